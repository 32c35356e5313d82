"""Per-kernel register / scratch / LDS use of the HIP translation unit (from the code-object
metadata hipcc emits with -S).  usage: python scripts/kernel_resources.py [filter]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-I",
                      os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", "-",
                      os.path.join(ROOT, "localrf_amd", "csrc", "lrf_render.hip")],
                     capture_output=True, text=True).stdout
flt = sys.argv[1] if len(sys.argv) > 1 else ""
meta = asm[asm.index("amdhsa.kernels:"):]
for blk in re.split(r"\n  - \.agpr_count:", meta)[1:]:
    get = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk)
    name = re.search(r"\n    \.name:\s+(\S+)", blk)[1]
    if flt not in name:
        continue
    short = re.sub(r"^_ZN3lrf\d+", "", name)[:28]
    print(f"{short:30s} vgpr {get('vgpr_count')[1]:>4s} agpr {blk.split()[0]:>3s} sgpr {get('sgpr_count')[1]:>4s} "
          f"scratch {get('private_segment_fixed_size')[1]:>5s} lds {get('group_segment_fixed_size')[1]:>6s} "
          f"spill_v {get('vgpr_spill_count')[1]:>4s}")
