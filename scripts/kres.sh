#!/bin/bash
# register / scratch / instruction counts of the kernels whose names match $1 (regex), compiled as __graft_entry__.build() does; extra flags follow
PAT=$1; shift
cd "$(dirname "$0")/../localrf_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize "$@" -I ../../include -S --cuda-device-only -o /tmp/hd.s lrf_render.hip 2>&1 | grep -E "error" 
python - "$PAT" <<'PY'
import re, sys
asm=open('/tmp/hd.s').read()
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel", asm, re.S|re.M):
    if re.search(sys.argv[1], m[1]):
        b=m[2]
        print(m[1][:64], 'vgpr', re.findall(r'\.amdhsa_next_free_vgpr (\d+)',b), 'agpr_off', re.findall(r'\.amdhsa_accum_offset (\d+)',b), 'scratch', re.findall(r'\.amdhsa_private_segment_fixed_size (\d+)',b),
              'ds_add_u64', len(re.findall(r'ds_add_u64',b)), 'bperm', len(re.findall('ds_bpermute',b)), 'gload', len(re.findall('global_load',b)), 'valu', len(re.findall(r'^\s+v_', b, re.M)))
PY
