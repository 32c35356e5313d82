#!/usr/bin/env python
"""Build an experiment variant of the library (same two translation units as __graft_entry__.build) with extra hipcc
flags: python scripts/build_variant.py <name> [--tu1 "<flags>"] [--tu2 "<flags>"] [--both "<flags>"]
-> localrf_amd/csrc/liblrf_hip_<name>.so; select it with LRF_LIB=<path> (localrf_amd/_native.py)."""
import argparse
import os
import shlex
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "localrf_amd", "csrc")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("--tu1", default="")
    ap.add_argument("--tu2", default="")
    ap.add_argument("--both", default="")
    a = ap.parse_args()
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    base = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include")]
    objs, procs = [], []
    for tu, extra in ((1, a.tu1), (2, "-fno-slp-vectorize " + a.tu2)):
        obj = os.path.join(CSRC, f"var_{a.name}.tu{tu}.o")
        cmd = base + shlex.split(extra) + shlex.split(a.both) + [f"-DLRF_TU={tu}", "-c", "-o", obj, os.path.join(CSRC, "lrf_render.hip")]
        procs.append(subprocess.Popen(cmd))
        objs.append(obj)
    rc = [p.wait() for p in procs]
    if any(rc):
        sys.exit(f"compile failed {rc}")
    out = os.path.join(CSRC, f"liblrf_hip_{a.name}.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    for o in objs:
        os.remove(o)
    print(out)


if __name__ == "__main__":
    main()
