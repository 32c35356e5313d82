#!/usr/bin/env python
"""Where the gradient scatter kernels spend their time: an LRF_SCATTER_PROF build (scripts/build_variant.sh prof
-DLRF_SCATTER_PROF=1) accumulates s_memtime per phase in wave 0 of every workgroup of k_scatter_plane; this probe runs a few
training steps at BASELINE configs[1] (or --grid N) and prints the per-phase totals over workgroups (mean / max).
LRF_LIB=localrf_amd/csrc/liblrf_prof.so python scripts/scatter_prof_probe.py [--grid 300] [--samples 1536]"""
import argparse, ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from localrf_amd import _native as N
from util import make_field, make_rays, quiet
ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=300)
ap.add_argument("--samples", type=int, default=1536)
ap.add_argument("--rays", type=int, default=4096)
a = ap.parse_args()
lib = N.lib()
f = quiet(make_field, [a.grid] * 3, "cpu", seed=0).to("cuda:0")
rays = make_rays(a.rays, 1).cuda()
g = torch.Generator().manual_seed(3)
gr, gd = torch.randn(a.rays, 3, generator=g).cuda(), torch.randn(a.rays, generator=g).cuda()
N.lib().lrf_debug_set_bwd_overlap(0)                      # one stream: uncontended phases
for _ in range(4):
    for p in f.parameters():
        p.grad = None
    rgb, depth = f(rays, white_bg=True, is_train=True, N_samples=a.samples)
    ((rgb * gr).sum() + (depth * gd).sum()).backward()
torch.cuda.synchronize()
raw = ctypes.CDLL(N.LIB_PATH)
buf = np.zeros((2, 2048, 12), dtype=np.uint64)
rc = raw.lrf_debug_scatter_prof(buf.ctypes.data_as(ctypes.c_void_p))
assert rc == 0, rc
names = ["zero tile", "phase A (entry -> taps)", "B: loads until ready", "B: math", "B: shuffles + LDS adds", "barrier after entries",
         "tile flush (global atomics)", "line flush", "tiles", "entries", "64-entry steps of wave 0", "total"]
for k, kind in enumerate(("density", "appearance")):
    d = buf[k].astype(np.float64)
    d = d[d[:, 11] > 0]
    if not len(d):
        continue
    print(f"== {kind} scatter: {len(d)} workgroups, cycles of wave 0 (s_memtime)")
    tot = d[:, 11]
    print(f"   total per workgroup: mean {tot.mean():.0f}  min {tot.min():.0f}  max {tot.max():.0f}")
    for i in range(8):
        print(f"   {names[i]:32s} mean {d[:, i].mean():10.0f} ({100 * d[:, i].mean() / tot.mean():5.1f} %)  max {d[:, i].max():10.0f}")
    for i in (8, 9, 10):
        print(f"   {names[i]:32s} mean {d[:, i].mean():10.1f}  min {d[:, i].min():8.0f}  max {d[:, i].max():10.0f}")
    steps = d[:, 10].sum()
    print(f"   per 64-entry step of a wave: A {d[:, 1].sum() / steps:.0f}  loads {d[:, 2].sum() / steps:.0f}  math {d[:, 3].sum() / steps:.0f}  adds {d[:, 4].sum() / steps:.0f} cycles")
