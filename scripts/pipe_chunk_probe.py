#!/usr/bin/env python
"""lrf_render_fwd's two-stream large-batch mode: a 65536-ray (or --rays) eval forward of BASELINE configs[1]'s field with
chunks of 0 (one pass, rounds 1-5) / 4096 / 8192 / 16384 rays -- time per call (median of 9) and the outputs against the
one-pass result.  python scripts/pipe_chunk_probe.py [--grid 300] [--rays 65536] [--chunks 0,4096,8192]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from localrf_amd import _native as N
from util import make_field, make_rays, quiet
ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=300)
ap.add_argument("--samples", type=int, default=512)
ap.add_argument("--rays", type=int, default=65536)
ap.add_argument("--chunks", default="0,4096,8192,16384")
a = ap.parse_args()
lib = N.lib()
f = quiet(make_field, [a.grid] * 3, "cpu", seed=0).to("cuda:0")
rays = torch.cat([make_rays(4096, 1 + i) for i in range((a.rays + 4095) // 4096)], 0)[:a.rays].cuda()
ref = None
for ch in [int(x) for x in a.chunks.split(",")]:
    lib.lrf_debug_set_pipe_chunk(ch)
    f._ws = None                                   # (the workspace size depends on the chunking)
    with torch.no_grad():
        for _ in range(3):
            rgb, depth = f(rays, white_bg=True, is_train=False, N_samples=a.samples)
        per = []
        for _ in range(9):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            rgb, depth = f(rays, white_bg=True, is_train=False, N_samples=a.samples)
            torch.cuda.synchronize(); per.append(time.perf_counter() - t0)
    per.sort()
    if ref is None:
        ref = (rgb.clone(), depth.clone())
    same = bool(torch.equal(rgb, ref[0]) and torch.equal(depth, ref[1]))
    print(f"chunk {ch:6d}: {per[4] * 1e3:.3f} ms per {a.rays}-ray call (min {per[0] * 1e3:.3f}) = {a.rays / per[4] / 1e6:.2f} M rays/s | bit-identical to the first setting: {same}", flush=True)
lib.lrf_debug_set_pipe_chunk(16384)
