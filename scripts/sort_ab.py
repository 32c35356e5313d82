#!/usr/bin/env python
"""A/B of LRF_FLAG_SORT_RAYS at BASELINE configs[1] (and 500^3 with GRID=500): eval forward and forward+backward, interleaved."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
ge.build()
from util import make_field, make_rays, quiet
G = int(os.environ.get("GRID", "300"))
NS = int(os.environ.get("NS", "1536"))
f = quiet(make_field, [G, G, G], "cpu", seed=0).to("cuda:0")
rays = make_rays(4096, 1).cuda()
g = torch.Generator().manual_seed(3)
gr, gd = torch.randn(4096, 3, generator=g).cuda(), torch.randn(4096, generator=g).cuda()

def fwd(n):
    with torch.no_grad():
        for _ in range(n):
            f(rays, white_bg=True, is_train=False, N_samples=NS)

def fb(n):
    for _ in range(n):
        for p in f.parameters():
            p.grad = None
        rgb, depth = f(rays, white_bg=True, is_train=False, N_samples=NS)
        ((rgb * gr).sum() + (depth * gd).sum()).backward()

def timed(fn, n):
    fn(5); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(n); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for rep in range(3):
    for srt in (False, True):
        f.sort_rays = srt
        print(f"grid {G} sort={srt}: forward {timed(fwd, 100):.4f} ms   forward+backward {timed(fb, 30):.3f} ms", flush=True)
