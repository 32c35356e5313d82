#!/usr/bin/env python
"""Host time to ENQUEUE one training step of a field (forward with a graph, backward, FusedAdam) against its GPU time; where the
host time goes (cProfile, by own time)."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
ge.build()
from localrf_amd import FusedAdam
from util import make_field, make_rays, quiet
f = quiet(make_field, [300, 300, 300], "cpu", seed=0).to("cuda:0")
rays = make_rays(4096, 1).cuda()
g = torch.Generator().manual_seed(3)
gr, gd = torch.randn(4096, 3, generator=g).cuda(), torch.randn(4096, generator=g).cuda()
opt = FusedAdam(f.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
def step():
    opt.zero_grad()
    rgb, depth = f(rays, white_bg=True, is_train=True, N_samples=1536)
    ((rgb * gr).sum() + (depth * gd).sum()).backward()
    opt.step()
for _ in range(20):
    step()
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(100):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"host enqueue {1e3 * (t1 - t0) / 100:.4f} ms per step; with the final sync {1e3 * (t2 - t0) / 100:.4f} ms per step", flush=True)
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
