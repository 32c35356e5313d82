#!/usr/bin/env python
"""Host time to ENQUEUE one eval forward (TensorVMSplit.forward -> lrf_render_fwd) against the GPU time of the step: if the first
is not well below the second, the headline loop is host-bound on that box."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from util import make_field, make_rays, quiet
f = quiet(make_field, [300, 300, 300], "cpu", seed=0).to("cuda:0")
rays = make_rays(4096, 1).cuda()
with torch.no_grad():
    for _ in range(50):
        f(rays, white_bg=True, is_train=False, N_samples=1536)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(200):
            f(rays, white_bg=True, is_train=False, N_samples=1536)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"host enqueue {1e3 * (t1 - t0) / 200:.4f} ms per call; with the final sync {1e3 * (t2 - t0) / 200:.4f} ms per step", flush=True)
import cProfile, pstats
pr = cProfile.Profile()
with torch.no_grad():
    pr.enable()
    for _ in range(300):
        f(rays, white_bg=True, is_train=False, N_samples=1536)
    pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
