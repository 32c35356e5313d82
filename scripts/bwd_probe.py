#!/usr/bin/env python
"""Forward + backward at BASELINE configs[1] (or --grid N) under several lrf_debug_set_train_fwd_engine settings: gradient
errors against the reference-recorded gradients of tests/golden/field_small_train_grad.npz, then wall time per step.
python scripts/bwd_probe.py --eng 1,17 [--grid 300] [--samples 1536] [--steps 40]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from localrf_amd import _native as N
from oracle import vm_render_np as oracle
from util import field_from_golden, load_golden, make_field, make_rays, quiet
ap = argparse.ArgumentParser()
ap.add_argument("--eng", default="1")
ap.add_argument("--grid", type=int, default=300)
ap.add_argument("--samples", type=int, default=1536)
ap.add_argument("--rays", type=int, default=4096)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--overlap", type=int, default=None, help="lrf_debug_set_bwd_overlap argument (0: the backward's two branches on one stream)")
a = ap.parse_args()
lib = N.lib()
if a.overlap is not None:
    lib.lrf_debug_set_bwd_overlap(a.overlap)
engs = [int(x) for x in a.eng.split(",")]
g = load_golden("field_small_train_grad")
rel = lambda x, y: float(np.abs(x - y).max() / max(np.abs(y).max(), 1e-12))
for eng in engs:
    lib.lrf_debug_set_train_fwd_engine(eng)
    f = quiet(field_from_golden, g, "cuda:0")
    f.z_override = torch.from_numpy(oracle.z_schedule(int(g["N_samples"]), np.float32, jitter=(g["U"], g["U2"])))
    rays = torch.from_numpy(g["rays"]).cuda().requires_grad_(True)
    rgb, depth = f(rays, white_bg=True, is_train=True, N_samples=int(g["N_samples"]))
    ((rgb * torch.from_numpy(g["g_rgb"]).cuda()).sum() + (depth * torch.from_numpy(g["g_depth"]).cuda()).sum()).backward()
    errs = {n: rel(p.grad.cpu().numpy(), g["grad." + n]) for n, p in f.named_parameters() if p.requires_grad}
    errs["rays"] = rel(rays.grad.cpu().numpy(), g["grad.rays"])
    print(f"eng {eng}: worst grad rel-to-max error {max(errs.values()):.2e}", {k: f"{v:.1e}" for k, v in errs.items() if "plane" in k or "line" in k}, flush=True)
f = quiet(make_field, [a.grid] * 3, "cpu", seed=0).to("cuda:0")
rays = make_rays(a.rays, 1).cuda()
gen = torch.Generator().manual_seed(3)
gr, gd = torch.randn(a.rays, 3, generator=gen).cuda(), torch.randn(a.rays, generator=gen).cuda()


def step():
    for p in f.parameters():
        p.grad = None
    rgb, depth = f(rays, white_bg=True, is_train=False, N_samples=a.samples)
    ((rgb * gr).sum() + (depth * gd).sum()).backward()


ref = None
for rnd in range(a.rounds):
    for eng in engs:
        lib.lrf_debug_set_train_fwd_engine(eng)
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize()
        dt = (time.time() - t0) / a.steps * 1e3
        gs = {n: p.grad.double().cpu().numpy() for n, p in f.named_parameters() if p.grad is not None}
        if ref is None:
            ref = gs
        worst = max(rel(gs[k], ref[k]) for k in ref)
        print(f"round {rnd} eng {eng}: grid {a.grid} fwd+bwd {dt:.3f} ms | worst rel-to-max difference from the first setting's gradients {worst:.2e}", flush=True)
lib.lrf_debug_set_train_fwd_engine(1)
