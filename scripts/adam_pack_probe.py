#!/usr/bin/env python
"""Optimiser step + layout refresh, two passes (lrf_adam_step, lrf_pack_field) against the fused one (lrf_adam_step_pack):
HIP-event time per iteration at several grid sizes.  python scripts/adam_pack_probe.py [--grids 300,500]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from localrf_amd import FusedAdam
from util import make_field, quiet
ap = argparse.ArgumentParser()
ap.add_argument("--grids", default="64,300,500")
a = ap.parse_args()
for g in [int(x) for x in a.grids.split(",")]:
    f = quiet(make_field, [g] * 3, "cpu", seed=0).to("cuda:0")
    for p in f.parameters():
        p.grad = torch.randn_like(p) * 0.01
    res = {}
    for mode in ("two passes", "fused"):
        opt = FusedAdam(f.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), pack_field=f if mode == "fused" else None)
        f._ensure_cache()
        for _ in range(5):
            opt.step(); f._ensure_cache()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(50):
            opt.step(); f._ensure_cache()
        e1.record(); torch.cuda.synchronize()
        res[mode] = e0.elapsed_time(e1) / 50 * 1e3
    print(f"grid {g}^3: Adam + layout refresh {res['two passes']:.1f} us in two passes, {res['fused']:.1f} us fused", flush=True)
