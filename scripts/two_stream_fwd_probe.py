#!/usr/bin/env python
"""Eval forward of BASELINE configs[1] issued on ONE stream against the same number of batches alternating over TWO streams
(one field: TensorVMSplit keeps a workspace per stream): does k_march of one batch run beside k_shade3 of another?
python scripts/two_stream_fwd_probe.py [--grid 300] [--batches 200]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from util import make_field, make_rays, quiet
ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, default=300)
ap.add_argument("--samples", type=int, default=512)
ap.add_argument("--rays", type=int, default=4096)
ap.add_argument("--batches", type=int, default=200)
a = ap.parse_args()
f0 = quiet(make_field, [a.grid] * 3, "cpu", seed=0).to("cuda:0")
fs = [f0, f0]
rays = [make_rays(a.rays, 1 + i).cuda() for i in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]


def run(nstreams):
    with torch.no_grad():
        for k in range(20 + a.batches):
            if k == 20:
                torch.cuda.synchronize(); t0 = time.perf_counter()
            i = k % nstreams
            with torch.cuda.stream(streams[i]):
                fs[i](rays[i], white_bg=True, is_train=False, N_samples=a.samples)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / a.batches * 1e3


for rnd in range(3):
    for ns in (1, 2):
        ms = run(ns)
        print(f"round {rnd}: {ns} stream(s): {ms:.4f} ms per {a.rays}-ray batch = {a.rays / ms / 1e3:.2f} M rays/s", flush=True)
