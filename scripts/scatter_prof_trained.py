#!/usr/bin/env python
"""Phase profile of the scatter kernels inside the captured progressive loop (trained field, sparse tiles): an LRF_SCATTER_PROF build
(scripts/build_variant.sh prof -DLRF_SCATTER_PROF=1), scripts/train_synth.run for --max-iters iterations, then the counters of the
LAST launches.  LRF_LIB=localrf_amd/csrc/liblrf_prof.so python scripts/scatter_prof_trained.py [--final 300] [--max-iters 1800]"""
import argparse, ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import torch
from localrf_amd import _native as N
import train_synth
ap = argparse.ArgumentParser()
ap.add_argument("--final", type=int, default=300)
ap.add_argument("--max-iters", type=int, default=1800)
a = ap.parse_args()
N.lib()
out = train_synth.run(frames=14, final=a.final, iters_per_frame=300, n_max_frames=8, max_iters=a.max_iters, dev="cuda:0", graph=True)
torch.cuda.synchronize()
print("resolution", out["final_resolution"], "ms/iter", out["ms_per_iteration_by_resolution"])
raw = ctypes.CDLL(N.LIB_PATH)
buf = np.zeros((2, 2048, 12), dtype=np.uint64)
assert raw.lrf_debug_scatter_prof(buf.ctypes.data_as(ctypes.c_void_p)) == 0
names = ["zero tile", "loads until ready", "-", "-", "run sums + adds", "barrier after entries", "tile flush", "line flush"]
for k, kind in enumerate(("density", "appearance")):
    d = buf[k].astype(np.float64)
    d = d[d[:, 11] > 0]
    if not len(d):
        continue
    tot = d[:, 11]
    print(f"== {kind}: {len(d)} workgroups, cycles of wave 0: total mean {tot.mean():.0f} min {tot.min():.0f} max {tot.max():.0f}; tile visits mean {d[:, 8].mean():.1f} max {d[:, 8].max():.0f}; entries mean {d[:, 9].mean():.0f}; steps of wave 0 mean {d[:, 10].mean():.1f}")
    for i in (0, 1, 4, 5, 6, 7):
        print(f"   {names[i]:24s} mean {d[:, i].mean():9.0f} ({100 * d[:, i].mean() / tot.mean():5.1f} %)  max {d[:, i].max():9.0f}")
