#!/usr/bin/env python
"""Eager loop vs captured iteration of the progressive driver on the same schedule: ms per iteration by resolution, the
capture statistics, and the first losses side by side.  gpurun -- 'python scripts/graph_train_probe.py [--final 200]'"""
import argparse, json, os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import __graft_entry__ as ge
ge.build()
import train_synth
ap = argparse.ArgumentParser()
ap.add_argument("--final", type=int, default=160)
ap.add_argument("--frames", type=int, default=10)
ap.add_argument("--iters-per-frame", type=int, default=120)
ap.add_argument("--n-max-frames", type=int, default=6)
ap.add_argument("--out", default=None)
a = ap.parse_args()
res = {}
for mode in ("graph", "eager"):
    try:
        t0 = time.perf_counter()
        out = train_synth.run(frames=a.frames, final=a.final, iters_per_frame=a.iters_per_frame, n_max_frames=a.n_max_frames,
                              dev="cuda:0", graph=(mode == "graph"), log=None)
        res[mode] = {k: out[k] for k in ("iterations", "ms_per_iteration_by_resolution", "iterations_by_resolution", "graph", "loss_first",
                                         "loss_last", "fields", "frames", "events", "peak_memory_GB", "param_checksum")}
        res[mode]["wall_s"] = time.perf_counter() - t0
    except Exception:
        res[mode] = {"error": traceback.format_exc()[-3000:]}
    print(mode, json.dumps(res[mode])[:3000], flush=True)
if a.out:
    json.dump(res, open(a.out, "w"), indent=1)
