#!/usr/bin/env python
"""What bounds the captured iteration: the graph's own execution time on the GPU (replays back to back, no host work in
between), against the host's share of an iteration (everything of the loop except the replay), per grid size."""
import argparse, cProfile, io, json, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import __graft_entry__ as ge
ge.build()
import torch
import train_synth
ap = argparse.ArgumentParser()
ap.add_argument("--final", type=int, default=300)
ap.add_argument("--max-iters", type=int, default=2600)
ap.add_argument("--profile", action="store_true")
a = ap.parse_args()
keep = {}
pr = cProfile.Profile() if a.profile else None
if pr: pr.enable()
out = train_synth.run(frames=14, final=a.final, iters_per_frame=300, n_max_frames=8, max_iters=a.max_iters, dev="cuda:0", graph=True, live=keep)
if pr: pr.disable()
gs = keep["captured"]
print("ms/iter by res", out["ms_per_iteration_by_resolution"], out["graph"], "res", out["final_resolution"])
if gs._graphs is not None:
    g = gs._graphs[0]
    for _ in range(20): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(300): g.replay()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 300 * 1e3
    print(f"graph replay back to back: {dt:.4f} ms per iteration at {out['final_resolution']}^3, nodes unknown")
    # sustained: blocks of 500 replays, seconds apart from the start -- does the chip hold its clock when it is never idle?
    for blk in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(500): g.replay()
        torch.cuda.synchronize()
        print(f"  sustained block {blk}: {(time.perf_counter() - t0) / 500 * 1e3:.4f} ms per replay", flush=True)
    # the same work launched eagerly, for the GPU time of the kernels alone (host-bound or not, the stream drains at the end)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(100): g.replay()
    ev1.record(); torch.cuda.synchronize()
    print(f"  HIP-event time per replay: {ev0.elapsed_time(ev1) / 100:.4f} ms")
if pr:
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
    print(s.getvalue()[:7000])
