#!/usr/bin/env python
"""Host-side profile of the progressive training driver: where does an iteration's wall time go on the CPU?"""
import cProfile, pstats, os, sys, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import __graft_entry__ as ge
ge.build()
import train_synth
pr = cProfile.Profile()
pr.enable()
out = train_synth.run(frames=12, final=300, iters_per_frame=60, max_iters=900, dev="cuda:0", log=lambda m: None)
pr.disable()
print(out["ms_per_iteration_by_resolution"], out["iterations"])
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
print(s.getvalue()[:9000])
