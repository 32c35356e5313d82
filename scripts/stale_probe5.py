#!/usr/bin/env python
"""Finding 17: does the first render after the GPU has been IDLE (host sleeps, clocks fall back) differ?  The pytest runs
that tripped had host-side work (golden comparisons in numpy) right before the render that came back different."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
ge.build()
from localrf_amd import _native as N
from util import make_field, make_rays, quiet
lib = N.lib()
f = quiet(make_field, [300, 300, 300], "cpu", seed=0).to("cuda:0")
rays = make_rays(4096, 1).cuda()
n = int(os.environ.get("N", "40"))
idle = float(os.environ.get("IDLE", "0.2"))


def rend(pipe, engine):
    lib.lrf_debug_set_shade_pipe(pipe)
    f.mlp_engine = engine
    try:
        with torch.no_grad():
            return f(rays, white_bg=True, is_train=False, N_samples=1536)
    finally:
        lib.lrf_debug_set_shade_pipe(0)
        f.mlp_engine = "bf16x3"


for pipe, engine in ((0, "bf16x3"), (0, "f32"), (0, "valu"), (9, "bf16x3")):
    for _ in range(5):
        ref, dref = rend(pipe, engine)
    ref, dref = ref.clone(), dref.clone()
    torch.cuda.synchronize()
    bad, worst, seen, dbad = 0, 0.0, {}, 0
    for it in range(n):
        time.sleep(idle)
        out, dep = rend(pipe, engine)
        d = (out - ref).abs().amax(-1)
        dbad += int(not torch.equal(dep, dref))
        if float(d.max()) > 0:
            bad += 1
            worst = max(worst, float(d.max()))
            for q in (d > 0).nonzero().flatten().tolist():
                seen[q] = seen.get(q, 0) + 1
    print(f"pipe {pipe} {engine:8s} first render after {idle:.2f} s idle: {bad}/{n} differ, worst {worst:.2e}, distinct rays {len(seen)}, depth differs in {dbad}", flush=True)
