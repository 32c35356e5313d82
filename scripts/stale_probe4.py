#!/usr/bin/env python
"""Finding 17: which colour engine shows the non-repeating ray differences when an unrelated, cache-flushing torch kernel
(a 4M-element sort) runs between renders of the same batch?"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
ge.build()
from localrf_amd import _native as N
from util import make_field, make_rays, quiet
lib = N.lib()
f = quiet(make_field, [300, 300, 300], "cpu", seed=0).to("cuda:0")
rays = make_rays(4096, 1).cuda()
n = int(os.environ.get("N", "300"))
big = torch.randn(1 << 22, device="cuda")
flush = torch.empty(96 << 20, device="cuda")          # 384 MB of fp32: larger than L2 + MALL


def rend(pipe, engine):
    lib.lrf_debug_set_shade_pipe(pipe)
    f.mlp_engine = engine
    try:
        with torch.no_grad():
            return f(rays, white_bg=True, is_train=False, N_samples=1536)
    finally:
        lib.lrf_debug_set_shade_pipe(0)
        f.mlp_engine = "bf16x3"


for pipe, engine in ((0, "bf16x3"), (9, "bf16x3"), (0, "bf16x3_fused"), (0, "bf16x3_split"), (0, "f32"), (0, "valu")):
    for _ in range(3):
        ref, dref = rend(pipe, engine)
    ref, dref = ref.clone(), dref.clone()
    for kind in os.environ.get("KINDS", "sort,fill 384 MB,sleep").split(","):
        bad, worst, seen, dbad = 0, 0.0, {}, 0
        for it in range(n):
            if kind == "sort":
                big.sort()
            elif kind == "fill 384 MB":
                flush.fill_(1.0)
            else:
                torch.cuda._sleep(2_000_000)
            out, dep = rend(pipe, engine)
            d = (out - ref).abs().amax(-1)
            dbad += int(not torch.equal(dep, dref))
            if float(d.max()) > 0:
                bad += 1
                worst = max(worst, float(d.max()))
                for q in (d > 0).nonzero().flatten().tolist():
                    seen[q] = seen.get(q, 0) + 1
        print(f"pipe {pipe} {engine:13s} after {kind:12s}: {bad}/{n} renders differ, worst {worst:.2e}, distinct rays {len(seen)}, depth differs in {dbad}", flush=True)
