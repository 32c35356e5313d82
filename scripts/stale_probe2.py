#!/usr/bin/env python
"""Reproducer for finding 17: a render of the big batch that follows a render of ANOTHER field / batch shape sometimes
has a ray off by 1e-6..1e-5.  Alternates 'foreign' renders with renders of the big batch and counts, per launch
sequence / engine, the renders that differ from the reference."""
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
empty = quiet(make_field, [64, 64, 64], "cpu", seed=3).to("cuda:0")
n = int(os.environ.get("N", "400"))


def rend(field, r, pipe, engine="bf16x3", N_samples=1536):
    lib.lrf_debug_set_shade_pipe(pipe)
    field.mlp_engine = engine
    try:
        with torch.no_grad():
            return field(r, white_bg=True, is_train=False, N_samples=N_samples)[0]
    finally:
        lib.lrf_debug_set_shade_pipe(0)
        field.mlp_engine = "bf16x3"


def foreign(kind):
    if kind == "other field":
        rend(empty, rays, 0, N_samples=192)
    elif kind == "other engine":
        rend(f, rays, 0, engine="bf16x3_fused")
    elif kind == "other shape":
        rend(f, rays[:1000], 0)
    elif kind == "torch op":
        torch.randn(1 << 22, device="cuda").sort()
    elif kind == "same":
        pass


for pipe in (0, 9):
    ref = rend(f, rays, pipe).clone()
    for kind in ("same", "other field", "other engine", "other shape", "torch op"):
        bad, worst, seen = 0, 0.0, {}
        for it in range(n):
            foreign(kind)
            out = rend(f, rays, pipe)
            d = (out - ref).abs().amax(-1)
            if float(d.max()) > 0:
                bad += 1
                worst = max(worst, float(d.max()))
                for q in (d > 0).nonzero().flatten().tolist():
                    seen[q] = seen.get(q, 0) + 1
        print(f"pipe {pipe} after '{kind}': {bad}/{n} renders differ from the first, worst {worst:.2e}, distinct rays {len(seen)}, top {sorted(seen.items(), key=lambda kv: -kv[1])[:5]}", flush=True)
