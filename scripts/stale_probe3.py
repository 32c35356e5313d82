#!/usr/bin/env python
"""Finding 17, by poisoning: before every render, every CU's LDS (and optionally a wave's vector registers) is filled
with a pattern (lrf_debug_poison_cu_state).  A NaN pattern turns any read of never-written LDS / registers into NaN
rays; a zero pattern is the control."""
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
n = int(os.environ.get("N", "200"))
st = torch.cuda.current_stream().cuda_stream


def rend(pipe, engine):
    lib.lrf_debug_set_shade_pipe(pipe)
    f.mlp_engine = engine
    try:
        with torch.no_grad():
            return f(rays, white_bg=True, is_train=False, N_samples=1536)[0]
    finally:
        lib.lrf_debug_set_shade_pipe(0)
        f.mlp_engine = "bf16x3"


for pipe, engine in ((0, "bf16x3"), (9, "bf16x3"), (0, "bf16x3_fused"), (0, "bf16x3_split"), (0, "f32")):
    ref = rend(pipe, engine).clone()
    for r in range(3):
        ref2 = rend(pipe, engine)
    ref = ref2.clone()
    for pattern, regs, name in ((0x7fc00000, 0, "LDS=NaN"), (0x7fc00000, 1, "LDS+regs=NaN"), (0, 1, "LDS+regs=0"), (0x3f800000, 1, "LDS+regs=1.0")):
        bad, nanr, worst, seen = 0, 0, 0.0, {}
        for it in range(n):
            N.check(lib.lrf_debug_poison_cu_state(pattern, regs, st), "poison")
            out = rend(pipe, engine)
            nan = torch.isnan(out).any(-1)
            d = torch.nan_to_num(out - ref, nan=0.0).abs().amax(-1)
            off = nan | (d > 0)
            if bool(off.any()):
                bad += 1
                nanr += int(nan.sum())
                worst = max(worst, float(d.max()))
                for q in off.nonzero().flatten().tolist():
                    seen[q] = seen.get(q, 0) + 1
        print(f"pipe {pipe} {engine:13s} {name:13s}: {bad}/{n} renders off, NaN rays {nanr}, worst finite {worst:.2e}, distinct rays {len(seen)}, top {sorted(seen.items(), key=lambda kv: -kv[1])[:4]}", flush=True)
