#!/usr/bin/env python
"""Finding 17: the two sequences that tripped tests/test_gpu_parity.py::test_two_launch_sequence_equals_four_launch_sequence
in two full-suite runs, replayed with the workspace and the output block poisoned with NaN right before the render
under test.  (a) a render by the round-1 fused engine, then the two-launch default; (b) a freshly built and uploaded
64^3 empty field, first render.  NaN rays = stale memory was read; finite differences = arithmetic."""
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
NAN = float("nan")
n = int(os.environ.get("N", "60"))


def rend(field, pipe, engine="bf16x3", N_samples=1536, poison=False):
    lib.lrf_debug_set_shade_pipe(pipe)
    field.mlp_engine = engine
    try:
        with torch.no_grad():
            if poison and field._ws is not None:
                field._ws.view(torch.float32).fill_(NAN)
                tmp = [torch.full((4096, 3), NAN, device="cuda"), torch.full((4096,), NAN, device="cuda")]
                del tmp
            return field(rays, white_bg=True, is_train=False, N_samples=N_samples)[0]
    finally:
        lib.lrf_debug_set_shade_pipe(0)
        field.mlp_engine = "bf16x3"


def tally(tag, outs, ref):
    bad = nanr = 0
    worst, seen = 0.0, {}
    for out in outs:
        nan = torch.isnan(out).any(-1)
        d = torch.nan_to_num(out - ref, nan=0.0).abs().amax(-1)
        off = nan | (d > 5e-7)
        if bool(off.any()):
            bad += 1
            nanr += int(nan.sum())
            worst = max(worst, float(d.max()))
            for q in off.nonzero().flatten().tolist():
                seen[q] = seen.get(q, 0) + 1
    print(f"{tag}: {bad}/{len(outs)} renders off (> 5e-7 or NaN), NaN rays {nanr}, worst finite {worst:.2e}, rays {sorted(seen.items(), key=lambda kv: -kv[1])[:6]}", flush=True)


ref = rend(f, 9).clone()
for poison in (False, True):
    outs = []
    for it in range(n):
        rend(f, 0, engine="bf16x3_split")
        rend(f, 0, engine="bf16x3_fused")
        outs.append(rend(f, 0, poison=poison).clone())
    tally(f"(a) after the split + fused engines, poison={poison}", outs, ref)
for poison in (False, True):
    outs, refs = [], None
    for it in range(max(4, n // 6)):
        e = quiet(make_field, [64, 64, 64], "cpu", seed=3)
        with torch.no_grad():
            for p_ in e.density_plane:
                p_.zero_()
        e = e.to("cuda:0")
        e.density_shift = -30.0
        if poison:
            rend(e, 9, N_samples=192)                      # allocates the workspace so that it can be poisoned
        outs.append(rend(e, 0, N_samples=192, poison=poison).clone())
        refs = rend(e, 9, N_samples=192).clone()
    tally(f"(b) first render of a fresh empty field, poison={poison}", outs, refs)
