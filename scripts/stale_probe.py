#!/usr/bin/env python
"""Does the two-launch forward ever hand back a value it did not compute in THIS render?  The workspace and the block the
output will be allocated from are filled with NaN before every render; any NaN (or any difference from the four-launch
reference) in the result is a read of stale memory.  Prints the offending rays.  (DESIGN.md finding 17.)"""
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


def render(pipe, poison):
    lib.lrf_debug_set_shade_pipe(pipe)
    try:
        with torch.no_grad():
            if poison and f._ws is not None:
                f._ws.view(torch.float32).fill_(NAN)
                tmp = [torch.full((4096, 3), NAN, device="cuda"), torch.full((4096,), NAN, device="cuda")]
                del tmp
            return f(rays, white_bg=True, is_train=False, N_samples=1536)
    finally:
        lib.lrf_debug_set_shade_pipe(0)




def case(field, r, white_bg=True, N_samples=1536, n=20, tag=""):
    def rend(pipe, poison):
        lib.lrf_debug_set_shade_pipe(pipe)
        try:
            with torch.no_grad():
                if poison and field._ws is not None:
                    field._ws.view(torch.float32).fill_(NAN)
                    tmp = [torch.full((r.shape[0], 3), NAN, device="cuda"), torch.full((r.shape[0],), NAN, device="cuda")]
                    del tmp
                return field(r, white_bg=white_bg, is_train=False, N_samples=N_samples)
        finally:
            lib.lrf_debug_set_shade_pipe(0)
    ref = rend(9, False)[0].clone()
    for pipe in (0, 9):
        bad, seen, worst, nans = 0, {}, 0.0, 0
        for it in range(n):
            out = rend(pipe, True)[0]
            nan = torch.isnan(out).any(-1)
            d = torch.nan_to_num(out - ref, nan=0.0).abs().amax(-1)
            off = nan | (d > 5e-7)
            if bool(off.any()):
                bad += 1
                nans += int(nan.sum())
                worst = max(worst, float(d.max()))
                for q in off.nonzero().flatten().tolist():
                    seen[q] = seen.get(q, 0) + 1
        print(f"{tag:28s} R {r.shape[0]:6d} pipe {pipe}: {bad}/{n} renders off, NaN rays {nans}, worst finite diff {worst:.2e}, rays {sorted(seen.items(), key=lambda kv: -kv[1])[:6]}", flush=True)


case(f, rays, tag="big")
case(f, rays, white_bg=False, tag="big black bg")
for R in (1, 63, 1000, 4095):
    case(f, rays[:R], tag="ragged")
case(f, rays, N_samples=1032, tag="S=344")
many = make_rays(20000, 5).cuda()
case(f, many, tag="20000 rays (four launches)")
case(f, many[:12000], tag="12000 rays")
case(f, many[:5000], N_samples=6144, tag="S=2048")
far = rays.clone()
far[::3, :3] = 50.0
far[::3, 3:] = torch.tensor([1.0, 0.2, 0.1], device="cuda")
case(f, far, tag="far rays")
empty = quiet(make_field, [64, 64, 64], "cpu", seed=3)
with torch.no_grad():
    for p_ in empty.density_plane:
        p_.zero_()
empty = empty.to("cuda:0")
empty.density_shift = -30.0
case(empty, rays, N_samples=192, tag="empty field")
# and once more in the test's own order without poison but with a foreign previous render in between
case(f, rays, tag="big again")
