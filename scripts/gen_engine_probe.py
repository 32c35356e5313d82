#!/usr/bin/env python
"""How slow is the generic colour-network engine (csrc/lrf_generic.inl)?  BASELINE configs[1] (300^3, 4096 rays x 512 samples)
with view_pe / fea_pe / featureC off opt.py's defaults: eval forward and forward+backward, beside the default configuration."""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
ge.build()
from util import make_field, make_rays, quiet
rays = make_rays(4096, 1).cuda()
g = torch.Generator().manual_seed(3)
gr, gd = torch.randn(4096, 3, generator=g).cuda(), torch.randn(4096, generator=g).cuda()
for cfg in (dict(), dict(view_pe=2), dict(fea_pe=2, view_pe=2), dict(fea_pe=6, view_pe=6, featureC=256)):
    f = quiet(make_field, [300, 300, 300], "cpu", seed=0, **cfg).to("cuda:0")
    def fwd():
        with torch.no_grad():
            f(rays, white_bg=True, is_train=False, N_samples=1536)
    def fb():
        for p in f.parameters():
            p.grad = None
        rgb, depth = f(rays, white_bg=True, is_train=False, N_samples=1536)
        ((rgb * gr).sum() + (depth * gd).sum()).backward()
    out = {}
    for name, fn, n in (("forward", fwd, 10), ("forward+backward", fb, 5)):
        fn(); fn(); torch.cuda.synchronize(); t0 = time.time()
        for _ in range(n):
            fn()
        torch.cuda.synchronize(); out[name] = (time.time() - t0) / n * 1e3
    print(cfg or "default (0 / 0 / 128)", {k: "%.2f ms" % v for k, v in out.items()})
    del f
