#!/usr/bin/env python
"""30 training steps at configs[1] with the backward's two branches on ONE stream (lrf_debug_set_bwd_overlap(0)): run under
rocprofv3 --kernel-trace to read every backward kernel's uncontended duration."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
ge.build()
from localrf_amd import _native as N
from util import make_field, make_rays, quiet
f = quiet(make_field, [300, 300, 300], "cpu", seed=0).to("cuda:0")
rays = make_rays(4096, 1).cuda()
g = torch.Generator().manual_seed(3)
gr, gd = torch.randn(4096, 3, generator=g).cuda(), torch.randn(4096, generator=g).cuda()
N.lib().lrf_debug_set_bwd_overlap(int(os.environ.get("OVERLAP", "0")))
N.lib().lrf_debug_set_train_fwd_engine(int(os.environ.get("TRAIN_ENG", "1"), 0))      # 1 = defaults; | 8 | 16: separate line-scatter kernels
for _ in range(30):
    for p in f.parameters():
        p.grad = None
    rgb, depth = f(rays, white_bg=True, is_train=False, N_samples=1536)
    ((rgb * gr).sum() + (depth * gd).sum()).backward()
torch.cuda.synchronize()
import time
N.lib().lrf_debug_set_bwd_overlap(int(os.environ.get("OVL2", "7")))
for _ in range(5):
    rgb, depth = f(rays, white_bg=True, is_train=False, N_samples=1536); ((rgb * gr).sum() + (depth * gd).sum()).backward()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(40):
    for p in f.parameters():
        p.grad = None
    rgb, depth = f(rays, white_bg=True, is_train=False, N_samples=1536); ((rgb * gr).sum() + (depth * gd).sum()).backward()
torch.cuda.synchronize()
print("fwd+bwd with the default overlap: %.3f ms" % ((time.time() - t0) / 40 * 1e3))
