#!/usr/bin/env python
"""Per-kernel HIP-event times of the eval forward with and without LRF_FLAG_SORT_RAYS (the sort launch sits in the first interval)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from util import make_field, make_rays, quiet
for G, NS in ((300, 1536), (500, -1), (640, -1)):
    f = quiet(make_field, [G, G, G], "cpu", seed=0).to("cuda:0")
    rays = make_rays(4096, 1).cuda()
    z = f.z_schedule(False, NS, rays.device)
    for srt in (False, True, False, True):
        f.sort_rays = srt
        p = bench.kernel_profile(f, rays, z, reps=20)
        print(f"grid {G} S {z.numel()} sort={srt}: first interval (sort + k_march) {p['march_ms']*1e3:.1f} us, k_shade3 {p['shade_ms']*1e3:.1f} us, shaded {p['n_shaded']}", flush=True)
    del f
