#!/usr/bin/env python
"""configs[2] and a full image of the same 4-field scene: lrf_scene_fwd with its fused multi-field launches against the
field-by-field form (lrf_debug_set_scene_fuse), same box, interleaved."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ge.build()
import torch
import bench
from localrf_amd import _native as N
dev = torch.device("cuda:0")
lt, ray_ids, view_ids, bw = bench.config3_scene(dev)
lib = N.lib()
g = torch.Generator().manual_seed(3)
cases = {"4096 rays, one chunk": (ray_ids, view_ids, bw, 4096),
         "65536 rays (4 views x 16384), chunks of 16384": (torch.randint(0, 64 * 48, (65536,), generator=g), view_ids, bw, 4 * 16384)}
def sync(): torch.cuda.synchronize()
for name, (rid, vid, b, chunk) in cases.items():
    lt.min_chunk = 1
    res = {}
    for rnd in range(3):
        for fuse in (1, 0):
            lib.lrf_debug_set_scene_fuse(fuse)
            with torch.no_grad():
                f = lambda: lt(rid, vid, 64, 48, is_train=False, blending_weights=b, chunk=chunk)
                res.setdefault(fuse, []).append(bench.timed(f, 20, 3, sync) / 20 * 1e3)
    lib.lrf_debug_set_scene_fuse(1)
    print(f"{name}: fused {min(res[1]):.4f} ms ({[round(x, 4) for x in res[1]]}), field by field {min(res[0]):.4f} ms ({[round(x, 4) for x in res[0]]})")
