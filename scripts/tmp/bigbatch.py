import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import make_field, make_rays, quiet
f = quiet(make_field, [300]*3, "cpu", seed=0).to("cuda:0")
for R in (4096, 8192, 15000, 16384, 32768, 65536, 262144):
    rays = make_rays(R, 1).cuda()
    with torch.no_grad():
        for _ in range(5): f(rays, white_bg=True, is_train=False, N_samples=1536)
        torch.cuda.synchronize(); t=time.time()
        n = max(3, 200000 // R)
        for _ in range(n): f(rays, white_bg=True, is_train=False, N_samples=1536)
        torch.cuda.synchronize(); dt=(time.time()-t)/n
    print(f"R {R}: {dt*1e3:.3f} ms  {R/dt/1e6:.2f} M rays/s")
