"""Does a small host->device copy wait for previously enqueued GPU work?  (decides whether a caller
that uploads view_ids / ray_ids every iteration serialises host and GPU)"""
import time
import torch
a = torch.randn(8192, 8192, device="cuda")
def busy():
    for _ in range(20):
        a @ a
torch.cuda.synchronize()
t = time.time(); busy(); torch.cuda.synchronize(); print("busy work ms", round((time.time() - t) * 1e3, 1))
pin = torch.tensor([1, 2, 3]).pin_memory()
import numpy as np
for name, fn in (("pageable tensor .to(cuda)", lambda: torch.tensor([1, 2, 3]).to("cuda")),
                 ("from_numpy .to(cuda)", lambda: torch.from_numpy(np.arange(4096)).to("cuda")),
                 ("torch.tensor(list, device=cuda)", lambda: torch.tensor([1, 2, 3], device="cuda")),
                 ("pinned .to(cuda, non_blocking)", lambda: pin.to("cuda", non_blocking=True)),
                 ("index with python list", lambda: a[[1, 2, 3]]),
                 (".tolist() of a cuda tensor", lambda: torch.ones(3, device="cuda").tolist())):
    torch.cuda.synchronize()
    busy()
    t = time.time(); fn(); dt = time.time() - t
    torch.cuda.synchronize()
    print(f"{name:36s} host blocked {dt * 1e3:8.2f} ms")
