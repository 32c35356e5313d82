import os, sys, time, torch
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from util import make_field, make_rays, quiet
f = quiet(make_field, [300, 300, 300], "cpu", seed=0).to("cuda:0")
rays = make_rays(4096, 1).cuda()
z = f.z_schedule(False, 1536, torch.device("cuda:0"))
with torch.no_grad():
    rgb0, d0 = f(rays, white_bg=True, is_train=False, N_samples=1536)
    out = (torch.empty_like(rgb0), torch.empty_like(d0))
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            f(rays, white_bg=True, is_train=False, N_samples=1536, out=out)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            f(rays, white_bg=True, is_train=False, N_samples=1536, out=out)
    except Exception as e:
        print("capture failed:", repr(e)[:300]); sys.exit(0)
    out[0].zero_(); out[1].zero_()
    g.replay(); torch.cuda.synchronize()
    print("replay equal:", torch.equal(out[0], rgb0), torch.equal(out[1], d0))
    for name, fn in (("eager", lambda: f(rays, white_bg=True, is_train=False, N_samples=1536, out=out)), ("graph", g.replay)):
        for _ in range(20): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): fn()
        torch.cuda.synchronize(); print(name, "%.4f ms per forward" % ((time.perf_counter() - t0) / 200 * 1e3))

# ---- the training step's forward + backward as one graph (the library's side stream and events are captured with it)
g2 = torch.Generator().manual_seed(3)
gr, gd = torch.randn(4096, 3, generator=g2).cuda(), torch.randn(4096, generator=g2).cuda()
f.z_override = z.clone()
static_rays = rays.clone().requires_grad_(True)
def fb():
    for p in f.parameters():
        p.grad = None
    static_rays.grad = None
    rgb, depth = f(static_rays, white_bg=True, is_train=False, N_samples=1536)
    ((rgb * gr).sum() + (depth * gd).sum()).backward()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        fb()
torch.cuda.current_stream().wait_stream(s)
ref = [p.grad.clone() for p in f.parameters() if p.grad is not None]
gg = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(gg):
        fb()
except Exception as e:
    print("training capture failed:", repr(e)[:400]); sys.exit(0)
gg.replay(); torch.cuda.synchronize()
got = [p.grad for p in f.parameters() if p.grad is not None]
print("training replay: max rel grad diff", max(float((a - b).abs().max() / (b.abs().max() + 1e-12)) for a, b in zip(got, ref)))
for name, fn in (("eager fwd+bwd", fb), ("graph fwd+bwd", gg.replay)):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(60): fn()
    torch.cuda.synchronize(); print(name, "%.4f ms" % ((time.perf_counter() - t0) / 60 * 1e3))
