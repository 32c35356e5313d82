"""Staged GPU diagnostics: each stage runs in its own process with a hard timeout and
appends to gpurun_out/diag.log, so a hang in one stage still leaves evidence."""
import faulthandler
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
LOG = os.path.join(ROOT, "gpurun_out", "diag.log")


def log(*a):
    msg = " ".join(str(x) for x in a)
    with open(LOG, "a") as f:
        f.write(msg + "\n")
    print(msg, flush=True)


def stage_import():
    t = time.time()
    import torch
    log("torch", torch.__version__, "cuda", torch.cuda.is_available(), "import s", round(time.time() - t, 1))
    p = torch.cuda.get_device_properties(0)
    log("device", p.name, "CUs", p.multi_processor_count, "mem GB", p.total_memory >> 30)
    x = torch.randn(1024, 1024, device="cuda")
    log("matmul ok", float((x @ x).sum()) != 0)
    import __graft_entry__ as ge
    deps = [os.path.join(ge.CSRC, s) for s in ge.HIP_SOURCES if os.path.exists(os.path.join(ge.CSRC, s))]
    log("lib exists", os.path.exists(ge.LIB), "stale", ge._stale(ge.LIB, deps))


def _field(grid=(20, 24, 28), seed=3):
    import torch
    from util import make_field, quiet
    f = quiet(make_field, list(grid), "cpu", seed=seed)
    with torch.no_grad():
        for p in f.density_plane:
            p.mul_(3.0)
    return f.to("cuda:0")


def stage_pack_density():
    import numpy as np
    import torch
    from oracle import vm_render_np as oracle
    f = _field()
    u = (torch.rand(300, 3) * 2.2 - 1.1).to("cuda:0")
    out = f.compute_densityfeature(u)
    torch.cuda.synchronize()
    fld = {k: v.detach().cpu().numpy() for k, v in f.state_dict().items()}
    ref = oracle.density_feature(fld, u.cpu().numpy())
    log("density_feature max err", float(np.abs(out.cpu().numpy() - ref).max()))
    out = f.compute_appfeature(u)
    torch.cuda.synchronize()
    ref = oracle.app_feature(fld, u.cpu().numpy())[0]
    log("app_feature max err", float(np.abs(out.cpu().numpy() - ref).max()))


def _render(engine, R=64, N=96, grid=(20, 24, 28)):
    import numpy as np
    import torch
    from oracle import vm_render_np as oracle
    from util import make_rays
    f = _field(grid)
    f.mlp_engine = engine
    rays = make_rays(R, 5).to("cuda:0")
    t = time.time()
    with torch.no_grad():
        rgb, depth, w, acc, z = f.render_weights(rays, N_samples=N)
    torch.cuda.synchronize()
    log(engine, "render done in", round(time.time() - t, 3), "s")
    fld = {k: v.detach().cpu().numpy() for k, v in f.state_dict().items()}
    ro, do, ex = oracle.render_field(fld, rays.cpu().numpy(), oracle.z_schedule(N), True, 0.0, return_extras=True)
    log(engine, "weights err", float(np.abs(w.cpu().numpy() - ex["weight"]).max()),
        "depth rel err", float((np.abs(depth.cpu().numpy() - do) / np.abs(do)).max()),
        "rgb err", float(np.abs(rgb.cpu().numpy() - ro).max()), "shaded", int(ex["shade"].sum()))
    return f, rays


def stage_render_valu():
    _render("valu")


def stage_render_mfma():
    _render("f32"); _render("bf16x3")


def stage_render_big():
    import torch
    f, rays = _render("bf16x3", R=4096, N=1536, grid=(300, 300, 300))
    with torch.no_grad():
        for _ in range(3):
            f(rays, N_samples=1536)
        torch.cuda.synchronize()
        t = time.time()
        for _ in range(10):
            f(rays, N_samples=1536)
        torch.cuda.synchronize()
    dt = (time.time() - t) / 10
    log("config2 forward ms", round(dt * 1e3, 3), "rays/s", round(4096 / dt))


STAGES = [("import", 240), ("pack_density", 60), ("render_valu", 60), ("render_mfma", 60), ("render_big", 120)]

if __name__ == "__main__":
    if len(sys.argv) > 1:
        faulthandler.enable()
        faulthandler.dump_traceback_later(int(sys.argv[2]) - 5, exit=True)
        globals()["stage_" + sys.argv[1]]()
        sys.exit(0)
    os.makedirs(os.path.dirname(LOG), exist_ok=True)
    only = os.environ.get("DIAG_STAGES")
    for name, tmo in STAGES:
        if only and name not in only.split(","):
            continue
        log(f"=== stage {name} (timeout {tmo}s)")
        t = time.time()
        try:
            r = subprocess.run([sys.executable, "-u", __file__, name, str(tmo)], timeout=tmo + 10,
                               capture_output=True, text=True)
            log(r.stdout[-3000:] if not r.stdout.strip() else "", "rc", r.returncode, "took", round(time.time() - t, 1))
            if r.returncode != 0:
                log("STDERR:", r.stderr[-3000:])
                if name != "render_mfma":
                    break
        except subprocess.TimeoutExpired as e:
            log("TIMEOUT in", name, "stderr:", (e.stderr or b"")[-2000:])
            break
