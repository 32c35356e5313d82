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


def stage_screen():
    """Box screen: alternates two fields (300^3 x 512 samples, an empty 64^3 x 64 samples) and the colour engines, one
    render each per round with a foreign kernel (a sort) thrown in, and compares every render with that configuration's
    own first result, colour and depth.  SCREEN_ROUNDS rounds (default 400).  Must report 0 odd renders."""
    import torch
    sys.path.insert(0, ROOT)
    from util import make_field, make_rays, quiet
    big = quiet(make_field, [300, 300, 300], "cpu", seed=0).to("cuda:0")
    empty = quiet(make_field, [64, 64, 64], "cpu", seed=3)
    with torch.no_grad():
        for p in empty.density_plane:
            p.zero_()
    empty = empty.to("cuda:0")
    empty.density_shift = -30.0
    rays = make_rays(4096, 1).cuda()
    cfgs = []
    for eng in os.environ.get("SCREEN_ENGINES", "bf16x3,f32").split(","):
        cfgs.append((eng, big, 1536, "300^3"))
        cfgs.append((eng, empty, 192, "empty64"))
    n = int(os.environ.get("SCREEN_ROUNDS", "400"))
    ref, odd = {}, {}
    scratch = torch.empty(1 << 22, device="cuda")
    with torch.no_grad():
        for rnd in range(n):
            for ci, (eng, fld, ns, name) in enumerate(cfgs):
                fld.mlp_engine = eng
                if rnd % 3 == 1:
                    scratch.sort()
                rgb, dep = fld(rays, white_bg=True, is_train=False, N_samples=ns)
                if ci not in ref:
                    ref[ci] = (rgb.clone(), dep.clone())
                    odd[ci] = []
                    continue
                if not (torch.equal(rgb, ref[ci][0]) and torch.equal(dep, ref[ci][1])):
                    dc = (rgb - ref[ci][0]).abs().amax(-1)
                    dd = (dep - ref[ci][1]).abs()
                    odd[ci].append((rnd, int((dc > 0).sum()), float(dc.max()), int((dd > 0).sum()), float(dd.max())))
    total = 0
    for ci, (eng, fld, ns, name) in enumerate(cfgs):
        total += len(odd[ci])
        log(f"screen {eng} {name}: {len(odd[ci])} of {n - 1} renders differ from the first "
            f"(round, rays colour, max colour, rays depth, max depth): {odd[ci][:6]}")
    log(f"screen total odd renders: {total}")


def stage_shade3_phases():
    """s_memtime phase totals of k_shade3 per wave (TIMED build), then per-kernel HIP-event times (median of 7)."""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from localrf_amd import _native as N
    from util import make_field, make_rays, quiet
    f = quiet(make_field, [300, 300, 300], "cpu", seed=0).to("cuda:0")
    rays = make_rays(4096, 1).cuda()
    z = f.z_schedule(False, 1536, rays.device).contiguous()
    lib = N.lib()
    with torch.no_grad():
        for _ in range(300):
            f(rays, white_bg=True, is_train=False, N_samples=1536)
    names = ["prologue rest", "header+position", "gather+split", "image copy", "scan", "chain", "tiles", "finalize"]
    nw = int(os.environ.get("DIAG_NW", "8"))                 # waves per workgroup of the build under test
    buf = torch.zeros(256 * nw * 8, dtype=torch.int64, device="cuda")
    with torch.no_grad():
        lib.lrf_debug_set_dump(buf.data_ptr())
        f(rays, white_bg=True, is_train=False, N_samples=1536)
        torch.cuda.synchronize()
        lib.lrf_debug_set_dump(None)
    t = buf.view(256 * nw, 8).double()
    tiles = t[:, 6].sum()
    tot = t[:, [0, 1, 2, 3, 4, 5, 7]].sum(1)
    log(f"k_shade3 {nw} waves: tiles {int(tiles)} | cycles per wave: prologue {float(t[:, 0].mean()):.0f} (+ image copy {float(t[:, 3].mean()):.0f}, scan {float(t[:, 4].mean()):.0f}) finalize {float(t[:, 7].mean()):.0f} | per tile: " +
        " ".join(f"{names[i]} {float(t[:, i].sum() / tiles):.0f}" for i in (1, 2, 5)) +
        f" | per-wave total mean/min/max {float(tot.mean()):.0f}/{float(tot.min()):.0f}/{float(tot.max()):.0f}")
    for eng in ("bf16x3", "f32"):
        f.mlp_engine = eng
        res = []
        for rep in range(7):
            with torch.no_grad():
                for _ in range(30):
                    f(rays, white_bg=True, is_train=False, N_samples=1536)
            p = bench.kernel_profile(f, rays, z, reps=30)
            res.append((p["shade_ms"] * 1e3, p["march_ms"] * 1e3, p["total_ms"] * 1e3))
        sh = sorted(v[0] for v in res)
        log(f"engine {eng}: colour stage median {sh[3]:.1f} us (min {sh[0]:.1f}, max {sh[-1]:.1f}) | k_march median {sorted(v[1] for v in res)[3]:.1f} us | total median {sorted(v[2] for v in res)[3]:.1f} us")
    f.mlp_engine = "bf16x3"


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


def stage_chunk():
    import torch
    from util import make_field, make_rays, quiet
    f = quiet(make_field, [300, 300, 300], "cpu", seed=0).to("cuda:0")
    rays = make_rays(4096, 1).to("cuda:0")
    for eng in ("bf16x3", "f32", "valu"):
        f.mlp_engine = eng
        with torch.no_grad():
            rgb, depth = f(rays, N_samples=1536)
            rgb2, depth2 = f(rays, N_samples=1536)
            ra, da = f(rays[:1000], N_samples=1536)
            rb, db = f(rays[1000:], N_samples=1536)
        rc, dc = torch.cat([ra, rb]), torch.cat([da, db])
        bad = ((rc - rgb).abs().amax(-1) > 0).nonzero()[:, 0]
        log(eng, "repeat equal", torch.equal(rgb, rgb2), "depth chunk equal", torch.equal(dc, depth),
            "rgb chunk max diff", float((rc - rgb).abs().max()), "n rays differ", int(bad.numel()),
            "first", bad[:8].tolist())


def stage_nondet():
    """Run the split-bf16 engine repeatedly on identical inputs and count tiles whose partial
    sums differ between runs (must be 0).  LRF_BF16_VARIANT selects the experiment build."""
    import torch
    from util import make_field, make_rays, quiet
    grid = [int(v) for v in os.environ.get("DIAG_GRID", "300,300,300").split(",")]
    nR, nN = int(os.environ.get("DIAG_R", "4096")), int(os.environ.get("DIAG_N", "1536"))
    f = quiet(make_field, grid, "cpu", seed=0).to("cuda:0")
    rays = make_rays(nR, 1, pinhole=bool(os.environ.get("DIAG_PINHOLE"))).to("cuda:0")
    f.mlp_engine = os.environ.get("DIAG_ENGINE", "bf16x3")
    outs = []
    with torch.no_grad():
        nrep = int(os.environ.get("DIAG_REPS", "40"))
        ref = None
        for i in range(nrep):
            rgb, _ = f(rays, N_samples=nN)
            if nrep <= 40:
                outs.append(rgb.clone())
            else:                                   # long run: keep only mismatch counts
                if ref is None:
                    ref = rgb.clone(); outs.append(ref)
                elif not torch.equal(rgb, ref):
                    outs.append(rgb.clone())
        if nrep > 40:
            log("long run", nrep, "renders; renders differing from the first:", len(outs) - 1,
                "max abs diff", max([float((o - ref).abs().max()) for o in outs[1:]] + [0.0]))
        torch.cuda.synchronize()
        t = time.time()
        for i in range(50):
            f(rays, N_samples=nN)
        torch.cuda.synchronize()
        dt = (time.time() - t) / 50
    nd = [int(((o - outs[0]).abs().amax(-1) > 0).sum()) for o in outs[1:]]
    log("variant", os.environ.get("LRF_BF16_VARIANT", "0"), "engine", f.mlp_engine, "rays differing per run", nd,
        "max abs diff", max(float((o - outs[0]).abs().max()) for o in outs[1:]), "ms/step", round(dt * 1e3, 4))


def stage_dump():
    """Find the first stage of the split-bf16 chain whose values differ between two runs."""
    import torch
    from localrf_amd import _native as N
    from util import make_field, make_rays, quiet
    f = quiet(make_field, [300, 300, 300], "cpu", seed=0).to("cuda:0")
    rays = make_rays(4096, 1).to("cuda:0")
    R, S = 4096, 512
    names = ["X00", "X13", "X25", "fe00", "fe03", "fe12", "h1_00", "h1_31", "h1_73", "h2_00", "h2_42", "h2_73",
             "o0", "o1", "o2", "w"]
    bufs = []
    with torch.no_grad():
        for i in range(4):
            buf = torch.zeros(R * S * 64, device="cuda:0")
            N.lib().lrf_debug_set_dump(buf.data_ptr())
            f(rays, N_samples=1536)
            torch.cuda.synchronize()
            bufs.append(buf.view(R, S, 4, 16))
        N.lib().lrf_debug_set_dump(None)
    for i in range(1, 4):
        d = (bufs[i] != bufs[0])
        log("run", i, "differing entries per stage value:", {n: int(d[..., k].sum()) for k, n in enumerate(names)})
        idx = d.any(-1).nonzero()[:6]
        for ray, j, g in idx.tolist():
            a, b = bufs[0][ray, j, g].tolist(), bufs[i][ray, j, g].tolist()
            log("  ray", ray, "j", j, "g", g, [(n, x, y) for n, x, y in zip(names, a, b) if x != y])


def stage_bwd():
    import numpy as np
    import torch
    from oracle import vm_render_np as oracle
    from util import field_from_golden, load_golden, make_field, make_rays, quiet
    g = load_golden("field_small_train_grad")
    f = quiet(field_from_golden, g, "cuda:0")
    f.z_override = torch.from_numpy(oracle.z_schedule(int(g["N_samples"]), np.float32, jitter=(g["U"], g["U2"])))
    rays = torch.from_numpy(g["rays"]).cuda().requires_grad_(True)
    rgb, depth = f(rays, white_bg=True, is_train=True, N_samples=int(g["N_samples"]))
    ((rgb * torch.from_numpy(g["g_rgb"]).cuda()).sum() + (depth * torch.from_numpy(g["g_depth"]).cuda()).sum()).backward()
    rel = lambda a, b: float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-12))
    errs = {n: rel(p.grad.cpu().numpy(), g["grad." + n]) for n, p in f.named_parameters() if p.requires_grad}
    errs["rays"] = rel(rays.grad.cpu().numpy(), g["grad.rays"])
    log("grad rel-to-max errors:", {k: f"{v:.2e}" for k, v in errs.items()})
    # config-2 timing: forward + backward
    f = quiet(make_field, [300, 300, 300], "cpu", seed=0).to("cuda:0")
    rays = make_rays(4096, 1).cuda()
    gr = torch.randn(4096, 3, device="cuda")
    gd = torch.randn(4096, device="cuda")
    for it in range(3):
        rgb, depth = f(rays, is_train=True, N_samples=1536)
        ((rgb * gr).sum() + (depth * gd).sum()).backward()
    torch.cuda.synchronize()
    t = time.time()
    n = 10
    for it in range(n):
        for p in f.parameters():
            p.grad = None
        rgb, depth = f(rays, is_train=True, N_samples=1536)
        ((rgb * gr).sum() + (depth * gd).sum()).backward()
    torch.cuda.synchronize()
    dt = (time.time() - t) / n
    log("config2 train fwd+bwd ms", round(dt * 1e3, 3), "rays/s", round(4096 / dt))
    log("max grad magnitudes", {n: float(p.grad.abs().max()) for n, p in list(f.named_parameters())[2:6]})


def stage_torch_train():
    """Stock PyTorch-ROCm (ATen op chain of the reference) forward+backward at config 2."""
    import torch
    from oracle import vm_render_torch as ot
    from util import make_field, make_rays, quiet
    f = quiet(make_field, [300, 300, 300], "cpu", seed=0).to("cuda:0")
    fld = {k: v.detach().clone().requires_grad_(v.dtype == torch.float32 and "aabb" not in k.lower())
           for k, v in f.state_dict().items()}
    rays = make_rays(4096, 1).cuda()
    z = ot.z_schedule(1536, device="cuda")
    gr = torch.randn(4096, 3, device="cuda")
    gd = torch.randn(4096, device="cuda")
    leaves = [v for v in fld.values() if v.requires_grad]

    def step():
        rgb, depth = ot.render_field(fld, rays, z)
        torch.autograd.grad((rgb * gr).sum() + (depth * gd).sum(), leaves, allow_unused=True)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    dt = (time.time() - t) / 5
    log("torch-ROCm port train fwd+bwd ms", round(dt * 1e3, 2), "rays/s", round(4096 / dt))


def stage_xcd():
    """Does ray order matter?  Same rays rendered in random order vs sorted by direction."""
    import torch
    from util import make_field, make_rays, quiet
    f = quiet(make_field, [300, 300, 300], "cpu", seed=0).to("cuda:0")
    rays = make_rays(4096, 1).cuda()
    d = rays[:, 3:] / rays[:, 3:].norm(dim=-1, keepdim=True)
    # direction key: octant (3 bits) then 5 bits each of |d| components (Morton-ish)
    q = ((d.abs() * 31.99).long())
    key = ((d[:, 0] < 0).long() << 17) | ((d[:, 1] < 0).long() << 16) | ((d[:, 2] < 0).long() << 15) \
        | (q[:, 0] << 10) | (q[:, 1] << 5) | q[:, 2]
    order = torch.argsort(key)
    for name, r in (("random", rays), ("sorted", rays[order].contiguous())):
        with torch.no_grad():
            for _ in range(20):
                f(r, N_samples=1536)
            torch.cuda.synchronize()
            t = time.time()
            for _ in range(200):
                f(r, N_samples=1536)
            torch.cuda.synchronize()
        log("ray order", name, "ms/step", round((time.time() - t) / 200 * 1e3, 4))


def stage_firstcall():
    """Is the FIRST render of a fresh field different from later ones?  Per engine, 4 fresh
    fields, 5 renders each: rays whose rgb / depth differ from the last render."""
    import torch
    from util import make_field, make_rays, quiet
    rays = make_rays(512, 78, pinhole=True).to("cuda:0")
    for eng in ("bf16x3", "f32", "valu"):
        res = []
        for trial in range(4):
            f = quiet(make_field, [400, 360, 440], "cpu", seed=77).to("cuda:0")
            f.mlp_engine = eng
            outs = []
            with torch.no_grad():
                for i in range(5):
                    rgb, depth = f(rays, N_samples=-1)
                    outs.append((rgb.clone(), depth.clone()))
            res.append([(int(((o[0] - outs[-1][0]).abs().amax(-1) > 0).sum()), int(((o[1] - outs[-1][1]).abs() > 0).sum()))
                        for o in outs[:-1]])
        log("firstcall", eng, res)


def stage_fwdbwdfwd():
    """forward (grad path) -> backward -> forward (no-grad): are the two forwards identical?"""
    import torch
    from util import make_field, make_rays, quiet
    for grid, R, N in (([400, 360, 440], 512, -1), ([300, 300, 300], 4096, 1536), ([20, 24, 28], 64, 96)):
        f = quiet(make_field, grid, "cpu", seed=77).to("cuda:0")
        rays = make_rays(R, 78, pinhole=True).to("cuda:0").requires_grad_(True)
        rgb, depth = f(rays, white_bg=True, is_train=False, N_samples=N)
        with torch.no_grad():
            rgb_b, depth_b = f(rays.detach(), white_bg=True, is_train=False, N_samples=N)
        (rgb.sum() + depth.sum()).backward()
        with torch.no_grad():
            rgb2, depth2 = f(rays.detach(), white_bg=True, is_train=False, N_samples=N)
            rgb3, depth3 = f(rays.detach(), white_bg=True, is_train=False, N_samples=N)
        d = lambda a, b: (int(((a - b).abs().reshape(a.shape[0], -1).amax(-1) > 0).sum()), float((a - b).abs().max()))
        log("fwdbwdfwd", grid, "grad-fwd vs nograd-fwd (before bwd)", d(rgb.detach(), rgb_b), "vs after bwd", d(rgb.detach(), rgb2),
            "after-bwd twice", d(rgb2, rgb3), "depth", d(depth.detach(), depth2))


def stage_scene():
    """LocalTensorfs.forward at BASELINE configs[2]/[3] scale: 300^3 fields, 4096 rays; 1 field
    (train-style call, no grad) and 4 blended fields (eval)."""
    import torch
    import localrf_amd.scene as scene_mod
    from localrf_amd import LocalTensorfs
    from util import FIELD_KW, quiet
    if os.environ.get("DIAG_TORCH_ADAM"):               # baseline: the reference's per-object torch.optim.Adam
        class TorchAdamShim(torch.optim.Adam):
            @staticmethod
            def step_many(opts):
                for o in opts:
                    o.step()
        scene_mod.FusedAdam = TorchAdamShim
        log("optimiser: torch.optim.Adam, one object per frame and kind (reference layout)")
    aabb = 2 * torch.tensor([[-1.0, -1, -1], [1, 1, 1]]).cuda()
    lt = quiet(LocalTensorfs, fov=85.6, n_init_frames=5, n_overlap=3, WH=(960, 540),
               n_iters_per_frame=600, n_iters_reg=100, lr_R_init=5e-3, lr_t_init=5e-4,
               lr_i_init=0, lr_exposure_init=1e-3, rf_lr_init=0.02, rf_lr_basis=1e-3,
               lr_decay_target_ratio=0.1, N_voxel_list={}, update_AlphaMask_list=[],
               camera_prior=None, device="cuda:0", lr_upsample_reset=True,
               aabb=aabb, gridSize=[300, 300, 300], **FIELD_KW)
    g = torch.Generator().manual_seed(3)
    for _ in range(3):
        for _ in range(3):
            lt.append_frame()
            with torch.no_grad():
                lt.t_c2w[-1].add_(0.05 * torch.randn(3, generator=g).cuda())
        quiet(lt.append_rf, 3)
    for _ in range(12):                                  # frames registered against the newest field
        lt.append_frame()
    n_frames = len(lt.r_c2w)
    view_ids = torch.arange(n_frames - 16 if n_frames >= 16 else 0, n_frames).cuda()[:16]
    V = view_ids.numel()
    ray_ids = torch.randint(0, 960 * 540, (V * (4096 // V),), generator=g).cuda()
    bw = torch.tensor([[.1, .2, .3, .4]]).repeat(V, 1).cuda()

    def timeit(fn, n=30):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t = time.time()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.time() - t) / n * 1e3
    with torch.no_grad():
        t1 = timeit(lambda: lt(ray_ids, view_ids, 960, 540, is_train=True))
        t4 = timeit(lambda: lt(ray_ids, view_ids, 960, 540, is_train=False, blending_weights=bw.clone(), chunk=16384))
    log("scene forward, 4096 rays, S=344: 1 field (train call, no grad) ms", round(t1, 3), "| 4 blended fields (eval) ms", round(t4, 3),
        "| rays/s", round(ray_ids.numel() / t1 * 1e3), round(ray_ids.numel() / t4 * 1e3))

    def train_step():
        for p in lt.parameters():
            p.grad = None
        rgb, depth, _, _ = lt(ray_ids, view_ids, 960, 540, is_train=True)
        (rgb.mean() + 0.01 * depth.mean()).backward()
    t_tr = timeit(train_step, 10)
    log("scene train step (forward + backward through poses/exposure/field), ms", round(t_tr, 3))

    lt.is_refining = True
    lt.rf_iter[-1] = 0                                   # fresh field: its linked poses are optimised
    def full_iter():
        rgb, depth, _, _ = lt(ray_ids, view_ids, 960, 540, is_train=True)
        loss = rgb.mean() + 0.01 * depth.mean()
        lt.optimizer_step(loss, optimize_poses=True)
    t_it = timeit(full_iter, 10)
    ray_h, view_h = ray_ids.cpu(), view_ids.cpu()
    def full_iter_host_ids():
        rgb, depth, _, _ = lt(ray_h, view_h, 960, 540, is_train=True)
        loss = rgb.mean() + 0.01 * depth.mean()
        lt.optimizer_step(loss, optimize_poses=True)
    t_ih = timeit(full_iter_host_ids, 20)
    log("scene full iteration with ids kept on the host (no blocking copies: host runs ahead of the GPU), ms", round(t_ih, 3))
    if os.environ.get("DIAG_CPROFILE"):
        import cProfile, pstats, io
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(20):
            full_iter()
        torch.cuda.synchronize()
        pr.disable()
        buf = io.StringIO()
        pstats.Stats(pr, stream=buf).sort_stats("cumulative").print_stats(45)
        log(buf.getvalue())
    log("scene full iteration (forward + optimizer_step: backward, Adam on field/poses/exposure), ms", round(t_it, 3),
        "| active poses", len(lt._active_pose_ids()))


def stage_cone():
    """Upper bound for gather locality: per-kernel time with all rays inside a narrow cone
    (every gather L2-resident) against the benchmark's random directions."""
    import ctypes as C
    import torch
    import bench as B
    from util import make_field, make_rays, quiet
    f = quiet(make_field, [300, 300, 300], "cpu", seed=0).to("cuda:0")
    rays = make_rays(4096, 1).cuda()
    g = torch.Generator().manual_seed(5)
    cone = rays.clone()
    d0 = torch.tensor([0.48, -0.62, 0.62])
    cone[:, 3:] = (d0[None] + float(os.environ.get("DIAG_CONE", "0.02")) * torch.randn(4096, 3, generator=g)).cuda()
    z = f.z_schedule(False, 1536, rays.device).contiguous()
    for name, r in (("random", rays), ("cone", cone.contiguous())):
        with torch.no_grad():
            f(r, N_samples=1536)
        p = B.kernel_profile(f, r, z)
        log("rays", name, "march ms", round(p["march_ms"], 4), "shade ms", round(p["shade_ms"], 4),
            "n_shaded", p["n_shaded"], "ns per shaded sample", round(p["shade_ms"] * 1e6 / max(1, p["n_shaded"]), 4))


def stage_adam():
    """Optimiser step of one 300^3 field: torch.optim.Adam (foreach) vs FusedAdam, and the layout
    repack that the next forward pays."""
    import torch
    from localrf_amd import FusedAdam
    from util import make_field, make_rays, quiet
    f = quiet(make_field, [300, 300, 300], "cpu", seed=0).to("cuda:0")
    rays = make_rays(4096, 1).cuda()
    for p in f.parameters():
        p.grad = torch.randn_like(p) * 1e-3
    groups = f.get_optparam_groups(0.02, 1e-3)
    for name, cls in (("torch.optim.Adam", torch.optim.Adam), ("FusedAdam", FusedAdam)):
        opt = cls([dict(g) for g in groups], betas=(0.9, 0.99))
        for _ in range(3):
            opt.step()
        torch.cuda.synchronize()
        t = time.time()
        for _ in range(20):
            opt.step()
        torch.cuda.synchronize()
        log(name, "step ms", round((time.time() - t) / 20 * 1e3, 3))
    with torch.no_grad():
        f(rays, N_samples=1536)
        torch.cuda.synchronize()
        t = time.time()
        for _ in range(20):
            torch.autograd.graph.increment_version(f.density_plane[0])
            f._ensure_cache()
        torch.cuda.synchronize()
        log("layout repack (lrf_pack_field) ms", round((time.time() - t) / 20 * 1e3, 3))
    def zero():
        for p in f.parameters():
            p.grad = None
    gr = torch.randn(4096, 3, device="cuda"); gd = torch.randn(4096, device="cuda")
    def fb():
        zero()
        rgb, depth = f(rays, is_train=True, N_samples=1536)
        ((rgb * gr).sum() + (depth * gd).sum()).backward()
    for _ in range(3):
        fb()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(10):
        fb()
    torch.cuda.synchronize()
    log("fwd+bwd ms (config 2)", round((time.time() - t) / 10 * 1e3, 3))


def stage_reg():
    """Regularisers of the reference (tensoRF.py:83-110) at 300^3: cost per training iteration while
    rf_iter < n_iters_reg (L1 weight 1e-2 by default, TV weights 0 by default, opt.py:111-113)."""
    import torch
    from util import make_field, quiet
    f = quiet(make_field, [300, 300, 300], "cpu", seed=0).to("cuda:0")

    class TV(torch.nn.Module):                         # utils/utils.py:293-309
        TVLoss_weight = 1.0

        def forward(self, x):
            h, w = x.size(2), x.size(3)
            tv = 0
            if h > 1:
                tv = tv + torch.pow(x[:, :, 1:, :] - x[:, :, :h - 1, :], 2).mean()
            if w > 1:
                tv = tv + torch.pow(x[:, :, :, 1:] - x[:, :, :, :w - 1], 2).mean()
            return 2 * tv
    reg = TV()

    def timeit(fn, n=10):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t = time.time()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.time() - t) / n * 1e3

    def l1():
        for p in f.parameters():
            p.grad = None
        (f.density_L1() * 1e-2).backward()

    def tv():
        for p in f.parameters():
            p.grad = None
        (f.TV_loss_density(reg) + f.TV_loss_app(reg)).backward()
    torch.cuda.reset_peak_memory_stats()
    log("density_L1 forward+backward ms", round(timeit(l1), 3), "| peak memory GB", round(torch.cuda.max_memory_allocated() / 2**30, 2))
    log("TV_loss_density + TV_loss_app forward+backward ms (lrf_tv_loss_*)", round(timeit(tv), 3))
    plain = lambda x: reg(x)                              # a callable without TVLoss_weight: torch op chain
    def tv_torch():
        for p in f.parameters():
            p.grad = None
        (f.TV_loss_density(plain) + f.TV_loss_app(plain)).backward()
    log("TV_loss_density + TV_loss_app forward+backward ms (reference op chain in torch)", round(timeit(tv_torch), 3))
    log("updateAlphaMask((150,150,150)) ms", round(timeit(lambda: f.updateAlphaMask((150, 150, 150)), 3), 2),
        "| upsample_volume_grid 300->330->300 ms", round(timeit(lambda: (f.upsample_volume_grid([330, 330, 330]), f.upsample_volume_grid([300, 300, 300])), 3), 2))


def stage_big():
    """BASELINE configs[4] grid sizes: forward and forward+backward at 500^3 and 640^3 (4096 rays,
    native sample count), plus the optimiser step and the layout repack."""
    import torch
    from localrf_amd import FusedAdam
    from util import make_field, make_rays, quiet
    for g in (500, 640):
        f = quiet(make_field, [g, g, g], "cpu", seed=0).to("cuda:0")
        rays = make_rays(4096, 1).cuda()
        gr = torch.randn(4096, 3, device="cuda"); gd = torch.randn(4096, device="cuda")
        opt = FusedAdam(f.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
        with torch.no_grad():
            for _ in range(3):
                f(rays)
            torch.cuda.synchronize(); t = time.time()
            for _ in range(20):
                f(rays)
            torch.cuda.synchronize(); t_f = (time.time() - t) / 20 * 1e3
        def it():
            opt.zero_grad()
            rgb, depth = f(rays, is_train=True)
            ((rgb * gr).sum() + (depth * gd).sum()).backward()
            opt.step()
        for _ in range(2):
            it()
        torch.cuda.synchronize(); t = time.time()
        for _ in range(5):
            it()
        torch.cuda.synchronize(); t_i = (time.time() - t) / 5 * 1e3
        log(f"grid {g}^3: S={f.nSamples // 6 * 2}, forward ms {t_f:.3f} ({4096 / t_f * 1e3:.0f} rays/s), "
            f"forward+backward+FusedAdam+repack ms {t_i:.3f}, peak memory GB {torch.cuda.max_memory_allocated() / 2**30:.2f}")
        del f, opt
        torch.cuda.empty_cache()


def stage_batch():
    """Forward throughput against the batch size (300^3, 512 samples): renderer.py evaluates whole
    images in large chunks, train.py uses 4096 rays."""
    import torch
    from util import make_field, make_rays, quiet
    f = quiet(make_field, [300, 300, 300], "cpu", seed=0).to("cuda:0")
    for R in (1024, 4096, 16384, 65536, 262144):
        rays = make_rays(R, 1).cuda()
        with torch.no_grad():
            for _ in range(3):
                f(rays, N_samples=1536)
            torch.cuda.synchronize(); t = time.time()
            n = 20 if R <= 65536 else 5
            for _ in range(n):
                f(rays, N_samples=1536)
            torch.cuda.synchronize(); dt = (time.time() - t) / n
        log(f"R={R}: {dt * 1e3:.3f} ms/step, {R / dt / 1e6:.2f} M rays/s")


def stage_soak_train():
    """Long training loop at scene level: memory must stay flat (the row-saving forward allocates its
    workspace per call), the loss must stay finite and go down on a fixed target."""
    import torch
    from localrf_amd import LocalTensorfs
    from util import FIELD_KW, quiet
    torch.manual_seed(0)
    aabb = 2 * torch.tensor([[-1.0, -1, -1], [1, 1, 1]]).cuda()
    lt = quiet(LocalTensorfs, fov=85.6, n_init_frames=8, n_overlap=3, WH=(320, 180),
               n_iters_per_frame=600, n_iters_reg=100, lr_R_init=5e-3, lr_t_init=5e-4,
               lr_i_init=0, lr_exposure_init=1e-3, rf_lr_init=0.02, rf_lr_basis=1e-3,
               lr_decay_target_ratio=0.1, N_voxel_list={}, update_AlphaMask_list=[],
               camera_prior=None, device="cuda:0", lr_upsample_reset=True,
               aabb=aabb, gridSize=[128, 128, 128], **FIELD_KW)
    lt.is_refining = True
    g = torch.Generator().manual_seed(1)
    view_ids = torch.arange(8)
    ray_ids = torch.randint(0, 320 * 180, (8 * 512,), generator=g)
    target = torch.rand(ray_ids.numel(), 3, generator=g).cuda() * 0.5 + 0.25
    n_it = int(os.environ.get("DIAG_ITERS", "600"))
    losses, mem = [], []
    t = time.time()
    for it in range(n_it):
        rgb, depth, _, _ = lt(ray_ids, view_ids, 320, 180, is_train=True)
        loss = ((rgb - target) ** 2).mean() + 1e-2 * lt.tensorfs[-1].density_L1()
        lt.optimizer_step(loss, optimize_poses=True)
        if it % 100 == 0 or it == n_it - 1:
            losses.append(float(loss))
            mem.append(torch.cuda.memory_allocated() / 2**20)
    torch.cuda.synchronize()
    log("soak:", n_it, "iterations,", round((time.time() - t) / n_it * 1e3, 3), "ms/iteration; loss every 100:",
        [round(v, 5) for v in losses], "| allocated MiB:", [round(m) for m in mem],
        "| peak MiB", round(torch.cuda.max_memory_allocated() / 2**20))
    assert all(v == v and v < 1e3 for v in losses), "loss diverged"
    assert losses[-1] < losses[0], "loss did not go down"
    assert max(mem[1:]) - min(mem[1:]) < 64, "allocated memory keeps growing"


def stage_fuzz():
    """Randomised shapes / options against the ATen-op port on the same GPU: eval forward (all
    cases) and forward+backward (small cases).  Reports the worst deviations; threshold flips
    (weight > 1e-3, tensorBase.py:622) may move single rays, counted separately."""
    import numpy as np
    import torch
    from localrf_amd import AlphaGridMask
    from oracle import vm_render_torch as ot
    from util import make_field, make_rays, quiet
    rng = np.random.default_rng(int(os.environ.get("DIAG_SEED", "0")))
    n_cases = int(os.environ.get("DIAG_CASES", "40"))
    worst = {"rgb": 0.0, "depth": 0.0, "grad": 0.0, "outlier_rays": 0, "cases": 0, "grad_cases": 0}
    for case in range(n_cases):
        grid = [int(rng.integers(8, 97)) for _ in range(3)]
        R = int(rng.choice([1, 3, 63, 64, 65, 200, 511, 700]))
        ns = int(rng.choice([-1, 36, 96, 200, 402]))
        act = str(rng.choice(["softplus", "relu"]))
        white = bool(rng.integers(0, 2))
        pin = bool(rng.integers(0, 2))
        f = quiet(make_field, grid, "cpu", seed=int(rng.integers(0, 1 << 30)), fea2denseAct=act).to("cuda:0")
        with torch.no_grad():
            for p in f.density_plane:
                p.mul_(float(rng.choice([1.0, 3.0, 6.0])))
        if rng.integers(0, 2):
            vol = (torch.rand(6, 7, 5, generator=torch.Generator().manual_seed(case)) > 0.3).float()
            vol = torch.nn.functional.interpolate(vol[None, None], size=(20, 22, 18), mode="nearest")[0, 0]
            f.alphaMask = AlphaGridMask(torch.device("cuda:0"), f.aabb.detach(), vol.cuda())
        rays = make_rays(R, 100 + case, pinhole=pin).cuda()
        z = f.z_schedule(False, ns, rays.device)
        fld = {k: v for k, v in f.state_dict().items()}
        with torch.no_grad():
            rgb, depth = f(rays, white_bg=white, is_train=False, N_samples=ns)
            rgb_p, depth_p = ot.render_field(fld, rays, z[None], white, 0.0,
                                             weight_thres=f.rayMarch_weight_thres) if act == "softplus" else (None, None)
        if rgb_p is not None:
            e_rgb = (rgb - rgb_p).abs().amax(-1)
            e_dep = (depth - depth_p).abs() / depth_p.abs().clamp(min=1e-3)
            out = int((e_rgb > 1e-4).sum())
            worst["outlier_rays"] += out
            worst["rgb"] = max(worst["rgb"], float(e_rgb[e_rgb <= 1e-4].max()) if (e_rgb <= 1e-4).any() else 0.0)
            worst["depth"] = max(worst["depth"], float(e_dep.max()))
            assert float(e_rgb.max()) < 5e-3 and out <= max(1, R // 50), (case, grid, R, ns, float(e_rgb.max()), out)
            assert float(e_dep.max()) < 1e-4, (case, grid, R, ns, float(e_dep.max()))
            worst["cases"] += 1
        if rgb_p is not None and R <= 200 and f.alphaMask is None:
            gr = torch.randn(R, 3, device="cuda"); gd = torch.randn(R, device="cuda")
            f.z_override = z.clone()
            for p in f.parameters():
                p.grad = None
            r1 = rays.clone().requires_grad_(True)
            a, b = f(r1, white_bg=white, is_train=True, N_samples=ns)
            ((a * gr).sum() + (b * gd).sum()).backward()
            g_native = {n: p.grad.clone() for n, p in f.named_parameters() if p.grad is not None}
            g_native["rays"] = r1.grad.clone()
            leaves = {k: v.detach().clone().requires_grad_(True) for k, v in f.named_parameters()}
            fl = {**fld, **leaves}
            r2 = rays.clone().requires_grad_(True)
            a2, b2 = ot.render_field(fl, r2, z[None], white, 0.0)
            ((a2 * gr).sum() + (b2 * gd).sum()).backward()
            for n, gn in g_native.items():
                gp = r2.grad if n == "rays" else leaves[n].grad
                den = float(gp.abs().max())
                if den > 0:
                    l2 = float((gn - gp).norm() / gp.norm())
                    if l2 > worst["grad"]:
                        worst["grad_where"] = (n, case, grid, R, ns, "max-norm err %.2e" % (float((gn - gp).abs().max()) / den))
                    worst["grad"] = max(worst["grad"], l2)
                    assert l2 < 2e-2, (case, grid, R, ns, n, l2)
            worst["grad_cases"] += 1
            f.z_override = None
    log("fuzz:", worst)


def stage_fuzz_case():
    """Re-run one fuzz case (DIAG_SEED, DIAG_CASE) and localise the gradient difference."""
    import numpy as np
    import torch
    from oracle import vm_render_torch as ot
    from util import make_field, make_rays, quiet
    rng = np.random.default_rng(int(os.environ.get("DIAG_SEED", "0")))
    want = int(os.environ.get("DIAG_CASE", "0"))
    for case in range(want + 1):
        grid = [int(rng.integers(8, 97)) for _ in range(3)]
        R = int(rng.choice([1, 3, 63, 64, 65, 200, 511, 700]))
        ns = int(rng.choice([-1, 36, 96, 200, 402]))
        act = str(rng.choice(["softplus", "relu"]))
        white = bool(rng.integers(0, 2)); pin = bool(rng.integers(0, 2))
        seed = int(rng.integers(0, 1 << 30)); scale = float(rng.choice([1.0, 3.0, 6.0])); mask = rng.integers(0, 2)
    f = quiet(make_field, grid, "cpu", seed=seed, fea2denseAct=act).to("cuda:0")
    with torch.no_grad():
        for p in f.density_plane:
            p.mul_(scale)
    rays = make_rays(R, 100 + want, pinhole=pin).cuda()
    z = f.z_schedule(False, ns, rays.device)
    log("case", want, grid, R, ns, act, white, pin, "mask", int(mask), "S", z.numel())
    rgbw = f.render_weights(rays, N_samples=ns, white_bg=white)
    w = rgbw[2]
    near = ((w - f.rayMarch_weight_thres).abs() < 2e-7).sum()
    log("samples with |w - thres| < 2e-7:", int(near), "| shaded:", int((w > f.rayMarch_weight_thres).sum()))
    gr = torch.randn(R, 3, device="cuda"); gd = torch.randn(R, device="cuda")
    f.z_override = z.clone()
    r1 = rays.clone().requires_grad_(True)
    a, b = f(r1, white_bg=white, is_train=True, N_samples=ns)
    ((a * gr).sum() + (b * gd).sum()).backward()
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in f.named_parameters()}
    fl = {**{k: v for k, v in f.state_dict().items()}, **leaves}
    r2 = rays.clone().requires_grad_(True)
    a2, b2 = ot.render_field(fl, r2, z[None], white, 0.0)
    ((a2 * gr).sum() + (b2 * gd).sum()).backward()
    for n, p in f.named_parameters():
        if p.grad is None or "plane" not in n:
            continue
        d = (p.grad - leaves[n].grad)[0]                     # [C,H,W]
        hot = (d.abs().amax(0) > 1e-3 * leaves[n].grad.abs().max())
        log(n, "max err / max", round(float(d.abs().max() / leaves[n].grad.abs().max()), 5), "| texels off by > 1e-3 of max:", int(hot.sum()),
            "at", hot.nonzero()[:6].tolist())


def stage_soak():
    """3000 renders of the config-2 batch with the shipped engine: every one must equal the first bit for bit."""
    import torch
    from util import make_field, make_rays, quiet
    f = quiet(make_field, [300, 300, 300], "cpu", seed=0).to("cuda:0")
    rays = make_rays(4096, 1).cuda()
    with torch.no_grad():
        first, d0 = f(rays, white_bg=True, is_train=False, N_samples=1536)
        bad = 0
        for i in range(3000):
            again, d1 = f(rays, white_bg=True, is_train=False, N_samples=1536)
            if not (torch.equal(first, again) and torch.equal(d0, d1)):
                bad += 1
    log(f"soak: 3000 renders, {bad} differ from the first")


def stage_geo():
    """Geometric losses (train.py:385-423): HIP kernels vs the ATen op chain on the same GPU, forward + backward,
    4096 rays over 16 views; and the grid upsample 300^3 -> 380^3 vs F.interpolate."""
    import torch
    import torch.nn.functional as F
    sys.path.insert(0, ROOT)
    from localrf_amd import _native as N, losses
    from oracle import vm_render_torch as ot
    dev = "cuda:0"
    V, n, Fr, W, H = 16, 256, 20, 640, 480
    gen = torch.Generator().manual_seed(7)
    rot = torch.linalg.qr(torch.eye(3)[None] + 0.05 * torch.randn(Fr, 3, 3, generator=gen))[0]
    c2w = torch.cat([rot, 0.2 * torch.randn(Fr, 3, 1, generator=gen)], -1).to(dev)
    col, row = torch.randint(0, W, (V, n), generator=gen), torch.randint(0, H, (V, n), generator=gen)
    kw = dict(ij=torch.stack([col, row], -1).to(dev), view_ids=(torch.randperm(Fr, generator=gen)[:V]).to(dev), starting_frame_id=0,
              fwd_flow=(4 * torch.randn(V, n, 2, generator=gen)).to(dev), bwd_flow=(4 * torch.randn(V, n, 2, generator=gen)).to(dev),
              fwd_mask=(torch.rand(V, n, generator=gen) > 0.2).float().to(dev), bwd_mask=(torch.rand(V, n, generator=gen) > 0.2).float().to(dev))
    dirs = torch.stack([(col + 0.5 - W / 2) / 500.0, -(row + 0.5 - H / 2) / 500.0, -torch.ones(V, n)], -1).to(dev)
    depth0 = (0.5 + 5 * torch.rand(V, n, generator=gen)).to(dev)
    inv = (0.1 + torch.rand(V, n, generator=gen)).to(dev)

    def step(impl):
        leaves = dict(depth_map=depth0.clone().requires_grad_(True), directions=dirs.clone().requires_grad_(True),
                      cam2world=c2w.clone().requires_grad_(True), focal=torch.tensor([500.0], device=dev, requires_grad=True),
                      center=torch.tensor([W / 2.0, H / 2.0], device=dev, requires_grad=True))
        if impl == "hip":
            total = losses.flow_loss(**leaves, **kw) + 0.1 * losses.depth_loss(leaves["depth_map"], inv, V)
        else:
            total = ot.flow_loss(**leaves, **kw)[0] + 0.1 * ot.depth_loss(leaves["depth_map"], inv)[0]
        total.backward()
        return float(total.detach()) if False else total.detach()
    for impl in ("hip", "aten", "hip", "aten"):
        for _ in range(20):
            step(impl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            v = step(impl)
        torch.cuda.synchronize()
        log(f"geometric losses ({impl}): {(time.perf_counter() - t0) / 200 * 1e3:.3f} ms per forward+backward (4096 rays, 16 views) | value {float(v):.6f}")
    lib = N.lib()
    src = torch.randn(1, 24, 300, 300, device=dev)
    dst = torch.empty(1, 24, 380, 380, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for name, fn in (("lrf_upsample_bilinear", lambda: N.check(lib.lrf_upsample_bilinear(N.ptr(src), 24, 300, 300, N.ptr(dst), 380, 380, st), "up")),
                     ("F.interpolate", lambda: F.interpolate(src, size=(380, 380), mode="bilinear", align_corners=True))):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        log(f"upsample 24 x 300^2 -> 380^2 ({name}): {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms")
    ref = F.interpolate(src, size=(380, 380), mode="bilinear", align_corners=True)
    N.check(lib.lrf_upsample_bilinear(N.ptr(src), 24, 300, 300, N.ptr(dst), 380, 380, st), "up")
    log(f"upsample max |kernel - ATen| {float((dst - ref).abs().max()):.2e}")


def stage_train_host():
    """Host-side profile of the progressive training iteration (scripts/train_synth.py at a small resolution, where
    the GPU work is short and the iteration time is what Python spends)."""
    import cProfile, pstats, io as _io
    import torch
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import train_synth
    train_synth.run(frames=6, final=80, iters_per_frame=20, max_iters=60, dev="cuda:0")          # warm everything
    for geo in (True, False):
        out = train_synth.run(frames=8, final=100, iters_per_frame=30, max_iters=300, dev="cuda:0", geo=geo)
        log(f"plain run (geometric losses {'on' if geo else 'off'}): ms/iteration by resolution {({k: round(v, 3) for k, v in out['ms_per_iteration_by_resolution'].items()})}")
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pr.enable()
    out = train_synth.run(frames=8, final=100, iters_per_frame=30, max_iters=300, dev="cuda:0")
    pr.disable()
    torch.cuda.synchronize()
    log(f"300 iterations in {time.perf_counter() - t0:.2f} s under cProfile; ms/iteration by resolution {out['ms_per_iteration_by_resolution']}")
    buf = _io.StringIO()
    pstats.Stats(pr, stream=buf).sort_stats("cumulative").print_stats(45)
    log(buf.getvalue()[-7000:])


def stage_bwd_overlap():
    """Training step (forward with a graph + backward) with the weight-gradient GEMMs on a side stream vs in line."""
    import torch
    from localrf_amd import _native as N
    from util import make_field, make_rays, quiet
    f = quiet(make_field, [300, 300, 300], "cpu", seed=0).to("cuda:0")
    rays = make_rays(4096, 1).cuda()
    g = torch.Generator().manual_seed(3)
    gr, gd = torch.randn(4096, 3, generator=g).cuda(), torch.randn(4096, generator=g).cuda()
    lib = N.lib()

    def step():
        for p in f.parameters():
            p.grad = None
        rgb, depth = f(rays, white_bg=True, is_train=False, N_samples=1536)
        ((rgb * gr).sum() + (depth * gd).sum()).backward()
    f.z_override = None
    for r in rays, :
        r.requires_grad_(False)
    for p in f.parameters():
        p.requires_grad_(True)
    ref = None
    for on, eng in ((5, 1), (3, 1), (7, 1), (9, 1), (11, 1), (0, 1), (5, 1), (3, 1), (7, 1), (5, 3), (5, 0)):
        lib.lrf_debug_set_bwd_overlap(on)
        lib.lrf_debug_set_train_fwd_engine(eng)
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(40):
            step()
        torch.cuda.synchronize()
        dt = (time.time() - t0) / 40 * 1e3
        with torch.no_grad():                                 # the row-saving forward alone
            torch.cuda.synchronize()
            t1 = time.time()
            for _ in range(40):
                with torch.enable_grad():
                    f(rays, white_bg=True, is_train=False, N_samples=1536)
            torch.cuda.synchronize()
            dtf = (time.time() - t1) / 40 * 1e3
        gsum = {n: p.grad.double().abs().sum().item() for n, p in f.named_parameters() if p.grad is not None}
        if ref is None:
            ref = gsum
        worst = max(abs(gsum[k] - ref[k]) / max(ref[k], 1e-30) for k in ref)
        log(f"bwd overlap {on & 1}{'' if on < 2 else f' ({(on >> 1) - 1} GEMMs on the caller stream)'}, dW2 on {'fp32' if eng & 2 else 'split-bf16'} MFMA: fwd+bwd {dt:.3f} ms (forward alone {dtf:.3f} ms) | "
            f"max relative change of a gradient's |sum| vs first run {worst:.2e}")
    lib.lrf_debug_set_bwd_overlap(7)
    lib.lrf_debug_set_train_fwd_engine(1)


def stage_scene_profile():
    """Host-side profile of LocalTensorfs.forward at BASELINE configs[2] (4 blended 300^3 fields, 4096 rays)."""
    import cProfile
    import pstats
    import torch
    sys.path.insert(0, ROOT)
    import bench
    lt, ray_ids, view_ids, bw = bench.config3_scene(torch.device("cuda:0"))

    def step():
        return lt(ray_ids, view_ids, 64, 48, is_train=False, blending_weights=bw, chunk=4096)
    with torch.no_grad():
        for _ in range(20):
            step()
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(100):
            step()
        torch.cuda.synchronize()
        log(f"config3 forward: {(time.time() - t0) * 10:.3f} ms/step (host + GPU, no sync inside)")
        t0 = time.time()
        for _ in range(100):
            step()
        host = (time.time() - t0) * 10
        torch.cuda.synchronize()
        log(f"config3 forward: host time to enqueue {host:.3f} ms/step")
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(100):
            step()
        pr.disable()
        torch.cuda.synchronize()
    import io
    buf = io.StringIO()
    pstats.Stats(pr, stream=buf).sort_stats("cumulative").print_stats(28)
    log(buf.getvalue()[-4500:])


STAGES = [("screen", 300), ("shade3_phases", 200), ("train_host", 300), ("geo", 200), ("scene_profile", 200), ("bwd_overlap", 100), ("soak", 100), ("fuzz_case", 200), ("fuzz", 400), ("soak_train", 300), ("batch", 200), ("big", 200), ("reg", 120), ("adam", 120), ("cone", 120), ("import", 240), ("pack_density", 60), ("render_valu", 60), ("render_mfma", 60), ("render_big", 120), ("chunk", 120), ("nondet", 120), ("dump", 120), ("bwd", 200), ("torch_train", 200), ("xcd", 100), ("firstcall", 150), ("fwdbwdfwd", 150), ("scene", 200)]

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
