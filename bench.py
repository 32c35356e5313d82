#!/usr/bin/env python
"""Headline benchmark: rays/sec of the per-ray render path (BASELINE.json metric).

Workload at N=1 = BASELINE.json configs[1]: one 300^3 TensorVMSplit (random-init, the
reference's initialiser under torch.manual_seed(0)), 4096 rays x 512 samples
(N_samples=1536), full density + appearance + MLP render, eval mode, white background.
A "step" is one TensorVMSplit.forward over one 4096-ray batch already resident in HBM.
For N>1 every rank renders its own 4096-ray shard of an N*4096 batch (weak scaling, no
collective on the forward path); value = N*4096*K / max-over-ranks time.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see the task contract) with
  roofline      the dominant kernel: HIP-event time on the launch stream, algorithmic bytes (or flops) per
                launch, HBM-side traffic from rocprofv3 --pmc passes of THIS run (child processes; falls back
                to the committed profiles/ summary and says so), plus what the counters say binds it:
                l2_frac / mfma_frac / hbm_frac and the per-kernel table
  cpu_baseline  the reference's TensorVMSplit itself on the host cores when /root/reference is importable
                ("reference"), else oracle/vm_render_torch.py, its ATen op chain ("port", with the reason)
  train_step    forward with a graph + backward + (N>1) the gradient all-reduce over RCCL + FusedAdam, on every
                N -- the design's only collective is inside this timed region
  workloads     N=1 only: the exact-fp32 colour engine, a trained-like scene (walls + rebuilt alpha mask:
                early termination), BASELINE configs[2] (4 blended 300^3 fields), and BASELINE configs[4]'s sizes:
                fwd_500, fwd_640 (eval forward) and train_500 (training step), each with its own PMC traffic / L2 hit rate
"""
import argparse
import contextlib
import ctypes as C
import glob
import io
import json
import os
import shutil
import sqlite3
import subprocess
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GRID, R_PER_GPU, N_SAMPLES_ARG = 300, 4096, 1536          # -> S = 2*(1536//6) = 512
DENS_BYTES_PER_SAMPLE = 3 * 4 * 8 * 4 + 3 * 2 * 8 * 4      # 576  B
APP_BYTES_PER_SAMPLE = 3 * 4 * 24 * 4 + 3 * 2 * 24 * 4     # 1728 B
BASIS_FLOP = 2 * 72 * 27
MLP_FLOP = 2 * (27 * 128 + 128 * 128 + 131 * 3)           # per shaded sample, after the basis
MFMA_FLOP = 2 * 32 * 32 * 16                               # one v_mfma_f32_32x32x16_bf16
MFMA_PER_TILE = 135                                        # k_shade3, per 32-sample tile: 15 basis + 24 layer 1 + 96 layer 2 (3-term split products)
VECTOR_INSTR_PER_TILE = 1535                               # k_shade3, VALU + MFMA instructions a wave issues per tile (profiles/r08c: SQ_INSTS_VALU / tiles)
HBM_PEAK_GBS, L2_PEAK_GBS, MFMA_BF16_PEAK_TF = 8000.0, 34500.0, 2500.0   # MI355X_MICROARCH.md

FIELD_KW = dict(density_n_comp=[8, 8, 8], appearance_n_comp=[24, 24, 24], app_dim=27,
                shadingMode="MLP_Fea_late_view", near_far=[0.1, 1e3], density_shift=-5,
                alphaMask_thres=1e-4, distance_scale=25, rayMarch_weight_thres=1e-3,
                pos_pe=0, view_pe=0, fea_pe=0, featureC=128, step_ratio=0.5,
                fea2denseAct="softplus")
REF = "/root/reference/localTensoRF"


def make_rays(R, seed):
    g = torch.Generator().manual_seed(seed)
    o = 0.05 * torch.randn(R, 3, generator=g)
    d = torch.randn(R, 3, generator=g)
    return torch.cat([o, d / d.norm(dim=-1, keepdim=True)], -1)


# ------------------------------------------------------------------------------ baselines
def import_reference():
    """The reference's own TensorVMSplit (read-only import; five third-party modules it never calls on
    this path are stubbed, as tests/golden/make_golden.py does).  None + reason where it does not exist
    (the GPU box has no /root/reference)."""
    if not os.path.isdir(REF):
        return None, f"{REF} is not present on this box (it exists only in the build container)"
    import types
    try:
        for name, attrs in (("kornia", {"create_meshgrid": None}), ("cv2", {"COLORMAP_JET": 2}), ("torchvision", {}),
                            ("torchvision.transforms", {}), ("plyfile", {}), ("skimage", {}), ("skimage.measure", {})):
            if name not in sys.modules:
                m = types.ModuleType(name)
                m.__dict__.update(attrs)
                sys.modules[name] = m
        if REF not in sys.path:
            sys.path.insert(0, REF)
        from models.tensoRF import TensorVMSplit as RefVM
        return RefVM, None
    except Exception as e:                                   # noqa: BLE001
        return None, f"import of the reference failed: {e!r}"


def _ref_field(RefVM, sd, device):
    with contextlib.redirect_stdout(io.StringIO()):
        aabb = 2 * torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
        f = RefVM(device, aabb.to(device), [GRID] * 3, **FIELD_KW)
    f.load_state_dict({k: v.to(device) for k, v in sd.items()})
    return f.to(device)


def cpu_baseline(field_sd, rays_cpu, runs=3):
    """The reference on the host cores: its own module when importable, else its ATen op chain."""
    RefVM, why = import_reference()
    sd = {k: v.detach().cpu() for k, v in field_sd.items()}
    if RefVM is not None:
        f = _ref_field(RefVM, sd, "cpu")
        fn = lambda r: f(r, white_bg=True, is_train=False, N_samples=N_SAMPLES_ARG)          # noqa: E731
        kind, what = "reference", "localTensoRF/models/tensoRF.py TensorVMSplit.forward (the real reference module)"
    else:
        from oracle import vm_render_torch as ot
        z = ot.z_schedule(N_SAMPLES_ARG)
        fn = lambda r: ot.render_field(sd, r, z)                                             # noqa: E731
        kind, what = "port", "oracle/vm_render_torch.py (F.grid_sample/cumprod/Linear, pinned to the reference goldens)"
    with torch.no_grad():
        fn(rays_cpu[:512])                                            # warm-up
        t0 = time.perf_counter()
        for _ in range(runs):
            fn(rays_cpu)
        dt = (time.perf_counter() - t0) / runs
    out = {"value": rays_cpu.shape[0] / dt, "unit": "rays/s", "cores": torch.get_num_threads(), "kind": kind,
           "sample": f"{runs} x {rays_cpu.shape[0]} rays x 512 samples on the 300^3 field, {what}, {dt * 1e3:.0f} ms/batch"}
    if why:
        out["why_port"] = why
    return out


def torch_rocm_baseline(field_sd, rays, iters=5):
    """The reference path under stock PyTorch-ROCm on this GPU: denominator of the >=10x target."""
    RefVM, why = import_reference()
    if RefVM is not None:
        f = _ref_field(RefVM, {k: v.detach() for k, v in field_sd.items()}, rays.device)
        fn = lambda: f(rays, white_bg=True, is_train=False, N_samples=N_SAMPLES_ARG)         # noqa: E731
        kind = "reference"
    else:
        from oracle import vm_render_torch as ot
        fld = {k: v.detach() for k, v in field_sd.items()}
        z = ot.z_schedule(N_SAMPLES_ARG, device=rays.device)
        fn = lambda: ot.render_field(fld, rays, z)                                           # noqa: E731
        kind = "port"
    with torch.no_grad():
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / iters
    out = {"value": rays.shape[0] / dt, "unit": "rays/s", "ms_per_step": dt * 1e3, "kind": kind,
           "what": "the reference module on cuda:0" if kind == "reference" else
                   "oracle/vm_render_torch.py on cuda:0 (the reference's ATen op chain under PyTorch-ROCm, fp32)"}
    if why:
        out["why_port"] = why
    return out


def torch_rocm_train_baseline(field_sd, rays, g_rgb, g_depth, iters=3):
    """One training step (forward with a graph + backward + Adam) of the reference path under stock PyTorch-ROCm on this GPU:
    the reference module when importable, else its ATen op chain -- same batch, same loss, same optimiser groups."""
    RefVM, why = import_reference()
    if RefVM is not None:
        f = _ref_field(RefVM, {k: v.detach() for k, v in field_sd.items()}, rays.device)
        opt = torch.optim.Adam(f.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99))
        render = lambda: f(rays, white_bg=True, is_train=True, N_samples=N_SAMPLES_ARG)      # noqa: E731
        kind = "reference"
    else:
        from oracle import vm_render_torch as ot
        fld = {k: v.detach().clone() for k, v in field_sd.items()}
        leaves = [k for k in fld if fld[k].dtype.is_floating_point and ("plane" in k or "line" in k or "basis" in k or "renderModule" in k)]
        for k in leaves:
            fld[k].requires_grad_(True)
        opt = torch.optim.Adam([fld[k] for k in leaves], lr=0.02, betas=(0.9, 0.99))
        z = ot.z_schedule(N_SAMPLES_ARG, device=rays.device)
        render = lambda: ot.render_field(fld, rays, z)                                       # noqa: E731
        kind = "port"

    def step():
        opt.zero_grad(set_to_none=True)
        rgb, depth = render()
        ((rgb * g_rgb).sum() + (depth * g_depth).sum()).backward()
        opt.step()
    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    out = {"ms_per_step": dt * 1e3, "rays_per_s": rays.shape[0] / dt, "kind": kind,
           "what": ("the reference module" if kind == "reference" else "oracle/vm_render_torch.py (the reference's ATen op chain)")
                   + " on cuda:0: forward with a graph + autograd backward + torch.optim.Adam, fp32, 4096 rays x 512 samples"}
    if why:
        out["why_port"] = why
    return out


def size_workloads(dev, sync, use_pmc):
    """BASELINE configs[4]'s own sizes in the bench line: the eval forward at 500^3 and 640^3 (the reference's default end
    size, opt.py:62) and the training step at 500^3, 4096 rays at each grid's default sample count (S = 576 / 738, what
    train.py runs), each with per-kernel HIP-event times, HBM-side PMC traffic and L2 hit rate of ITS OWN run -- at these
    sizes the field (96 / 158 MB) no longer fits the 32 MB of aggregate L2."""
    from localrf_amd import FusedAdam, TensorVMSplit
    out = {}
    aabb = 2 * torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    rays = make_rays(R_PER_GPU, 1).to(dev)
    for g in (500, 640):
        torch.manual_seed(0)
        f = TensorVMSplit(torch.device("cpu"), aabb, [g] * 3, **FIELD_KW).to(dev)
        z = f.z_schedule(False, -1, dev).contiguous()
        S = int(z.shape[0])
        with torch.no_grad():                                    # two timed blocks, the faster one: the first block after a new 100-MB-class
            fwd_g = lambda: f(rays, white_bg=True, is_train=False, N_samples=-1)   # noqa: E731  field is built has been seen 8 x slow
            d = min(timed(fwd_g, 20, 10, sync), timed(fwd_g, 20, 0, sync))
        prof = kernel_profile(f, rays, z)
        n_par = sum(p.numel() for p in f.parameters())
        rec = {"what": f"single {g}^3 TensorVMSplit ({n_par * 4 / 1e6:.0f} MB of parameters), 4096 rays x {S} samples (the grid's default), eval forward",
               "rays_per_s": R_PER_GPU * 20 / d, "ms_per_step": d / 20 * 1e3, "samples_per_ray": S,
               "k_march_ms": prof["march_ms"], "k_shade3_ms": prof["shade_ms"], "shaded_fraction": prof["n_shaded"] / (R_PER_GPU * S),
               "alg_bytes": R_PER_GPU * S * DENS_BYTES_PER_SAMPLE + prof["n_shaded"] * APP_BYTES_PER_SAMPLE}
        if use_pmc:
            tr, src = pmc_traffic(mode="fwd", grid=g, n_samples=-1, steps=5)
            rec["traffic_source"] = src
            if tr:
                rec["kernels"] = {k: {"traffic_bytes": v["traffic_bytes"], "l2_hit_rate": v["l2_hit_rate"], "profiled_us": v["profiled_us"]}
                                  for k, v in tr.items() if k in ("k_march", "k_shade3")}
                rec["traffic"] = sum(v["traffic_bytes"] for v in rec["kernels"].values())
        out[f"fwd_{g}"] = rec
        if g == 500:
            opt = FusedAdam(f.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), pack_field=f)
            gr, gd = torch.randn(R_PER_GPU, 3, device=dev), torch.randn(R_PER_GPU, device=dev)

            def step():
                opt.zero_grad()
                rgb, depth = f(rays, white_bg=True, is_train=True, N_samples=-1)
                ((rgb * gr).sum() + (depth * gd).sum()).backward()
                opt.step()
            dtt = min(timed(step, 10, 3, sync), timed(step, 10, 0, sync))
            rt = {"what": f"training step at 500^3: lrf_render_fwd_train + lrf_render_bwd + FusedAdam ({n_par * 4 / 1e6:.0f} MB of parameters, "
                          f"m, v), 4096 rays x {S} jittered samples", "ms_per_step": dtt / 10 * 1e3, "rays_per_s": R_PER_GPU * 10 / dtt}
            if use_pmc:
                tr, src = pmc_traffic(mode="train", grid=g, n_samples=-1, steps=5)
                rt["traffic_source"] = src
                if tr:
                    n_steps = 3 + 5                                          # the child runs 3 warm-up + 5 timed training steps
                    tab = {k: {"traffic_bytes_per_step": v["traffic_bytes"] * v["launches"] / n_steps, "l2_hit_rate": v["l2_hit_rate"],
                               "profiled_us": v["profiled_us"], "launches_per_step": v["launches"] / n_steps} for k, v in tr.items()}
                    rt["kernels"] = dict(sorted(tab.items(), key=lambda kv: -kv[1]["traffic_bytes_per_step"]))
                    rt["traffic"] = sum(v["traffic_bytes_per_step"] for v in tab.values())
                    rt["hbm_frac"] = rt["traffic"] / (dtt / 10) / 1e9 / HBM_PEAK_GBS
            out["train_500"] = rt
            del opt
        del f
        torch.cuda.empty_cache()
    return out


def progressive_loop_workload(dev):
    """BASELINE configs[4]: the progressive loop of train.py:349-474 (scripts/train_synth.py: synthetic frames, append_frame /
    append_rf, the upsample ladder, alpha-mask rebuilds, photometric + flow + depth + density_L1 losses, every Adam) on the
    reference's schedule (600 iterations per frame, a frame every 100), bounded to ~3 s per mode: ms per iteration at each
    grid size, the iteration captured as one replayed hipGraph (localrf_amd/graph_step.py) beside the eager loop."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import train_synth
    out = {"what": "scripts/train_synth.py, 8 frames 64x48, 4096 rays per iteration, 64^3 -> 200^3, iters_per_frame 600 (the reference's), "
                   "first 2400 iterations; full runs to 500^3: profiles/r14_train_synth_500_*.json"}
    for mode in ("graph", "eager"):
        r = train_synth.run(frames=8, final=200, iters_per_frame=600, n_max_frames=6, max_iters=2400, dev=str(dev), graph=(mode == "graph"))
        out[mode] = {"ms_per_iteration_by_resolution": r["ms_per_iteration_by_resolution"], "iterations_by_resolution": r["iterations_by_resolution"],
                     "loss_first": r["loss_first"], "loss_last": r["loss_last"], "capture_stats": r["graph"]}
    return out


def geometric_losses_workload(dev, V=16, n=256, Fr=20, W=640, H=480, iters=100):
    """the flow + depth losses of train.py:385-423 on a 4096-ray batch (16 views), forward + backward:
    localrf_amd.losses (HIP) and, beside it, the ATen op chain (oracle/vm_render_torch.py: baseline leg only)."""
    from localrf_amd import losses
    from oracle import vm_render_torch as ot
    gen = torch.Generator().manual_seed(7)
    rot = torch.linalg.qr(torch.eye(3)[None] + 0.05 * torch.randn(Fr, 3, 3, generator=gen))[0]
    c2w = torch.cat([rot, 0.2 * torch.randn(Fr, 3, 1, generator=gen)], -1).to(dev)
    col, row = torch.randint(0, W, (V, n), generator=gen), torch.randint(0, H, (V, n), generator=gen)
    kw = dict(ij=torch.stack([col, row], -1).to(dev), view_ids=torch.randperm(Fr, generator=gen)[:V].to(dev), starting_frame_id=0,
              fwd_flow=(4 * torch.randn(V, n, 2, generator=gen)).to(dev), bwd_flow=(4 * torch.randn(V, n, 2, generator=gen)).to(dev),
              fwd_mask=(torch.rand(V, n, generator=gen) > 0.2).float().to(dev), bwd_mask=(torch.rand(V, n, generator=gen) > 0.2).float().to(dev))
    dirs = torch.stack([(col + 0.5 - W / 2) / 500.0, -(row + 0.5 - H / 2) / 500.0, -torch.ones(V, n)], -1).to(dev)
    depth0 = (0.5 + 5 * torch.rand(V, n, generator=gen)).to(dev)
    inv = (0.1 + torch.rand(V, n, generator=gen)).to(dev)

    def step(impl):
        lv = dict(depth_map=depth0.clone().requires_grad_(True), directions=dirs.clone().requires_grad_(True),
                  cam2world=c2w.clone().requires_grad_(True), focal=torch.tensor([500.0], device=dev, requires_grad=True),
                  center=torch.tensor([W / 2.0, H / 2.0], device=dev, requires_grad=True))
        if impl == "hip":
            total = losses.flow_loss(**lv, **kw) + 0.1 * losses.depth_loss(lv["depth_map"], inv, V)
        else:
            total = ot.flow_loss(**lv, **kw)[0] + 0.1 * ot.depth_loss(lv["depth_map"], inv)[0]
        total.backward()
        return total.detach()
    out = {"what": "flow + depth losses of train.py:385-423, forward + backward, 4096 rays over 16 views"}
    for impl in ("hip", "aten"):
        for _ in range(10):
            step(impl)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(iters):
            v = step(impl)
        torch.cuda.synchronize(dev)
        out[impl + "_ms"] = (time.perf_counter() - t0) / iters * 1e3
        out[impl + "_value"] = float(v)
    return out


# ------------------------------------------------------------------------------ measurement helpers
def timed(fn, steps, warmup, sync):
    for _ in range(warmup):
        fn()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    sync()
    return time.perf_counter() - t0


def kernel_profile(field, rays, z, reps=5):
    """Per-kernel HIP-event timing through lrf_render_fwd_profile (events on the launch stream)."""
    from localrf_amd import _native as N
    lib = N.lib()
    field._ensure_cache()
    R, S = rays.shape[0], z.shape[0]
    dev = rays.device
    rgb = torch.empty(R, 3, device=dev)
    depth = torch.empty(R, device=dev)
    ws = field._workspace(R, S, dev)
    f = field._c_field()
    st = torch.cuda.current_stream(dev).cuda_stream
    ms = (C.c_float * 6)()
    nsh = C.c_int32(0)
    acc = [0.0] * 6
    for i in range(reps + 1):
        N.check(lib.lrf_render_fwd_profile(C.byref(f), N.ptr(rays), N.ptr(z), R, S, field._flags(True), 0.0,
                                           N.ptr(rgb), N.ptr(depth), ws.data_ptr(), st, ms, C.byref(nsh)),
                "lrf_render_fwd_profile")
        if i:                                   # first call is a warm-up
            for j in range(6):
                acc[j] += ms[j] / reps
    return {"march_ms": acc[0], "shade_ms": acc[1], "finalize_ms": acc[2], "total_ms": acc[3], "n_shaded": int(nsh.value)}


PMC_PASSES = (("FETCH_SIZE",), ("WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"))


def pmc_traffic(timeout_s=240, mode="fwd", grid=GRID, n_samples=N_SAMPLES_ARG, steps=10):
    """HBM-side bytes per launch of every lrf kernel, measured now: one rocprofv3 --kernel-trace --pmc
    child per counter group (FETCH_SIZE and WRITE_SIZE do not fit one pass), each re-running this script
    for a few steps.  traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE) KB -- FETCH_SIZE reports half the bytes
    of 16-B-per-lane reads on gfx950 (MI355X_MICROARCH.md, HBM).  None if rocprofv3 is unavailable."""
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    per = {}
    for ctrs in PMC_PASSES:
        d = tempfile.mkdtemp(prefix="lrf_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", *ctrs, "-d", d, "-o", "pmc", "--", sys.executable,
               os.path.join(ROOT, "bench.py"), "--steps", str(steps), "--warmup", "2", "--child", mode,
               "--grid", str(grid), "--n-samples", str(n_samples)]
        if mode == "train":
            cmd += ["--preroll-ms", "0"]                         # no eval renders in the counted run
        try:
            r = subprocess.run(cmd, cwd="/tmp", env={**os.environ, "TMPDIR": "/tmp"}, timeout=timeout_s,
                               capture_output=True, text=True)
            dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
            if r.returncode != 0 or not dbs:
                return None, f"rocprofv3 pass {ctrs} failed (rc {r.returncode}): {r.stderr[-300:]}"
            db = sqlite3.connect(dbs[0])
            q = ("select kernel_name, counter_name, dispatch_id, sum(value) from counters_collection "
                 "group by kernel_name, counter_name, dispatch_id")
            acc = {}
            for name, ctr, _, val in db.execute(q):
                short = name.split("(")[0].replace("lrf::", "").replace("void ", "")
                if short.startswith("k_"):
                    acc.setdefault((short.split("<")[0], ctr), []).append(val)
            for (k, ctr), vals in acc.items():
                per.setdefault(k, {})[ctr] = sum(vals) / len(vals)
                per[k]["launches"] = len(vals)
            for name, n, avg in db.execute("select name, count(*), avg(duration) from kernels group by name"):
                short = name.split("(")[0].replace("lrf::", "").replace("void ", "")
                if short.startswith("k_"):                      # (durations under counter collection: serialised, slightly inflated)
                    per.setdefault(short.split("<")[0], {})["profiled_us"] = avg / 1e3
        except Exception as e:                               # noqa: BLE001
            return None, f"rocprofv3 pass {ctrs}: {e!r}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out = {}
    for k, c in per.items():
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            out[k] = {"traffic_bytes": (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024, "launches": c.get("launches", 0),
                      "l2_hit_rate": c.get("TCC_HIT_sum", 0.0) / max(c.get("TCC_HIT_sum", 0.0) + c.get("TCC_MISS_sum", 0.0), 1.0),
                      "profiled_us": c.get("profiled_us")}
    return out, "rocprofv3 --kernel-trace --pmc, 2 passes of this run; traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024"


def committed_traffic():
    for name in ("r03_pmc_traffic.json",):
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", name)))
            out = {k: {"traffic_bytes": v["traffic_bytes"],
                                                           "l2_hit_rate": v["TCC_HIT"] / max(v["TCC_HIT"] + v["TCC_MISS"], 1.0)}
                   for k, v in pmc["kernels"].items()}
            return out, f"NOT measured in this run: committed profiles/{name} ({pmc.get('source', '')[:80]})"
        except Exception:                                    # noqa: BLE001
            continue
    return {}, None


def roofline_object(prof, S, traffic, traffic_src, cus=256, clock_ghz=2.4):
    """SURVEY.md s8d figures per launch over HIP-event durations (DESIGN.md s5 says how to read them).  The 34.6 MB
    the forward touches of the field are cache-resident, so the HBM roof cannot bind either kernel; what can bind is reported beside the contract's
    figure: matrix-pipe occupancy (issued MFMAs, three per product), the vector issue port (VALU + MFMA instructions at
    four cycles each per SIMD), L2 rate, HBM-side traffic from the counters."""
    n_sh = prof["n_shaded"]
    tiles = (n_sh + 31) // 32                              # lower bound (rays pad their last tile; the kernel counts ~3 % more)
    dens_bytes = R_PER_GPU * S * DENS_BYTES_PER_SAMPLE
    app_bytes = n_sh * APP_BYTES_PER_SAMPLE
    kern = {"k_march": {"ms": prof["march_ms"], "alg_bytes": dens_bytes, "bound": "hbm"},
            "k_shade3": {"ms": prof["shade_ms"], "alg_bytes": app_bytes, "bound": "mfma",
                         "alg_flop": n_sh * (BASIS_FLOP + MLP_FLOP), "issued_mfma_flop": tiles * MFMA_PER_TILE * MFMA_FLOP,
                         "vector_instr": tiles * VECTOR_INSTR_PER_TILE}}
    if prof["finalize_ms"] > 1e-3:
        kern["k_finalize"] = {"ms": prof["finalize_ms"]}
    for name, k in kern.items():
        t = k["ms"] * 1e-3
        if "alg_bytes" in k and t > 0:
            k["GBps"] = k["alg_bytes"] / t / 1e9
            k["l2_frac"] = k["GBps"] / L2_PEAK_GBS        # against the aggregate L2 bandwidth: gathers are cache-served
        if "alg_flop" in k and t > 0:
            k["alg_TFLOPs"] = k["alg_flop"] / t / 1e12
            k["mfma_frac"] = k["issued_mfma_flop"] / t / 1e12 / MFMA_BF16_PEAK_TF   # matrix-pipe occupancy, 3 MFMAs per product
            k["issue_frac"] = k["vector_instr"] * 4.0 / (t * clock_ghz * 1e9 * cus * 4)   # vector issue port: 4 cycles per instruction and SIMD
        tr = traffic.get(name)
        if tr:
            k["pmc_traffic_bytes"] = tr["traffic_bytes"]
            k["l2_hit_rate"] = tr["l2_hit_rate"]
            if t > 0:
                k["hbm_frac"] = tr["traffic_bytes"] / t / 1e9 / HBM_PEAK_GBS
        k["binding_frac"] = max(k.get(n, 0.0) or 0.0 for n in ("l2_frac", "mfma_frac", "issue_frac", "hbm_frac"))
    dom = max(("k_march", "k_shade3"), key=lambda n: kern[n]["ms"])
    d = kern[dom]
    if d["bound"] == "mfma":
        ach, peak, unit = d["alg_TFLOPs"], MFMA_BF16_PEAK_TF, "TFLOP/s"
    else:
        ach, peak, unit = d["GBps"], HBM_PEAK_GBS, "GB/s"
    return {"bound": d["bound"], "kernel": dom, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak,
            "traffic": d.get("pmc_traffic_bytes"), "traffic_source": traffic_src,
            "l2_frac": d.get("l2_frac"), "mfma_frac": d.get("mfma_frac"), "issue_frac": d.get("issue_frac"), "hbm_frac": d.get("hbm_frac"),
            "binding_frac": d.get("binding_frac"),
            "limiter": "vector instruction issue: a 32-sample tile costs a wave ~1400 VALU + 135 MFMA instructions through one issue "
                       "port per SIMD (issue_frac); the matrix pipe runs the three-term split products at mfma_frac of its dense "
                       "bf16 rate (frac counts algorithmic flops: one product per weight); the 34.6 MB the forward touches of the field is cache-resident, "
                       "so neither HBM (hbm_frac) nor L2 (l2_frac) binds",
            "whole_path_GBps": (dens_bytes + app_bytes) / (prof["total_ms"] * 1e-3) / 1e9,
            "shaded_fraction": n_sh / (R_PER_GPU * S), "kernels": kern,
            "note": "achieved = SURVEY.md s8d algorithmic flops (bytes for k_march) of the dominant kernel per launch / its "
                    "HIP-event duration on the launch stream"}


def walls_field(dev, grid=GRID):
    """Trained-like scene: near-empty space (density planes x 0.1) inside a closed box of dense walls at
    |x|,|y|,|z| ~ 0.9 (six rank-1 components: plane = 1, line = smooth bump of height 40), then the
    alpha mask rebuilt on the device as train.py does at its update iterations."""
    from localrf_amd import TensorVMSplit
    torch.manual_seed(0)
    aabb = 2 * torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    f = TensorVMSplit(torch.device("cpu"), aabb, [grid] * 3, **FIELD_KW)
    with torch.no_grad():
        for p in f.density_plane:
            p.mul_(0.1)
        c = torch.linspace(-2, 2, grid)
        for p in range(3):
            for comp, centre in ((0, 0.9), (1, -0.9)):
                f.density_plane[p][0, comp].fill_(1.0)
                f.density_line[p][0, comp, :, 0] = 40.0 * torch.exp(-((c - centre) / 0.08) ** 2)
    return f.to(dev)


def config3_scene(dev):
    """BASELINE.json configs[2]: LocalTensorfs with 4 overlapping 300^3 fields, built as SURVEY.md s8d says."""
    from localrf_amd import LocalTensorfs
    torch.manual_seed(33)
    aabb = 2 * torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    with contextlib.redirect_stdout(io.StringIO()):
        lt = LocalTensorfs(fov=85.6, n_init_frames=5, n_overlap=3, WH=(64, 48), n_iters_per_frame=600, n_iters_reg=100,
                           lr_R_init=5e-3, lr_t_init=5e-4, lr_i_init=0, lr_exposure_init=1e-3, rf_lr_init=0.02,
                           rf_lr_basis=1e-3, lr_decay_target_ratio=0.1, N_voxel_list={}, update_AlphaMask_list=[],
                           camera_prior=None, device="cpu", lr_upsample_reset=True, aabb=aabb, gridSize=[GRID] * 3, **FIELD_KW)
        g = torch.Generator().manual_seed(34)
        for _ in range(3):
            for _ in range(3):
                lt.append_frame()
                with torch.no_grad():
                    lt.t_c2w[-1].add_(0.05 * torch.randn(3, generator=g))
                    lt.r_c2w[-1].add_(0.05 * torch.randn(3, 2, generator=g))
            lt.append_rf(3)
    lt = lt.to(dev)
    lt.device = torch.device(dev)
    for f in lt.tensorfs:
        f.to(dev)
    view_ids = [2, 7, 11, len(lt.r_c2w) - 1]
    ray_ids = torch.randint(0, 64 * 48, (4 * 1024,), generator=g)
    bw = torch.tensor([[.1, .2, .3, .4]]).repeat(4, 1)        # on the host: reading the active set off a device tensor syncs
    return lt, ray_ids, view_ids, bw


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)        # what the driver runs
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-baselines", action="store_true", help="skip the CPU / torch-ROCm baselines and extra workloads")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="process-group backend for N > 1: nccl = RCCL over xGMI (one GPU per rank); gloo = host-staged, ranks may share a GPU (launch test)")
    ap.add_argument("--no-pmc", action="store_true", help="do not spawn the rocprofv3 --pmc passes for roofline.traffic")
    ap.add_argument("--child", nargs="?", const="fwd", default=None, choices=("fwd", "train"),
                    help="(internal) the timed loop only: what the PMC passes profile (fwd: eval forward; train: training step)")
    ap.add_argument("--preroll-ms", type=float, default=1500.0,
                    help="untimed rendering before the warm-up steps: after an idle period the GPU clock of the boxes of this pool needs ~1 s of work "
                         "to reach its sustained value (same box, round 5: 0.189 ms per step behind 150 ms of work, 0.163 ms behind 1500 or 3000 ms)")
    ap.add_argument("--grid", type=int, default=GRID, help="(internal, with --child) grid size of the profiled workload")
    ap.add_argument("--n-samples", type=int, default=N_SAMPLES_ARG, help="(internal, with --child) N_samples argument (-1 = the grid's default)")
    args = ap.parse_args()
    if not args.child and (args.grid != GRID or args.n_samples != N_SAMPLES_ARG):
        raise SystemExit("--grid / --n-samples are for the PMC child runs; the headline workload is BASELINE configs[1]")
    grid, ns_arg = args.grid, args.n_samples
    if args.child:                                            # the PMC passes count bytes, not time: no need to warm the clock
        args.preroll_ms = min(args.preroll_ms, 150.0)

    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` on its own: become the launcher the driver would have used -- one rank per GPU under
        # torch.distributed.run, same arguments (rank 0 prints the JSON line)
        import socket
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.execv(sys.executable, cmd)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch.distributed as dist
    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        raise SystemExit("bench.py needs an MI355X (torch.cuda.device_count() == 0)")
    if local >= n_dev and args.backend == "nccl":
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local} but {n_dev} GPU(s) visible -- RCCL needs one GPU per rank "
                         "(--backend gloo shares GPUs between ranks: a launch test, not a measurement)")
    dev_index = local % n_dev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # under torch.distributed.run (RANK set) the process group is created even for one rank, so the
    # RCCL code path below is the same for every N
    ddp = (world > 1 or "RANK" in os.environ) and not args.child
    if ddp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI
        else:
            dist.init_process_group("gloo")                  # host-staged: several ranks may share a GPU
        if world == 1:
            os.environ.setdefault("LRF_DIST_FORCE", "1")     # one rank: still issue every collective of the N-rank step

    def barrier():
        if args.backend == "nccl":
            dist.barrier(device_ids=[dev_index])
        else:
            dist.barrier()

    import __graft_entry__ as ge
    if ddp and local != 0:                                    # one rank compiles, the rest wait for it
        barrier()
    ge.build()
    if ddp and local == 0:
        barrier()
    from localrf_amd import FusedAdam, TensorVMSplit
    from localrf_amd.dist import allreduce_grads

    torch.manual_seed(0)                                      # identical replica on every rank
    aabb = 2 * torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    field = TensorVMSplit(torch.device("cpu"), aabb, [grid] * 3, **FIELD_KW).to(dev)
    rays_cpu = make_rays(R_PER_GPU, 1 + rank)                 # this rank's shard
    rays = rays_cpu.to(dev)

    def sync():
        torch.cuda.synchronize(dev)
        if ddp:
            barrier()
            torch.cuda.synchronize(dev)

    def fwd():
        return field(rays, white_bg=True, is_train=False, N_samples=ns_arg)

    with torch.no_grad():
        # clock ramp: the first second of work after an idle period runs at a lower GPU clock; render untimed
        # for --preroll-ms before the W warm-up steps so that K is measured at the sustained clock
        t_pre = time.perf_counter()
        while (time.perf_counter() - t_pre) * 1e3 < args.preroll_ms:
            for _ in range(20):
                fwd()
            torch.cuda.synchronize(dev)
        dt = timed(fwd, args.steps, args.warmup, sync) if args.child != "train" else 1.0
    if args.child == "fwd":
        return
    def max_over_ranks(vals):
        t = torch.tensor(vals, dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        if ddp:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t.tolist()]

    dt = max_over_ranks([dt])[0]
    ms_per_step = dt / args.steps * 1e3
    value = world * R_PER_GPU * args.steps / dt

    # ---- training step, every N: forward with a graph, backward, gradient all-reduce (RCCL), FusedAdam
    sd_init = {k: v.detach().clone() for k, v in field.state_dict().items()}     # the workloads below render the untrained field again
    opt = FusedAdam(field.get_optparam_groups(0.02, 1e-3), betas=(0.9, 0.99), pack_field=field)
    gr = torch.randn(R_PER_GPU, 3, device=dev)
    gd = torch.randn(R_PER_GPU, device=dev)
    reduced = [0]

    comm = [True]
    comm_stats = {}

    def train_step():
        opt.zero_grad()
        rgb, depth = field(rays, white_bg=True, is_train=True, N_samples=ns_arg)
        ((rgb * gr).sum() + (depth * gd).sum()).backward()
        if ddp and comm[0]:
            reduced[0] = allreduce_grads(field, stats=comm_stats)
        opt.step()
    t_steps = max(5, min(args.steps, 20))
    dtt_nocomm = None
    if ddp:                                                  # the same step without the collectives: what the exchange costs on this N
        comm[0] = False
        dtt_nocomm = timed(train_step, t_steps, 3, sync)
        comm[0] = True
    dtt = timed(train_step, t_steps, 3, sync)
    if args.child == "train":                                 # the PMC passes count exactly (3 + t_steps) training steps
        return
    field.load_state_dict(sd_init)
    del sd_init
    tm = max_over_ranks([dtt, dtt_nocomm if dtt_nocomm is not None else 0.0])
    dtt = tm[0]
    if dtt_nocomm is not None:
        dtt_nocomm = tm[1]

    if rank == 0:
        z = field.z_schedule(False, N_SAMPLES_ARG, dev).contiguous()
        S = z.shape[0]
        prof = kernel_profile(field, rays, z)
        traffic, traffic_src = ({}, None)
        if world == 1 and not args.no_pmc:
            traffic, traffic_src = pmc_traffic()
        if not traffic:
            why = traffic_src
            traffic, traffic_src = committed_traffic()
            if why and traffic_src:
                traffic_src += f" [{why}]"
        roofline = roofline_object(prof, S, traffic or {}, traffic_src, cus=torch.cuda.get_device_properties(dev).multi_processor_count)
        n_sh = prof["n_shaded"]
        rows = ((n_sh + 15) // 16) * 16
        # bytes the training step moves through HBM-side memory by construction: the saved activation / gradient rows
        # (ACT 32 + GRD 128 floats per row since round 4 -- hidden activations and the plane x line products are recomputed, not
        # stored; dW1 and dbasis are accumulated in the kernels that hold their operands), the density features (R*S floats,
        # written + read twice), reference-layout gradients + Adam (param, m, v read+write) -- cache-served gathers not counted
        n_par = sum(p.numel() for p in field.parameters() if p.requires_grad)
        # per row: written feat 128 B + GRD 512 B (go block 64, dfeat 128, dX 320) + mask bits 32 B + rgb / rowinfo 16 B;
        # read: k_wgrad_w2w3 feat + go + bits 224 B, k_train_dgrad3 bits + rgb + feat 172 B, k_train_app3 dfeat 128 B,
        # the appearance scatter dX + ids 304 B
        train_bytes = rows * ((128 + 512 + 32 + 16) + (224 + 172 + 128 + 304)) + R_PER_GPU * S * 4 * 3 + n_par * 4 * (3 + 7)
        train = {"ms_per_step": dtt / t_steps * 1e3, "rays_per_s": world * R_PER_GPU * t_steps / dtt, "steps": t_steps,
                 "ms_per_step_without_allreduce": (dtt_nocomm / t_steps * 1e3 if dtt_nocomm is not None else None),
                 "allreduce": ({"backend": args.backend, "n_ranks_seen": dist.get_world_size(), "collectives_per_step": comm_stats.get("collectives"),
                                "bytes_per_piece": comm_stats.get("chunks"), "field_bytes": comm_stats.get("field_bytes"),
                                "order": "density planes+lines | colour network | appearance plane 0 | plane 1 | plane 2 + lines: each handed to the "
                                         "collective behind the event lrf_render_bwd records when that piece is final (side stream), in place in the flat gradient buffer",
                                "forced_at_one_rank": world == 1} if ddp else None),
                 "what": "lrf_render_fwd_train + lrf_render_bwd + "
                         + (f"allreduce_grads over RCCL ({reduced[0] / 1e6:.1f} MB in place) + " if ddp else "")
                         + "FusedAdam, 4096 rays x 512 samples per GPU, jittered samples",
                 "roofline": {"bound": "hbm", "achieved": train_bytes / (dtt / t_steps) / 1e9, "peak": HBM_PEAK_GBS,
                              "unit": "GB/s", "frac": train_bytes / (dtt / t_steps) / 1e9 / HBM_PEAK_GBS,
                              "bytes_per_step": train_bytes,
                              "note": "materialised rows + feature / gradient buffers + Adam, by construction; per-kernel "
                                      "times in profiles/"}}
        if world == 1 and not args.no_pmc:                   # HBM-side bytes of one training step, measured (not by construction)
            tr_k, tr_src = pmc_traffic(mode="train")
            if tr_k:
                per_step = {k: v["traffic_bytes"] * v["launches"] / (3 + 10) for k, v in tr_k.items()}   # child: --steps 10, 3 warm-up
                total = sum(per_step.values())
                train["roofline"].update({"traffic": total, "traffic_source": tr_src + "; sum over the lrf kernels of one step "
                                                                                 "(the optimiser and the few ATen elementwise kernels not included)",
                                          "achieved": total / (dtt / t_steps) / 1e9, "frac": total / (dtt / t_steps) / 1e9 / HBM_PEAK_GBS,
                                          "by_construction_bytes": train_bytes,
                                          "traffic_top": dict(sorted(per_step.items(), key=lambda kv: -kv[1])[:6]),
                                          "note": "achieved = PMC traffic of the step's kernels / step time; by_construction_bytes = "
                                                  "materialised rows + feature / gradient buffers + Adam"})
            else:
                train["roofline"]["traffic"] = None
                train["roofline"]["traffic_source"] = tr_src
        out = {"metric": "rays/sec (4096-ray batch, 512 samples, 300^3 grid)", "value": value,
               "unit": "rays/s", "n_gpus": world, "n_ranks_seen": (dist.get_world_size() if ddp else 1), "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32 (colour network: split-bf16 3-term products on v_mfma_f32_32x32x16_bf16, fp32 accumulate, <= 3e-5 vs the exact-fp32 engine; gathers and compositing fp32)",
               "data": "synthetic",
               "config": {"workload": "configs[1]: single 300^3 TensorVMSplit, 4096 rays x 512 samples "
                                      "per GPU, full density+appearance+MLP render, eval forward",
                          "rays_per_gpu": R_PER_GPU, "samples_per_ray": S, "grid": GRID,
                          "parallelism": f"ray-shard x{world}", "mlp_engine": field.mlp_engine},
               "roofline": roofline, "train_step": train}
        if not args.no_baselines and world == 1:              # baselines and extra workloads are an N=1 report
            with torch.no_grad():
                field.mlp_engine = "f32"
                d32 = timed(fwd, 20, 3, sync)
                field.mlp_engine = "bf16x3"
            work = {"exact_f32_engine": {"rays_per_s": R_PER_GPU * 20 / d32, "ms_per_step": d32 / 20 * 1e3,
                                         "what": "same batch, colour MLP on v_mfma_f32_16x16x4_f32 (LRF_FLAG_MLP_F32)"}}
            try:
                wf = walls_field(dev)
                wf.updateAlphaMask((GRID // 2,) * 3)
                wf.early_term_T = 1e-9                        # the opt-in (default 0 = reference semantics): north_star's early termination
                with torch.no_grad():
                    wfwd = lambda: wf(rays, white_bg=True, is_train=False, N_samples=N_SAMPLES_ARG)   # noqa: E731
                    d_on = timed(wfwd, 20, 3, sync)
                    p_on = kernel_profile(wf, rays, z)
                    wf.early_term_T = 0.0
                    d_off = timed(wfwd, 20, 3, sync)
                    p_off = kernel_profile(wf, rays, z)
                work["trained_like"] = {
                    "what": "300^3 field, empty space inside a box of dense walls + device-rebuilt alpha mask, same 4096 x 512 batch; early termination opted in (early_term_T = 1e-9; the class default is 0 = every sample evaluated)",
                    "rays_per_s": R_PER_GPU * 20 / d_on, "ms_per_step": d_on / 20 * 1e3,
                    "k_march_ms": p_on["march_ms"], "shaded_fraction": p_on["n_shaded"] / (R_PER_GPU * S),
                    "without_early_termination": {"rays_per_s": R_PER_GPU * 20 / d_off, "ms_per_step": d_off / 20 * 1e3,
                                                  "k_march_ms": p_off["march_ms"]}}
                del wf
            except Exception as e:                           # noqa: BLE001
                work["trained_like"] = {"error": repr(e)}
            try:                                             # a non-default colour network (opt.py:148-157): the generic fp32 engine
                torch.manual_seed(0)
                gf = TensorVMSplit(torch.device("cpu"), 2 * torch.tensor([[-1.0, -1, -1], [1, 1, 1]]), [GRID] * 3,
                                   **{**FIELD_KW, "view_pe": 2, "fea_pe": 2}).to(dev)
                with torch.no_grad():
                    gfw = lambda: gf(rays, white_bg=True, is_train=False, N_samples=N_SAMPLES_ARG)   # noqa: E731
                    dg = timed(gfw, 5, 2, sync)
                ggr, ggd = torch.randn(R_PER_GPU, 3, device=dev), torch.randn(R_PER_GPU, device=dev)

                def gfb():
                    for p_ in gf.parameters():
                        p_.grad = None
                    a_, b_ = gf(rays, white_bg=True, is_train=True, N_samples=N_SAMPLES_ARG)
                    ((a_ * ggr).sum() + (b_ * ggd).sum()).backward()
                dgb = timed(gfb, 3, 1, sync)
                work["nondefault_network"] = {
                    "what": "same grid and batch, view_pe = fea_pe = 2 (MLPRender_Fea_late_view with positional encodings): "
                            "csrc/lrf_generic.inl, fp32 tile GEMMs on v_mfma_f32_16x16x4_f32 -- supported, not tuned",
                    "rays_per_s": R_PER_GPU * 5 / dg, "ms_per_step": dg / 5 * 1e3, "forward_backward_ms": dgb / 3 * 1e3}
                del gf
            except Exception as e:                           # noqa: BLE001
                work["nondefault_network"] = {"error": repr(e)}
            try:                                             # the same field on a 16 x larger batch (an eval image is rendered this way)
                big = make_rays(16 * R_PER_GPU, 7).to(dev)
                with torch.no_grad():
                    fb = lambda: field(big, white_bg=True, is_train=False, N_samples=N_SAMPLES_ARG)   # noqa: E731
                    per = sorted(timed(fb, 1, 1 if i == 0 else 0, sync) for i in range(9))     # each step on its own: the median
                    d_big = per[4]                                                                  # is not moved by a one-off stall
                work["batch_65536"] = {"what": "same field and sample count, 65536 rays per call: four chunks of 16384 rays alternating over two "
                                               "streams (lrf_render_fwd's large-batch mode; three launches per chunk: the tile offsets of this many "
                                               "rays do not fit in LDS beside the weight image); median of 9 single steps",
                                       "rays_per_s": 16 * R_PER_GPU / d_big, "ms_per_step": d_big * 1e3, "ms_min_max": [per[0] * 1e3, per[-1] * 1e3]}
                del big
            except Exception as e:                           # noqa: BLE001
                work["batch_65536"] = {"error": repr(e)}
            try:                                             # the headline's batches, alternating over two streams (a workspace per stream)
                sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
                rays_b = make_rays(R_PER_GPU, 9).to(dev)
                state = {"k": 0}

                def two():
                    state["k"] ^= 1
                    with torch.cuda.stream(sa if state["k"] else sb):
                        field(rays if state["k"] else rays_b, white_bg=True, is_train=False, N_samples=N_SAMPLES_ARG)
                with torch.no_grad():
                    sa.wait_stream(torch.cuda.current_stream(dev)); sb.wait_stream(torch.cuda.current_stream(dev))
                    per2 = sorted(timed(two, 40, 6 if i == 0 else 0, sync) / 40 for i in range(5))
                work["two_streams_4096"] = {"what": "the headline's 4096-ray batches issued alternately on two streams (TensorVMSplit keeps a workspace per "
                                                    "stream): k_march of one batch and the tail of its k_shade3 run beside the other stream's colour kernel; "
                                                    "not the headline (one stream, one batch at a time); median of 5 x 40 batches",
                                            "rays_per_s": R_PER_GPU / per2[2], "ms_per_batch": per2[2] * 1e3, "ms_min_max": [per2[0] * 1e3, per2[-1] * 1e3]}
            except Exception as e:                           # noqa: BLE001
                work["two_streams_4096"] = {"error": repr(e)}
            try:
                lt, ray_ids, view_ids, bw = config3_scene(dev)
                with torch.no_grad():
                    c3 = lambda: lt(ray_ids, view_ids, 64, 48, is_train=False, blending_weights=bw, chunk=4096)   # noqa: E731
                    d3 = timed(c3, 20, 3, sync)
                    # per-kernel table: every field on the rays the scene forward hands it (same kernels, HIP events)
                    from localrf_amd.scene_ops import scene_rays
                    ids_d = ray_ids.to(dev)
                    rays4, _, _ = scene_rays(ids_d, lt.get_cam2world(view_ids), torch.stack(list(lt.world2rf), 0), lt.focal(64),
                                             lt.center(64, 48), ids_d.shape[0] // len(view_ids), 64, 48, False)
                    per_field = []
                    for k, fk in enumerate(lt.tensorfs):
                        zk = fk.z_schedule(False, -1, dev).contiguous()
                        pk = kernel_profile(fk, rays4[k].contiguous(), zk, reps=10)
                        per_field.append({"k_march_ms": pk["march_ms"], "k_shade3_ms": pk["shade_ms"], "n_shaded": pk["n_shaded"]})
                    t0 = time.perf_counter()
                    for _ in range(50):
                        c3()
                    host_ms = (time.perf_counter() - t0) / 50 * 1e3
                    torch.cuda.synchronize(dev)
                work["config3_4x300"] = {"what": "configs[2]: LocalTensorfs, 4 blended 300^3 fields, 4096 rays, default S=344, "
                                                 "ids and blending weights handed over on the host; one native call (lrf_scene_fwd)",
                                         "rays_per_s": 4096 * 20 / d3, "ms_per_step": d3 / 20 * 1e3,
                                         "host_enqueue_ms": host_ms, "kernels_per_field": per_field,
                                         "kernel_sum_ms": sum(q["k_march_ms"] + q["k_shade3_ms"] for q in per_field)}
                del lt
            except Exception as e:                           # noqa: BLE001
                work["config3_4x300"] = {"error": repr(e)}
            try:
                work.update(size_workloads(dev, sync, use_pmc=not args.no_pmc))
            except Exception as e:                           # noqa: BLE001
                work["fwd_500"] = {"error": repr(e)}
            try:
                work["geometric_losses"] = geometric_losses_workload(dev)
            except Exception as e:                           # noqa: BLE001
                work["geometric_losses"] = {"error": repr(e)}
            try:
                work["progressive_loop"] = progressive_loop_workload(dev)
            except Exception as e:                           # noqa: BLE001
                work["progressive_loop"] = {"error": repr(e)}
            out["workloads"] = work
            sd = field.state_dict()
            out["torch_rocm_baseline"] = torch_rocm_baseline(sd, rays)
            try:
                out["torch_rocm_baseline"]["train_step"] = torch_rocm_train_baseline(sd, rays, gr, gd)
                out["train_step"]["speedup_vs_torch_rocm"] = out["torch_rocm_baseline"]["train_step"]["ms_per_step"] / out["train_step"]["ms_per_step"]
            except Exception as e:                           # noqa: BLE001
                out["torch_rocm_baseline"]["train_step"] = {"error": repr(e)}
            out["cpu_baseline"] = cpu_baseline(sd, rays_cpu)
            out["speedup_vs_torch_rocm"] = value / world / out["torch_rocm_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if ddp:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
