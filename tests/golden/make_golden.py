"""Generate tests/golden/*.npz by running the REAL reference (read-only import from
/root/reference) on CPU.  Runs only in the build container; the fixtures it writes are
what travels.  Usage:  python tests/golden/make_golden.py

The reference needs five third-party modules it never calls on this path
(kornia, cv2, torchvision, plyfile, skimage): they are stubbed in sys.modules.
Recorded with: see `meta` in each file (torch version, seeds).
"""
import contextlib
import io
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/localTensoRF"


def import_reference():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    stub("kornia", create_meshgrid=None)
    stub("cv2", COLORMAP_JET=2)
    tv = stub("torchvision")
    tv.transforms = stub("torchvision.transforms")
    stub("plyfile")
    sk = stub("skimage")
    sk.measure = stub("skimage.measure")
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from models.tensoRF import TensorVMSplit            # noqa
    from models.tensorBase import AlphaGridMask         # noqa
    from local_tensorfs import LocalTensorfs            # noqa
    return TensorVMSplit, AlphaGridMask, LocalTensorfs


FIELD_KW = dict(density_n_comp=[8, 8, 8], appearance_n_comp=[24, 24, 24], app_dim=27,
                shadingMode="MLP_Fea_late_view", near_far=[0.1, 1e3], density_shift=-5,
                alphaMask_thres=1e-4, distance_scale=25, rayMarch_weight_thres=1e-3,
                pos_pe=0, view_pe=0, fea_pe=0, featureC=128, step_ratio=0.5,
                fea2denseAct="softplus")


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def make_field(TensorVMSplit, grid, seed, scale_density=1.0, **over):
    torch.manual_seed(seed)
    kw = dict(FIELD_KW)
    kw.update(over)
    aabb = 2 * torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    f = quiet(TensorVMSplit, "cpu", aabb, list(grid), **kw)
    if scale_density != 1.0:                 # make density interesting (opaque-ish regions)
        with torch.no_grad():
            for p in f.density_plane:
                p.mul_(scale_density)
    return f


def make_rays(R, seed, pinhole=False):
    g = torch.Generator().manual_seed(seed)
    o = 0.05 * torch.randn(R, 3, generator=g)
    d = torch.randn(R, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    if pinhole:                              # un-normalised directions, |d| > 1
        d = d / d[:, 2:3].abs().clamp(min=0.2)
    return torch.cat([o, d], -1)


def sd_np(f):
    out = {k: v.detach().cpu().numpy() for k, v in f.state_dict().items()}
    return out


def save(name, **arrs):
    meta = f"torch={torch.__version__};numpy={np.__version__};ref=facebookresearch/localrf@v1"
    np.savez_compressed(os.path.join(HERE, name), meta=np.array(meta), **arrs)
    print("wrote", name, len(arrs), "arrays")


def pack_field(prefix, f):
    return {f"{prefix}{k}": v for k, v in sd_np(f).items()}


def case_field(TensorVMSplit, AlphaGridMask, name, grid, R, N_samples, seed, floater=0.0,
               pinhole=False, mask=False, scale_density=1.0, store_field=True, **over):
    f = make_field(TensorVMSplit, grid, seed, scale_density, **over)
    if mask:
        g = np.random.default_rng(seed + 7)
        # blocky occupancy (3^3-voxel blocks, ~45% empty) so trilinear "alpha > 0" really culls
        zyx = (grid[2] // 2, grid[1] // 2, grid[0] // 2)
        coarse = (g.random(tuple((n + 2) // 3 for n in zyx)) > 0.45).astype(np.float32)
        vol = np.kron(coarse, np.ones((3, 3, 3), np.float32))[:zyx[0], :zyx[1], :zyx[2]].copy()
        f.alphaMask = AlphaGridMask("cpu", f.aabb, torch.from_numpy(vol))
    rays = make_rays(R, seed + 1, pinhole)
    with torch.no_grad():
        vd = rays[:, 3:6] / rays[:, 3:6].norm(dim=-1, keepdim=True)
        xyz, z, _ = f.sample_ray_contracted(rays[:, :3], vd, is_train=False, N_samples=N_samples)
        u = f.normalize_coord(xyz)
        sig_feat = f.compute_densityfeature(u[:4].reshape(-1, 3))
        app_feat = f.compute_appfeature(u[:4].reshape(-1, 3))
        rgb, depth = f(rays.clone(), white_bg=True, is_train=False, N_samples=N_samples,
                       floater_thresh=floater)
    arrs = dict(rays=rays.numpy(), z=z[0].numpy(), N_samples=np.array(N_samples),
                floater=np.array(floater, np.float32), xyz0=xyz[:4].numpy(),
                sig_feat=sig_feat.numpy(), app_feat=app_feat.numpy(),
                rgb=rgb.numpy(), depth=depth.numpy(), grid=np.array(grid), seed=np.array(seed),
                scale_density=np.array(scale_density, np.float32),
                nSamples=np.array(f.nSamples), stepSize=np.array(float(f.stepSize)))
    if store_field:
        arrs.update(pack_field("f.", f))
    else:                                    # big field: regenerate from seed, pin by checksum
        arrs["field_sum"] = np.array([float(sum(v.double().abs().sum() for v in f.state_dict().values()))])
    save(name, **arrs)
    return f


def case_train_grad(TensorVMSplit, name, grid, R, N_samples, seed):
    """Train-mode forward with recorded jitter + autograd gradients (SURVEY.md s4 'gradient')."""
    f = make_field(TensorVMSplit, grid, seed, scale_density=3.0)
    rays = make_rays(R, seed + 1, pinhole=True).requires_grad_(True)
    h = N_samples // 6
    torch.manual_seed(seed + 2)
    U = torch.rand(1, h)
    U2 = torch.rand(1, h)
    torch.manual_seed(seed + 2)              # same stream -> forward draws (U, U2)
    rgb, depth = f(rays, white_bg=True, is_train=True, N_samples=N_samples)
    g = torch.Generator().manual_seed(seed + 3)
    g_rgb = torch.randn(R, 3, generator=g)
    g_depth = torch.randn(R, generator=g)
    loss = (rgb * g_rgb).sum() + (depth * g_depth).sum()
    params = dict(f.named_parameters())
    params = {k: v for k, v in params.items() if v.requires_grad}
    grads = torch.autograd.grad(loss, list(params.values()) + [rays], allow_unused=True)
    arrs = dict(rays=rays.detach().numpy(), U=U[0].numpy(), U2=U2[0].numpy(),
                N_samples=np.array(N_samples), rgb=rgb.detach().numpy(),
                depth=depth.detach().numpy(), g_rgb=g_rgb.numpy(), g_depth=g_depth.numpy(),
                grid=np.array(grid))
    for (k, _), gr in zip(list(params.items()) + [("rays", None)], grads):
        arrs[f"grad.{k}"] = (gr if gr is not None else torch.zeros(1)).numpy()
    arrs.update(pack_field("f.", f))
    save(name, **arrs)


def case_pe(TensorVMSplit, name, grid, R, N_samples, seed, **over):
    """A non-default MLPRender_Fea_late_view configuration (opt.py:148-157: view_pe / fea_pe / featureC; tensorBase.py:14-21,
    97-135): eval forward with and without the feature encodings (refine), train-mode forward with recorded jitter +
    autograd gradients."""
    f = make_field(TensorVMSplit, grid, seed, scale_density=3.0, **over)
    rays = make_rays(R, seed + 1, pinhole=True).requires_grad_(True)
    with torch.no_grad():
        rgb_e, depth_e = f(rays.detach().clone(), white_bg=True, is_train=False, N_samples=N_samples)
        rgb_n, depth_n = f(rays.detach().clone(), white_bg=True, is_train=False, N_samples=N_samples, refine=False)
    h = N_samples // 6
    torch.manual_seed(seed + 2)
    U = torch.rand(1, h)
    U2 = torch.rand(1, h)
    torch.manual_seed(seed + 2)              # same stream -> forward draws (U, U2)
    rgb, depth = f(rays, white_bg=True, is_train=True, N_samples=N_samples)
    g = torch.Generator().manual_seed(seed + 3)
    g_rgb = torch.randn(R, 3, generator=g)
    g_depth = torch.randn(R, generator=g)
    loss = (rgb * g_rgb).sum() + (depth * g_depth).sum()
    params = {k: v for k, v in f.named_parameters() if v.requires_grad}
    grads = torch.autograd.grad(loss, list(params.values()) + [rays], allow_unused=True)
    arrs = dict(rays=rays.detach().numpy(), U=U[0].numpy(), U2=U2[0].numpy(), N_samples=np.array(N_samples),
                rgb=rgb.detach().numpy(), depth=depth.detach().numpy(), g_rgb=g_rgb.numpy(), g_depth=g_depth.numpy(),
                rgb_eval=rgb_e.numpy(), depth_eval=depth_e.numpy(), rgb_eval_norefine=rgb_n.numpy(), depth_eval_norefine=depth_n.numpy(),
                grid=np.array(grid), view_pe=np.array(over.get("view_pe", 0)), fea_pe=np.array(over.get("fea_pe", 0)),
                featureC=np.array(over.get("featureC", 128)))
    for (k, _), gr in zip(list(params.items()) + [("rays", None)], grads):
        arrs[f"grad.{k}"] = (gr if gr is not None else torch.zeros(1)).numpy()
    arrs.update(pack_field("f.", f))
    save(name, **arrs)


def case_local(LocalTensorfs, name, grid, seed):
    """LocalTensorfs.forward, 4 blended fields, exposure on."""
    torch.manual_seed(seed)
    aabb = 2 * torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    W, H = 32, 24
    lt = quiet(LocalTensorfs, fov=85.6, n_init_frames=5, n_overlap=3, WH=(W, H),
               n_iters_per_frame=600, n_iters_reg=100, lr_R_init=5e-3, lr_t_init=5e-4,
               lr_i_init=0, lr_exposure_init=1e-3, rf_lr_init=0.02, rf_lr_basis=1e-3,
               lr_decay_target_ratio=0.1, N_voxel_list={}, update_AlphaMask_list=[],
               camera_prior=None, device="cpu", lr_upsample_reset=True,
               aabb=aabb, gridSize=list(grid), **FIELD_KW)
    g = torch.Generator().manual_seed(seed + 1)
    for _ in range(3):
        for _ in range(3):
            quiet(lt.append_frame)
            with torch.no_grad():
                lt.t_c2w[-1].add_(0.05 * torch.randn(3, generator=g))
                lt.r_c2w[-1].add_(0.05 * torch.randn(3, 2, generator=g))
                lt.exposure[-1].add_(0.05 * torch.randn(3, 3, generator=g))
        quiet(lt.append_rf, 3)
    with torch.no_grad():
        for f in lt.tensorfs:
            for p in f.density_plane:
                p.mul_(3.0)
    n_frames = len(lt.r_c2w)
    view_ids = torch.tensor([2, 7, 11, n_frames - 1])
    per = 48
    ray_ids = torch.randint(0, W * H, (view_ids.numel() * per,), generator=g)
    bw = torch.tensor([[.1, .2, .3, .4]]).repeat(view_ids.numel(), 1)
    with torch.no_grad():
        rgbs, depths, dirs, ij = lt(ray_ids, view_ids, W, H, is_train=False,
                                    blending_weights=bw.clone(), chunk=4096, floater_thresh=0.0)
        rgbs_t, depths_t, _, _ = lt(ray_ids, view_ids, W, H, is_train=False, chunk=4096,
                                    test_id=True)
    arrs = dict(ray_ids=ray_ids.numpy(), view_ids=view_ids.numpy(), W=np.array(W), H=np.array(H),
                bw=bw.numpy(), rgbs=rgbs.numpy(), depths=depths.numpy(), dirs=dirs.numpy(),
                ij=ij.numpy(), rgbs_testid=rgbs_t.numpy(), depths_testid=depths_t.numpy(),
                grid=np.array(grid), n_fields=np.array(len(lt.tensorfs)),
                nSamples=np.array([f.nSamples for f in lt.tensorfs]))
    arrs.update({f"lt.{k}": v.detach().numpy() for k, v in lt.state_dict().items()})
    save(name, **arrs)


def case_local_train(LocalTensorfs, name, grid, seed):
    """LocalTensorfs.forward in train mode (recorded jitter) + autograd gradients of poses,
    intrinsics, exposure, world2rf and the active field: pins lrf_scene_rays/_blend backward."""
    torch.manual_seed(seed)
    aabb = 2 * torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    W, H = 40, 30
    lt = quiet(LocalTensorfs, fov=85.6, n_init_frames=5, n_overlap=3, WH=(W, H),
               n_iters_per_frame=600, n_iters_reg=100, lr_R_init=5e-3, lr_t_init=5e-4,
               lr_i_init=1e-3, lr_exposure_init=1e-3, rf_lr_init=0.02, rf_lr_basis=1e-3,
               lr_decay_target_ratio=0.1, N_voxel_list={}, update_AlphaMask_list=[],
               camera_prior=None, device="cpu", lr_upsample_reset=True,
               aabb=aabb, gridSize=list(grid), **FIELD_KW)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for i in range(len(lt.r_c2w)):
            lt.t_c2w[i].add_(0.05 * torch.randn(3, generator=g))
            lt.r_c2w[i].add_(0.05 * torch.randn(3, 2, generator=g))
            lt.exposure[i].add_(0.2 * torch.randn(3, 3, generator=g))   # large: some channels clamp
        lt.exposure[2].mul_(1.9)                     # pushes part of view 2 above 1: clamp mask
        lt.center_rel.add_(0.02 * torch.randn(2, generator=g))
        lt.focal_offset.add_(0.03)
        for p in lt.tensorfs[-1].density_plane:
            p.mul_(3.0)
    # 4 views, not 3: with exactly 3 views the reference's dim-less torch.cross (utils/utils.py:386)
    # picks dim 0 (across views) and builds non-orthonormal rotations -- a quirk not reproduced.
    view_ids = torch.tensor([0, 2, 3, 4])
    per = 40
    ray_ids = torch.randint(0, W * H, (view_ids.numel() * per,), generator=g)
    h = lt.tensorfs[-1].nSamples // 6
    torch.manual_seed(seed + 2)
    U, U2 = torch.rand(1, h), torch.rand(1, h)
    torch.manual_seed(seed + 2)                      # forward draws (U, U2) from the same stream
    rgbs, depths, dirs, ij = lt(ray_ids, view_ids, W, H, is_train=True, white_bg=True)
    R = ray_ids.numel()
    g_rgb, g_depth = torch.randn(R, 3, generator=g), torch.randn(R, generator=g)
    g_dirs = torch.randn(R, 3, generator=g)
    loss = (rgbs * g_rgb).sum() + (depths * g_depth).sum() + (dirs * g_dirs).sum()
    loss.backward()
    arrs = dict(ray_ids=ray_ids.numpy(), view_ids=view_ids.numpy(), W=np.array(W), H=np.array(H),
                U=U[0].numpy(), U2=U2[0].numpy(), rgbs=rgbs.detach().numpy(), depths=depths.detach().numpy(),
                dirs=dirs.detach().numpy(), ij=ij.numpy(), g_rgb=g_rgb.numpy(), g_depth=g_depth.numpy(),
                g_dirs=g_dirs.numpy(), grid=np.array(grid), nSamples=np.array(lt.tensorfs[-1].nSamples),
                clamped=np.array(float(((rgbs <= 0) | (rgbs >= 1)).float().mean())))
    for k, p in lt.named_parameters():
        if p.grad is not None:
            arrs[f"grad.{k}"] = p.grad.numpy()
    arrs.update({f"lt.{k}": v.detach().numpy() for k, v in lt.state_dict().items()})
    save(name, **arrs)


# ----------------------------------------------------------------------------- round 2 cases
def field_checksum(sd):
    return np.array([float(sum(v.double().abs().sum() for v in sd.values() if v.dtype.is_floating_point))])


def case_config2(TensorVMSplit, name):
    """BASELINE.json configs[1] at full size: 300^3 field (seed 0, regenerated by the tests from the
    seed and pinned by checksum), ALL 4096 bench rays x 512 samples (N_samples=1536).  Also records,
    per ray, how close the reference's own weights come to the shading threshold (tensorBase.py:622):
    a ray whose colour differs by more than the tolerance must be explained by such a sample."""
    import models.tensorBase as tb
    f = make_field(TensorVMSplit, (300, 300, 300), 0)
    rays = make_rays(4096, 1)
    seen = {}
    orig = tb.alpha2weights

    def spy(alpha):
        w, T = orig(alpha)
        seen["w"] = w.detach().clone()
        return w, T
    tb.alpha2weights = spy
    try:
        with torch.no_grad():
            rgb, depth = f(rays.clone(), white_bg=True, is_train=False, N_samples=1536)
    finally:
        tb.alpha2weights = orig
    w = seen["w"]
    thres = f.rayMarch_weight_thres
    save(name, rays=rays.numpy(), rgb=rgb.numpy(), depth=depth.numpy(), N_samples=np.array(1536),
         seed=np.array(0), grid=np.array([300, 300, 300]), field_sum=field_checksum(f.state_dict()),
         near_thres=(w - thres).abs().amin(-1).numpy(), n_shaded=np.array(int((w > thres).sum())),
         acc=w.sum(-1).numpy())


def pack_grad(arrs, key, g, rng, keep=16384):
    """Full tensor when small; for the big plane gradients a seeded random subset + max + L2."""
    g = g.detach().reshape(-1)
    arrs[f"gmax.{key}"] = np.array(float(g.abs().max()))
    arrs[f"gl2.{key}"] = np.array(float(g.double().norm()))
    if g.numel() <= 4 * keep:
        arrs[f"grad.{key}"] = g.numpy()
    else:
        idx = np.sort(rng.choice(g.numel(), keep, replace=False))
        top = torch.topk(g.abs(), 2048).indices.numpy()          # and the largest entries
        idx = np.unique(np.concatenate([idx, top]))
        arrs[f"gidx.{key}"] = idx.astype(np.int64)
        arrs[f"grad.{key}"] = g.numpy()[idx]


def case_train_grad_big(TensorVMSplit, name, grid=(128, 128, 128), R=512, seed=5):
    """Train-mode forward (recorded jitter, default sample count) + autograd gradients at 128^3;
    field regenerated from the seed by the tests."""
    f = make_field(TensorVMSplit, grid, seed, scale_density=3.0)
    rays = make_rays(R, seed + 1, pinhole=True).requires_grad_(True)
    h = f.nSamples // 6
    torch.manual_seed(seed + 2)
    U, U2 = torch.rand(1, h), torch.rand(1, h)
    torch.manual_seed(seed + 2)
    rgb, depth = f(rays, white_bg=True, is_train=True, N_samples=-1)
    g = torch.Generator().manual_seed(seed + 3)
    g_rgb, g_depth = torch.randn(R, 3, generator=g), torch.randn(R, generator=g)
    loss = (rgb * g_rgb).sum() + (depth * g_depth).sum()
    params = {k: v for k, v in f.named_parameters() if v.requires_grad}
    grads = torch.autograd.grad(loss, list(params.values()) + [rays], allow_unused=True)
    arrs = dict(rays=rays.detach().numpy(), U=U[0].numpy(), U2=U2[0].numpy(), nSamples=np.array(f.nSamples),
                rgb=rgb.detach().numpy(), depth=depth.detach().numpy(), g_rgb=g_rgb.numpy(),
                g_depth=g_depth.numpy(), grid=np.array(grid), seed=np.array(seed),
                scale_density=np.array(3.0, np.float32), field_sum=field_checksum(f.state_dict()))
    rng = np.random.default_rng(seed)
    for (k, _), gr in zip(list(params.items()) + [("rays", None)], grads):
        pack_grad(arrs, k, gr if gr is not None else torch.zeros(1), rng)
    save(name, **arrs)


def case_ladder(TensorVMSplit, name, seed=45, R=128):
    """The upsample ladder of train.py (train.py:275-288 with opt.py:61-69: 64^3 -> 101, 161, 255, 404 -> 640^3,
    resolutions through utils.N_to_reso as local_tensorfs.py:251-253 does): one seeded 64^3 field taken through
    upsample_volume_grid five times; after every stage an eval render at that stage's own default sample count."""
    from utils.utils import N_to_reso
    n_list = torch.round(torch.exp(torch.linspace(np.log(64 ** 3), np.log(640 ** 3), 6))).long().tolist()[1:]
    n_list = [round(n ** (1 / 3)) ** 3 for n in n_list]
    f = make_field(TensorVMSplit, (64, 64, 64), seed, scale_density=3.0)
    rays = make_rays(R, seed + 1, pinhole=True)
    arrs = dict(rays=rays.numpy(), seed=np.array(seed), n_voxels=np.array(n_list, np.int64), scale_density=np.array(3.0, np.float32),
                field_sum=field_checksum(f.state_dict()))
    for i, n in enumerate(n_list):
        reso = N_to_reso(n, f.aabb)
        quiet(f.upsample_volume_grid, reso)
        with torch.no_grad():
            rgb, depth = f(rays.clone(), white_bg=True, is_train=False, N_samples=-1)
        arrs[f"reso{i}"] = np.array(reso)
        arrs[f"nSamples{i}"] = np.array(f.nSamples)
        arrs[f"rgb{i}"] = rgb.numpy()
        arrs[f"depth{i}"] = depth.numpy()
        arrs[f"field_sum{i}"] = field_checksum(f.state_dict())
        print("ladder stage", i, reso, f.nSamples, flush=True)
    save(name, **arrs)


def case_sample_ray(TensorVMSplit, name):
    """TensorBase.sample_ray (tensorBase.py:396-417; dead code on train.py's path, named by north_star)."""
    f = make_field(TensorVMSplit, (32, 32, 32), 5)
    rays = make_rays(64, 9, pinhole=True)
    rays[0, 4] = 0.0                                     # exercises the d == 0 branch
    rays[1, :3] = torch.tensor([2.5, 0.1, -0.2])         # origin outside the box
    with torch.no_grad():
        pts, t, inside = f.sample_ray(rays[:, :3], rays[:, 3:], is_train=False, N_samples=50)
        torch.manual_seed(77)
        U = torch.rand(64, 1)
        torch.manual_seed(77)
        pts_j, t_j, inside_j = f.sample_ray(rays[:, :3], rays[:, 3:], is_train=True, N_samples=50)
    save(name, rays=rays.numpy(), pts=pts.numpy(), t=t.numpy(), inside=inside.numpy(), U=U[:, 0].numpy(),
         pts_j=pts_j.numpy(), t_j=t_j.numpy(), inside_j=inside_j.numpy(), N_samples=np.array(50),
         stepSize=np.array(float(f.stepSize)), aabb=f.aabb.numpy(), near_far=np.array(f.near_far, np.float32))


def case_reg(TensorVMSplit, name, grid=(20, 24, 28), seed=51):
    """density_L1 (tensoRF.py:83-92), TV_loss_density / TV_loss_app (tensoRF.py:94-110) with the
    reference's TVLoss module (utils/utils.py:293-309): values and autograd gradients."""
    from utils.utils import TVLoss
    f = make_field(TensorVMSplit, grid, seed, scale_density=20.0)
    arrs = dict(grid=np.array(grid), seed=np.array(seed), scale_density=np.array(20.0, np.float32),
                field_sum=field_checksum(f.state_dict()))
    reg = TVLoss()
    for key, fn in (("l1", lambda: f.density_L1()), ("tv_density", lambda: f.TV_loss_density(reg)),
                    ("tv_app", lambda: f.TV_loss_app(reg))):
        for p in f.parameters():
            p.grad = None
        out = fn()
        out = out.mean() if out.dim() else out
        out.backward()
        arrs[f"{key}.value"] = np.array(float(out))
        for n, p in f.named_parameters():
            if p.grad is not None:
                arrs[f"{key}.grad.{n}"] = p.grad.numpy().copy()
    save(name, **arrs)


def case_alpha_mask(TensorVMSplit, name, grid=(40, 36, 44), seed=61):
    """updateAlphaMask (tensorBase.py:501-536) on a field with real structure (density_shift -10 and
    strong planes, so a good part of the lattice falls below alphaMask_thres), twice: the second
    rebuild goes through the first mask (compute_alpha, :538-558)."""
    f = make_field(TensorVMSplit, grid, seed, scale_density=60.0, density_shift=-10)
    g1 = tuple(int(x) // 2 for x in grid)
    quiet(f.updateAlphaMask, g1)
    m1 = f.alphaMask.alpha_volume.detach().numpy()[0, 0].copy()
    g2 = tuple(int(x) * 3 // 4 for x in grid)
    quiet(f.updateAlphaMask, g2)
    m2 = f.alphaMask.alpha_volume.detach().numpy()[0, 0].copy()
    rays = make_rays(64, seed + 1)
    with torch.no_grad():
        rgb, depth = f(rays.clone(), white_bg=True, is_train=False, N_samples=120)
    print("alpha mask kept fractions", m1.mean(), m2.mean())
    assert 0.05 < m1.mean() < 0.95 and 0.05 < m2.mean() < 0.95
    save(name, grid=np.array(grid), seed=np.array(seed), scale_density=np.array(60.0, np.float32),
         density_shift=np.array(-10.0, np.float32), field_sum=field_checksum({k: v for k, v in f.state_dict().items() if "alphaMask" not in k}),
         g1=np.array(g1), g2=np.array(g2), m1=np.packbits(m1.astype(np.uint8)), m2=np.packbits(m2.astype(np.uint8)),
         m1_shape=np.array(m1.shape), m2_shape=np.array(m2.shape), rays=rays.numpy(), rgb=rgb.numpy(),
         depth=depth.numpy(), N_samples=np.array(120))


def case_sixd(name):
    """sixD_to_mtx (utils/utils.py:381-388) including its dim-less torch.cross: with exactly three
    views the cross product runs over the view axis."""
    import warnings
    from utils.utils import sixD_to_mtx
    arrs = {}
    g = torch.Generator().manual_seed(7)
    for V in (1, 2, 3, 4, 7):
        r = (torch.eye(3, 2)[None] + 0.3 * torch.randn(V, 3, 2, generator=g)).requires_grad_(True)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m = sixD_to_mtx(r)
        ct = torch.randn(V, 3, 3, generator=g)
        (m * ct).sum().backward()
        arrs.update({f"r{V}": r.detach().numpy(), f"m{V}": m.detach().numpy(), f"ct{V}": ct.numpy(),
                     f"g{V}": r.grad.numpy()})
    save(name, **arrs)


def case_upsample(TensorVMSplit, name, grid=(16, 20, 12), target=(24, 27, 30), seed=71):
    """TensorVMSplit.upsample_volume_grid (tensoRF.py:198-233): F.interpolate(bilinear, align_corners=True) of the 6
    planes and 6 lines + update_stepSize."""
    f = make_field(TensorVMSplit, grid, seed)
    fsum = field_checksum(f.state_dict())
    quiet(f.upsample_volume_grid, list(target))
    arrs = dict(grid=np.array(grid), target=np.array(target), seed=np.array(seed), field_sum=fsum,
                nSamples=np.array(int(f.nSamples)), stepSize=np.array(float(f.stepSize), np.float32))
    for k, v in f.state_dict().items():
        if "plane" in k or "line" in k:
            arrs[f"up.{k}"] = v.detach().numpy().copy()
    save(name, **arrs)


def case_geo_losses(name, seed=81, F=6, V=4, n=96, W=64, H=48):
    """The optical-flow and monocular-depth losses of train.py:385-423, computed with the reference's own
    utils/utils.py functions (get_fwd_bwd_cam2cams, get_pred_flow, compute_depth_loss); the lines of train.py that
    combine them are inline code there and are repeated here verbatim in meaning (masks, quantile clipping, mean)."""
    from utils.utils import compute_depth_loss, get_fwd_bwd_cam2cams, get_pred_flow, sixD_to_mtx
    g = torch.Generator().manual_seed(seed)
    r6 = torch.eye(3)[:, :2][None].repeat(F, 1, 1) + 0.08 * torch.randn(F, 3, 2, generator=g)
    r6[4] = torch.tensor([[0.36, 0.0], [0.0, 1.0], [-0.93, 0.05]])        # frame 4 is turned by ~1.2 rad: a few rays reproject behind it (z clip in pts2px)
    rot = torch.stack([sixD_to_mtx(r6[i:i + 1])[0] for i in range(F)])   # one view at a time (no dim-less cross quirk)
    trans = 0.2 * torch.randn(F, 3, 1, generator=g)
    cam2world = torch.cat([rot, trans], -1).detach().clone().requires_grad_(True)          # [F,3,4]
    # absolute view ids with a non-zero starting frame: train.py:396 compares the ABSOLUTE id with the length of the
    # cam2world slice, so here view 5 (relative 4, not the last) loses its forward mask and view 6 (the last) keeps it
    view_ids = torch.tensor([1, 3, 6, 5])
    starting_frame_id = 1
    focal = torch.tensor([60.0], requires_grad=True)
    center = torch.tensor([W * 0.5, H * 0.48], requires_grad=True)
    col = torch.randint(0, W, (V, n), generator=g)
    row = torch.randint(0, H, (V, n), generator=g)
    ij = torch.stack([col, row], -1)                                                       # [V,n,2] int64
    with torch.no_grad():
        dirs0 = torch.stack([(col + 0.5 - center[0]) / focal, -(row + 0.5 - center[1]) / focal, -torch.ones(V, n)], -1)
    directions = dirs0.clone().requires_grad_(True)
    depth_map = (0.5 + 3.5 * torch.rand(V, n, generator=g)).requires_grad_(True)
    fwd_flow = 3.0 * torch.randn(V, n, 2, generator=g)
    bwd_flow = 3.0 * torch.randn(V, n, 2, generator=g)
    fwd_mask = (torch.rand(V, n, generator=g) > 0.3).float()
    bwd_mask = (torch.rand(V, n, generator=g) > 0.3).float()
    invdepths = 0.2 + torch.rand(V, n, generator=g)

    # train.py:389-410
    fm = fwd_mask.clone()
    fm[view_ids == len(cam2world) - 1] = 0
    fwd_cam2cams, bwd_cam2cams = get_fwd_bwd_cam2cams(cam2world, view_ids - starting_frame_id)
    pts = directions * depth_map[..., None]
    pred_fwd_flow = get_pred_flow(pts, ij, fwd_cam2cams, focal, center)
    pred_bwd_flow = get_pred_flow(pts, ij, bwd_cam2cams, focal, center)
    flow_loss_arr = torch.sum(torch.abs(pred_bwd_flow - bwd_flow), dim=-1) * bwd_mask
    flow_loss_arr += torch.sum(torch.abs(pred_fwd_flow - fwd_flow), dim=-1) * fm
    flow_loss_arr[flow_loss_arr > torch.quantile(flow_loss_arr, 0.9, dim=1)[..., None]] = 0
    flow_mean = flow_loss_arr.mean()
    flow_mean.backward()
    arrs = dict(seed=np.array(seed), WH=np.array([W, H]), view_ids=view_ids.numpy(), starting_frame_id=np.array(starting_frame_id),
                cam2world=cam2world.detach().numpy(), focal=focal.detach().numpy(), center=center.detach().numpy(), ij=ij.numpy(),
                directions=directions.detach().numpy(), depth=depth_map.detach().numpy(), fwd_flow=fwd_flow.numpy(),
                bwd_flow=bwd_flow.numpy(), fwd_mask=fwd_mask.numpy(), bwd_mask=bwd_mask.numpy(), invdepths=invdepths.numpy())
    arrs.update({"flow.arr": flow_loss_arr.detach().numpy(), "flow.mean": np.array(float(flow_mean)),
                 "flow.fwd_cam2cams": fwd_cam2cams.detach().numpy(), "flow.bwd_cam2cams": bwd_cam2cams.detach().numpy(),
                 "flow.g_depth": depth_map.grad.numpy().copy(), "flow.g_dirs": directions.grad.numpy().copy(),
                 "flow.g_cam2world": cam2world.grad.numpy().copy(), "flow.g_focal": focal.grad.numpy().copy(),
                 "flow.g_center": center.grad.numpy().copy()})
    print("flow: mean", float(flow_mean), "zeroed", int((flow_loss_arr == 0).sum()), "of", V * n)

    # train.py:417-421
    depth_map.grad = None
    _, _, depth_loss_arr = compute_depth_loss(1 / depth_map.clamp(1e-6), invdepths)
    depth_loss_arr[depth_loss_arr > torch.quantile(depth_loss_arr, 0.8, dim=1)[..., None]] = 0
    depth_mean = depth_loss_arr.mean()
    depth_mean.backward()
    arrs.update({"depth.arr": depth_loss_arr.detach().numpy(), "depth.mean": np.array(float(depth_mean)),
                 "depth.g_depth": depth_map.grad.numpy().copy()})
    print("depth: mean", float(depth_mean))
    save(name, **arrs)


def build_local(LocalTensorfs, grid, seed, WH, camera_prior=None, lr_i=1e-3):
    torch.manual_seed(seed)
    aabb = 2 * torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    return quiet(LocalTensorfs, fov=85.6, n_init_frames=5, n_overlap=3, WH=WH,
                 n_iters_per_frame=600, n_iters_reg=100, lr_R_init=5e-3, lr_t_init=5e-4,
                 lr_i_init=lr_i, lr_exposure_init=1e-3, rf_lr_init=0.02, rf_lr_basis=1e-3,
                 lr_decay_target_ratio=0.1, N_voxel_list={}, update_AlphaMask_list=[],
                 camera_prior=camera_prior, device="cpu", lr_upsample_reset=True,
                 aabb=aabb, gridSize=list(grid), **FIELD_KW)


def small_state(lt):
    """Everything in the state dict except the field tensors (those are regenerated from the seed)."""
    return {f"lt.{k}": v.detach().numpy() for k, v in lt.state_dict().items() if not k.startswith("tensorfs.")}


def case_local_train_views(LocalTensorfs, name, grid, seed, view_ids, camera_prior=False):
    """LocalTensorfs train-mode forward + gradients with the field regenerated from the seed:
    (a) exactly 3 views (the torch.cross quirk of sixD_to_mtx), (b) camera priors, which make
    r_c2w [3,3] parameters (local_tensorfs.py:171-176)."""
    W, H = 40, 30
    prior = None
    g = torch.Generator().manual_seed(seed + 1)
    if camera_prior:
        rel = []
        for _ in range(8):
            a = 0.08 * torch.randn(3, generator=g)
            K = torch.tensor([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
            m = torch.eye(4)
            m[:3, :3] = torch.matrix_exp(K)
            m[:3, 3] = 0.05 * torch.randn(3, generator=g)
            rel.append(m)
        prior = {"transforms": {"fl_x": 30.0, "w": 44.0}, "rel_poses": torch.stack(rel)}
    lt = build_local(LocalTensorfs, grid, seed, (W, H), prior)
    with torch.no_grad():
        for i in range(len(lt.r_c2w)):
            lt.t_c2w[i].add_(0.05 * torch.randn(3, generator=g))
            lt.r_c2w[i].add_(0.05 * torch.randn(*lt.r_c2w[i].shape, generator=g))
            lt.exposure[i].add_(0.1 * torch.randn(3, 3, generator=g))
        for p in lt.tensorfs[-1].density_plane:
            p.mul_(3.0)
    view_ids = torch.tensor(view_ids)
    per = 40
    ray_ids = torch.randint(0, W * H, (view_ids.numel() * per,), generator=g)
    h = lt.tensorfs[-1].nSamples // 6
    torch.manual_seed(seed + 2)
    U, U2 = torch.rand(1, h), torch.rand(1, h)
    torch.manual_seed(seed + 2)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        rgbs, depths, dirs, ij = lt(ray_ids, view_ids, W, H, is_train=True, white_bg=True)
    R = ray_ids.numel()
    g_rgb, g_depth = torch.randn(R, 3, generator=g), torch.randn(R, generator=g)
    ((rgbs * g_rgb).sum() + (depths * g_depth).sum()).backward()
    arrs = dict(ray_ids=ray_ids.numpy(), view_ids=view_ids.numpy(), W=np.array(W), H=np.array(H),
                U=U[0].numpy(), U2=U2[0].numpy(), rgbs=rgbs.detach().numpy(), depths=depths.detach().numpy(),
                g_rgb=g_rgb.numpy(), g_depth=g_depth.numpy(), grid=np.array(grid), seed=np.array(seed),
                field_sum=field_checksum({k: v for k, v in lt.state_dict().items() if k.startswith("tensorfs.")}))
    if prior is not None:
        arrs["rel_poses"] = prior["rel_poses"].numpy()
    for k, p in lt.named_parameters():
        if p.grad is not None and not k.startswith("tensorfs."):
            arrs[f"grad.{k}"] = p.grad.numpy()
    rng = np.random.default_rng(seed)
    for k, p in lt.named_parameters():
        if p.grad is not None and k.startswith("tensorfs."):
            pack_grad(arrs, k, p.grad, rng, keep=4096)
    arrs.update(small_state(lt))
    save(name, **arrs)


def case_config3(LocalTensorfs, name, grid=(300, 300, 300), seed=33):
    """BASELINE.json configs[2] at full size: 4 overlapping 300^3 fields, 4096 rays, blended with
    explicit weights, exposure on.  Fields are regenerated from the seed by the tests."""
    W, H = 64, 48
    lt = build_local(LocalTensorfs, grid, seed, (W, H), lr_i=0)
    g = torch.Generator().manual_seed(seed + 1)
    for _ in range(3):
        for _ in range(3):
            quiet(lt.append_frame)
            with torch.no_grad():
                lt.t_c2w[-1].add_(0.05 * torch.randn(3, generator=g))
                lt.r_c2w[-1].add_(0.05 * torch.randn(3, 2, generator=g))
                lt.exposure[-1].add_(0.05 * torch.randn(3, 3, generator=g))
        quiet(lt.append_rf, 3)
    n_frames = len(lt.r_c2w)
    view_ids = torch.tensor([2, 7, 11, n_frames - 1])
    per = 1024
    ray_ids = torch.randint(0, W * H, (view_ids.numel() * per,), generator=g)
    bw = torch.tensor([[.1, .2, .3, .4]]).repeat(view_ids.numel(), 1)
    with torch.no_grad():
        rgbs, depths, dirs, ij = lt(ray_ids, view_ids, W, H, is_train=False,
                                    blending_weights=bw.clone(), chunk=4096, floater_thresh=0.0)
    arrs = dict(ray_ids=ray_ids.numpy(), view_ids=view_ids.numpy(), W=np.array(W), H=np.array(H), bw=bw.numpy(),
                rgbs=rgbs.numpy(), depths=depths.numpy(), grid=np.array(grid), seed=np.array(seed),
                n_fields=np.array(len(lt.tensorfs)), nSamples=np.array([f.nSamples for f in lt.tensorfs]),
                field_sum=field_checksum({k: v for k, v in lt.state_dict().items() if k.startswith("tensorfs.")}))
    arrs.update(small_state(lt))
    save(name, **arrs)


def case_trajectory(TensorVMSplit, LocalTensorfs, name, seed=91):
    """30 optimisation iterations of the REAL LocalTensorfs on CPU through tests/trajectory.py::run: photometric L1 +
    density L1 -> optimizer_step, crossing two append_frame, the switch to refining, one upsample (fresh Adam), one
    alpha-mask rebuild, the end of regularisation, append_rf and the first iterations of the second field.  Recorded:
    the initial state, the initial state of the second field, the sample distances z of every forward (the train-mode
    jitter), losses / colours / depths of every iteration and the final state.  A second run from an initial state
    perturbed by 1e-6 (relative) measures how far the reference's OWN trajectory moves under a last-bit change: the
    replay's tolerance on the final parameters is stated against that, not guessed."""
    sys.path.insert(0, os.path.dirname(HERE))
    import trajectory as tj
    import warnings
    kw = dict(FIELD_KW)
    kw.update(tj.FIELD_OVER)
    aabb = 2 * torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    view_u, ray_ids = tj.batches(seed + 3)
    target = tj.targets()

    def fresh():
        torch.manual_seed(seed)
        scene_kw = {k: (dict(v) if isinstance(v, dict) else v) for k, v in tj.SCENE_KW.items()}
        lt = quiet(LocalTensorfs, device="cpu", aabb=aabb, gridSize=list(tj.GRID), **scene_kw, **kw)
        g = torch.Generator().manual_seed(seed + 1)
        with torch.no_grad():
            for i in range(len(lt.r_c2w)):
                lt.t_c2w[i].add_(0.04 * i + 0.02 * torch.randn(3, generator=g))
                lt.r_c2w[i].add_(0.03 * torch.randn(3, 2, generator=g))
            for p in lt.tensorfs[-1].density_plane:             # a density field with structure: the alpha mask rebuilt
                p.mul_(10.0)                                    # at it 14 keeps about half of its cells
            for p in lt.tensorfs[-1].density_line:
                p.mul_(3.0)
        return lt

    z_log = []
    orig = TensorVMSplit.sample_ray_contracted

    def spy(self, *a, **k):
        out = orig(self, *a, **k)
        z_log.append(out[1][0].detach().numpy().copy())
        return out

    def one_run(lt, rf1_state=None):
        rf1 = {}

        def after_append_rf(scene):
            if rf1_state is not None:                                  # second run: same second field as the first
                scene.tensorfs[-1].load_state_dict(rf1_state)
            rf1.update({k: v.detach().clone() for k, v in scene.tensorfs[-1].state_dict().items()})
        z_log.clear()
        TensorVMSplit.sample_ray_contracted = spy
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                torch.manual_seed(seed + 2)
                log = quiet(tj.run, lt, view_u, ray_ids, target, "cpu", after_append_rf=after_append_rf)
        finally:
            TensorVMSplit.sample_ray_contracted = orig
        return log, [z.copy() for z in z_log], rf1

    lt = fresh()
    init = {k: v.detach().numpy().copy() for k, v in lt.state_dict().items()}
    log, zs, rf1 = one_run(lt)
    assert len(zs) == tj.N_ITERS, len(zs)
    final = {k: v.detach().numpy().copy() for k, v in lt.state_dict().items()}
    final_rgb, final_depth = quiet(tj.final_render, lt, "cpu")

    # the reference against itself: initial parameters moved by 1e-6 relative, same jitter, same second field
    lt2 = fresh()
    g = torch.Generator().manual_seed(seed + 5)
    with torch.no_grad():
        for p in lt2.parameters():
            if p.is_floating_point() and p.requires_grad:
                p.mul_(1 + 1e-6 * (2 * torch.rand(p.shape, generator=g) - 1))
    z_first = [z.copy() for z in zs]
    torch.manual_seed(seed + 2)
    log2, zs2, _ = one_run(lt2, rf1_state=rf1)
    assert all(np.array_equal(a, b) for a, b in zip(z_first, zs2)), "the two reference runs must share their jitter"
    final2 = {k: v.detach().numpy().copy() for k, v in lt2.state_dict().items()}
    final_rgb2, final_depth2 = quiet(tj.final_render, lt2, "cpu")

    # ... and when every gradient it computes is off by 1e-5 of that tensor's largest gradient element, which is the
    # bar the gradient parity tests hold the HIP path to (tests/test_gpu_parity.py): Adam divides by the running
    # gradient magnitude, so elements whose own gradient is that small move by a visible fraction of the learning rate
    lt3 = fresh()
    g3 = torch.Generator().manual_seed(seed + 6)

    def noisy(grad):
        return grad + (grad != 0) * 1e-5 * grad.abs().max() * (2 * torch.rand(grad.shape, generator=g3) - 1)   # exact zeros stay zero

    def hook_all(scene):
        for p_ in scene.parameters():
            if p_.requires_grad and not getattr(p_, "_noisy", False):
                p_.register_hook(noisy)
                p_._noisy = True
    hook_all(lt3)
    orig_append_rf, orig_append_frame = LocalTensorfs.append_rf, LocalTensorfs.append_frame

    def append_rf_hooked(self, *a, **k):
        out = orig_append_rf(self, *a, **k)
        if self is lt3:
            self.tensorfs[-1].load_state_dict(rf1)
            hook_all(self)
        return out

    def append_frame_hooked(self, *a, **k):
        out = orig_append_frame(self, *a, **k)
        if self is lt3:
            hook_all(self)
        return out
    LocalTensorfs.append_rf, LocalTensorfs.append_frame = append_rf_hooked, append_frame_hooked
    try:
        log3, zs3, _ = one_run(lt3, rf1_state=rf1)
    finally:
        LocalTensorfs.append_rf, LocalTensorfs.append_frame = orig_append_rf, orig_append_frame
    assert all(np.array_equal(a, b) for a, b in zip(z_first, zs3))
    final3 = {k: v.detach().numpy().copy() for k, v in lt3.state_dict().items()}

    arrs = dict(seed=np.array(seed), view_u=view_u, ray_ids=ray_ids,
                photo=np.array([r["photo"] for r in log], np.float64), l1=np.array([r["l1"] for r in log], np.float64),
                total=np.array([r["total"] for r in log], np.float64),
                regularize=np.array([r["regularize"] for r in log]), rf_iter=np.array([r["rf_iter"] for r in log]),
                n_fields=np.array([r["n_fields"] for r in log]), grid_it=np.array([r["grid"] for r in log]),
                nSamples_it=np.array([r["nSamples"] for r in log]), has_mask=np.array([r["has_mask"] for r in log]),
                can_add_rf=np.array([r["can_add_rf"] for r in log]),
                views=np.array([r["views"] for r in log]),
                rgb=np.stack([r["rgb"] for r in log]), depth=np.stack([r["depth"] for r in log]),
                photo_perturbed=np.array([r["photo"] for r in log2], np.float64))
    for it, z in enumerate(zs):
        arrs[f"z.{it}"] = z
    arrs.update({f"init.{k}": v for k, v in init.items()})
    arrs.update({f"rf1.{k}": v.numpy() for k, v in rf1.items()})
    arrs.update({f"final.{k}": v for k, v in final.items()})
    for k in final:                                                    # reference-vs-reference drift per tensor
        if final[k].dtype.kind == "f" and final[k].shape == final2[k].shape:
            arrs[f"drift.{k}"] = np.array(float(np.abs(final[k] - final2[k]).max()))
            arrs[f"drift_l2.{k}"] = np.array(float(np.linalg.norm(final[k] - final2[k]) / max(np.linalg.norm(final[k]), 1e-30)))
            d3 = np.abs(final[k] - final3[k])
            arrs[f"gdrift.{k}"] = np.array(float(d3.max()))
            arrs[f"gdrift_l2.{k}"] = np.array(float(np.linalg.norm(d3) / max(np.linalg.norm(final[k]), 1e-30)))
            arrs[f"gdrift_n.{k}"] = np.array(int((d3 > 1e-3 * np.abs(final[k]).max()).sum()))
    arrs["photo_gnoise"] = np.array([r["photo"] for r in log3], np.float64)
    arrs.update(final_rgb=final_rgb, final_depth=final_depth,
                final_rgb_drift=np.array(float(np.abs(final_rgb - final_rgb2).max())),
                final_depth_drift=np.array(float((np.abs(final_depth - final_depth2) / np.abs(final_depth)).max())))
    mask = lt.tensorfs[0].alphaMask
    arrs["mask_kept"] = np.array(float(mask.alpha_volume.mean()))
    print("trajectory: photo", np.round(arrs["photo"], 5).tolist())
    print("trajectory: l1", np.round(arrs["l1"], 6).tolist())
    print("trajectory: rf_iter", arrs["rf_iter"].tolist(), "fields", arrs["n_fields"].tolist())
    print("trajectory: grids", [tuple(g) for g in arrs["grid_it"][[0, 12, 29]]], "mask kept", float(arrs["mask_kept"]),
          "has_mask", arrs["has_mask"].tolist())
    worst = max((float(v), k) for k, v in arrs.items() if k.startswith("drift."))
    print("trajectory: final render drift rgb", float(arrs["final_rgb_drift"]), "depth rel", float(arrs["final_depth_drift"]),
          "worst l2 drift", max((float(v), k) for k, v in arrs.items() if k.startswith("drift_l2.")))
    print("trajectory: reference with 1e-5 gradient noise: worst final max drift / max|p|",
          sorted(((float(v) / float(np.abs(arrs["final." + k[7:]]).max()), float(arrs["gdrift_l2." + k[7:]]), int(arrs["gdrift_n." + k[7:]]), k)
                  for k, v in arrs.items() if k.startswith("gdrift.") and np.abs(arrs["final." + k[7:]]).max() > 0), reverse=True)[:4],
          "photo", float((np.abs(arrs["photo"] - arrs["photo_gnoise"]) / arrs["photo"]).max()))
    print("trajectory: reference-vs-perturbed-reference worst final drift", worst,
          "photo drift", float(np.abs(arrs["photo"] - arrs["photo_perturbed"]).max()))
    save(name, **arrs)


def main():
    TensorVMSplit, AlphaGridMask, LocalTensorfs = import_reference()
    only = set(sys.argv[1:])
    if only:                                    # python make_golden.py config2 sixd ...  (round-2 cases by key)
        r2 = {"config2": lambda: case_config2(TensorVMSplit, "config2_300cube.npz"),
              "train128": lambda: case_train_grad_big(TensorVMSplit, "field_128_train_grad.npz"),
              "sample_ray": lambda: case_sample_ray(TensorVMSplit, "sample_ray.npz"),
              "reg": lambda: case_reg(TensorVMSplit, "reg_losses.npz"),
              "alpha_mask": lambda: case_alpha_mask(TensorVMSplit, "alpha_mask_rebuild.npz"),
              "sixd": lambda: case_sixd("sixd_to_mtx.npz"),
              "local3": lambda: case_local_train_views(LocalTensorfs, "local_train_3views.npz", (20, 24, 28), 43, [0, 2, 4]),
              "prior": lambda: case_local_train_views(LocalTensorfs, "local_train_prior.npz", (20, 24, 28), 45, [0, 1, 3, 4], camera_prior=True),
              "config3": lambda: case_config3(LocalTensorfs, "config3_4x300.npz"),
              "upsample": lambda: case_upsample(TensorVMSplit, "upsample_grid.npz"),
              "geo": lambda: case_geo_losses("geo_losses.npz"),
              "trajectory": lambda: case_trajectory(TensorVMSplit, LocalTensorfs, "trajectory_30it.npz"),
              "train500": lambda: case_train_grad_big(TensorVMSplit, "field_500_train_grad.npz", grid=(500, 500, 500), R=512, seed=41),
              "train640": lambda: case_train_grad_big(TensorVMSplit, "field_640_train_grad.npz", grid=(640, 640, 640), R=256, seed=43),
              "ladder": lambda: case_ladder(TensorVMSplit, "ladder_64_to_640.npz"),
              "pe": lambda: (case_pe(TensorVMSplit, "field_pe_2_3_64.npz", (20, 24, 28), 64, 96, 101, fea_pe=2, view_pe=3, featureC=64),
                             case_pe(TensorVMSplit, "field_pe_0_2_128.npz", (24, 20, 22), 48, 96, 103, fea_pe=0, view_pe=2, featureC=128),
                             case_pe(TensorVMSplit, "field_pe_6_6_200.npz", (16, 18, 20), 32, 60, 105, fea_pe=6, view_pe=6, featureC=200))}
        for k in only:
            r2[k]()
        return
    # non-cubic grid: catches axis-order mistakes (first coord indexes W)
    case_field(TensorVMSplit, AlphaGridMask, "field_small_eval.npz", (20, 24, 28), 64, 96, 11,
               scale_density=3.0)
    case_field(TensorVMSplit, AlphaGridMask, "field_small_floater.npz", (20, 24, 28), 64, 96, 12,
               floater=0.5, pinhole=True, scale_density=3.0)
    case_field(TensorVMSplit, AlphaGridMask, "field_small_mask.npz", (20, 24, 28), 64, 96, 13,
               mask=True, scale_density=3.0)
    case_field(TensorVMSplit, AlphaGridMask, "field_small_default_ns.npz", (20, 24, 28), 32, -1, 14)
    # BASELINE.json configs[0]: 64^3, 256 rays x 64 samples (N_samples=192)
    case_field(TensorVMSplit, AlphaGridMask, "config1_64cube.npz", (64, 64, 64), 256, 192, 0,
               store_field=False)
    case_train_grad(TensorVMSplit, "field_small_train_grad.npz", (20, 24, 28), 48, 96, 21)
    case_local(LocalTensorfs, "local_4fields.npz", (16, 16, 16), 31)
    case_local_train(LocalTensorfs, "local_train_grad.npz", (20, 24, 28), 41)
    # round 2
    case_config2(TensorVMSplit, "config2_300cube.npz")
    case_train_grad_big(TensorVMSplit, "field_128_train_grad.npz")
    case_sample_ray(TensorVMSplit, "sample_ray.npz")
    case_reg(TensorVMSplit, "reg_losses.npz")
    case_alpha_mask(TensorVMSplit, "alpha_mask_rebuild.npz")
    case_sixd("sixd_to_mtx.npz")
    case_local_train_views(LocalTensorfs, "local_train_3views.npz", (20, 24, 28), 43, [0, 2, 4])
    case_local_train_views(LocalTensorfs, "local_train_prior.npz", (20, 24, 28), 45, [0, 1, 3, 4], camera_prior=True)
    case_config3(LocalTensorfs, "config3_4x300.npz")
    case_upsample(TensorVMSplit, "upsample_grid.npz")
    case_geo_losses("geo_losses.npz")
    # round 3
    case_trajectory(TensorVMSplit, LocalTensorfs, "trajectory_30it.npz")
    # round 4: BASELINE configs[4]'s own sizes (500^3; the reference's default end size 640^3; train.py's upsample ladder)
    case_train_grad_big(TensorVMSplit, "field_500_train_grad.npz", grid=(500, 500, 500), R=512, seed=41)
    case_train_grad_big(TensorVMSplit, "field_640_train_grad.npz", grid=(640, 640, 640), R=256, seed=43)
    case_ladder(TensorVMSplit, "ladder_64_to_640.npz")
    # round 4: non-default colour-network configurations (the generic engine, csrc/lrf_generic.inl)
    case_pe(TensorVMSplit, "field_pe_2_3_64.npz", (20, 24, 28), 64, 96, 101, fea_pe=2, view_pe=3, featureC=64)
    case_pe(TensorVMSplit, "field_pe_0_2_128.npz", (24, 20, 22), 48, 96, 103, fea_pe=0, view_pe=2, featureC=128)
    case_pe(TensorVMSplit, "field_pe_6_6_200.npz", (16, 18, 20), 32, 60, 105, fea_pe=6, view_pe=6, featureC=200)


if __name__ == "__main__":
    main()
