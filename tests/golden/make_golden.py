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


def main():
    TensorVMSplit, AlphaGridMask, LocalTensorfs = import_reference()
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


if __name__ == "__main__":
    main()
