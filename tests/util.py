"""Shared helpers for the tests: golden loading and field construction."""
import contextlib
import io
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

FIELD_KW = dict(density_n_comp=[8, 8, 8], appearance_n_comp=[24, 24, 24], app_dim=27,
                shadingMode="MLP_Fea_late_view", near_far=[0.1, 1e3], density_shift=-5,
                alphaMask_thres=1e-4, distance_scale=25, rayMarch_weight_thres=1e-3,
                pos_pe=0, view_pe=0, fea_pe=0, featureC=128, step_ratio=0.5,
                fea2denseAct="softplus")


def load_golden(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    return {k: g[k] for k in g.files}


def golden_field_dict(g, prefix="f."):
    """state-dict-named numpy arrays of a stored field -> oracle `fld` dict."""
    return {k[len(prefix):]: v for k, v in g.items() if k.startswith(prefix)}


def make_field(grid, device="cpu", seed=None, **over):
    from localrf_amd import TensorVMSplit
    if seed is not None:
        torch.manual_seed(seed)
    kw = dict(FIELD_KW)
    kw.update(over)
    aabb = 2 * torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    return TensorVMSplit(torch.device(device), aabb.to(device), list(grid), **kw)


def field_from_golden(g, device, prefix="f."):
    """Build a TensorVMSplit on `device` holding the golden's parameters."""
    from localrf_amd import AlphaGridMask
    fld = golden_field_dict(g, prefix)
    grid = [int(v) for v in g["grid"]]
    f = make_field(grid, device)
    sd = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in fld.items()}
    if "alphaMask.alpha_volume" in sd:
        f.alphaMask = AlphaGridMask(torch.device(device), sd["alphaMask.aabb"].to(device),
                                    sd["alphaMask.alpha_volume"][0, 0].to(device))
    f.load_state_dict(sd)
    return f.to(device)


def make_rays(R, seed, pinhole=False):
    gen = torch.Generator().manual_seed(seed)
    o = 0.05 * torch.randn(R, 3, generator=gen)
    d = torch.randn(R, 3, generator=gen)
    d = d / d.norm(dim=-1, keepdim=True)
    if pinhole:
        d = d / d[:, 2:3].abs().clamp(min=0.2)
    return torch.cat([o, d], -1)


def rel_err(a, b, floor=1e-3):
    """max |a-b| / max(|b|, floor): the 1e-4 relative fp32 bar of BASELINE.json's north_star."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float((np.abs(a - b) / np.maximum(np.abs(b), floor)).max())


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def torch_scene_chain(ray_ids, c2w, shifts, focal, center, per_view, W, H, fov360):
    """The reference's op chain (local_tensorfs.py:397-431,448-452) in plain torch."""
    from localrf_amd.rays import (get_ray_directions_360, get_ray_directions_lean, get_rays_lean,
                                  ids2pixel)
    col, row = ids2pixel(W, H, ray_ids)
    dirs = (get_ray_directions_360(col, row, W, H) if fov360
            else get_ray_directions_lean(col, row, focal, center))
    rays = []
    for k in range(shifts.shape[0]):
        m = c2w.clone()
        m[:, :3, 3] += shifts[k]
        o, d = get_rays_lean(dirs, m.repeat_interleave(per_view, dim=0))
        rays.append(torch.cat([o, d], -1))
    return torch.stack(rays, 0), dirs, torch.stack([col, row], -1)


# ----------------------------------------------------------------- seed-regenerated goldens (round 2)
def state_checksum(sd):
    return float(sum(v.double().abs().sum() for v in sd.values() if v.dtype.is_floating_point))


def field_from_seed(g, device="cpu", **over):
    """Rebuild the field a golden was recorded on from its seed (the package creates parameters in
    the reference's order, so torch.manual_seed(seed) reproduces it); verified by checksum."""
    if "density_shift" in g:
        over.setdefault("density_shift", float(g["density_shift"]))
    f = quiet(make_field, [int(v) for v in g["grid"]], "cpu", seed=int(g["seed"]), **over)
    sc = float(g["scale_density"]) if "scale_density" in g else 1.0
    if sc != 1.0:
        with torch.no_grad():
            for p in f.density_plane:
                p.mul_(sc)
    got = state_checksum(f.state_dict())
    want = float(g["field_sum"][0])
    assert abs(got - want) <= 1e-6 * want, ("seeded field differs from the reference's", got, want)
    return f.to(device) if str(device) != "cpu" else f


def packed_grad_check(g, key, grad, tol, floor=1e-12):
    """Compare a gradient tensor with a golden written by make_golden.pack_grad (full tensor, or a
    seeded subset + the largest entries + max + L2).  Returns max|diff| / max|ref|."""
    a = np.asarray(grad, np.float64).reshape(-1)
    ref = g["grad." + key].astype(np.float64).reshape(-1)
    if ("gidx." + key) in g:
        a = a[g["gidx." + key]]
    gmax = max(float(g["gmax." + key]), floor)
    err = float(np.abs(a - ref).max() / gmax)
    assert err <= tol, (key, err, tol)
    return err


def local_from_golden_seed(g, device="cpu", camera_prior=None, lr_i=1e-3, n_grow=0, scale_density_last=1.0):
    """LocalTensorfs rebuilt from the seed of a golden written by make_golden.build_local (+ the
    recorded poses / exposure / intrinsics loaded on top); fields verified by checksum."""
    from localrf_amd import LocalTensorfs
    torch.manual_seed(int(g["seed"]))
    aabb = 2 * torch.tensor([[-1.0, -1, -1], [1, 1, 1]])
    lt = quiet(LocalTensorfs, fov=85.6, n_init_frames=5, n_overlap=3, WH=(int(g["W"]), int(g["H"])),
               n_iters_per_frame=600, n_iters_reg=100, lr_R_init=5e-3, lr_t_init=5e-4,
               lr_i_init=lr_i, lr_exposure_init=1e-3, rf_lr_init=0.02, rf_lr_basis=1e-3,
               lr_decay_target_ratio=0.1, N_voxel_list={}, update_AlphaMask_list=[],
               camera_prior=camera_prior, device="cpu", lr_upsample_reset=True,
               aabb=aabb, gridSize=[int(v) for v in g["grid"]], **FIELD_KW)
    for _ in range(n_grow):                       # make_golden.case_config3: 3 x (3 frames, 1 field)
        for _ in range(3):
            quiet(lt.append_frame)
        quiet(lt.append_rf, 3)
    small = {k[3:]: torch.from_numpy(np.ascontiguousarray(v)) for k, v in g.items() if k.startswith("lt.")}
    sd = lt.state_dict()
    missing = [k for k in sd if not k.startswith("tensorfs.") and k not in small]
    assert not missing, missing
    with torch.no_grad():
        for k, v in small.items():
            assert sd[k].shape == v.shape, (k, sd[k].shape, v.shape)
            sd[k].copy_(v)
        if scale_density_last != 1.0:
            for p in lt.tensorfs[-1].density_plane:
                p.mul_(scale_density_last)
    got = state_checksum({k: v for k, v in lt.state_dict().items() if k.startswith("tensorfs.")})
    want = float(g["field_sum"][0])
    assert abs(got - want) <= 1e-6 * want, ("seeded fields differ from the reference's", got, want)
    if str(device) != "cpu":
        lt = lt.to(device)
        lt.device = torch.device(device)
        for f in lt.tensorfs:
            f.to(device)
    return lt



# ----------------------------------------------------------------- gradient parity with forced ReLU masks
MAT_MODE = ((0, 1), (0, 2), (1, 2))
VEC_MODE = (2, 1, 0)


class capture_train_ws:
    """Keeps the training workspace of the field's next row-saving forward alive, so that a test can
    read the activation rows after backward (lrf_workspace_layout_bwd)."""

    def __init__(self, f):
        self.f, self.ws = f, None

    def __enter__(self):
        orig = self.f._native_forward_train

        def wrap(*a):
            out = orig(*a)
            self.ws = out[2]
            return out
        self.f._native_forward_train = wrap
        return self

    def __exit__(self, *exc):
        del self.f._native_forward_train


def kernel_relu_masks(f, rays, z, ws):
    """The ReLU masks the kernel's colour network used in the training forward that filled workspace `ws`
    (lrf_render_fwd_train): the mask bits it saved for the data-gradient kernel (relu(h) > 0 per unit; the hidden
    activations themselves are not stored since round 4), per shaded sample.
    Returns (lin, m1, m2, n_shaded): lin = ray * S + sample, sorted; m1, m2 [n,128] bool.  Call after backward (rowinfo
    is written by the data-gradient kernel)."""
    import ctypes as C
    from localrf_amd import _native as N
    R, S = rays.shape[0], z.numel()
    out = (C.c_uint64 * 9)()
    N.lib().lrf_workspace_layout_bwd(R, S, (C.c_int32 * 3)(*f._grid_host), out)
    _, _, ri_off, toff_off, _, _, bits_off, perm_off, _ = [int(v) for v in out]
    toff = ws[toff_off:toff_off + 4 * (R + 1)].view(torch.int32)
    tiles = int(toff[R])
    rows = tiles * 16
    # relu_bits[tile][layer][lane = s + 16 g]: bit 4 t1 + r = unit 16 t1 + 4 g + r of the tile's sample s (k_shade3<SAVE>)
    bits = ws[bits_off:bits_off + tiles * 2 * 64 * 4].view(torch.int32).view(tiles, 2, 4, 16)        # [tile][layer][g][s]
    u = torch.arange(128, device=ws.device)
    t1, gq, r = u >> 4, (u >> 2) & 3, u & 3
    sel = bits[:, :, gq, :]                                                                           # [tile][layer][unit][s]
    m = ((sel >> (4 * t1 + r)[None, None, :, None]) & 1).bool().permute(0, 3, 1, 2).reshape(rows, 2, 128)
    rowinfo = ws[ri_off:ri_off + rows * 4].view(torch.int32)
    valid = rowinfo >= 0
    lin = rowinfo[valid].long()
    if getattr(f, "sort_rays", False) and 2 <= R <= 32768:                # rows are indexed by SLOT of the direction-sorted batch
        perm = ws[perm_off:perm_off + 4 * R].view(torch.int32).long()
        lin = perm[lin // S] * S + lin % S
    m1, m2 = m[valid][:, 0], m[valid][:, 1]
    order = torch.argsort(lin)
    return lin[order], m1[order], m2[order], int(valid.sum())


def port_gradients(f, rays, z, g_rgb, g_depth, white, masks, names):
    """Gradients of (rgb * g_rgb).sum() + (depth * g_depth).sum() through the reference's ATen op chain
    (oracle/vm_render_torch.py, pinned to the reference goldens) on the field's parameters and on the rays, with the
    colour network's ReLU masks FORCED to `masks` (kernel_relu_masks) for the samples the kernel shaded -- both sides then
    differentiate the same piecewise-linear function.  Returns (grads by name, info): info counts the (sample, unit)
    pairs where the forced mask differs from the port's own sign (n_flips) and their largest |pre-activation|."""
    from oracle import vm_render_torch as ot
    fld = {k: v.detach().clone() for k, v in f.state_dict().items()}
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in f.named_parameters()}
    r2 = rays.detach().clone().requires_grad_(True)
    info = {}
    a2, b2 = ot.render_field({**fld, **leaves}, r2, z.reshape(1, -1), white, 0.0, density_shift=float(f.density_shift),
                             weight_thres=f.rayMarch_weight_thres, fea2dense_act=f.fea2denseAct,
                             relu_masks=masks[:3] if masks is not None else None, info=info)
    ((a2 * g_rgb).sum() + (b2 * g_depth).sum()).backward()
    grads = {n: (r2.grad if n == "rays" else leaves[n].grad) for n in names}
    return grads, info


def check_grads(mine, ref, tol=1e-4, subset=None, gmax=None, tol_for=None):
    """Every gradient tensor within `tol` of its largest reference magnitude -- no exceptions.  subset / gmax:
    name -> flat indices / max magnitude when `ref` holds only part of a tensor (make_golden.pack_grad).
    tol_for: name -> tolerance for tensors with a stated, separately justified bar."""
    worst = {}
    for name, gm in mine.items():
        gr = ref[name]
        if gr is None:
            gr = torch.zeros_like(gm)
        if subset is not None and name in subset:
            gm = gm.reshape(-1)[subset[name]]
        gm, gr = gm.reshape(-1), gr.reshape(-1)
        den = max(float(gmax[name]) if gmax is not None else float(gr.abs().max()), 1e-12)
        worst[name] = float((gm - gr).abs().max()) / den
        assert worst[name] <= (tol_for or {}).get(name, tol), (name, worst[name])
    return worst
