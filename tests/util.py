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
