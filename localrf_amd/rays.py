"""Per-ray host math around the native field renderer: pixel ids -> camera rays, 6D
rotations, scene contraction.  These are O(R) torch ops kept differentiable so pose,
intrinsic and exposure gradients flow through autograd exactly as in the reference
(cited lines are relative to /root/reference/localTensoRF)."""
import torch


def contract(x):
    """L-infinity scene contraction (utils/ray_utils.py:9-12).  The render kernels apply
    this per sample in registers; this torch form exists for callers outside the path."""
    m = x.abs().amax(dim=-1, keepdim=True).clamp(min=1e-6)
    return torch.where(m <= 1, x, ((2 * m - 1) / (m * m)) * x)


def ids2pixel(W, H, ids):
    """Ray index -> (column, row) (local_tensorfs.py:23-29)."""
    return ids % W, (ids // W) % H


def ids2pixel_view(W, H, ids):
    """Ray index -> (column, row, view) (local_tensorfs.py:14-21)."""
    return ids % W, (ids // W) % H, ids // (W * H)


def get_ray_directions_lean(i, j, focal, center):
    """Pinhole directions for pixel (i, j) (utils/ray_utils.py:14-24)."""
    x = (i.float() + 0.5 - center[0]) / focal
    y = -(j.float() + 0.5 - center[1]) / focal
    return torch.stack([x, y, -torch.ones_like(x)], -1)


def get_ray_directions_360(i, j, W, H):
    """Equirectangular directions (utils/ray_utils.py:26-37)."""
    phi = (j.float() + 0.5) * torch.pi / H - torch.pi / 2.0
    theta = (i.float() + 0.5) * 2.0 * torch.pi / W + torch.pi
    return torch.stack([torch.cos(phi) * torch.sin(theta), torch.sin(phi),
                        torch.cos(phi) * torch.cos(theta)], -1)


def get_rays_lean(directions, c2w):
    """(origin, direction) in field space from per-ray [B,3,4] cam-to-field transforms
    (utils/ray_utils.py:39-54).  Directions are NOT normalised."""
    rays_d = torch.bmm(c2w[:, :3, :3], directions[..., None])[..., 0]
    return c2w[:, :3, 3], rays_d


def sixD_to_mtx(r, reference_cross=True):
    """Gram-Schmidt 6D -> rotation matrix, columns (b1,b2,b3) (utils/utils.py:381-388).
    The reference's `torch.cross(b1, b2)` has no `dim`, i.e. it runs over the first axis of size
    3 -- the view axis when exactly 3 views are stacked.  reference_cross=True reproduces that."""
    a1, a2 = r[..., 0], r[..., 1]
    b1 = a1 / torch.norm(a1, dim=-1)[:, None]
    b2 = a2 - torch.sum(b1 * a2, dim=-1)[:, None] * b1
    b2 = b2 / torch.norm(b2, dim=-1)[:, None]
    dim = 0 if (reference_cross and b1.dim() == 2 and b1.shape[0] == 3) else -1
    b3 = torch.linalg.cross(b1, b2, dim=dim)
    return torch.stack([b1, b2, b3], dim=-1)


def mtx_to_sixD(r):
    """utils/utils.py:391-392."""
    return torch.stack([r[..., 0], r[..., 1]], dim=-1)


def N_to_reso(n_voxels, bbox):
    """Voxel budget -> per-axis resolution (utils/utils.py:200-203)."""
    lo, hi = bbox
    voxel = ((hi - lo).prod() / n_voxels).pow(1 / 3)
    return ((hi - lo) / voxel).long().tolist()
