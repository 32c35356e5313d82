"""CPU ORACLE, torch flavour (test infrastructure, NOT product code).

A functional restatement of the reference's PyTorch path using the same ATen operators
the reference calls (F.grid_sample, cumprod, softplus, boolean-mask compaction), so that
timing it reproduces what the reference costs on this machine's host cores (bench.py's
`cpu_baseline`, kind "port") and, moved to the GPU, what the stock PyTorch-ROCm path
costs (the denominator of BASELINE.json's >=10x target).  localrf_amd/ never imports it.

Parity status: PINNED against tests/golden/*.npz (recorded from the real reference) by
tests/test_oracle_golden.py::test_torch_port_matches_reference.

`fld` is a dict of torch tensors with the reference's state-dict names (see
oracle/vm_render_np.py).  Cited lines: /root/reference/localTensoRF.
"""
import torch
import torch.nn.functional as F

MAT_MODE = ((0, 1), (0, 2), (1, 2))
VEC_MODE = (2, 1, 0)


def z_schedule(n_samples_arg, device="cpu", jitter=None):
    """models/tensorBase.py:419-437."""
    h = int(n_samples_arg) // 6
    t = torch.linspace(0.0, h - 1, h, device=device)[None] / h
    a = t.clone()
    if jitter is not None:
        a = a + jitter[0].to(device)[None] / h
        t = t + jitter[1].to(device)[None] / h
    b = 1.0 / ((1.0 - t) + t / 1e3)
    return torch.cat([a, b], 1) + 0.1


def _vm_lookup(planes, lines, u):
    """grid_sample of 3 planes and 3 lines at normalised points u [P,3]
    (models/tensoRF.py:115-146 / 156-191).  Returns lists of [C,P] tensors."""
    P = u.shape[0]
    outs_p, outs_l = [], []
    for p in range(3):
        gp = u[:, list(MAT_MODE[p])].view(1, P, 1, 2)
        gl = torch.stack([torch.zeros_like(u[:, 0]), u[:, VEC_MODE[p]]], -1).view(1, P, 1, 2)
        outs_p.append(F.grid_sample(planes[p], gp, align_corners=True, padding_mode="border").view(-1, P))
        outs_l.append(F.grid_sample(lines[p], gl, align_corners=True, padding_mode="border").view(-1, P))
    return outs_p, outs_l


def density_feature(fld, u):
    """models/tensoRF.py:112-151."""
    pp, ll = _vm_lookup([fld[f"density_plane.{i}"] for i in range(3)],
                        [fld[f"density_line.{i}"] for i in range(3)], u)
    out = torch.zeros(u.shape[0], device=u.device)
    for a, b in zip(pp, ll):
        out = out + (a * b).sum(0)
    return out


def app_feature(fld, u):
    """models/tensoRF.py:153-196."""
    pp, ll = _vm_lookup([fld[f"app_plane.{i}"] for i in range(3)],
                        [fld[f"app_line.{i}"] for i in range(3)], u)
    x = (torch.cat(pp) * torch.cat(ll)).T
    return F.linear(x, fld["basis_mat.weight"])


def late_view_mlp(fld, feat, viewdirs, masks=None, info=None):
    """models/tensorBase.py:115-135 with fea_pe = view_pe = 0.

    masks = (has [A] bool, m1 [A,128] bool, m2 [A,128] bool): for the samples with `has`, the two ReLUs are replaced by
    multiplication with the GIVEN 0/1 masks (the masks another implementation of the same network used on these samples).
    ReLU'(0) is a jump: a unit whose pre-activation differs from 0 by less than the rounding difference of two
    implementations gets mask 1 in one and 0 in the other, and the gradients differ by that unit's whole contribution.
    With the masks forced both sides differentiate the SAME piecewise-linear function, so gradients can be compared at
    rounding level; the forward value moves by |pre-activation| of the flipped units (~1e-8).  info (dict, optional)
    receives the number of units whose given mask differs from this chain's own sign and the largest |pre-activation|
    among them."""
    p1 = F.linear(feat, fld["renderModule.mlp.0.weight"], fld["renderModule.mlp.0.bias"])
    if info is not None and info.get("want_masks"):
        info["own_m1"] = p1.detach() > 0
    if masks is None:
        h = F.relu(p1)
    else:
        has, m1, m2 = masks
        own = p1.detach() > 0
        use = torch.where(has[:, None], m1, own)
        h = p1 * use.to(p1.dtype)
    p2 = F.linear(h, fld["renderModule.mlp.2.weight"], fld["renderModule.mlp.2.bias"])
    if info is not None and info.get("want_masks"):
        info["own_m2"] = p2.detach() > 0
    if masks is None:
        h = F.relu(p2)
    else:
        own2 = p2.detach() > 0
        use2 = torch.where(has[:, None], m2, own2)
        h = p2 * use2.to(p2.dtype)
        if info is not None:
            f1, f2 = (use != own), (use2 != own2)
            info["n_flips"] = info.get("n_flips", 0) + int(f1.sum() + f2.sum())
            mp = torch.cat([p1.detach()[f1].abs(), p2.detach()[f2].abs(), torch.zeros(1, device=p1.device)]).max()
            info["max_pre"] = max(info.get("max_pre", 0.0), float(mp))
            info["n_forced"] = info.get("n_forced", 0) + int(has.sum())
    o = F.linear(torch.cat([h, viewdirs], -1), fld["renderModule.mlp_view.0.weight"],
                 fld["renderModule.mlp_view.0.bias"])
    return torch.sigmoid(o)


def alpha2weights(alpha):
    """models/tensorBase.py:23-32."""
    alpha[:, -1] = 1
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1.0 - alpha + 1e-10], -1), -1)
    return alpha * T[:, :-1]


def render_field(fld, rays, z, white_bg=True, floater_thresh=0.0, density_shift=-5.0,
                 distance_scale=25.0, weight_thres=1e-3, fea2dense_act="softplus", relu_masks=None, info=None):
    """models/tensorBase.py:567-636 (softplus or relu density :495-499, late-view shading).  z: [1,S].
    relu_masks = (lin [A'] int64 sorted = ray * S + sample of the samples another implementation shaded, m1, m2 [A',128]
    bool): forced ReLU masks of the colour network for those samples (late_view_mlp)."""
    o, d = rays[:, :3], rays[:, 3:6]
    n = torch.norm(d, dim=-1, keepdim=True)
    dh = d / n
    x = o[:, None, :] + dh[:, None, :] * z[..., None]
    m = x.abs().amax(dim=-1, keepdim=True).clamp(min=1e-6)              # utils/ray_utils.py:9-12
    x = torch.where(m <= 1, x, ((2 * m - 1) / (m ** 2)) * x)
    dists = torch.cat([z[:, 1:] - z[:, :-1], torch.zeros_like(z[:, :1])], -1)
    valid = torch.ones(x.shape[:2], dtype=torch.bool, device=x.device)
    aabb = fld["aabb"]
    if fld.get("alphaMask.alpha_volume") is not None:
        maabb = fld.get("alphaMask.aabb", aabb)
        pm = (x.reshape(-1, 3) - maabb[0]) * (1.0 / (maabb[1] - maabb[0]) * 2) - 1
        a = F.grid_sample(fld["alphaMask.alpha_volume"], pm.view(1, -1, 1, 1, 3), align_corners=True).view(-1)
        valid &= (a > 0).view(valid.shape)
    valid[:, -1] = False
    u = (x - aabb[0]) * (2.0 / (aabb[1] - aabb[0])) - 1
    sigma = torch.zeros(x.shape[:2], device=x.device)
    if valid.any():
        df = density_feature(fld, u[valid])
        sigma[valid] = F.softplus(df + density_shift) if fea2dense_act == "softplus" else F.relu(df)     # tensorBase.py:495-499
    alpha = 1.0 - torch.exp(-sigma * dists * distance_scale)
    w = alpha2weights(alpha)
    acc = w.sum(-1)
    depth = (w * z).sum(-1) / n[:, 0]
    if floater_thresh > 0:
        k = torch.arange(alpha.shape[1], device=x.device)[None]
        idx = (w * k).sum(-1, keepdim=True)
        alpha[k < idx * floater_thresh] = 0
        w = alpha2weights(alpha)
    shade = w > weight_thres
    rgb = torch.zeros(x.shape[:2] + (3,), device=x.device)
    if shade.any():
        vd = dh[:, None, :].expand(x.shape)[shade].clone().detach()     # tensorBase.py:628
        masks = None
        if info is not None and info.get("want_masks"):
            info["own_lin"] = torch.nonzero(shade.reshape(-1)).flatten()
        if relu_masks is not None and relu_masks[0].numel() > 0:
            lin, m1, m2 = relu_masks
            mine = torch.nonzero(shade.reshape(-1)).flatten()            # row-major = the order of u[shade]
            pos = torch.searchsorted(lin.contiguous(), mine).clamp(max=max(lin.numel() - 1, 0))
            has = lin[pos] == mine
            masks = (has, m1[pos], m2[pos])
        rgb[shade] = late_view_mlp(fld, app_feature(fld, u[shade]), vd, masks, info)
    rgb_map = (w[..., None] * rgb).sum(-2)
    if white_bg:
        rgb_map = rgb_map + (1.0 - acc[:, None])
    return rgb_map, depth


# ------------------------------------------------------------- rows either side of the path
def sample_ray_aabb(rays_o, rays_d, aabb, step_size, n_samples, near_far, jitter=None):
    """models/tensorBase.py:396-417 (TensoRF AABB march; jitter [R,1] replaces rand_like)."""
    vec = torch.where(rays_d == 0, torch.full_like(rays_d, 1e-6), rays_d)
    rate_a = (aabb[1] - rays_o) / vec
    rate_b = (aabb[0] - rays_o) / vec
    t_min = torch.minimum(rate_a, rate_b).amax(-1).clamp(min=near_far[0], max=near_far[1])
    rng = torch.arange(n_samples)[None].float()
    if jitter is not None:
        rng = rng.repeat(rays_d.shape[-2], 1) + jitter
    interpx = t_min[..., None] + step_size * rng.to(rays_o.device)
    pts = rays_o[..., None, :] + rays_d[..., None, :] * interpx[..., None]
    outside = ((aabb[0] > pts) | (pts > aabb[1])).any(dim=-1)
    return pts, interpx, ~outside


def sixd_to_mtx(r):
    """utils/utils.py:381-388.  torch.cross there has no `dim`: it runs over the first axis of size
    3, i.e. over the VIEWS when exactly three are stacked (a reference quirk, reproduced)."""
    b1 = r[..., 0]
    b1 = b1 / torch.norm(b1, dim=-1)[:, None]
    b2 = r[..., 1] - torch.sum(b1 * r[..., 1], dim=-1)[:, None] * b1
    b2 = b2 / torch.norm(b2, dim=-1)[:, None]
    b3 = torch.linalg.cross(b1, b2, dim=0 if (b1.dim() == 2 and b1.shape[0] == 3) else -1)
    return torch.stack([b1, b2, b3], dim=-1)


def density_l1(fld, grid, density_shift=-5.0):
    """models/tensoRF.py:83-92: every plane x line product over the dense lattice, each plane in
    its own flattening order, summed element by element; sqrt(clamp(softplus)).mean()."""
    n = int(grid[0]) * int(grid[1]) * int(grid[2])
    feat = torch.zeros(n)
    for p in range(3):
        pl = fld[f"density_plane.{p}"]
        ln = fld[f"density_line.{p}"]
        a = pl.reshape(pl.shape[1], -1)
        b = ln.reshape(ln.shape[1], -1)
        feat = feat + torch.bmm(a[..., None], b[:, None]).reshape(a.shape[0], n).sum(0)
    return torch.sqrt(F.softplus(feat + density_shift).clamp(1e-5)).mean()


def tv_loss(fld, kind, weight=1.0):
    """models/tensoRF.py:94-110 with utils/utils.py:293-309 (TVLoss) as `reg`."""
    def reg(x):
        tv = 0
        if x.shape[2] > 1:
            tv = tv + torch.pow(x[:, :, 1:, :] - x[:, :, :-1, :], 2).mean()
        if x.shape[3] > 1:
            tv = tv + torch.pow(x[:, :, :, 1:] - x[:, :, :, :-1], 2).mean()
        return weight * 2 * tv
    total = 0
    for p in range(3):
        total = total + reg(fld[f"{kind}_plane.{p}"].transpose(0, 1)) * 1e-2 \
                      + reg(fld[f"{kind}_line.{p}"].transpose(0, 1)) * 1e-3
    return total


def update_alpha_mask(fld, grid_size, step_size, alpha_mask_thres=1e-4, density_shift=-5.0):
    """models/tensorBase.py:501-558: dense alpha on the lattice (through the current mask, if any),
    3x3x3 max-pool, threshold.  Returns the binary volume [gz,gy,gx] (the new mask's alpha_volume)."""
    aabb = fld["aabb"]
    lin = [torch.linspace(0, 1, int(g)) for g in grid_size]
    dense = torch.stack(torch.meshgrid(*lin, indexing="ij"), -1)
    dense = aabb[0] * (1 - dense) + aabb[1] * dense
    alpha = torch.zeros_like(dense[..., 0])
    for i in range(int(grid_size[0])):
        x = dense[i].reshape(-1, 3)
        keep = torch.ones(x.shape[0], dtype=torch.bool)
        if fld.get("alphaMask.alpha_volume") is not None:
            maabb = fld["alphaMask.aabb"]
            pm = (x - maabb[0]) * (1.0 / (maabb[1] - maabb[0]) * 2) - 1
            keep = F.grid_sample(fld["alphaMask.alpha_volume"], pm.view(1, -1, 1, 1, 3), align_corners=True).view(-1) > 0
        sigma = torch.zeros(x.shape[0])
        if keep.any():
            u = (x[keep] - aabb[0]) * (2.0 / (aabb[1] - aabb[0])) - 1
            sigma[keep] = F.softplus(density_feature(fld, u) + density_shift)
        alpha[i] = (1 - torch.exp(-sigma * step_size)).view(int(grid_size[1]), int(grid_size[2]))
    alpha = alpha.clamp(0, 1).transpose(0, 2).contiguous()[None, None]
    alpha = F.max_pool3d(alpha, kernel_size=3, padding=1, stride=1).view(tuple(int(g) for g in grid_size)[::-1])
    return (alpha >= alpha_mask_thres).float()


def upsample_vm(fld, res_target, mat_mode=((0, 1), (0, 2), (1, 2)), vec_mode=(2, 1, 0)):
    """models/tensoRF.py:198-221 (up_sampling_VM for the density and the appearance tensors): bilinear,
    align_corners=True.  Returns a dict of the 12 resized tensors."""
    out = {}
    for kind in ("density", "app"):
        for i in range(3):
            m0, m1 = mat_mode[i]
            out[f"{kind}_plane.{i}"] = F.interpolate(fld[f"{kind}_plane.{i}"], size=(int(res_target[m1]), int(res_target[m0])),
                                                     mode="bilinear", align_corners=True)
            out[f"{kind}_line.{i}"] = F.interpolate(fld[f"{kind}_line.{i}"], size=(int(res_target[vec_mode[i]]), 1),
                                                    mode="bilinear", align_corners=True)
    return out


def _cam2cams(cam2worlds, indices, offset):
    """utils/utils.py:22-35 (inverse_pose + get_cam2cams)."""
    idx = torch.clamp(indices + offset, 0, len(cam2worlds) - 1)
    a = cam2worlds[idx]
    w = a[:, :3, :3].transpose(1, 2)
    tw = -torch.bmm(w, a[:, :3, 3:])[..., 0]
    b = cam2worlds[indices]
    rot = torch.bmm(w, b[:, :3, :3])
    t = torch.bmm(w, b[:, :3, 3:])[..., 0] + tw
    return rot, t


def _pred_flow(pts, ij, rot, t, focal, center):
    """utils/utils.py:15-21,43-48 (pts2px + get_pred_flow)."""
    q = torch.bmm(rot, pts.transpose(1, 2)).transpose(1, 2) + t[:, None]
    x, y, zc = q[..., 0], -q[..., 1], torch.clip(-q[..., 2], min=1e-6)
    px = torch.stack([x / zc * focal + center[0] - 0.5, y / zc * focal + center[1] - 0.5], -1)
    return px - ij.float()


def flow_loss(depth_map, directions, ij, cam2world, view_ids, starting_frame_id, fwd_flow, fwd_mask, bwd_flow, bwd_mask,
              focal, center, quantile=0.9):
    """train.py:385-410: reprojection flow loss per view, entries above the view's 0.9-quantile zeroed.
    Shapes [V,n,...]; returns (flow_loss_arr.mean(), flow_loss_arr)."""
    fm = fwd_mask.clone()
    fm[view_ids == len(cam2world) - 1] = 0
    idx = view_ids - starting_frame_id
    pts = directions * depth_map[..., None]
    arr = torch.sum(torch.abs(_pred_flow(pts, ij, *_cam2cams(cam2world, idx, -1), focal, center) - bwd_flow), -1) * bwd_mask
    arr = arr + torch.sum(torch.abs(_pred_flow(pts, ij, *_cam2cams(cam2world, idx, 1), focal, center) - fwd_flow), -1) * fm
    keep = ~(arr > torch.quantile(arr.detach(), quantile, dim=1)[..., None])
    arr = arr * keep
    return arr.mean(), arr


def depth_loss(depth_map, invdepths, quantile=0.8):
    """train.py:414-421 with compute_depth_loss (utils/utils.py:50-59): returns (depth_loss_arr.mean(), arr)."""
    def norm(x):
        t = torch.median(x, dim=-1, keepdim=True).values
        s = torch.mean(torch.abs(x - t), dim=-1, keepdim=True)
        return (x - t) / s
    arr = (norm(1 / depth_map.clamp(1e-6)) - norm(invdepths)) ** 2
    keep = ~(arr > torch.quantile(arr.detach(), quantile, dim=1)[..., None])
    arr = arr * keep
    return arr.mean(), arr
