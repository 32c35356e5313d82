"""CPU ORACLE (test infrastructure, NOT product code).

A numpy restatement of localrf's per-ray volume-rendering path, written from the
specification in SURVEY.md Appendix A.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module; localrf_amd/ never does.

Parity status: PINNED.  The reference ships no tests or golden vectors, so the pin is
tests/golden/*.npz, produced by tests/golden/make_golden.py, which imports the real
reference modules from /root/reference in the build container (torch 2.10.0 CPU) and
records their outputs; tests/test_oracle_golden.py checks this file against them.

Every function cites the reference lines it restates (paths relative to
/root/reference/localTensoRF).  All arithmetic runs in `dt` (float32 to mimic the
reference, float64 for a tight ground truth).

Field description (`fld`): a dict with the reference's state-dict tensor names
  density_plane.{0,1,2} [1,Cd,H,W]   density_line.{0,1,2} [1,Cd,L,1]
  app_plane.{0,1,2}     [1,Ca,H,W]   app_line.{0,1,2}     [1,Ca,L,1]
  basis_mat.weight [app_dim, 3*Ca]
  renderModule.mlp.0.{weight,bias} renderModule.mlp.2.{weight,bias}
  renderModule.mlp_view.0.{weight,bias}
  aabb [2,3]
plus optional 'alphaMask.alpha_volume' [1,1,Z,Y,X], 'alphaMask.aabb' [2,3] and scalars
  density_shift, distance_scale, rayMarch_weight_thres, fea2denseAct, fea_pe, view_pe.
"""
import numpy as np

MAT_MODE = ((0, 1), (0, 2), (1, 2))   # models/tensorBase.py:274
VEC_MODE = (2, 1, 0)                  # models/tensorBase.py:275

DEFAULTS = dict(density_shift=-5.0, distance_scale=25.0, rayMarch_weight_thres=1e-3,
                fea2denseAct="softplus", fea_pe=0, view_pe=0)


def _f(fld, key):
    return fld.get(key, DEFAULTS[key])


# ----------------------------------------------------------------------------- rays
def contract(x):
    """utils/ray_utils.py:9-12 -- L-infinity scene contraction."""
    m = np.maximum(np.abs(x).max(axis=-1, keepdims=True), x.dtype.type(1e-6))
    with np.errstate(divide="ignore", invalid="ignore"):
        scaled = ((2 * m - 1) / (m * m)) * x
    return np.where(m <= 1, x, scaled).astype(x.dtype)


def z_schedule(n_samples_arg, dt=np.float32, jitter=None):
    """models/tensorBase.py:419-437 -- ray-independent sample distances.

    n_samples_arg is the N_samples passed to forward (already resolved from nSamples);
    the count actually used is 2*(N//6).  `jitter` = (U, U') two arrays [h] of uniform
    draws (train mode), or None (eval)."""
    h = int(n_samples_arg) // 6
    t = (np.arange(h, dtype=dt) / dt(h)).astype(dt)
    a = t.copy()
    tb = t.copy()
    if jitter is not None:
        a = (a + np.asarray(jitter[0], dt) / dt(h)).astype(dt)
        tb = (tb + np.asarray(jitter[1], dt) / dt(h)).astype(dt)
    near, far = dt(1.0), dt(1e3)
    b = (dt(1.0) / (dt(1.0) / near * (dt(1.0) - tb) + dt(1.0) / far * tb)).astype(dt)
    return (np.concatenate([a, b]) + dt(1e-1)).astype(dt)


def sample_ray_contracted(rays_o, dirs_unit, z):
    """models/tensorBase.py:438-440 -- positions then contraction."""
    x = rays_o[:, None, :] + dirs_unit[:, None, :] * z[None, :, None]
    return contract(x.astype(rays_o.dtype))


def sample_ray_aabb(rays_o, rays_d, aabb, step_size, n_samples, near_far, jitter=None):
    """models/tensorBase.py:396-417 -- TensoRF AABB march (dead code in the reference's
    forward, kept because BASELINE.json's north_star names it)."""
    dt = rays_o.dtype.type
    near, far = dt(near_far[0]), dt(near_far[1])
    vec = np.where(rays_d == 0, dt(1e-6), rays_d)
    rate_a = (aabb[1] - rays_o) / vec
    rate_b = (aabb[0] - rays_o) / vec
    t_min = np.clip(np.minimum(rate_a, rate_b).max(-1), near, far)
    rng = np.arange(n_samples, dtype=rays_o.dtype)[None]
    if jitter is not None:
        rng = rng + np.asarray(jitter, rays_o.dtype)[:, None]
    interpx = t_min[:, None] + dt(step_size) * rng
    pts = rays_o[:, None, :] + rays_d[:, None, :] * interpx[..., None]
    outside = ((aabb[0] > pts) | (pts > aabb[1])).any(-1)
    return pts.astype(rays_o.dtype), interpx.astype(rays_o.dtype), ~outside


# ------------------------------------------------------------------- interpolation
def _unnorm(u, size, dt):
    """ATen GridSampler.h grid_sampler_unnormalize(align_corners=True) + border clip."""
    ix = ((u + dt(1)) / dt(2)) * dt(size - 1)
    return np.clip(ix, dt(0), dt(size - 1))


def bilerp_plane(plane, ua, ub):
    """F.grid_sample(plane[1,C,H,W], (ua,ub), align_corners=True, padding='border')
    as called at models/tensoRF.py:135-140,177-183.  ua indexes W, ub indexes H.
    plane: [C,H,W]; returns [C,P]."""
    dt = plane.dtype.type
    C, H, W = plane.shape
    ix = _unnorm(ua.astype(plane.dtype), W, dt)
    iy = _unnorm(ub.astype(plane.dtype), H, dt)
    x0 = np.floor(ix)
    y0 = np.floor(iy)
    tx = ix - x0
    ty = iy - y0
    x0 = x0.astype(np.int64)
    y0 = y0.astype(np.int64)
    x1 = x0 + 1
    y1 = y0 + 1
    vx1 = x1 <= W - 1          # out-of-range neighbour contributes zero
    vy1 = y1 <= H - 1
    x1c = np.minimum(x1, W - 1)
    y1c = np.minimum(y1, H - 1)
    w00 = (dt(1) - tx) * (dt(1) - ty)
    w10 = tx * (dt(1) - ty) * vx1
    w01 = (dt(1) - tx) * ty * vy1
    w11 = tx * ty * (vx1 & vy1)
    return (plane[:, y0, x0] * w00 + plane[:, y0, x1c] * w10
            + plane[:, y1c, x0] * w01 + plane[:, y1c, x1c] * w11).astype(plane.dtype)


def lerp_line(line, u):
    """F.grid_sample(line[1,C,L,1], (0,u)) as called at models/tensoRF.py:141-146:
    1-D linear interpolation along L.  line: [C,L]; returns [C,P]."""
    dt = line.dtype.type
    C, L = line.shape
    iy = _unnorm(u.astype(line.dtype), L, dt)
    y0 = np.floor(iy)
    ty = iy - y0
    y0 = y0.astype(np.int64)
    y1 = y0 + 1
    vy1 = y1 <= L - 1
    y1c = np.minimum(y1, L - 1)
    return (line[:, y0] * (dt(1) - ty) + line[:, y1c] * (ty * vy1)).astype(line.dtype)


def trilerp_zeros(vol, p):
    """F.grid_sample(vol[1,1,Z,Y,X], p, align_corners=True) with the default zeros
    padding (models/tensorBase.py:51-55).  vol: [Z,Y,X]; p: [P,3] in [-1,1]; -> [P]."""
    dt = vol.dtype.type
    Z, Y, X = vol.shape
    out = np.zeros(p.shape[0], vol.dtype)
    ix = ((p[:, 0] + dt(1)) / dt(2)) * dt(X - 1)
    iy = ((p[:, 1] + dt(1)) / dt(2)) * dt(Y - 1)
    iz = ((p[:, 2] + dt(1)) / dt(2)) * dt(Z - 1)
    x0, y0, z0 = np.floor(ix), np.floor(iy), np.floor(iz)
    tx, ty, tz = ix - x0, iy - y0, iz - z0
    x0, y0, z0 = x0.astype(np.int64), y0.astype(np.int64), z0.astype(np.int64)
    for dz in (0, 1):
        for dy in (0, 1):
            for dx in (0, 1):
                xi, yi, zi = x0 + dx, y0 + dy, z0 + dz
                ok = (xi >= 0) & (xi < X) & (yi >= 0) & (yi < Y) & (zi >= 0) & (zi < Z)
                w = ((tx if dx else dt(1) - tx) * (ty if dy else dt(1) - ty)
                     * (tz if dz else dt(1) - tz))
                v = vol[np.clip(zi, 0, Z - 1), np.clip(yi, 0, Y - 1), np.clip(xi, 0, X - 1)]
                out += np.where(ok, v * w, dt(0)).astype(vol.dtype)
    return out


# ------------------------------------------------------------------------ VM grids
def normalize_coord(x, aabb):
    """models/tensorBase.py:342-345."""
    dt = x.dtype.type
    inv = (dt(2.0) / (aabb[1] - aabb[0])).astype(x.dtype)
    return ((x - aabb[0]) * inv - dt(1)).astype(x.dtype)


def density_feature(fld, u):
    """models/tensoRF.py:112-151 -- sum_p sum_c plane_p,c * line_p,c.  u: [P,3]."""
    acc = np.zeros(u.shape[0], u.dtype)
    for p in range(3):
        m0, m1 = MAT_MODE[p]
        pl = bilerp_plane(fld[f"density_plane.{p}"][0], u[:, m0], u[:, m1])
        ln = lerp_line(fld[f"density_line.{p}"][0, :, :, 0], u[:, VEC_MODE[p]])
        acc = acc + (pl * ln).sum(0, dtype=u.dtype)
    return acc


def app_feature(fld, u):
    """models/tensoRF.py:153-196 -- 3*Ca plane*line products (p-major) then basis_mat."""
    prods = []
    for p in range(3):
        m0, m1 = MAT_MODE[p]
        pl = bilerp_plane(fld[f"app_plane.{p}"][0], u[:, m0], u[:, m1])
        ln = lerp_line(fld[f"app_line.{p}"][0, :, :, 0], u[:, VEC_MODE[p]])
        prods.append(pl * ln)
    x = np.concatenate(prods, 0).T                      # [P, 3*Ca]
    return (x @ fld["basis_mat.weight"].T).astype(u.dtype), x


def feature2density(fld, f):
    """models/tensorBase.py:495-499; softplus has torch's threshold 20."""
    dt = f.dtype.type
    if _f(fld, "fea2denseAct") == "relu":
        return np.maximum(f, dt(0))
    y = f + dt(_f(fld, "density_shift"))
    with np.errstate(over="ignore"):
        sp = np.log1p(np.exp(np.minimum(y, dt(20)))).astype(f.dtype)
    return np.where(y > 20, y, sp).astype(f.dtype)


def positional_encoding(v, freqs):
    """models/tensorBase.py:14-21."""
    bands = (2.0 ** np.arange(freqs)).astype(v.dtype)
    pts = (v[..., None] * bands).reshape(v.shape[:-1] + (freqs * v.shape[-1],))
    return np.concatenate([np.sin(pts), np.cos(pts)], -1).astype(v.dtype)


def mlp_late_view(fld, feat, viewdirs, refine=True):
    """models/tensorBase.py:115-135 -- MLPRender_Fea_late_view.forward."""
    dt = feat.dtype
    fea_pe, view_pe = int(_f(fld, "fea_pe")), int(_f(fld, "view_pe"))
    x = feat
    if fea_pe > 0:
        if refine:
            x = np.concatenate([feat, positional_encoding(feat, fea_pe)], -1)
        else:
            x = np.concatenate([feat, np.zeros((feat.shape[0], 2 * fea_pe * feat.shape[1]), dt)], -1)
    v = viewdirs
    if view_pe > 0:
        v = np.concatenate([viewdirs, positional_encoding(viewdirs, view_pe)], -1)
    h1 = np.maximum(x @ fld["renderModule.mlp.0.weight"].T + fld["renderModule.mlp.0.bias"], 0)
    h2 = np.maximum(h1 @ fld["renderModule.mlp.2.weight"].T + fld["renderModule.mlp.2.bias"], 0)
    o = (np.concatenate([h2, v], -1) @ fld["renderModule.mlp_view.0.weight"].T
         + fld["renderModule.mlp_view.0.bias"])
    return (1.0 / (1.0 + np.exp(-o))).astype(dt)


# --------------------------------------------------------------------- compositing
def alpha2weights(alpha):
    """models/tensorBase.py:23-32 -- note: forces the last alpha to 1 IN PLACE."""
    dt = alpha.dtype.type
    alpha[:, -1] = 1
    T = np.cumprod(np.concatenate([np.ones((alpha.shape[0], 1), alpha.dtype),
                                   dt(1.0) - alpha + dt(1e-10)], -1), -1, dtype=alpha.dtype)
    return (alpha * T[:, :-1]).astype(alpha.dtype), T


def render_field(fld, rays, z, white_bg=True, floater_thresh=0.0, refine=True,
                 return_extras=False):
    """models/tensorBase.py:567-636 -- TensorBase.forward with z (step 2) supplied.

    rays [R,6] = (o, d); z [S].  Returns (rgb_map [R,3], depth_map [R])."""
    dt = rays.dtype.type
    fld = {k: (np.asarray(v, rays.dtype) if isinstance(v, np.ndarray) and v.dtype.kind == "f" else v)
           for k, v in fld.items()}
    z = np.asarray(z, rays.dtype)
    aabb = fld["aabb"]
    R, S = rays.shape[0], z.shape[0]
    d = rays[:, 3:6]
    n = np.sqrt((d * d).sum(-1, keepdims=True, dtype=rays.dtype))        # :578-580
    dh = (d / n).astype(rays.dtype)
    x = sample_ray_contracted(rays[:, :3], dh, z)                         # :581-583
    dists = np.concatenate([z[1:] - z[:-1], np.zeros(1, rays.dtype)])[None]   # :584-587
    valid = np.ones((R, S), bool)
    if fld.get("alphaMask.alpha_volume") is not None:                     # :593-598
        maabb = fld.get("alphaMask.aabb", aabb)
        pm = normalize_coord(x.reshape(-1, 3), maabb)
        a = trilerp_zeros(fld["alphaMask.alpha_volume"][0, 0], pm).reshape(R, S)
        valid &= a > 0
    valid[:, -1] = False                                                  # :600
    u = normalize_coord(x, aabb)                                          # :602
    sigma = np.zeros((R, S), rays.dtype)
    if valid.any():
        f = density_feature(fld, u[valid])                                # :603-606
        sigma[valid] = feature2density(fld, f)                            # :607-608
    alpha = (dt(1.0) - np.exp(-sigma * dists * dt(_f(fld, "distance_scale")))).astype(rays.dtype)  # :610
    weight, T = alpha2weights(alpha)                                      # :612
    acc = weight.sum(-1, dtype=rays.dtype)                                # :614
    depth = ((weight * z[None]).sum(-1, dtype=rays.dtype) / n[:, 0]).astype(rays.dtype)  # :615
    if floater_thresh > 0:                                                # :617-620
        k = np.arange(S, dtype=rays.dtype)[None]
        idx_map = (weight * k).sum(-1, keepdims=True, dtype=rays.dtype)
        alpha[k < idx_map * dt(floater_thresh)] = 0
        weight, T = alpha2weights(alpha)
    shade = weight > dt(_f(fld, "rayMarch_weight_thres"))                 # :622
    rgb = np.zeros((R, S, 3), rays.dtype)
    feat = x72 = None
    if shade.any():
        feat, x72 = app_feature(fld, u[shade])                            # :624-626
        vd = np.broadcast_to(dh[:, None, :], (R, S, 3))[shade]
        rgb[shade] = mlp_late_view(fld, feat, vd, refine)                 # :627-630
    rgb_map = (weight[..., None] * rgb).sum(-2, dtype=rays.dtype)         # :632
    if white_bg:
        rgb_map = rgb_map + (dt(1.0) - acc[:, None])                      # :633-634
    if return_extras:
        return rgb_map, depth, dict(x=x, u=u, sigma=sigma, alpha=alpha, weight=weight,
                                    acc=acc, shade=shade, rgb=rgb, feat=feat, x72=x72)
    return rgb_map.astype(rays.dtype), depth


# ---------------------------------------------------------------- LocalTensorfs side
def sixd_to_mtx(r):
    """utils/utils.py:381-388 -- Gram-Schmidt 6D -> 3x3 (columns b1,b2,b3).  The reference's
    torch.cross has no `dim` and so runs over the first axis of size 3: the VIEW axis when exactly
    three views are stacked (reproduced here; pinned by tests/golden/sixd_to_mtx.npz)."""
    b1 = r[..., 0]
    b1 = b1 / np.linalg.norm(b1, axis=-1)[:, None]
    b2 = r[..., 1] - (b1 * r[..., 1]).sum(-1)[:, None] * b1
    b2 = b2 / np.linalg.norm(b2, axis=-1)[:, None]
    b3 = np.cross(b1, b2, axis=0) if r.shape[0] == 3 else np.cross(b1, b2)
    return np.stack([b1, b2, b3], -1).astype(r.dtype)


def pixel_rays(ray_ids, W, H, focal, center):
    """local_tensorfs.py:23-29 + utils/ray_utils.py:14-24."""
    col = ray_ids % W
    row = (ray_ids // W) % H
    i = col.astype(np.float32) + np.float32(0.5)
    j = row.astype(np.float32) + np.float32(0.5)
    dirs = np.stack([(i - center[0]) / focal, -(j - center[1]) / focal, -np.ones_like(i)], -1)
    return dirs.astype(np.float32), np.stack([col, row], -1)


def render_local(fields, world2rf, ray_ids, view_ids, W, H, r_c2w, t_c2w, focal, center,
                 blending_weights, exposure=None, z_per_field=None, floater_thresh=0.0,
                 refine=False, dt=np.float32):
    """local_tensorfs.py:382-499 -- eval-mode blend of the active fields (one chunk).

    fields: list of fld dicts; world2rf: [F,3]; r_c2w [V,3,2]; t_c2w [V,3];
    blending_weights [V,F]; exposure [V,3,3] or None; z_per_field: list of z arrays."""
    dirs, ij = pixel_rays(ray_ids, W, H, np.float32(focal), np.asarray(center, np.float32))
    rot = sixd_to_mtx(np.asarray(r_c2w, np.float32)[view_ids])          # get_cam2world(view_ids), :292-299
    per = ray_ids.shape[0] // view_ids.shape[0]
    rot_r = np.repeat(rot, per, 0)
    t_r = np.repeat(np.asarray(t_c2w, np.float32)[view_ids], per, 0)
    bw = np.repeat(np.asarray(blending_weights, np.float32), per, 0)
    active = np.nonzero(np.asarray(blending_weights).sum(0))[0].tolist()
    rgbs = np.zeros((ray_ids.shape[0], 3), dt)
    depths = np.zeros(ray_ids.shape[0], dt)
    for f in active:
        o = (t_r + np.asarray(world2rf[f], np.float32)).astype(dt)
        d = np.einsum("rij,rj->ri", rot_r, dirs).astype(dt)               # ray_utils.py:39-54
        rgb, dep = render_field(fields[f], np.concatenate([o, d], -1), z_per_field[f],
                                True, floater_thresh, refine)
        rgbs += rgb * bw[:, f:f + 1].astype(dt)
        depths += dep * bw[:, f].astype(dt)
    if exposure is not None:
        e = np.repeat(np.asarray(exposure, dt)[view_ids], per, 0)
        rgbs = np.einsum("rij,rj->ri", e, rgbs)
    return np.clip(rgbs, 0, 1).astype(dt), depths, dirs, ij
