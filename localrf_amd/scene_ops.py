"""Autograd seams for the scene-level kernels (lrf_scene_* in include/lrf.h).

`scene_rays` replaces the torch chain ids2pixel -> get_ray_directions_lean/_360 ->
cam2rf -> repeat_interleave -> get_rays_lean -> cat (local_tensorfs.py:397-431,448-452;
utils/ray_utils.py:14-54) and `scene_blend` replaces the weighted sum over fields, the
per-view exposure bmm and the clamp (local_tensorfs.py:468-474,481-499), each with one
HIP launch; their backward functions return what autograd derives for those chains
(pose, intrinsic, world2rf and exposure gradients).  No CPU fallback.
Cited lines are relative to /root/reference/localTensoRF."""
import ctypes as C

import torch

from . import _native as N


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def _f32c(t):
    """Contiguous fp32 view of t for a native call (only data_ptr() is taken, so an already conforming tensor is
    passed through as is: detach() alone costs ~4 us, 70 of them per training iteration)."""
    if t.dtype is torch.float32 and t.is_contiguous():
        return t
    return t.detach().contiguous().float()


class _SceneRaysFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ray_ids, cam2world, world2rf, focal, center, per_view, W, H, fov360, squeeze=False):
        dev = ray_ids.device
        R, n_rf = ray_ids.shape[0], world2rf.shape[0]
        ids = ray_ids.detach().contiguous().long()
        c2w, w2rf = _f32c(cam2world), _f32c(world2rf)
        fo = None if focal is None else _f32c(focal)
        ce = None if center is None else _f32c(center)
        rays = torch.empty(n_rf, R, 6, dtype=torch.float32, device=dev)
        dirs = torch.empty(R, 3, dtype=torch.float32, device=dev)
        ij = torch.empty(R, 2, dtype=torch.int64, device=dev)
        N.check(N.lib().lrf_scene_rays(ids.data_ptr(), R, per_view, N.ptr(c2w), N.ptr(w2rf), n_rf, N.ptr(fo),
                                       N.ptr(ce), W, H, int(fov360), N.ptr(rays), N.ptr(dirs), ij.data_ptr(),
                                       _stream(dev)), "lrf_scene_rays")
        ctx.save_for_backward(ids, c2w, fo, ce)
        ctx.set_materialize_grads(False)                         # an unused output (directions without the flow loss) costs no zero fill
        ctx.meta = (R, per_view, n_rf, W, H, int(fov360))
        ctx.mark_non_differentiable(ij)
        if squeeze:                                              # one field: its rays [R,6] as the output itself (indexing the [1,R,6]
            rays = rays.view(R, 6)                               # result costs autograd a zero fill and a copy on the way back)
        return rays, dirs, ij

    @staticmethod
    def backward(ctx, g_rays, g_dirs, _g_ij):
        ids, c2w, fo, ce = ctx.saved_tensors
        R, per_view, n_rf, W, H, fov360 = ctx.meta
        dev, V = ids.device, R // per_view
        if g_rays is None:
            g_rays = torch.zeros(n_rf, R, 6, dtype=torch.float32, device=dev)
        g_rays = _f32c(g_rays)
        g_dirs = None if g_dirs is None else _f32c(g_dirs)
        g_c2w = torch.empty(V, 3, 4, dtype=torch.float32, device=dev)
        g_intr = torch.empty(V, 3, dtype=torch.float32, device=dev)        # (every entry is written: k_scene_rays_bwd, one block per view)
        g_w2rf = torch.empty(V, n_rf, 3, dtype=torch.float32, device=dev)
        N.check(N.lib().lrf_scene_rays_bwd(ids.data_ptr(), R, per_view, N.ptr(c2w), n_rf, N.ptr(fo), N.ptr(ce),
                                           W, H, fov360, N.ptr(g_rays), N.ptr(g_dirs), N.ptr(g_c2w),
                                           N.ptr(g_intr), N.ptr(g_w2rf), _stream(dev)), "lrf_scene_rays_bwd")
        g_focal = g_center = None
        if not fov360 and (ctx.needs_input_grad[3] or ctx.needs_input_grad[4]):
            s = g_intr.sum(0)
            g_focal, g_center = s[0:1], s[1:3]
        return None, g_c2w, (g_w2rf.sum(0) if ctx.needs_input_grad[2] else None), g_focal, g_center, None, None, None, None, None


class _SceneBlendFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb_f, dep_f, blend_w, exposure, per_view):
        dev = rgb_f.device
        n_rf, R = rgb_f.shape[0], rgb_f.shape[1]
        rgb_f, dep_f, bw = _f32c(rgb_f), _f32c(dep_f), _f32c(blend_w)
        ex = None if exposure is None else _f32c(exposure)
        need_bwd = any(ctx.needs_input_grad)
        rgbs = torch.empty(R, 3, dtype=torch.float32, device=dev)
        depth = torch.empty(R, dtype=torch.float32, device=dev)
        pre = torch.empty(R, 3, dtype=torch.float32, device=dev) if need_bwd else None
        N.check(N.lib().lrf_scene_blend(N.ptr(rgb_f), N.ptr(dep_f), N.ptr(bw), N.ptr(ex), R, per_view, n_rf,
                                        N.ptr(rgbs), N.ptr(depth), N.ptr(pre), _stream(dev)), "lrf_scene_blend")
        if need_bwd:
            ctx.save_for_backward(pre, bw, ex)
        ctx.set_materialize_grads(False)                         # (depth_map unused by the loss: no zero fill, the kernel takes a null pointer)
        ctx.meta = (R, per_view, n_rf)
        return rgbs, depth

    @staticmethod
    def backward(ctx, g_rgbs, g_depth):
        pre, bw, ex = ctx.saved_tensors
        R, per_view, n_rf = ctx.meta
        dev, V = pre.device, R // per_view
        g_rgbs = torch.zeros(R, 3, device=dev) if g_rgbs is None else _f32c(g_rgbs)
        g_depth = None if g_depth is None else _f32c(g_depth)
        g_rgb_f = torch.empty(n_rf, R, 3, dtype=torch.float32, device=dev)
        g_dep_f = torch.empty(n_rf, R, dtype=torch.float32, device=dev)
        g_ex = torch.empty(V, 3, 3, dtype=torch.float32, device=dev) if ex is not None else None
        N.check(N.lib().lrf_scene_blend_bwd(N.ptr(g_rgbs), N.ptr(g_depth), N.ptr(pre), N.ptr(bw), N.ptr(ex),
                                            R, per_view, n_rf, N.ptr(g_rgb_f), N.ptr(g_dep_f), N.ptr(g_ex),
                                            _stream(dev)), "lrf_scene_blend_bwd")
        return g_rgb_f, g_dep_f, None, g_ex, None


class _PoseFn(torch.autograd.Function):
    """lrf_pose_assemble / _bwd over V <= LRF_POSE_MAX frames: inputs r_0..r_{V-1}, t_0..t_{V-1}."""

    @staticmethod
    def _ptrs(tensors):
        return (C.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])

    @staticmethod
    def forward(ctx, quirk, *params):
        V = len(params) // 2
        rs = [_f32c(p) for p in params[:V]]
        ts = [_f32c(p) for p in params[V:]]
        dev = rs[0].device
        c2w = torch.empty(V, 3, 4, dtype=torch.float32, device=dev)
        N.check(N.lib().lrf_pose_assemble(_PoseFn._ptrs(rs), _PoseFn._ptrs(ts), V, int(quirk), N.ptr(c2w),
                                          _stream(dev)), "lrf_pose_assemble")
        ctx.save_for_backward(*rs)
        ctx.quirk = int(quirk)
        return c2w

    @staticmethod
    def backward(ctx, g_c2w):
        rs = ctx.saved_tensors
        V, dev = len(rs), rs[0].device
        g_r = torch.empty(V, 3, 2, dtype=torch.float32, device=dev)
        g_t = torch.empty(V, 3, dtype=torch.float32, device=dev)
        g_c = _f32c(g_c2w)
        N.check(N.lib().lrf_pose_assemble_bwd(_PoseFn._ptrs(rs), V, ctx.quirk, N.ptr(g_c), N.ptr(g_r), N.ptr(g_t),
                                              _stream(dev)), "lrf_pose_assemble_bwd")
        return (None,) + tuple(g_r.unbind(0)) + tuple(g_t.unbind(0))


def pose_assemble(r_list, t_list, cross_over_views=False):
    """[V,3,4] camera-to-world from per-frame 6D rotations [3,2] and translations [3]
    (LocalTensorfs.get_cam2world, local_tensorfs.py:292-299; sixD_to_mtx, utils/utils.py:381-388).
    cross_over_views (V == 3 only): b3 = cross(b1, b2) over the VIEW axis, which is what the
    reference's dim-less torch.cross computes for a batch of exactly three views."""
    _require_gpu(r_list[0])
    if cross_over_views and len(r_list) != 3:
        raise ValueError("cross_over_views reproduces a reference quirk that exists for exactly 3 views")
    parts = []
    for lo in range(0, len(r_list), N.LRF_POSE_MAX):
        hi = lo + N.LRF_POSE_MAX
        parts.append(_PoseFn.apply(bool(cross_over_views), *r_list[lo:hi], *t_list[lo:hi]))
    return parts[0] if len(parts) == 1 else torch.cat(parts, 0)


def _require_gpu(t):
    if not t.is_cuda:
        raise N.NativeError("localrf_amd: the scene kernels run only on an AMD GPU (HIP); got a "
                            f"{t.device} tensor. There is no CPU fallback.")


def scene_rays(ray_ids, cam2world, world2rf, focal, center, per_view, W, H, fov360=False, squeeze=False):
    """-> rays [n_rf,R,6] (origin | unnormalised direction, per field), directions [R,3], ij [R,2].
    cam2world [V,3,4] (or [V,4,4]); world2rf [n_rf,3]; focal [1] / center [2] tensors (None for 360).
    squeeze (one field only): rays come back as [R,6]."""
    _require_gpu(ray_ids)
    if ray_ids.shape[0] % per_view:
        raise ValueError("number of rays must be a multiple of the number of views")
    if squeeze and world2rf.shape[0] != 1:
        raise ValueError("squeeze needs exactly one field")
    c2w = cam2world if cam2world.shape[1] == 3 else cam2world[:, :3, :]
    return _SceneRaysFn.apply(ray_ids, c2w, world2rf, focal, center, int(per_view),
                              int(W), int(H), bool(fov360), bool(squeeze))


def scene_blend(rgb_f, dep_f, blend_w, exposure, per_view):
    """-> (clamp(E_v * sum_k w[v,k] rgb_k, 0, 1) [R,3], sum_k w[v,k] depth_k [R])."""
    _require_gpu(rgb_f)
    return _SceneBlendFn.apply(rgb_f, dep_f, blend_w, exposure, int(per_view))


def scene_forward(ray_ids, cam2world, world2rf, focal, center, per_view, W, H, fov360, fields, white_bg, floater_thresh,
                  chunk, blend_w, exposure, refine=True):
    """LocalTensorfs.forward without a tape as ONE native call (lrf_scene_fwd): rays of every active field, the per-field
    renders in the reference's chunk / field order (local_tensorfs.py:440-474), blend, exposure, clamp.
    `fields`: the active TensorVMSplit objects; world2rf [n_rf,3].  -> (rgbs [R,3], depth [R], directions [R,3], ij [R,2])."""
    _require_gpu(ray_ids)
    dev = ray_ids.device
    # the kernel reads int64 ids at unit stride (as the taped path's _SceneRaysFn coerces them): an int32 or strided
    # tensor handed over as is would be read out of bounds
    ids = ray_ids.detach().contiguous().long()
    R, n_rf = int(ids.shape[0]), len(fields)
    if R % per_view:
        raise ValueError("number of rays must be a multiple of the number of views")
    c2w = _f32c(cam2world[:, :3, :])
    w2rf = _f32c(world2rf)
    fo = None if fov360 else _f32c(focal)
    ce = None if fov360 else _f32c(center)
    bw = _f32c(blend_w)
    ex = None if exposure is None else _f32c(exposure)
    n_chunk = R if chunk <= 0 else min(int(chunk), R)
    arr = (N.LrfSceneField * n_rf)()
    keep = []
    for k, f in enumerate(fields):
        f._require_gpu(ray_ids)
        f._ensure_cache()
        z = f.z_schedule(False, -1, dev).detach().contiguous().float().view(-1)
        cf = f._c_field()
        ws = f._workspace(max(n_chunk, 1), int(z.shape[0]), dev)
        keep += [z, cf, ws]
        arr[k].field = C.pointer(cf)
        arr[k].z = z.data_ptr()
        arr[k].S = int(z.shape[0])
        arr[k].flags = f._flags(bool(white_bg)) | (N.LRF_FLAG_PE_OFF if (f.fea_pe > 0 and not refine) else 0)   # local_tensorfs.py:446 passes refine=self.is_refining
        arr[k].workspace = ws.data_ptr()
    # per-ray state of the fused form (several fields in one march + one colour launch over their field-major rays): a group
    # of up to 4 fields of one shape, chunk by chunk (chunks of a multiple of 16 rays)
    sws, sws_bytes = None, 0
    if n_rf >= 2 and n_chunk % 16 == 0 and R > 0:
        S0 = int(arr[0].S)
        if all(int(arr[k].S) == S0 for k in range(n_rf)):
            sws_bytes = int(N.lib().lrf_workspace_bytes(min(n_rf, 4) * n_chunk, S0))
            sws = _scene_workspace(dev, sws_bytes)
    rays = torch.empty(n_rf, R, 6, dtype=torch.float32, device=dev)
    rgb_f = torch.empty(n_rf, R, 3, dtype=torch.float32, device=dev)
    dep_f = torch.empty(n_rf, R, dtype=torch.float32, device=dev)
    dirs = torch.empty(R, 3, dtype=torch.float32, device=dev)
    ij = torch.empty(R, 2, dtype=torch.int64, device=dev)
    rgbs = torch.empty(R, 3, dtype=torch.float32, device=dev)
    depth = torch.empty(R, dtype=torch.float32, device=dev)
    if R:
        N.check(N.lib().lrf_scene_fwd(ids.data_ptr(), R, int(per_view), N.ptr(c2w), N.ptr(w2rf), n_rf, N.ptr(fo), N.ptr(ce),
                                      int(W), int(H), int(bool(fov360)), arr, float(floater_thresh), int(n_chunk),
                                      N.ptr(bw), N.ptr(ex), N.ptr(rays), N.ptr(rgb_f), N.ptr(dep_f), N.ptr(dirs), ij.data_ptr(),
                                      N.ptr(rgbs), N.ptr(depth), None if sws is None else sws.data_ptr(), sws_bytes,
                                      _stream(dev)), "lrf_scene_fwd")
    return rgbs, depth, dirs, ij


_scene_ws = {}


def _scene_workspace(dev, nbytes):
    """One scratch buffer per device for lrf_scene_fwd's fused launches (launches on a device's stream are serialised, as with
    the per-field workspaces)."""
    t = _scene_ws.get(dev)
    if t is None or t.numel() < nbytes:
        t = _scene_ws[dev] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    return t


class _RowsGatherFn(torch.autograd.Function):
    """out[v] = src[idx[v]] through lrf_rows_gather / _bwd (one launch each way, deterministic sums for repeated ids)."""

    @staticmethod
    def forward(ctx, src, idx):
        s = _f32c(src)
        F, K = s.shape
        V = idx.shape[0]
        out = torch.empty(V, K, dtype=torch.float32, device=s.device)
        N.check(N.lib().lrf_rows_gather(N.ptr(s), idx.data_ptr(), V, K, F, N.ptr(out), _stream(s.device)), "lrf_rows_gather")
        ctx.save_for_backward(idx)
        ctx.shape = (F, K)
        return out

    @staticmethod
    def backward(ctx, g_out):
        (idx,) = ctx.saved_tensors
        F, K = ctx.shape
        g = _f32c(g_out)
        g_src = torch.empty(F, K, dtype=torch.float32, device=g.device)
        N.check(N.lib().lrf_rows_gather_bwd(N.ptr(g), idx.data_ptr(), idx.shape[0], K, F, N.ptr(g_src), _stream(g.device)), "lrf_rows_gather_bwd")
        return g_src, None


def rows_gather(src, idx):
    """src [F, ...] indexed by a DEVICE int64 vector idx [V] along dim 0 -> [V, ...]; differentiable in src.  What
    torch.stack(params)[view_ids] (local_tensorfs.py:292-299,496) costs autograd eight launches for."""
    if not (src.is_cuda and idx.is_cuda and idx.dtype == torch.int64 and idx.dim() == 1):
        raise N.NativeError("localrf_amd: rows_gather takes a device tensor and a device int64 index vector")
    F = src.shape[0]
    out = _RowsGatherFn.apply(src.reshape(F, -1), idx.contiguous())
    return out.reshape((idx.shape[0],) + tuple(src.shape[1:]))
