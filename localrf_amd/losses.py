"""Geometric losses around the render path, as HIP kernels behind the reference's expressions.

  flow_loss(...)   train.py:385-412 with utils/utils.py:15-48 (pts2px, inverse_pose, get_cam2cams, get_fwd_bwd_cam2cams,
                   get_pred_flow): returns `flow_loss_arr.mean()` after the 0.9-quantile clipping, i.e. the value
                   train.py multiplies by loss_flow_weight * reg_loss_weight / ((W + H) / 2)
  depth_loss(...)  train.py:414-423 with compute_depth_loss (utils/utils.py:50-59): returns `depth_loss_arr.mean()`
                   after the 0.8-quantile clipping

Both take what `LocalTensorfs.forward` returns (depth_map, directions, ij) and are differentiable with respect to
depth_map (-> field and poses), directions, cam2world (-> poses) and focal / center.  One workgroup per view; the
per-view torch.median / torch.quantile come from an LDS sort (csrc/lrf_losses.inl).  No torch fallback.
"""
import ctypes as C

import torch

from . import _native as N
from .scene_ops import _f32c, _stream


def _i32(t, dev):
    """int32 on the device.  Host ids are staged through pinned memory: a pageable host->device copy blocks the host
    until the stream has drained (docs/GFX950_FINDINGS.md finding 7), which would serialise the iteration."""
    if not torch.is_tensor(t):
        t = torch.as_tensor(t)
    if t.is_cuda:
        return t.to(dtype=torch.int32).contiguous()
    stage = torch.empty(t.shape, dtype=torch.int32, pin_memory=True)
    stage.copy_(t)
    return stage.to(dev, non_blocking=True)


class _FlowLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, dirs, cam2world, focal, center, ij, frame, fwd_off, fwd_flow, fwd_mask, bwd_flow, bwd_mask, q):
        dev = depth.device
        V, n = depth.shape
        a = N.LrfFlowLoss()
        keep = [_f32c(cam2world), frame, _f32c(dirs), _f32c(depth), ij.contiguous(), _f32c(fwd_flow), _f32c(fwd_mask),
                _f32c(bwd_flow), _f32c(bwd_mask), _f32c(focal).reshape(-1), _f32c(center).reshape(-1), fwd_off]
        if keep[0].dim() != 3 or keep[0].shape[1:] != (3, 4):
            raise ValueError("cam2world must be [F,3,4]")
        if keep[4].dtype != torch.int64:
            raise ValueError("ij must be int64 (LocalTensorfs.forward's fourth output)")
        for name, t in zip(("cam2world", "frame", "dirs", "depth", "ij", "fwd_flow", "fwd_mask", "bwd_flow", "bwd_mask",
                            "focal", "center", "fwd_off"), keep):
            setattr(a, name, t.data_ptr())
        a.F, a.V, a.n, a.quantile = keep[0].shape[0], V, n, float(q)
        arr = torch.empty(V, n, dtype=torch.float32, device=dev)
        vsum = torch.empty(V, dtype=torch.float32, device=dev)
        N.check(N.lib().lrf_flow_loss_fwd(C.byref(a), N.ptr(arr), N.ptr(vsum), _stream(dev)), "lrf_flow_loss_fwd")
        ctx.args, ctx.keep, ctx.arr = a, [t.detach() for t in keep], arr     # (kept alive for the raw pointers in `a`)
        ctx.focal_shape = focal.shape
        ctx.mark_non_differentiable(arr)
        return vsum.sum() / float(V * n), arr

    @staticmethod
    def backward(ctx, g_loss, _g_arr):
        a, keep, arr = ctx.args, ctx.keep, ctx.arr
        dev = arr.device
        V, n = arr.shape
        g_depth = torch.empty(V, n, dtype=torch.float32, device=dev)
        g_dirs = torch.empty(V, n, 3, dtype=torch.float32, device=dev)
        g_c2w = torch.empty(a.F, 3, 4, dtype=torch.float32, device=dev)
        g_intr = torch.empty(V, 3, dtype=torch.float32, device=dev)
        ws = torch.empty(V * 36, dtype=torch.float32, device=dev)
        g = _f32c(g_loss).reshape(1)
        N.check(N.lib().lrf_flow_loss_bwd(C.byref(a), N.ptr(arr), N.ptr(g), 1.0 / float(V * n), N.ptr(g_depth), N.ptr(g_dirs),
                                          N.ptr(g_c2w), N.ptr(g_intr), N.ptr(ws), _stream(dev)), "lrf_flow_loss_bwd")
        if not (ctx.needs_input_grad[3] or ctx.needs_input_grad[4]):   # intrinsics without a tape (LocalTensorfs.freeze_intrinsics)
            return g_depth, g_dirs, g_c2w, None, None, None, None, None, None, None, None, None, None
        s = g_intr.sum(0)
        return g_depth, g_dirs, g_c2w, s[0:1].reshape(ctx.focal_shape), s[1:3], None, None, None, None, None, None, None, None


def flow_loss(depth_map, directions, ij, cam2world, view_ids, starting_frame_id, fwd_flow, fwd_mask, bwd_flow, bwd_mask,
              focal, center, quantile=0.9, return_arr=False):
    """`flow_loss_arr.mean()` of train.py:385-410.  depth_map [V*n] or [V,n]; directions [V*n,3]; ij [V*n,2] int64;
    cam2world = local_tensorfs.get_cam2world(starting_id=starting_frame_id) [F,3,4]; view_ids [V]; flows [V*n,2],
    masks [V*n]; focal = local_tensorfs.focal(W) (tensor [1] or float), center = local_tensorfs.center(W, H) [2]."""
    dev = depth_map.device
    if dev.type != "cuda":
        raise N.NativeError("localrf_amd.losses: tensors must be on the GPU (there is no CPU fallback)")
    view_ids = torch.as_tensor(view_ids)
    V = int(view_ids.shape[0])
    depth = depth_map.reshape(V, -1)
    n = depth.shape[1]
    if n > N.LRF_LOSS_MAX_PER_VIEW:
        raise ValueError(f"at most {N.LRF_LOSS_MAX_PER_VIEW} rays per view")
    if view_ids.device.type == "cpu" and V:                       # (ids on the device are not checked: that would synchronise)
        lo, hi = int(view_ids.min()) - int(starting_frame_id), int(view_ids.max()) - int(starting_frame_id)
        if lo < 0 or hi >= int(cam2world.shape[0]):
            raise IndexError(f"view ids {int(view_ids.min())}..{int(view_ids.max())} outside cam2world[{starting_frame_id}:"
                             f"{starting_frame_id + int(cam2world.shape[0])}]")
    # train.py:396 compares the ABSOLUTE view id with the length of the cam2world slice; reproduced as is
    if view_ids.device.type == "cpu":                              # one staged upload: [frame index | forward-mask-off flag]
        both = _i32(torch.stack([view_ids.to(torch.int64) - int(starting_frame_id),
                                 (view_ids == int(cam2world.shape[0]) - 1).to(torch.int64)]), dev)
        frame, fwd_off = both[0], both[1]
    else:
        ids = _i32(view_ids, dev)
        frame = ids - int(starting_frame_id)
        fwd_off = (ids == int(cam2world.shape[0]) - 1).to(torch.int32)
    if not torch.is_tensor(focal):
        focal = torch.tensor([float(focal)], device=dev)
    loss, arr = _FlowLossFn.apply(depth, directions.reshape(V, n, 3), cam2world, focal, center, ij.reshape(V, n, 2), frame.contiguous(), fwd_off.contiguous(),
                                  fwd_flow.reshape(V, n, 2), fwd_mask.reshape(V, n).float(), bwd_flow.reshape(V, n, 2),
                                  bwd_mask.reshape(V, n).float(), quantile)
    return (loss, arr) if return_arr else loss


class _DepthLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, gt, q):
        dev = depth.device
        V, n = depth.shape
        d, g = _f32c(depth), _f32c(gt)
        arr = torch.empty(V, n, dtype=torch.float32, device=dev)
        stats = torch.empty(V, 6, dtype=torch.float32, device=dev)
        vsum = torch.empty(V, dtype=torch.float32, device=dev)
        N.check(N.lib().lrf_depth_loss_fwd(N.ptr(d), N.ptr(g), V, n, float(q), N.ptr(arr), N.ptr(stats), N.ptr(vsum), _stream(dev)),
                "lrf_depth_loss_fwd")
        ctx.keep = (d.detach(), g.detach(), arr, stats)
        ctx.mark_non_differentiable(arr)
        return vsum.sum() / float(V * n), arr

    @staticmethod
    def backward(ctx, g_loss, _g_arr):
        d, g, arr, stats = ctx.keep
        V, n = arr.shape
        g_depth = torch.empty_like(d)
        gl = _f32c(g_loss).reshape(1)
        N.check(N.lib().lrf_depth_loss_bwd(N.ptr(d), N.ptr(g), V, n, N.ptr(arr), N.ptr(stats), N.ptr(gl), 1.0 / float(V * n),
                                           N.ptr(g_depth), _stream(d.device)), "lrf_depth_loss_bwd")
        return g_depth, None, None


def depth_loss(depth_map, invdepths, n_views, quantile=0.8, return_arr=False):
    """`depth_loss_arr.mean()` of train.py:414-421: compute_depth_loss(1 / depth_map.clamp(1e-6), invdepths) per view,
    entries above the view's 0.8-quantile zeroed."""
    dev = depth_map.device
    if dev.type != "cuda":
        raise N.NativeError("localrf_amd.losses: tensors must be on the GPU (there is no CPU fallback)")
    depth = depth_map.reshape(int(n_views), -1)
    if depth.shape[1] > N.LRF_LOSS_MAX_PER_VIEW:
        raise ValueError(f"at most {N.LRF_LOSS_MAX_PER_VIEW} rays per view")
    loss, arr = _DepthLossFn.apply(depth, invdepths.reshape(depth.shape), quantile)
    return (loss, arr) if return_arr else loss


class _PhotoLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rgb, target, weights, w_mean):
        r, t = _f32c(rgb), _f32c(target)
        w = None if weights is None else _f32c(weights).reshape(-1)
        wm = None if w_mean is None else _f32c(w_mean).reshape(-1)
        dev = r.device
        out = torch.empty(2, dtype=torch.float32, device=dev)
        N.check(N.lib().lrf_photo_loss_fwd(N.ptr(r), N.ptr(t), N.ptr(w), N.ptr(wm), r.shape[0], N.ptr(out[0:1]), N.ptr(out[1:2]), _stream(dev)),
                "lrf_photo_loss_fwd")
        ctx.keep = (r.detach(), t.detach(), w, out)
        return out[0]

    @staticmethod
    def backward(ctx, g_loss):
        r, t, w, out = ctx.keep
        g = _f32c(g_loss).reshape(1)
        g_rgb = torch.empty_like(r)
        N.check(N.lib().lrf_photo_loss_bwd(N.ptr(r), N.ptr(t), N.ptr(w), N.ptr(out[1:2]), N.ptr(g), r.shape[0], N.ptr(g_rgb), _stream(r.device)),
                "lrf_photo_loss_bwd")
        return g_rgb, None, None, None


def photometric_loss(rgb_map, rgb_train, loss_weights=None, weights_mean=None):
    """train.py:369-371: `(0.25 * |rgb_map - rgb_train| * loss_weights / loss_weights.mean()).mean()` as one launch each way
    (differentiable in rgb_map).  rgb_map, rgb_train [R,3]; loss_weights [R] / [R,1] or None (ones); weights_mean: a device
    scalar to divide by instead of this batch's own mean -- under ray sharding the batch-global mean
    (localrf_amd.dist.global_mean), so that an N-rank step equals the 1-rank step."""
    if rgb_map.device.type != "cuda":
        raise N.NativeError("localrf_amd.losses: tensors must be on the GPU (there is no CPU fallback)")
    if rgb_map.dim() != 2 or rgb_map.shape[1] != 3 or rgb_train.shape != rgb_map.shape:
        raise ValueError("rgb_map and rgb_train must both be [R,3]")
    if loss_weights is not None and loss_weights.numel() != rgb_map.shape[0]:
        raise ValueError("loss_weights must hold one weight per ray")
    return _PhotoLossFn.apply(rgb_map, rgb_train, loss_weights, weights_mean)
