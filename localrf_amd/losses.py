"""Geometric losses around the render path, as HIP kernels behind the reference's expressions.

  flow_loss(...)   train.py:385-412 with utils/utils.py:15-48 (pts2px, inverse_pose, get_cam2cams, get_fwd_bwd_cam2cams,
                   get_pred_flow): returns `flow_loss_arr.mean()` after the 0.9-quantile clipping, i.e. the value
                   train.py multiplies by loss_flow_weight * reg_loss_weight / ((W + H) / 2)
  depth_loss(...)  train.py:414-423 with compute_depth_loss (utils/utils.py:50-59): returns `depth_loss_arr.mean()`
                   after the 0.8-quantile clipping

  batch_gather(...) train.py:352-358,385-420: the batch's target colours, flows, flow masks and inverse depths out of the
                   dataset tensors in one launch
  combine(...)     train.py:425-437: the weighted sum of the iteration's loss terms, one launch each way

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
    def forward(ctx, depth, dirs, cam2world, focal, center, ij, frame, fwd_off, fwd_flow, fwd_mask, bwd_flow, bwd_mask, q, per_view=False):
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
        ctx.per_view = bool(per_view)
        return (vsum if per_view else vsum.sum() / float(V * n)), arr

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
        # per_view: the output was the V per-view sums, every one of which enters the total with the same weight (combine):
        # the first element of the incoming gradient is that weight
        g = _f32c(g_loss[0:1] if ctx.per_view else g_loss).reshape(1)
        N.check(N.lib().lrf_flow_loss_bwd(C.byref(a), N.ptr(arr), N.ptr(g), 1.0 if ctx.per_view else 1.0 / float(V * n), N.ptr(g_depth), N.ptr(g_dirs),
                                          N.ptr(g_c2w), N.ptr(g_intr), N.ptr(ws), _stream(dev)), "lrf_flow_loss_bwd")
        if not (ctx.needs_input_grad[3] or ctx.needs_input_grad[4]):   # intrinsics without a tape (LocalTensorfs.freeze_intrinsics)
            return g_depth, g_dirs, g_c2w, None, None, None, None, None, None, None, None, None, None, None
        s = g_intr.sum(0)
        return g_depth, g_dirs, g_c2w, s[0:1].reshape(ctx.focal_shape), s[1:3], None, None, None, None, None, None, None, None, None


def flow_loss(depth_map, directions, ij, cam2world, view_ids, starting_frame_id, fwd_flow, fwd_mask, bwd_flow, bwd_mask,
              focal, center, quantile=0.9, return_arr=False, per_view=False, frame_ids=None):
    """`flow_loss_arr.mean()` of train.py:385-410.  depth_map [V*n] or [V,n]; directions [V*n,3]; ij [V*n,2] int64;
    cam2world = local_tensorfs.get_cam2world(starting_id=starting_frame_id) [F,3,4]; view_ids [V]; flows [V*n,2],
    masks [V*n]; focal = local_tensorfs.focal(W) (tensor [1] or float), center = local_tensorfs.center(W, H) [2].
    per_view: return the V per-view sums of the clipped array instead of its mean (mean = sum / (V n): `combine` folds that
    factor into the term's weight).  frame_ids: int32 device [2, V] = (view - starting_frame_id, view == F - 1) prepared by the
    caller (the captured iteration stages it with its other inputs) -- otherwise formed here from view_ids."""
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
    if frame_ids is not None:
        frame, fwd_off = frame_ids[0], frame_ids[1]
    elif view_ids.device.type == "cpu":                            # one staged upload: [frame index | forward-mask-off flag]
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
                                  bwd_mask.reshape(V, n).float(), quantile, per_view)
    return (loss, arr) if return_arr else loss


class _DepthLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, gt, q, per_view=False):
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
        ctx.per_view = bool(per_view)
        return (vsum if per_view else vsum.sum() / float(V * n)), arr

    @staticmethod
    def backward(ctx, g_loss, _g_arr):
        d, g, arr, stats = ctx.keep
        V, n = arr.shape
        g_depth = torch.empty_like(d)
        gl = _f32c(g_loss[0:1] if ctx.per_view else g_loss).reshape(1)
        N.check(N.lib().lrf_depth_loss_bwd(N.ptr(d), N.ptr(g), V, n, N.ptr(arr), N.ptr(stats), N.ptr(gl), 1.0 if ctx.per_view else 1.0 / float(V * n),
                                           N.ptr(g_depth), _stream(d.device)), "lrf_depth_loss_bwd")
        return g_depth, None, None, None


def depth_loss(depth_map, invdepths, n_views, quantile=0.8, return_arr=False, per_view=False):
    """`depth_loss_arr.mean()` of train.py:414-421: compute_depth_loss(1 / depth_map.clamp(1e-6), invdepths) per view,
    entries above the view's 0.8-quantile zeroed.  per_view: the n_views per-view sums instead of the mean (see flow_loss)."""
    dev = depth_map.device
    if dev.type != "cuda":
        raise N.NativeError("localrf_amd.losses: tensors must be on the GPU (there is no CPU fallback)")
    depth = depth_map.reshape(int(n_views), -1)
    if depth.shape[1] > N.LRF_LOSS_MAX_PER_VIEW:
        raise ValueError(f"at most {N.LRF_LOSS_MAX_PER_VIEW} rays per view")
    loss, arr = _DepthLossFn.apply(depth, invdepths.reshape(depth.shape), quantile, per_view)
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


def batch_gather(view_ids, pix, images=None, fwd_flow=None, bwd_flow=None, invdepths=None):
    """The batch's rows of the dataset tensors in one launch (train.py:352-358 `rgb_train`, :385-420 the flows, their masks
    and the inverse depths; the reference indexes each tensor with the (view, pixel) ids and forms the masks with tensor
    expressions).  view_ids int64 [V], pix int64 [V, n] (pixel ids inside the view), both on the device; dataset tensors
    [n_images, H*W, 3 | 2 | 2] and [n_images, H*W] float32 on the device, None = not wanted.  Returns a dict: "target" [V n, 3],
    "fwd_flow" / "bwd_flow" [V n, 2], "fwd_mask" / "bwd_mask" [V n] (1 where the view has a next / previous image),
    "invdepths" [V n].  Not differentiable (the dataset is data)."""
    dev = pix.device
    if dev.type != "cuda":
        raise N.NativeError("localrf_amd.losses: tensors must be on the GPU (there is no CPU fallback)")
    if view_ids.dtype != torch.int64 or pix.dtype != torch.int64:
        raise ValueError("view_ids and pix must be int64")
    V, n = int(pix.shape[0]), int(pix.shape[1])
    if int(view_ids.numel()) != V:
        raise ValueError("one view id per row of pix")
    srcs = {"images": images, "fwd_flow": fwd_flow, "bwd_flow": bwd_flow, "invdepths": invdepths}
    ref = next((t for t in srcs.values() if t is not None), None)
    if ref is None:
        raise ValueError("no dataset tensor given")
    a = N.LrfBatchGather()
    keep = [view_ids.contiguous(), pix.contiguous()]
    a.view_ids, a.pix = keep[0].data_ptr(), keep[1].data_ptr()
    a.V, a.n, a.HW, a.n_images = V, n, int(ref.shape[1]), int(ref.shape[0])
    for name, t in srcs.items():
        if t is not None:
            if t.dtype != torch.float32 or not t.is_contiguous() or int(t.shape[0]) != a.n_images or int(t.shape[1]) != a.HW:
                raise ValueError(f"{name}: contiguous float32 [n_images, H*W, ...] expected")
            setattr(a, name, t.data_ptr())
    out = {}
    if images is not None:
        out["target"] = torch.empty(V * n, 3, dtype=torch.float32, device=dev)
    if fwd_flow is not None:
        out["fwd_flow"] = torch.empty(V * n, 2, dtype=torch.float32, device=dev)
        out["fwd_mask"] = torch.empty(V * n, dtype=torch.float32, device=dev)
    if bwd_flow is not None:
        out["bwd_flow"] = torch.empty(V * n, 2, dtype=torch.float32, device=dev)
        out["bwd_mask"] = torch.empty(V * n, dtype=torch.float32, device=dev)
    if invdepths is not None:
        out["invdepths"] = torch.empty(V * n, dtype=torch.float32, device=dev)
    N.check(N.lib().lrf_batch_gather(C.byref(a), N.ptr(out.get("target")), N.ptr(out.get("fwd_flow")), N.ptr(out.get("fwd_mask")),
                                     N.ptr(out.get("bwd_flow")), N.ptr(out.get("bwd_mask")), N.ptr(out.get("invdepths")), _stream(dev)),
            "lrf_batch_gather")
    return out


class _CombineFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, s, coef, *xs):
        dev = xs[0].device
        t = N.LrfLossTerms()
        keep = [_f32c(x).reshape(-1) for x in xs]
        for k, (x, (a, b)) in enumerate(zip(keep, coef)):
            t.x[k], t.n[k], t.a[k], t.b[k] = x.data_ptr(), int(x.numel()), float(a), float(b)
        t.count = len(keep)
        sk = None if s is None else _f32c(s).reshape(-1)
        t.s = None if sk is None else sk.data_ptr()
        out = torch.empty(1 + N.LRF_LOSS_TERMS_MAX, dtype=torch.float32, device=dev)
        N.check(N.lib().lrf_loss_combine_fwd(C.byref(t), N.ptr(out[0:1]), N.ptr(out[1:]), _stream(dev)), "lrf_loss_combine_fwd")
        ctx.w, ctx.shapes = out[1:], [x.shape for x in xs]
        return out[0]

    @staticmethod
    def backward(ctx, g_total):
        n = len(ctx.shapes)
        g = torch.empty(n, dtype=torch.float32, device=ctx.w.device)
        gt = _f32c(g_total).reshape(1)
        N.check(N.lib().lrf_loss_combine_bwd(N.ptr(ctx.w), N.ptr(gt), n, N.ptr(g), _stream(g.device)), "lrf_loss_combine_bwd")
        return (None, None) + tuple(g[k].expand(sh) for k, sh in enumerate(ctx.shapes))


def combine(terms, s=None):
    """total = sum_k (a_k + b_k s) sum(x_k): the loss assembly of train.py:425-437 as one launch each way.  terms: up to 8
    (x_k, a_k, b_k) with x_k a device scalar or a vector of partial sums (flow_loss / depth_loss with per_view=True) and a_k,
    b_k host floats; s: a device scalar (the schedule weight of the iteration, lr_factor ** rf_iter) or None.  E.g.
    combine([(photo, 1, 0), (flow_sums, 0, w_flow / ((W + H) / 2) / (V n)), (depth_sums, 0, w_depth / (V n)), (l1, w_l1, 0)], reg_w).
    Differentiable in every x_k."""
    if not 0 < len(terms) <= N.LRF_LOSS_TERMS_MAX:
        raise ValueError(f"1..{N.LRF_LOSS_TERMS_MAX} terms")
    if terms[0][0].device.type != "cuda":
        raise N.NativeError("localrf_amd.losses: tensors must be on the GPU (there is no CPU fallback)")
    return _CombineFn.apply(s, tuple((float(a), float(b)) for _, a, b in terms), *[x for x, _, _ in terms])
