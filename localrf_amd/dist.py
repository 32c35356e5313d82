"""Ray-batch data parallelism for the render path (new: the reference is single-device,
SURVEY.md s2.3).  One process per GPU, replicated scene, each rank renders its own ray shard;
forward needs no communication.  After backward the gradients are summed in place in the flat
fp32 buffer lrf_render_bwd wrote them into, piece by piece as the backward finishes the pieces
(RCCL all-reduce over xGMI when the backend is "nccl"; the same code runs on gloo for the CPU tests)."""
import os

import torch
import torch.distributed as dist


def _forced():
    """LRF_DIST_FORCE=1: issue every collective even in a one-rank group, so that the whole exchange (side stream,
    lrf_render_bwd_wait events, RCCL calls, in-place division) runs on a single GPU exactly as it does on eight."""
    return os.environ.get("LRF_DIST_FORCE", "0") == "1"


def active(group=None, force=None):
    """True when gradients are exchanged after a backward: a process group exists and has more than one rank
    (or LRF_DIST_FORCE=1 / force=True).  TensorVMSplit's backward asks this to decide whether to run its appearance
    scatter per plane with an event behind each (LRF_FLAG_PLANE_EVENTS)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or (_forced() if force is None else bool(force))


def shard_views(ray_ids, view_ids, rank=None, world=None):
    """Split a batch by views so that ray_ids.shape[0] // view_ids.shape[0] stays integral
    (local_tensorfs.py:437).  Returns this rank's (ray_ids, view_ids)."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    V = view_ids.shape[0]
    if V % world:
        raise ValueError(f"{V} views do not split evenly over {world} ranks")
    per = ray_ids.shape[0] // V
    v0, v1 = rank * (V // world), (rank + 1) * (V // world)
    return ray_ids[v0 * per:v1 * per], view_ids[v0:v1]


def _all_reduce(t, group, async_op=False):
    """dist.all_reduce(SUM).  RCCL ("nccl") reduces device buffers in place over xGMI.  With the gloo
    backend (CPU tests, and the two-ranks-on-one-GPU test, which RCCL refuses) device tensors are
    staged through host memory."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        host = t.detach().cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        t.copy_(host)
        return None
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def global_mean(x, group=None):
    """Mean of `x` over the elements of EVERY rank's shard (sum and count all-reduced): what a batch-global statistic of
    the reference's loop becomes under ray sharding -- train.py:369 normalises the photometric loss by
    loss_weights.mean() over the whole batch; the per-shard mean would make an N-rank step differ from the 1-rank step
    whenever the weights are not uniform.  No gradient flows through it (the reference's weights are data)."""
    x = x.detach()
    pair = torch.stack([x.sum(dtype=torch.float32), torch.tensor(float(x.numel()), dtype=torch.float32, device=x.device)])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        _all_reduce(pair, group)
    return pair[0] / pair[1]


def _field_buckets(params_or_module):
    """(field, flat parameter-gradient buffer, parameters it holds) for every TensorVMSplit whose LAST backward wrote its
    gradients into one flat buffer (TensorVMSplit.grad_bucket).  A field that took no part in
    the last backward -- a finished field of a LocalTensorfs keeps its old .grad forever, nothing zeroes it -- is not
    fresh and is left alone: its stale buffer is neither reduced nor divided.  When autograd accumulated (some of) a fresh
    field's gradients outside the buffer (a regulariser in the loss), they are copied back first (rebucket_grads); a field
    for which even that fails comes back as (field, None, None) and takes the small-bucket path, every tensor of it."""
    if not isinstance(params_or_module, torch.nn.Module):
        return []
    out = []
    for m in params_or_module.modules():
        gb = getattr(m, "grad_bucket", None)
        if gb is not None and getattr(m, "_grad_fresh", False):
            b = gb()
            if b is None and hasattr(m, "rebucket_grads"):
                b = m.rebucket_grads()
            out.append((m,) + (tuple(b) if b is not None else (None, None)))
            m._grad_fresh = False
    return out


_prep_streams = {}


def _reduce_field_chunks(field, flat, group, works):
    """All-reduce one field's flat gradient buffer piece by piece (TensorVMSplit.grad_chunks): density planes / lines
    (8.7 MB at 300^3), colour network (0.1 MB), then the appearance planes one by one (8.6 MB each; lines with the last) --
    instead of one 34.8 MB collective behind the whole backward.  The backward finishes the density branch early (its own
    stream, lrf_render_bwd) and, when ranks exchange gradients, runs the appearance scatter as one pass per plane: on the
    GPU each piece is handed to the collective from a side stream that waits only for the event lrf_render_bwd recorded
    when that piece became final (lrf_render_bwd_wait), so RCCL moves finished gradients over xGMI while the rest of the
    backward still runs and only the last plane's ~ 9 MB are exposed.  Same sums as one flat all-reduce (the pieces are
    disjoint views).  Falls back to the caller's stream when the events are not those of this backward (empty batch,
    captured graph, gradients copied back by rebucket_grads) or the wait fails: every rank still issues the same
    collectives in the same order."""
    chunks = getattr(field, "grad_chunks", lambda: None)()
    if not chunks:
        segs = getattr(field, "grad_segments", lambda: None)()
        chunks = [(i, a, b) for i, (a, b) in enumerate(segs)] if segs else None
    if not chunks:
        works.append(_all_reduce(flat, group, async_op=True))
        return 1
    early = (flat.is_cuda and dist.get_backend(group) != "gloo" and hasattr(field, "_wait_bwd_bucket")
             and getattr(field, "grad_events_valid", lambda: True)())
    n = 0
    for which, a, b in chunks:
        if b <= a:
            continue
        chunk = flat[a:b]
        issued = False
        if early:
            dev = flat.device
            prep = _prep_streams.get(dev)
            if prep is None:
                prep = _prep_streams[dev] = torch.cuda.Stream(dev)
            try:
                field._wait_bwd_bucket(which, prep)           # prep waits for that piece's event only
                with torch.cuda.stream(prep):
                    works.append(_all_reduce(chunk, group, async_op=True))
                issued = True
            except Exception:                                 # noqa: BLE001 -- no events on this device: plain stream order below
                early = False
        if not issued:
            works.append(_all_reduce(chunk, group, async_op=True))
        n += 1
    return n


def allreduce_grads(params, group=None, average=False, has_grad=None, force=None, stats=None):
    """Sum (or average) .grad of `params` (module or iterable) across ranks.

    The field gradients -- 34.8 MB at 300^3, 96 MB at 500^3 -- are all-reduced IN PLACE in the flat buffer the backward
    kernels wrote them into (zero copies: the 19 parameter gradients are views of that buffer; the d/d rays tail behind
    them is rank-local and is not sent), one collective per piece of the backward (_reduce_field_chunks).  Everything
    else (poses, exposure, intrinsics: a few hundred bytes) travels in one small concatenated bucket.  A parameter that
    received no gradient on ANY rank (a view nobody sampled this iteration) keeps .grad = None, so that Adam leaves it
    and its step counter alone exactly as in a one-rank run; one that received a gradient on some rank gets the sum on
    every rank.  Which is which:

    * `has_grad` given (an iterable of the parameters OUTSIDE the fields that are differentiated on SOME rank this
      iteration; every rank passes the same set -- e.g. localrf_amd.dist.scene_has_grad from the GLOBAL view batch every
      rank knows before it takes its shard): no flags travel and the host never waits for the device.  The hint never
      removes a field's own tensors: a fresh field's gradients are always reduced;
    * otherwise one has-gradient flag per parameter rides in the small bucket (MAX over ranks) and the host reads the
      flags back -- one synchronisation per step.

    Parameters of fields that sat the backward out (finished fields of a LocalTensorfs) are never touched, whatever
    their .grad is.  Every rank issues identical collectives.  `force` (default: LRF_DIST_FORCE=1) runs all of it in a
    one-rank group too.  `stats`: a dict that receives {"field_bytes", "small_bytes", "collectives", "chunks"}.
    Returns the number of gradient bytes reduced."""
    module = params if isinstance(params, torch.nn.Module) else None
    if module is not None:
        params = [p for p in module.parameters() if p.requires_grad]
    params = list(params)
    if not params or not active(group, force):
        return 0
    world = dist.get_world_size(group)
    nbytes, covered, works, flats = 0, set(), [], []
    n_coll, chunk_bytes = 0, []
    unbucketed = set()                                        # fresh fields whose gradients could not be brought into a bucket
    always = set()                                            # ... their tensors: reduced whatever the hint says
    for field, flat, held in _field_buckets(module):
        if flat is None:
            unbucketed.add(id(field))
            always.update(id(p) for p in field.parameters())
            continue
        n_coll += _reduce_field_chunks(field, flat, group, works)
        flats.append(flat)
        covered.update(id(p) for p in held)
        nbytes += sum(p.numel() for p in held) * 4
        ch = getattr(field, "grad_chunks", lambda: None)()
        if ch:
            chunk_bytes += [(b - a) * 4 for _, a, b in ch]
    field_bytes = nbytes
    if module is not None:
        # Fields whose gradients live in a flat bucket but which took no part in this backward (every rank is in the same
        # lifecycle state): not ours to touch, whatever .grad holds (None after append_rf, a stale buffer otherwise)
        for m in module.modules():
            if getattr(m, "grad_bucket", None) is not None and hasattr(m, "_grad_flat") and id(m) not in unbucketed:
                for p in m.parameters():
                    covered.add(id(p))
    rest = [p for p in params if id(p) not in covered]
    if has_grad is not None:
        want = {id(p) for p in has_grad} | always
        rest = [p for p in rest if id(p) in want]
    if rest:
        dev = rest[0].device
        pieces = [(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(dev, torch.float32) for p in rest]
        if has_grad is None:
            pieces.append(torch.tensor([0.0 if p.grad is None else 1.0 for p in rest], dtype=torch.float32, device=dev))
        small = torch.cat(pieces)
        _all_reduce(small, group)
        n_coll += 1
        if has_grad is None:
            any_grad = small[small.numel() - len(rest):].tolist()     # the one host synchronisation of this path
        else:
            any_grad = [1.0] * len(rest)
        if average:
            small /= world
        off = 0
        for p, flag in zip(rest, any_grad):
            n = p.numel()
            if flag > 0:                                      # some rank differentiated through it
                g = small[off:off + n].view_as(p).to(p.dtype)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
                nbytes += n * 4
            off += n
    for work in works:
        if work is not None:
            work.wait()
    if average and world > 1:
        for flat in flats:
            flat /= world
    if stats is not None:
        stats.update({"field_bytes": field_bytes, "small_bytes": nbytes - field_bytes, "collectives": n_coll,
                      "chunks": chunk_bytes, "world": world})
    return nbytes


def scene_has_grad(scene, global_view_ids, optimize_poses=True):
    """The parameters of a LocalTensorfs outside its field buckets that receive a gradient on some rank when the ranks
    together render `global_view_ids` (the batch before shard_views): rotation / translation / exposure of the sampled
    frames and the intrinsics.  For allreduce_grads(..., has_grad=...)."""
    views = sorted({int(v) for v in (global_view_ids.tolist() if hasattr(global_view_ids, "tolist") else global_view_ids)})
    out = []
    for v in views:
        if optimize_poses:
            out += [scene.r_c2w[v], scene.t_c2w[v]]
        if getattr(scene, "lr_exposure_init", 0) > 0:
            out.append(scene.exposure[v])
    out += [p for p in (getattr(scene, "focal_offset", None), getattr(scene, "center_rel", None)) if p is not None]
    return [p for p in out if p.requires_grad]


def allreduce_scalar(x, group=None, average=True):
    """Loss / metric value for logging, reduced across ranks."""
    t = torch.as_tensor(x, dtype=torch.float32).detach().clone().reshape(1)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, group=group)
        if average:
            t /= dist.get_world_size(group)
    return float(t)
