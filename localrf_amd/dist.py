"""Ray-batch data parallelism for the render path (new: the reference is single-device,
SURVEY.md s2.3).  One process per GPU, replicated scene, each rank renders its own ray shard;
forward needs no communication.  After backward the gradients are summed with ONE collective
over a flattened fp32 bucket (RCCL all-reduce over xGMI when the backend is "nccl"; the same
code runs on gloo for the CPU tests)."""
import torch
import torch.distributed as dist


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


def _field_buckets(params_or_module):
    """(flat parameter-gradient buffer, parameters it holds) for every TensorVMSplit whose LAST backward wrote its
    gradients into one flat buffer that .grad still views (TensorVMSplit.grad_bucket).  A field that took no part in
    the last backward -- a finished field of a LocalTensorfs keeps its old .grad forever, nothing zeroes it -- is not
    fresh and is left alone: its stale buffer is neither reduced nor divided."""
    if not isinstance(params_or_module, torch.nn.Module):
        return []
    out = []
    for m in params_or_module.modules():
        gb = getattr(m, "grad_bucket", None)
        if gb is not None and getattr(m, "_grad_fresh", False):
            b = gb()
            if b is not None:
                out.append(b)
            m._grad_fresh = False
    return out


def allreduce_grads(params, group=None, average=False):
    """Sum (or average) .grad of `params` (module or iterable) across ranks.

    The field gradients -- 34.8 MB at 300^3, 96 MB at 500^3 -- are all-reduced IN PLACE in the flat buffer the backward
    kernels wrote them into (one collective per field that took part in the backward, zero copies: the 19 parameter
    gradients are views of that buffer; the d/d rays tail behind them is rank-local and is not sent).  Everything else
    (poses, exposure, intrinsics: a few hundred bytes) travels in one small concatenated bucket together with a
    has-gradient flag per parameter: a parameter that received no gradient on ANY rank (a view nobody sampled this
    iteration) keeps .grad = None, so that Adam leaves it and its step counter alone exactly as in a one-rank run;
    one that received a gradient on some rank gets the sum on every rank.  Every rank issues identical collectives.
    Returns the number of gradient bytes reduced."""
    module = params if isinstance(params, torch.nn.Module) else None
    if module is not None:
        params = [p for p in module.parameters() if p.requires_grad]
    params = list(params)
    if not params or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    world = dist.get_world_size(group)
    nbytes, covered, works = 0, set(), []
    for flat, held in _field_buckets(module):
        works.append((_all_reduce(flat, group, async_op=True), flat))
        covered.update(id(p) for p in held)
        nbytes += sum(p.numel() for p in held) * 4
    if module is not None:                                    # parameters of fields that sat the backward out: not ours to touch
        for m in module.modules():
            if getattr(m, "grad_bucket", None) is not None and hasattr(m, "_grad_flat"):
                for p in m.parameters():
                    if id(p) not in covered and p.grad is not None and m._grad_flat is not None \
                            and p.grad.untyped_storage().data_ptr() == m._grad_flat[0].untyped_storage().data_ptr():
                        covered.add(id(p))
    rest = [p for p in params if id(p) not in covered]
    if rest:
        dev = rest[0].device
        has = torch.tensor([0.0 if p.grad is None else 1.0 for p in rest], dtype=torch.float32, device=dev)
        small = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(dev, torch.float32)
                           for p in rest] + [has])
        _all_reduce(small, group)
        any_grad = small[small.numel() - len(rest):].tolist()
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
    for work, flat in works:
        if work is not None:
            work.wait()
        if average:
            flat /= world
    return nbytes


def allreduce_scalar(x, group=None, average=True):
    """Loss / metric value for logging, reduced across ranks."""
    t = torch.as_tensor(x, dtype=torch.float32).detach().clone().reshape(1)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, group=group)
        if average:
            t /= dist.get_world_size(group)
    return float(t)
