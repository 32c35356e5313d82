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


def allreduce_grads(params, group=None, average=False):
    """Sum (or average) .grad of `params` (module or iterable) across ranks with a single
    all-reduce of one flattened bucket.  Parameters without a grad contribute zeros so every
    rank issues an identical collective."""
    if isinstance(params, torch.nn.Module):
        params = [p for p in params.parameters() if p.requires_grad]
    params = list(params)
    if not params or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    dev = params[0].device
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1).to(dev, torch.float32)
                      for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= dist.get_world_size(group)
    off = 0
    for p in params:
        n = p.numel()
        g = flat[off:off + n].view_as(p).to(p.dtype)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += n
    return flat.numel() * 4


def allreduce_scalar(x, group=None, average=True):
    """Loss / metric value for logging, reduced across ranks."""
    t = torch.as_tensor(x, dtype=torch.float32).detach().clone().reshape(1)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, group=group)
        if average:
            t /= dist.get_world_size(group)
    return float(t)
