"""world_size-2 gloo tests of the data-parallel design (CPU).  The render kernels need the GPU, so
on the CPU the per-rank render is the ORACLE (oracle/vm_render_torch.py, the reference's ATen op chain
pinned to the reference goldens): a 2-rank run on a view-sharded batch + localrf_amd.dist.allreduce_grads
must produce the loss and the gradients of the 1-rank run on the whole batch (rays shard
with no forward collective, one gradient all-reduce after backward).  The same code path with the HIP
kernels and RCCL is tests/test_gpu_training.py::test_two_rank_data_parallel_step_on_one_gpu."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from util import make_field, quiet, torch_scene_chain


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class OracleScene(torch.nn.Module):
    """A one-field scene whose forward is the reference's op chain: per-view 6D poses + translations,
    pixel ids -> rays (the torch chain of tests/util.py), field render by the ATen port."""

    def __init__(self, n_views=8, W=16, H=12):
        super().__init__()
        from localrf_amd.rays import sixD_to_mtx
        self._sixd = sixD_to_mtx
        self.W, self.H = W, H
        f = quiet(make_field, [12, 14, 10], "cpu", seed=3)
        with torch.no_grad():
            for p in f.density_plane:
                p.mul_(4.0)
        self.fld_names = [k for k, v in f.state_dict().items()]
        self.field = f
        g = torch.Generator().manual_seed(4)
        self.r = torch.nn.ParameterList([torch.nn.Parameter(torch.eye(3, 2) + 0.05 * torch.randn(3, 2, generator=g))
                                         for _ in range(n_views)])
        self.t = torch.nn.ParameterList([torch.nn.Parameter(0.05 * torch.randn(3, generator=g)) for _ in range(n_views)])

    def forward(self, ray_ids, view_ids):
        from oracle import vm_render_torch as ot
        r = torch.stack([self.r[int(v)] for v in view_ids])
        t = torch.stack([self.t[int(v)] for v in view_ids])
        c2w = torch.cat([self._sixd(r), t[..., None]], -1)
        per = ray_ids.shape[0] // len(view_ids)
        focal, center = torch.tensor([10.0]), torch.tensor([self.W / 2.0, self.H / 2.0])
        rays, _, _ = torch_scene_chain(ray_ids, c2w, torch.zeros(1, 3), focal, center, per, self.W, self.H, False)
        fld = dict(self.field.state_dict(keep_vars=True))
        return ot.render_field(fld, rays[0], ot.z_schedule(60), True, 0.0)


def _batch():
    g = torch.Generator().manual_seed(1)
    view_ids = torch.tensor([0, 1, 2, 3, 4, 5, 6, 7])
    ray_ids = torch.randint(0, 16 * 12, (8 * 24,), generator=g)
    target = torch.rand(8 * 24, 3, generator=g)
    return ray_ids, view_ids, target


def _loss(model, ray_ids, view_ids, target):
    rgb, depth = model(ray_ids, view_ids)
    return ((rgb - target) ** 2).sum() + 0.1 * depth.sum()          # sum-reduced: shards add up


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from localrf_amd.dist import allreduce_grads, allreduce_scalar, shard_views
    model = OracleScene()
    ray_ids, view_ids, target = _batch()
    r_ids, v_ids = shard_views(ray_ids, view_ids)
    per = ray_ids.shape[0] // view_ids.shape[0]
    assert r_ids.shape[0] // v_ids.shape[0] == per and v_ids.shape[0] == 4
    lo = rank * 4 * per
    loss = _loss(model, r_ids, v_ids, target[lo:lo + 4 * per])
    loss.backward()
    nbytes = allreduce_grads(model)
    total = allreduce_scalar(loss, average=False)
    if rank == 0:
        torch.save({"grads": {n: (p.grad.clone() if p.grad is not None else None) for n, p in model.named_parameters()},
                    "loss": total, "bytes": nbytes}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_render_gradients_match_single_rank(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    model = OracleScene()
    ray_ids, view_ids, target = _batch()
    loss = _loss(model, ray_ids, view_ids, target)
    loss.backward()
    n_checked = 0
    for n, p in model.named_parameters():
        if not p.requires_grad:
            continue
        ref = p.grad if p.grad is not None else torch.zeros_like(p)
        a = got["grads"][n]
        assert a is not None, n
        assert float((a - ref).abs().max()) <= 2e-5 * max(float(ref.abs().max()), 1e-6), n
        n_checked += 1
    assert n_checked >= 19 + 16
    assert abs(got["loss"] - float(loss.detach())) < 1e-4 * abs(float(loss.detach()))
    assert got["bytes"] == 4 * sum(p.numel() for p in model.parameters() if p.requires_grad)


# three iterations; every iteration samples FOUR of the eight views, so that each rank gets two and four views get no
# gradient anywhere (the reference steps one tiny Adam per view: local_tensorfs.py:229-249)
_STEP_VIEWS = ([0, 1, 2, 3], [4, 5, 2, 7], [0, 6, 1, 5])


def _train_steps(model, shard=None):
    """Per-view Adam optimisers for rotations / translations + one for the field, stepped the way
    LocalTensorfs.optimizer_step does: zero_grad(set_to_none) -> backward -> [gradient all-reduce] -> step."""
    from localrf_amd.dist import allreduce_grads
    opts_r = [torch.optim.Adam([p], lr=5e-3, betas=(0.9, 0.99)) for p in model.r]
    opts_t = [torch.optim.Adam([p], lr=5e-4, betas=(0.9, 0.99)) for p in model.t]
    opt_f = torch.optim.Adam(model.field.parameters(), lr=1e-3, betas=(0.9, 0.99))
    g = torch.Generator().manual_seed(11)
    for views in _STEP_VIEWS:
        view_ids = torch.tensor(views)
        ray_ids = torch.randint(0, 16 * 12, (4 * 24,), generator=g)
        target = torch.rand(4 * 24, 3, generator=g)
        for o in opts_r + opts_t + [opt_f]:
            o.zero_grad(set_to_none=True)
        if shard is None:
            loss = _loss(model, ray_ids, view_ids, target)
        else:
            rank, world = shard
            per, n = 24, 4 // world
            loss = _loss(model, ray_ids[rank * n * per:(rank + 1) * n * per], view_ids[rank * n:(rank + 1) * n],
                         target[rank * n * per:(rank + 1) * n * per])
        loss.backward()
        if shard is not None:
            allreduce_grads(model)
        for o in opts_r + opts_t + [opt_f]:
            o.step()
    steps_r = [int(o.state[o.param_groups[0]["params"][0]].get("step", torch.tensor(0))) for o in opts_r]
    return steps_r


def _worker_steps(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = OracleScene()
    steps = _train_steps(model, shard=(rank, world))
    if rank == 0:
        torch.save({"params": {n: p.detach().clone() for n, p in model.named_parameters()}, "steps": steps}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_training_steps_match_single_rank_with_unsampled_views(tmp_path):
    """Views nobody sampled in an iteration must stay untouched under data parallelism, exactly as on one rank: their
    .grad stays None after allreduce_grads (no zero gradient is invented), so Adam neither moves them by momentum nor
    advances their step counter.  Three iterations, per-view optimisers, 2 ranks vs 1 rank: same parameters, same
    per-view step counts."""
    out = str(tmp_path / "steps.pt")
    mp.spawn(_worker_steps, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    model = OracleScene()
    steps = _train_steps(model)
    assert steps == got["steps"], (steps, got["steps"])
    assert steps == [2, 2, 2, 1, 1, 2, 1, 1]                      # how often each view was sampled
    for n, p in model.named_parameters():
        a = got["params"][n]
        assert float((a - p.detach()).abs().max()) <= 1e-5 * max(float(p.detach().abs().max()), 1e-3), n


class _BucketField(torch.nn.Module):
    """Stand-in for a TensorVMSplit after lrf_render_bwd: all gradients are views of ONE flat buffer, in three branches
    (density | appearance | network), exposed through grad_bucket() / grad_segments() as localrf_amd.field does."""

    def __init__(self, seed, fresh=True):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.dens = torch.nn.Parameter(torch.randn(37, generator=g))
        self.app = torch.nn.Parameter(torch.randn(101, generator=g))
        self.net = torch.nn.Parameter(torch.randn(13, generator=g))
        self._grad_flat, self._grad_fresh = None, False
        if fresh:
            self.fill(seed)

    def fill(self, seed):
        g = torch.Generator().manual_seed(1000 + seed)
        offs = [0, 64, 192, 256]                              # 64-float aligned views, as _native_backward lays them out
        flat = torch.zeros(offs[-1])
        ps = [self.dens, self.app, self.net]
        for p, o in zip(ps, offs):
            flat[o:o + p.numel()] = torch.randn(p.numel(), generator=g)
            p.grad = flat[o:o + p.numel()].view_as(p)
        self._grad_flat = (flat, ps, offs[-1], (0, 64, 192, 256))
        self._grad_fresh = True

    def grad_bucket(self):
        if self._grad_flat is None:
            return None
        return self._grad_flat[0][:self._grad_flat[2]], [self.dens, self.app, self.net]

    def grad_segments(self):
        d0, a0, n0, end = self._grad_flat[3]
        return [(d0, a0), (n0, end), (a0, n0)]


class _BucketScene(torch.nn.Module):
    def __init__(self, rank):
        super().__init__()
        self.done = _BucketField(5, fresh=False)              # finished field: .grad None, _grad_flat None (append_rf)
        self.live = _BucketField(7 + rank)
        g = torch.Generator().manual_seed(50 + rank)
        self.poses = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(3, generator=g)) for _ in range(4)])
        for i in ((0, 1) if rank == 0 else (1, 2)):           # view 3 is sampled by nobody
            self.poses[i].grad = torch.randn(3, generator=g)


def _worker_buckets(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from localrf_amd.dist import allreduce_grads
    res = {}
    for mode in ("flags", "hint"):
        m = _BucketScene(rank)
        hint = None if mode == "flags" else [m.poses[0], m.poses[1], m.poses[2]]
        nbytes = allreduce_grads(m, has_grad=hint)
        res[mode] = {"flat": m.live._grad_flat[0].clone(), "dens": m.live.dens.grad.clone(), "app": m.live.app.grad.clone(),
                     "net": m.live.net.grad.clone(), "poses": [None if p.grad is None else p.grad.clone() for p in m.poses],
                     "done": [p.grad for p in m.done.parameters()], "bytes": nbytes, "fresh": m.live._grad_fresh}
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


def test_chunked_field_reduction_equals_the_flat_sum_and_finished_fields_are_left_alone(tmp_path):
    """allreduce_grads reduces a field's flat gradient buffer as three collectives (density / network / appearance: the
    branches lrf_render_bwd finishes at different times) -- the result must be the plain sum of the ranks' buffers; the
    parameters of a finished field (no fresh bucket, .grad None) are neither reduced nor counted nor given a gradient;
    the has_grad hint (no flag exchange, no host synchronisation) gives the same result as the flag path, and a view no
    rank sampled keeps .grad None either way."""
    out = str(tmp_path / "b.pt")
    mp.spawn(_worker_buckets, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    a, b = _BucketScene(0), _BucketScene(1)
    want_flat = a.live._grad_flat[0] + b.live._grad_flat[0]
    for mode in ("flags", "hint"):
        r = got[mode]
        assert torch.equal(r["flat"], want_flat), mode
        assert torch.equal(r["dens"], want_flat[0:37]) and torch.equal(r["app"], want_flat[64:165]) and torch.equal(r["net"], want_flat[192:205])
        assert torch.equal(r["poses"][0], a.poses[0].grad) and torch.equal(r["poses"][2], b.poses[2].grad)
        assert torch.allclose(r["poses"][1], a.poses[1].grad + b.poses[1].grad)
        assert r["poses"][3] is None                          # sampled by nobody: no gradient is invented
        assert all(g is None for g in r["done"])
        assert r["bytes"] == 4 * (37 + 101 + 13 + 9) and r["fresh"] is False


def test_shard_views_keeps_rays_per_view_integral():
    from localrf_amd.dist import shard_views
    ray_ids, view_ids = torch.arange(16 * 10), torch.arange(16)
    seen = []
    for r in range(8):
        ri, vi = shard_views(ray_ids, view_ids, rank=r, world=8)
        assert vi.tolist() == [2 * r, 2 * r + 1] and ri.shape[0] == 20
        seen.append(ri)
    assert torch.equal(torch.cat(seen), ray_ids)
    import pytest
    with pytest.raises(ValueError):
        shard_views(ray_ids, view_ids[:15], rank=0, world=8)


class _StrayField(_BucketField):
    """A fresh field whose gradients autograd accumulated outside the flat buffer and that cannot bring them back (no
    rebucket_grads): grad_bucket() is None, every tensor of it must travel in the small bucket."""

    def fill(self, seed):
        super().fill(seed)
        for p in (self.dens, self.app, self.net):
            p.grad = p.grad.clone() + 1.0

    def grad_bucket(self):
        return None


class _StrayScene(torch.nn.Module):
    def __init__(self, rank):
        super().__init__()
        self.live = _StrayField(7 + rank)
        g = torch.Generator().manual_seed(50 + rank)
        self.poses = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(3, generator=g)) for _ in range(3)])
        for i in ((0, 1) if rank == 0 else (1,)):            # view 2 is sampled by nobody
            self.poses[i].grad = torch.randn(3, generator=g)


class _RealFieldScene(torch.nn.Module):
    """A real TensorVMSplit with the gradient bucket of its backward (field._new_grad_bucket: the product's layout), whose
    density tensors carry a regulariser: autograd summed that contribution first, so .grad of those six tensors is NOT a
    view of the flat buffer (what density_L1 does on every iteration of its phase)."""

    def __init__(self, rank):
        super().__init__()
        self.field = quiet(make_field, [10, 12, 14], "cpu", seed=3)
        keep = self.field._param_list()
        grads, _ = self.field._new_grad_bucket(keep, 5, torch.device("cpu"), plane_events=True)
        g = torch.Generator().manual_seed(100 + rank)
        self.full = []
        for i, (p, v) in enumerate(zip(keep, grads)):
            v.copy_(torch.randn(v.shape, generator=g))
            p.grad = v if i >= 6 else v.clone() + 1.0
            self.full.append(p.grad.clone())
        self.pose = torch.nn.Parameter(torch.zeros(3))
        self.pose.grad = torch.full((3,), float(rank + 1))


def _worker_strays(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from localrf_amd.dist import allreduce_grads
    res = {}
    for mode in ("flags", "hint"):
        m = _StrayScene(rank)
        hint = None if mode == "flags" else [m.poses[0], m.poses[1]]
        nbytes = allreduce_grads(m, has_grad=hint)
        res[mode] = {"dens": m.live.dens.grad.clone(), "app": m.live.app.grad.clone(), "net": m.live.net.grad.clone(),
                     "poses": [None if p.grad is None else p.grad.clone() for p in m.poses], "bytes": nbytes}
        r = _RealFieldScene(rank)
        st = {}
        nb = allreduce_grads(r, has_grad=None if mode == "flags" else [r.pose], stats=st)
        bucket = r.field.grad_bucket()
        base = r.field._grad_flat["flat"].untyped_storage().data_ptr()
        res["real_" + mode] = {"grads": [p.grad.clone() for p in r.field._param_list()], "pose": r.pose.grad.clone(), "bytes": nb,
                               "stats": st, "bucket": bucket is not None,
                               "views": all(p.grad.untyped_storage().data_ptr() == base for p in r.field._param_list())}
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


def test_fresh_field_with_gradients_outside_its_bucket_is_still_reduced(tmp_path):
    """with a regulariser in the loss autograd sums the contributions to a density tensor before it writes
    .grad, so .grad is not the view lrf_render_bwd wrote.  (a) A real TensorVMSplit brings the strays back into its flat
    buffer (rebucket_grads: one multi-tensor copy) and is reduced in place, piece by piece -- five pieces with the per-plane
    appearance split, 19 views of the buffer afterwards; (b) a field that cannot do that travels in the small bucket and
    the has_grad hint (which names poses only) must not drop it: ranks would silently diverge."""
    out = str(tmp_path / "s.pt")
    mp.spawn(_worker_strays, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    a, b = _StrayScene(0), _StrayScene(1)
    ra, rb = _RealFieldScene(0), _RealFieldScene(1)
    n_field = sum(p.numel() for p in ra.field._param_list())
    for mode in ("flags", "hint"):
        r = got[mode]
        for name in ("dens", "app", "net"):
            want = getattr(a.live, name).grad + getattr(b.live, name).grad
            assert torch.equal(r[name], want), (mode, name)
        assert torch.equal(r["poses"][0], a.poses[0].grad) and torch.allclose(r["poses"][1], a.poses[1].grad + b.poses[1].grad)
        assert r["poses"][2] is None
        assert r["bytes"] == 4 * (37 + 101 + 13 + 6)
        q = got["real_" + mode]
        assert q["bucket"] and q["views"], mode
        for i, (ga, gb_) in enumerate(zip(ra.full, rb.full)):
            assert torch.equal(q["grads"][i], ga + gb_), (mode, i)
        assert torch.equal(q["pose"], torch.full((3,), 3.0))
        assert q["bytes"] == 4 * (n_field + 3) and q["stats"]["field_bytes"] == 4 * n_field
        assert q["stats"]["collectives"] == 5 + 1 and len(q["stats"]["chunks"]) == 5      # density, network, plane 0, plane 1, plane 2 + lines; small bucket
        assert sum(q["stats"]["chunks"]) >= 4 * n_field


def _weighted_loss(model, ray_ids, view_ids, target, weights, norm):
    """train.py:369-371 with a sum in place of the mean (shards add up): 0.25 |rgb - target| w / mean(w)."""
    rgb, _ = model(ray_ids, view_ids)
    return (0.25 * (rgb - target).abs() * weights[:, None] / norm).sum()


def _worker_weighted(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from localrf_amd.dist import allreduce_grads, global_mean, shard_views
    model = OracleScene()
    ray_ids, view_ids, target = _batch()
    weights = 0.2 + 3.0 * torch.rand(ray_ids.shape[0], generator=torch.Generator().manual_seed(9)) * (torch.arange(ray_ids.shape[0]) / 96.0)
    r_ids, v_ids = shard_views(ray_ids, view_ids)
    per = ray_ids.shape[0] // view_ids.shape[0]
    lo, hi = rank * 4 * per, (rank + 1) * 4 * per
    w = weights[lo:hi]
    res = {}
    for name, norm in (("global", global_mean(w)), ("local", w.mean())):
        model.zero_grad(set_to_none=True)
        _weighted_loss(model, r_ids, v_ids, target[lo:hi], w, norm).backward()
        allreduce_grads(model)
        res[name] = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
    res["norm"] = float(global_mean(w))
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


def test_batch_global_loss_normalisation_is_rank_count_invariant(tmp_path):
    """train.py:369 divides the photometric loss by loss_weights.mean() over the WHOLE batch -- the one batch-global
    statistic of the loop (the quantile clips of train.py:406,419 are per view, and shard_views keeps views whole).
    With dist.global_mean the 2-rank gradients equal the 1-rank gradients for non-uniform weights; with the per-shard
    mean they do not (the test makes sure it would notice)."""
    out = str(tmp_path / "w.pt")
    mp.spawn(_worker_weighted, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    model = OracleScene()
    ray_ids, view_ids, target = _batch()
    weights = 0.2 + 3.0 * torch.rand(ray_ids.shape[0], generator=torch.Generator().manual_seed(9)) * (torch.arange(ray_ids.shape[0]) / 96.0)
    assert abs(got["norm"] - float(weights.mean())) < 1e-6
    _weighted_loss(model, ray_ids, view_ids, target, weights, weights.mean()).backward()
    worst_local = 0.0
    for n, p in model.named_parameters():
        if p.grad is None:
            continue
        den = max(float(p.grad.abs().max()), 1e-9)
        assert float((got["global"][n] - p.grad).abs().max()) <= 2e-5 * den, n
        worst_local = max(worst_local, float((got["local"][n] - p.grad).abs().max()) / den)
    assert worst_local > 1e-2, worst_local                   # the per-shard mean is a different loss
