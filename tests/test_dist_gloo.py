"""world_size-2 gloo test of the data-parallel helpers (CPU): a 2-rank run on a sharded batch
must produce the gradients of the 1-rank run on the whole batch (SURVEY.md s4 'distributed')."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _toy():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.ReLU(), torch.nn.Linear(16, 3))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from localrf_amd.dist import allreduce_grads, allreduce_scalar, shard_views
    model = _toy()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(8 * 32, 6, generator=g)             # 8 views x 32 rays
    ray_ids = torch.arange(8 * 32)
    view_ids = torch.arange(8)
    r_ids, v_ids = shard_views(ray_ids, view_ids)
    assert r_ids.shape[0] // v_ids.shape[0] == 32
    loss = model(x[r_ids]).square().sum()                # sum-reduced loss: shards add up
    loss.backward()
    nbytes = allreduce_grads(model)
    total = allreduce_scalar(loss, average=False)
    if rank == 0:
        torch.save({"grads": [p.grad.clone() for p in model.parameters()], "loss": total,
                    "bytes": nbytes}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradients_match_single_rank(tmp_path):
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    model = _toy()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(8 * 32, 6, generator=g)
    loss = model(x).square().sum()
    loss.backward()
    for a, p in zip(got["grads"], model.parameters()):
        assert torch.allclose(a, p.grad, rtol=1e-5, atol=1e-6)
    assert abs(got["loss"] - float(loss)) < 1e-3 * abs(float(loss))
    assert got["bytes"] == 4 * sum(p.numel() for p in model.parameters())
