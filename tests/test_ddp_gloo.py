"""CPU, world_size 2, gloo: the N>1 path of the step -- ONE all-reduce (sum) of the flat gradient buffer, then the
1/world scale folded into the optimiser -- reproduces single-process gradients of the concatenated batch."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from fal_net_amd import train


class _FlatModel:
    """Stand-in exposing the two accessors train.allreduce_gradients uses (the real module needs a GPU)."""

    def __init__(self, n):
        self.g = torch.zeros(n)

    def flat_gradients(self):
        return self.g


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    w = torch.randn(16, 5)
    x = torch.randn(world * 4, 16, generator=torch.Generator().manual_seed(1))
    shard = x[rank * 4:(rank + 1) * 4]
    wl = w.clone().requires_grad_(True)
    (shard @ wl).pow(2).mean().backward()  # local mean loss on this rank's shard of the batch
    m = _FlatModel(wl.numel())
    m.g.copy_(wl.grad.reshape(-1))
    scale = train.allreduce_gradients(m)
    if rank == 0:
        torch.save({"g": m.g * scale, "scale": scale}, out)
    dist.destroy_process_group()


def test_allreduce_of_flat_gradients_equals_big_batch(tmp_path):
    world, out = 2, str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(world, 29533, out), nprocs=world, join=True)
    got = torch.load(out)
    assert abs(got["scale"] - 0.5) < 1e-12
    torch.manual_seed(0)
    w = torch.randn(16, 5, requires_grad=True)
    x = torch.randn(world * 4, 16, generator=torch.Generator().manual_seed(1))
    (x @ w).pow(2).mean().backward()
    assert torch.allclose(got["g"], w.grad.reshape(-1), atol=1e-6)


def test_single_process_is_identity():
    m = _FlatModel(8)
    m.g.fill_(3.0)
    assert train.allreduce_gradients(m) == 1.0 and float(m.g.sum()) == 24.0


def _worker_buckets(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 1000
    m = _FlatModel(n)
    m.g.copy_(torch.arange(n, dtype=torch.float32) * (rank + 1))
    train.enable_overlapped_allreduce(m)
    assert m.bucket_hook is not None
    # backward finalises contiguous ranges of the flat buffer back to front (FAL_net.gradient_buckets): one async all-reduce each
    for i, (lo, hi) in enumerate([(700, n), (300, 700), (10, 300), (0, 10)]):
        m.bucket_hook(i, m.g[lo:hi])
    scale = train.allreduce_gradients(m)  # waits for the pending bucket reductions; no second collective
    if rank == 0:
        torch.save({"g": m.g.clone(), "scale": scale, "pending": len(m._pending_reduces)}, out)
    dist.destroy_process_group()


def test_bucketed_allreduce_covers_the_flat_buffer_once(tmp_path):
    world, out = 2, str(tmp_path / "r0b.pt")
    mp.spawn(_worker_buckets, args=(world, 29534, out), nprocs=world, join=True)
    got = torch.load(out)
    assert got["scale"] == 0.5 and got["pending"] == 0
    assert torch.equal(got["g"], torch.arange(1000, dtype=torch.float32) * 3)  # rank0 (x1) + rank1 (x2), every element exactly once
