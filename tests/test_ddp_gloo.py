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
