"""Helper of tests/test_gpu_step.py::test_data_parallel_two_ranks_one_gpu: the N > 1 code path with REAL kernels.  Two processes share
cuda:0 (RCCL refuses two ranks on one device, so the transport is gloo over CUDA tensors; everything else -- start-up broadcast + checksum,
the bucket hook fired from the weight-gradient side stream, asynchronous per-bucket all-reduce, 1/world folded into Adam -- is the path
`bench.py --gpus N` runs).  Each rank takes half of a 4-pair batch; rank 0 then repeats the step single-process on all 4 pairs: data
parallelism over equal shards of a mean loss must reproduce the full-batch update."""
import json
import os
import sys

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402


def worker(rank, world, port, out_path, dtype_name):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from fal_net_amd import loss_functions as LF
    from fal_net_amd import synthetic, train
    from fal_net_amd.models import FAL_netB
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dtype = {"f32": torch.float32, "bf16": torch.bfloat16}[dtype_name]
    LF.set_compute_dtype(dtype)
    left, right, mn, mx = synthetic.synthetic_pair(4, 64, 128, seed=77, distinct=True)
    sd = synthetic.seeded_falnetb_state_dict(49)
    if rank == 1:  # a rank that starts from OTHER weights: sync_parameters must overwrite them with rank 0's
        sd = {k: v * 1.01 for k, v in sd.items()}
    m = FAL_netB({"state_dict": sd}, no_levels=49, compute_dtype=dtype).to("cuda").train()
    assert train.sync_parameters(m)
    opt = train.FlatAdam(m, lr=1e-4)
    sl = slice(2 * rank, 2 * rank + 2)
    losses = []
    for _ in range(2):
        out = train.stage1_step(m, opt, left[sl].cuda(), right[sl].cuda(), mx[sl].cuda())
        losses.append(float(out["loss"]))
    assert getattr(m, "bucket_hook", None) is not None  # the overlapped bucket all-reduce was installed and used
    torch.cuda.synchronize()
    flat = m.flat_parameters().double()
    mine = torch.stack([flat.sum(), (flat * flat).sum()]).cpu()
    lo, hi = mine.clone(), mine.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    same = bool(torch.equal(lo, hi))
    res = {"rank": rank, "same_params_across_ranks": same, "losses": losses}
    if rank == 0:
        w_dp = m.flat_parameters().clone()
        g_dp = m.flat_gradients().clone()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:  # the same two steps, single process, all four pairs
        m1 = FAL_netB({"state_dict": synthetic.seeded_falnetb_state_dict(49)}, no_levels=49, compute_dtype=dtype).to("cuda").train()
        o1 = train.FlatAdam(m1, lr=1e-4)
        l1 = []
        for _ in range(2):
            l1.append(float(train.stage1_step(m1, o1, left.cuda(), right.cuda(), mx.cuda())["loss"]))
        torch.cuda.synchronize()
        w1, g1 = m1.flat_parameters(), m1.flat_gradients()
        res.update(single_losses=l1, w_maxabs=float((w_dp - w1).abs().max()),
                   grad_rel=float((g_dp / world - g1).norm() / g1.norm()),  # (the DP buffer holds the SUM over ranks; Adam applies 1/world)
                   grad_cos=float(torch.nn.functional.cosine_similarity(g_dp.double(), g1.double(), dim=0)))
        LF.set_compute_dtype(torch.float32)
        with open(out_path, "w") as f:
            json.dump(res, f)
    else:
        assert same


def main():
    out_path, dtype_name = sys.argv[1], sys.argv[2]
    port = 29600 + os.getpid() % 300
    mp.spawn(worker, args=(2, port, out_path, dtype_name), nprocs=2, join=True)


if __name__ == "__main__":
    main()
