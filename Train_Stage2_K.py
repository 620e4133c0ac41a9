#!/usr/bin/env python3
"""Stage-2 fine-tuning entry point (reference: Train_Stage2_K.py): mirror loss against a frozen Stage-1 teacher,
occlusion masks, 2B flip batch.  Flags follow Train_Stage2_K.py:30-71 (typed); see Train_Stage1_K.py for the
differences from the reference (synthetic input, one process per GPU, flat Adam)."""
import argparse
import json
import os
import time

parser = argparse.ArgumentParser(description='FAL_net Stage 2 on MI355X', formatter_class=argparse.ArgumentDefaultsHelpFormatter)
parser.add_argument('-maxd', '--max_disp', type=float, default=300)
parser.add_argument('-mind', '--min_disp', type=float, default=2)
parser.add_argument('-mm', '--m_model', default='FAL_netB', choices=['FAL_netA', 'FAL_netB', 'FAL_netC'])
parser.add_argument('-no_levels', '--no_levels', type=int, default=49)
parser.add_argument('-perc', '--a_p', type=float, default=0.01)
parser.add_argument('-smooth', '--a_sm', type=float, default=0.4 * 2 / 512)
parser.add_argument('-mirror_loss', '--a_mr', type=float, default=1)
parser.add_argument('-b', '--batch_size', type=int, default=4)
parser.add_argument('-ch', '--crop_height', type=int, default=192)
parser.add_argument('-cw', '--crop_width', type=int, default=640)
parser.add_argument('--lr', type=float, default=0.00005)
parser.add_argument('--beta', type=float, default=0.999)
parser.add_argument('--momentum', type=float, default=0.5)
parser.add_argument('--milestones', type=int, nargs='*', default=[5, 10])
parser.add_argument('--epochs', type=int, default=20)
parser.add_argument('--epoch_size', type=int, default=0)
parser.add_argument('--print-freq', '-p', type=int, default=100)
parser.add_argument('--start-epoch', type=int, default=0)
parser.add_argument('--fix_model', default=None, help='Stage-1 checkpoint of the frozen teacher (reference format)')
parser.add_argument('--pretrained', default=None, help='Stage-1 checkpoint to fine-tune (reference format)')
parser.add_argument('--synthetic', action='store_true')
parser.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])


def main():
    import torch
    import torch.distributed as dist
    from fal_net_amd import loss_functions as LF
    from fal_net_amd import synthetic, train
    import models

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank, local_rank = int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=dev)
    dtype = torch.bfloat16 if args.dtype == 'bf16' else torch.float32
    LF.set_compute_dtype(dtype)

    def load(path):
        return torch.load(path, map_location='cpu') if path else {'state_dict': synthetic.seeded_state_dict(args.m_model[-1], args.no_levels)}
    m_model = models.__dict__[args.m_model](load(args.pretrained), no_levels=args.no_levels, compute_dtype=dtype).to(dev).train()
    fix_model = models.__dict__[args.m_model](load(args.fix_model), no_levels=args.no_levels, compute_dtype=dtype).to(dev).eval()
    for p in fix_model.parameters():
        p.requires_grad_(False)
    train.sync_parameters(m_model)  # N > 1: rank 0's weights to every rank, once (then one gradient all-reduce per step)
    opt = train.FlatAdam(m_model, lr=args.lr, betas=(args.momentum, args.beta))
    if not args.synthetic:
        raise SystemExit('only --synthetic input is wired in this build (data pipeline out of scope, SURVEY.md 8f-3)')
    steps = args.epoch_size or 100
    pool = []
    for k in range(4):  # seeded batches resident in HBM, cycled (see Train_Stage1_K.py)
        l_, r_, _, mx_ = synthetic.synthetic_pair(args.batch_size, args.crop_height, args.crop_width, seed=4321 + rank + 977 * k, max_disp=args.max_disp)
        pool.append((l_.to(dev), r_.to(dev), mx_.to(dev)))
    for epoch in range(args.start_epoch, args.epochs):
        opt.param_groups[0]['lr'] = args.lr * (0.5 ** sum(1 for m in args.milestones if epoch >= m))
        t0 = time.time()
        for i in range(steps):
            left, right, mx = pool[i % len(pool)]
            out = train.stage2_step(m_model, fix_model, opt, left, right, mx, a_p=args.a_p, a_sm=args.a_sm,
                                    a_mr=args.a_mr, min_disp_arg=args.min_disp, max_disp_arg=args.max_disp)
            if i % args.print_freq == 0 and rank == 0:
                print(json.dumps({'epoch': epoch, 'iter': i, 'loss': float(out['loss']), 'rec': float(out['rec']),
                                  'mirror': float(out['mirror']), 'pairs_per_s': world * args.batch_size * (i + 1) / (time.time() - t0)}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    args = parser.parse_args()
    main()
