#!/usr/bin/env python3
"""Stage-1 training entry point (reference: Train_Stage1_K.py) on the MI355X implementation.

Same flag names and defaults as the reference (Train_Stage1_K.py:30-70), with types added (the reference's
untyped flags only work at their defaults).  Differences, all outside the hot path:
  * real data (`-d <root>`): the pairs of `--train_list` are decoded by `--workers` loader processes (fal_net_amd.datasets),
    uploaded as uint8 and augmented ON THE GPU (fal_net_amd.data_transforms.StereoAugment: the reference's co_transform +
    input_transform chain, Train_Stage1_K.py:116-128); every epoch ends with validate() on KITTI 2015 (:279-347) when
    `<root>/<vdataName>` holds it, and the best RMSE keeps `model_best.pth.tar` (:199-207);
  * `--synthetic`: seeded random stereo pairs of the crop size, so the script also runs with no KITTI files;
  * data parallelism is one process per GPU (`torchrun --nproc-per-node N Train_Stage1_K.py ...`) with ONE RCCL
    all-reduce of the flat gradient buffer per step, instead of nn.DataParallel (Train_Stage1_K.py:172);
  * logging is stdout / JSON lines; losses are read back every `--print-freq` steps only (the reference syncs twice
    per step, :249,259).
"""
import argparse
import datetime
import json
import os
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "5")  # application-level choice, before HIP initialises (fal_net_amd/__init__.py)

parser = argparse.ArgumentParser(description='FAL_net Stage 1 on MI355X', formatter_class=argparse.ArgumentDefaultsHelpFormatter)
parser.add_argument('-d', '--data', metavar='DIR', default=None, help='path to dataset (unused with --synthetic)')
parser.add_argument('-n0', '--dataName0', default='Kitti')
parser.add_argument('-train_split', '--train_split', default='eigen_train_split')
parser.add_argument('-vdn', '--vdataName', default='Kitti2015')
parser.add_argument('-relbase_test', '--rel_baset', type=float, default=1)
parser.add_argument('-maxd', '--max_disp', type=float, default=300)
parser.add_argument('-mind', '--min_disp', type=float, default=2)
parser.add_argument('-gpu_no', '--gpu_no', default='0')
parser.add_argument('-mm', '--m_model', default='FAL_netB', choices=['FAL_netA', 'FAL_netB', 'FAL_netC'])
parser.add_argument('-no_levels', '--no_levels', type=int, default=49)
parser.add_argument('-perc', '--a_p', type=float, default=0.01)
parser.add_argument('-smooth', '--a_sm', type=float, default=0.2 * 2 / 512)
parser.add_argument('-w', '--workers', type=int, default=4)
parser.add_argument('-b', '--batch_size', type=int, default=8)
parser.add_argument('-ch', '--crop_height', type=int, default=192)
parser.add_argument('-cw', '--crop_width', type=int, default=640)
parser.add_argument('-tbs', '--tbatch_size', type=int, default=1)
parser.add_argument('-op', '--optimizer', default='adam')
parser.add_argument('--lr', type=float, default=0.0001)
parser.add_argument('--beta', type=float, default=0.999)
parser.add_argument('--momentum', type=float, default=0.5)
parser.add_argument('--milestones', type=int, nargs='*', default=[30, 40])
parser.add_argument('--weight-decay', '--wd', type=float, default=0.0)
parser.add_argument('--bias-decay', type=float, default=0.0)
parser.add_argument('--epochs', type=int, default=50)
parser.add_argument('--epoch_size', type=int, default=0)
parser.add_argument('--sparse', default=True, action='store_true')
parser.add_argument('--print-freq', '-p', type=int, default=100)
parser.add_argument('--start-epoch', type=int, default=0)
parser.add_argument('--pretrained', default=None, help='checkpoint.pth.tar in the reference format')
# --- additions ---
parser.add_argument('--synthetic', action='store_true', help='seeded random pairs instead of a dataset')
parser.add_argument('--gpu-augment', action='store_true',
                    help='synthetic mode: start from KITTI-sized uint8 "decoded" pairs and run the reference augmentation chain on '
                         'the GPU every step (fal_net_amd.data_transforms.StereoAugment) instead of cycling pre-made float batches')
parser.add_argument('--dtype', default='bf16', choices=['bf16', 'f16', 'f32'], help='compute dtype (f32 = exact-f32 MFMA parity path)')
parser.add_argument('--train_list', default=os.path.join('Datasets', 'kitti_eigen_train.txt'),
                    help="training pairs, one 'left right' pair of paths relative to <data>/<dataName0> per line (the reference opens this "
                         "file relative to the working directory, Datasets/Kitti.py:37)")
parser.add_argument('--save-path', default=None)


def main(step='stage1_step'):
    import torch
    import torch.distributed as dist
    from fal_net_amd import loss_functions as LF
    from fal_net_amd import myUtils as utils
    from fal_net_amd import synthetic, train
    import models

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank, local_rank = int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=dev)
    if args.weight_decay or args.bias_decay:
        raise SystemExit('weight decay is 0 in the reference defaults; the fused flat Adam implements wd=0 only')
    dtype = {'bf16': torch.bfloat16, 'f16': torch.float16, 'f32': torch.float32}[args.dtype]
    LF.set_compute_dtype(dtype)

    stage2 = step == 'stage2_step'
    save_path = args.save_path or os.path.join(args.dataName0 + ('_stage2' if stage2 else '_stage1'), datetime.datetime.now().strftime('%m-%d-%H_%M'),
                                               '{},e{}es{},b{},lr{}'.format(args.m_model, args.epochs, args.epoch_size or '', args.batch_size, args.lr))
    if rank == 0:
        os.makedirs(save_path, exist_ok=True)
        with open(os.path.join(save_path, 'settings.txt'), 'w') as f:
            f.write(''.join('%15s: %s\n' % (k, v) for k, v in vars(args).items()))

    network_data = torch.load(args.pretrained, map_location='cpu') if args.pretrained else None
    if stage2 and not args.synthetic and not (args.pretrained and args.fix_model) and not getattr(args, 'allow_seeded_teacher', False):
        # the reference torch.loads both (Train_Stage2_K.py:66-71,190-198): a real-data Stage-2 run must not silently distil from a
        # random, untrained teacher or fine-tune random weights
        raise SystemExit('Stage 2 on real data needs --pretrained <stage-1 checkpoint> and --fix_model <stage-1 checkpoint> '
                         '(seeded stand-ins are used with --synthetic only; --allow-seeded-teacher overrides, for tests)')
    if stage2 and network_data is None:  # --synthetic: seeded stand-in for the Stage-1 model
        network_data = {'state_dict': synthetic.seeded_state_dict(args.m_model[-1], args.no_levels)}
    m_model = models.__dict__[args.m_model](network_data, no_levels=args.no_levels, compute_dtype=dtype).to(dev)
    if isinstance(network_data, dict) and 'loss_scaler' in network_data and train.loss_scaler(m_model) is not None:
        train.loss_scaler(m_model).load_state_dict(network_data['loss_scaler'])  # resumed f16 run: continue at the scale it had reached
    fix_model = None
    if stage2:  # frozen Stage-1 teacher (Train_Stage2_K.py:190-198)
        fix_data = torch.load(args.fix_model, map_location='cpu') if args.fix_model else {'state_dict': synthetic.seeded_state_dict(args.m_model[-1], args.no_levels)}
        fix_model = models.__dict__[args.m_model](fix_data, no_levels=args.no_levels, compute_dtype=dtype).to(dev).eval()
        for q in fix_model.parameters():
            q.requires_grad_(False)
    if rank == 0:
        print("=> Number of parameters m-model '{}'".format(utils.get_n_params(m_model)))
    train.sync_parameters(m_model)  # N > 1: rank 0's weights to every rank, once (then one gradient all-reduce per step)
    opt = train.FlatAdam(m_model, lr=args.lr, betas=(args.momentum, args.beta))

    def lr_at(epoch):  # MultiStepLR(milestones, gamma=0.5), fast-forwarded like Train_Stage1_K.py:181-184
        return args.lr * (0.5 ** sum(1 for m in args.milestones if epoch >= m))

    train_loader = val_loader = None
    if not args.synthetic:
        if not args.data:
            raise SystemExit('give the dataset root with -d/--data (or run with --synthetic)')
        from fal_net_amd import data_transforms as DT
        from fal_net_amd import datasets as DS
        root = os.path.join(args.data, args.dataName0)
        pairs = DS.read_pair_list(args.train_list, root)
        if not pairs:
            raise SystemExit('no training pair of {} exists under {}'.format(args.train_list, root))
        train_loader = DS.make_loader(DS.StereoPairDataset(root, pairs, max_pix=args.max_disp, fix=True), args.batch_size, args.workers,
                                      shuffle=True, rank=rank, world=world)
        vroot = os.path.join(args.data, args.vdataName)
        vtriples = DS.kitti2015_pairs(vroot) if os.path.isdir(vroot) else []
        if vtriples and rank == 0:  # only rank 0 validates: the other ranks start no validation workers
            val_loader = DS.make_loader(DS.StereoValDataset(vroot, vtriples), args.tbatch_size, args.workers, shuffle=False, drop_last=False)
        real_augment = DT.StereoAugment(args.crop_height, args.crop_width)
        if rank == 0:
            print('=> {} training pairs, {} validation pairs'.format(len(pairs), len(vtriples)))
    vtriples_any = train_loader is not None and bool(vtriples)  # the same on every rank (same files): gates the post-validation barrier
    steps_per_epoch = args.epoch_size or (len(train_loader) if train_loader is not None else 100)
    best = -1
    # synthetic mode: a small pool of seeded batches resident in HBM, cycled (generating 25 MB of noise on the CPU every
    # step would make the script loader-bound; a real loader prefetches asynchronously)
    pool = []
    for k in range(4):
        l_, r_, _, mx_ = synthetic.synthetic_pair(args.batch_size, args.crop_height, args.crop_width, seed=1234 + rank + 977 * k, max_disp=args.max_disp)
        pool.append((l_.to(dev), r_.to(dev), mx_.to(dev)))
    raw, augment = None, None
    if args.gpu_augment:
        from fal_net_amd import data_transforms as DT
        import torch as _t
        g = _t.Generator().manual_seed(4321 + rank)
        raw = [[_t.randint(0, 256, (375, 1242, 3), generator=g, dtype=_t.uint8).to(dev) for _ in range(2)] for _ in range(2 * args.batch_size)]
        augment = DT.StereoAugment(args.crop_height, args.crop_width)  # down=0.75, up=1.5, gamma / brightness ranges of :117-122
        mx_aug = _t.full((args.batch_size, 1, 1), float(args.max_disp), device=dev)

    def next_batch(i):
        if augment is None:
            return pool[i % len(pool)]
        import torch as _t
        views = [augment(raw[(i * args.batch_size + b) % len(raw)]) for b in range(args.batch_size)]
        return _t.stack([v[0] for v in views]), _t.stack([v[1] for v in views]), mx_aug
    def real_batches():
        """Decoded uint8 pairs -> GPU -> augmented (B, 3, ch, cw) tensors; the upload of the next list overlaps the current step
        (pinned memory, non_blocking)."""
        import torch as _t
        for batch in train_loader:
            views = [real_augment([l.to(dev, non_blocking=True), r.to(dev, non_blocking=True)]) for l, r, _ in batch]
            mxs = _t.tensor([abs(x) for _, _, x in batch], device=dev).view(-1, 1, 1)
            yield _t.stack([v[0] for v in views]), _t.stack([v[1] for v in views]), mxs

    for epoch in range(args.start_epoch, args.epochs):
        opt.param_groups[0]['lr'] = lr_at(epoch)
        m_model.train()
        losses, rec_losses = utils.AverageMeter(), utils.AverageMeter()
        end = time.time()
        if train_loader is not None and getattr(train_loader, 'sampler', None) is not None and hasattr(train_loader.sampler, 'set_epoch'):
            train_loader.sampler.set_epoch(epoch)
        stream = real_batches() if train_loader is not None else None
        for i in range(steps_per_epoch):
            if stream is not None:
                try:
                    left, right, mx = next(stream)
                except StopIteration:
                    break
            else:
                left, right, mx = next_batch(i)
            if stage2:
                out = train.stage2_step(m_model, fix_model, opt, left, right, mx, a_p=args.a_p, a_sm=args.a_sm, a_mr=args.a_mr,
                                        min_disp_arg=args.min_disp, max_disp_arg=args.max_disp)
            else:
                out = getattr(train, step)(m_model, opt, left, right, mx, a_p=args.a_p, a_sm=args.a_sm,
                                           min_disp_arg=args.min_disp, max_disp_arg=args.max_disp)
            if i % args.print_freq == 0:
                losses.update(float(out['loss']), args.batch_size)
                rec_losses.update(float(out['rec']), args.batch_size)
                if not (losses.val == losses.val and abs(losses.val) != float('inf')):  # fail loudly: a NaN loss never recovers
                    raise FloatingPointError('non-finite loss {} at epoch {} iteration {}'.format(losses.val, epoch, i))
                f16_scale = out['scaler'].check() if out.get('scaler') is not None else None  # raises when f16 gradients overflow at scale 1
                if rank == 0:
                    rec = {'epoch': epoch, 'iter': i, 'of': steps_per_epoch, 'loss': losses.val, 'rec_loss': rec_losses.val,
                           'pairs_per_s': world * args.batch_size * (i + 1) / (time.time() - end)}
                    if f16_scale is not None:
                        rec['f16_loss_scale'], rec['f16_skipped_steps'] = f16_scale
                    if stage2:
                        rec['mirror'] = float(out['mirror'])
                    print(json.dumps(rec), flush=True)
        is_best = False
        if val_loader is not None and rank == 0:  # :190-207: validate, keep the best RMSE
            res = train.validate(m_model, val_loader, max_disp=args.max_disp, min_disp=args.min_disp, rel_baset=args.rel_baset,
                                 sparse=args.sparse, print_freq=args.print_freq)
            print(json.dumps({'epoch': epoch, 'val_rmse': res['rmse'], 'val_epe': res['epe'], 'kitti': res['kitti']}), flush=True)
            if best < 0:
                best = res['rmse']
            is_best = res['rmse'] <= best
            best = min(res['rmse'], best)
        if world > 1 and vtriples_any:
            # rank 0 validated (200 full-size frames, the first of each size builds a plan): hold the other ranks HERE rather than
            # inside the next epoch's first gradient all-reduce, whose RCCL timeout that time would count against
            dist.barrier()
        if rank == 0:
            ckpt = {'epoch': epoch + 1, 'm_model': args.m_model, 'state_dict': m_model.state_dict(), 'best_rmse': best}
            sc = train.loss_scaler(m_model)
            if sc is not None:
                ckpt['loss_scaler'] = sc.state_dict()  # (an extra key: the reference's loader reads only the four above)
            utils.save_checkpoint(ckpt, is_best, save_path)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    args = parser.parse_args()
    main()
