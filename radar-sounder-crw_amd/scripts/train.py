#!/usr/bin/env python3
"""CRW training entrypoint -- same flags and defaults as the reference's scripts/train.py:17-37
(--tune and the Ray-Tune branch are out of scope: third-party orchestration, SURVEY.md section 2
row 10).  Additions: --data_path / --synthetic H W (the reference's dataset paths are private),
--steps (stop early), --save (skip writing the checkpoint when empty), --shared_encode (a batch is
--batch_size CONSECUTIVE overlapping items whose patch-columns are encoded once, CRW.forward_columns).

Single GPU:   python radar-sounder-crw_amd/scripts/train.py --model 0 --synthetic 512 4096
Multi GPU:    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \
                  radar-sounder-crw_amd/scripts/train.py --model 0 --synthetic 512 32768
One process per GPU; items (independent sequences) are sharded over ranks and the only exchange is
one RCCL all-reduce of the flat encoder gradient per step (the reference instead wraps the encoder
in nn.DataParallel and runs the walk on GPU 0, scripts/train.py:45-47).
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch
from torch.optim import Adam
from torch.utils.data import DataLoader, Subset

from utils import create_model, create_dataset
from model import CRW
import dist as crw_dist

torch.manual_seed(11)


def get_args_parser():
    p = argparse.ArgumentParser('CRW Train', add_help=True)
    p.add_argument('--tune', default=False, type=bool, help='Ray Tune search (not supported here)')
    p.add_argument('--model', default=1, type=int, help='0=CNN,1=Resnet18')
    p.add_argument('--dataset', default=3, type=int, help='0=MCORDS1,1=Miguel,3=SHARAD')
    p.add_argument('--patch_size', default=(16, 16), nargs=2, type=int)
    p.add_argument('--seq_length', default=20, type=int)
    p.add_argument('--overlap', default=(8, 0), nargs='+', type=int)
    p.add_argument('--batch_size', default=8, type=int, help='global batch (items per step over all ranks)')
    p.add_argument('--epochs', default=2, type=int)
    p.add_argument('--lr', default=1E-3, type=float)
    p.add_argument('--tau', default=0.01, type=float)
    p.add_argument('--pos_embed', default=False, type=bool)
    p.add_argument('--dataset_full', default=True)
    p.add_argument('--output_folder', default='./resources/')
    p.add_argument('--output_name', default='sharad16_3')
    p.add_argument('--data_path', default=None, help='H x W radargram .pt file')
    p.add_argument('--synthetic', default=None, nargs=2, type=int, metavar=('H', 'W'))
    p.add_argument('--steps', default=0, type=int, help='stop after this many steps (0 = full epochs)')
    p.add_argument('--shared_encode', action='store_true',
                   help='batches of consecutive overlapping items share one encoder pass (needs --dataset_full)')
    p.add_argument('--host_data', action='store_true',
                   help='keep the radargram in host memory and feed items through a DataLoader like the reference '
                        '(default: the radargram lives on the GPU and items are cut there, no per-step H2D copy)')
    p.add_argument('--save', default='', help='checkpoint path for encoder.state_dict() (default: '
                                              '<output_folder>/models/<output_name>.pt)')
    return p


def main(args, on_step=None):
    """on_step(step_index, loss): optional observer called after every optimizer step with the step's (detached, on-device)
    loss of this rank."""
    rank, world, local = crw_dist.init_from_env()
    if rank == 0:
        print(args)
    if args.tune:
        raise SystemExit('--tune (Ray Tune) is outside the scope of the MI355X build')
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)

    encoder = create_model(args.model, args.pos_embed)  # same seed on every rank -> identical replicas
    model = CRW(encoder, args.tau, args.pos_embed).to(device)
    dataset = create_dataset(id=args.dataset, length=args.seq_length, dim=tuple(args.patch_size),
                             full=args.dataset_full, overlap=tuple(args.overlap), data_path=args.data_path,
                             synthetic=args.synthetic)
    if args.batch_size % world:
        raise SystemExit(f'--batch_size {args.batch_size} must be a multiple of the number of ranks ({world})')
    per_rank = args.batch_size // world
    # Like a DataLoader with drop_last=True the tail of an epoch that does not fill a global batch is dropped (the
    # reference's loader keeps a final partial batch, scripts/train.py:53): equal shards per rank are what make the mean of
    # the rank gradients the global gradient.  A dataset shorter than one batch would silently train nothing: refuse it.
    if len(dataset) < args.batch_size:
        raise SystemExit(f'the dataset holds {len(dataset)} items, fewer than one global batch (--batch_size {args.batch_size}): '
                         'no step would run; lower --batch_size or --seq_length, or use a longer radargram')
    if args.shared_encode and not hasattr(dataset, 'columns'):
        raise SystemExit('--shared_encode needs the overlapping dataset (--dataset_full True)')

    base = dataset.dataset if isinstance(dataset, Subset) else dataset
    if not args.host_data:
        # the item cut is a strided view of the radargram (src/dataset.py:19-39): with the radargram resident in
        # HBM a batch is one gather kernel instead of a CPU unfold + 16 MB host-to-device copy per step
        base.T = base.T.to(device)
    bucket = crw_dist.FlatGradBucket(model.parameters(), lazy=True)
    if device.type == "cuda":  # torch.optim.Adam's update (the reference's optimizer) as one launch over flat buffers
        import optim as crw_optim
        optimizer = crw_optim.FlatAdam(bucket, lr=args.lr)
    else:
        optimizer = Adam(model.parameters(), lr=args.lr)
    model.train(True)
    loss_tot, nsteps = [], 0
    for epoch in range(args.epochs):
        t0 = time.time()
        g = torch.Generator().manual_seed(11 + epoch)  # identical permutation on every rank
        order = torch.randperm(len(dataset), generator=g).tolist()
        usable = len(order) // args.batch_size * args.batch_size
        mine = [order[i] for i in range(usable) if (i % args.batch_size) // per_rank == rank]
        if args.shared_encode:
            # batch b = items [b*batch_size, (b+1)*batch_size): rank r walks per_rank consecutive items whose
            # per_rank + seq_length - 1 patch-columns are encoded once; batch order is shuffled per epoch
            nb = len(dataset) // args.batch_size
            loader = (dataset.columns(b * args.batch_size + rank * per_rank, per_rank + args.seq_length - 1)[None]
                      for b in torch.randperm(nb, generator=g).tolist())
        elif args.host_data:
            loader = DataLoader(Subset(dataset, mine), batch_size=per_rank, shuffle=False)
        else:
            loader = (torch.stack([dataset[i] for i in mine[b:b + per_rank]])
                      for b in range(0, len(mine) - per_rank + 1, per_rank))
        loss_epoch = []
        for seq in loader:
            seq = seq.to(device, non_blocking=True)
            bucket.zero()
            loss, _ = model.forward_columns(seq, args.seq_length) if args.shared_encode else model(seq)
            loss.backward()
            bucket.all_reduce_mean()
            optimizer.step()
            loss_epoch.append(loss.detach())
            nsteps += 1
            if on_step is not None:
                on_step(nsteps, loss_epoch[-1])
            if args.steps and nsteps >= args.steps:
                break
        mean = torch.stack(loss_epoch).mean() if loss_epoch else torch.zeros((), device=device)
        if world > 1:
            torch.distributed.all_reduce(mean, op=torch.distributed.ReduceOp.SUM)
            mean /= world
        loss_tot.append(mean.item())
        cols = len(loss_epoch) * args.batch_size * (args.seq_length * (args.patch_size[1] - args.overlap[1])
                                                   + args.overlap[1])
        if rank == 0:
            dt = time.time() - t0
            print('Epoch:', epoch, 'Loss:', loss_tot[-1], 'Time:', dt, 'columns/s:', cols / max(dt, 1e-9))
        if args.steps and nsteps >= args.steps:
            break

    if rank == 0:
        path = args.save or os.path.join(args.output_folder, 'models', args.output_name + '.pt')
        os.makedirs(os.path.dirname(path), exist_ok=True)
        torch.save(encoder.state_dict(), path)
        print('Finished training.')
    if world > 1:
        torch.distributed.destroy_process_group()
    return loss_tot


if __name__ == '__main__':
    a = get_args_parser().parse_args()
    a.overlap = tuple(a.overlap)
    main(a)
