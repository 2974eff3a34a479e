#!/usr/bin/env python
# coding: utf-8
"""Training entry point with the command line, the outputs and the epoch logic of the reference's scripts/train.py
(train.py:69-166): `train.py data_dir tb_dir [--embed_dim 128] [--n_layers 3] [--n_heads 8] [--lr_init 1e-3]
[--lr_decay 0.99] [--min_delta 1e-4] [--patience 20] [--batch_size 32] [--n_epochs 100] [--checkpoint_freq N]
[--target regret|in_solution] [--use_gpu]`; Adam + ExponentialLR, MSELoss / BCEWithLogitsLoss, early stopping on the
validation loss, checkpoints `checkpoint_{epoch}.pt`, `checkpoint_best_val.pt`, `checkpoint_final.pt` (keys epoch,
model_state_dict, optimizer_state_dict, loss, val_loss) and `params.json` in `tb_dir/<timestamp>_<uuid>/`.

What differs is the execution: `model(batch, x)` and `loss.backward()` run on the MI355X training kernels
(gnngls_amd.models, include/gnngls_hip.h N4); batches are `gnngls_amd.models.batch` unions of line graphs instead of
dgl.batch.  `--use_gpu` is accepted for compatibility; there is no CPU path.  Scalars go to TensorBoard when it is
installed, and always to `scalars.jsonl` in the run directory.

Kept on purpose: the reference evaluates its "validation" loss on the TRAINING loader (train.py:137 passes train_loader
to test()), so early stopping and checkpoint_best_val follow the training-set loss in eval mode; `--val_on_val_set`
switches to the validation set.
"""
import argparse
import datetime
import json
import os
import pathlib
import sys
import uuid

import torch
import tqdm.auto as tqdm
from torch.utils.data import DataLoader

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))

from gnngls_amd import datasets, models  # noqa: E402


class ScalarLog:
    """SummaryWriter when tensorboard is importable, plus a scalars.jsonl that is always written."""

    def __init__(self, log_dir):
        log_dir.mkdir(parents=True, exist_ok=True)
        self.file = open(log_dir / 'scalars.jsonl', 'w')
        try:
            from torch.utils.tensorboard import SummaryWriter
            self.tb = SummaryWriter(log_dir)
        except Exception:                       # tensorboard is an optional dependency here
            self.tb = None

    def add_scalar(self, tag, value, step):
        self.file.write(json.dumps({'tag': tag, 'value': float(value), 'step': int(step)}) + '\n')
        self.file.flush()
        if self.tb is not None:
            self.tb.add_scalar(tag, value, step)

    def close(self):
        self.file.close()
        if self.tb is not None:
            self.tb.close()


# the reference's command line (train.py:70-86): flag, type, default, help
OPTIONS = (
    ('embed_dim', int, 128, 'Maximum hidden feature dimension'),
    ('n_layers', int, 3, 'Number of message passing steps'),
    ('n_heads', int, 8, 'Number of attention heads for GAT'),
    ('lr_init', float, 1e-3, 'Initial learning rate'),
    ('lr_decay', float, 0.99, 'Learning rate decay'),
    ('min_delta', float, 1e-4, 'Early stopping min delta'),
    ('patience', int, 20, 'Early stopping patience'),
    ('batch_size', int, 32, 'Batch size'),
    ('n_epochs', int, 100, 'Number of epochs'),
    ('checkpoint_freq', int, None, 'Checkpoint frequency'),
)


def parse_args():
    ap = argparse.ArgumentParser(description='Train model')
    ap.add_argument('data_dir', type=pathlib.Path, help='Where to load dataset')
    ap.add_argument('tb_dir', type=pathlib.Path, help='Where to log Tensorboard data')
    for name, kind, default, text in OPTIONS:
        ap.add_argument('--' + name, type=kind, default=default, help=text)
    ap.add_argument('--target', type=str, default='regret', choices=['regret', 'in_solution'])
    ap.add_argument('--use_gpu', action='store_true')
    ap.add_argument('--num_workers', type=int, default=os.cpu_count(), help='DataLoader workers (train.py:118-121)')
    ap.add_argument('--val_on_val_set', action='store_true', help='evaluate the validation loss on val.txt')
    return ap.parse_args()


def run_epoch(model, loader, target, criterion, device, optimizer=None):
    """train.py:20-38 (optimizer given) / train.py:41-58 (evaluation): mean of the per-batch losses."""
    training = optimizer is not None
    model.train(training)
    total, batches = 0.0, 0
    with torch.set_grad_enabled(training):
        for batch in loader:
            batch = batch.to(device)
            x, y = batch.ndata['features'], batch.ndata[target]
            if training:
                optimizer.zero_grad()
            y_pred = model(batch, x)
            loss = criterion(y_pred, y.type_as(y_pred))
            if training:
                loss.backward()
                optimizer.step()
            total += loss.detach().item()
            batches += 1
    return total / batches


def save(model, optimizer, epoch, train_loss, val_loss, save_path):
    """train.py:61-68"""
    torch.save({'epoch': epoch, 'model_state_dict': model.state_dict(), 'optimizer_state_dict': optimizer.state_dict(),
                'loss': train_loss, 'val_loss': val_loss}, save_path)


def make_criterion(target, train_set, device):
    if target == 'regret':
        return torch.nn.MSELoss()                                             # train.py:106-107
    y = train_set[0].ndata['in_solution']                                     # only works for a homogenous dataset
    pos_weight = len(y) / y.sum() - 1                                         # train.py:110-112
    return torch.nn.BCEWithLogitsLoss(pos_weight=pos_weight.to(device))


def main():
    args = parse_args()
    train_set = datasets.TSPDataset(args.data_dir / 'train.txt')
    val_set = datasets.TSPDataset(args.data_dir / 'val.txt')
    if not torch.cuda.is_available():
        raise RuntimeError('gnngls_amd has no CPU path: a HIP device is required')
    device = torch.device('cuda')
    print('device =', device)

    _, feat_dim = train_set[0].ndata['features'].shape
    model = models.EdgePropertyPredictionModel(feat_dim, args.embed_dim, 1, args.n_layers, n_heads=args.n_heads).to(device)
    optimizer = torch.optim.Adam(model.parameters(), lr=args.lr_init)
    lr_scheduler = torch.optim.lr_scheduler.ExponentialLR(optimizer, args.lr_decay)
    criterion = make_criterion(args.target, train_set, device)

    loader_args = dict(batch_size=args.batch_size, shuffle=True, collate_fn=models.batch, num_workers=args.num_workers)
    train_loader = DataLoader(train_set, **loader_args)
    val_loader = DataLoader(val_set, **loader_args)
    eval_loader = val_loader if args.val_on_val_set else train_loader          # train.py:137 (see the module docstring)

    stamp = datetime.datetime.now().strftime('%b%d_%H-%M-%S')
    log_dir = args.tb_dir / f'{stamp}_{uuid.uuid4().hex}'
    writer = ScalarLog(log_dir)

    best_score, stale = None, 0                                               # early stopping, train.py:127-129
    epoch = epoch_loss = epoch_val_loss = None
    pbar = tqdm.trange(args.n_epochs)
    for epoch in pbar:
        epoch_loss = run_epoch(model, train_loader, args.target, criterion, device, optimizer)
        writer.add_scalar('Loss/train', epoch_loss, epoch)
        epoch_val_loss = run_epoch(model, eval_loader, args.target, criterion, device)
        writer.add_scalar('Loss/validation', epoch_val_loss, epoch)
        pbar.set_postfix({'Train Loss': '{:.4f}'.format(epoch_loss), 'Validation Loss': '{:.4f}'.format(epoch_val_loss)})

        if args.checkpoint_freq is not None and epoch > 0 and epoch % args.checkpoint_freq == 0:
            save(model, optimizer, epoch, epoch_loss, epoch_val_loss, log_dir / f'checkpoint_{epoch}.pt')
        if best_score is None or epoch_val_loss < best_score - args.min_delta:
            save(model, optimizer, epoch, epoch_loss, epoch_val_loss, log_dir / 'checkpoint_best_val.pt')
            best_score, stale = epoch_val_loss, 0
        else:
            stale += 1
        if stale >= args.patience:
            pbar.close()
            break
        lr_scheduler.step()
    writer.close()

    params = dict(vars(args))
    params['data_dir'], params['tb_dir'] = str(params['data_dir']), str(params['tb_dir'])
    json.dump(params, open(log_dir / 'params.json', 'w'))
    save(model, optimizer, epoch, epoch_loss, epoch_val_loss, log_dir / 'checkpoint_final.pt')


if __name__ == '__main__':
    main()
