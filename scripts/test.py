#!/usr/bin/env python
# coding: utf-8
"""Evaluation entry point with the command line and the output of the reference's scripts/test.py
(test.py:20-123): `test.py data_path model_path run_dir guides... [--time_limit 10.] [--perturbation_moves 20]
[--use_gpu]`, per-instance semantics unchanged (the budget starts before the forward pass, nearest-neighbour start
on 'regret_pred' whenever that guide is used and on 'weight' otherwise (test.py:70-88), guided_local_search, gap vs the
optimum stored in the instances) and the same DataFrame pickle (columns instance, time, opt_cost, cost, best_cost, gap,
dt) in `run_dir/<timestamp>_<uuid>.pkl`.

Search progress (test.py:97-117).  The reference appends one row per accepted move; a 10 s TSP100 search on the GPU
accepts ~2e6 moves per instance, i.e. ~20 GB of rows for a 1024-instance batch.  The default record here is therefore
the bounded one the device keeps at any run length: one row whenever the returned best improves plus a terminal row
(end of the search, returned best cost) -- exactly the rows that determine the `best_cost` (cummin), `gap` and `dt`
columns; `best_cost` of an instance's last row always equals the returned cost.  `--full_trace CAP` additionally keeps
the first CAP per-move rows of every instance (the reference's record verbatim while moves <= CAP); if an instance
accepts more moves than that, its rows continue with the improvement record and a warning is printed -- never a
silently truncated DataFrame.

What differs is the execution: instead of one instance at a time (test.py:59) whole batches are searched on the GPU,
every instance of a batch getting the full --time_limit concurrently (gnngls_amd.pipeline.solve_batch).  `--use_gpu`
is accepted for compatibility; there is no CPU path.  Under torchrun (one process per GPU) the instance list is
split into contiguous blocks (gnngls_amd.parallel.shard_range), every rank searches its block, and rank 0 gathers
the records once and writes the single DataFrame.
"""
import argparse
import datetime
import json
import os
import pathlib
import sys
import time
import uuid

import numpy as np
import pandas as pd
import torch
import torch.distributed as dist
import tqdm.auto as tqdm

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))

import gnngls_amd  # noqa: E402
from gnngls_amd import datasets, models, ops, parallel, pipeline  # noqa: E402
from gnngls_amd.algorithms import _attr_matrix  # noqa: E402

IMP_CAP = 1024


def parse_args():
    parser = argparse.ArgumentParser(description='Test model')
    parser.add_argument('data_path', type=pathlib.Path)
    parser.add_argument('model_path', type=pathlib.Path)
    parser.add_argument('run_dir', type=pathlib.Path)
    parser.add_argument('guides', type=str, nargs='+')
    parser.add_argument('--time_limit', type=float, default=10.)
    parser.add_argument('--perturbation_moves', type=int, default=20)
    parser.add_argument('--use_gpu', action='store_true')
    parser.add_argument('--batch_size', type=int, default=0, help='instances searched concurrently (0 = device capacity)')
    parser.add_argument('--per_batch_budget', type=int, default=0, metavar='R',
                        help='search R x batch_size instances within ONE --time_limit (each gets time_limit / R) '
                             'instead of giving every instance the full budget')
    parser.add_argument('--full_trace', type=int, default=0, metavar='CAP',
                        help='also record the first CAP accepted moves of every instance (the reference records all)')
    return parser.parse_args()


def init_ranks():
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')) % max(torch.cuda.device_count(), 1))
    if world > 1 or os.environ.get('GNNGLS_DIST_SINGLE', '0') == '1':       # the latter: one rank through RCCL (GPU tests)
        dist.init_process_group(os.environ.get('GNNGLS_DIST_BACKEND', 'nccl'))
    return world, rank


def load_model(args, params, test_set):
    """test.py:37-54: architecture from params.json, weights from checkpoint['model_state_dict']."""
    if 'regret_pred' not in args.guides:
        return None, None
    device = torch.device('cuda')
    print('device =', device)
    if datasets.is_lfs_pointer(args.model_path):
        raise FileNotFoundError(f'{args.model_path} is a git-LFS pointer stub')
    _, feat_dim = test_set[0].ndata['features'].shape                        # test.py:41 (after efeat_drop_idx)
    model = models.EdgePropertyPredictionModel(feat_dim, params['embed_dim'], 1, params['n_layers'], n_heads=params['n_heads'])
    state = torch.load(args.model_path, map_location=device)['model_state_dict']
    model.load_state_dict(state)
    return model.to(device).eval(), pipeline.Scalers.from_sklearn(test_set.scalers)


def progress_rows(res, k, full_trace):
    """(dt since the instance's search kernel started, cost) rows of instance k, see the module docstring."""
    rows, t_cut = [], -1.0
    if full_trace > 0:
        moves = int(res.moves[k])
        kept = min(moves, full_trace)
        rows = list(zip(res.trace_time[k, :kept].tolist(), res.trace_cost[k, :kept].tolist()))
        t_cut = rows[-1][0] if rows else -1.0
    m = min(int(res.imp_len[k]), res.imp_cost.shape[1])
    imp = list(zip(res.imp_time[k, :m].tolist(), res.imp_cost[k, :m].tolist()))
    if full_trace > 0 and moves <= full_trace:
        # complete per-move record: only the terminal entry (returned best, end of the search) is added, so that the
        # record reaches the returned cost also when no move was accepted or the start tour was never improved
        return rows + imp[-1:], False
    rows += [(dt, c) for dt, c in imp[:-1] if dt > t_cut] + imp[-1:]
    return rows, full_trace > 0


def default_feature_set(test_set, G):
    """True if the instances carry the reference's default edge features -- the edge weight alone (datasets.py:14-20) --
    and nothing is dropped: the scaled features are then packed from the distance matrix on the device.  Any other
    feature set goes through TSPDataset.get_scaled_features on the host, exactly as test.py:72 does."""
    if len(test_set.feat_drop_idx):
        return False
    return all(np.shape(G.edges[e]['features']) == (1,) and G.edges[e]['features'][0] == np.float32(G.edges[e]['weight'])
               for e in G.edges)


def solve_block(names, test_set, model, scalers, args, chunk, budget='per_instance'):
    """One batch of instances -> (search-progress records, gaps), the body of the loop at test.py:59-109."""
    graphs = [datasets.read_gpickle(test_set.root_dir / name) for name in names]
    optima = [gnngls_amd.optimal_cost(G, weight='weight') for G in graphs]
    D = torch.from_numpy(np.stack([_attr_matrix(G, 'weight') for G in graphs])).cuda()
    features = None
    if model is not None and not default_feature_set(test_set, graphs[0]):
        features = torch.stack([test_set.get_scaled_features(G).ndata['features'] for G in graphs])      # test.py:72-74
    res = pipeline.solve_batch(D, model, scalers, guides=args.guides, time_limit=args.time_limit,
                               perturbation_moves=args.perturbation_moves, trace_cap=args.full_trace,
                               want_trace_time=args.full_trace > 0, chunk=chunk, budget=budget, imp_cap=IMP_CAP,
                               features=features)
    res.imp_cost, res.imp_time, res.imp_len = res.imp_cost.cpu(), res.imp_time.cpu(), res.imp_len.cpu()
    res.moves = res.moves.cpu()
    if args.full_trace > 0:
        res.trace_cost, res.trace_time = res.trace_cost.cpu(), res.trace_time.cpu()
    best, started, launched = res.best_cost.cpu().numpy(), res.start_time.numpy(), res.launch_time.numpy()
    records, gaps, cut = [], [], 0
    for k, (name, opt) in enumerate(zip(names, optima)):
        # the budget of an instance starts before its forward pass (test.py:64); device times count from the launch
        # of its search kernel
        records.append({'instance': name, 'time': float(started[k]), 'opt_cost': opt})
        rows, truncated = progress_rows(res, k, args.full_trace)
        cut += truncated
        records += [{'instance': name, 'opt_cost': opt, 'time': float(launched[k]) + dt, 'cost': c} for dt, c in rows]
        # cummin over the rows = the `best_cost` column: it must end on the returned cost (a complete per-move record ends
        # on the last iteration's cost, which may be above the best; its minimum is the best, bit for bit)
        assert rows and min(c for _, c in rows) == best[k], 'the search-progress record must reach the returned cost'
        gaps.append((best[k] / opt - 1) * 100)                                # test.py:104
    if cut:
        print(f'warning: {cut} instance(s) accepted more than --full_trace {args.full_trace} moves; their rows '
              f'continue with new-best events only', file=sys.stderr)
    return records, gaps


def records_to_array(records, n_instances, width):
    """The records of this rank's instances (in order: a start row, then progress rows) as ONE fixed-width fp64 array
    [n_instances, 3 + 2 width]: opt_cost, start time, number of progress rows, their times, their costs (NaN padded).
    width = --full_trace + IMP_CAP: a function of the command line alone, so every rank knows every rank's shape."""
    out = np.full((n_instances, 3 + 2 * width), np.nan)
    k = -1
    for r in records:
        if 'cost' not in r:                                                   # start row of the next instance (test.py:65-68)
            k += 1
            out[k, 0], out[k, 1], out[k, 2] = r['opt_cost'], r['time'], 0
        else:
            m = int(out[k, 2])
            out[k, 3 + m], out[k, 3 + width + m] = r['time'], r['cost']
            out[k, 2] = m + 1
    assert k == n_instances - 1
    return out


def array_to_records(arr, names, width):
    records = []
    for row, name in zip(arr, names):
        opt = float(row[0])
        records.append({'instance': name, 'time': float(row[1]), 'opt_cost': opt})
        m = int(row[2])
        records += [{'instance': name, 'opt_cost': opt, 'time': float(t), 'cost': float(c)}
                    for t, c in zip(row[3:3 + m], row[3 + width:3 + width + m])]
    return records


def gather_records(records, world, rank, names_all, n_local, width):
    """The one exchange of the run: a tensor `gather` of fixed-width record arrays (gnngls_amd.parallel.gather_results: shard
    sizes are a function of (total, world), no size exchange, no pickling through the collective)."""
    if not dist.is_initialized():
        return records
    local = torch.from_numpy(records_to_array(records, n_local, width))
    on_gpu = dist.get_backend() == 'nccl'                                     # RCCL moves device memory
    g = parallel.gather_results(local.cuda() if on_gpu else local, parallel.shard_sizes(len(names_all), world))
    dist.barrier()
    dist.destroy_process_group()
    if rank != 0:
        return None
    return array_to_records(g.cpu().numpy(), names_all, width)


def write_progress(records, run_dir):
    """test.py:113-123."""
    df = pd.DataFrame.from_records(records)
    by_instance = df.groupby('instance')
    df['best_cost'] = by_instance['cost'].cummin()
    df['gap'] = (df['best_cost'] / df['opt_cost'] - 1) * 100
    df['dt'] = df['time'] - by_instance['time'].transform('min')
    stamp = datetime.datetime.now().strftime('%b%d_%H-%M-%S')
    if not run_dir.exists():
        run_dir.mkdir()
    df.to_pickle(run_dir / f'{stamp}_{uuid.uuid4().hex}.pkl')


def main():
    args = parse_args()
    world, rank = init_ranks()
    params = json.load(open(args.model_path.parent / 'params.json'))
    test_set = datasets.TSPDataset(args.data_path, feat_drop_idx=params.get('efeat_drop_idx', []))
    model, scalers = load_model(args, params, test_set)

    chunk = args.batch_size or max(ops.gls_resident_capacity(test_set.G.n), 1)
    lo, hi = parallel.shard_range(len(test_set.instances), world, rank)       # this rank's block of instances
    mine = test_set.instances[lo:hi]
    records, gaps = [], []
    block = chunk * max(args.per_batch_budget, 1)
    budget = 'per_batch' if args.per_batch_budget > 1 else 'per_instance'
    with tqdm.tqdm(total=len(mine), disable=rank != 0) as pbar:
        for start in range(0, len(mine), block):
            names = mine[start:start + block]
            rec, g = solve_block(names, test_set, model, scalers, args, chunk, budget)
            records += rec
            gaps += g
            pbar.set_postfix({'Avg Gap': '{:.4f}'.format(np.mean(gaps))})
            pbar.update(len(names))

    records = gather_records(records, world, rank, test_set.instances, len(mine), args.full_trace + IMP_CAP + 1)
    if records is not None:
        write_progress(records, args.run_dir)


if __name__ == '__main__':
    main()
