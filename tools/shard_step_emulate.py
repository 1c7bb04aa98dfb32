#!/usr/bin/env python3
"""Whole level-sharded training step of ONE rank of a pretend world, on one GPU, with the collectives replaced by
local stand-ins of the same shapes (all-gather = tile the local rays, all-to-all = leave the remote blocks zero,
all-reduce = nothing).  Everything a rank computes in a real run is executed -- encode of its levels for world x the
points, MLPs from features, backward, scatter of its levels for world x the points, Adam on its slices, the Python
and launch overhead of the exchange code -- only the time on the wire is missing.

    python tools/shard_step_emulate.py --world 8 [--rank 0] [--steps 20]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

import bench
from nerf_amd import level_parallel, parallel


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--world', type=int, default=8)
    ap.add_argument('--rank', type=int, default=-1, help='-1: every rank in turn')
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--rays', type=int, default=4096)
    args = ap.parse_args()
    W = args.world
    state = {'rank': 0}
    parallel.world_size = lambda: W
    parallel.rank = lambda: state['rank']
    parallel.all_gather_rows = lambda t: t.repeat(*([W] + [1] * (t.dim() - 1)))

    def fake_a2a(recv_blocks, send_blocks, async_op=False):
        for i, b in enumerate(recv_blocks):
            if i != state['rank']:
                b.zero_()
        return None
    parallel.all_to_all_blocks = fake_a2a
    parallel.all_reduce_grads = lambda params, async_op=False, skip=(): []
    out = []
    for r in (range(W) if args.rank < 0 else [args.rank]):
        state['rank'] = r
        model, renderer, loader, opt = bench.build_gpu(torch.device('cuda', 0), r, args.rays, shard_levels=True)
        assert model.level_sharding.world == W and model.level_sharding.rank == r
        for it in range(3):
            bench.train_step(model, renderer, loader, opt, args.rays, it, W)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for it in range(args.steps):
            bench.train_step(model, renderer, loader, opt, args.rays, it, W)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        rec = {'world': W, 'rank': r, 'levels': model.level_sharding.mine, 'ms_per_step_without_wire': round(ms, 3)}
        print(json.dumps(rec), flush=True)
        out.append(rec)
        del model, renderer, loader, opt
        torch.cuda.empty_cache()
    ms = [o['ms_per_step_without_wire'] for o in out]
    print(json.dumps({'world': W, 'max_ms': max(ms), 'mean_ms': sum(ms) / len(ms)}))


if __name__ == '__main__':
    main()
