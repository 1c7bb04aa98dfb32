#!/usr/bin/env python3
"""Whole level-sharded training step of ONE rank of a pretend world, on one GPU, with the collectives replaced by
local stand-ins of the same shapes (all-gather = tile the local rays, all-to-all = leave the remote blocks zero,
all-reduce = nothing).  Everything a rank computes in a real run is executed -- encode of its levels for world x the
points, MLPs from features, backward, scatter of its levels for world x the points, Adam on its slices, the Python
and launch overhead of the exchange code.

The WIRE is modelled (--link-gbs > 0): every collective becomes a spin kernel on a separate "wire" stream that lasts
latency + bytes-to-the-busiest-peer / link rate and is ordered against the compute stream exactly like the RCCL
call it stands for (it starts when the compute stream reaches the call; work.wait() makes the compute stream wait
for its end), so the overlap the step really gets is what the clock measures.  xGMI on an 8-GPU MI355X node is a
full mesh, one link per pair, 153.6 GB/s per link both directions together = 76.8 GB/s each way; an all-to-all uses
all 7 links of a GPU at once, each carrying one peer's block.  Defaults: 80 % of 76.8 GB/s achieved, 20 us per
collective.  Projected weak-scaling efficiency = single-GPU ms / slowest rank's ms.

    python tools/shard_step_emulate.py --world 8 [--rank 0] [--steps 20] [--link-gbs 61.4] [--single-ms 4.40]
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
    ap.add_argument('--link-gbs', type=float, default=0.8 * 76.8, help='achieved GB/s per link and direction; 0: no wire')
    ap.add_argument('--latency-us', type=float, default=20.0, help='fixed cost of one collective')
    ap.add_argument('--single-ms', type=float, default=0.0, help='single-GPU ms/step for the efficiency line')
    args = ap.parse_args()
    W = args.world
    state = {'rank': 0}
    parallel.world_size = lambda: W
    parallel.rank = lambda: state['rank']
    parallel.all_gather_rows = lambda t: t.repeat(*([W] + [1] * (t.dim() - 1)))

    wire = torch.cuda.Stream()
    # spin-kernel calibration: cycles of torch.cuda._sleep per microsecond
    torch.cuda._sleep(1000)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    torch.cuda._sleep(20000000)
    e1.record()
    torch.cuda.synchronize()
    cyc_per_us = 20000000 / (e0.elapsed_time(e1) * 1e3)
    wire_log = {'us': 0.0, 'calls': 0, 'bytes': 0}

    class Wire(object):
        def __init__(self, ev):
            self.ev = ev

        def wait(self):
            torch.cuda.current_stream().wait_event(self.ev)

    def on_wire(nbytes_busiest_peer, async_op, hops=1):
        """A collective whose busiest link carries `nbytes_busiest_peer` in one direction, in `hops` dependent rounds."""
        if args.link_gbs <= 0:
            return None
        us = hops * args.latency_us + nbytes_busiest_peer / (args.link_gbs * 1e3)
        wire_log['us'] += us
        wire_log['calls'] += 1
        wire_log['bytes'] += nbytes_busiest_peer
        start = torch.cuda.Event()
        start.record()
        with torch.cuda.stream(wire):
            wire.wait_event(start)
            torch.cuda._sleep(int(us * cyc_per_us))
            done = torch.cuda.Event()
            done.record()
        if async_op:
            return Wire(done)
        torch.cuda.current_stream().wait_event(done)
        return None

    def fake_gather(t, async_op=False):
        out = t.repeat(*([W] + [1] * (t.dim() - 1)))
        work = on_wire(t.numel() * t.element_size(), async_op)      # every peer sends me its block over its own link
        return (out, work) if async_op else out
    parallel.all_gather_rows = fake_gather

    def fake_a2a(recv_blocks, send_blocks, async_op=False):
        for i, b in enumerate(recv_blocks):
            if i != state['rank']:
                b.zero_()
        peer = max([b.numel() * b.element_size() for i, b in enumerate(send_blocks) if i != state['rank']] +
                   [b.numel() * b.element_size() for i, b in enumerate(recv_blocks) if i != state['rank']] + [0])
        return on_wire(peer, async_op)
    parallel.all_to_all_blocks = fake_a2a

    def fake_reduce(params, async_op=False, skip=(), flags=()):
        # the MLP / latent gradients + the non-finite flag: ONE small all-reduce (~60 KB: latency-bound, modelled as
        # two latencies whatever the world -- RCCL's low-latency protocols, not a 2 (W-1)-hop ring)
        skip_ids = {id(p) for p in skip}
        n = sum(p.numel() * 4 for p in params if p is not None and p.grad is not None and id(p) not in skip_ids)
        h = on_wire(2.0 * n * (W - 1) / W, async_op, hops=2)
        return [h] if (async_op and h is not None) else []
    parallel.all_reduce_grads = fake_reduce
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
        rec = {'world': W, 'rank': r, 'levels': model.level_sharding.mine,
               ('ms_per_step_with_modelled_wire' if args.link_gbs > 0 else 'ms_per_step_without_wire'): round(ms, 3),
               'groups': len(model.level_sharding.plan(dict(outer_bound=2.5, x=torch.zeros(args.rays * 128, 3, device='cuda'), n_points=args.rays * 128))['groups'])}
        print(json.dumps(rec), flush=True)
        out.append(rec)
        del model, renderer, loader, opt
        torch.cuda.empty_cache()
    ms = [o.get('ms_per_step_with_modelled_wire', o.get('ms_per_step_without_wire')) for o in out]
    steps_run = (3 + args.steps) * len(out)
    summary = {'world': W, 'max_ms': max(ms), 'mean_ms': sum(ms) / len(ms), 'link_gbs_per_direction': args.link_gbs,
               'latency_us': args.latency_us, 'spin_cycles_per_us': round(cyc_per_us, 2),
               'wire_ms_per_step_if_serial': round(wire_log['us'] / steps_run / 1e3, 3),
               'collectives_per_step': round(wire_log['calls'] / steps_run, 1),
               'busiest_link_MB_per_step': round(wire_log['bytes'] / steps_run / 1e6, 1)}
    if args.single_ms > 0:
        summary['single_gpu_ms'] = args.single_ms
        summary['projected_weak_scaling_efficiency'] = round(args.single_ms / max(ms), 3)
    print(json.dumps(summary))


if __name__ == '__main__':
    main()
