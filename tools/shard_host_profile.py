#!/usr/bin/env python3
"""cProfile of the host side of the emulated level-sharded step (tools/shard_step_emulate.py, one rank, no wire)."""
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import torch

import bench
from nerf_amd import parallel

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
parallel.world_size = lambda: W
parallel.rank = lambda: 0
parallel.all_gather_rows = lambda t: t.repeat(*([W] + [1] * (t.dim() - 1)))


def fake_a2a(recv_blocks, send_blocks, async_op=False):
    return None


parallel.all_to_all_blocks = fake_a2a
parallel.all_reduce_grads = lambda params, async_op=False, skip=(), flags=(): []
model, renderer, loader, opt = bench.build_gpu(torch.device('cuda', 0), 0, 4096, shard_levels=True)
for it in range(5):
    bench.train_step(model, renderer, loader, opt, 4096, it, W)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for it in range(20):
    bench.train_step(model, renderer, loader, opt, 4096, it, W)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(45)
