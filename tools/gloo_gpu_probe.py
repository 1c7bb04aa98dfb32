#!/usr/bin/env python3
"""Probe (no nerf_amd code): two gloo ranks sharing cuda:0, many async all-reduces of CUDA slices issued while
kernels are still queued on the main stream, then a synchronous one.  Tells a torch/gloo/HIP quirk from a bug in
our pipeline when the single-GPU rehearsal of bench.py --gpus 2 stalls."""
import faulthandler
import os
import sys
import time

import torch
import torch.distributed as dist

faulthandler.dump_traceback_later(30, repeat=True)
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
torch.cuda.set_device(0)
dist.init_process_group('gloo', rank=rank, world_size=world)
n_async, n = int(sys.argv[1]), int(sys.argv[2])
from_thread = len(sys.argv) > 3 and sys.argv[3] == 'thread'
big = torch.ones(n_async * n, device='cuda')
a = torch.randn(4096, 4096, device='cuda')
small = [torch.ones(k, device='cuda') for k in (4800, 3000, 9000)]
for it in range(3):
    t0 = time.time()
    works = []

    def issue():
        for i in range(n_async):
            b = a @ a                       # keeps the main stream busy while the collectives are issued
            big[i * n:(i + 1) * n] += b.view(-1)[:n] * 0
            works.append(dist.all_reduce(big[i * n:(i + 1) * n], async_op=True))
    if from_thread:
        import threading
        th = threading.Thread(target=issue)
        th.start()
        th.join()
    else:
        issue()
    for s in small:
        dist.all_reduce(s)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for w in works:
            w.wait()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    print('rank', rank, 'iter', it, 'ok', round(time.time() - t0, 3), float(big[0]), flush=True)
dist.destroy_process_group()
