#!/usr/bin/env python3
"""ms per iteration of Trainer.train_epoch (the user-facing loop, with its per-iteration scalar logging and LR
scheduler) on the bench configuration, next to bench.py's bare step.

    python tools/trainer_bench.py
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    from nerf_amd.logger import Logger
    from nerf_amd.trainer import Trainer
    dev = torch.device('cuda', 0)
    model, renderer, loader, opt = bench.build_gpu(dev, 0, 4096)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda it: 0.1 ** (it / 10000))
    tr = Trainer(model, loader, Logger(None, verbose=False), None, renderer, opt, sched, 4096, 100, 100, None, None, None,
                 1e-5, 1e-2, 1e-1, 1e-1)
    tr.train_epoch(20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.train_epoch(100)
    torch.cuda.synchronize()
    t_tr = (time.perf_counter() - t0) / 100
    t0 = time.perf_counter()
    for it in range(100):
        bench.train_step(model, renderer, loader, opt, 4096, it, 1)
    torch.cuda.synchronize()
    t_b = (time.perf_counter() - t0) / 100
    print(json.dumps({'trainer_ms_per_iter': t_tr * 1e3, 'bench_step_ms': t_b * 1e3}))


if __name__ == '__main__':
    main()
