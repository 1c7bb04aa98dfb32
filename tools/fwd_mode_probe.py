#!/usr/bin/env python3
"""Why did nerf_field_fwd run at 0.49 ms inside the training loop and at 0.68 ms everywhere else?  Times the fused
forward of the bench configuration in isolation under the suspected conditions (allocator state, cache contents, busy
or idle GPU before it, training steps before it).  Answer: none of them -- 0.68 ms is the kernel's
time on an untrained or correctly trained table; the 0.49 ms came from a table damaged by the first-step Adam-moment
race, whose exploded entries collapsed part of the fine samples and made the gathers more local."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def t_fwd(model, renderer, loader, n=20, before=None):
    from nerf_amd import _lib
    n_, h, w, K, E, gt, bg, _ = loader.get_random_batch(4096)
    with torch.no_grad():
        for _ in range(3):
            renderer.render(n_, h, w, K, E, bg)
    _lib.kernel_timing(True)
    with torch.no_grad():
        for _ in range(n):
            if before is not None:
                before()
            renderer.render(n_, h, w, K, E, bg)
    prof = _lib.kernel_timing(False)
    return {k: round(v[1] / v[0], 4) for k, v in prof.items() if 'fwd' in k and ('field' in k or 'density' in k)}


def main():
    dev = torch.device('cuda', 0)
    model, renderer, loader, opt = bench.build_gpu(dev, 0, 4096, pipeline=True)
    print('fresh', t_fwd(model, renderer, loader), flush=True)
    a = torch.empty(256 << 20, dtype=torch.float32, device=dev)       # 1 GB, unrelated to the table
    b = torch.empty_like(a)
    print('1 GB copy of unrelated memory before each render', t_fwd(model, renderer, loader, before=lambda: b.copy_(a)), flush=True)
    m1 = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    print('bf16 8192^3 matmul before each render', t_fwd(model, renderer, loader, before=lambda: torch.mm(m1, m1)), flush=True)
    tbl = model.encoder.params
    print('table.mul_(1.0) before each render', t_fwd(model, renderer, loader, before=lambda: tbl.data.mul_(1.0)), flush=True)
    dense_end = int(model.encoder.cfg.offset[9]) * 2
    print('dense levels .mul_(1.0) before each render', t_fwd(model, renderer, loader, before=lambda: tbl.data[:dense_end].mul_(1.0)), flush=True)
    del a, b, m1
    big = torch.empty(700 << 20, dtype=torch.uint8, device=dev)
    for it in range(3):
        bench.train_step(model, renderer, loader, opt, 4096, it, 1)
    torch.cuda.synchronize()
    print('after 3 train steps (binned=%s)' % os.environ.get('NERF_TABLE_BINNED', '1'), t_fwd(model, renderer, loader), flush=True)
    del big
    torch.cuda.empty_cache()
    print('after freeing', t_fwd(model, renderer, loader), flush=True)
    time.sleep(2)
    print('after 2 s idle', t_fwd(model, renderer, loader), flush=True)


if __name__ == '__main__':
    main()
