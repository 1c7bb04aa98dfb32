#!/usr/bin/env python3
"""Full-size timing of the binned table update against scatter + Adam (config_base table, 4096 x 128 points).

    python tools/table_update_bench.py [--points-mult 1]
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--points-mult', type=int, default=1)
    ap.add_argument('--reps', type=int, default=5)
    a = ap.parse_args()
    from nerf_amd._lib import NerfLevelSet, check, lib, ptr, stream
    from nerf_amd.nets import HashGridEncoding, make_point_src
    enc = HashGridEncoding(20, 2, 24, device='cuda', seed=0)
    cfg, F = enc.cfg, 2
    R, S = 4096 * a.points_mult, 128
    g = torch.Generator().manual_seed(0)
    ro = ((torch.rand(R, 3, generator=g) * 2 - 1) * 1.5).cuda()
    rd = torch.randn(R, 3, generator=g).cuda()
    z = torch.sort(torch.rand(R, S, generator=g) * 3 + 0.3, dim=-1).values.cuda().contiguous()
    M = R * S
    src = make_point_src(outer_bound=2.5, inner_bound=2.0, rays_o=ro, rays_d=rd, z=z, z_stride=S, n_samples=S, n_points=M)
    dfeat = torch.randn(M, 40, device='cuda') * 1e-3
    levels = [l for l in range(20) if int(cfg.entries[l]) * F > 16 * 32768]
    ls = NerfLevelSet()
    ls.count = len(levels)
    for i, l in enumerate(levels):
        ls.level[i] = l
    cols = (C.c_int32 * len(levels))(*[l * F for l in levels])
    p = enc.params.detach().clone()
    m, v, gr = torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p)
    hyper = (1e-2, 0.9, 0.99, 1e-15, 0.0)
    nbytes = lib().nerf_table_bin_workspace_bytes(C.byref(cfg), C.byref(ls), M)
    ws = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
    print('levels', levels, 'workspace %.1f MB' % (nbytes / 1e6), 'params %.1f M' % (sum(int(cfg.entries[l]) * F for l in levels) / 1e6))

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / a.reps

    def bins():
        check(lib().nerf_table_bin_levels(C.byref(cfg), C.byref(src), C.byref(ls), ptr(dfeat), 40, cols, ptr(ws), nbytes, stream()), 'bin')

    def upd():
        check(lib().nerf_table_update_binned(C.byref(cfg), C.byref(ls), M, ptr(ws), nbytes, ptr(p), ptr(m), ptr(v), *hyper, 1, 1.0,
                                             stream()), 'upd')

    def scatter():
        for l in levels:
            check(lib().nerf_hashgrid_scatter(C.byref(cfg), C.byref(src), ptr(dfeat), ptr(gr), l, 1, stream()), 'scatter')

    def adam():
        for l in levels:
            a0 = int(cfg.offset[l]) * F
            b0 = a0 + int(cfg.entries[l]) * F
            check(lib().nerf_adam_step(ptr(p[a0:b0]), ptr(gr[a0:b0]), ptr(m[a0:b0]), ptr(v[a0:b0]), b0 - a0, *hyper, 1, 1.0, 1,
                                       stream()), 'adam')

    tb, tu, ts, ta = timed(bins), timed(upd), timed(scatter), timed(adam)
    print('bin passes %.3f ms   binned update %.3f ms   | scatter %.3f ms   adam %.3f ms (run back to back, not overlapped)' % (tb, tu, ts, ta))


if __name__ == '__main__':
    main()
