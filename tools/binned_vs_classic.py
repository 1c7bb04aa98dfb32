#!/usr/bin/env python3
"""Train the bench configuration for a few steps with the binned table update and with scatter + Adam, same seeds:
the loss curves and the tables must agree to summation-order noise."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def run(binned, steps):
    os.environ['NERF_TABLE_BINNED'] = '1' if binned else '0'
    dev = torch.device('cuda', 0)
    model, renderer, loader, opt = bench.build_gpu(dev, 0, 4096, pipeline=True)
    assert bool(model.table_pipeline.binned_levels) == binned
    losses = []
    for it in range(steps):
        losses.append(float(bench.train_step(model, renderer, loader, opt, 4096, it, 1)))
    torch.cuda.synchronize()
    st = opt.state[model.encoder.params]
    return losses, model.encoder.params.detach().clone(), st['exp_avg'].clone(), st['exp_avg_sq'].clone(), model.encoder.cfg


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    l1, p1, m1, v1, cfg = run(True, steps)
    l0, p0, m0, v0, _ = run(False, steps)
    for i in range(0, steps, max(1, steps // 10)):
        print('step %3d  loss binned %.6f  classic %.6f' % (i, l1[i], l0[i]))
    print('final     loss binned %.6f  classic %.6f' % (l1[-1], l0[-1]))
    for l in range(cfg.n_levels):
        a = int(cfg.offset[l]) * 2
        b = a + int(cfg.entries[l]) * 2
        dp = float((p1[a:b] - p0[a:b]).abs().max())
        print('level %2d  max|dp| %.3e  max|p| %.3e  max|dm| %.3e  max|m| %.3e  max|dv| %.3e max|v| %.3e' % (
            l, dp, float(p0[a:b].abs().max()), float((m1[a:b] - m0[a:b]).abs().max()), float(m0[a:b].abs().max()),
            float((v1[a:b] - v0[a:b]).abs().max()), float(v0[a:b].abs().max())))


if __name__ == '__main__':
    main()
