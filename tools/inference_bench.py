#!/usr/bin/env python3
"""Timing of the three Inferencer entry points (inference.py:54-223) on one 800x800 view of the bench scene.

    python tools/inference_bench.py
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    from nerf_amd.inference import Inferencer
    dev = torch.device('cuda', 0)
    model, renderer, loader, opt = bench.build_gpu(dev, 0, 4096)
    for it in range(20):
        bench.train_step(model, renderer, loader, opt, 4096, it, 1)
    inf = Inferencer(renderer, 65536, 0, 0, 0.1, 0.05, 0.4, [1, 2, 3, 4], 20)
    n, h, w, K, E, _, _, _ = loader.get_image_batch(0)
    out = {'rays': int(n.numel())}
    out['render_image_s'] = timed(lambda: inf.render_image(n, h, w, K, E))
    out['render_invdepth_thresh_s'] = timed(lambda: inf.render_invdepth_thresh(n, h, w, K, E))
    out['extract_surface_geometry_s'] = timed(lambda: inf.extract_surface_geometry(n, h, w, K, E))
    for k in list(out):
        if k.endswith('_s'):
            out[k[:-2] + '_rays_per_s'] = out['rays'] / out[k]
    print(json.dumps(out))


if __name__ == '__main__':
    main()
