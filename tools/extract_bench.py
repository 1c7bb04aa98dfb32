#!/usr/bin/env python3
"""Timing of the dense extraction at SURVEY's C5 size (512^3 voxel centres, batches of 524 288, 128^3 camera mask,
sigma threshold 100; allign/config.yaml:21-24) on the bench network and scene.

    python tools/extract_bench.py
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    from nerf_amd.extract_dense_geometry import ExtractDenseGeometry
    dev = torch.device('cuda', 0)
    model, renderer, loader, opt = bench.build_gpu(dev, 0, 4096)
    for it in range(20):                    # a few training steps so that some density exists
        bench.train_step(model, renderer, loader, opt, 4096, it, 1)
    N, H, W, K, E = loader.get_calibration()
    ex = ExtractDenseGeometry(model, loader, 128, 512, 524288, 100.0, bench.OUTER)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mask = ex.get_valid_positions(N, H, W, K, E, res=128)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    pts, col = ex.extract_dense_geometry(N, H, W, K, E, mask=mask)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    kept = int(mask.sum())
    queried = kept / float(128 ** 3) * 512 ** 3
    print(json.dumps({'workload': 'C5: 512^3 voxels, batch 524288, mask 128^3, thresh 100', 'mask_seconds': t1 - t0,
                      'extract_seconds': t2 - t1, 'mask_fraction_kept': kept / float(128 ** 3),
                      'points_queried': queried, 'field_points_per_s': queried / (t2 - t1), 'points_out': int(pts.shape[0])}))


if __name__ == '__main__':
    main()
