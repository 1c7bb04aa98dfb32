#!/usr/bin/env python3
"""The per-ray kernels (resampling, compositing, the fused training tail and the backward passes) at 65 536 rays x
(64 coarse, 128 fine) samples, each entry point through the C ABI: prints one JSON line of
{name, avg_launch_ms, algorithmic bytes, GB/s, fraction of 8 TB/s} (bench.per_ray_kernels_at).  Meant to run under
`rocprofv3 --kernel-trace --stats` so that the kernel durations of the same launches are in the committed summary.

    python tools/per_ray_bench.py [--rays 65536]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--rays', type=int, default=65536)
    ap.add_argument('--reps', type=int, default=20)
    a = ap.parse_args()
    print(json.dumps({'rays': a.rays, 'kernels': bench.per_ray_kernels_at(a.rays, reps=a.reps)}))
