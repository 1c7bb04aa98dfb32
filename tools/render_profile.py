#!/usr/bin/env python3
"""render_image of one 800x800 view of the bench scene, a few times: the program to put under
`rocprofv3 --kernel-trace --stats` for the inference path (inference.py:54-94).

    python tools/render_profile.py [reps] [rays per chunk]
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    from nerf_amd.inference import Inferencer
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    chunk = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    dev = torch.device('cuda', 0)
    model, renderer, loader, opt = bench.build_gpu(dev, 0, 4096)
    for it in range(int(os.environ.get('TRAIN_STEPS', '20'))):
        bench.train_step(model, renderer, loader, opt, 4096, it, 1)
    inf = Inferencer(renderer, chunk, 0, 0, 0.1, 0.05, 0.4, [1, 2, 3, 4], 20)
    n, h, w, K, E, _, _, _ = loader.get_image_batch(0)
    inf.render_image(n, h, w, K, E)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        inf.render_image(n, h, w, K, E)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(json.dumps({'rays': int(n.numel()), 'chunk': chunk, 'ms_per_image': dt * 1e3, 'rays_per_s': n.numel() / dt}))


if __name__ == '__main__':
    main()
