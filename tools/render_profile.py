"""Kernel timing of one 800 x 800 render_image (coherent traversal, 65 536-ray chunks) on the metric network: which kernels
the 68 ms are made of.  Run under `rocprofv3 --kernel-trace --stats` or alone (prints the library's own per-entry-point timing)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    prec = sys.argv[1] if len(sys.argv) > 1 else 'float32'
    bench.NET['encoding_precision'] = prec
    model, renderer, loader, opt = bench.build_gpu(dev, 0, 4096, pipeline=True)
    from nerf_amd.inference import Inferencer
    from nerf_amd import _lib
    inf = Inferencer(renderer, 65536, 0, 0, 0.1, 0.05, 0.4, [1, 2, 3, 4], 20)
    n, h, w, K, E, _, _, _ = loader.get_image_batch(0)
    for _ in range(2):
        inf.render_image(n, h, w, K, E)
    torch.cuda.synchronize()
    _lib.kernel_timing(True)
    t0 = time.perf_counter()
    for _ in range(3):
        inf.render_image(n, h, w, K, E)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    rows = _lib.kernel_timing(False)
    print('%s: %.2f ms per image' % (prec, dt * 1e3))
    for name, (calls, ms) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print('  %-34s %4d calls  %8.3f ms per image' % (name, calls // 3, ms / 3))


if __name__ == '__main__':
    main()
