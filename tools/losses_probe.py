#!/usr/bin/env python3
"""Where do the 33 us of nerf_losses go?  Times the kernel on the bench shape (4096 rays x 128 samples) with parts of
its work switched off through its own arguments."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from nerf_amd._lib import check, lib, ptr, stream
    N, S = 4096, 128
    g = torch.Generator().manual_seed(0)
    img, gt = torch.rand(N, 3, generator=g).cuda(), torch.rand(N, 3, generator=g).cuda()
    w = (torch.rand(N, S, generator=g) / S).cuda()
    s = torch.sort(torch.rand(N, S, generator=g), dim=-1).values.cuda().contiguous()
    sums = torch.zeros(8, dtype=torch.float64, device='cuda')
    dimg, dw = torch.empty_like(img), torch.empty_like(w)

    def run(use_w, grads, n=N):
        check(lib().nerf_losses(ptr(img), ptr(gt), ptr(w) if use_w else None, ptr(s) if use_w else None, n, S, 1e-3, 1e-1, 1.0, 1,
                                ptr(sums), ptr(dimg) if grads else None, ptr(dw) if (grads and use_w) else None, stream()), 'nerf_losses')

    for name, a, b, n in (('full', True, True, N), ('no gradients', True, False, N), ('rgb only', False, True, N),
                          ('full, 1024 rays', True, True, 1024), ('full, 256 rays', True, True, 256)):
        for _ in range(5):
            run(a, b, n)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            run(a, b, n)
        e1.record()
        torch.cuda.synchronize()
        print('%-18s %.2f us' % (name, e0.elapsed_time(e1) / 200 * 1e3))


if __name__ == '__main__':
    main()
