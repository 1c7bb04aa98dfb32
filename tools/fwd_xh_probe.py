#!/usr/bin/env python3
"""What do the x_hashtable rows cost in nerf_field_fwd?  Times the fused forward of the bench configuration on one
batch of fine samples with and without the feature output (8-byte stores per lane and level)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    from nerf_amd._lib import check, lib, ptr, stream
    from nerf_amd.nets import make_point_src
    dev = torch.device('cuda:0')
    net, renderer, loader, opt = bench.build_gpu(dev, 0, 4096)
    R, S = 4096, 128
    g = torch.Generator().manual_seed(0)
    ro = ((torch.rand(R, 3, generator=g) * 2 - 1) * 1.5).cuda()
    rd = torch.randn(R, 3, generator=g).cuda()
    z = torch.sort(torch.rand(R, S, generator=g) * 3 + 0.3, dim=-1).values.cuda().contiguous()
    n = torch.randint(0, net.N, (R,), generator=g).cuda()
    M = R * S
    src = make_point_src(outer_bound=2.5, inner_bound=2.0, rays_o=ro, rays_d=rd, z=z, z_stride=S, n_samples=S, n=n, n_points=M)
    sigma, rgb = torch.empty(M, device=dev), torch.empty(M, 3, device=dev)
    xh, xyz = torch.empty(M, 40, device=dev), torch.empty(M, 3, device=dev)

    def run(with_xh, with_xyz):
        check(lib().nerf_field_fwd(C.byref(net.encoder.cfg), C.byref(net.net_cfg), ptr(net.encoder.params),
                                   ptr(net.sigma_net.params), ptr(net.color_net.params), ptr(net.latent_emb), net.N,
                                   C.byref(src), ptr(sigma), ptr(rgb), ptr(xh) if with_xh else None,
                                   ptr(xyz) if with_xyz else None, stream()), 'nerf_field_fwd')

    for name, a, b in (('sigma+rgb only', False, False), ('+ x_hashtable', True, False), ('+ x_hashtable + xyz', True, True),
                       ('sigma+rgb only', False, False)):
        for _ in range(3):
            run(a, b)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run(a, b)
        e1.record()
        torch.cuda.synchronize()
        print('%-22s %.4f ms' % (name, e0.elapsed_time(e1) / 20))


if __name__ == '__main__':
    main()
