#!/usr/bin/env python3
"""Golden vectors for the Inferencer entry points (run in the build container only): the REFERENCE's own
``Inferencer.render_image`` / ``render_invdepth_thresh`` / ``extract_surface_geometry`` (/root/reference/inference.py:54-223,
imported, never copied) driven on the CPU by the reference renderer with the closed-form ``AnalyticField``.

inference.py names the device 'cuda' literally (``torch.ones(3, device='cuda')``, ``.to('cuda')``); this container has no
GPU, so the calls run under a TorchFunctionMode that maps every 'cuda' device argument to 'cpu' -- the reference's code
and arithmetic are untouched.  The stratification noise of renderer.py:102 is injected chunk by chunk and stored.

    python tests/golden/make_golden_infer.py        ->  tests/golden/infer.npz (numeric arrays only)
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, '/root/reference')

import numpy as np
import torch
from torch.overrides import TorchFunctionMode

import renderer as ref_renderer      # /root/reference/renderer.py
import inference as ref_inference    # /root/reference/inference.py

from make_golden import make_cameras
from oracle.ref_cpu import AnalyticField

torch.set_num_threads(1)


class CudaToCpu(TorchFunctionMode):
    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if kwargs.get('device') == 'cuda':
            kwargs['device'] = 'cpu'
        args = tuple('cpu' if (isinstance(a, str) and a == 'cuda') else a for a in args)
        return func(*args, **kwargs)


def main():
    H, W, n_img, Sf, chunk = 10, 15, 7, 128, 64
    K_all, E_all = make_cameras(n_img, 5)
    # an opaque blob in front of a thin haze: rays through the blob have a sharp surface (small depth spread), rays past
    # it accumulate slowly (rejected by the variance masks) or never reach the thresholds (depth 0)
    field = AnalyticField(centers=((0.1, 0.0, 0.0), (-0.5, 0.4, 0.2)), widths=(0.12, 0.3), peaks=(1e8, 40.0), floor=0.05)
    R = ref_renderer.NerfRenderer(field, 2, 2.5, [1.6, 4.8], [64], Sf, 0.2)
    hh, ww = torch.meshgrid(torch.arange(H) * 70 + 50, torch.arange(W) * 50 + 20, indexing='ij')
    n = torch.full(hh.shape, 3)
    K, E = K_all[n], E_all[n]
    total = H * W
    g = torch.Generator().manual_seed(77)
    noise = torch.rand(total, Sf, generator=g)
    inf = ref_inference.Inferencer(R, chunk, 3, 0, 0.1, 0.05, 0.4, [1, 2, 3, 4], 20)

    real_rand = torch.rand

    def run(fn):
        pos = [0]

        def fake_rand(shape, device=None):
            a = pos[0]
            pos[0] += shape[0]
            assert shape[1] == Sf
            return noise[a:a + shape[0]].clone()
        torch.rand = fake_rand
        try:
            with CudaToCpu():
                return fn(n, hh, ww, K, E)
        finally:
            torch.rand = real_rand
    image, invdepth = run(inf.render_image)
    invdepth_thresh = run(inf.render_invdepth_thresh)
    pc = run(inf.extract_surface_geometry)
    # what render() returned for the same rays and noise (inputs of the restatement tests)
    pos = [0]

    def fake_rand(shape, device=None):
        a = pos[0]
        pos[0] += shape[0]
        return noise[a:a + shape[0]].clone()
    torch.rand = fake_rand
    try:
        _, weights, _, aux = R.render(n.reshape(-1), hh.reshape(-1), ww.reshape(-1), K.reshape(-1, 3, 3), E.reshape(-1, 4, 4),
                                      torch.ones(3))
    finally:
        torch.rand = real_rand
    out = {'n': n, 'h': hh, 'w': ww, 'K': K, 'E': E, 'noise': noise, 'image': torch.from_numpy(image),
           'invdepth': torch.from_numpy(invdepth), 'invdepth_thresh': torch.from_numpy(invdepth_thresh),
           'points': torch.from_numpy(pc['points']), 'colors': torch.from_numpy(pc['colors']),
           'depth_variance': torch.from_numpy(pc['depth_variance']), 'weights': weights, 'z_vals': aux['z_vals'],
           'xyzs': aux['xyzs'], 'rgbs': aux['rgbs'],
           'field_centers': field.centers, 'field_widths': field.widths, 'field_peaks': field.peaks,
           'field_floor': torch.tensor(field.floor), 'chunk': torch.tensor(chunk)}
    np.savez_compressed(os.path.join(HERE, 'infer.npz'), **{k: v.detach().numpy() for k, v in out.items()})
    print('infer.npz: %d surface points, %d of %d pixels with a thresholded depth' %
          (pc['points'].shape[0], int((invdepth_thresh > 0).sum()), total))


if __name__ == '__main__':
    main()
