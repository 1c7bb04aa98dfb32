#!/usr/bin/env python3
"""Capture golden vectors from the REFERENCE renderer (run in the build container only).

Imports ``renderer.py`` / ``helpers.py`` / ``inference.py`` from ``/root/reference`` (never
copied, never shipped), drives them with the closed-form ``oracle.ref_cpu.AnalyticField`` and
injected stratification noise, and writes inputs + outputs of every renderer method to
``tests/golden/*.npz``.  Only numeric arrays are stored.

    python tests/golden/make_golden.py

The reference network (``nets.py``) needs the CUDA-only ``tinycudann`` and is not importable;
network parity is therefore "unpinned" (see oracle/ref_cpu.py header).
"""
import contextlib
import json
import math
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, '/root/reference')

import numpy as np
import torch

import renderer as ref_renderer      # /root/reference/renderer.py
import helpers as ref_helpers        # /root/reference/helpers.py
import inference as ref_inference    # /root/reference/inference.py

from oracle.ref_cpu import AnalyticField

torch.set_num_threads(1)   # deterministic reductions in the captured values


def make_cameras(n_img, seed, radius=3.2, H=800, W=800, focal=1111.1):
    """OpenCV-style camera-to-world poses on a sphere looking at the origin (+z forward)."""
    g = torch.Generator().manual_seed(seed)
    az = torch.rand(n_img, generator=g) * 2 * math.pi
    el = torch.rand(n_img, generator=g) * 0.9 + 0.1
    pos = radius * torch.stack([torch.cos(el) * torch.cos(az), torch.cos(el) * torch.sin(az), torch.sin(el)], -1)
    fwd = -pos / pos.norm(dim=-1, keepdim=True)
    up = torch.tensor([0.0, 0.0, 1.0]).expand_as(fwd)
    right = torch.cross(fwd, up, dim=-1)
    right = right / right.norm(dim=-1, keepdim=True)
    down = torch.cross(fwd, right, dim=-1)
    E = torch.zeros(n_img, 4, 4)
    E[:, :3, 0], E[:, :3, 1], E[:, :3, 2], E[:, :3, 3] = right, down, fwd, pos
    E[:, 3, 3] = 1
    K = torch.zeros(n_img, 3, 3)
    K[:, 0, 0] = focal
    K[:, 1, 1] = focal
    K[:, 0, 2] = W / 2
    K[:, 1, 2] = H / 2
    K[:, 2, 2] = 1
    return K, E


@contextlib.contextmanager
def patched(noise, record):
    """Inject `noise` for torch.rand and record searchsorted inputs/outputs inside sample_pdf."""
    real_rand, real_ss = torch.rand, torch.searchsorted

    def fake_rand(shape, device=None):
        assert tuple(shape) == tuple(noise.shape)
        return noise.clone()

    def rec_ss(seq, vals, right=False):
        out = real_ss(seq, vals, right=right)
        record['u'] = vals.clone()
        record['inds'] = (out + 1).clone()
        return out

    torch.rand, torch.searchsorted = fake_rand, rec_ss
    try:
        yield
    finally:
        torch.rand, torch.searchsorted = real_rand, real_ss


def capture_case(name, seed, N, steps, z_bounds, inner, outer, Sf=128, alpha=0.2, n_img=7):
    g = torch.Generator().manual_seed(1000 + seed)
    K_all, E_all = make_cameras(n_img, seed)
    n = torch.randint(0, n_img, (N,), generator=g)
    h = torch.randint(0, 800, (N,), generator=g)
    w = torch.randint(0, 800, (N,), generator=g)
    K, E = K_all[n], E_all[n]
    bg = torch.rand(3, generator=g)
    noise = torch.rand(N, Sf, generator=g)
    rgb_gt = torch.rand(N, 3, generator=g)
    field = AnalyticField()
    R = ref_renderer.NerfRenderer(field, inner, outer, z_bounds, steps, Sf, alpha)

    out = {'n': n, 'h': h, 'w': w, 'K': K, 'E': E, 'bg': bg, 'noise': noise, 'rgb_gt': rgb_gt}
    rays_o, rays_d = R.get_rays(h, w, K, E)
    out['rays_o'], out['rays_d'] = rays_o.contiguous(), rays_d
    zlog_c, z_c = R.get_uniform_z_vals(rays_o, rays_d, steps)
    out['z_vals_log_coarse'], out['z_vals_coarse'] = zlog_c[0].contiguous(), z_c[0].contiguous()   # ray independent
    xyz_c, dirs_c = R.get_sample_points(rays_o, rays_d, z_c)
    out['dirs'] = dirs_c[:, 0, :].contiguous()
    s_xyz_c = R.mipnerf360_scale(xyz_c, inner, outer)
    if N <= 64:
        out['xyzs_coarse'], out['s_xyzs_coarse'] = xyz_c, s_xyz_c
    sig_c, _ = field.density(s_xyz_c, outer)
    out['sigmas_coarse'] = sig_c
    _, _, w_c = R.render_rays_log(sig_c, torch.zeros(*sig_c.shape, 3), z_c, zlog_c)   # same formula as renderer.py:154-160
    out['weights_coarse'] = w_c
    w_re = R.reweight_the_weights(w_c, alpha, steps)
    out['weights_re'] = w_re
    rec = {}
    with patched(noise, rec):
        zlog_f = R.sample_pdf(zlog_c, w_re, Sf)
    out['cdf'] = torch.cumsum(w_re, dim=-1)
    out['u'], out['inds'], out['z_vals_log_fine_from_pdf'] = rec['u'], rec['inds'], zlog_f

    rec2 = {}
    with patched(noise, rec2):
        image, weights, zlog_s, aux = R.render(n, h, w, K, E, bg)
    out['inds_render'] = rec2['inds']
    out['image'], out['weights'], out['z_vals_log_s'] = image, weights, zlog_s
    for k in ('invdepth', 'z_vals', 'z_vals_log', 'sigmas', 'rgbs'):
        out['aux_' + k] = aux[k]
    if N <= 64:
        out['aux_xyzs'] = aux['xyzs']

    # losses + gradients w.r.t. the field outputs (trainer.py:153-169, helpers.py:11-32)
    sig_leaf = aux['sigmas'].clone().requires_grad_(True)
    rgb_leaf = aux['rgbs'].clone().requires_grad_(True)
    img2, inv2, w2 = R.render_rays_log(sig_leaf, rgb_leaf, aux['z_vals'], aux['z_vals_log'])
    img2 = img2 + (1 - torch.sum(w2, dim=-1)[..., None]) * bg
    l_rgb = ref_helpers.criterion_rgb(img2, rgb_gt)
    l_dist = ref_helpers.criterion_dist(w2, zlog_s)
    l_depth = torch.mean(w2 * (1 - zlog_s))
    lam_d, lam_z = 3e-3, 1e-1
    loss = l_rgb + lam_d * l_dist + lam_z * l_depth
    loss.backward()
    out['loss_rgb'], out['loss_dist'], out['loss_depth'], out['loss'] = l_rgb.detach(), l_dist.detach(), l_depth.detach(), loss.detach()
    out['lam_dist'], out['lam_depth'] = torch.tensor(lam_d), torch.tensor(lam_z)
    out['psnr'] = torch.tensor(ref_helpers.psnr(img2.detach(), rgb_gt))
    out['grad_sigmas'], out['grad_rgbs'] = sig_leaf.grad, rgb_leaf.grad

    inf = ref_inference.Inferencer(R, 4096, 0, 0, 0.1, 0.05, 0.4, [1, 2, 3, 4], 20)
    for t in (0.3, 0.5, 0.7):
        out['cum_thresh_%d' % int(t * 10)] = inf.calculate_cumlative_weights_thresh(weights, t)

    meta = {'seed': seed, 'N': N, 'steps': steps, 'z_bounds': z_bounds, 'inner': inner, 'outer': outer,
            'Sf': Sf, 'alpha': alpha, 'n_img': n_img, 'torch': torch.__version__}
    np.savez_compressed(os.path.join(HERE, name + '.npz'),
                        **{k: v.detach().numpy() for k, v in out.items()})
    return meta


def capture_edges():
    """Crafted inputs for the per-ray methods: zero density, opaque spike, cdf ties, r == inner."""
    R = ref_renderer.NerfRenderer(None, 2, 2.5, [1.6, 4.8], [64], 128, 0.2)
    S, Sf = 64, 128
    zlog_c, z_c = R.get_uniform_z_vals(torch.zeros(5, 3), torch.zeros(5, 3), [64])
    sig = torch.zeros(5, S)
    sig[1, 20] = 1e6                      # single opaque spike
    sig[2, :] = 50.0                      # uniform fog
    sig[3, 10:12] = 1e4                   # two-sample wall
    sig[4, 63] = 1e6                      # spike on the last sample (delta = 0 -> no weight)
    _, _, w_c = R.render_rays_log(sig, torch.zeros(5, S, 3), z_c, zlog_c)
    w_re = R.reweight_the_weights(w_c, 0.2, [64])
    g = torch.Generator().manual_seed(7)
    noise = torch.rand(5, Sf, generator=g)
    noise[0, :] = 0.0                     # u exactly on the stratification grid
    rec = {}
    with patched(noise, rec):
        zlog_f = R.sample_pdf(zlog_c, w_re, Sf)
    # exact ties in the cdf: pdf with zero entries (alpha = 0 would give this)
    pdf_t = torch.zeros(2, 16)
    pdf_t[0, [2, 3, 9]] = torch.tensor([0.25, 0.5, 0.25])
    pdf_t[1, :] = 1.0 / 16
    zl_t = torch.linspace(0.0, 1.5, 16)[None].expand(2, -1).contiguous()
    noise_t = torch.rand(2, 8, generator=g)
    rec_t = {}
    with patched(noise_t, rec_t):
        zf_t = R.sample_pdf(zl_t, pdf_t, 8)
    # contraction exactly at / around r == inner (F10)
    pts = torch.tensor([[[2.0, 0, 0], [0, 1.9999, 0], [0, 0, 2.0001], [10.0, 0, 0], [300.0, 400.0, 1200.0], [0, 0, 0]]])
    s21 = R.mipnerf360_scale(pts, 2, 2.5)
    s12 = R.mipnerf360_scale(pts, 1, 2)
    out = {'sigmas': sig, 'z_vals_log': zlog_c[0].contiguous(), 'z_vals': z_c[0].contiguous(), 'weights': w_c, 'weights_re': w_re, 'noise': noise,
           'u': rec['u'], 'inds': rec['inds'], 'z_vals_log_fine': zlog_f,
           'tie_pdf': pdf_t, 'tie_zlog': zl_t, 'tie_noise': noise_t, 'tie_u': rec_t['u'], 'tie_inds': rec_t['inds'],
           'tie_fine': zf_t, 'contract_pts': pts, 'contract_2_25': s21, 'contract_1_2': s12}
    np.savez_compressed(os.path.join(HERE, 'edges.npz'), **{k: v.detach().numpy() for k, v in out.items()})


if __name__ == '__main__':
    metas = {}
    metas['a0'] = capture_case('a0', 0, 64, [64], [1.6, 4.8], 2, 2.5)
    metas['a1'] = capture_case('a1', 1, 257, [64], [1.6, 4.8], 1, 2)
    metas['b0'] = capture_case('b0', 2, 64, [256, 128, 64, 32], [0.1, 1, 10, 100, 1000], 2, 2.5)
    metas['b1'] = capture_case('b1', 0, 33, [256, 128, 64, 32], [0.1, 1, 10, 100, 1000], 1, 2)
    capture_edges()
    with open(os.path.join(HERE, 'meta.json'), 'w') as f:
        json.dump(metas, f, indent=1)
    print('golden vectors written to', HERE)
