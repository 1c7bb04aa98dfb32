"""The CPU oracle (oracle/ref_cpu.py) must reproduce the golden vectors captured from the
reference renderer (tests/golden/make_golden.py).  Indices bit-exact, values <= 1e-6."""
import math

import pytest
import torch

from conftest import load_golden
from oracle import ref_cpu

CASES = ['a0', 'a1', 'b0', 'b1']


def _renderer(meta):
    return ref_cpu.OracleRenderer(ref_cpu.AnalyticField(), meta['inner'], meta['outer'], meta['z_bounds'],
                                  meta['steps'], meta['Sf'], meta['alpha'])


@pytest.mark.parametrize('case', CASES)
def test_stagewise(case, golden_meta):
    torch.set_num_threads(1)
    meta, g = golden_meta[case], load_golden(case)
    R = _renderer(meta)
    rays_o, rays_d = R.get_rays(g['h'], g['w'], g['K'], g['E'])
    assert torch.equal(rays_o, g['rays_o']) and torch.equal(rays_d, g['rays_d'])
    zlog_c, z_c = R.get_uniform_z_vals(meta['N'])
    assert torch.equal(zlog_c[0], g['z_vals_log_coarse']) and torch.equal(z_c[0], g['z_vals_coarse'])
    xyz, dirs = R.get_sample_points(rays_o, rays_d, z_c)
    assert torch.equal(dirs[:, 0], g['dirs'])
    s_xyz = R.mipnerf360_scale(xyz, meta['inner'], meta['outer'])
    if 'xyzs_coarse' in g:
        assert torch.equal(xyz, g['xyzs_coarse']) and torch.equal(s_xyz, g['s_xyzs_coarse'])
    sig, _ = R.model.density(s_xyz, meta['outer'])
    torch.testing.assert_close(sig, g['sigmas_coarse'], rtol=1e-6, atol=1e-6)
    # from the golden sigma on, each stage is checked on the golden inputs
    _, _, w_c = R.render_rays_log(g['sigmas_coarse'], torch.zeros(*sig.shape, 3), z_c, zlog_c)
    assert torch.equal(w_c, g['weights_coarse'])
    w_re = R.reweight_the_weights(g['weights_coarse'], meta['alpha'], meta['steps'])
    torch.testing.assert_close(w_re, g['weights_re'], rtol=1e-6, atol=1e-9)
    fine, dbg = R.sample_pdf(zlog_c, g['weights_re'], meta['Sf'], noise=g['noise'], return_all=True)
    assert torch.equal(dbg['cdf'], g['cdf'])
    assert torch.equal(dbg['u'], g['u'])
    assert torch.equal(dbg['inds'], g['inds'])                   # bit-exact indices (boundary P1)
    assert torch.equal(fine, g['z_vals_log_fine_from_pdf'])
    assert int(g['inds'].min()) >= 1 and int(g['inds'].max()) <= sum(meta['steps']) - 2


@pytest.mark.parametrize('case', CASES)
def test_render_and_losses(case, golden_meta):
    torch.set_num_threads(1)
    meta, g = golden_meta[case], load_golden(case)
    R = _renderer(meta)
    image, weights, zlog_s, aux = R.render(g['n'], g['h'], g['w'], g['K'], g['E'], g['bg'], noise=g['noise'])
    torch.testing.assert_close(aux['z_vals_log'], g['aux_z_vals_log'], rtol=0, atol=1e-6)
    torch.testing.assert_close(image, g['image'], rtol=0, atol=1e-6)
    torch.testing.assert_close(weights, g['weights'], rtol=0, atol=1e-6)
    torch.testing.assert_close(aux['invdepth'], g['aux_invdepth'], rtol=0, atol=1e-6)
    torch.testing.assert_close(zlog_s, g['z_vals_log_s'], rtol=0, atol=1e-6)
    # compositing + losses + autograd on the golden field outputs
    sig = g['aux_sigmas'].clone().requires_grad_(True)
    rgb = g['aux_rgbs'].clone().requires_grad_(True)
    img, inv, w = R.render_rays_log(sig, rgb, g['aux_z_vals'], g['aux_z_vals_log'])
    img = img + (1 - torch.sum(w, dim=-1)[..., None]) * g['bg']
    assert torch.equal(w, g['weights']) and torch.equal(inv, g['aux_invdepth'])
    l_rgb = ref_cpu.criterion_rgb(img, g['rgb_gt'])
    l_dist = ref_cpu.criterion_dist(w, g['z_vals_log_s'])
    l_depth = torch.mean(w * (1 - g['z_vals_log_s']))
    loss = l_rgb + float(g['lam_dist']) * l_dist + float(g['lam_depth']) * l_depth
    loss.backward()
    for a, b in ((l_rgb, g['loss_rgb']), (l_dist, g['loss_dist']), (l_depth, g['loss_depth']), (loss, g['loss'])):
        torch.testing.assert_close(a.detach(), b, rtol=1e-6, atol=1e-9)
    torch.testing.assert_close(sig.grad, g['grad_sigmas'], rtol=1e-5, atol=1e-10)
    torch.testing.assert_close(rgb.grad, g['grad_rgbs'], rtol=1e-5, atol=1e-10)
    assert abs(ref_cpu.psnr(img.detach(), g['rgb_gt']) - float(g['psnr'])) < 1e-5
    for t in (0.3, 0.5, 0.7):
        assert torch.equal(ref_cpu.cumulative_weights_thresh(g['weights'], t), g['cum_thresh_%d' % int(t * 10)])


def test_edges():
    torch.set_num_threads(1)
    g = load_golden('edges')
    R = ref_cpu.OracleRenderer(None, 2, 2.5, [1.6, 4.8], [64], 128, 0.2)
    zlog_c, z_c = R.get_uniform_z_vals(5)
    assert torch.equal(zlog_c[0], g['z_vals_log'])
    _, _, w = R.render_rays_log(g['sigmas'], torch.zeros(5, 64, 3), z_c, zlog_c)
    assert torch.equal(w, g['weights'])
    assert float(w[4].sum()) == 0.0            # spike on the last sample has delta 0
    w_re = R.reweight_the_weights(w, 0.2, [64])
    torch.testing.assert_close(w_re, g['weights_re'], rtol=1e-6, atol=1e-9)
    fine, dbg = R.sample_pdf(zlog_c, g['weights_re'], 128, noise=g['noise'], return_all=True)
    assert torch.equal(dbg['inds'], g['inds']) and torch.equal(fine, g['z_vals_log_fine'])
    fine_t, dbg_t = R.sample_pdf(g['tie_zlog'], g['tie_pdf'], 8, noise=g['tie_noise'], return_all=True)
    assert torch.equal(dbg_t['inds'], g['tie_inds'])
    assert torch.equal(torch.nan_to_num(fine_t, nan=-7.0), torch.nan_to_num(g['tie_fine'], nan=-7.0))
    assert torch.equal(R.mipnerf360_scale(g['contract_pts'], 2, 2.5), g['contract_2_25'])
    assert torch.equal(R.mipnerf360_scale(g['contract_pts'], 1, 2), g['contract_1_2'])


def test_grid_level_table():
    """Appendix B of SURVEY.md: config_base table = 200 132 568 entries, levels 0-8 dense."""
    levels, total = ref_cpu.grid_levels(20, 24)
    assert [lv.res for lv in levels[:9]] == [16, 23, 31, 43, 59, 81, 112, 154, 213]
    assert [lv.hashed for lv in levels] == [False] * 9 + [True] * 11
    assert total == 200132568


def test_inferencer_entry_points_vs_reference_golden():
    """tests/golden/infer.npz holds what the REFERENCE's Inferencer.render_image / render_invdepth_thresh /
    extract_surface_geometry (inference.py:54-223) returned on the CPU (tests/golden/make_golden_infer.py): the oracle's
    restatements must reproduce them -- image / inverse depth from the same rays, field and noise; thresholded depth and
    surface points bit for bit from the same weights."""
    g = load_golden('infer')
    field = ref_cpu.AnalyticField(centers=g['field_centers'].tolist(), widths=g['field_widths'].tolist(),
                                  peaks=g['field_peaks'].tolist(), floor=float(g['field_floor']))
    R = ref_cpu.OracleRenderer(field, 2, 2.5, [1.6, 4.8], [64], 128, 0.2)
    flat = lambda t, *s: t.reshape(-1, *s)
    with torch.no_grad():
        img, w, _, aux = R.render(flat(g['n']), flat(g['h']), flat(g['w']), flat(g['K'], 3, 3), flat(g['E'], 4, 4),
                                  torch.ones(3), noise=g['noise'])
    assert torch.equal(w, g['weights']) and torch.equal(aux['z_vals'], g['z_vals'])
    torch.testing.assert_close(img, flat(g['image'], 3), rtol=0, atol=1e-6)
    torch.testing.assert_close(aux['invdepth'], flat(g['invdepth']), rtol=0, atol=1e-6)
    got = ref_cpu.invdepth_thresh(g['weights'], g['z_vals'], 0.4, 0.05)
    assert torch.equal(got, flat(g['invdepth_thresh'])) and int((got > 0).sum()) == 33
    pts, col, var = ref_cpu.surface_geometry(g['weights'], g['z_vals'], g['xyzs'], g['rgbs'], 0.4, 0.1, 2)
    assert pts.shape[0] == 46
    assert torch.equal(pts, g['points']) and torch.equal(col, g['colors']) and torch.equal(var, g['depth_variance'])


def test_oracle_float16_table_mode():
    """`encoding_precision='float16'` of the oracle (nets.py:31-36,49; tinycudann's fp16 arithmetic itself is absent and
    unpinned): the table is rounded to fp16 where it is gathered, the gradient goes straight through to the fp32 master."""
    from oracle.ref_cpu import OracleNetwork
    kw = dict(encoding_n_levels=6, encoding_log2_hashmap_size=12, latent_embedding_dim=0, encoding_dir_degree=0)
    o16 = OracleNetwork(3, encoding_precision='float16', seed=1, **kw)
    o32 = OracleNetwork(3, encoding_precision='float32', seed=1, **kw)
    with torch.no_grad():
        o16.encoder_params.mul_(1e3)
        o32.encoder_params.copy_(o16.encoder_params.half().float())
    assert o16.half_table and not o32.half_table
    x = torch.rand(200, 3, generator=torch.Generator().manual_seed(2))
    e16, e32 = o16.encode(x), o32.encode(x)
    assert torch.equal(e16, e32)
    w = torch.randn(200, e16.shape[1], generator=torch.Generator().manual_seed(3))
    (e16 * w).sum().backward()
    (e32 * w).sum().backward()
    assert o16.encoder_params.grad is not None and torch.equal(o16.encoder_params.grad, o32.encoder_params.grad)
    assert float((o16.encoder_params.detach() - o16.encoder_params.detach().half().float()).abs().max()) > 0     # master keeps its fp32 bits
