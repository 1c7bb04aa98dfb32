"""GPU parity tests: the HIP path (through the C ABI of libnerf_hip.so) against
 (a) the golden vectors captured from the reference renderer and
 (b) the CPU oracle (oracle/ref_cpu.py) on the same seeded inputs.
Bars (BASELINE.json north_star): resampled indices bit-exact; composited RGB / depth / weights
within 1e-4 (fp32)."""
import math

import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu

CASES = ['a0', 'a1', 'b0', 'b1']


def _cuda(d):
    return {k: v.cuda() for k, v in d.items()}


@pytest.fixture(scope='module')
def R_for():
    from nerf_amd.renderer import NerfRenderer
    from oracle.ref_cpu import AnalyticField

    def make(meta, model=None):
        return NerfRenderer(model if model is not None else AnalyticField(), meta['inner'], meta['outer'],
                            meta['z_bounds'], meta['steps'], meta['Sf'], meta['alpha'])
    return make


@pytest.mark.parametrize('case', CASES)
def test_stages_vs_golden(case, golden_meta, R_for):
    meta, g = golden_meta[case], _cuda(load_golden(case))
    R = R_for(meta)
    N = meta['N']
    rays_o, rays_d = R.get_rays(g['h'], g['w'], g['K'], g['E'])
    assert torch.equal(rays_o, g['rays_o'])
    torch.testing.assert_close(rays_d, g['rays_d'], rtol=2e-6, atol=2e-6)
    zlog_c, z_c = R.get_uniform_z_vals(rays_o, rays_d, meta['steps'])
    assert torch.equal(zlog_c[0], g['z_vals_log_coarse']) and torch.equal(z_c[0], g['z_vals_coarse'])
    # sample points / contraction on the golden rays
    xyz, dirs = R.get_sample_points(g['rays_o'], g['rays_d'], z_c)
    torch.testing.assert_close(dirs[:, 0], g['dirs'], rtol=1e-6, atol=1e-7)
    if 'xyzs_coarse' in g:
        torch.testing.assert_close(xyz, g['xyzs_coarse'], rtol=1e-6, atol=1e-6)
        s = R.mipnerf360_scale(g['xyzs_coarse'], meta['inner'], meta['outer'])
        torch.testing.assert_close(s, g['s_xyzs_coarse'], rtol=1e-6, atol=1e-6)
    # coarse weights, re-weighting, resampling on golden inputs
    w_c = R.coarse_weights(g['sigmas_coarse'], zlog_c)
    torch.testing.assert_close(w_c, g['weights_coarse'], rtol=1e-5, atol=1e-7)
    w_re = R.reweight_the_weights(g['weights_coarse'], meta['alpha'], meta['steps'])
    torch.testing.assert_close(w_re, g['weights_re'], rtol=1e-6, atol=1e-9)
    fine, inds = R.sample_pdf(zlog_c, g['weights_re'], meta['Sf'], noise=g['noise'], return_inds=True)
    assert torch.equal(inds, g['inds'])                                   # bit-exact indices (boundary P1)
    assert torch.equal(fine, g['z_vals_log_fine_from_pdf'])
    # compositing forward on the golden field outputs
    image, weights, invdepth = None, None, None
    rgb, invdepth, weights = R.render_rays_log(g['aux_sigmas'], g['aux_rgbs'], g['aux_z_vals'], g['aux_z_vals_log'])
    torch.testing.assert_close(weights, g['weights'], rtol=0, atol=1e-6)
    torch.testing.assert_close(invdepth, g['aux_invdepth'], rtol=2e-5, atol=2e-6)   # reference sums w/z in fp32
    image = rgb + (1 - weights.sum(-1))[:, None] * g['bg']
    torch.testing.assert_close(image, g['image'], rtol=0, atol=1e-6)


@pytest.mark.parametrize('case', CASES)
def test_composite_losses_backward_vs_golden(case, golden_meta, R_for):
    from nerf_amd import helpers
    from nerf_amd.renderer import _CompositeFn
    meta, g = golden_meta[case], _cuda(load_golden(case))
    sig = g['aux_sigmas'].clone().requires_grad_(True)
    rgb = g['aux_rgbs'].clone().requires_grad_(True)
    lo = math.log10(meta['z_bounds'][0])
    rng = math.log10(meta['z_bounds'][-1]) - lo
    image, invdepth, weights, zlog_s = _CompositeFn.apply(sig, rgb, g['aux_z_vals'], g['aux_z_vals_log'], g['bg'], lo, rng)
    torch.testing.assert_close(image, g['image'], rtol=0, atol=1e-6)
    torch.testing.assert_close(zlog_s, g['z_vals_log_s'], rtol=0, atol=1e-6)
    loss, parts = helpers.fused_train_loss(image, g['rgb_gt'], weights, zlog_s, float(g['lam_dist']), float(g['lam_depth']))
    loss.backward()
    torch.testing.assert_close(parts[0], g['loss_rgb'], rtol=1e-5, atol=1e-9)
    torch.testing.assert_close(parts[1], g['loss_dist'], rtol=1e-5, atol=1e-9)
    torch.testing.assert_close(parts[2], g['loss_depth'], rtol=1e-5, atol=1e-9)
    torch.testing.assert_close(loss.detach(), g['loss'], rtol=1e-5, atol=1e-9)
    scale = float(g['grad_sigmas'].abs().max())
    torch.testing.assert_close(sig.grad, g['grad_sigmas'], rtol=1e-4, atol=1e-5 * scale)
    torch.testing.assert_close(rgb.grad, g['grad_rgbs'], rtol=1e-4, atol=1e-5 * float(g['grad_rgbs'].abs().max()))
    assert abs(helpers.psnr(image.detach(), g['rgb_gt']) - float(g['psnr'])) < 1e-4
    # individual reference-named losses, and the unsorted O(S^2) form of criterion_dist
    torch.testing.assert_close(helpers.criterion_rgb(g['image'], g['rgb_gt']), g['loss_rgb'], rtol=1e-5, atol=1e-9)
    torch.testing.assert_close(helpers.criterion_dist(g['weights'], g['z_vals_log_s']), g['loss_dist'], rtol=1e-5, atol=1e-9)
    torch.testing.assert_close(helpers.criterion_dist(g['weights'], g['z_vals_log_s'], assume_sorted=False), g['loss_dist'],
                               rtol=1e-5, atol=1e-9)
    from nerf_amd.inference import Inferencer
    inf = Inferencer(None, 4096, 0, 0, 0.1, 0.05, 0.4, [1, 2, 3, 4], 20)
    for t in (0.3, 0.5, 0.7):
        assert torch.equal(inf.calculate_cumlative_weights_thresh(g['weights'], t), g['cum_thresh_%d' % int(t * 10)])


@pytest.mark.parametrize('case', CASES)
def test_render_stub_field_vs_golden(case, golden_meta, R_for):
    """Whole render() with the analytic stub field: every renderer stage in HIP, field in torch."""
    meta, g = golden_meta[case], _cuda(load_golden(case))
    R = R_for(meta)
    image, weights, zlog_s, aux = R.render(g['n'], g['h'], g['w'], g['K'], g['E'], g['bg'], noise=g['noise'])
    # the resampled depths go through an fp32 pdf normalisation and a GPU-vs-CPU expf: looser bar
    torch.testing.assert_close(aux['z_vals_log'], g['aux_z_vals_log'], rtol=0, atol=2e-4)
    torch.testing.assert_close(image, g['image'], rtol=0, atol=1e-4)
    torch.testing.assert_close(weights, g['weights'], rtol=0, atol=1e-4)
    torch.testing.assert_close(aux['invdepth'], g['aux_invdepth'], rtol=0, atol=1e-4)
    torch.testing.assert_close(zlog_s, g['z_vals_log_s'], rtol=0, atol=1e-4)


def test_edges(R_for):
    g = _cuda(load_golden('edges'))
    meta = dict(inner=2, outer=2.5, z_bounds=[1.6, 4.8], steps=[64], Sf=128, alpha=0.2)
    R = R_for(meta)
    zlog_c, _ = R.get_uniform_z_vals(torch.zeros(5, 3, device='cuda'), None, None)
    w = R.coarse_weights(g['sigmas'], zlog_c)
    torch.testing.assert_close(w, g['weights'], rtol=1e-5, atol=1e-7)
    assert float(w[4].sum()) == 0.0
    fine, inds = R.sample_pdf(zlog_c, g['weights_re'], 128, noise=g['noise'], return_inds=True)
    assert torch.equal(inds, g['inds']) and torch.equal(fine, g['z_vals_log_fine'])
    fine_t, inds_t = R.sample_pdf(g['tie_zlog'], g['tie_pdf'], 8, noise=g['tie_noise'], return_inds=True)
    assert torch.equal(inds_t, g['tie_inds'])
    assert torch.equal(torch.nan_to_num(fine_t, nan=-7.0), torch.nan_to_num(g['tie_fine'], nan=-7.0))
    torch.testing.assert_close(R.mipnerf360_scale(g['contract_pts'], 2, 2.5), g['contract_2_25'], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(R.mipnerf360_scale(g['contract_pts'], 1, 2), g['contract_1_2'], rtol=1e-6, atol=1e-6)
    # ragged / tail sizes: 1, 3 and 5 rays (a partial 4-ray block), empty input
    for n in (1, 3, 5):
        f2, i2 = R.sample_pdf(zlog_c[:n], g['weights_re'][:n], 128, noise=g['noise'][:n], return_inds=True)
        assert torch.equal(i2, g['inds'][:n]) and torch.equal(f2, g['z_vals_log_fine'][:n])
    e = R.sample_pdf(zlog_c[:0], g['weights_re'][:0], 128, noise=g['noise'][:0])
    assert e.shape == (0, 128)


NET_CFGS = {
    'small': dict(encoding_n_levels=8, encoding_log2_hashmap_size=15),
    'odd': dict(encoding_n_levels=5, encoding_log2_hashmap_size=14),
    'base19': dict(encoding_n_levels=20, encoding_log2_hashmap_size=19),
    'f4': dict(encoding_n_levels=8, encoding_log2_hashmap_size=15, encoding_n_features_per_level=4),
    'f1': dict(encoding_n_levels=11, encoding_log2_hashmap_size=14, encoding_n_features_per_level=1),
    'sh2lat16': dict(encoding_n_levels=8, encoding_log2_hashmap_size=15, encoding_dir_degree=2, latent_embedding_dim=16),
}


def _make_pair(cfg, n_img=7, seed=3):
    """NeRFNetwork on the GPU and an OracleNetwork on the CPU holding identical parameters."""
    from nerf_amd.nets import NeRFNetwork
    from oracle.ref_cpu import OracleNetwork
    kw = dict(encoding_precision='float32', encoding_n_features_per_level=2, geo_feat_dim=15, sigma_hidden_dim=64,
              sigma_num_layers=2, encoding_dir_precision='float32', encoding_dir_encoding='SphericalHarmonics',
              encoding_dir_degree=4, latent_embedding_dim=48, color_hidden_dim=64, color_num_layers=3)
    kw.update(cfg)
    ora = OracleNetwork(n_img, seed=seed, **kw)
    with torch.no_grad():
        ora.encoder_params.mul_(1e3)      # lift the table out of its 1e-4 init so features matter
        ora.latent_emb.copy_(torch.randn(n_img, kw["latent_embedding_dim"], generator=torch.Generator().manual_seed(seed)) * 0.3)
    net = NeRFNetwork(n_img, **kw)
    with torch.no_grad():
        assert net.encoder.params.numel() == ora.encoder_params.numel()
        net.encoder.params.copy_(ora.encoder_params)
        net.sigma_net.params.copy_(ora.sigma_params)
        net.color_net.params.copy_(ora.color_params)
        net.latent_emb.copy_(ora.latent_emb)
    return net, ora


@pytest.mark.parametrize('name', list(NET_CFGS))
def test_field_forward_backward_vs_oracle(name):
    net, ora = _make_pair(NET_CFGS[name])
    g = torch.Generator().manual_seed(11)
    M = 1000 if name != 'base19' else 517     # ragged: not a multiple of 32 / 128
    x = (torch.rand(M, 3, generator=g) * 2 - 1) * 2.4
    d = torch.randn(M, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    n = torch.randint(0, 7, (M,), generator=g)
    ob = 2.5
    # stand-alone encoding
    x01 = (x + ob) / (2 * ob)
    enc_o = ora.encode(x01)
    enc_g = net.encoder(x01.cuda())
    torch.testing.assert_close(enc_g.cpu(), enc_o, rtol=1e-5, atol=1e-7)
    # density + full field
    sig_o, rgb_o, aux_o = ora(x, d, n, ob)
    sig_d, aux_d = net.density(x.cuda(), ob)
    sig_g, rgb_g, aux_g = net(x.cuda(), d.cuda(), n.cuda(), ob)
    scale = float(sig_o.abs().max()) + 1e-6
    torch.testing.assert_close(sig_g.cpu(), sig_o, rtol=1e-4, atol=1e-5 * scale)
    torch.testing.assert_close(sig_d.cpu(), sig_o, rtol=1e-4, atol=1e-5 * scale)
    torch.testing.assert_close(rgb_g.cpu(), rgb_o, rtol=0, atol=1e-5)
    torch.testing.assert_close(aux_g['x_hashtable'].cpu(), aux_o['x_hashtable'], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(aux_d['x_hashtable'].cpu(), aux_o['x_hashtable'], rtol=1e-5, atol=1e-7)
    # backward: random upstream gradients
    gs = torch.randn(M, generator=g)
    gc = torch.randn(M, 3, generator=g)
    (sig_o * gs).sum().add((rgb_o * gc).sum()).backward()
    (sig_g * gs.cuda()).sum().add((rgb_g * gc.cuda()).sum()).backward()
    for a, b, what in ((net.sigma_net.params.grad, ora.sigma_params.grad, 'sigma_net'),
                       (net.color_net.params.grad, ora.color_params.grad, 'color_net'),
                       (net.latent_emb.grad, ora.latent_emb.grad, 'latent'),
                       (net.encoder.params.grad, ora.encoder_params.grad, 'table')):
        s = float(b.abs().max()) + 1e-12
        torch.testing.assert_close(a.cpu(), b, rtol=1e-3, atol=2e-5 * s, msg=lambda m, w=what: w + ': ' + m)


def _batch(loader_seed, N, n_img=7):
    import sys
    sys.path.insert(0, 'tests/golden')
    from nerf_amd.loaders.procedural import sphere_cameras
    K_all, E_all = sphere_cameras(n_img, loader_seed)
    g = torch.Generator().manual_seed(50 + loader_seed)
    n = torch.randint(0, n_img, (N,), generator=g)
    h = torch.randint(0, 800, (N,), generator=g)
    w = torch.randint(0, 800, (N,), generator=g)
    return n, h, w, K_all[n], E_all[n], torch.rand(3, generator=g), torch.rand(N, 128, generator=g), torch.rand(N, 3, generator=g)


@pytest.mark.parametrize('steps,zb', [([64], [1.6, 4.8]), ([256, 128, 64, 32], [0.1, 1, 10, 100, 1000])])
def test_render_fused_vs_oracle(steps, zb):
    """Full render() with the HIP field network vs the CPU oracle on identical rays and weights."""
    from nerf_amd.renderer import NerfRenderer
    from oracle.ref_cpu import OracleRenderer
    net, ora = _make_pair(NET_CFGS['small'])
    n, h, w, K, E, bg, noise, gt = _batch(1, 130)
    Ro = OracleRenderer(ora, 2, 2.5, zb, steps, 128, 0.2)
    Rg = NerfRenderer(net, 2, 2.5, zb, steps, 128, 0.2)
    with torch.no_grad():
        img_o, w_o, zs_o, aux_o = Ro.render(n, h, w, K, E, bg, noise=noise)
    img_g, w_g, zs_g, aux_g = Rg.render(n.cuda(), h.cuda(), w.cuda(), K.cuda(), E.cuda(), bg.cuda(), noise=noise.cuda())
    torch.testing.assert_close(aux_g['z_vals_log'].cpu(), aux_o['z_vals_log'], rtol=0, atol=5e-5)
    torch.testing.assert_close(img_g.detach().cpu(), img_o, rtol=0, atol=1e-4)
    torch.testing.assert_close(w_g.detach().cpu(), w_o, rtol=0, atol=1e-4)
    torch.testing.assert_close(aux_g['invdepth'].cpu(), aux_o['invdepth'], rtol=0, atol=1e-4)
    torch.testing.assert_close(aux_g['xyzs'].cpu(), aux_o['xyzs'], rtol=1e-4, atol=1e-3)


def test_adam_vs_torch():
    from nerf_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(5)
    p0 = torch.randn(10007, generator=g)
    pc = torch.nn.Parameter(p0.clone())
    pg = torch.nn.Parameter(p0.clone().cuda())
    oc = torch.optim.Adam([{'params': [pc], 'lr': 2e-2, 'weight_decay': 1e-6}], betas=(0.9, 0.99), eps=1e-15)
    og = FusedAdam([{'params': [pg], 'lr': 2e-2, 'weight_decay': 1e-6}], betas=(0.9, 0.99), eps=1e-15)
    for it in range(5):
        gr = torch.randn(10007, generator=g) * (0.1 if it != 2 else 0.0)   # step 2: all-zero gradient
        pc.grad = gr.clone()
        pg.grad = gr.clone().cuda()
        oc.step()
        og.step()
        assert float(pg.grad.abs().max()) == 0.0          # cleared in the same pass
        torch.testing.assert_close(pg.detach().cpu(), pc.detach(), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize('name,binned', [('small', '1'), ('base19', '1'), ('base19', '0'), ('f4', '1'), ('f1', '1'), ('sh2lat16', '1')])
def test_train_steps_vs_oracle(name, binned, monkeypatch):
    """A short training trace (loss per step) of the HIP path against the CPU oracle:
    same rays, same noise, same init, Adam groups of train.py:84-88.  'base19' has levels large enough for the
    binned table update (csrc/table_update.hip); binned='0' forces the scatter -> Adam pipeline on the same levels."""
    from nerf_amd.optim import FusedAdam
    from nerf_amd.renderer import NerfRenderer
    from nerf_amd import helpers
    from oracle import ref_cpu
    monkeypatch.setenv('NERF_TABLE_BINNED', binned)
    net, ora = _make_pair(NET_CFGS[name])
    with torch.no_grad():
        ora.latent_emb.zero_()
        net.latent_emb.zero_()
    Ro = ref_cpu.OracleRenderer(ora, 2, 2.5, [1.6, 4.8], [64], 128, 0.2)
    Rg = NerfRenderer(net, 2, 2.5, [1.6, 4.8], [64], 128, 0.2)
    oo = ref_cpu.make_optimizer(ora)
    og = FusedAdam([{'params': list(net.encoder.parameters()), 'lr': 2e-2},
                    {'params': [net.latent_emb], 'lr': 2e-2},
                    {'params': list(net.sigma_net.parameters()) + list(net.color_net.parameters()), 'weight_decay': 1e-6, 'lr': 1e-3}],
                   betas=(0.9, 0.99), eps=1e-15).attach(net)
    assert bool(net.table_pipeline.binned_levels) == (name == 'base19' and binned == '1')
    lo, lg = [], []
    n_steps = 20 if (name, binned) in (('small', '1'), ('base19', '1')) else 4      # a 20-step trace
    w0 = ora.sigma_params.detach().clone()
    steady = torch.ones_like(w0, dtype=torch.bool)     # weights whose oracle gradient is above summation noise at EVERY step
    for it in range(n_steps):
        n, h, w, K, E, bg, noise, gt = _batch(10 + it, 96)
        oo.zero_grad()
        img, wts, zs, _ = Ro.render(n, h, w, K, E, bg, noise=noise)
        loss, _ = ref_cpu.train_loss(img, gt, wts, zs, it, 100, 1e-5, 1e-2, 1e-1, 1e-1)
        loss.backward()
        g_o = ora.sigma_params.grad.detach()
        steady &= g_o.abs() > 1e-2 * float(g_o.abs().max())
        oo.step()
        lo.append(float(loss))
        og.zero_grad()
        img, wts, zs, _ = Rg.render(n.cuda(), h.cuda(), w.cuda(), K.cuda(), E.cuda(), bg.cuda(), noise=noise.cuda())
        lam_d = ref_cpu.dist_scalar(it, 100, 1e-5, 1e-2)
        loss_g, _ = helpers.fused_train_loss(img, gt.cuda(), wts, zs, lam_d, 1e-1)
        loss_g.backward()
        og.step()
        lg.append(float(loss_g))
        if it == 0:
            # after ONE step no sample can have moved to another cell yet: the table's Adam moments must agree with
            # torch.optim.Adam's entry by entry (m = 0.1 g, v = 0.01 g^2), up to the summation order of the corner updates
            torch.cuda.synchronize()
            st1, so1 = og.state[net.encoder.params], oo.state[ora.encoder_params]
            m1, v1, mo1, vo1 = st1['exp_avg'].cpu(), st1['exp_avg_sq'].cpu(), so1['exp_avg'], so1['exp_avg_sq']
            # (a fine sample a few 1e-5 next to a cell face may sit in the neighbouring cell on the other side -- the fine
            # depths agree to 5e-5, not bit for bit -- so a small share of entries differs; their corner weights are ~0)
            for a_, b_, what in ((m1, mo1, 'exp_avg'), (v1, vo1, 'exp_avg_sq')):
                bad = (a_ - b_).abs() > 1e-3 * b_.abs() + 1e-4 * float(b_.abs().max())
                assert float(bad.float().mean()) < 5e-3, (what, float(bad.float().mean()))
                assert float((a_ - b_).norm() / b_.norm()) < 2e-2, what
            assert int((m1 != 0).sum()) > 1000 and float(((m1 != 0) != (mo1 != 0)).float().mean()) < 5e-3
    # the first steps agree to the parity bar; later ones drift apart slowly (Adam with eps = 1e-15 turns summation noise on
    # near-zero gradients into +-lr parameter steps), so the bar widens with the step count
    for k, (a, b) in enumerate(zip(lg, lo)):
        assert abs(a - b) <= (2e-4 if k < 4 else 2e-3) * max(1.0, abs(b)), (k, lg, lo)
    # Weights.  Adam with eps = 1e-15 moves EVERY weight by about lr per step whatever the size of its gradient, so a weight
    # whose gradient is summation noise (|g| << max |g|: its sign depends on the order of the sum) can differ by up to
    # 2 lr per step and says nothing about parity.  The bar is therefore set on the weights whose oracle gradient stayed
    # above 1 % of the largest one at every step -- they moved by ~n_steps * lr = 4e-3 .. 2e-2 and must agree to a small
    # fraction of that -- and on the distribution of all of them.
    wg, wo = net.sigma_net.params.detach().cpu(), ora.sigma_params.detach()
    diff = (wg - wo).abs()
    moved = (wo - w0).abs()
    assert int(steady.sum()) >= 50, int(steady.sum())
    assert float(moved[steady].median()) > 0.5e-3 * n_steps, float(moved[steady].median())
    stats = (name, binned, n_steps, int(steady.sum()), float(diff[steady].max()), float(diff.median()), float(diff.max()))
    assert float(diff[steady].max()) <= 0.1e-3 * n_steps, stats          # <= 10 % of the distance they travelled
    assert float(diff.median()) <= 0.05e-3 * n_steps, stats
    assert float(diff.max()) <= 2.2e-3 * n_steps, stats                  # nothing moved further than Adam can move it
    # Adam invariants of the table: a parameter that ever saw a gradient has both moments, and the moments agree with
    # the oracle's (torch.optim.Adam) wherever they are not summation noise.  (The first pipelined step once lost its
    # moments to a late zero-fill: v = 0 under m != 0 divides by eps on the next step.)
    st = og.state[net.encoder.params]
    m, v = st['exp_avg'].cpu(), st['exp_avg_sq'].cpu()
    assert int(((m != 0) & (v == 0)).sum()) == 0
    so = oo.state[ora.encoder_params]
    # (4 steps of Adam with eps = 1e-15 turn summation noise on near-zero gradients into +-lr parameter steps and the
    # fine samples then fall into other cells of the fine levels: entries are not comparable one by one, totals are)
    mo, vo = so['exp_avg'], so['exp_avg_sq']
    assert abs(float(m.abs().sum()) / float(mo.abs().sum()) - 1) < 5e-2
    assert abs(float(v.sum()) / float(vo.sum()) - 1) < 5e-2


def test_inferencer_chunking_and_thresh_depth():
    """inference.py:54-157: chunk size must not change the image; thresholded depth vs the oracle."""
    from nerf_amd.inference import Inferencer
    from nerf_amd.renderer import NerfRenderer
    from oracle import ref_cpu
    net, ora = _make_pair(NET_CFGS['small'])
    Rg = NerfRenderer(net, 2, 2.5, [1.6, 4.8], [64], 128, 0.2)
    n, h, w, K, E, bg, noise, gt = _batch(3, 150)
    c = lambda t: t.cuda()
    shape = (10, 15)
    args = [c(t).reshape(*shape, *t.shape[1:]) for t in (n, h, w, K, E)]
    img_a, inv_a = Inferencer(Rg, 37, 0, 0, 0.1, 0.05, 0.4, [1], 20).render_image(*args, noise=c(noise))
    img_b, inv_b = Inferencer(Rg, 4096, 0, 0, 0.1, 0.05, 0.4, [1], 20).render_image(*args, noise=c(noise))
    assert img_a.shape == (10, 15, 3) and inv_a.shape == (10, 15)
    assert (img_a == img_b).all() and (inv_a == inv_b).all()
    Ro = ref_cpu.OracleRenderer(ora, 2, 2.5, [1.6, 4.8], [64], 128, 0.2)
    with torch.no_grad():
        img_o, w_o, _, aux_o = Ro.render(n, h, w, K, E, torch.ones(3), noise=noise)
    torch.testing.assert_close(torch.from_numpy(img_a).reshape(-1, 3), img_o, rtol=0, atol=1e-4)
    # thresholded depth on identical weights (the first-crossing index is discrete: feed the oracle's weights)
    inf = Inferencer(Rg, 4096, 0, 0, 0.1, 0.05, 0.4, [1], 20)
    (d0, d1, d2), _ = inf._thresh_depths(c(w_o), c(aux_o['z_vals']))
    want = ref_cpu.invdepth_thresh(w_o, aux_o['z_vals'], 0.4, 0.05)
    got = 1 / d1
    got[torch.abs(d0 - d2) > 0.05] = 0
    got[d1 == 0] = 0
    torch.testing.assert_close(got.cpu(), want, rtol=1e-6, atol=1e-7)


def test_extract_dense_geometry_vs_oracle():
    """allign/extract_dense_geometry.py:28-101 on a tiny grid (voxel_res 12, mask_res 4)."""
    from nerf_amd.extract_dense_geometry import ExtractDenseGeometry
    from nerf_amd.loaders.procedural import sphere_cameras
    from oracle import ref_cpu
    net, ora = _make_pair(NET_CFGS['small'])
    K, E = sphere_cameras(7, 3)
    mask_o = ref_cpu.valid_positions(7, 800, 800, K, E, 4)
    ex = ExtractDenseGeometry(net, None, 4, 12, 500, 0.0, 2.5)
    mask_g = ex.get_valid_positions(7, 800, 800, K.cuda(), E.cuda(), 4)
    assert torch.equal(mask_g.cpu(), mask_o)
    full = torch.ones(4, 4, 4, dtype=torch.bool)            # exercise every voxel whatever the cameras see
    with torch.no_grad():
        vox = torch.linspace(-1 + 1 / 12, 1 - 1 / 12, 12)
        grid = torch.stack(torch.meshgrid(vox, vox, vox, indexing='ij'), -1).reshape(-1, 3)
        sig_all, _, _ = ora(grid, torch.tensor([0.0, 0.0, 1.0]).expand(grid.shape[0], 3), torch.zeros(grid.shape[0], dtype=torch.int64), 2.5)
    thresh = float(sig_all.median())
    ex.thresh = thresh
    pts_g, col_g = ex.extract_dense_geometry(7, 800, 800, K.cuda(), E.cuda(), mask=full.cuda())
    pts_o, col_o = ref_cpu.extract_dense_geometry(ora, full, 12, 500, thresh, 2.5)
    safe = (sig_all - thresh).abs() > 1e-4 * float(sig_all.abs().max())   # away from the threshold the sets must agree
    key = lambda p: list(map(tuple, (p * 1e4).round().to(torch.int64).tolist()))
    kg, ko, ks = key(pts_g), key(pts_o), set(key(grid[safe]))
    assert (set(kg) & ks) == (set(ko) & ks) and len(ko) > 0
    # colours of every point both sides kept, whatever the order and whatever happened next to the threshold
    ig, io = {k: i for i, k in enumerate(kg)}, {k: i for i, k in enumerate(ko)}
    common = sorted(set(kg) & set(ko))
    assert len(common) >= len(set(ko) & ks) > 100
    torch.testing.assert_close(col_g[[ig[k] for k in common]], col_o[[io[k] for k in common]], rtol=0, atol=1e-5)
    # order: ascending voxel index, as the reference's batch loop produces it
    assert [k for k in kg if k in ks] == [k for k in ko if k in ks]


def test_chunked_backward_matches_unchunked():
    """Batches of >= 187 k points split the backward into two ray chunks whose table scatters run on two streams
    (nets._FieldFn.backward); the trained parameters must equal the single-chunk path up to atomic ordering."""
    from nerf_amd import helpers
    from nerf_amd.optim import FusedAdam
    from nerf_amd.renderer import NerfRenderer

    def run(split):
        net, _ = _make_pair(NET_CFGS['small'])
        R = NerfRenderer(net, 2, 2.5, [1.6, 4.8], [64], 128, 0.2)
        opt = FusedAdam([{'params': list(net.encoder.parameters()) + [net.latent_emb], 'lr': 2e-2},
                         {'params': list(net.sigma_net.parameters()) + list(net.color_net.parameters()), 'lr': 1e-3}],
                        betas=(0.9, 0.99), eps=1e-15).attach(net)
        net.table_pipeline.early_fraction = split
        n, h, w, K, E, bg, noise, gt = [t.cuda() for t in _batch(2, 1600)]
        grads = None
        for it in range(2):
            opt.zero_grad()
            img, wts, zs, _ = R.render(n, h, w, K, E, bg, noise=noise)
            loss, _ = helpers.fused_train_loss(img, gt, wts, zs, 1e-3, 1e-1)
            loss.backward()
            used_chunks = bool(net.table_pipeline.early_events)
            if it == 0:
                torch.cuda.synchronize()        # both scatter streams
                grads = [p.grad.detach().clone() for p in (net.encoder.params, net.sigma_net.params, net.color_net.params,
                                                           net.latent_emb)]
            opt.step()
        torch.cuda.synchronize()
        return used_chunks, grads, [p.detach().clone() for p in (net.encoder.params, net.sigma_net.params,
                                                                  net.color_net.params, net.latent_emb)]
    c0, g_want, want = run(0.0)
    c1, g_got, got = run(0.35)
    assert not c0 and c1
    for g, w, name in zip(g_got, g_want, ('table', 'sigma_net', 'color_net', 'latent')):     # gradients: summation order only
        torch.testing.assert_close(g, w, rtol=1e-4, atol=1e-5 * float(w.abs().max()), msg=lambda m, n=name: n + ': ' + m)
    for g, w, name in zip(got, want, ('table', 'sigma_net', 'color_net', 'latent')):
        # Adam divides by sqrt(v): entries whose gradient is ~0 amplify summation-order noise to +-lr
        torch.testing.assert_close(g, w, rtol=0, atol=2e-3 * 40, msg=lambda m, n=name: n + ': ' + m)
        assert float(((g - w).abs() > 1e-4).float().mean()) < 0.02, name


def test_module_level_composition_matches_fused_field():
    """The tinycudann-level boundary: encoder / sigma_net / encoder_dir / color_net called as separate modules and
    glued with torch ops exactly as the reference's nets.py:109-146 does, forward and backward, against the fused
    field kernels (nerf_hashgrid_fwd/bwd, nerf_sh_fwd, nerf_mlp_fwd/bwd vs nerf_field_fwd/bwd)."""
    import torch.nn.functional as F
    net, _ = _make_pair(NET_CFGS['small'])
    g = torch.Generator().manual_seed(31)
    M, ob = 777, 2.5
    x = ((torch.rand(M, 3, generator=g) * 2 - 1) * 2.4).cuda()
    d = torch.randn(M, 3, generator=g)
    d = (d / d.norm(dim=-1, keepdim=True)).cuda()
    n = torch.randint(0, 7, (M,), generator=g).cuda()
    gs, gc = torch.randn(M, generator=g).cuda(), torch.randn(M, 3, generator=g).cuda()
    params = [net.encoder.params, net.sigma_net.params, net.color_net.params, net.latent_emb]

    def grads(fn):
        for p in params:
            p.grad = None
        sigma, color = fn()
        ((sigma * gs).sum() + (color * gc).sum()).backward()
        return sigma.detach(), color.detach(), [p.grad.detach().clone() for p in params]

    def modules():                                      # nets.py:109-146, line by line
        xx = (x + ob) / (2 * ob)
        x_hashtable = net.encoder(xx)
        sigma_out = net.sigma_net(x_hashtable)
        sigma = F.relu(sigma_out[..., 0])
        geo_feat = sigma_out[..., 1:]
        dd = net.encoder_dir((d + 1) / 2)
        color_in = torch.cat([geo_feat, dd, net.latent_emb[n]], dim=-1)
        color = torch.sigmoid(net.color_net(color_in))
        return sigma, color

    def fused():
        sigma, color, _ = net(x, d, n, ob)
        return sigma, color
    s_m, c_m, g_m = grads(modules)
    s_f, c_f, g_f = grads(fused)
    scale = float(s_f.abs().max()) + 1e-6
    torch.testing.assert_close(s_m, s_f, rtol=1e-4, atol=1e-5 * scale)
    torch.testing.assert_close(c_m, c_f, rtol=0, atol=1e-5)
    for a, b, name in zip(g_m, g_f, ('table', 'sigma_net', 'color_net', 'latent')):
        s = float(b.abs().max()) + 1e-12
        torch.testing.assert_close(a, b, rtol=1e-3, atol=2e-5 * s, msg=lambda m, w=name: w + ': ' + m)
    for p in params:
        p.grad = None


@pytest.mark.parametrize('name,M', [('small', 65536 + 37), ('f1', 70001), ('odd', 65536 + 5), ('f4', 66000)])
def test_feature_rows_of_large_batches(name, M):
    """From 64 K points on, the fused forward collects the x_hashtable rows of a 32-point tile in LDS and writes them as
    one contiguous block (16-byte pieces when L*F is a multiple of 4, single floats otherwise).  The rows must equal
    those of the direct-store path bit for bit, including the ragged last tile, for forward() and density()."""
    from nerf_amd.nets import NeRFNetwork
    kw = dict(encoding_precision='float32', encoding_n_features_per_level=2, geo_feat_dim=15, sigma_hidden_dim=64,
              sigma_num_layers=2, encoding_dir_precision='float32', encoding_dir_encoding='SphericalHarmonics',
              encoding_dir_degree=4, latent_embedding_dim=48, color_hidden_dim=64, color_num_layers=3)
    kw.update(NET_CFGS[name])
    net = NeRFNetwork(5, seed=1, **kw)
    g = torch.Generator().manual_seed(9)
    with torch.no_grad():
        net.encoder.params.copy_((torch.rand(net.encoder.params.shape, generator=g) - 0.5).cuda())
        x = ((torch.rand(M, 3, generator=g) * 2 - 1) * 2.4).cuda()
        d = torch.nn.functional.normalize(torch.randn(M, 3, generator=g), dim=-1).cuda()
        n = torch.randint(0, 5, (M,), generator=g).cuda()
        sigma, rgb, aux = net(x, d, n, 2.5)
        # batches below 64 K points take the direct-store path: the same rows, bit for bit, whole batch covered
        parts = [net(x[a:a + 30000], d[a:a + 30000], n[a:a + 30000], 2.5) for a in range(0, M, 30000)]
        assert torch.equal(aux['x_hashtable'], torch.cat([p[2]['x_hashtable'] for p in parts]))
        assert torch.equal(sigma, torch.cat([p[0] for p in parts]))
        assert torch.equal(rgb, torch.cat([p[1] for p in parts]))
        sigma_d, aux_d = net.density(x, 2.5)
        assert torch.equal(aux_d['x_hashtable'], aux['x_hashtable'])
        assert torch.equal(sigma_d, sigma)
        # and the stand-alone encoder's (torch divides by the constant as a multiplication: last-bit input differences)
        want = net.encoder((x + 2.5) / 5.0)
        torch.testing.assert_close(aux['x_hashtable'], want, rtol=0, atol=2e-4)
