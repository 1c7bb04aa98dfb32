"""Guards around the hot path: image indices outside the latent table (the reference's latent_emb[n] raises,
nets.py:132), the non-finite-loss guard that stands in for GradScaler (trainer.py:134-136), and the order of
zero_grad / backward / step calls around a discarded backward."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _pair(name='small'):
    from test_gpu_parity import NET_CFGS, _make_pair
    return _make_pair(NET_CFGS[name])


def test_image_index_outside_the_latent_table_is_memory_safe():
    from nerf_amd._lib import check, lib, ptr, stream
    from nerf_amd.nets import make_point_src
    net, _ = _pair()
    N_img, lat = net.N, net.latent_emb_dim
    g = torch.Generator().manual_seed(2)
    M = 300
    x = ((torch.rand(M, 3, generator=g) * 2 - 1) * 2.4).cuda()
    d = torch.nn.functional.normalize(torch.randn(M, 3, generator=g), dim=-1).cuda()
    n = torch.randint(0, N_img, (M,), generator=g).cuda()
    bad = torch.zeros(M, dtype=torch.bool, device='cuda')
    bad[::7] = True
    n_bad = n.clone()
    n_bad[bad] = torch.where(torch.arange(M, device='cuda')[bad] % 2 == 0, N_img, -1)      # one past the end / before the start
    with torch.no_grad():
        net.latent_emb[0].zero_()
        n_ref = n.clone()
        n_ref[bad] = 0                                  # row 0 is all zero: what an out-of-range index must read
        s_bad, c_bad, aux = net(x, d, n_bad, 2.5)
        s_ref, c_ref, _ = net(x, d, n_ref, 2.5)
    assert torch.equal(s_bad, s_ref) and torch.equal(c_bad, c_ref) and bool(torch.isfinite(c_bad).all())
    # backward through the C ABI with the latent gradient inside a guard-padded buffer
    pad = 4096
    big = torch.zeros(2 * pad + N_img * lat, device='cuda')
    dlat = big[pad:pad + N_img * lat]
    cfg, ncfg = net.encoder.cfg, net.net_cfg
    xh = aux['x_hashtable'].contiguous()
    dsig = torch.randn(M, generator=g).cuda()
    drgb = torch.randn(M, 3, generator=g).cuda()

    def bwd(n_idx, dl):
        src = make_point_src(2.5, x=x, d=d, n=n_idx, n_points=M)
        dfeat = torch.empty_like(xh)
        dsw, dcw = torch.zeros_like(net.sigma_net.params), torch.zeros_like(net.color_net.params)
        check(lib().nerf_field_bwd(C.byref(cfg), C.byref(ncfg), None, ptr(net.sigma_net.params), ptr(net.color_net.params),
                                   ptr(net.latent_emb.detach()), N_img, C.byref(src), ptr(xh), ptr(dsig), ptr(drgb), ptr(dfeat),
                                   None, ptr(dsw), ptr(dcw), ptr(dl), stream()), 'nerf_field_bwd')
        torch.cuda.synchronize()
        return dfeat, dsw, dcw
    f_bad, sw_bad, cw_bad = bwd(n_bad, dlat)
    assert float(big[:pad].abs().max()) == 0.0 and float(big[pad + N_img * lat:].abs().max()) == 0.0
    ref_l = torch.zeros(N_img * lat, device='cuda')
    f_ref, sw_ref, cw_ref = bwd(n_ref, ref_l)
    assert torch.equal(f_bad, f_ref)                      # per-point rows: identical arithmetic
    torch.testing.assert_close(sw_bad, sw_ref, rtol=1e-4, atol=1e-5 * float(sw_ref.abs().max()))
    # the out-of-range points add nothing anywhere: rows 1.. equal the reference run's, row 0 lacks their share
    got, want = dlat.view(N_img, lat), ref_l.view(N_img, lat)
    torch.testing.assert_close(got[1:], want[1:], rtol=1e-4, atol=1e-5 * float(want.abs().max()))
    assert float((got[0] - want[0]).abs().max()) > 0


def _opt(net):
    from nerf_amd.optim import FusedAdam
    return FusedAdam([{'params': list(net.encoder.parameters()), 'lr': 2e-2}, {'params': [net.latent_emb], 'lr': 2e-2},
                      {'params': list(net.sigma_net.parameters()) + list(net.color_net.parameters()), 'weight_decay': 1e-6,
                       'lr': 1e-3}], betas=(0.9, 0.99), eps=1e-15).attach(net)


def _step(net, R, opt, seed, poison=False):
    from test_gpu_parity import _batch
    from nerf_amd import helpers
    n, h, w, K, E, bg, noise, gt = [t.cuda() for t in _batch(seed, 200)]
    if poison:
        gt = gt.clone()
        gt[17, 1] = float('nan')
    opt.zero_grad()
    img, wts, zs, _ = R.render(n, h, w, K, E, bg, noise=noise)
    loss, _ = helpers.fused_train_loss(img, gt, wts, zs, 1e-3, 1e-1)
    loss.backward()
    opt.skip_flag = loss.nonfinite_flag
    opt.step()
    return loss


@pytest.mark.parametrize('name', ['small', 'base19'])
def test_nonfinite_loss_drops_the_step(name):
    """A NaN in the batch makes the loss NaN: the guarded step must leave parameters and Adam moments exactly as they were
    (decided on the device), clear the gradients, and training continues from there.  'base19' takes the binned table
    update, 'small' the per-level scatter -> Adam pipeline."""
    from nerf_amd.renderer import NerfRenderer
    net, _ = _pair(name)
    R = NerfRenderer(net, 2, 2.5, [1.6, 4.8], [64], 128, 0.2)
    opt = _opt(net)
    assert bool(net.table_pipeline.binned_levels) == (name == 'base19')
    _step(net, R, opt, 1)
    torch.cuda.synchronize()
    params = (net.encoder.params, net.sigma_net.params, net.color_net.params, net.latent_emb)
    snap = [p.detach().clone() for p in params]
    mom = [(opt.state[p]['exp_avg'].clone(), opt.state[p]['exp_avg_sq'].clone()) for p in params]
    steps_before = [opt.state[p]['step'] for p in params]
    loss = _step(net, R, opt, 2, poison=True)
    torch.cuda.synchronize()
    assert not torch.isfinite(loss) and float(loss.nonfinite_flag) != 0
    for p, s, (m, v) in zip(params, snap, mom):
        assert torch.equal(p.detach(), s) and torch.equal(opt.state[p]['exp_avg'], m) and torch.equal(opt.state[p]['exp_avg_sq'], v)
        assert float(p.grad.abs().max()) == 0.0         # NaN gradients are gone as well
    opt._collect_flags(wait=True)
    assert opt.skipped_steps == 1 and [opt.state[p]['step'] for p in params] == steps_before
    loss3 = _step(net, R, opt, 3)
    torch.cuda.synchronize()
    assert torch.isfinite(loss3) and all(bool(torch.isfinite(p).all()) for p in params)
    assert not torch.equal(net.sigma_net.params.detach(), snap[1])
    opt._collect_flags(wait=True)
    assert opt.skipped_steps == 1


@pytest.mark.parametrize('name', ['small', 'base19'])
def test_discarded_backward_leaves_no_trace(name):
    """Call order step -> backward -> zero_grad (discard) -> backward -> step: the discarded pass must not reach the
    parameters -- neither through p.grad nor through the binned corner updates that were waiting for step()."""
    from test_gpu_parity import _batch
    from nerf_amd import helpers
    from nerf_amd.renderer import NerfRenderer

    def run(discard):
        net, _ = _pair(name)
        R = NerfRenderer(net, 2, 2.5, [1.6, 4.8], [64], 128, 0.2)
        opt = _opt(net)
        _step(net, R, opt, 1)
        if discard:
            n, h, w, K, E, bg, noise, gt = [t.cuda() for t in _batch(9, 200)]
            img, wts, zs, _ = R.render(n, h, w, K, E, bg, noise=noise)      # NO zero_grad before: step() cleared everything
            loss, _ = helpers.fused_train_loss(img, gt, wts, zs, 1e-3, 1e-1)
            loss.backward()
            opt.zero_grad()                                                  # throw that pass away
            torch.cuda.synchronize()
            assert net.table_pipeline.binned_pending is None and not net.table_pipeline.pending
            for p in (net.encoder.params, net.sigma_net.params, net.color_net.params, net.latent_emb):
                assert float(p.grad.abs().max()) == 0.0
        _step(net, R, opt, 2)
        torch.cuda.synchronize()
        return [p.detach().clone() for p in (net.encoder.params, net.sigma_net.params, net.color_net.params, net.latent_emb)]
    a, b = run(False), run(True)
    for x, y, nm in zip(a, b, ('table', 'sigma_net', 'color_net', 'latent')):
        # (two runs differ only in the order of float atomics; Adam's 1/sqrt(v) turns that into steps of at most ~lr on
        # entries whose gradient is noise)
        torch.testing.assert_close(x, y, rtol=0, atol=2e-3 * 40, msg=lambda m, n=nm: n + ': ' + m)
        assert float(((x - y).abs() > 1e-4).float().mean()) < 0.02, nm
