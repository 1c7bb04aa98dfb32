"""Guards around the hot path: image indices outside the latent table (the reference's latent_emb[n] raises,
nets.py:132), the non-finite guard that stands in for GradScaler (trainer.py:134-136: a non-finite loss, non-finite
gradients of a finite loss, a single non-finite gradient entry), and the order of zero_grad / backward / step calls around
a discarded backward."""
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


@pytest.mark.parametrize('name', ['small', 'base19'])
def test_finite_loss_with_nonfinite_gradients_drops_the_step(name):
    """GradScaler.step skips on non-finite GRADIENTS (trainer.py:134-136), not on the loss: a finite loss whose backward
    overflows (here: an Inf upstream gradient handed to backward()) must drop the step as well.  The fused tail
    (Trainer.train_step's route) raises the step's flag from nerf_composite_loss_bwd when d(sigma) / d(rgb) are not finite."""
    from test_gpu_parity import _batch
    from nerf_amd.renderer import NerfRenderer
    net, _ = _pair(name)
    R = NerfRenderer(net, 2, 2.5, [1.6, 4.8], [64], 128, 0.2)
    opt = _opt(net)
    _step(net, R, opt, 1)
    torch.cuda.synchronize()
    params = (net.encoder.params, net.sigma_net.params, net.color_net.params, net.latent_emb)
    snap = [p.detach().clone() for p in params]
    mom = [(opt.state[p]['exp_avg'].clone(), opt.state[p]['exp_avg_sq'].clone()) for p in params]
    steps_before = [opt.state[p]['step'] for p in params]
    n, h, w, K, E, bg, noise, gt = [t.cuda() for t in _batch(2, 200)]
    opt.zero_grad()
    loss = R.render_loss(n, h, w, K, E, bg, gt, 1e-3, 1e-1, noise=noise)[0]
    assert torch.isfinite(loss) and float(loss.nonfinite_flag) == 0
    opt.skip_flag = loss.nonfinite_flag
    loss.backward(torch.tensor(float('inf'), device='cuda'))
    assert float(loss.nonfinite_flag) != 0                 # raised by the backward kernel, on the device
    opt.step()
    torch.cuda.synchronize()
    for p, s_, (m, v) in zip(params, snap, mom):
        assert torch.equal(p.detach(), s_) and torch.equal(opt.state[p]['exp_avg'], m) and torch.equal(opt.state[p]['exp_avg_sq'], v)
        assert float(p.grad.abs().max()) == 0.0
    # a checkpoint written right now must not count the dropped step (state_dict reads the pending flags first)
    sd = opt.state_dict()
    assert [st['step'] for st in sd['state'].values()] == steps_before and opt.skipped_steps == 1
    loss3 = _step(net, R, opt, 3)
    torch.cuda.synchronize()
    assert torch.isfinite(loss3) and all(bool(torch.isfinite(p).all()) for p in params)
    assert not torch.equal(net.sigma_net.params.detach(), snap[1])


def test_nonfinite_gradient_entry_keeps_its_state():
    """Under the step-level flag every Adam kernel leaves an ENTRY whose own gradient is NaN / Inf exactly as it was
    (parameter and both moments; adam_common.h) and clears the gradient; all other entries take torch.optim.Adam's step."""
    from nerf_amd.optim import FusedAdam
    g = torch.Generator().manual_seed(9)
    p0 = torch.randn(5003, generator=g)
    pc, pg = torch.nn.Parameter(p0.clone()), torch.nn.Parameter(p0.clone().cuda())
    oc = torch.optim.Adam([{'params': [pc], 'lr': 2e-2, 'weight_decay': 1e-6}], betas=(0.9, 0.99), eps=1e-15)
    og = FusedAdam([{'params': [pg], 'lr': 2e-2, 'weight_decay': 1e-6}], betas=(0.9, 0.99), eps=1e-15)
    bad = torch.tensor([3, 1000, 4097, 5002])
    for it in range(3):
        gr = torch.randn(5003, generator=g) * 0.1
        before = (pg.detach().clone(), None, None)
        if it == 1:
            before = (pg.detach().clone(), og.state[pg]['exp_avg'].clone(), og.state[pg]['exp_avg_sq'].clone())
            ref_state = (pc.detach().clone(), oc.state[pc]['exp_avg'].clone(), oc.state[pc]['exp_avg_sq'].clone())
        pc.grad = gr.clone()
        gg = gr.clone()
        if it == 1:
            gg[bad] = torch.tensor([float('inf'), float('nan'), float('-inf'), float('nan')])
        pg.grad = gg.cuda()
        oc.step()
        og.step()
        torch.cuda.synchronize()
        assert float(pg.grad.abs().nan_to_num(nan=1.0).max()) == 0.0
        if it == 1:
            st = og.state[pg]
            assert torch.equal(pg.detach()[bad.cuda()], before[0][bad.cuda()])
            assert torch.equal(st['exp_avg'][bad.cuda()], before[1][bad.cuda()])
            assert torch.equal(st['exp_avg_sq'][bad.cuda()], before[2][bad.cuda()])
            with torch.no_grad():       # give the CPU reference the same "nothing happened" for those entries
                pc[bad] = ref_state[0][bad]
                oc.state[pc]['exp_avg'][bad] = ref_state[1][bad]
                oc.state[pc]['exp_avg_sq'][bad] = ref_state[2][bad]
        assert bool(torch.isfinite(pg).all())
        torch.testing.assert_close(pg.detach().cpu(), pc.detach(), rtol=1e-5, atol=1e-6)


def test_nonfinite_corner_update_keeps_the_table_entries_it_touches():
    """The same safety net in the binned table update: an Inf row of d(features) poisons only the sums of the table
    entries its corners touch; those keep p, m, v, every other entry equals the run without the bad row."""
    import ctypes as C
    from nerf_amd.nets import make_point_src
    from test_gpu_level_sharding import _rays
    from test_gpu_table_update import _grid, _run_binned
    F, levels, R_, S = 2, [14, 15, 10], 512, 16
    enc = _grid(16, F, 19)
    ro, rd, z, _ = _rays(R_, S, 5)
    M = R_ * S
    src = make_point_src(outer_bound=2.5, inner_bound=2.0, rays_o=ro, rays_d=rd, z=z, z_stride=S, n_samples=S, n_points=M)
    dfeat = torch.randn(M, len(levels) * F, device='cuda') * 1e-3
    hyper = (1e-2, 0.9, 0.99, 1e-15, 0.0, 1.0)
    p_ok, m_ok, v_ok = _run_binned(enc, src, levels, dfeat, len(levels) * F, False, 2, hyper)
    poisoned = dfeat.clone()
    poisoned[777] = float('inf')
    p_b, m_b, v_b = _run_binned(enc, src, levels, poisoned, len(levels) * F, False, 2, hyper)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(p_b).all()) and bool(torch.isfinite(m_b).all()) and bool(torch.isfinite(v_b).all())
    cfg = enc.cfg
    in_set = torch.zeros_like(p_b, dtype=torch.bool)
    for l in levels:
        a = int(cfg.offset[l]) * F
        in_set[a:a + int(cfg.entries[l]) * F] = True
    # every entry of the listed levels takes an Adam step (m decays even under a zero gradient): the ones that did not
    # are the entries the bad point's corners touch
    kept = in_set & (m_b == 0.01)
    assert 0 < int(kept.sum()) <= 8 * F * len(levels)
    p_init = enc.params.detach()
    assert torch.equal(p_b[kept], p_init[kept]) and torch.equal(v_b[kept], torch.full_like(v_b[kept], 0.002))
    assert int((in_set & (m_ok == 0.01)).sum()) == 0
    rest = in_set & ~kept           # (two runs sum a bin's items in different orders: fp32 noise of the sum, as in test_gpu_table_update)
    torch.testing.assert_close(p_b[rest], p_ok[rest], rtol=0, atol=1e-5 * hyper[0])
    torch.testing.assert_close(m_b[rest], m_ok[rest], rtol=0, atol=1e-8)
