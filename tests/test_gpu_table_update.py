"""Binned table update (nerf_table_bin_levels + nerf_table_update_binned) against the pair it replaces: the atomic
scatter of d(features) into a zeroed gradient (nerf_hashgrid_scatter_levels) followed by nerf_adam_step on the levels'
slices.  Same arithmetic per parameter; the corner contributions are summed in a different order, so the bar is the
fp32 noise of that sum (relative to the largest gradient), carried through one Adam step."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

from test_gpu_level_sharding import _rays  # noqa: E402


def _grid(levels, F, log2_T):
    from nerf_amd.nets import HashGridEncoding
    return HashGridEncoding(levels, F, log2_T, device='cuda', seed=3)


def _level_set(levels):
    from nerf_amd._lib import NerfLevelSet
    ls = NerfLevelSet()
    ls.count = len(levels)
    for i, l in enumerate(levels):
        ls.level[i] = l
    return ls


def _run_pair(enc, src, levels, dfeat, stride, by_level, step, hyper):
    """Reference: scatter + Adam per level on clones.  Returns (p, m, v)."""
    from nerf_amd._lib import check, lib, ptr, stream
    cfg, F = enc.cfg, enc.cfg.n_features
    p = enc.params.detach().clone()
    g = torch.zeros_like(p)
    m = torch.full_like(p, 0.01)
    v = torch.full_like(p, 0.002)
    cols = torch.cat([dfeat[:, (l if by_level else i) * F:(l if by_level else i) * F + F] for i, l in enumerate(levels)], 1).contiguous()
    ls = _level_set(levels)
    check(lib().nerf_hashgrid_scatter_levels(C.byref(cfg), C.byref(src), C.byref(ls), ptr(cols), cols.shape[1], ptr(g),
                                             stream()), 'scatter_levels')
    g_kept = g.clone()                       # the Adam pass below clears g
    for l in levels:
        a = int(cfg.offset[l]) * F
        b = a + int(cfg.entries[l]) * F
        check(lib().nerf_adam_step(ptr(p[a:b]), ptr(g[a:b]), ptr(m[a:b]), ptr(v[a:b]), b - a, *hyper[:5], step, hyper[5], 1,
                                   stream()), 'adam')
    return p, m, v, g_kept


def _run_binned(enc, src, levels, dfeat, stride, by_level, step, hyper):
    from nerf_amd._lib import check, lib, ptr, stream
    cfg = enc.cfg
    p = enc.params.detach().clone()
    m = torch.full_like(p, 0.01)
    v = torch.full_like(p, 0.002)
    ls = _level_set(levels)
    nbytes = lib().nerf_table_bin_workspace_bytes(C.byref(cfg), C.byref(ls), src.n_points)
    assert nbytes > 0
    ws = torch.empty(nbytes, dtype=torch.uint8, device='cuda')
    cols = (C.c_int32 * len(levels))(*[l * cfg.n_features for l in levels]) if by_level else None
    check(lib().nerf_table_bin_levels(C.byref(cfg), C.byref(src), C.byref(ls), ptr(dfeat), stride, cols, ptr(ws),
                                      nbytes, stream()), 'bin_levels')
    check(lib().nerf_table_update_binned(C.byref(cfg), C.byref(ls), src.n_points, ptr(ws), nbytes, ptr(p), ptr(m), ptr(v),
                                         *hyper[:5], step, hyper[5], stream()), 'update_binned')
    return p, m, v


@pytest.mark.parametrize('F,log2_T,levels,R,S,by_level', [
    (2, 16, [15, 3, 9, 12], 301, 17, True),        # hashed + dense, ragged point count, columns by level number
    (2, 19, [14, 15, 10, 6], 2048, 24, False),     # several workgroup chunks (49152 points), columns by set position
    (4, 15, [11, 5], 97, 9, False),
    (1, 17, [15, 13, 8], 640, 8, True),
])
def test_binned_update_matches_scatter_plus_adam(F, log2_T, levels, R, S, by_level):
    from nerf_amd.nets import make_point_src
    enc = _grid(16, F, log2_T)
    cfg = enc.cfg
    ro, rd, z, _ = _rays(R, S, 11)
    M = R * S
    src = make_point_src(outer_bound=2.5, inner_bound=2.0, rays_o=ro, rays_d=rd, z=z, z_stride=S, n_samples=S, n_points=M)
    stride = 16 * F if by_level else len(levels) * F
    dfeat = torch.randn(M, stride, device='cuda') * 1e-3
    hyper = (1e-2, 0.9, 0.99, 1e-15, 0.0, 0.5)        # lr, beta1, beta2, eps, weight_decay, grad_scale
    p0, m0, v0, g0 = _run_pair(enc, src, levels, dfeat, stride, by_level, 3, hyper)
    p1, m1, v1 = _run_binned(enc, src, levels, dfeat, stride, by_level, 3, hyper)
    gmax = float(g0.abs().max())
    assert gmax > 0
    touched = torch.zeros_like(p0, dtype=torch.bool)
    for l in levels:
        a = int(cfg.offset[l]) * F
        touched[a:a + int(cfg.entries[l]) * F] = True
    # levels outside the set are untouched by the binned update
    assert torch.equal(p1[~touched], enc.params.detach()[~touched])
    assert torch.equal(m1[~touched], torch.full_like(m1[~touched], 0.01))
    # exp_avg is linear in the gradient: error = (1-beta1) * grad_scale * (summation noise of the gradient)
    torch.testing.assert_close(m1[touched], m0[touched], rtol=0, atol=2e-6 * gmax)
    torch.testing.assert_close(v1[touched], v0[touched], rtol=1e-4, atol=1e-6 * gmax * gmax)
    torch.testing.assert_close(p1[touched], p0[touched], rtol=0, atol=1e-5 * hyper[0])
    # and something did change
    assert not torch.equal(p1[touched], enc.params.detach()[touched])


def test_binned_update_rejects_bad_arguments():
    from nerf_amd._lib import lib, ptr, stream
    from nerf_amd.nets import make_point_src
    enc = _grid(16, 2, 15)
    cfg = enc.cfg
    x = torch.rand(64, 3, device='cuda')
    src = make_point_src(outer_bound=1.0, x=x, n_points=64)
    ls = _level_set([3, 3])
    assert lib().nerf_table_bin_workspace_bytes(C.byref(cfg), C.byref(ls), 64) < 0           # level listed twice
    ls = _level_set([3, 40])
    assert lib().nerf_table_bin_workspace_bytes(C.byref(cfg), C.byref(ls), 64) < 0           # level out of range
    ls = _level_set([3, 9])
    n = lib().nerf_table_bin_workspace_bytes(C.byref(cfg), C.byref(ls), 64)
    assert n > 0
    ws = torch.empty(n, dtype=torch.uint8, device='cuda')
    dfeat = torch.zeros(64, 4, device='cuda')
    assert lib().nerf_table_bin_levels(C.byref(cfg), C.byref(src), C.byref(ls), ptr(dfeat), 4, None, ptr(ws), n - 1, stream()) != 0
    assert lib().nerf_table_bin_levels(C.byref(cfg), C.byref(src), C.byref(ls), ptr(dfeat), 3, None, ptr(ws), n, stream()) != 0
    p = enc.params.detach().clone()
    assert lib().nerf_table_update_binned(C.byref(cfg), C.byref(ls), 64, ptr(ws), n, ptr(p), ptr(p), None, 1e-2, 0.9, 0.99,
                                          1e-15, 0.0, 1, 1.0, stream()) != 0
    assert lib().nerf_table_update_binned(C.byref(cfg), C.byref(ls), 64, ptr(ws), n, ptr(p), ptr(p), ptr(p), 1e-2, 0.9, 0.99,
                                          1e-15, 0.0, 0, 1.0, stream()) != 0               # step must be >= 1


def _accum_run(binned, monkeypatch, n_backward, discard_first=False):
    """`n_backward` backward passes on different batches, then one optimizer step, in the given table mode."""
    from nerf_amd import helpers
    from nerf_amd.optim import FusedAdam
    from nerf_amd.renderer import NerfRenderer
    from test_gpu_parity import NET_CFGS, _batch, _make_pair
    monkeypatch.setenv('NERF_TABLE_BINNED', binned)
    net, _ = _make_pair(NET_CFGS['base19'])
    R = NerfRenderer(net, 2, 2.5, [1.6, 4.8], [64], 128, 0.2)
    opt = FusedAdam([{'params': list(net.encoder.parameters()) + [net.latent_emb], 'lr': 2e-2},
                     {'params': list(net.sigma_net.parameters()) + list(net.color_net.parameters()), 'lr': 1e-3}],
                    betas=(0.9, 0.99), eps=1e-8).attach(net)
    assert bool(net.table_pipeline.binned_levels) == (binned == '1')
    for step in range(2):
        opt.zero_grad()
        if discard_first and step == 0:
            # a backward whose gradients are thrown away: zero_grad() without step() must drop the binned updates too
            n, h, w, K, E, bg, noise, gt = [t.cuda() for t in _batch(40, 64)]
            img, wts, zs, _ = R.render(n, h, w, K, E, bg, noise=noise)
            helpers.fused_train_loss(img, gt, wts, zs, 1e-3, 1e-1)[0].backward()
            opt._grads_cleared = False
            opt.zero_grad()
        for j in range(n_backward):
            n, h, w, K, E, bg, noise, gt = [t.cuda() for t in _batch(20 + 3 * step + j, 64)]
            img, wts, zs, _ = R.render(n, h, w, K, E, bg, noise=noise)
            loss, _ = helpers.fused_train_loss(img, gt, wts, zs, 1e-3, 1e-1)
            loss.backward()
        opt.step()
    torch.cuda.synchronize()
    st = opt.state[net.encoder.params]
    return net.encoder.params.detach().clone(), st['exp_avg'].clone(), st['exp_avg_sq'].clone()


@pytest.mark.parametrize('n_backward,discard_first', [(2, False), (1, True)])
def test_binned_mode_handles_gradient_accumulation_and_discarded_backward(n_backward, discard_first, monkeypatch):
    """Two backward passes before one step (the binned mode falls back to the atomic scatter into param.grad), and a
    backward that is discarded by zero_grad(): both must give what the scatter -> Adam mode gives.  eps = 1e-8 keeps
    Adam from amplifying summation noise, so the tables are comparable entry by entry."""
    p1, m1, v1 = _accum_run('1', monkeypatch, n_backward, discard_first)
    p0, m0, v0 = _accum_run('0', monkeypatch, n_backward, discard_first)
    # the second step's samples depend on the first step's parameters to the last bit, so a few hundred of the 16 M
    # entries see other corner updates; a lost or doubled backward pass would be off by ~50 % everywhere
    assert float(m0.abs().max()) > 0
    for a, b in ((m1, m0), (v1, v0)):
        assert float((a - b).norm() / b.norm()) < 1e-2
        assert float(((a - b).abs() > 1e-4 * float(b.abs().max())).float().mean()) < 1e-3
    assert float(((p1 - p0).abs() > 1e-5).float().mean()) < 1e-3


def _early_run(early, monkeypatch, interleave=False):
    from nerf_amd import helpers, nets
    from nerf_amd.optim import FusedAdam
    from nerf_amd.renderer import NerfRenderer
    from test_gpu_parity import NET_CFGS, _batch, _make_pair
    monkeypatch.setattr(nets, 'EARLY_BIN_COUNT', early)
    net, _ = _make_pair(NET_CFGS['base19'])
    R = NerfRenderer(net, 2, 2.5, [1.6, 4.8], [64], 128, 0.2)
    opt = FusedAdam([{'params': list(net.encoder.parameters()) + [net.latent_emb], 'lr': 2e-2},
                     {'params': list(net.sigma_net.parameters()) + list(net.color_net.parameters()), 'lr': 1e-3}],
                    betas=(0.9, 0.99), eps=1e-8).attach(net)
    used = 0
    for step in range(2):
        opt.zero_grad()
        n, h, w, K, E, bg, noise, gt = [t.cuda() for t in _batch(30 + step, 160)]
        img, wts, zs, _ = R.render(n, h, w, K, E, bg, noise=noise)
        if interleave and step == 1:
            # a second forward (other rays, never back-propagated) between this forward and its backward: its early count
            # pass overwrites the counters, so the backward must count again
            n2, h2, w2, K2, E2, bg2, noise2, _ = [t.cuda() for t in _batch(77, 96)]
            R.render(n2, h2, w2, K2, E2, bg2, noise=noise2)
        loss, _ = helpers.fused_train_loss(img, gt, wts, zs, 1e-3, 1e-1)
        loss.backward()
        used += int(net.table_pipeline.binned_pending is not None)
        opt.step()
    torch.cuda.synchronize()
    assert used == 2        # the binned path, not the accumulation fallback
    st = opt.state[net.encoder.params]
    return net.encoder.params.detach().clone(), st['exp_avg'].clone(), st['exp_avg_sq'].clone()


@pytest.mark.parametrize('interleave', [False, True])
def test_early_count_pass_changes_nothing(interleave, monkeypatch):
    """The count + scan passes of the binned update issued from the forward pass on a side stream (TablePipeline.count_early)
    against the same passes issued in the backward: the same counters, so the same bins (the order of the items inside a
    bin depends on the LDS atomics of the fill pass in either case, hence the comparison up to summation order, with the
    bars of the accumulation test above: a lost, doubled or misplaced run of items would be off by tens of percent).
    With a foreign forward between a forward and its backward the early counts are stale and must not be used.
    (Run-to-run spread of either mode on one box, tools/early_count_probe.py: moments 2e-4 ... 1e-3 in relative L2.)"""
    p1, m1, v1 = _early_run(True, monkeypatch, interleave)
    p0, m0, v0 = _early_run(False, monkeypatch, interleave)
    assert float(m0.abs().max()) > 0
    for a, b in ((m1, m0), (v1, v0)):
        assert float((a - b).norm() / b.norm()) < 1e-2
        assert float(((a - b).abs() > 1e-4 * float(b.abs().max())).float().mean()) < 2e-3
    assert float(((p1 - p0).abs() > 1e-5).float().mean()) < 1e-2


def test_ranged_fill_equals_one_fill():
    """nerf_table_bin_fill_range over [0, 2 chunks), [2 chunks, M) (M not a multiple of the chunk) after one count pass
    against nerf_table_bin_levels: the same Adam step up to the summation order inside a bin; ranges that are not whole
    chunks are refused."""
    from nerf_amd._lib import check, lib, ptr, stream
    from nerf_amd.nets import make_point_src
    enc = _grid(16, 2, 19)
    cfg = enc.cfg
    levels = [15, 12, 9]
    ls = _level_set(levels)
    M = 3 * 16384 + 777
    g = torch.Generator(device='cuda').manual_seed(5)
    x = torch.rand(M, 3, device='cuda', generator=g) * 2 - 1
    src = make_point_src(outer_bound=1.0, x=x, n_points=M)
    dfeat = torch.randn(M, 8, device='cuda', generator=g) * 1e-3
    cols = (C.c_int32 * 3)(0, 2, 4)
    n = int(lib().nerf_table_bin_workspace_bytes(C.byref(cfg), C.byref(ls), M))

    def step(ranged):
        ws = torch.zeros(n, dtype=torch.uint8, device='cuda')
        if ranged:
            check(lib().nerf_table_bin_count(C.byref(cfg), C.byref(src), C.byref(ls), ptr(ws), n, stream()), 'count')
            for a, b in ((2 * 16384, M), (0, 2 * 16384)):       # any order
                check(lib().nerf_table_bin_fill_range(C.byref(cfg), C.byref(src), C.byref(ls), ptr(dfeat), 8, cols, ptr(ws), n,
                                                      a, b, stream()), 'fill_range')
        else:
            check(lib().nerf_table_bin_levels(C.byref(cfg), C.byref(src), C.byref(ls), ptr(dfeat), 8, cols, ptr(ws), n,
                                              stream()), 'bin')
        p = enc.params.detach().clone()
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        check(lib().nerf_table_update_binned(C.byref(cfg), C.byref(ls), M, ptr(ws), n, ptr(p), ptr(m), ptr(v), 1e-2, 0.9, 0.99,
                                             1e-8, 0.0, 1, 1.0, stream()), 'update')
        torch.cuda.synchronize()
        return p, m, v

    p1, m1, v1 = step(True)
    p0, m0, v0 = step(False)
    assert float(m0.abs().max()) > 0
    assert float((m1 - m0).norm() / m0.norm()) < 1e-5
    assert float((v1 - v0).norm() / v0.norm()) < 1e-5
    assert float((p1 - p0).abs().max()) <= 2e-2 and float(((p1 - p0).abs() > 1e-6).float().mean()) < 1e-3
    ws = torch.zeros(n, dtype=torch.uint8, device='cuda')
    assert lib().nerf_table_bin_fill_range(C.byref(cfg), C.byref(src), C.byref(ls), ptr(dfeat), 8, cols, ptr(ws), n, 100, M,
                                           stream()) != 0
    assert lib().nerf_table_bin_fill_range(C.byref(cfg), C.byref(src), C.byref(ls), ptr(dfeat), 8, cols, ptr(ws), n, 0, 16384 + 5,
                                           stream()) != 0


def test_update_on_a_foreign_workspace_stays_in_bounds():
    """nerf_table_update_binned on a workspace that nerf_table_bin_levels never filled (random bytes): the result is
    garbage, but every read and write stays inside the workspace and the listed levels (no GPU fault, levels outside
    the set untouched)."""
    from nerf_amd._lib import check, lib, ptr, stream
    enc = _grid(16, 2, 16)
    cfg = enc.cfg
    levels = [15, 9]
    ls = _level_set(levels)
    M = 3000
    n = lib().nerf_table_bin_workspace_bytes(C.byref(cfg), C.byref(ls), M)
    ws = torch.randint(0, 256, (n,), dtype=torch.uint8, device='cuda', generator=torch.Generator(device='cuda').manual_seed(1))
    p = enc.params.detach().clone()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    check(lib().nerf_table_update_binned(C.byref(cfg), C.byref(ls), M, ptr(ws), n, ptr(p), ptr(m), ptr(v), 1e-2, 0.9, 0.99,
                                         1e-8, 0.0, 1, 1.0, stream()), 'update')
    torch.cuda.synchronize()
    untouched = torch.ones_like(p, dtype=torch.bool)
    for l in levels:
        a = int(cfg.offset[l]) * 2
        untouched[a:a + int(cfg.entries[l]) * 2] = False
    assert torch.equal(p[untouched], enc.params.detach()[untouched])
    assert float(m[untouched].abs().max()) == 0


def test_binned_update_with_no_points_is_a_zero_gradient_adam_step():
    """Empty input: nerf_table_bin_levels is a no-op and nerf_table_update_binned(n_points = 0) is Adam with g = 0 on
    the listed levels (moments decay, parameters follow them), exactly what nerf_adam_step does with a zero gradient."""
    from nerf_amd._lib import check, lib, ptr, stream
    from nerf_amd.nets import make_point_src
    enc = _grid(16, 2, 16)
    cfg = enc.cfg
    levels = [15, 9]
    ls = _level_set(levels)
    n = lib().nerf_table_bin_workspace_bytes(C.byref(cfg), C.byref(ls), 0)
    assert n >= 0
    ws = torch.empty(max(int(n), 256), dtype=torch.uint8, device='cuda')
    x = torch.zeros(0, 3, device='cuda')
    src = make_point_src(outer_bound=1.0, x=torch.zeros(1, 3, device='cuda'), n_points=0)
    dfeat = torch.zeros(1, 4, device='cuda')
    check(lib().nerf_table_bin_levels(C.byref(cfg), C.byref(src), C.byref(ls), ptr(dfeat), 4, None, ptr(ws), ws.numel(),
                                      stream()), 'bin')
    g = torch.Generator(device='cuda').manual_seed(2)
    p0 = enc.params.detach().clone()
    m0 = torch.randn(p0.shape, device='cuda', generator=g) * 1e-3
    v0 = torch.rand(p0.shape, device='cuda', generator=g) * 1e-6
    p, m, v = p0.clone(), m0.clone(), v0.clone()
    hyper = (1e-2, 0.9, 0.99, 1e-8, 0.0)
    check(lib().nerf_table_update_binned(C.byref(cfg), C.byref(ls), 0, ptr(ws), ws.numel(), ptr(p), ptr(m), ptr(v), *hyper, 5,
                                         1.0, stream()), 'update')
    pr, mr, vr = p0.clone(), m0.clone(), v0.clone()
    gz = torch.zeros_like(p0)
    for l in levels:
        a = int(cfg.offset[l]) * 2
        b = a + int(cfg.entries[l]) * 2
        check(lib().nerf_adam_step(ptr(pr[a:b]), ptr(gz[a:b]), ptr(mr[a:b]), ptr(vr[a:b]), b - a, *hyper, 5, 1.0, 1, stream()),
              'adam')
    assert torch.equal(p, pr) and torch.equal(m, mr) and torch.equal(v, vr)
    assert not torch.equal(p, p0)
