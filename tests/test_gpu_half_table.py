"""`precision: float16` for the hash grid (reference nets.py:31-36, passed to tcnn.Encoding as dtype at nets.py:49):
the fp32 table stays the master copy the optimizer updates, every kernel that reads the table gathers from its
half-precision image (nerf_table_to_half + the `_f16` entry points, include/nerf_hip.h).

Bars.  tinycudann is absent from /root/reference, so its fp16 arithmetic is UNPINNED; what is checked here is
 (a) the image is the nearest-even fp16 rounding of the master, bit for bit (torch's .half());
 (b) the `_f16` kernels are BIT-IDENTICAL to the fp32 kernels run on a table that was rounded to fp16 beforehand
     (widening fp16 -> fp32 is exact, the arithmetic after the load is the same code);
 (c) against the CPU oracle with `encoding_precision='float16'` (table rounded where it is gathered, gradients straight
     through to the master): the same fp32 bars as tests/test_gpu_parity.py, forward, backward and a training trace;
 (d) the image follows the master through FusedAdam.step, in-place torch updates and load_state_dict, and is not part
     of the checkpoint."""
import pytest
import torch

pytestmark = pytest.mark.gpu

KW = dict(encoding_n_features_per_level=2, geo_feat_dim=15, sigma_hidden_dim=64, sigma_num_layers=2,
          encoding_dir_precision='float32', encoding_dir_encoding='SphericalHarmonics', encoding_dir_degree=4,
          latent_embedding_dim=48, color_hidden_dim=64, color_num_layers=3)
CFGS = {
    'small': dict(encoding_n_levels=8, encoding_log2_hashmap_size=15),
    'base19': dict(encoding_n_levels=20, encoding_log2_hashmap_size=19),
    'f4': dict(encoding_n_levels=8, encoding_log2_hashmap_size=15, encoding_n_features_per_level=4),
    'f1': dict(encoding_n_levels=11, encoding_log2_hashmap_size=14, encoding_n_features_per_level=1),
}


def _nets(cfg, n_img=7, seed=3, with_oracle=True):
    """(half-table network, fp32 network holding the fp16-ROUNDED table, oracle in float16 mode): same master values."""
    from nerf_amd.nets import NeRFNetwork
    from oracle.ref_cpu import OracleNetwork
    kw = dict(KW)
    kw.update(cfg)
    ora = OracleNetwork(n_img, seed=seed, encoding_precision='float16', **kw)
    with torch.no_grad():
        ora.encoder_params.mul_(1e3)
        ora.latent_emb.copy_(torch.randn(n_img, kw['latent_embedding_dim'], generator=torch.Generator().manual_seed(seed)) * 0.3)
    n16 = NeRFNetwork(n_img, encoding_precision='float16', **kw)
    n32 = NeRFNetwork(n_img, encoding_precision='float32', **kw)
    with torch.no_grad():
        for net in (n16, n32):
            net.encoder.params.copy_(ora.encoder_params)
            net.sigma_net.params.copy_(ora.sigma_params)
            net.color_net.params.copy_(ora.color_params)
            net.latent_emb.copy_(ora.latent_emb)
        n32.encoder.params.copy_(n32.encoder.params.half().float())
    assert n16.encoding_precision == torch.float16 and n16.encoder.half and not n32.encoder.half
    return n16, n32, ora


def _points(M, seed=11):
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(M, 3, generator=g) * 2 - 1) * 2.4
    d = torch.randn(M, 3, generator=g)
    d = d / d.norm(dim=-1, keepdim=True)
    n = torch.randint(0, 7, (M,), generator=g)
    return x, d, n, g


def test_table_to_half_is_nearest_even_rounding():
    import ctypes as C
    from nerf_amd._lib import check, lib, ptr, stream
    g = torch.Generator().manual_seed(0)
    for n in (0, 1, 3, 4, 1027, 1 << 20):
        t = torch.randn(n, generator=g) * torch.tensor(10.0).pow(torch.randint(-9, 5, (n,), generator=g).float())
        if n >= 1027:
            t[:9] = torch.tensor([0.0, -0.0, 65504.0, 65520.0, 7e4, -7e4, 6e-8, 2.98e-8, 1.0 + 2.0 ** -11])   # max, overflow, subnormals, a tie
        src = t.cuda()
        dst = torch.full((n + 4,), 7.0, dtype=torch.float16, device='cuda')
        check(lib().nerf_table_to_half(ptr(src), ptr(dst), n, stream()), 'nerf_table_to_half')
        assert torch.equal(dst[:n].view(torch.int16), src.half().view(torch.int16)), n
        assert bool((dst[n:] == 7.0).all())        # nothing written past the end
    assert lib().nerf_table_to_half(None, None, 4, None) < 0 and b'null' in lib().nerf_last_error()
    del C


@pytest.mark.parametrize('name', list(CFGS))
def test_half_gathers_equal_fp32_kernels_on_the_rounded_table(name):
    n16, n32, _ = _nets(CFGS[name])
    M = 1000 if name != 'base19' else 517
    x, d, n, g = _points(M)
    ob = 2.5
    xc, dc, nc = x.cuda(), d.cuda(), n.cuda()
    x01 = (xc + ob) / (2 * ob)
    assert torch.equal(n16.encoder(x01), n32.encoder(x01))                       # nerf_hashgrid_fwd_f16
    s16, a16 = n16.density(xc, ob)                                               # nerf_density_fwd_f16
    s32, a32 = n32.density(xc, ob)
    assert torch.equal(s16, s32) and torch.equal(a16['x_hashtable'], a32['x_hashtable'])
    f16, f32_ = n16(xc, dc, nc, ob), n32(xc, dc, nc, ob)                         # nerf_field_fwd_f16
    assert torch.equal(f16[0], f32_[0]) and torch.equal(f16[1], f32_[1])
    assert torch.equal(f16[2]['x_hashtable'], f32_[2]['x_hashtable'])
    with torch.no_grad():
        r16 = n16.field_nograd(dict(outer_bound=ob, x=xc, d=dc, n=nc, n_points=M))
        r32 = n32.field_nograd(dict(outer_bound=ob, x=xc, d=dc, n=nc, n_points=M))
    assert torch.equal(r16[0], r32[0]) and torch.equal(r16[1], r32[1])
    # gradients never read the table: the same kernels on the same inputs (equal up to the order of their atomic sums),
    # and they land on the fp32 master
    gs, gc = torch.randn(M, generator=g).cuda(), torch.randn(M, 3, generator=g).cuda()
    for net, out in ((n16, f16), (n32, f32_)):
        (out[0] * gs).sum().add((out[1] * gc).sum()).backward()
    for a, b in ((n16.encoder.params.grad, n32.encoder.params.grad), (n16.sigma_net.params.grad, n32.sigma_net.params.grad),
                 (n16.color_net.params.grad, n32.color_net.params.grad), (n16.latent_emb.grad, n32.latent_emb.grad)):
        assert a.dtype == torch.float32 and float(b.abs().max()) > 0
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6 * float(b.abs().max()))
    # d(sigma)/d(position) for the pose alignment: nerf_density_bwd_input_f16
    dx = []
    for net in (n16, n32):
        xg = xc.clone().requires_grad_(True)
        sig, _ = net.density(xg, ob)
        (sig * gs).sum().backward()
        dx.append(xg.grad)
    assert torch.equal(dx[0], dx[1]) and float(dx[0].abs().max()) > 0


@pytest.mark.parametrize('name', ['small', 'base19'])
def test_half_table_field_vs_oracle(name):
    n16, _, ora = _nets(CFGS[name])
    M = 1000 if name != 'base19' else 517
    x, d, n, g = _points(M)
    ob = 2.5
    sig_o, rgb_o, aux_o = ora(x, d, n, ob)
    sig_g, rgb_g, aux_g = n16(x.cuda(), d.cuda(), n.cuda(), ob)
    scale = float(sig_o.abs().max()) + 1e-6
    torch.testing.assert_close(aux_g['x_hashtable'].cpu(), aux_o['x_hashtable'], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(sig_g.cpu(), sig_o, rtol=1e-4, atol=1e-5 * scale)
    torch.testing.assert_close(rgb_g.cpu(), rgb_o, rtol=0, atol=1e-5)
    # the rounding is really there: the fp32 evaluation of the same master differs by about 2^-11 relative
    with torch.no_grad():
        exact = ora.encoder_params
        ora.half_table = False
        xh_exact = ora.encode((x + ob) / (2 * ob))
        ora.half_table = True
        assert exact.dtype == torch.float32
    rel = float((aux_o['x_hashtable'] - xh_exact).abs().max() / xh_exact.abs().max())
    assert 1e-6 < rel < 2e-3, rel
    gs, gc = torch.randn(M, generator=g), torch.randn(M, 3, generator=g)
    (sig_o * gs).sum().add((rgb_o * gc).sum()).backward()
    (sig_g * gs.cuda()).sum().add((rgb_g * gc.cuda()).sum()).backward()
    for a, b, what in ((n16.sigma_net.params.grad, ora.sigma_params.grad, 'sigma_net'),
                       (n16.color_net.params.grad, ora.color_params.grad, 'color_net'),
                       (n16.latent_emb.grad, ora.latent_emb.grad, 'latent'),
                       (n16.encoder.params.grad, ora.encoder_params.grad, 'table')):
        s = float(b.abs().max()) + 1e-12
        torch.testing.assert_close(a.cpu(), b, rtol=1e-3, atol=2e-5 * s, msg=lambda m, w=what: w + ': ' + m)


def _batch(loader_seed, N, n_img=7):
    from nerf_amd.loaders.procedural import sphere_cameras
    K_all, E_all = sphere_cameras(n_img, loader_seed)
    g = torch.Generator().manual_seed(50 + loader_seed)
    n = torch.randint(0, n_img, (N,), generator=g)
    h = torch.randint(0, 800, (N,), generator=g)
    w = torch.randint(0, 800, (N,), generator=g)
    return n, h, w, K_all[n], E_all[n], torch.rand(3, generator=g), torch.rand(N, 128, generator=g), torch.rand(N, 3, generator=g)


@pytest.mark.parametrize('name,fused', [('small', True), ('base19', True), ('base19', False)])
def test_half_table_train_steps_vs_oracle(name, fused, monkeypatch):
    """Training with the half-precision image: fp32 master + fp32 Adam (FusedAdam; the binned update on 'base19'), the image
    refreshed after every step.  Loss trace against the oracle in float16 mode + torch.optim.Adam; the master must keep
    what fp16 could not (updates below the fp16 spacing accumulate).  fused: the table passes of FusedAdam write the image
    themselves (nerf_table_update_binned_f16, nerf_adam_step_f16); otherwise a nerf_table_to_half pass runs before the next
    forward.  Either way the image must be the rounding of the master, bit for bit, at every step."""
    from nerf_amd import helpers, optim
    monkeypatch.setattr(optim, 'FUSED_HALF_IMAGE', fused)
    from nerf_amd.optim import FusedAdam
    from nerf_amd.renderer import NerfRenderer
    from oracle import ref_cpu
    n16, _, ora = _nets(CFGS[name])
    with torch.no_grad():
        ora.latent_emb.zero_()
        n16.latent_emb.zero_()
    Ro = ref_cpu.OracleRenderer(ora, 2, 2.5, [1.6, 4.8], [64], 128, 0.2)
    Rg = NerfRenderer(n16, 2, 2.5, [1.6, 4.8], [64], 128, 0.2)
    oo = ref_cpu.make_optimizer(ora)
    og = FusedAdam([{'params': list(n16.encoder.parameters()), 'lr': 2e-2},
                    {'params': [n16.latent_emb], 'lr': 2e-2},
                    {'params': list(n16.sigma_net.parameters()) + list(n16.color_net.parameters()), 'weight_decay': 1e-6, 'lr': 1e-3}],
                   betas=(0.9, 0.99), eps=1e-15).attach(n16)
    assert bool(n16.table_pipeline.binned_levels) == (name == 'base19')
    lo, lg = [], []
    for it in range(6):
        n, h, w, K, E, bg, noise, gt = _batch(10 + it, 96)
        oo.zero_grad()
        img, wts, zs, _ = Ro.render(n, h, w, K, E, bg, noise=noise)
        loss, _ = ref_cpu.train_loss(img, gt, wts, zs, it, 100, 1e-5, 1e-2, 1e-1, 1e-1)
        loss.backward()
        oo.step()
        lo.append(float(loss))
        og.zero_grad()
        img, wts, zs, _ = Rg.render(n.cuda(), h.cuda(), w.cuda(), K.cuda(), E.cuda(), bg.cuda(), noise=noise.cuda())
        loss_g, _ = helpers.fused_train_loss(img, gt.cuda(), wts, zs, ref_cpu.dist_scalar(it, 100, 1e-5, 1e-2), 1e-1)
        loss_g.backward()
        og.step()
        lg.append(float(loss_g))
        # fused: the image was written by the optimizer's own passes and is still marked current (no conversion pass)
        assert (n16.encoder._half_state is not None) == fused
        # the image in use at the NEXT forward is the rounding of the updated master
        _, tab = n16.encoder.gather_table()
        assert tab.dtype == torch.float16 and torch.equal(tab, n16.encoder.params.detach().half())
    for k, (a, b) in enumerate(zip(lg, lo)):
        assert abs(a - b) <= (2e-4 if k < 4 else 2e-3) * max(1.0, abs(b)), (k, lg, lo)
    assert lg[-1] < lg[0]
    master = n16.encoder.params.detach()
    assert master.dtype == torch.float32
    touched = og.state[n16.encoder.params]['exp_avg'] != 0
    assert int(touched.sum()) > 1000
    # most touched entries are no longer fp16-representable: the master carries more than the image
    assert float((master[touched] != master[touched].half().float()).float().mean()) > 0.9
    # ... and agrees with the oracle's master wherever the gradient was not summation noise (Adam with eps = 1e-15 turns
    # noise on a near-zero gradient into a +-lr step: a minority of the touched entries)
    d = (master.cpu() - ora.encoder_params.detach()).abs()[touched.cpu()]
    assert float(d.median()) <= 1e-3, (float(d.median()), float(d.max()))


def test_half_image_follows_the_master_and_stays_out_of_checkpoints():
    n16, n32, _ = _nets(CFGS['small'])
    x, d, n, _ = _points(300)
    xc = x.cuda()
    s0, _ = n16.density(xc, 2.5)
    img0 = n16.encoder.params_half
    assert img0 is not None and img0.dtype == torch.float16 and img0.numel() == n16.encoder.params.numel()
    sd = {k: v.clone() for k, v in n16.state_dict().items()}
    assert set(sd) == set(n32.state_dict()) and all(v.dtype != torch.float16 for v in sd.values())
    # an in-place torch update of the master (what torch.optim.Adam does) is seen at the next forward
    with torch.no_grad():
        n16.encoder.params.mul_(0.5)
        n32.encoder.params.copy_(n16.encoder.params.half().float())
    s1, _ = n16.density(xc, 2.5)
    assert torch.equal(s1, n32.density(xc, 2.5)[0]) and not torch.equal(s1, s0)
    # load_state_dict copies into the parameter in place: seen as well
    n16.load_state_dict(sd)
    assert torch.equal(n16.density(xc, 2.5)[0], s0)
    # no conversion when nothing changed (the image buffer is reused and not rewritten)
    before = n16.encoder._half_state
    n16.density(xc, 2.5)
    assert n16.encoder._half_state == before and n16.encoder.params_half.data_ptr() == img0.data_ptr()


def test_half_image_under_the_non_finite_guard():
    """A dropped step (non-finite loss, trainer.py:134-136) leaves master AND image untouched; the next good step moves both."""
    from nerf_amd import helpers
    from nerf_amd.optim import FusedAdam
    from nerf_amd.renderer import NerfRenderer
    n16, _, _ = _nets(CFGS['base19'])
    R = NerfRenderer(n16, 2, 2.5, [1.6, 4.8], [64], 128, 0.2)
    opt = FusedAdam([{'params': list(n16.encoder.parameters()) + [n16.latent_emb], 'lr': 2e-2},
                     {'params': list(n16.sigma_net.parameters()) + list(n16.color_net.parameters()), 'lr': 1e-3}],
                    betas=(0.9, 0.99), eps=1e-15).attach(n16)
    n, h, w, K, E, bg, noise, gt = [t.cuda() for t in _batch(3, 64)]
    for bad in (False, True, False):
        before = n16.encoder.params.detach().clone()
        opt.zero_grad()
        img, wts, zs, _ = R.render(n, h, w, K, E, bg, noise=noise)
        target = gt.clone()
        if bad:
            target[5, 1] = float('nan')
        loss, _ = helpers.fused_train_loss(img, target, wts, zs, 1e-3, 1e-1)
        opt.skip_flag = loss.nonfinite_flag
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        assert n16.encoder._half_state is not None          # written (or, when dropped, left alone) by the table passes
        _, tab = n16.encoder.gather_table()
        assert torch.equal(tab, n16.encoder.params.detach().half())
        assert torch.equal(before, n16.encoder.params.detach()) == bad
