"""BASELINE configs[3]: configs/config_bothsides.yaml -- a CameraGeometryLoader store of TWO scans (the second registered
onto the first by a rigid transform, RGBA + depth, quarter-size images; configs/config_bothsides.yaml:1-16,
loaders/camera_geometry_loader.py:36-66,91-117) feeding the HIP path at the reference's own sampling
(renderer.steps [256,128,64,32] + 128 importance samples, 16 384-ray batches; configs/config_base.yaml:28,62-76).

 (a) batch values of the device-resident store against a numpy restatement of camera_geometry_loader.py:68-117;
 (b) HIP render + loss against the CPU oracle at 96 rays drawn from the store;
 (c) at the full size (16 384 rays x (480 + 128) samples, 2^24 table) the size-independent property set: sorted fine
     depths, affine dependence on the background, bit-identical results under any split of the batch, additivity of
     the gradients, and one Trainer.train_epoch iteration on the store."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS, ZB, SF = [256, 128, 64, 32], [0.1, 1, 10, 100, 1000], 128


@pytest.fixture(scope='module')
def store():
    from nerf_amd.config import configurator
    from nerf_amd.loaders.camera_geometry_loader import CameraGeometryLoader
    from nerf_amd.loaders.synthetic_scans import make_scans
    cfg = configurator(None, argv=['--configs', os.path.join(ROOT, 'configs', 'config_base.yaml'),
                                   os.path.join(ROOT, 'configs', 'config_bothsides.yaml')])
    assert cfg.renderer.steps == STEPS and cfg.trainer.n_rays == 16384
    scans = make_scans(cfg.scene.scene_paths, cfg.scene.transforms, cfg.scene.image_scale)
    loader = CameraGeometryLoader.from_arrays(scans, device='cuda', batch_seed=5)
    return cfg, scans, loader


def _restate(scans, n, h, w, bg):
    """camera_geometry_loader.py:41-62 (stores), :68-88 (ground truth), :91-108 (batch) in numpy."""
    E_all, K_all, rgba, depth = [], [], [], []
    for s in scans:
        T = np.eye(4, dtype=np.float32) if s.get('transform') is None else np.asarray(s['transform'], np.float32)
        E_all.append(T @ s['extrinsics'])
        K_all.append(s['intrinsics'])
        rgb = s['rgb']
        if rgb.shape[-1] == 3:
            rgb = np.concatenate([rgb, np.full(rgb.shape[:-1] + (1,), 255, np.uint8)], -1)
        rgba.append(rgb)
        depth.append(s['depth'])
    E_all, K_all, rgba, depth = (np.concatenate(x) for x in (E_all, K_all, rgba, depth))
    tc = E_all[:, :3, 3].mean(0, keepdims=True)
    E = E_all[n].copy()
    E[:, :3, 3] -= tc
    px = rgba[n, h, w].astype(np.float32) / 255
    gt = px[:, :3] * px[:, 3:] + np.asarray(bg, np.float32) * (1 - px[:, 3:])
    return K_all[n], E, gt, depth[n, h, w]


def test_two_scan_store_batches_on_the_gpu(store):
    cfg, scans, L = store
    assert L.images.is_cuda and L.N == 192 and (L.H, L.W) == (128, 160) and L.depths_bool
    g = torch.Generator().manual_seed(3)
    n, h, w = (torch.randint(0, hi, (4096,), generator=g) for hi in (L.N, L.H, L.W))
    bn, bh, bw, K, E, gt, bg, depth = L.get_custom_batch(n, h, w, background=(0.25, 0.5, 0.75))
    assert all(t.is_cuda for t in (bn, bh, bw, K, E, gt, bg, depth))
    K_o, E_o, gt_o, d_o = _restate(scans, n.numpy(), h.numpy(), w.numpy(), (0.25, 0.5, 0.75))
    assert np.array_equal(K.cpu().numpy(), K_o)
    np.testing.assert_allclose(E.cpu().numpy(), E_o, rtol=0, atol=1e-6)
    np.testing.assert_allclose(gt.cpu().numpy(), gt_o, rtol=0, atol=1e-6)
    assert np.array_equal(depth.cpu().numpy(), d_o)
    # the random batch: indices in range and from both scans, ONE background colour per batch (:72), values consistent
    rn, rh, rw, rK, rE, rgt, rbg, rdepth = L.get_random_batch(16384)
    assert rn.shape == (16384,) and int(rn.min()) >= 0 and int(rn.max()) < L.N and int(rh.max()) < L.H and int(rw.max()) < L.W
    assert bool((rn < 96).any()) and bool((rn >= 96).any()) and rbg.shape == (3,)
    K_o, E_o, gt_o, d_o = _restate(scans, rn.cpu().numpy(), rh.cpu().numpy(), rw.cpu().numpy(), rbg.cpu().numpy())
    np.testing.assert_allclose(rE.cpu().numpy(), E_o, rtol=0, atol=1e-6)
    np.testing.assert_allclose(rgt.cpu().numpy(), gt_o, rtol=0, atol=1e-6)
    assert np.array_equal(rdepth.cpu().numpy(), d_o)
    a = L.get_random_batch(64)
    b = L.get_random_batch(64)
    assert not torch.equal(a[0], b[0]) and not torch.equal(a[6], b[6])      # a fresh draw (and colour) per batch


def test_render_and_loss_from_the_store_vs_oracle(store):
    from test_gpu_parity import NET_CFGS, _make_pair
    from nerf_amd import helpers
    from nerf_amd.renderer import NerfRenderer
    from oracle import ref_cpu
    cfg, scans, L = store
    net, ora = _make_pair(NET_CFGS['small'], n_img=L.N)
    n, h, w, K, E, gt, bg, _ = L.get_random_batch(96)
    noise = torch.rand(96, SF, generator=torch.Generator().manual_seed(4))
    Ro = ref_cpu.OracleRenderer(ora, 2, 2.5, ZB, STEPS, SF, 0.2)
    Rg = NerfRenderer(net, 2, 2.5, ZB, STEPS, SF, 0.2)
    c = lambda t: t.cpu()
    with torch.no_grad():
        img_o, w_o, zs_o, aux_o = Ro.render(c(n), c(h), c(w), c(K), c(E), c(bg), noise=noise)
        loss_o, _ = ref_cpu.train_loss(img_o, c(gt), w_o, zs_o, 10, 100, 1e-5, 1e-2, 1e-1, 1e-1)
    img, wts, zs, aux = Rg.render(n, h, w, K, E, bg, noise=noise.cuda())
    torch.testing.assert_close(aux['z_vals_log'].cpu(), aux_o['z_vals_log'], rtol=0, atol=2e-4)
    torch.testing.assert_close(img.detach().cpu(), img_o, rtol=0, atol=1e-4)
    torch.testing.assert_close(wts.detach().cpu(), w_o, rtol=0, atol=1e-4)
    torch.testing.assert_close(aux['invdepth'].cpu(), aux_o['invdepth'], rtol=0, atol=1e-4 * float(aux_o['invdepth'].abs().max() + 1))
    loss, _ = helpers.fused_train_loss(img, gt, wts, zs, ref_cpu.dist_scalar(10, 100, 1e-5, 1e-2), 1e-1)
    assert abs(float(loss) - float(loss_o)) <= 1e-4 * max(1.0, abs(float(loss_o)))


@pytest.fixture(scope='module')
def full(store):
    from test_gpu_fullsize_properties import NET
    from nerf_amd.nets import NeRFNetwork
    from nerf_amd.renderer import NerfRenderer
    cfg, scans, L = store
    net = NeRFNetwork(L.N, seed=0, **NET)
    with torch.no_grad():
        net.encoder.params.mul_(3e3)
        net.latent_emb.normal_(0, 0.3, generator=torch.Generator(device='cuda').manual_seed(1))
    R = NerfRenderer(net, cfg.scene.inner_bound, cfg.scene.outer_bound, cfg.renderer.z_bounds, cfg.renderer.steps,
                     cfg.renderer.importance_steps, cfg.renderer.alpha)
    N = cfg.trainer.n_rays
    batch = L.get_random_batch(N)
    noise = torch.rand(N, SF, device='cuda', generator=torch.Generator(device='cuda').manual_seed(2))
    yield net, R, batch, noise, N
    del net, R
    torch.cuda.empty_cache()


def test_bothsides_render_properties_at_full_size(full):
    net, R, (n, h, w, K, E, gt, bg, depth), noise, N = full
    assert N == 16384 and sum(R.steps_firstpass) == 480
    with torch.no_grad():
        img, wts, zs, aux = R.render(n, h, w, K, E, bg, noise=noise)
        zl = aux['z_vals_log']
        assert tuple(zl.shape) == (N, SF) and bool((zl[:, 1:] >= zl[:, :-1]).all())           # sorted fine depths
        assert float(zl.min()) >= math.log10(0.1) - 1e-6 and float(zl.max()) <= math.log10(1000) + 1e-6
        assert bool((zs >= -1e-6).all()) and bool((zs <= 1 + 1e-6).all())
        assert bool(torch.isfinite(img).all()) and bool((wts >= 0).all()) and float(wts.sum(-1).max()) <= 1 + 1e-5
        acc = wts.sum(-1, keepdim=True)
        bg2 = torch.tensor([0.9, 0.1, 0.4], device='cuda')
        img2, wts2, _, aux2 = R.render(n, h, w, K, E, bg2, noise=noise)
        assert torch.equal(wts, wts2) and torch.equal(aux['rgbs'], aux2['rgbs'])
        torch.testing.assert_close(img2 - img, (1 - acc) * (bg2 - bg), rtol=0, atol=2e-6)      # affine in the background
        torch.testing.assert_close(img, (wts[..., None] * aux['rgbs']).sum(1) + (1 - acc) * bg, rtol=0, atol=1e-5)
        cut = 5003                                                                             # any split of the batch
        a = R.render(n[:cut], h[:cut], w[:cut], K[:cut], E[:cut], bg, noise=noise[:cut])
        b = R.render(n[cut:], h[cut:], w[cut:], K[cut:], E[cut:], bg, noise=noise[cut:])
        assert torch.equal(torch.cat([a[0], b[0]]), img) and torch.equal(torch.cat([a[1], b[1]]), wts)
        assert torch.equal(torch.cat([a[3]['z_vals_log'], b[3]['z_vals_log']]), zl)
        # rays of both scans took part, and the second scan's rays start on the other side of the row
        ro, _ = R.get_rays(h, w, K, E)
        assert float(ro[n < 96, 0].mean()) * float(ro[n >= 96, 0].mean()) < 0


def test_bothsides_gradients_add_up_at_full_size(full):
    from nerf_amd import helpers
    net, R, (n, h, w, K, E, gt, bg, depth), noise, N = full
    params = [net.sigma_net.params, net.color_net.params, net.latent_emb, net.encoder.params]

    def grads(sl, scale):
        for p in params:
            p.grad = None
        img, wts, zs, _ = R.render(n[sl], h[sl], w[sl], K[sl], E[sl], bg, noise=noise[sl])
        loss, _ = helpers.fused_train_loss(img, gt[sl], wts, zs, 1e-3, 1e-1)
        (loss * scale).backward()
        torch.cuda.synchronize()
        return [p.grad.detach().clone() for p in params]
    cut = 6000
    whole = grads(slice(None), 1.0)
    ga = grads(slice(0, cut), cut / N)
    gb = grads(slice(cut, None), (N - cut) / N)
    for p in params:
        p.grad = None
    for gw, a, b, name in zip(whole, ga, gb, ('sigma_net', 'color_net', 'latent', 'table')):
        s = float(gw.abs().max())
        assert s > 0, name
        torch.testing.assert_close(a + b, gw, rtol=1e-3, atol=2e-5 * s, msg=lambda m, nm=name: nm + ': ' + m)


def test_trainer_iterations_on_the_store(store, full):
    """Trainer.train_epoch on the two-scan store at config_bothsides' batch size: the wiring of train.py (three Adam
    groups, LambdaLR) with CameraGeometryLoader batches; losses finite, every parameter group moves, no step dropped."""
    from nerf_amd.logger import Logger
    from nerf_amd.optim import FusedAdam
    from nerf_amd.trainer import Trainer
    cfg, scans, L = store
    net, R, _, _, N = full
    with torch.no_grad():
        net.encoder.params.mul_(1.0 / 3e3)
    opt = FusedAdam([
        {'name': 'encoding', 'params': list(net.encoder.parameters()), 'lr': cfg.optimizer.encoding.lr},
        {'name': 'latent_emb', 'params': [net.latent_emb], 'lr': cfg.optimizer.latent_emb.lr},
        {'name': 'net', 'params': list(net.sigma_net.parameters()) + list(net.color_net.parameters()),
         'weight_decay': cfg.optimizer.net.weight_decay, 'lr': cfg.optimizer.net.lr}],
        betas=tuple(cfg.optimizer.betas), eps=cfg.optimizer.eps).attach(net)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda=lambda x: 1, last_epoch=-1)
    lg = Logger(None, verbose=False)
    tr = Trainer(net, L, lg, None, R, opt, sched, N, 1, 6, None, None, None, cfg.trainer.dist_loss_lambda1,
                 cfg.trainer.dist_loss_lambda2, cfg.trainer.depth_loss_lambda1, cfg.trainer.depth_loss_lambda2)
    before = [p.detach().clone() for p in (net.sigma_net.params, net.color_net.params, net.latent_emb)]
    t_before = net.encoder.params.detach()[-(1 << 20):].clone()
    tr.train_epoch(6)
    torch.cuda.synchronize()
    losses = [float(v) for v in lg.scalars['loss']]
    assert len(losses) == 6 and all(math.isfinite(v) and v > 0 for v in losses)
    opt._collect_flags(wait=True)
    assert tr.iter == 6 and opt.skipped_steps == 0
    for b, p in zip(before, (net.sigma_net.params, net.color_net.params, net.latent_emb)):
        assert float((p.detach() - b).abs().max()) > 0
    assert float((net.encoder.params.detach()[-(1 << 20):] - t_before).abs().max()) > 0
    net.enable_direct_grads(False)
    net.table_pipeline = None
