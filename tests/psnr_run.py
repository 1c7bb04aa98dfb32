#!/usr/bin/env python3
"""PSNR evidence on the procedural synthetic scene (run on the GPU box; lives under tests/ because it checks the HIP path against the CPU oracle -- test infrastructure, not collected by pytest).

 (1) quality: train the HIP path on ground-truth images rendered from an analytic field and report the PSNR of a
     training view (how the reference validates, trainer.py:95-101);
 (2) PSNR delta: train the HIP path and the CPU oracle from identical weights on identical batches / noise for the
     same budget and compare loss curves and final PSNR.

    python tests/psnr_run.py --out profiles/psnr_r01.json
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from nerf_amd import helpers
from nerf_amd.inference import Inferencer
from nerf_amd.loaders.procedural import ProceduralSceneLoader
from nerf_amd.nets import NeRFNetwork
from nerf_amd.optim import FusedAdam
from nerf_amd.renderer import NerfRenderer
from oracle import ref_cpu

STEPS, ZB, SF, ALPHA, INNER, OUTER = [64], [1.6, 4.8], 128, 0.2, 2, 2.5


def net_kwargs(L, T, precision='float32'):
    return dict(encoding_precision=precision, encoding_n_levels=L, encoding_n_features_per_level=2,
                encoding_log2_hashmap_size=T, geo_feat_dim=15, sigma_hidden_dim=64, sigma_num_layers=2,
                encoding_dir_precision='float32', encoding_dir_encoding='SphericalHarmonics', encoding_dir_degree=4,
                latent_embedding_dim=48, color_hidden_dim=64, color_num_layers=3)


@torch.no_grad()
def render_ground_truth(loader, res):
    """RGBA uint8 images of the analytic field through the HIP renderer (stub-model route)."""
    field = ref_cpu.AnalyticField(centers=((0.5, 0.2, -0.3), (-0.6, -0.4, 0.4), (0.0, 0.7, 0.2)), widths=(0.45, 0.35, 0.5),
                                  peaks=(80.0, 150.0, 40.0), floor=0.0)
    R = NerfRenderer(field, INNER, OUTER, ZB, STEPS, SF, ALPHA)
    imgs = torch.zeros(loader.N, res, res, 4, dtype=torch.uint8, device='cuda')
    zero = torch.zeros(3, device='cuda')
    for i in range(loader.N):
        n, h, w, K, E, _, _, _ = loader.get_image_batch(i)
        flat = lambda t: t.reshape(-1, *t.shape[2:])
        noise = torch.full((res * res, SF), 0.5, device='cuda')
        rgb, wts, _, _ = R.render(flat(n), flat(h), flat(w), flat(K), flat(E), zero, noise=noise)
        a = wts.sum(-1, keepdim=True).clamp(0, 1)
        col = (rgb / a.clamp_min(1e-6)).clamp(0, 1)
        imgs[i] = (torch.cat([col, a], -1) * 255 + 0.5).to(torch.uint8).reshape(res, res, 4)
    return imgs


def make_opt(net):
    return FusedAdam([{'params': list(net.encoder.parameters()), 'lr': 2e-2}, {'params': [net.latent_emb], 'lr': 2e-2},
                      {'params': list(net.sigma_net.parameters()) + list(net.color_net.parameters()), 'weight_decay': 1e-6,
                       'lr': 1e-3}], betas=(0.9, 0.99), eps=1e-15).attach(net)


def eval_psnr(renderer, loader, image_num):
    inf = Inferencer(renderer, 65536, image_num, 0, 0.1, 0.05, 0.4, [1], 20)
    n, h, w, K, E, gt, _, _ = loader.get_image_batch(image_num)
    noise = torch.full((n.numel(), SF), 0.5, device='cuda')
    img, _ = inf.render_image(n, h, w, K, E, noise=noise)
    pred = torch.as_tensor(img).cuda().reshape(-1, 3).clamp(0, 1)
    return helpers.psnr(pred, gt.reshape(-1, 3))


def quality_run(res, n_img, L, T, iters, n_rays):
    loader = ProceduralSceneLoader(n_img, res, res, seed=0, batch_seed=1)
    loader.set_images(render_ground_truth(loader, res))
    net = NeRFNetwork(n_img, seed=0, **net_kwargs(L, T))
    R = NerfRenderer(net, INNER, OUTER, ZB, STEPS, SF, ALPHA)
    opt = make_opt(net)
    p0 = eval_psnr(R, loader, n_img // 2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(iters):
        opt.zero_grad()
        n, h, w, K, E, gt, bg, _ = loader.get_random_batch(n_rays)
        img, wts, zs, _ = R.render(n, h, w, K, E, bg)
        lam = ref_cpu.dist_scalar(it, iters, 1e-5, 1e-2)
        loss, _ = helpers.fused_train_loss(img, gt, wts, zs, lam, 1e-1)
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    p1 = eval_psnr(R, loader, n_img // 2)
    # throughput on the TRAINED field (sparser sigma => other gather locality than the random-init bench)
    def one_step(it):
        opt.zero_grad()
        n, h, w, K, E, gt, bg, _ = loader.get_random_batch(n_rays)
        img, wts, zs, _ = R.render(n, h, w, K, E, bg)
        loss, _ = helpers.fused_train_loss(img, gt, wts, zs, 1e-2, 1e-1)
        loss.backward()
        opt.step()
    for it in range(5):
        one_step(it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(50):
        one_step(it)
    torch.cuda.synchronize()
    ms_trained = (time.perf_counter() - t0) / 50 * 1e3
    inf = Inferencer(R, 65536, 0, 0, 0.1, 0.05, 0.4, [1], 20)
    views = [loader.get_image_batch(i) for i in range(16)]
    inf.render_image(*views[0][:5])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for (n, h, w, K, E, _, _, _) in views:
        inf.render_image(n, h, w, K, E)
    torch.cuda.synchronize()
    render_rps = 16 * res * res / (time.perf_counter() - t0)
    return {'config': 'L%d T%d, %d images %dx%d, %d iters x %d rays' % (L, T, n_img, res, res, iters, n_rays),
            'psnr_before': p0, 'psnr_after': p1, 'train_seconds': dt, 'final_loss': float(loss),
            'trained_ms_per_step': ms_trained, 'trained_train_rays_per_s': n_rays / ms_trained * 1e3,
            'trained_render_rays_per_s_%dx%d_views' % (res, res): render_rps}


def delta_run(res, n_img, L, T, iters, n_rays, fused_tail=True, precision='float32'):
    """HIP vs CPU oracle: identical init, batches, noise, budget.  `fused_tail`: the HIP side takes the route
    Trainer.train_step takes (NerfRenderer.render_loss: compositing + losses in one kernel); False = render() followed by
    helpers.fused_train_loss.  The result names the table levels that went through the binned update.
    precision 'float16': both sides gather from the fp16 rounding of their fp32 master table (nets.py:31-36)."""
    loader = ProceduralSceneLoader(n_img, res, res, seed=0, batch_seed=1)
    imgs = render_ground_truth(loader, res)
    loader.set_images(imgs)
    imgs_c = imgs.cpu()
    ora = ref_cpu.OracleNetwork(n_img, seed=0, **net_kwargs(L, T, precision))
    net = NeRFNetwork(n_img, **net_kwargs(L, T, precision))
    with torch.no_grad():
        net.encoder.params.copy_(ora.encoder_params)
        net.sigma_net.params.copy_(ora.sigma_params)
        net.color_net.params.copy_(ora.color_params)
    Ro = ref_cpu.OracleRenderer(ora, INNER, OUTER, ZB, STEPS, SF, ALPHA)
    Rg = NerfRenderer(net, INNER, OUTER, ZB, STEPS, SF, ALPHA)
    oo, og = ref_cpu.make_optimizer(ora), make_opt(net)
    K_all, E_all = loader.intrinsics.cpu(), loader.extrinsics.cpu()
    g = torch.Generator().manual_seed(123)
    lo, lg = [], []
    for it in range(iters):
        n = torch.randint(0, n_img, (n_rays,), generator=g)
        h = torch.randint(0, res, (n_rays,), generator=g)
        w = torch.randint(0, res, (n_rays,), generator=g)
        bg, noise = torch.rand(3, generator=g), torch.rand(n_rays, SF, generator=g)
        rgba = imgs_c[n, h, w].to(torch.float32) / 255
        gt = rgba[:, :3] * rgba[:, 3:] + bg * (1 - rgba[:, 3:])
        lam = ref_cpu.dist_scalar(it, iters, 1e-5, 1e-2)
        oo.zero_grad()
        img, wts, zs, _ = Ro.render(n, h, w, K_all[n], E_all[n], bg, noise=noise)
        loss_o = ref_cpu.criterion_rgb(img, gt) + lam * ref_cpu.criterion_dist(wts, zs) + 0.1 * torch.mean(wts * (1 - zs))
        loss_o.backward()
        oo.step()
        og.zero_grad()
        c = lambda t: t.cuda()
        if fused_tail:
            loss_g = Rg.render_loss(c(n), c(h), c(w), c(K_all[n]), c(E_all[n]), c(bg), c(gt), lam, 0.1, noise=c(noise))[0]
        else:
            img, wts, zs, _ = Rg.render(c(n), c(h), c(w), c(K_all[n]), c(E_all[n]), c(bg), noise=c(noise))
            loss_g, _ = helpers.fused_train_loss(img, c(gt), wts, zs, lam, 0.1)
        og.skip_flag = loss_g.nonfinite_flag
        loss_g.backward()
        og.step()
        lo.append(float(loss_o))
        lg.append(float(loss_g))
    # evaluate both on one view with fixed noise
    i = n_img // 2
    n, h, w, K, E, gt, _, _ = loader.get_image_batch(i)
    flat = lambda t: t.reshape(-1, *t.shape[2:])
    noise = torch.full((res * res, SF), 0.5)
    with torch.no_grad():
        img_o, _, _, _ = Ro.render(flat(n).cpu(), flat(h).cpu(), flat(w).cpu(), flat(K).cpu(), flat(E).cpu(), torch.ones(3), noise=noise)
    img_g, _, _, _ = Rg.render(flat(n), flat(h), flat(w), flat(K), flat(E), torch.ones(3, device='cuda'), noise=noise.cuda())
    gtc = gt.reshape(-1, 3)
    p_o = ref_cpu.psnr(img_o.clamp(0, 1), gtc.cpu())
    p_g = helpers.psnr(img_g.detach().clamp(0, 1), gtc)
    pipe = net.table_pipeline
    return {'config': 'L%d T%d, %d images %dx%d, %d iters x %d rays, identical init/batches/noise' % (L, T, n_img, res, res, iters, n_rays)
            + ('' if precision == 'float32' else ', table precision ' + precision),
            'binned_levels': list(pipe.binned_levels) if pipe is not None else [], 'fused_tail': bool(fused_tail),
            'skipped_steps': og.skipped_steps, 'psnr_hip': p_g, 'psnr_cpu_oracle': p_o, 'psnr_delta': p_g - p_o,
            'max_abs_loss_diff': max(abs(a - b) for a, b in zip(lg, lo)), 'loss_first': [lg[0], lo[0]], 'loss_last': [lg[-1], lo[-1]],
            'max_abs_image_diff': float((img_g.detach().cpu() - img_o).abs().max())}


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'psnr_r01.json'))
    ap.add_argument('--quality-iters', type=int, default=2000)
    ap.add_argument('--delta-iters', type=int, default=60)
    ap.add_argument('--metric-delta-iters', type=int, default=60,
                    help='PSNR delta at the METRIC network (L20 T24, 4096 rays): the oracle takes ~2.5 s per step')
    ap.add_argument('--build', default='', help='commit / label of the build that produced the file')
    args = ap.parse_args()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    out = {'build': args.build}
    out['quality_full'] = quality_run(res=200, n_img=100, L=20, T=24, iters=args.quality_iters, n_rays=4096)
    print(json.dumps(out['quality_full']), flush=True)
    out['delta_small'] = delta_run(res=64, n_img=20, L=8, T=15, iters=args.delta_iters, n_rays=1024)
    print(json.dumps(out['delta_small']), flush=True)
    out['delta_binned_T19'] = delta_run(res=64, n_img=20, L=20, T=19, iters=40, n_rays=1024)
    print(json.dumps(out['delta_binned_T19']), flush=True)
    if args.metric_delta_iters > 0:
        out['delta_metric_network'] = delta_run(res=200, n_img=100, L=20, T=24, iters=args.metric_delta_iters, n_rays=4096)
        print(json.dumps(out['delta_metric_network']), flush=True)
    with open(args.out, 'w') as f:
        json.dump(out, f, indent=1)
