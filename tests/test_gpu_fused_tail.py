"""nerf_composite_loss_fwd / _bwd (compositing + the three losses in one kernel, their backward in one more) against
the separate entry points they fuse: nerf_composite_fwd -> nerf_losses -> (autograd multiply) -> nerf_composite_bwd.
The arithmetic is the same operation for operation, so image, weights, zlog_s and both gradients must be BIT-IDENTICAL;
the loss totals differ only in the order the workgroups add them."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _inputs(N, S, seed):
    g = torch.Generator().manual_seed(seed)
    sig = (torch.rand(N, S, generator=g) * 8) * (torch.rand(N, S, generator=g) > 0.4)
    rgb = torch.rand(N, S, 3, generator=g)
    zl = torch.sort(torch.rand(N, S, generator=g) * (math.log10(4.8) - math.log10(1.6)) + math.log10(1.6), dim=-1).values
    z = torch.pow(10, zl)
    bg, gt = torch.rand(3, generator=g), torch.rand(N, 3, generator=g)
    return [t.cuda().contiguous() for t in (sig, rgb, z, zl, bg, gt)]


@pytest.mark.parametrize('N,S,upstream', [(1, 4, 1.0), (37, 128, 1.0), (4096, 128, 1.0), (513, 200, 0.25), (70, 512, 1.0)])
def test_fused_tail_equals_separate_kernels(N, S, upstream):
    from nerf_amd import helpers
    from nerf_amd.renderer import _CompositeFn, _CompositeLossFn
    sig, rgb, z, zl, bg, gt = _inputs(N, S, 5)
    lo, rng = math.log10(1.6), math.log10(4.8) - math.log10(1.6)
    lam_dist, lam_depth = 3e-3, 1e-1
    # separate
    s0, c0 = sig.clone().requires_grad_(True), rgb.clone().requires_grad_(True)
    img0, inv0, w0, zs0 = _CompositeFn.apply(s0, c0, z, zl, bg, lo, rng)
    loss0, parts0 = helpers.fused_train_loss(img0, gt, w0, zs0, lam_dist, lam_depth)
    (loss0 * upstream).backward()
    # fused
    s1, c1 = sig.clone().requires_grad_(True), rgb.clone().requires_grad_(True)
    loss1, parts1, flag1, img1, inv1, w1, zs1 = _CompositeLossFn.apply(s1, c1, z, zl, bg, gt, lo, rng, lam_dist, lam_depth)
    (loss1 * upstream).backward()
    assert torch.equal(img1, img0.detach()) and torch.equal(w1, w0.detach()) and torch.equal(zs1, zs0) and torch.equal(inv1, inv0)
    assert torch.equal(s1.grad, s0.grad) and torch.equal(c1.grad, c0.grad)
    assert float(s0.grad.abs().max()) > 0 and float(c0.grad.abs().max()) > 0
    torch.testing.assert_close(loss1, loss0.detach(), rtol=2e-6, atol=1e-9)
    torch.testing.assert_close(parts1, parts0, rtol=2e-6, atol=1e-9)
    assert float(flag1) == 0.0 and not img1.requires_grad and not w1.requires_grad


def test_fused_tail_raises_the_nonfinite_flag():
    from nerf_amd.renderer import _CompositeLossFn
    sig, rgb, z, zl, bg, gt = _inputs(64, 128, 6)
    gt[5, 1] = float('nan')
    loss, parts, flag, *_ = _CompositeLossFn.apply(sig, rgb, z, zl, bg, gt, math.log10(1.6), 0.477, 1e-3, 1e-1)
    assert not torch.isfinite(loss) and float(flag) != 0.0


def test_render_loss_matches_render_plus_helpers():
    """NerfRenderer.render_loss (what Trainer.train_step and bench.py call) against render() + helpers.fused_train_loss on a
    full training step: same loss, same parameter gradients."""
    from nerf_amd import helpers
    from nerf_amd.renderer import NerfRenderer
    from test_gpu_parity import NET_CFGS, _batch, _make_pair
    out = []
    for fused in (True, False):
        net, _ = _make_pair(NET_CFGS['small'])
        R = NerfRenderer(net, 2, 2.5, [1.6, 4.8], [64], 128, 0.2)
        n, h, w, K, E, bg, noise, gt = [t.cuda() for t in _batch(9, 96)]
        if fused:
            loss, parts, img, wts, zs, aux = R.render_loss(n, h, w, K, E, bg, gt, 1e-3, 1e-1, noise=noise)
            assert aux['x_hashtable'] is not None and hasattr(loss, 'nonfinite_flag')
        else:
            img, wts, zs, aux = R.render(n, h, w, K, E, bg, noise=noise)
            loss, parts = helpers.fused_train_loss(img, gt, wts, zs, 1e-3, 1e-1)
        loss.backward()
        out.append((loss.detach(), parts, img.detach(), [p.grad.clone() for p in net.parameters() if p.grad is not None]))
    (l1, p1, i1, g1), (l0, p0, i0, g0) = out
    assert torch.equal(i1, i0)
    torch.testing.assert_close(l1, l0, rtol=2e-6, atol=1e-9)
    torch.testing.assert_close(p1, p0, rtol=2e-6, atol=1e-9)
    assert len(g1) == len(g0) > 0
    for a, b in zip(g1, g0):        # atomics: summation order differs from run to run
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6 * float(b.abs().max()) + 1e-12)
