"""Stratification noise drawn inside the resampling kernel (nerf_coarse_to_fine_rng, Philox4x32-10) instead of
torch.rand (renderer.py:102): the stream itself, its equivalence with the explicit-noise entry point, and the renderer's
seed / counter bookkeeping."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_uniform_noise_stream():
    from nerf_amd.renderer import NerfRenderer
    from test_gpu_parity import NET_CFGS, _make_pair
    net, _ = _make_pair(NET_CFGS['small'])
    R = NerfRenderer(net, 2, 2.5, [1.6, 4.8], [64], 128, 0.2)
    a = R.uniform_noise(1234, 7, 4096, 128)
    assert float(a.min()) >= 0.0 and float(a.max()) < 1.0
    assert abs(float(a.mean()) - 0.5) < 2e-3 and abs(float(a.var()) - 1 / 12) < 2e-3
    # 16 equal bins within 5 sigma of uniform; neighbouring elements uncorrelated
    hist = torch.histc(a, bins=16, min=0, max=1)
    n = a.numel()
    assert float((hist - n / 16).abs().max()) < 5 * (n / 16 * 15 / 16) ** 0.5
    f = a.flatten()
    assert abs(float(torch.corrcoef(torch.stack([f[:-1], f[1:]]))[0, 1])) < 5e-3
    assert torch.equal(a, R.uniform_noise(1234, 7, 4096, 128))                    # a pure function of (seed, counter)
    assert torch.equal(a.flatten()[:1000], R.uniform_noise(1234, 7, 1000, 1).flatten())   # ... and of the element index
    assert not torch.equal(a, R.uniform_noise(1234, 8, 4096, 128)) and not torch.equal(a, R.uniform_noise(1235, 7, 4096, 128))


def test_in_kernel_noise_equals_explicit_noise():
    from nerf_amd.renderer import NerfRenderer
    from test_gpu_parity import NET_CFGS, _batch, _make_pair
    net, _ = _make_pair(NET_CFGS['small'])
    with torch.no_grad():
        net.encoder.params.mul_(1e3)        # some structure in the coarse density
    R = NerfRenderer(net, 2, 2.5, [1.6, 4.8], [64], 128, 0.2)
    n, h, w, K, E, bg, _noise, gt = [t.cuda() for t in _batch(5, 301)]
    torch.manual_seed(99)
    with torch.no_grad():
        img1, w1, zs1, aux1 = R.render(n, h, w, K, E, bg)            # draw 1 of seed 99
        seed, counter = R.last_noise_stream
        img2, _, zs2, _ = R.render(n, h, w, K, E, bg)               # draw 2: other noise
        torch.manual_seed(99)
        img3, _, zs3, _ = R.render(n, h, w, K, E, bg)               # reseeded: draw 1 again
        img4, _, zs4, _ = R.render(n, h, w, K, E, bg, noise=R.uniform_noise(seed, counter, 301, 128))     # the same noise, handed in
    assert torch.equal(zs1, zs3) and torch.equal(img1, img3)
    assert torch.equal(zs1, zs4) and torch.equal(img1, img4)
    assert not torch.equal(zs1, zs2)
    assert bool((zs1[:, 1:] >= zs1[:, :-1]).all())


def test_resampling_a_slice_of_the_batch_draws_the_batch_noise():
    """nerf_coarse_to_fine_rng_at: rays [first, first + n) of a batch, resampled alone, get the depths one launch over the
    whole batch gives them, bit for bit (the ray groups of the level-sharded step resample one after the other)."""
    from nerf_amd._lib import check, lib, ptr, stream
    from nerf_amd.renderer import NerfRenderer
    from oracle.ref_cpu import AnalyticField
    R = NerfRenderer(AnalyticField(), 2, 2.5, [1.6, 4.8], [64], 128, 0.2)
    N, Sc, Sf = 301, 64, 127                      # odd Sf: the pair draws of the kernel straddle Philox blocks
    g = torch.Generator().manual_seed(3)
    sig = (torch.rand(N, Sc, generator=g) * 4).cuda()
    zlog_c, _ = R._coarse_table(sig.device)
    lin, steps = R._lin_table(Sf, sig.device), R._steps_array(R.steps_firstpass)

    def run(first, n, at):
        zl, z = torch.empty(n, Sf, device='cuda'), torch.empty(n, Sf, device='cuda')
        s = sig[first:first + n]
        if at:
            check(lib().nerf_coarse_to_fine_rng_at(ptr(s), ptr(zlog_c), 0, 77, 5, first, ptr(lin), n, Sc, Sf, 0.2, steps, 1, ptr(zl),
                                                   ptr(z), stream()), 'nerf_coarse_to_fine_rng_at')
        else:
            check(lib().nerf_coarse_to_fine_rng(ptr(s), ptr(zlog_c), 0, 77, 5, ptr(lin), n, Sc, Sf, 0.2, steps, 1, ptr(zl), ptr(z),
                                                stream()), 'nerf_coarse_to_fine_rng')
        return zl, z
    whole = run(0, N, False)
    assert torch.equal(run(0, N, True)[0], whole[0])
    for first, n in ((0, 100), (100, 101), (201, 100), (300, 1)):
        part = run(first, n, True)
        assert torch.equal(part[0], whole[0][first:first + n]) and torch.equal(part[1], whole[1][first:first + n]), (first, n)
    if N > 100:
        assert not torch.equal(run(100, 101, False)[0], whole[0][100:201])        # without the offset: other noise
    assert lib().nerf_coarse_to_fine_rng_at(ptr(sig), ptr(zlog_c), 0, 77, 5, -1, ptr(lin), 4, Sc, Sf, 0.2, steps, 1,
                                            ptr(whole[0]), ptr(whole[1]), stream()) < 0
