"""Torch-only parts of nerf_amd.allign against the oracle's line-by-line restatement of allign/trainer.py:174-208,
allign/allign2.py:35-56 and allign/rotation.py:8-27 (CPU)."""
import torch

from nerf_amd import allign
from oracle import ref_cpu


def test_pair_distance_prefix_form_equals_outer_product():
    g = torch.Generator().manual_seed(0)
    wa, wb = torch.rand(9, 37, generator=g), torch.rand(9, 37, generator=g)
    s = torch.sort(torch.rand(9, 37, generator=g), dim=-1).values
    s[3, 5:9] = s[3, 5]                                                       # ties
    want = torch.mean(torch.sum(torch.bmm(wa[:, :, None], wb[:, None, :]) * torch.abs(s[:, :, None] - s[:, None, :]), dim=[1, 2]))
    got = allign.pair_distance_sorted(wa, wb, s)
    assert abs(float(got) - float(want)) < 1e-5 * float(want)


def test_transform_and_contraction_match_restatement():
    r = torch.tensor([0.3, -0.2, 0.1])
    torch.testing.assert_close(allign.Exp(r), ref_cpu.so3_exp(r))
    torch.testing.assert_close(allign.Exp(torch.zeros(3)), torch.eye(3))
    init = torch.eye(4)
    init[:3, :3] = ref_cpu.so3_exp(torch.tensor([0.1, 0.2, -0.3]))
    init[:3, 3] = torch.tensor([0.5, -1.0, 0.25])
    tr = allign.Transform(init)
    with torch.no_grad():
        tr.R.copy_(r)
        tr.T.copy_(torch.tensor([0.1, 0.2, 0.3]))
    x = torch.randn(4, 6, 3, generator=torch.Generator().manual_seed(1))
    torch.testing.assert_close(tr(x), ref_cpu.rigid_transform(x, init, tr.R.detach(), tr.T.detach()), rtol=1e-5, atol=1e-6)
    R = ref_cpu.OracleRenderer(None, 2, 2.5, [1.6, 4.8], [64], 128, 0.2)
    pts = torch.cat([x * 2, torch.zeros(4, 1, 3)], 1).requires_grad_(True)    # includes r = 0 and r > inner
    out = allign.contract(pts, 2, 2.5)
    torch.testing.assert_close(out, R.mipnerf360_scale(pts.detach(), 2, 2.5))
    out.sum().backward()
    assert bool(torch.isfinite(pts.grad).all())
