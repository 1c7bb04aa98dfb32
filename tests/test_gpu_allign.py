"""Input gradient of density() and the pose-alignment consistency step (allign/trainer.py:174-208) against the CPU
oracle (torch autograd through the restated network; tinycudann absent => parity unpinned at the network boundary,
as for the rest of the field tests)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_gpu_parity import NET_CFGS, _batch, _make_pair  # noqa: E402


@pytest.mark.parametrize('name', list(NET_CFGS))
def test_density_input_gradient_vs_oracle(name):
    net, ora = _make_pair(NET_CFGS[name])
    g = torch.Generator().manual_seed(21)
    M = 700
    x = ((torch.rand(M, 3, generator=g) * 2 - 1) * 2.4).requires_grad_(True)
    gs = torch.randn(M, generator=g)
    sig_o, _ = ora.density(x, 2.5)
    (sig_o * gs).sum().backward()
    xg = x.detach().cuda().requires_grad_(True)
    sig_g, aux = net.density(xg, 2.5)
    assert sig_g.requires_grad and not aux['x_hashtable'].requires_grad
    (sig_g * gs.cuda()).sum().backward()
    torch.testing.assert_close(sig_g.detach().cpu(), sig_o.detach(), rtol=1e-4, atol=1e-5 * float(sig_o.abs().max()))
    scale = float(x.grad.abs().max())
    assert scale > 0
    torch.testing.assert_close(xg.grad.cpu(), x.grad, rtol=1e-3, atol=2e-5 * scale)
    # without grad the fast path is taken and nothing is recorded
    with torch.no_grad():
        s2, _ = net.density(xg, 2.5)
    assert not s2.requires_grad and torch.equal(s2, sig_g.detach())


def test_scene_consistency_vs_oracle():
    from nerf_amd import allign
    from nerf_amd.renderer import NerfRenderer
    from oracle import ref_cpu
    net_a, ora_a = _make_pair(NET_CFGS['small'], seed=3)
    net_b, ora_b = _make_pair(NET_CFGS['small'], seed=5)
    n, h, w, K, E, _, noise, _ = _batch(4, 96)
    init = torch.eye(4)
    init[:3, :3] = ref_cpu.so3_exp(torch.tensor([0.02, -0.03, 0.01]))
    init[:3, 3] = torch.tensor([0.05, -0.02, 0.03])
    R = torch.tensor([0.01, 0.02, -0.015], requires_grad=True)
    T = torch.tensor([-0.01, 0.03, 0.02], requires_grad=True)
    Ro = ref_cpu.OracleRenderer(ora_a, 2, 2.5, [1.6, 4.8], [64], 128, 0.2)
    loss_o = ref_cpu.scene_consistency(Ro, ora_a, ora_b, init, R, T, h, w, K, E, noise=noise)
    loss_o.backward()

    Rg = NerfRenderer(net_a, 2, 2.5, [1.6, 4.8], [64], 128, 0.2)
    tr = allign.Transform(init.cuda())
    with torch.no_grad():
        tr.R.copy_(R.detach())
        tr.T.copy_(T.detach())
    loss_g = allign.scene_consistency(Rg, net_a, net_b, tr, h.cuda(), w.cuda(), K.cuda(), E.cuda(), noise=noise.cuda())
    loss_g.backward()
    assert abs(float(loss_g) - float(loss_o)) <= 2e-4 * max(1e-3, abs(float(loss_o))), (float(loss_g), float(loss_o))
    for got, want, name in ((tr.R.grad, R.grad, 'R'), (tr.T.grad, T.grad, 'T')):
        s = float(want.abs().max())
        torch.testing.assert_close(got.cpu(), want, rtol=5e-3, atol=5e-3 * s, msg=lambda m, n=name: n + ': ' + m)
    assert net_a.encoder.params.grad is None and net_b.encoder.params.grad is None        # the fields stay frozen
