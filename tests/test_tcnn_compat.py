"""nerf_amd.tcnn_compat: padding / stripping of FullyFusedMLP parameters (tiny-cuda-nn's published layout; UNPINNED against
a real reference checkpoint, see the module docstring)."""
import pytest
import torch

from nerf_amd import tcnn_compat as tc


def test_strip_and_pad_round_trip():
    g = torch.Generator().manual_seed(0)
    for widths in ([40, 64, 16], [79, 64, 64, 3], [32, 64, 64, 64, 16], [16, 64, 16]):
        n = sum(widths[i + 1] * widths[i] for i in range(len(widths) - 1))
        flat = torch.randn(n, generator=g)
        padded = tc.add_padding(flat, widths)
        assert padded.numel() == tc.padded_numel(widths)
        assert torch.equal(tc.strip_padding(padded, widths), flat)
        assert torch.equal(tc.strip_padding(padded.to(torch.float16), widths), flat.to(torch.float16).to(torch.float32))
        # the padding is zero and sits where tiny-cuda-nn puts it: extra INPUT columns of layer 0, extra OUTPUT rows of the last
        first = padded[:widths[1] * tc._pad16(widths[0])].view(widths[1], -1)
        assert float(first[:, widths[0]:].abs().sum()) == 0.0
        last = padded[-tc._pad16(widths[-1]) * widths[-2]:].view(-1, widths[-2])
        assert float(last[widths[-1]:].abs().sum()) == 0.0
    assert tc.padded_numel([40, 64, 16]) == 48 * 64 + 64 * 16           # sigma net: 4 096
    assert tc.padded_numel([79, 64, 64, 3]) == 80 * 64 + 64 * 64 + 16 * 64   # colour net: 10 240
    with pytest.raises(ValueError):
        tc.strip_padding(torch.zeros(5), [40, 64, 16])


@pytest.mark.gpu
def test_padded_state_dict_loads_and_evaluates_identically():
    from test_gpu_parity import NET_CFGS, _make_pair
    net, _ = _make_pair(NET_CFGS['small'])
    other, _ = _make_pair(NET_CFGS['small'], seed=11)
    ref_sd = tc.to_reference_state_dict(net, dtype=torch.float32)
    assert ref_sd['color_net.params'].numel() == tc.padded_numel(net.color_net.widths) != net.color_net.params.numel()
    other.load_state_dict(tc.from_reference_state_dict(ref_sd, other))
    g = torch.Generator().manual_seed(1)
    x = ((torch.rand(500, 3, generator=g) * 2 - 1) * 2.4).cuda()
    d = torch.nn.functional.normalize(torch.randn(500, 3, generator=g), dim=-1).cuda()
    n = torch.randint(0, 7, (500,), generator=g).cuda()
    with torch.no_grad():
        s0, c0, _ = net(x, d, n, 2.5)
        s1, c1, _ = other(x, d, n, 2.5)
    assert torch.equal(s0, s1) and torch.equal(c0, c1)
