"""nerf_mlp_bwd_noalloc (the entry point that allocates nothing: its workspace is a static device array inside
libnerf_hip.so) against nerf_mlp_bwd with a caller-allocated workspace, through the C ABI.  Two calls in a row on the
same GPU, the second with a smaller input (the shared array keeps the first call's contents), then the size limit.
Bar: dx bit-equal (one thread per row, no atomics); dparams bit-equal when the rows fit one 2048-row chunk, 1e-5 of
the largest entry otherwise (the chunks' partial sums meet in float atomics, whose order varies between launches)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(lib_, rows, widths, seed, noalloc):
    from nerf_amd._lib import check, ptr, stream
    g = torch.Generator().manual_seed(seed)
    n_par = sum(widths[i] * widths[i + 1] for i in range(len(widths) - 1))
    params = (torch.rand(n_par, generator=g) - 0.5).cuda()
    x = torch.randn(rows, widths[0], generator=g).cuda()
    dout = torch.randn(rows, widths[-1], generator=g).cuda()
    dx = torch.empty_like(x)
    dparams = torch.zeros_like(params)
    dims = (C.c_int32 * len(widths))(*widths)
    if noalloc:
        check(lib_.nerf_mlp_bwd_noalloc(ptr(params), dims, len(widths) - 1, ptr(x), ptr(dout), rows, ptr(dx), ptr(dparams),
                                        stream()), 'nerf_mlp_bwd_noalloc')
    else:
        n_ws = lib_.nerf_mlp_bwd_workspace_floats(dims, len(widths) - 1, rows)
        ws = torch.full((int(n_ws),), float('nan'), device='cuda')
        check(lib_.nerf_mlp_bwd(ptr(params), dims, len(widths) - 1, ptr(x), ptr(dout), rows, ptr(dx), ptr(dparams), ptr(ws),
                                ws.numel(), stream()), 'nerf_mlp_bwd')
    torch.cuda.synchronize()
    return dx, dparams


def test_noalloc_matches_workspace_variant_twice():
    from nerf_amd import _lib
    lib_ = _lib.lib()
    widths = [79, 64, 64, 3]                    # the colour net (nets.py:96-106)
    # the warm-up a capturing caller makes once per device: no rows, no pointers, nothing launched
    assert lib_.nerf_mlp_bwd_noalloc(None, (C.c_int32 * 4)(*widths), 3, None, None, 0, None, None, None) == 0
    for rows, seed in ((5000, 1), (777, 2)):    # the second, smaller call runs on the first one's leftovers
        dx_a, dp_a = _run(lib_, rows, widths, seed, noalloc=True)
        dx_b, dp_b = _run(lib_, rows, widths, seed, noalloc=False)
        assert torch.equal(dx_a, dx_b)
        if rows <= 2048:
            assert torch.equal(dp_a, dp_b)
        else:
            torch.testing.assert_close(dp_a, dp_b, rtol=0, atol=1e-5 * float(dp_b.abs().max()))
    # and the sigma net's shape right after, again smaller
    dx_a, dp_a = _run(lib_, 300, [40, 64, 16], 3, noalloc=True)
    dx_b, dp_b = _run(lib_, 300, [40, 64, 16], 3, noalloc=False)
    assert torch.equal(dx_a, dx_b) and torch.equal(dp_a, dp_b)


def test_noalloc_rejects_oversized_input():
    from nerf_amd import _lib
    from nerf_amd._lib import ptr, stream
    lib_ = _lib.lib()
    widths = [79, 64, 64, 3]
    dims = (C.c_int32 * 4)(*widths)
    rows = 20000                                # 20000 x 515 floats > the 18 MiB array
    x = torch.zeros(rows, 79, device='cuda')
    dout = torch.zeros(rows, 3, device='cuda')
    params = torch.zeros(79 * 64 + 64 * 64 + 64 * 3, device='cuda')
    dparams = torch.zeros_like(params)
    rc = lib_.nerf_mlp_bwd_noalloc(ptr(params), dims, 3, ptr(x), ptr(dout), rows, None, ptr(dparams), stream())
    assert rc == -1 and b'at most' in lib_.nerf_last_error()
