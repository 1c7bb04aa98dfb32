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
