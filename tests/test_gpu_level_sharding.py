"""Level-sharded table, single process: the cut entry points (encode a level subset / MLPs from ready features /
scatter a level subset) must reproduce the fused kernels they were cut from -- bit-exactly in the forward
direction (same arithmetic in the same order), to atomic-ordering noise in the scatter."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

from test_gpu_parity import NET_CFGS, _make_pair  # noqa: E402


def _rays(M_rays, S, seed):
    g = torch.Generator().manual_seed(seed)
    ro = (torch.rand(M_rays, 3, generator=g) * 2 - 1) * 1.5
    rd = torch.randn(M_rays, 3, generator=g)
    z = torch.sort(torch.rand(M_rays, S, generator=g) * 3 + 0.3, dim=-1).values
    n = torch.randint(0, 7, (M_rays,), generator=g)
    return ro.cuda(), rd.cuda(), z.cuda().contiguous(), n.cuda()


@pytest.mark.parametrize('name', list(NET_CFGS))
@pytest.mark.parametrize('world', [2, 3])
def test_cut_entry_points_match_fused(name, world):
    from nerf_amd._lib import check, lib, ptr, stream
    from nerf_amd.level_parallel import LevelSharding
    from nerf_amd.nets import make_point_src
    net, _ = _make_pair(NET_CFGS[name])
    cfg, ncfg = net.encoder.cfg, net.net_cfg
    L, F = cfg.n_levels, cfg.n_features
    R, S = 37, 13                                    # ragged: 481 points
    ro, rd, z, n = _rays(R, S, 5)
    M = R * S
    kw = dict(outer_bound=2.5, inner_bound=2.0, rays_o=ro, rays_d=rd, z=z, z_stride=S, n_samples=S, n_points=M, n=n)
    # fused reference
    sig_f, rgb_f, xh_f, xyz_f = net._field(kw, want_xyz=True)
    sig_d, xh_d = net.density_src(make_point_src(**kw), want_features=True)
    assert torch.equal(xh_f, xh_d)
    # every "rank" encodes its levels; assemble like exchange_features does
    shards = [LevelSharding(cfg, 'cuda', rank=r, world=world) for r in range(world)]
    assert sorted(l for s in shards for l in s.mine) == list(range(L))
    src = make_point_src(**kw)
    chunks = []
    for s in shards:
        w = s.width[s.rank]
        out = torch.full((M, max(w, 1)), float('nan'), device='cuda')
        if w:
            check(lib().nerf_hashgrid_encode_levels(C.byref(cfg), ptr(net.encoder.params), C.byref(src), C.byref(s.my_set),
                                                    ptr(out), w, stream()), 'encode_levels')
        chunks.append(out[:, :w])
    xh = torch.cat(chunks, 1).index_select(1, shards[0].inv).contiguous()
    assert torch.equal(xh, xh_f.detach())
    # MLP halves from ready features
    sig2 = torch.empty(M, device='cuda')
    check(lib().nerf_density_from_features(C.byref(cfg), C.byref(ncfg), ptr(net.sigma_net.params), ptr(xh), M, ptr(sig2),
                                           stream()), 'density_from_features')
    assert torch.equal(sig2, sig_d)
    sig3, rgb3, xyz3 = torch.empty(M, device='cuda'), torch.empty(M, 3, device='cuda'), torch.empty(M, 3, device='cuda')
    check(lib().nerf_field_from_features(C.byref(cfg), C.byref(ncfg), ptr(net.sigma_net.params), ptr(net.color_net.params),
                                         ptr(net.latent_emb), net.N, C.byref(src), ptr(xh), ptr(sig3), ptr(rgb3), ptr(xyz3),
                                         stream()), 'field_from_features')
    assert torch.equal(sig3, sig_f.detach()) and torch.equal(rgb3, rgb_f.detach()) and torch.equal(xyz3, xyz_f)
    # scatter of level subsets == the per-level slices of the whole scatter
    dfeat = torch.randn(M, L * F, device='cuda')
    whole = torch.zeros_like(net.encoder.params)
    check(lib().nerf_hashgrid_scatter(C.byref(cfg), C.byref(src), ptr(dfeat), ptr(whole), 0, L, stream()), 'scatter')
    parts = torch.zeros_like(net.encoder.params)
    for s in shards:
        w = s.width[s.rank]
        if not w:
            continue
        mine = dfeat.index_select(1, s.perm).split(s.width, 1)[s.rank].contiguous()
        check(lib().nerf_hashgrid_scatter_levels(C.byref(cfg), C.byref(src), C.byref(s.my_set), ptr(mine), w, ptr(parts),
                                                 stream()), 'scatter_levels')
    torch.testing.assert_close(parts, whole, rtol=1e-5, atol=1e-5 * float(whole.abs().max()))


def test_level_entry_points_reject_bad_arguments():
    from nerf_amd._lib import NerfLevelSet, lib, ptr, stream
    from nerf_amd.nets import make_point_src
    net, _ = _make_pair(NET_CFGS['odd'])
    cfg = net.encoder.cfg
    x = torch.rand(8, 3, device='cuda')
    src = make_point_src(2.5, x=x, n_points=8)
    out = torch.empty(8, 4, device='cuda')
    bad = NerfLevelSet()
    bad.count, bad.level[0] = 1, cfg.n_levels          # level out of range
    assert lib().nerf_hashgrid_encode_levels(C.byref(cfg), ptr(net.encoder.params), C.byref(src), C.byref(bad), ptr(out), 4,
                                             stream()) == -1
    ok = NerfLevelSet()
    ok.count, ok.level[0], ok.level[1] = 2, 0, 3
    assert lib().nerf_hashgrid_encode_levels(C.byref(cfg), ptr(net.encoder.params), C.byref(src), C.byref(ok), ptr(out), 3,
                                             stream()) == -1      # stride < count*F
    assert lib().nerf_hashgrid_encode_levels(C.byref(cfg), ptr(net.encoder.params), C.byref(src), C.byref(ok), ptr(out), 4,
                                             stream()) == 0
    torch.cuda.synchronize()


@pytest.mark.parametrize('name', ['small', 'base19'])
def test_lds_privatised_scatter_matches_global_atomic_path(name):
    """Calls with >= 65536 points send the small dense levels through the LDS-privatised kernel (one and several
    chunks per level); the same points in pieces below that bar take the global-atomic kernel.  Same sums."""
    from nerf_amd._lib import check, lib, ptr, stream
    from nerf_amd.nets import make_point_src
    net, _ = _make_pair(NET_CFGS[name])
    cfg = net.encoder.cfg
    L, F = cfg.n_levels, cfg.n_features
    R, S = 1171, 64                                   # 74944 points, ragged against 1024-thread blocks
    ro, rd, z, _ = _rays(R, S, 9)
    M = R * S
    dfeat = torch.randn(M, L * F, device='cuda')
    src = make_point_src(2.5, rays_o=ro, rays_d=rd, z=z, z_stride=S, n_samples=S, n_points=M, inner_bound=2.0)
    got = torch.zeros_like(net.encoder.params)
    check(lib().nerf_hashgrid_scatter(C.byref(cfg), C.byref(src), ptr(dfeat), ptr(got), 0, L, stream()), 'scatter')
    want = torch.zeros_like(net.encoder.params)
    step = 500                                        # rays per piece: 32000 points
    for a in range(0, R, step):
        b = min(R, a + step)
        part = make_point_src(2.5, rays_o=ro[a:b].contiguous(), rays_d=rd[a:b].contiguous(), z=z[a:b].contiguous(),
                              z_stride=S, n_samples=S, n_points=(b - a) * S, inner_bound=2.0)
        check(lib().nerf_hashgrid_scatter(C.byref(cfg), C.byref(part), ptr(dfeat[a * S:b * S].contiguous()), ptr(want), 0, L,
                                          stream()), 'scatter')
    small = [l for l in range(L) if not cfg.hashed[l] and cfg.entries[l] * F <= 4 * 32768]
    assert small, 'configuration has no small dense level'
    torch.testing.assert_close(got, want, rtol=1e-4, atol=2e-5 * float(want.abs().max()))
    # and in points mode through the stand-alone encoding backward
    x01 = torch.rand(M, 3, device='cuda')
    got2, want2 = torch.zeros_like(got), torch.zeros_like(got)
    check(lib().nerf_hashgrid_bwd(C.byref(cfg), ptr(x01), ptr(dfeat), M, ptr(got2), stream()), 'bwd')
    for a in range(0, M, 30000):
        b = min(M, a + 30000)
        check(lib().nerf_hashgrid_bwd(C.byref(cfg), ptr(x01[a:b].contiguous()), ptr(dfeat[a:b].contiguous()), b - a,
                                      ptr(want2), stream()), 'bwd')
    torch.testing.assert_close(got2, want2, rtol=1e-4, atol=2e-5 * float(want2.abs().max()))


@pytest.mark.parametrize('world', [2, 3, 8])
def test_feature_shuffle_matches_torch_permutation(world):
    """nerf_feature_shuffle (one pass) == the cat / index_select / split it replaces, in both directions."""
    from nerf_amd._lib import check, lib, ptr, stream
    from nerf_amd.level_parallel import LevelSharding
    net, _ = _make_pair(NET_CFGS['base19'])
    cfg = net.encoder.cfg
    LF, M = cfg.n_levels * cfg.n_features, 777
    for rank in (0, world - 1):
        sh = LevelSharding(cfg, 'cuda', rank=rank, world=world)
        recv = torch.randn(M * LF, device='cuda')
        chunks, o = [], 0
        for w in sh.width:
            chunks.append(recv[o:o + M * w].view(M, w))
            o += M * w
        want = torch.cat(chunks, 1).index_select(1, sh.inv)
        got = torch.empty(M, LF, device='cuda')
        check(lib().nerf_feature_shuffle(ptr(recv), ptr(got), M, LF, C.byref(sh.col_map(M)), 0, stream()), 'shuffle')
        assert torch.equal(got, want)
        back = torch.empty(M * LF, device='cuda')
        check(lib().nerf_feature_shuffle(ptr(got), ptr(back), M, LF, C.byref(sh.col_map(M)), 1, stream()), 'shuffle')
        assert torch.equal(back, recv)
        want_send = torch.cat([c.reshape(-1) for c in got.index_select(1, sh.perm).split(sh.width, 1)])
        assert torch.equal(back, want_send)
