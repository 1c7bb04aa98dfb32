"""Multi-process logic on CPU (gloo, world_size 2): the data-parallel gradient path, the
inference slice gather and the variable-length extraction gather.  The gradient PRODUCER here is
the CPU oracle (tests may use it); what is under test is nerf_amd.parallel."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)


def _batch(N, seed):
    from nerf_amd.loaders.procedural import sphere_cameras
    K_all, E_all = sphere_cameras(5, 0)
    g = torch.Generator().manual_seed(seed)
    n = torch.randint(0, 5, (N,), generator=g)
    h = torch.randint(0, 800, (N,), generator=g)
    w = torch.randint(0, 800, (N,), generator=g)
    return n, h, w, K_all[n], E_all[n], torch.rand(3, generator=g), torch.rand(N, 128, generator=g), torch.rand(N, 3, generator=g)


def _oracle(seed=2):
    from oracle import ref_cpu
    kw = dict(encoding_n_levels=4, encoding_log2_hashmap_size=12, latent_embedding_dim=48)
    net = ref_cpu.OracleNetwork(5, seed=seed, **kw)
    with torch.no_grad():
        net.encoder_params.mul_(1e3)
    return net, ref_cpu.OracleRenderer(net, 2, 2.5, [1.6, 4.8], [64], 128, 0.2)


def _grads(net, R, batch):
    from oracle import ref_cpu
    n, h, w, K, E, bg, noise, gt = batch
    net.zero_grad()
    img, wts, zs, _ = R.render(n, h, w, K, E, bg, noise=noise)
    loss, _ = ref_cpu.train_loss(img, gt, wts, zs, 3, 100, 1e-5, 1e-2, 1e-1, 1e-1)
    loss.backward()
    return [p.grad.clone() for p in net.parameters()]


def _dp_worker(rank, world, port, q):
    from nerf_amd import parallel
    _init(rank, world, port)
    net, R = _oracle()
    full = _batch(32, 7)
    a, b = parallel.shard_range(32, rank, world)
    mine = tuple(t[a:b] if (torch.is_tensor(t) and t.dim() > 0 and t.shape[0] == 32) else t for t in full)
    _grads(net, R, mine)
    params = list(net.parameters())
    flag = torch.tensor([float(rank)], dtype=torch.float64)    # the non-finite flag rides in the same collective
    parallel.all_reduce_grads(params, flags=[flag])
    assert float(flag) == sum(range(world)) and flag.dtype == torch.float64
    out = [p.grad / world for p in params]                 # the 1/world factor the fused Adam applies
    if rank == 0:
        q.put([o.numpy() for o in out])
    dist.barrier()
    dist.destroy_process_group()


class _FakeSharding(object):
    """Stands in for LevelSharding in backward_and_reduce: fires the hook from inside the backward pass."""
    after_last_backward = None


def _early_reduce_worker(rank, world, port, q):
    from nerf_amd import parallel
    _init(rank, world, port)
    torch.manual_seed(3)
    w_ = torch.nn.Parameter(torch.randn(5, 3))
    b_ = torch.nn.Parameter(torch.randn(3))
    big = torch.nn.Parameter(torch.randn(4))        # stands for the table: reduced elsewhere, skipped here
    x = torch.arange(10.0).reshape(2, 5) + rank
    sh = _FakeSharding()

    class Model(object):
        level_sharding, direct_grads = sh, True

        def _sharded(self):
            return True

    class Fire(torch.autograd.Function):            # the place _ShardedFieldFn.backward calls the hook from
        @staticmethod
        def forward(ctx, t):
            return t.clone()

        @staticmethod
        def backward(ctx, g):
            w_.grad = x.t() @ g                     # "direct gradients": written by the backward pass itself
            b_.grad = g.sum(0)
            fn, sh.after_last_backward = sh.after_last_backward, None
            assert fn is not None
            fn()
            return None

    seen = []
    orig = parallel.all_reduce_grads
    parallel.all_reduce_grads = lambda *a, **k: (seen.append(k.get('async_op', False)), orig(*a, **k))[1]
    out = Fire.apply(x @ w_.detach().requires_grad_() + b_.detach())
    big.grad = torch.ones(4) * (rank + 1)
    flag = torch.tensor([float(rank)])
    parallel.backward_and_reduce(out.sum(), Model(), [w_, b_, big], skip=lambda: [big], flags=[flag])
    assert seen == [True], seen                     # queued once, asynchronously, from inside the backward pass
    assert sh.after_last_backward is None and float(flag) == sum(range(world))
    assert torch.equal(big.grad, torch.ones(4) * (rank + 1))        # skipped
    want_w = sum((torch.arange(10.0).reshape(2, 5) + r).t() @ torch.ones(2, 3) for r in range(world))
    assert torch.allclose(w_.grad, want_w) and torch.allclose(b_.grad, torch.full((3,), 2.0 * world))
    # without a sharded model: plain backward, then the synchronous reduce
    seen.clear()
    v = torch.nn.Parameter(torch.ones(3) * (rank + 1))
    parallel.backward_and_reduce((v * v).sum(), object(), [v], flags=[])
    assert seen == [False] and torch.allclose(v.grad, torch.full((3,), 2.0 * sum(r + 1 for r in range(world))))
    if rank == 0:
        q.put('ok')
    dist.barrier()
    dist.destroy_process_group()


def test_backward_and_reduce_queues_the_small_collective_from_the_backward_pass():
    """parallel.backward_and_reduce over gloo, two ranks: the hook fires inside backward(), the collective is asynchronous,
    `skip` is evaluated late, the flag is summed with the gradients; a model without level sharding takes the plain path."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_early_reduce_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    assert q.get(timeout=300) == 'ok'
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)


def test_data_parallel_gradients_match_single_process():
    """mean-over-rays loss: all-reduced (SUM) shard gradients / world == full-batch gradients."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    got = q.get(timeout=300)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    net, R = _oracle()
    want = _grads(net, R, _batch(32, 7))
    for g, w in zip(got, want):
        scale = float(w.abs().max()) + 1e-12
        torch.testing.assert_close(torch.from_numpy(g), w, rtol=1e-4, atol=1e-5 * scale)


def _gather_worker(rank, world, port, q):
    from nerf_amd import parallel
    _init(rank, world, port)
    total = 11                                             # ragged: 6 + 5
    full = torch.arange(total * 3, dtype=torch.float32).reshape(total, 3)
    a, b = parallel.shard_range(total, rank, world)
    got = parallel.gather_slices(full[a:b].clone(), total)
    rows = full[full[:, 0] % 2 == rank]                    # different row counts per rank (incl. possibly none)
    var = parallel.gather_varlen(rows.clone())
    empty = parallel.gather_varlen(torch.zeros(0, 3) if rank == 1 else full[:2].clone())
    if rank == 0:
        q.put((got.numpy(), var.numpy(), empty.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_slice_and_varlen_gathers():
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    got, var, empty = q.get(timeout=120)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    full = torch.arange(33, dtype=torch.float32).reshape(11, 3)
    assert torch.equal(torch.from_numpy(got), full)
    want_var = torch.cat([full[full[:, 0] % 2 == 0], full[full[:, 0] % 2 == 1]])
    assert torch.equal(torch.from_numpy(var), want_var)
    assert torch.equal(torch.from_numpy(empty), full[:2])


def test_shard_range_covers_everything():
    from nerf_amd.parallel import shard_range
    for total in (0, 1, 7, 640000, 134217728):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


# ---- level-sharded table: ownership and the two feature exchanges (torch-only part, gloo) ----
def _fake_cfg(L=7, F=2, log2T=10):
    from nerf_amd._lib import NerfGridCfg
    cfg = NerfGridCfg()
    cfg.n_levels, cfg.n_features, cfg.log2_hashmap_size = L, F, log2T
    off = 0
    for l in range(L):
        cfg.entries[l] = min(1 << log2T, (4 + 3 * l) ** 3)
        cfg.hashed[l] = 1 if (4 + 3 * l) ** 3 > (1 << log2T) else 0
        cfg.offset[l] = off
        off += cfg.entries[l]
    cfg.total_entries = off
    return cfg


def test_assign_levels_covers_and_balances():
    from nerf_amd.level_parallel import assign_levels, level_costs
    cfg = _fake_cfg(L=20, log2T=12)
    for world in (1, 2, 3, 4, 8):
        owner = assign_levels(cfg, world)
        assert len(owner) == 20 and set(owner) <= set(range(world))
        cost = level_costs(cfg, world)
        load = [sum(c for c, o in zip(cost, owner) if o == r) for r in range(world)]
        assert max(load) - min(load) <= max(cost) + 1e-9       # LPT bound


def _base_cfg(log2T):
    """config_base's 20-level grid (configs/config_base.yaml:78-83) through the library's own (host-only) level table."""
    import ctypes as C
    from nerf_amd._lib import NerfGridCfg, lib
    cfg = NerfGridCfg()
    assert lib().nerf_grid_cfg_init(C.byref(cfg), 20, 2, log2T, 16, 1.3819) > 0
    return cfg


def _exchange_worker(rank, world, port, q, grid='fake7'):
    import os
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from nerf_amd.level_parallel import LevelSharding
    cfg = _fake_cfg() if grid == 'fake7' else _base_cfg(int(grid[1:]))
    L, F, M = cfg.n_levels, cfg.n_features, 5
    sh = LevelSharding(cfg, 'cpu')
    # feature of (owner-of-points r, point p, level l, f) = 1000 r + 100 p + 10 l + f
    val = lambda r, p, l, f: 1000.0 * r + 100.0 * p + 10.0 * l + f
    send = torch.tensor([[[val(r, p, l, f) for l in sh.mine for f in range(F)] for p in range(M)] for r in range(world)])
    xh = sh.exchange_features(send.reshape(-1), M)
    want = torch.tensor([[val(rank, p, l, f) for l in range(L) for f in range(F)] for p in range(M)])
    ok_fwd = torch.equal(xh, want)
    back = sh.exchange_dfeat(want)           # every rank returns its own rows; owners get [world*M, width_mine]
    want_back = torch.tensor([[val(r, p, l, f) for l in sh.mine for f in range(F)] for r in range(world) for p in range(M)])
    ok_bwd = torch.equal(back, want_back.reshape(world * M, -1))
    # table sync: every level comes from its owner
    ok_sync = True
    if int(cfg.total_entries) * F <= (1 << 22):      # (the 2^24 grid is 400 M floats per rank: its sync is not rehearsed here)
        table = torch.full((int(cfg.total_entries) * F,), float(rank))
        sh.dirty = True
        sh.sync_table(table)
        ok_sync = all(bool((table[sh.level_slice(l)[0]:sh.level_slice(l)[1]] == sh.owner[l]).all()) for l in range(L))
    # ray groups (LevelSharding.plan): 5 rays x 3 samples per rank in groups of 3 + 2 rays; the gathered source and the
    # d(features) matrix the owners receive are group-major ([group][rank][ray])
    R, S = 5, 3
    ray = lambda r, i: 100.0 * r + i
    ro = torch.tensor([[ray(rank, i), 0.0, 1.0] for i in range(R)])
    z = torch.tensor([[ray(rank, i) + 0.1 * k for k in range(S)] for i in range(R)])
    kw = dict(outer_bound=2.0, inner_bound=1.0, rays_o=ro, rays_d=-ro, z=z, z_stride=S, n_samples=S, n_points=R * S,
              n=torch.arange(R))
    plan = sh.plan(kw, groups=2)
    order = [(r, i) for (ra, rb) in ((0, 3), (3, 5)) for r in range(world) for i in range(ra, rb)]
    want_o = torch.tensor([ray(r, i) for r, i in order])
    ok_plan = (torch.equal(plan['kw_all']['rays_o'][:, 0], want_o) and torch.equal(plan['kw_all']['z'][:, 0], want_o)
               and plan['kw_all']['n_points'] == world * R * S
               and [(g['a'], g['b']) for g in plan['groups']] == [(0, 9), (9, 15)]
               and torch.equal(plan['groups'][1]['kw_all']['rays_o'][:, 0], want_o[world * 3:])
               and torch.equal(plan['groups'][1]['kw_local']['n'], torch.arange(3, 5))
               and torch.equal(plan['groups'][1]['kw_local']['z'], z[3:5]))
    # the same plan from rows all-gathered group by group BEFOREHAND (pregather_*: what the renderer does from inside the
    # coarse pass, so that the fine pass's plan() issues no collective): identical group-major source, filled per group
    cuts = sh.cuts_for(R, R * S, groups=2)
    st = sh.pregather_start(R, cuts)
    for gi, (ra, rb) in enumerate(cuts):
        sh.pregather_group(st, gi, ro[ra:rb], (-ro)[ra:rb], z[ra:rb])
    sh.pregather_finish(st, z)
    plan2 = sh.plan(kw, groups=2)
    ok_pre = cuts == [(0, 3), (3, 5)] and all(g['ready'] is not None for g in plan2['groups'])
    for g in plan2['groups']:
        g['ready']()
    ok_pre = (ok_pre and getattr(sh, 'pregather_hits', 0) == 1 and sh._pregathered is None
              and all(torch.equal(plan2['kw_all'][k], plan['kw_all'][k]) for k in ('rays_o', 'rays_d', 'z'))
              and torch.equal(plan2['groups'][1]['kw_all']['z'], plan['groups'][1]['kw_all']['z']))
    # ... and a plan for ANOTHER depth tensor must not pick them up (they are waited for and dropped)
    st = sh.pregather_start(R, cuts)
    for gi, (ra, rb) in enumerate(cuts):
        sh.pregather_group(st, gi, ro[ra:rb], (-ro)[ra:rb], z[ra:rb])
    sh.pregather_finish(st, z)
    plan3 = sh.plan(dict(kw, z=z.clone()), groups=2)
    ok_pre = (ok_pre and all(g['ready'] is None for g in plan3['groups']) and sh.pregather_hits == 1 and sh._pregathered is None
              and torch.equal(plan3['kw_all']['z'], plan['kw_all']['z']))
    ok_plan = ok_plan and ok_pre
    # backward data movement of the groups: value = 1000 rank + 100 point + 10 level + f at point p = ray * S + k
    dfeat = torch.tensor([[val(rank, p, l, f) for l in range(L) for f in range(F)] for p in range(R * S)])
    state = None
    for grp in plan['groups']:
        state = sh.dfeat_exchange_start(plan, grp, dfeat[grp['a']:grp['b']].contiguous(), state)
    pts = [(r, i * S + k) for r, i in order for k in range(S)]
    want_recv = torch.tensor([[val(r, p, l, f) for l in sh.mine for f in range(F)] for r, p in pts])
    ok_plan = ok_plan and torch.equal(state['recv'].view(world * R * S, -1), want_recv)
    q.put((rank, ok_fwd, ok_bwd and ok_plan, ok_sync, sorted(sh.mine)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('world,grid', [(2, 'fake7'), (4, 'T24'), (8, 'T24'), (8, 'T12'), (3, 'T24')])
def test_level_sharding_exchanges_gloo(world, grid):
    """Both feature exchanges, the table sync and the ray-group plan at worlds 2, 3, 4 and 8.  'T24' is config_base's
    20-level grid: 20 levels over 8 ranks = uneven owner blocks of 2-3 levels, some ranks owning only small dense levels,
    two ray groups -- the shapes the first 8-GPU run will see ('T12' shrinks the table so that the sync runs too)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, q, grid)) for r in range(world)]
    [p.start() for p in procs]
    got = [q.get(timeout=240) for _ in range(world)]
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    owned, widths = [], []
    for rank, ok_fwd, ok_bwd, ok_sync, mine in got:
        assert ok_fwd and ok_bwd and ok_sync, (rank, ok_fwd, ok_bwd, ok_sync)
        owned += mine
        widths.append(len(mine))
    assert sorted(owned) == list(range(7 if grid == 'fake7' else 20))
    if grid == 'T24' and world == 8:
        assert min(widths) >= 1 and max(widths) > min(widths)       # uneven blocks: 20 levels do not divide by 8
