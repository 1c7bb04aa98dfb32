"""Multi-rank training on the GPU path: two ranks (gloo, both on cuda:0 -- the test box has one GPU) each train on
their half of every batch; the result must equal one process training on the whole batch (mean-over-rays loss =>
SUM of shard grads / world).  Two table strategies:
  'dp'       replicated table, per-level scatter -> all-reduce -> Adam pipeline;
  'sharded'  level-sharded table (level_parallel.py): feature all-to-all, no table all-reduce, owner-only Adam."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

KW = dict(encoding_precision='float32', encoding_n_levels=8, encoding_n_features_per_level=2,
          encoding_log2_hashmap_size=15, geo_feat_dim=15, sigma_hidden_dim=64, sigma_num_layers=2,
          encoding_dir_precision='float32', encoding_dir_encoding='SphericalHarmonics', encoding_dir_degree=4,
          latent_embedding_dim=48, color_hidden_dim=64, color_num_layers=3)
N, STEPS = 132, 3       # divisible by 2, 3 and 4: the level-sharded exchange needs the same ray count on every rank


def _batches():
    from nerf_amd.loaders.procedural import sphere_cameras
    K_all, E_all = sphere_cameras(5, 0)
    g = torch.Generator().manual_seed(4)
    out = []
    for _ in range(STEPS):
        n = torch.randint(0, 5, (N,), generator=g)
        h = torch.randint(0, 800, (N,), generator=g)
        w = torch.randint(0, 800, (N,), generator=g)
        out.append((n, h, w, K_all[n], E_all[n], torch.rand(3, generator=g), torch.rand(N, 128, generator=g),
                    torch.rand(N, 3, generator=g)))
    return out


def _train(rank, world, log2_T=15, precision='float32'):
    from nerf_amd import helpers, parallel
    from nerf_amd.nets import NeRFNetwork
    from nerf_amd.optim import FusedAdam
    from nerf_amd.renderer import NerfRenderer
    net = NeRFNetwork(5, seed=1, **dict(KW, encoding_log2_hashmap_size=log2_T, encoding_precision=precision))
    with torch.no_grad():
        net.encoder.params.mul_(1e3)
    R = NerfRenderer(net, 2, 2.5, [1.6, 4.8], [64], 128, 0.2)
    opt = FusedAdam([{'params': list(net.encoder.parameters()) + [net.latent_emb], 'lr': 2e-2},
                     {'params': list(net.sigma_net.parameters()) + list(net.color_net.parameters()), 'lr': 1e-3,
                      'weight_decay': 1e-6}], betas=(0.9, 0.99), eps=1e-15).attach(net)
    prefetch = world > 1 and os.environ.get('NERF_TEST_PREFETCH') == '1'
    a, e = parallel.shard_range(N, rank, world)
    mine = [[t[a:e].cuda() if t.dim() > 0 and t.shape[0] == N else t.cuda() for t in b] for b in _batches()]
    for i, (n, h, w, K, E, bg, noise, gt) in enumerate(mine):
        opt.zero_grad()
        img, wts, zs, _ = R.render(n, h, w, K, E, bg, noise=noise)
        loss, _ = helpers.fused_train_loss(img, gt, wts, zs, 1e-3, 1e-1)
        params = [p for g in opt.param_groups for p in g['params']]
        if world > 1 and int(os.environ.get('NERF_SHARD_GROUPS', '1')) != 1:
            # trainer.py:134 as the Trainer and bench.py call it: under level sharding the small all-reduce is queued from
            # inside the backward pass (parallel.backward_and_reduce)
            parallel.backward_and_reduce(loss, net, params,
                                         skip=lambda: [opt._table_param] if opt.table_reduced_by_pipeline() else [])
            opt.grad_scale = 1.0 / world
        else:
            loss.backward()
            if world > 1:
                skip = [opt._table_param] if opt.table_reduced_by_pipeline() else []
                assert skip, 'the table gradient must not need an all-reduce of its own'
                parallel.all_reduce_grads(params, skip=skip)
                opt.grad_scale = 1.0 / world
        if prefetch and i + 1 < STEPS:
            # Trainer.train_epoch / bench.py: the next batch's rays and the ray all-gather of its coarse pass are issued
            # before the optimizer step and picked up by the next render()
            R.prefetch_batch(*mine[i + 1][:5])
        opt.step()
    if world > 1:
        assert (net.level_sharding is not None) == (os.environ.get('NERF_TABLE_PARALLEL') != '0')
        if net.level_sharding is not None and int(os.environ.get('NERF_SHARD_GROUPS', '1')) != 1:
            assert net.level_sharding.early_reduces == STEPS        # the hook fired once per backward pass
    if world > 1 and net.level_sharding is not None:
        # ray groups: every step's fine-pass rays + depths came from the per-group all-gathers started in the coarse pass
        want = STEPS if int(os.environ.get('NERF_SHARD_GROUPS', '1')) != 1 else 0
        assert getattr(net.level_sharding, 'pregather_hits', 0) == want, getattr(net.level_sharding, 'pregather_hits', 0)
    if prefetch and net.level_sharding is not None:
        assert net.level_sharding.prefetch_hits == STEPS - 1, getattr(net.level_sharding, 'prefetch_hits', 0)
    if log2_T >= 19 and (world == 1 or net.level_sharding is not None):
        # levels 5-7 are hashed and large enough for the binned table update (every rank together owns all three)
        assert world > 1 or net.table_pipeline.binned_levels == [5, 6, 7], net.table_pipeline.binned_levels
        assert all(l in (5, 6, 7) for l in net.table_pipeline.binned_levels)
    net.sync_table()
    torch.cuda.synchronize()
    return [p.detach().cpu() for p in (net.encoder.params, net.sigma_net.params, net.color_net.params, net.latent_emb)]


def _worker(rank, world, port, q, mode, log2_T, groups, prefetch=False, precision='float32'):
    os.environ['NERF_TABLE_PARALLEL'] = '1' if mode == 'sharded' else '0'
    os.environ['NERF_SHARD_GROUPS'] = str(groups)
    os.environ['NERF_TEST_PREFETCH'] = '1' if prefetch else '0'
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    out = _train(rank, world, log2_T, precision)
    if rank == 0:
        q.put([o.numpy() for o in out])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('mode,world,log2_T,groups,prefetch', [
    ('dp', 2, 15, 1, False), ('sharded', 2, 15, 1, False), ('sharded', 3, 15, 1, False), ('sharded', 2, 19, 1, False),
    ('dp', 2, 19, 1, False), ('sharded', 2, 15, 3, False), ('sharded', 3, 19, 2, False), ('sharded', 2, 19, 4, True),
    ('sharded', 4, 19, 2, True)])
def test_multi_rank_training_matches_single_process(mode, world, log2_T, groups, prefetch, precision='float32'):
    """world 3 puts a rank in the middle (remote points on both sides of its own block) and splits 8 levels 3/3/2.
    log2_T 19 makes levels 5-7 hashed and large: the single process and the level owners then take the binned table
    update (csrc/table_update.hip), the replicated 'dp' ranks the scatter -> all-reduce -> Adam pipeline.
    groups > 1: both feature exchanges run per ray group (level_parallel.LevelSharding.plan; 63 or 42 rays per rank in
    groups of 22 / 16-17 / 11 rays), the owners see the points of all ranks in group-major order.  world 4 with two ray
    groups splits the 8 levels 2/2/2/2 over four ranks on the one GPU.  prefetch: the ray all-gather of the next batch's
    coarse pass is issued before the optimizer step (NerfRenderer.prefetch_batch) and must be the one the next step uses."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, mode, log2_T, groups, prefetch, precision)) for r in range(world)]
    [p.start() for p in procs]
    got = q.get(timeout=150)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    want = _train(0, 1, log2_T, precision)
    for g, w, name in zip(got, want, ('table', 'sigma_net', 'color_net', 'latent')):
        # Adam divides by sqrt(v): elements whose gradient is ~0 amplify rounding differences between the two
        # reduction orders, so compare with an absolute bar at the scale of one update step (lr)
        torch.testing.assert_close(torch.from_numpy(g), w, rtol=0, atol=2e-3, msg=lambda m, n=name: n + ': ' + m)
        frac = float(((torch.from_numpy(g) - w).abs() > 1e-5).float().mean())
        assert frac < 0.02, (name, frac)


@pytest.mark.parametrize('mode,world,groups', [('sharded', 2, 2), ('dp', 2, 1)])
def test_multi_rank_training_with_half_precision_table(mode, world, groups):
    """`precision: float16` (tests/test_gpu_half_table.py) under both multi-GPU table strategies: the level owners encode
    from the half-precision image of THEIR levels (refreshed after their own update pass), the replicated ranks from the
    image of the whole table; either way the result equals the single process with the same precision."""
    test_multi_rank_training_matches_single_process(mode, world, 19, groups, mode == 'sharded', precision='float16')
