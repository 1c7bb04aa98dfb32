"""The level-sharded training step over REAL RCCL (backend "nccl") in a world of one rank: the only way to run the
RCCL calls of nerf_amd.parallel on a one-GPU box (RCCL refuses two ranks on one device).  A world of one is the
degenerate case of the same code: all_gather_into_tensor, the list all-to-all whose own block is an empty tensor
(async + wait), the asynchronous ray all-gather a step issues for the NEXT batch (NerfRenderer.prefetch_batch), the packed
all-reduce of the small gradients, the per-level broadcasts of sync_table, barrier.
Runs in a child process so that the process group does not leak into the other tests."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch, torch.distributed as dist
from nerf_amd import helpers, parallel
from nerf_amd.optim import FusedAdam
from nerf_amd.renderer import NerfRenderer
from test_gpu_parity import NET_CFGS, _batch, _make_pair

def train(shard, cfg="small"):
    net, _ = _make_pair(NET_CFGS[cfg])
    R = NerfRenderer(net, 2, 2.5, [1.6, 4.8], [64], 128, 0.2)
    opt = FusedAdam([{"params": list(net.encoder.parameters()) + [net.latent_emb], "lr": 2e-2},
                     {"params": list(net.sigma_net.parameters()) + list(net.color_net.parameters()), "lr": 1e-3}],
                    betas=(0.9, 0.99), eps=1e-15).attach(net, shard_levels=shard)
    assert (net.level_sharding is not None) == bool(shard)
    assert bool(net.table_pipeline.binned_levels) == (cfg == "base19")    # sharded: the hashed levels; plain: all large ones
    n, h, w, K, E, bg, noise, gt = [t.cuda() for t in _batch(3, 96)]
    for _ in range(3):
        opt.zero_grad()
        img, wts, zs, _ = R.render(n, h, w, K, E, bg, noise=noise)
        loss, _ = helpers.fused_train_loss(img, gt, wts, zs, 1e-3, 1e-1)
        loss.backward()
        skip = [opt._table_param] if opt.table_reduced_by_pipeline() else []
        parallel.all_reduce_grads([p for g in opt.param_groups for p in g["params"]], skip=skip)
        if shard:       # the next step's ray all-gather, asynchronous over RCCL, issued before the optimizer step
            R.prefetch_batch(n, h, w, K, E)
        opt.step()
    if shard:
        assert net.level_sharding.prefetch_hits == 2, getattr(net.level_sharding, "prefetch_hits", 0)
        # ray groups: the fine pass's rays + depths were all-gathered group by group from inside the coarse pass
        # (asynchronous RCCL all-gathers, waited for one by one right before each group's encode)
        want = 3 if os.environ.get("NERF_SHARD_GROUPS") else 0
        assert getattr(net.level_sharding, "pregather_hits", 0) == want, getattr(net.level_sharding, "pregather_hits", 0)
    net.sync_table()
    torch.cuda.synchronize()
    return float(loss), [p.detach().clone() for p in (net.encoder.params, net.sigma_net.params, net.color_net.params, net.latent_emb)]

rank, world, local = parallel.init_from_env(force=True)   # backend nccl: GPUs are present
assert dist.get_backend() == "nccl" and world == 1
l1, sharded = train(True)
l0, plain = train(False)
dist.barrier()
assert abs(l0 - l1) < 1e-5, (l0, l1)
for a, b in zip(sharded, plain):
    assert float(((a - b).abs() > 1e-5).float().mean()) < 0.02
    assert float((a - b).abs().max()) < 2e-3
# the same with levels large enough for the binned table update behind the sharded exchange (T = 2^19)
l1, sharded = train(True, "base19")
l0, plain = train(False, "base19")
dist.barrier()
assert abs(l0 - l1) < 1e-5, (l0, l1)
for a, b in zip(sharded, plain):
    assert float(((a - b).abs() > 1e-5).float().mean()) < 0.02
    assert float((a - b).abs().max()) < 4e-2 + 1e-3      # a table entry whose tiny gradient flips sign moves by 2 lr
# the same again with the exchanges cut into three ray groups (level_parallel.LevelSharding.plan): several RCCL
# all-to-alls in flight at once, waited for in order, flags riding with the small gradients
os.environ["NERF_SHARD_GROUPS"] = "3"
l2, grouped = train(True, "base19")
del os.environ["NERF_SHARD_GROUPS"]
dist.barrier()
assert abs(l2 - l1) < 1e-5, (l2, l1)
for a, b in zip(grouped, sharded):
    assert float(((a - b).abs() > 1e-5).float().mean()) < 0.02
    assert float((a - b).abs().max()) < 4e-2 + 1e-3
# in-kernel noise: resampling group by group (nerf_coarse_to_fine_rng_at) draws what one launch over the batch draws
import nerf_amd.renderer as RM
os.environ["NERF_SHARD_GROUPS"] = "3"
net_, _ = _make_pair(NET_CFGS["small"])
R_ = NerfRenderer(net_, 2, 2.5, [1.6, 4.8], [64], 128, 0.2)
FusedAdam([{"params": list(net_.parameters()), "lr": 1e-3}], betas=(0.9, 0.99), eps=1e-15).attach(net_, shard_levels=True)
n_, h_, w_2, K_, E_, bg_, _, _ = [t.cuda() for t in _batch(3, 96)]
seen = []
for on in (True, False):
    RM.PIPELINED_DEPTH_GATHER = on
    torch.manual_seed(7)
    with torch.no_grad():
        _, _, _, aux_ = R_.render(n_, h_, w_2, K_, E_, bg_)
    seen.append((aux_["z_vals"].clone(), aux_["sigmas"].clone()))
RM.PIPELINED_DEPTH_GATHER = True
del os.environ["NERF_SHARD_GROUPS"]
assert getattr(net_.level_sharding, "pregather_hits", 0) == 1
assert torch.equal(seen[0][0], seen[1][0]) and float((seen[0][1] - seen[1][1]).abs().max()) < 1e-6
flag = torch.tensor([3.0], dtype=torch.float64, device="cuda")
w_ = torch.nn.Parameter(torch.ones(4, device="cuda")); w_.grad = torch.full((4,), 2.0, device="cuda")
parallel.all_reduce_grads([w_], flags=[flag])
assert float(flag) == 3.0 and float(w_.grad.sum()) == 8.0
# the helpers directly, incl. the empty own block of the list all-to-all
x = torch.arange(12., device="cuda").view(4, 3)
assert torch.equal(parallel.all_gather_rows(x), x)
g_out, g_work = parallel.all_gather_rows(x, async_op=True)
if g_work is not None:
    g_work.wait()
assert torch.equal(g_out, x)
send, recv = torch.ones(5, device="cuda"), torch.zeros(5, device="cuda")
work = parallel.all_to_all_blocks([recv], [send], async_op=True)
if work is not None:
    work.wait()
torch.cuda.synchronize()
assert float(recv.sum()) == 0.0                        # the own block is never transferred
dist.destroy_process_group()
print("RCCL_SINGLE_RANK_OK")
'''


def test_sharded_step_over_rccl_world_of_one():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    env.pop('NERF_DIST_BACKEND', None)
    out = subprocess.run([sys.executable, '-c', CHILD, ROOT], env=env, cwd=ROOT, capture_output=True, text=True, timeout=170)
    assert out.returncode == 0 and 'RCCL_SINGLE_RANK_OK' in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])
