#!/usr/bin/env python3
"""Per-rank compute cost of the level-sharded training step, emulated on ONE GPU (no collectives): for every
rank of a pretend world, time the kernels that rank would run on the metric workload -- encode of its levels for
the points of all ranks, MLPs from features, MLP backward, scatter of its levels, Adam on its slices -- and
report max / mean over ranks.  Used to tune level_parallel.level_costs (the owner assignment).

    python tools/shard_emulate.py --world 8 [--owner 0,1,2,...]
"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from nerf_amd._lib import check, lib, ptr, stream
from nerf_amd.level_parallel import LevelSharding, _level_set, level_costs
from nerf_amd.nets import NeRFNetwork, make_point_src

NET = dict(encoding_precision='float32', encoding_n_levels=20, encoding_n_features_per_level=2,
           encoding_log2_hashmap_size=24, geo_feat_dim=15, sigma_hidden_dim=64, sigma_num_layers=2,
           encoding_dir_precision='float32', encoding_dir_encoding='SphericalHarmonics', encoding_dir_degree=4,
           latent_embedding_dim=48, color_hidden_dim=64, color_num_layers=3)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--world', type=int, default=8)
    ap.add_argument('--rays', type=int, default=4096)
    ap.add_argument('--owner', default=None)
    ap.add_argument('--per-level', action='store_true', help='time encode / scatter of every level on its own')
    args = ap.parse_args()
    W, N, Sc, Sf = args.world, args.rays, 64, 128
    net = NeRFNetwork(100, seed=0, **NET)
    cfg = net.encoder.cfg
    with torch.no_grad():
        net.encoder.params.mul_(1e3)
    g = torch.Generator(device='cuda').manual_seed(0)
    # rays roughly like the bench scene: origins on a sphere of radius 4, looking at the centre
    o = torch.randn(W * N, 3, device='cuda', generator=g)
    o = 4.0 * o / o.norm(dim=-1, keepdim=True)
    d = -o / 4.0 + 0.15 * torch.randn(W * N, 3, device='cuda', generator=g)
    zc = torch.logspace(torch.log10(torch.tensor(1.6)).item(), torch.log10(torch.tensor(4.8)).item(), Sc, device='cuda')
    zf = torch.sort(1.6 + 3.2 * torch.rand(W * N, Sf, device='cuda', generator=g), dim=-1).values.contiguous()
    kw_c = dict(outer_bound=2.5, inner_bound=2.0, rays_o=o, rays_d=d, z=zc, z_stride=0, n_samples=Sc, n_points=W * N * Sc)
    kw_f = dict(outer_bound=2.5, inner_bound=2.0, rays_o=o, rays_d=d, z=zf, z_stride=Sf, n_samples=Sf, n_points=W * N * Sf)
    src_c, src_f = make_point_src(**kw_c), make_point_src(**kw_f)
    table, dtable = net.encoder.params, torch.zeros_like(net.encoder.params)
    LF = cfg.n_levels * cfg.n_features
    out = {}
    if args.per_level:
        buf = torch.empty(W * N * Sf, 2, device='cuda')
        dbuf = torch.randn(W * N * Sf, 2, device='cuda')
        rows = []
        for l in range(cfg.n_levels):
            one = _level_set([l])
            te = timed(lambda: check(lib().nerf_hashgrid_encode_levels(C.byref(cfg), ptr(table), C.byref(src_f), C.byref(one),
                                                                       ptr(buf), 2, stream()), 'enc'))
            ts = timed(lambda: check(lib().nerf_hashgrid_scatter_levels(C.byref(cfg), C.byref(src_f), C.byref(one), ptr(dbuf),
                                                                        2, ptr(dtable), stream()), 'sc'))
            rows.append({'level': l, 'hashed': int(cfg.hashed[l]), 'entries': int(cfg.entries[l]),
                         'encode_ms': round(te, 4), 'scatter_ms': round(ts, 4)})
            print(json.dumps(rows[-1]), flush=True)
        out['per_level'] = rows
    owner = [int(x) for x in args.owner.split(',')] if args.owner else None
    m = torch.zeros_like(table)
    v = torch.zeros_like(table)
    ranks = []
    for r in range(W):
        sh = LevelSharding(cfg, 'cuda', owner=owner, rank=r, world=W)
        wm = sh.width[r]
        fc = torch.empty(W * N * Sc, max(wm, 1), device='cuda')
        ff = torch.empty(W * N * Sf, max(wm, 1), device='cuda')
        dm = torch.randn(W * N * Sf, max(wm, 1), device='cuda')
        t = {}
        t['encode_coarse'] = timed(lambda: check(lib().nerf_hashgrid_encode_levels(
            C.byref(cfg), ptr(table), C.byref(src_c), C.byref(sh.my_set), ptr(fc), wm, stream()), 'enc')) if wm else 0.0
        t['encode_fine'] = timed(lambda: check(lib().nerf_hashgrid_encode_levels(
            C.byref(cfg), ptr(table), C.byref(src_f), C.byref(sh.my_set), ptr(ff), wm, stream()), 'enc')) if wm else 0.0
        t['scatter'] = timed(lambda: check(lib().nerf_hashgrid_scatter_levels(
            C.byref(cfg), C.byref(src_f), C.byref(sh.my_set), ptr(dm), wm, ptr(dtable), stream()), 'sc')) if wm else 0.0

        def adam():
            for l in sh.mine:
                a, b = sh.level_slice(l)
                check(lib().nerf_adam_step(ptr(table.view(-1)[a:b]), ptr(dtable.view(-1)[a:b]), ptr(m.view(-1)[a:b]),
                                           ptr(v.view(-1)[a:b]), b - a, 1e-9, 0.9, 0.99, 1e-15, 0.0, 1, 1.0, 1, stream()), 'adam')
        t['adam'] = timed(adam)
        # assemble / split passes (torch): xh = cat(...).index_select ; dfeat.index_select -> split -> cat
        xh_parts = [torch.empty(N * Sf, w, device='cuda') for w in sh.width]
        recv_flat = torch.randn(N * Sf * LF, device="cuda")
        xh_out = torch.empty(N * Sf, LF, device="cuda")
        t["assemble_fine"] = timed(lambda: check(lib().nerf_feature_shuffle(ptr(recv_flat), ptr(xh_out), N * Sf, LF, C.byref(sh.col_map(N * Sf)), 0, stream()), "sh"))
        dfeat = torch.randn(N * Sf, LF, device='cuda')
        t["split_dfeat"] = timed(lambda: check(lib().nerf_feature_shuffle(ptr(dfeat), ptr(recv_flat), N * Sf, LF, C.byref(sh.col_map(N * Sf)), 1, stream()), "sh"))
        t = {k: round(x, 4) for k, x in t.items()}
        t['levels'] = sh.mine
        t['total_table_ms'] = round(sum(x for k, x in t.items() if k != 'levels'), 4)
        ranks.append(t)
        print(json.dumps({'rank': r, **t}), flush=True)
    # the MLP halves are the same on every rank
    xh = torch.randn(N * Sf, LF, device='cuda') * 0.1
    sig, rgb = torch.empty(N * Sf, device='cuda'), torch.empty(N * Sf, 3, device='cuda')
    n = torch.randint(0, 100, (N,), device='cuda')
    kw_l = dict(kw_f, rays_o=o[:N].contiguous(), rays_d=d[:N].contiguous(), z=zf[:N].contiguous(), n_points=N * Sf, n=n)
    src_l = make_point_src(**kw_l)
    mlp = {}
    mlp['density_from_features'] = timed(lambda: check(lib().nerf_density_from_features(
        C.byref(cfg), C.byref(net.net_cfg), ptr(net.sigma_net.params), ptr(xh), N * Sc, ptr(sig), stream()), 'd'))
    mlp['field_from_features'] = timed(lambda: check(lib().nerf_field_from_features(
        C.byref(cfg), C.byref(net.net_cfg), ptr(net.sigma_net.params), ptr(net.color_net.params), ptr(net.latent_emb), 100,
        C.byref(src_l), ptr(xh), ptr(sig), ptr(rgb), None, stream()), 'f'))
    out['mlp_ms'] = {k: round(x, 4) for k, x in mlp.items()}
    tot = [t['total_table_ms'] for t in ranks]
    out.update(world=W, owner=LevelSharding(cfg, 'cuda', owner=owner, rank=0, world=W).owner, costs=level_costs(cfg, W),
               max_table_ms=max(tot), mean_table_ms=sum(tot) / W)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
