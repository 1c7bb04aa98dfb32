"""Level-sharded hash table for multi-GPU training (no counterpart in the reference,
which is single-GPU).

Plain data parallelism would all-reduce the dense 1.6 GB table gradient every step -- 9 ms over xGMI against
a 7 ms step.  The path has a much cheaper exchange point: the per-point FEATURE rows (L*F floats = 160 B per
sample point).  So the table is sharded by level:

    every rank owns a subset of the hash-grid levels (table slice, Adam moments, gradient);
    all ranks all-gather the ray batch (rays_o, rays_d, depths: a few MB);
    the owner encodes ITS levels for the sample points of ALL ranks      (nerf_hashgrid_encode_levels);
    one all-to-all brings each rank the feature rows of its own points    (RCCL, ~10 MB per peer);
    MLPs, compositing, losses and their backward run where the rays live  (nerf_*_from_features, nerf_field_bwd);
    one all-to-all returns d(loss)/d(features) to the level owners, which scatter into their slice of the
    table gradient (nerf_hashgrid_scatter_levels) and run Adam on that slice only.

Only the small MLP / latent gradients are all-reduced.  Per step and rank the exchanged volume is
~(64 + 128 + 128) samples x rays x 160 B = 210 MB instead of 2 x 1.6 GB, the table-bound work (gathers, atomics)
is unchanged, and the Adam pass over the table shrinks by the number of ranks.

Between optimizer steps a rank's copy of the levels it does NOT own is stale; `sync_table()` broadcasts every
level from its owner (needed before inference, extraction or a checkpoint).
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import parallel
from ._lib import NerfColMap, NerfLevelSet, check, lib, ptr, stream


def level_costs(cfg):
    """Relative per-level cost of owning a level (encode gathers + scatter atomics + Adam on it), from
    tools/shard_emulate.py --per-level on MI355X (profiles/r01_shard_emulate_w8.log): a level's scatter runs at the
    memory-side atomic rate whatever its size (~0.9 ms for 4.2 M points), except for the small dense levels that
    the LDS-privatised kernel handles; the gathers of both passes add 0.05 (dense, L2-resident) to 0.5 (finest
    hashed level)."""
    out = []
    hashed = [l for l in range(cfg.n_levels) if cfg.hashed[l]]
    for l in range(cfg.n_levels):
        floats = cfg.entries[l] * cfg.n_features
        if cfg.hashed[l]:
            out.append(1.15 + 0.25 * (l - hashed[0]) / max(1, len(hashed) - 1))
        elif floats <= 2 * 32768:
            out.append(0.4)
        elif floats <= 16 * 32768:
            out.append(0.95)
        else:
            out.append(1.1)
    return out


def assign_levels(cfg, world):
    """owner[l] for every level: longest-processing-time greedy on level_costs, then moves / swaps while they
    lower the sum of squared loads (deterministic; identical on every rank)."""
    cost = level_costs(cfg)
    L = cfg.n_levels
    load = [0.0] * world
    owner = [0] * L
    for l in sorted(range(L), key=lambda l: (-cost[l], l)):
        r = min(range(world), key=lambda r: (load[r], r))
        owner[l] = r
        load[r] += cost[l]
    improved = True
    while improved:
        improved = False
        for l in range(L):
            for r in range(world):          # move l to r
                a = owner[l]
                if r != a and (load[r] + cost[l]) ** 2 + (load[a] - cost[l]) ** 2 < load[r] ** 2 + load[a] ** 2 - 1e-12:
                    load[a] -= cost[l]
                    load[r] += cost[l]
                    owner[l] = r
                    improved = True
            for k in range(l + 1, L):       # swap l and k
                a, b = owner[l], owner[k]
                d = cost[l] - cost[k]
                if a != b and (load[a] - d) ** 2 + (load[b] + d) ** 2 < load[a] ** 2 + load[b] ** 2 - 1e-12:
                    load[a] -= d
                    load[b] += d
                    owner[l], owner[k] = b, a
                    improved = True
    return owner


def _level_set(levels):
    s = NerfLevelSet()
    s.count = len(levels)
    for i, l in enumerate(levels):
        s.level[i] = l
    return s


class LevelSharding(object):
    def __init__(self, cfg, device, owner=None, rank=None, world=None):
        self.cfg = cfg
        self.world = parallel.world_size() if world is None else world
        self.rank = parallel.rank() if rank is None else rank
        self.F = cfg.n_features
        self.owner = list(owner) if owner is not None else assign_levels(cfg, self.world)
        if len(self.owner) != cfg.n_levels or any(not 0 <= o < self.world for o in self.owner):
            raise ValueError('owner must name a rank for each of the %d levels' % cfg.n_levels)
        self.levels = [[l for l in range(cfg.n_levels) if self.owner[l] == r] for r in range(self.world)]
        self.mine = self.levels[self.rank]
        self.my_set = _level_set(self.mine)
        self.width = [len(lv) * self.F for lv in self.levels]      # feature columns held by each owner
        cols = [l * self.F + f for lv in self.levels for l in lv for f in range(self.F)]
        self.identity = cols == list(range(len(cols)))
        self.perm = torch.tensor(cols, device=device, dtype=torch.int64)          # owner-major -> level-major source
        inv = [0] * len(cols)
        for i, c in enumerate(cols):
            inv[c] = i
        self.inv = torch.tensor(inv, device=device, dtype=torch.int64)
        self._colmaps = {}
        self.dirty = False      # True once an optimizer step made the non-owned levels of the local table stale

    # ---- point sources of all ranks -------------------------------------------------------
    def gather_src(self, kw):
        """kwargs of make_point_src for the points of ALL ranks (rank-major), from this rank's kwargs."""
        out = dict(outer_bound=kw['outer_bound'], inner_bound=kw.get('inner_bound', 0.0))
        W = self.world
        if kw.get('x') is not None:
            x = parallel.all_gather_rows(kw['x'])
            out.update(x=x, n_points=x.shape[0])
            return out
        ro, rd = kw['rays_o'], kw['rays_d']
        S, zs = kw['n_samples'], kw.get('z_stride', 0)
        if zs == 0:     # one depth table shared by every ray of every rank
            g = parallel.all_gather_rows(torch.cat([ro, rd], 1))
            z = kw['z']
        else:
            g = parallel.all_gather_rows(torch.cat([ro, rd, kw['z'].view(ro.shape[0], zs)], 1))
            z = g[:, 6:].contiguous()
        out.update(rays_o=g[:, 0:3].contiguous(), rays_d=g[:, 3:6].contiguous(), z=z, z_stride=zs, n_samples=S,
                   n_points=W * kw['n_points'])
        return out

    def col_map(self, m_local):
        """NerfColMap of the all-to-all buffers that hold `m_local` rows per rank: level-major column c of the
        feature matrix sits in the block of the level's owner."""
        cm = self._colmaps.get(m_local)
        if cm is None:
            cm = NerfColMap()
            base = 0
            for r, lv in enumerate(self.levels):
                w = self.width[r]
                for k, l in enumerate(lv):
                    for f in range(self.F):
                        cm.offset[l * self.F + f] = base + k * self.F + f
                        cm.stride[l * self.F + f] = w
                base += m_local * w
            self._colmaps[m_local] = cm
        return cm

    # ---- forward: encode my levels for everybody, all-to-all, assemble [M, L*F] -------------
    def encode_exchange(self, table, src_all, m_local):
        from .nets import make_point_src
        W, wm = self.world, self.width[self.rank]
        src = make_point_src(**src_all)
        send = torch.empty(W * m_local, max(wm, 1), device=table.device, dtype=torch.float32)
        if wm:
            check(lib().nerf_hashgrid_encode_levels(C.byref(self.cfg), ptr(table), C.byref(src), C.byref(self.my_set),
                                                    ptr(send), wm, stream()), 'nerf_hashgrid_encode_levels')
        return self.exchange_features(send.view(-1)[:W * m_local * wm], m_local)

    def exchange_features(self, send, m_local):
        """send: flat [world, m_local, width_mine] (my levels, everybody's points) -> xh [m_local, L*F] (all levels,
        my points, level-major columns)."""
        W, wm = self.world, self.width[self.rank]
        recv = parallel.all_to_all_flat(send, [m_local * wm] * W, [m_local * w for w in self.width])
        if recv.is_cuda:        # one pass: owner-major blocks -> [m_local, L*F]
            LF = self.cfg.n_levels * self.F
            xh = torch.empty(m_local, LF, device=recv.device, dtype=torch.float32)
            check(lib().nerf_feature_shuffle(ptr(recv), ptr(xh), m_local, LF, C.byref(self.col_map(m_local)), 0, stream()),
                  'nerf_feature_shuffle')
            return xh
        chunks, o = [], 0       # host tensors (the CPU tests of the exchange logic)
        for w in self.width:
            chunks.append(recv[o:o + m_local * w].view(m_local, w))
            o += m_local * w
        xh = torch.cat(chunks, 1)
        return xh if self.identity else xh.index_select(1, self.inv)

    # ---- backward: d(features) back to the level owners ------------------------------------
    def exchange_dfeat(self, dfeat):
        """dfeat [M, L*F] (level-major) -> [world*M, width_mine]: gradient columns of MY levels for the points of
        all ranks."""
        M, W, wm = dfeat.shape[0], self.world, self.width[self.rank]
        if dfeat.is_cuda:
            send = torch.empty(dfeat.numel(), device=dfeat.device, dtype=torch.float32)
            check(lib().nerf_feature_shuffle(ptr(dfeat), ptr(send), M, dfeat.shape[1], C.byref(self.col_map(M)), 1, stream()),
                  'nerf_feature_shuffle')
        else:
            g = dfeat if self.identity else dfeat.index_select(1, self.perm)
            send = torch.cat([c.reshape(-1) for c in g.split(self.width, 1)])
        recv = parallel.all_to_all_flat(send, [M * w for w in self.width], [M * wm] * W)
        return recv.view(W * M, wm) if wm else recv.view(W * M, 0)

    def scatter(self, src_all, dfeat_mine, dtable, pipe=None):
        """dtable[my levels] += scatter(dfeat_mine); with a TablePipeline, level by level with an event after each."""
        from .nets import make_point_src
        wm = self.width[self.rank]
        if not wm:
            if pipe is not None:
                pipe.begin(dtable)
            return
        src = make_point_src(**src_all)
        if pipe is None:
            check(lib().nerf_hashgrid_scatter_levels(C.byref(self.cfg), C.byref(src), C.byref(self.my_set), ptr(dfeat_mine),
                                                     wm, ptr(dtable), stream()), 'nerf_hashgrid_scatter_levels')
            return
        pipe.begin(dtable)
        slot = {l: i for i, l in enumerate(self.mine)}
        base = ptr(dfeat_mine)
        for (l0, cnt) in pipe.groups:
            for l in range(l0, l0 + cnt):
                one = _level_set([l])
                check(lib().nerf_hashgrid_scatter_levels(C.byref(self.cfg), C.byref(src), C.byref(one),
                                                         base + 4 * slot[l] * self.F, wm, ptr(dtable), stream()),
                      'nerf_hashgrid_scatter_levels')
            pipe.level_done(l0, cnt)

    # ---- table consistency ---------------------------------------------------------------
    def level_slice(self, l):
        a = int(self.cfg.offset[l]) * self.F
        b = (int(self.cfg.offset[l + 1]) if l + 1 < self.cfg.n_levels else int(self.cfg.total_entries)) * self.F
        return a, b

    @torch.no_grad()
    def sync_table(self, table):
        """Broadcast every level from its owner so that all ranks hold the whole, current table."""
        if self.world > 1 and self.dirty:
            flat = table.view(-1)
            for l in range(self.cfg.n_levels):
                a, b = self.level_slice(l)
                parallel.broadcast_(flat[a:b], self.owner[l])
        self.dirty = False
