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
import os

import torch
import torch.distributed as dist

from . import parallel
from ._lib import NerfColMap, NerfLevelSet, check, lib, ptr, stream


RANGED_FILL = os.environ.get('NERF_SHARD_RANGED_FILL', '1') != '0'


def _binned_hashed():
    """Whether the hashed levels take the binned update under level sharding (optim.FusedAdam.attach reads the same
    switch; it must be set alike on every rank)."""
    return os.environ.get('NERF_TABLE_BINNED_SHARDED', 'auto') != '0'


def sharded_binned_levels(cfg, world, mode=None):
    """The levels whose OWNER applies the binned table update (csrc/table_update.hip) in a world of `world` ranks.
    mode (NERF_TABLE_BINNED_SHARDED): '0' none; '1' every level of more than 512 K floats; 'hashed' the hashed ones of
    those; 'auto' (default) the hashed ones plus the dense ones whose partitions stay light: an owner sees the points of ALL
    ranks, the clustered points of a dense level pile into its few partitions, and one workgroup only sums ~0.3 G items/s
    (a rank owning levels 3, 4, 5 at world 8: 4.7 -> 9.2 ms with everything binned), so a dense level is binned while
    8 corners x world x 524 288 points / partitions stays under 110 K items per partition (measured with
    tools/shard_step_emulate.py: at world 2 binning level 5 of config_base -- 64 partitions, 131 K items each -- costs
    0.1 ms, at world 4 binning level 6 -- 163 partitions, 103 K items each -- gains 0.06 ms; levels 8-11 are binned at
    every world)."""
    mode = os.environ.get('NERF_TABLE_BINNED_SHARDED', 'auto') if mode is None else mode
    out = []
    F = cfg.n_features
    for l in range(cfg.n_levels):
        floats = int(cfg.entries[l]) * F
        if mode == '0' or not (16 * 32768 < floats <= 8192 * 16384):
            continue
        if mode == '1' or cfg.hashed[l]:
            out.append(l)
        elif mode == 'auto':
            parts = (floats + 16383) // 16384
            if 8 * world * 524288 / parts <= float(os.environ.get('NERF_SHARD_BIN_ITEMS', '110000')):
                out.append(l)
    return out


def level_costs(cfg, world=8, binned_hashed=None):
    """Cost (ms per step, MI355X, 4096 rays x 128 samples per rank) of owning a level in a world of `world` ranks,
    from tools/shard_emulate.py --per-level (profiles/r01_shard_emulate_w8.log):

        scatter  grows with the points of ALL ranks: 0.11 ms per 524 k points at the memory-side atomic rate whatever
                 the level's size, 0.09 on the big dense levels, 0.03 on the smallest dense levels (one LDS copy per
                 workgroup) and 0.03-0.06, growing with the number of LDS chunks, on the chunked ones of the LDS-privatised kernel (per-rank step times of two
                 different assignments at world 8, tools/shard_step_emulate.py, solved for the levels);
        encode   (coarse + fine pass = 1.5 x the fine pass) likewise: 0.005 ms (L2-resident dense level) to 0.04 ms
                 (finest hashed level) per 524 k points;
        Adam     does not grow with the world: 32 B per float of the level at ~5.4 TB/s (0.2 ms for a hashed level).

    With the binned update on the hashed levels (csrc/table_update.hip; tools/shard_step_emulate.py: a rank owning
    two hashed levels 4.6 -> 3.95 ms per step at world 8) their "scatter" is the count / fill passes plus the item
    reads of the update pass, 0.055 ms per 524 k points, and their Adam share 24 B per float (0.87 ms per hashed
    level and step at world 8, whatever the level).
    """
    if binned_hashed is None:
        binned_hashed = _binned_hashed()
    binned_set = set(sharded_binned_levels(cfg, world)) if binned_hashed else set()
    out = []
    hashed = [l for l in range(cfg.n_levels) if cfg.hashed[l]]
    for l in range(cfg.n_levels):
        floats = cfg.entries[l] * cfg.n_features
        if cfg.hashed[l]:
            scatter = 0.112
            encode = 0.019 + 0.021 * (l - hashed[0]) / max(1, len(hashed) - 1)
        else:
            scatter = 0.03 if floats <= 2 * 32768 else (0.03 + 0.03 * floats / (16 * 32768.0) if floats <= 16 * 32768 else 0.09)
            encode = 0.005 + 0.007 * min(1.0, cfg.entries[l] / float(1 << 24))
        adam = floats * 32 / 5.4e9
        if cfg.hashed[l] and l in binned_set:
            scatter, adam = 0.055, floats * 24 / 5.0e9
            encode = 0.015 + 0.010 * (l - hashed[0]) / max(1, len(hashed) - 1)
        elif l in binned_set:       # a dense level under the binned update: bin passes instead of memory-side atomics
            scatter, adam = 0.055, floats * 24 / 5.0e9
        out.append(world * (scatter + 1.5 * encode) + adam + 0.05)     # + launches and exchange bookkeeping per level
    return out


def assign_levels(cfg, world):
    """owner[l] for every level: longest-processing-time greedy on level_costs, then moves / swaps while they
    lower the sum of squared loads (deterministic; identical on every rank)."""
    cost = level_costs(cfg, world)
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


def _slice_rows(kw, ra, rb, unit):
    """make_point_src kwargs restricted to rays [ra, rb) (ray sources) or points [ra, rb) (point sources)."""
    out = dict(kw)
    if kw.get('x') is not None:
        out.update(x=kw['x'][ra:rb], n_points=rb - ra)
        for k in ('d', 'n'):
            if kw.get(k) is not None:
                out[k] = kw[k][ra:rb]
        return out
    zs = kw.get('z_stride', 0)
    out.update(rays_o=kw['rays_o'][ra:rb], rays_d=kw['rays_d'][ra:rb], z=kw['z'] if zs == 0 else kw['z'][ra:rb],
               n_points=(rb - ra) * unit)
    if kw.get('n') is not None:
        out['n'] = kw['n'][ra:rb]
    return out


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
        self._group_perm = {}
        self._prefetched = None  # (rays_o, rays_d, packed source, gathered rows, work) of prefetch_rays
        self._pregathered = None # per-group all-gathers of the fine pass's rays + depths, started from the coarse pass
        self.force = False      # run the exchange code even in a world of one rank (exercises the RCCL calls on one GPU)
        self.dirty = False      # True once an optimizer step made the non-owned levels of the local table stale

    # ---- point sources of all ranks -------------------------------------------------------
    def prefetch_rays(self, rays_o, rays_d):
        """Start the ray all-gather of the NEXT batch's coarse pass now (asynchronously): its depth table is shared, so the
        collective needs nothing but the rays, which do not depend on the parameters.  Issued before the optimizer step,
        it completes under the table update instead of standing, for its whole latency, at the top of the next step
        (one of the exposed collectives of a sharded step).  The next plan() / gather_src() that is
        handed these very tensors picks the result up; any other call gathers as usual."""
        src = torch.cat([rays_o, rays_d], 1)
        out, work = parallel.all_gather_rows(src, async_op=True)
        self._prefetched = (rays_o, rays_d, src, out, work)

    def _gather_rays(self, ro, rd):
        pf, self._prefetched = self._prefetched, None
        if pf is not None and pf[0] is ro and pf[1] is rd:
            if pf[4] is not None:
                pf[4].wait()
            self.prefetch_hits = getattr(self, 'prefetch_hits', 0) + 1
            return pf[3]
        if pf is not None and pf[4] is not None:
            pf[4].wait()            # a prefetch nobody asked for: let it finish before its buffers go away
        return parallel.all_gather_rows(torch.cat([ro, rd], 1))

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
            g = self._gather_rays(ro, rd)
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

    # ---- ray groups: the exchanges of consecutive groups overlap the compute of their neighbours ------------
    def n_groups(self, n_points):
        """How many ray groups a batch of `n_points` local sample points is cut into.  One all-to-all per group and
        pass: while group g is on the wire the owner encodes group g+1 and the MLPs run on group g-1, so only the
        first encode and the last MLP of a pass are not covered (the case that matters is a world of two ranks, where
        everything a rank exchanges crosses ONE xGMI link).  NERF_SHARD_GROUPS overrides."""
        env = os.environ.get('NERF_SHARD_GROUPS')
        if env:     # a count (equal groups) or comma-separated fractions of the rays ("0.25,0.5,0.25")
            return [float(f) for f in env.split(',')] if ',' in env else max(1, int(env))
        if self.world <= 1 or n_points < 131072:
            return 1
        # measured with the modelled wire (tools/shard_step_emulate.py): two groups win at every world once the scatters
        # run beside the exchange on their own stream (4.75 vs 5.33 / 4.32 vs 4.63 / 4.33 vs 4.45 ms at 2 / 4 / 8 ranks,
        # profiles/r02h_shard_*.log); four groups lose more to the smaller launches than they hide
        return 2

    def cuts_for(self, rows, n_points, groups=None):
        """The ray groups [(first ray, end ray)] plan() cuts a batch of `rows` rays / `n_points` sample points into."""
        groups = groups or self.n_groups(n_points)
        if isinstance(groups, (list, tuple)):
            edges, acc = [0], 0.0
            for f in groups[:-1]:
                acc += f / float(sum(groups))
                edges.append(min(rows, max(edges[-1], int(round(acc * rows)))))
            edges.append(rows)
            return [(a, b) for a, b in zip(edges[:-1], edges[1:]) if b > a] or [(0, rows)]
        G = max(1, min(int(groups), max(rows, 1)))
        return [parallel.shard_range(rows, g, G) for g in range(G)]

    # ---- fine-pass depths gathered group by group, from inside the coarse pass ---------------------------------
    # The fine pass needs the rays AND resampled depths of all ranks; the depths exist only after the coarse pass, so their
    # all-gather stood, for its whole latency, between the two passes.  With ray groups the renderer resamples group g as
    # soon as its coarse densities exist and starts the all-gather of that group's rows at once: it completes under the
    # coarse pass of group g+1 (or, for the last group, under the fine encode of group 0) and plan() picks the rows up.
    def _drop_pregathered(self):
        pg, self._pregathered = self._pregathered, None
        if pg is not None:
            for part in pg['parts']:
                if part is not None and part[2] is not None:
                    part[2].wait()      # let an all-gather nobody asked for finish before its buffers go away

    def pregather_start(self, rows, cuts):
        self._drop_pregathered()
        return {'rows': rows, 'cuts': list(cuts), 'parts': [None] * len(cuts), 'z': None}

    def pregather_group(self, st, g, rays_o, rays_d, z):
        """Start the all-gather of group g's rows [rays_o | rays_d | z] (this rank's rays [ra, rb) of the batch)."""
        packed = torch.cat([rays_o, rays_d, z], 1)
        out, work = parallel.all_gather_rows(packed, async_op=True)
        st['parts'][g] = (packed, out, work)

    def pregather_finish(self, st, z):
        """`z`: the [rows, S] depth tensor the fine pass will be called with (its identity is the key)."""
        st['z'] = z
        self._pregathered = st
        self.pregather_count = getattr(self, 'pregather_count', 0) + 1

    def plan(self, kw, groups=None):
        """Gather the point source of every rank (one all-gather, as gather_src) and cut it into ray groups.
        The gathered rays are kept GROUP-major ([group][rank][ray]) so that the points of all ranks of one group are
        one contiguous source, and so are the points of all groups: the d(features) matrix the owners receive in the
        backward pass has its rows in exactly this order."""
        rays = kw.get('x') is None
        unit = int(kw['n_samples']) if rays else 1
        M = int(kw['n_points'])
        rows = M // unit
        W = self.world
        cuts = self.cuts_for(rows, M, groups)
        G = len(cuts)
        ready = None
        if G == 1:
            if self._pregathered is not None:
                self._drop_pregathered()
            whole = self.gather_src(kw)
        else:
            whole, ready = self._gather_group_major(kw, cuts, rows)
        out = []
        for gi, (ra, rb) in enumerate(cuts):
            grp = {'ra': ra, 'rb': rb, 'a': ra * unit, 'b': rb * unit, 'm': (rb - ra) * unit}
            grp['kw_local'] = _slice_rows(kw, ra, rb, unit)
            grp['kw_all'] = whole if G == 1 else _slice_rows(whole, W * ra, W * rb, unit)
            # (pre-gathered depths: the rows of this group in `whole` are filled by this call, on the current stream, right
            # before the group's first use -- encode_exchange_groups -- so that a later group's all-gather is not waited for early)
            grp['ready'] = ready[gi] if ready is not None else None
            out.append(grp)
        return {'groups': out, 'kw_all': whole, 'm_local': M, 'unit': unit}

    def _gather_group_major(self, kw, cuts, rows):
        W = self.world
        key = (rows, tuple(cuts), W)
        perm = self._group_perm.get(key)
        if perm is None:
            idx = [q * rows + i for (ra, rb) in cuts for q in range(W) for i in range(ra, rb)]
            perm = self._group_perm[key] = torch.tensor(idx, dtype=torch.int64, device=self.perm.device)
        out = dict(outer_bound=kw['outer_bound'], inner_bound=kw.get('inner_bound', 0.0))
        pg = self._pregathered
        if kw.get('x') is not None:
            if pg is not None:
                self._drop_pregathered()
            x = parallel.all_gather_rows(kw['x']).index_select(0, perm.to(kw['x'].device))
            out.update(x=x, n_points=x.shape[0])
            return out, None
        ro, rd = kw['rays_o'], kw['rays_d']
        S, zs = kw['n_samples'], kw.get('z_stride', 0)
        perm = perm.to(ro.device)
        if zs != 0 and pg is not None and pg['z'] is kw['z'] and pg['cuts'] == list(cuts) and pg['rows'] == rows:
            # the groups' rows were gathered from inside the coarse pass (pregather_group): a group's gathered block
            # [rank][ray of the group] IS its stretch of the group-major order -- no collective and no permutation here
            self._pregathered = None
            self.pregather_hits = getattr(self, 'pregather_hits', 0) + 1
            ro_all = ro.new_empty(W * rows, 3)
            rd_all = ro.new_empty(W * rows, 3)
            z_all = ro.new_empty(W * rows, zs)
            parts = pg['parts']

            def make_ready(gi, ra, rb):
                def ready():
                    _packed, got, work = parts[gi]
                    if work is not None:
                        work.wait()
                    ro_all[W * ra:W * rb].copy_(got[:, 0:3])
                    rd_all[W * ra:W * rb].copy_(got[:, 3:6])
                    z_all[W * ra:W * rb].copy_(got[:, 6:])
                    parts[gi] = None
                return ready
            out.update(rays_o=ro_all, rays_d=rd_all, z=z_all, z_stride=zs, n_samples=S, n_points=W * kw['n_points'])
            return out, [make_ready(gi, ra, rb) for gi, (ra, rb) in enumerate(cuts)]
        if pg is not None and zs != 0:
            self._drop_pregathered()
        if zs == 0:
            g = self._gather_rays(ro, rd).index_select(0, perm)
            z = kw['z']
        else:
            g = parallel.all_gather_rows(torch.cat([ro, rd, kw['z'].view(ro.shape[0], zs)], 1)).index_select(0, perm)
            z = g[:, 6:].contiguous()
        out.update(rays_o=g[:, 0:3].contiguous(), rays_d=g[:, 3:6].contiguous(), z=z, z_stride=zs, n_samples=S,
                   n_points=W * kw['n_points'])
        return out, None

    def encode_exchange_groups(self, table, plan, xh, after_last_start=None):
        """Generator over the plan's groups, in order: yields a group once rows [a, b) of `xh` hold its feature rows.
        Schedule on the compute stream: E0 | E1 F0 | E2 F1 | ... | F(G-1), where E = encode my levels for the
        group (remote points, start the all-to-all, own points) and F = wait + shuffle into xh; the caller's MLP on
        the yielded rows follows F, so the all-to-all of group g runs under E(g+1) and the MLP of group g-1."""
        pend = None
        for grp in plan['groups']:
            if grp.get('ready') is not None:
                grp['ready']()
                grp['ready'] = None
            st = self._encode_start(table, grp['kw_all'], grp['m'])
            if after_last_start is not None and grp is plan['groups'][-1]:
                after_last_start()      # every exchange of this pass is on its way: side work from here delays no wire
            if pend is not None:
                self._encode_finish(pend[0], pend[1], xh)
                yield pend[1]
            pend = (st, grp)
        if pend is not None:
            self._encode_finish(pend[0], pend[1], xh)
            yield pend[1]

    def _encode_start(self, table, src_all, m_local):
        W, r, wm = self.world, self.rank, self.width[self.rank]
        dev = table.device
        if os.environ.get('NERF_TEST_SHARD_FAIL'):      # rehearsal hook for bench.py's replicated-table fallback
            raise RuntimeError('injected failure (NERF_TEST_SHARD_FAIL)')
        send = torch.empty(W * m_local * wm, device=dev, dtype=torch.float32)
        recv = torch.empty(m_local * sum(self.width), device=dev, dtype=torch.float32)
        send_blocks = self._blocks(send, m_local, [wm] * W)
        recv_blocks = self._blocks(recv, m_local, self.width)
        if wm:
            for (a, b) in ((0, r), (r + 1, W)):
                if b > a:
                    self._encode(table, self._sub_src(src_all, a, b, m_local), send_blocks[a], wm)
        work = parallel.all_to_all_blocks(recv_blocks, send_blocks, async_op=True)
        if wm:
            self._encode(table, self._sub_src(src_all, r, r + 1, m_local), recv_blocks[r], wm)
        return (work, send, recv, m_local)

    def _encode_finish(self, st, grp, xh):
        work, _send, recv, m_local = st
        if work is not None:
            work.wait()
        self.assemble(recv, m_local, out=xh[grp['a']:grp['b']])

    # ---- helpers ---------------------------------------------------------------------------
    def _sub_src(self, src_all, a, b, m_local):
        """make_point_src kwargs for the points of ranks [a, b) out of the gathered source."""
        kw = dict(outer_bound=src_all['outer_bound'], inner_bound=src_all.get('inner_bound', 0.0))
        if src_all.get('x') is not None:
            kw.update(x=src_all['x'][a * m_local:b * m_local], n_points=(b - a) * m_local)
            return kw
        S, zs = src_all['n_samples'], src_all['z_stride']
        R = m_local // S
        z = src_all['z'] if zs == 0 else src_all['z'][a * R:b * R]
        kw.update(rays_o=src_all['rays_o'][a * R:b * R], rays_d=src_all['rays_d'][a * R:b * R], z=z, z_stride=zs,
                  n_samples=S, n_points=(b - a) * m_local)
        return kw

    def _blocks(self, flat, m_local, widths):
        out, o = [], 0
        for w in widths:
            out.append(flat[o:o + m_local * w])
            o += m_local * w
        return out

    def _encode(self, table, kw, out, stride):
        from .nets import make_point_src
        src = make_point_src(**kw)
        enc = getattr(self, 'encoder', None)        # precision float16: gather from the half-precision image of the table
        sfx, tab = enc.gather_table() if enc is not None and table.data_ptr() == enc.params.data_ptr() else ('', table)
        check(getattr(lib(), 'nerf_hashgrid_encode_levels' + sfx)(C.byref(self.cfg), ptr(tab), C.byref(src), C.byref(self.my_set),
                                                                  ptr(out), stride, stream()), 'nerf_hashgrid_encode_levels')

    # ---- forward: encode my levels for everybody, all-to-all, assemble [M, L*F] -------------
    def encode_exchange(self, table, src_all, m_local):
        """x_hashtable [m_local, L*F] of this rank's points (one group).  Order of work: encode my levels for the
        points of the OTHER ranks, start the all-to-all, encode them for my own points (straight into my block of the
        receive buffer) while the exchange is in flight, wait, one shuffle pass into level-major rows."""
        work, _send, recv, _m = self._encode_start(table, src_all, m_local)
        if work is not None:
            work.wait()
        return self.assemble(recv, m_local)

    def assemble(self, recv, m_local, out=None):
        """Owner-major blocks of the receive buffer -> [m_local, L*F] level-major rows (into `out` when given)."""
        if recv.is_cuda:
            LF = self.cfg.n_levels * self.F
            xh = out if out is not None else torch.empty(m_local, LF, device=recv.device, dtype=torch.float32)
            check(lib().nerf_feature_shuffle(ptr(recv), ptr(xh), m_local, LF, C.byref(self.col_map(m_local)), 0, stream()),
                  'nerf_feature_shuffle')
            return xh
        chunks = [blk.view(m_local, w) for blk, w in zip(self._blocks(recv, m_local, self.width), self.width)]
        xh = torch.cat(chunks, 1)       # host tensors (the CPU tests of the exchange logic)
        xh = xh if self.identity else xh.index_select(1, self.inv)
        if out is not None:
            out.copy_(xh)
            return out
        return xh

    def exchange_features(self, send, m_local):
        """send: flat [world, m_local, width_mine] (my levels, everybody's points, own block included) ->
        xh [m_local, L*F].  The plain form of encode_exchange's data movement (used by the CPU tests)."""
        W, r, wm = self.world, self.rank, self.width[self.rank]
        recv = torch.empty(m_local * sum(self.width), device=send.device, dtype=send.dtype)
        send_blocks, recv_blocks = self._blocks(send, m_local, [wm] * W), self._blocks(recv, m_local, self.width)
        parallel.all_to_all_blocks(recv_blocks, send_blocks)
        recv_blocks[r].copy_(send_blocks[r])
        return self.assemble(recv, m_local)

    # ---- backward: d(features) back to the level owners ------------------------------------
    def split(self, dfeat):
        """[M, L*F] level-major rows -> flat owner-major blocks ([M, width_r] for every owner r)."""
        M = dfeat.shape[0]
        if dfeat.is_cuda:
            send = torch.empty(dfeat.numel(), device=dfeat.device, dtype=torch.float32)
            check(lib().nerf_feature_shuffle(ptr(dfeat), ptr(send), M, dfeat.shape[1], C.byref(self.col_map(M)), 1, stream()),
                  'nerf_feature_shuffle')
            return send
        g = dfeat if self.identity else dfeat.index_select(1, self.perm)
        return torch.cat([c.reshape(-1) for c in g.split(self.width, 1)])

    def exchange_dfeat(self, dfeat):
        """dfeat [M, L*F] (level-major) -> [world*M, width_mine]: gradient columns of MY levels for the points of
        all ranks (own block included).  The plain form of exchange_scatter's data movement."""
        M, W, r, wm = dfeat.shape[0], self.world, self.rank, self.width[self.rank]
        send = self.split(dfeat)
        recv = torch.empty(W * M * wm, device=dfeat.device, dtype=dfeat.dtype)
        send_blocks, recv_blocks = self._blocks(send, M, self.width), self._blocks(recv, M, [wm] * W)
        parallel.all_to_all_blocks(recv_blocks, send_blocks)
        recv_blocks[r].copy_(send_blocks[r])
        return recv.view(W * M, wm)

    def _scatter(self, kw, levels, dfeat_ptr, stride, dtable):
        from .nets import make_point_src
        src = make_point_src(**kw)
        check(lib().nerf_hashgrid_scatter_levels(C.byref(self.cfg), C.byref(src), C.byref(levels), dfeat_ptr, stride,
                                                 ptr(dtable), stream()), 'nerf_hashgrid_scatter_levels')

    def exchange_scatter(self, dfeat, src_all, dtable, pipe=None):
        """dtable[my levels] += scatter of d(features) from the points of all ranks.  The gradient rows of my own
        points are scattered while the all-to-all that brings the other ranks' rows is in flight; with a
        TablePipeline the remote part then goes level by level with an event after each (Adam follows on the side
        stream).  Levels that take the binned update (TablePipeline.binned_levels) are not scattered at all: the
        corner updates of all ranks' points are sorted into table partitions once the exchange has completed."""
        M, W, r, wm = dfeat.shape[0], self.world, self.rank, self.width[self.rank]
        send = self.split(dfeat)
        recv = torch.empty(W * M * wm, device=dfeat.device, dtype=torch.float32)
        send_blocks, recv_blocks = self._blocks(send, M, self.width), self._blocks(recv, M, [wm] * W)
        work = parallel.all_to_all_blocks(recv_blocks, send_blocks, async_op=True)
        self._fire_after_last_backward()
        if pipe is not None:
            pipe.begin(dtable)
        binned = list(pipe.binned_levels) if pipe is not None else []
        slot = {l: i for i, l in enumerate(self.mine)}
        # the levels that keep the atomic scatter (all of mine without a pipeline, the small ones with the binned
        # update): the rows of my own points go first, under the exchange
        early = [l for l in self.mine if l not in binned]
        if wm and early:
            own = self._sub_src(src_all, r, r + 1, M)
            with self._scatter_side(pipe):      # (the scatter stream with a pipeline: beside the exchange AND what follows)
                if len(early) == len(self.mine):
                    self._scatter(own, self.my_set, ptr(send_blocks[r]), wm, dtable)
                else:
                    for l in early:
                        self._scatter(own, _level_set([l]), ptr(send_blocks[r]) + 4 * slot[l] * self.F, wm, dtable)
        if wm and binned:
            recv_blocks[r].copy_(send_blocks[r])        # recv becomes the whole [W*M, wm] gradient matrix, rank-major
        if not wm:
            if work is not None:
                work.wait()
            return
        remote = [(a, b) for (a, b) in ((0, r), (r + 1, W)) if b > a]
        if pipe is None:
            if work is not None:
                work.wait()
            for (a, b) in remote:
                self._scatter(self._sub_src(src_all, a, b, M), self.my_set, ptr(recv_blocks[a]), wm, dtable)
            return
        pipe.keep += [send, recv, src_all]
        with self._scatter_side(pipe):
            if work is not None:
                work.wait()         # the scatter stream waits for the exchange ...
            for (l0, cnt) in pipe.groups:
                for l in range(l0, l0 + cnt):
                    one = _level_set([l])
                    for (a, b) in remote:
                        self._scatter(self._sub_src(src_all, a, b, M), one, ptr(recv_blocks[a]) + 4 * slot[l] * self.F, wm, dtable)
                pipe.level_done(l0, cnt)
        if work is not None:
            work.wait()             # ... and so does the main stream, for the bin pass below
        if binned:
            # binned update of the large levels: the corner updates of the points of ALL ranks are sorted into table
            # partitions here; FusedAdam.step sums and applies them in one pass (no gradient array, no atomics)
            from .nets import make_point_src
            pipe.bin(self.cfg, make_point_src(**self._sub_src(src_all, 0, W, M)), recv, wm,
                     {l: slot[l] * self.F for l in binned}, dtable, keepalive=src_all)

    def dfeat_exchange_start(self, plan, grp, dfeat_rows, state=None, dtable=None, pipe=None):
        """Backward of one ray group: send the gradient columns of every owner's levels for this group's points
        (asynchronously; the caller's backward of the next group runs meanwhile).  `state` carries the receive matrix
        of the whole plan, [world * m_local, width_mine] with its rows in the plan's group-major point order.
        With `dtable`, the levels that keep the atomic scatter take the rows of my OWN points of this group now, and the
        rows the PREVIOUS group's exchange has brought: both run under this group's transfer."""
        W, r, wm = self.world, self.rank, self.width[self.rank]
        M, m = plan['m_local'], grp['m']
        if state is None:
            state = {'recv': torch.empty(W * M * wm, device=dfeat_rows.device, dtype=torch.float32), 'works': [], 'keep': [],
                     'scattered': 0, 'begun': False, 'ranged': False, 'filled': 0}
        if pipe is not None and dtable is not None and not state['begun']:
            pipe.begin(dtable)
            state['begun'] = True
            # binned levels: the rows of a group are sorted into the partitions while the NEXT group is on the wire (the
            # count + scan passes ran under the forward pass, see _ShardedFieldFn.forward); needs groups of whole chunks
            chunk = 16384
            if wm and pipe.binned_levels and all((W * g['a']) % chunk == 0 for g in plan['groups']) and RANGED_FILL:
                from .nets import make_point_src
                slot = {l: i for i, l in enumerate(self.mine)}
                state['ranged'] = pipe.bin_ranges_start(self.cfg, make_point_src(**plan['kw_all']), state['recv'], wm,
                                                        {l: slot[l] * self.F for l in pipe.binned_levels}, dtable,
                                                        keepalive=plan['kw_all'], tag=plan.get('count_tag'))
        send = self.split(dfeat_rows)
        recv_g = state['recv'][W * grp['a'] * wm:W * grp['b'] * wm]
        send_blocks, recv_blocks = self._blocks(send, m, self.width), self._blocks(recv_g, m, [wm] * W)
        work = parallel.all_to_all_blocks(recv_blocks, send_blocks, async_op=True)
        if grp is plan['groups'][-1]:
            self._fire_after_last_backward()
        if wm:
            recv_blocks[r].copy_(send_blocks[r])        # my own points' rows never travel
        state['works'].append((work, grp, recv_blocks))
        state['keep'].append(send)
        if wm and dtable is not None:
            binned = list(pipe.binned_levels) if pipe is not None else []
            with self._scatter_side(pipe):
                self._scatter_rows(grp['kw_all'], r, r + 1, m, ptr(send_blocks[r]), binned, dtable)
                self._scatter_arrived(state, len(state['works']) - 1, binned, dtable)
            self._fill_arrived(state, len(state['works']) - 1, pipe)
        return state

    def _fire_after_last_backward(self):
        """Hook (parallel.backward_and_reduce): every MLP / latent gradient of this backward pass has been issued and the last
        d(features) exchange is on its way -- a small collective queued now completes under that exchange and the bin fill."""
        fn, self.after_last_backward = getattr(self, 'after_last_backward', None), None
        if fn is not None:
            self.early_reduces = getattr(self, 'early_reduces', 0) + 1
            fn()

    def _fill_arrived(self, state, upto, pipe):
        """Ranged fill (main stream) of the groups [state['filled'], upto): their exchanges are waited for first."""
        if not state['ranged']:
            return
        W = self.world
        while state['filled'] < upto:
            work, grp, _blocks = state['works'][state['filled']]
            if work is not None:
                work.wait()
            pipe.bin_range(W * grp['a'], W * grp['b'])
            state['filled'] += 1

    def _scatter_side(self, pipe):
        """Context for the atomic / LDS scatter of the levels that are not binned: with a TablePipeline they run on its
        scatter stream, BESIDE the backward kernel of the next group, the bin passes and the update pass (one rank at world 2
        spent 0.6 ms of a 5 ms step in eight such launches on the main stream); the stream first waits for what the main
        stream has produced so far (the rows just split off d(features))."""
        if pipe is None or not torch.cuda.is_available():
            import contextlib
            return contextlib.nullcontext()
        if pipe.scatter_stream is None:
            pipe.scatter_stream = torch.cuda.Stream()
        ready = torch.cuda.Event()
        ready.record()
        pipe.scatter_stream.wait_event(ready)
        return torch.cuda.stream(pipe.scatter_stream)

    def _scatter_rows(self, src_all, qa, qb, m, base_ptr, binned, dtable):
        """Atomic scatter of the non-binned levels of mine for the points of ranks [qa, qb) of one group; `base_ptr`:
        first row of rank qa's block in a [*, width_mine] gradient matrix."""
        early = [l for l in self.mine if l not in binned]
        if not early or qb <= qa:
            return
        kw = self._sub_src(src_all, qa, qb, m)
        if early == self.mine[:len(early)]:         # a prefix of my columns: one call for all of them
            self._scatter(kw, _level_set(early), base_ptr, self.width[self.rank], dtable)
            return
        slot = {l: i for i, l in enumerate(self.mine)}
        for l in early:
            self._scatter(kw, _level_set([l]), base_ptr + 4 * slot[l] * self.F, self.width[self.rank], dtable)

    def _scatter_arrived(self, state, upto, binned, dtable):
        """Scatter the remote rows of the groups [state['scattered'], upto) -- their exchanges are waited for."""
        W, r = self.world, self.rank
        while state['scattered'] < upto:
            work, grp, recv_blocks = state['works'][state['scattered']]
            if work is not None:
                work.wait()
            for (qa, qb) in ((0, r), (r + 1, W)):
                if qb > qa:
                    self._scatter_rows(grp['kw_all'], qa, qb, grp['m'], ptr(recv_blocks[qa]), binned, dtable)
            state['scattered'] += 1

    def dfeat_exchange_finish(self, plan, state, dtable, pipe=None):
        """Wait for the remaining all-to-alls and scatter what they brought; then the levels of the binned update are
        sorted into table partitions (TablePipeline.bin) from the gradient rows of the points of ALL ranks."""
        wm = self.width[self.rank]
        binned = list(pipe.binned_levels) if pipe is not None else []
        if wm:
            with self._scatter_side(pipe):
                self._scatter_arrived(state, len(state['works']), binned, dtable)
                if pipe is not None:
                    for (l0, cnt) in pipe.groups:       # events on the scatter stream: the Adam slices wait for them
                        pipe.level_done(l0, cnt)
        for work, _g, _b in state['works']:     # the main stream needs every group's rows too (bin pass below)
            if work is not None:
                work.wait()
        if not wm or pipe is None:
            return
        if state['ranged']:
            self._fill_arrived(state, len(state['works']), pipe)
            pipe.keep += [state['recv'], plan['kw_all'], state['keep']]
            pipe.bin_ranges_finish()
            return
        recv, src_all = state['recv'], plan['kw_all']
        pipe.keep += [recv, src_all, state['keep']]
        if binned:
            from .nets import make_point_src
            slot = {l: i for i, l in enumerate(self.mine)}
            pipe.bin(self.cfg, make_point_src(**src_all), recv, wm, {l: slot[l] * self.F for l in binned}, dtable,
                     keepalive=src_all)

    # ---- table consistency ---------------------------------------------------------------
    def level_slice(self, l):
        a = int(self.cfg.offset[l]) * self.F
        b = (int(self.cfg.offset[l + 1]) if l + 1 < self.cfg.n_levels else int(self.cfg.total_entries)) * self.F
        return a, b

    @torch.no_grad()
    def sync_levels(self, flat):
        """Broadcast every level slice of a table-shaped tensor (the table, an Adam moment) from its owner."""
        if dist.is_initialized():
            flat = flat.view(-1)
            for l in range(self.cfg.n_levels):
                a, b = self.level_slice(l)
                parallel.broadcast_(flat[a:b], self.owner[l])

    @torch.no_grad()
    def sync_table(self, table):
        """Broadcast every level from its owner so that all ranks hold the whole, current table."""
        if self.dirty:
            self.sync_levels(table)
        self.dirty = False
