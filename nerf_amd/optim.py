"""Fused dense Adam (nerf_adam_step) behind torch.optim's param_groups interface, so the
reference's optimizer wiring (train.py:84-96: three groups, per-group lr / weight_decay,
LambdaLR on top) works unchanged.  Update rule = torch.optim.Adam's single-tensor path."""
import ctypes as C
import os

import torch
import torch.distributed as dist

from . import parallel
from ._lib import NerfLevelSet, check, lib, ptr, stream


SMALL_TENSORS_ON_SIDE_STREAM = os.environ.get('NERF_SMALL_ADAM_SIDE', '1') != '0'
# precision float16: the table passes write the half-precision gather image themselves ('0': a conversion pass before the next forward)
FUSED_HALF_IMAGE = os.environ.get('NERF_FUSED_HALF_IMAGE', '1') != '0'
SPLIT_SMALL_GROUP = os.environ.get('NERF_SPLIT_SMALL', '1') != '0'


class TablePipeline(object):
    """Per-level pipeline for the hash-table gradient: scatter(level group) on the main stream ->
    [all-reduce of that slice, when distributed] -> Adam on that slice on a side stream.

    The scatter is bound by the rate of memory-side atomics and the Adam pass by HBM streaming, so the two
    overlap almost perfectly; without the pipeline they run back to back.  Level groups are ordered with the
    large hashed levels first so that the tail left after the last scatter is a small dense level."""

    def __init__(self, grid_cfg, n_features, levels=None, reduce=True, binned=None):
        """`levels`: restrict to these levels (the ones this rank owns under a level-sharded table, whose
        gradient is complete without any all-reduce: pass reduce=False).
        `binned` (default: on whenever no table gradient has to be all-reduced, NERF_TABLE_BINNED=0 turns it off):
        the levels larger than the LDS-privatised scatter takes go through the binned update (csrc/table_update.hip:
        scatter and Adam in one pass, no gradient array); only the small dense levels keep scatter -> Adam."""
        L = grid_cfg.n_levels
        need_reduce = reduce and dist.is_initialized() and dist.get_world_size() > 1
        if binned is None:
            binned = not need_reduce and os.environ.get('NERF_TABLE_BINNED', '1') != '0'
        self.binned_levels = []
        if binned:
            own = list(range(L)) if levels is None else sorted(levels)
            # (upper bound: nerf_table_bin_levels takes at most 8192 partitions of 16 K floats per level)
            if isinstance(binned, (list, tuple, set)):      # an explicit choice (level sharding: level_parallel.sharded_binned_levels)
                self.binned_levels = [l for l in own if l in binned]
            else:
                self.binned_levels = [l for l in own if 16 * 32768 < int(grid_cfg.entries[l]) * n_features <= 8192 * 16384
                                      and (binned != 'hashed' or grid_cfg.hashed[l])]
            rest = [l for l in own if l not in self.binned_levels]
            if levels is not None:
                levels = rest
        self.binned_set = None          # NerfLevelSet of binned_levels (made on first use)
        self.binned_ws = None           # workspace of nerf_table_bin_levels, reused while large enough
        self.binned_pending = None      # (n_points, workspace bytes) between backward and step
        self.binned_fallback = False    # a second backward arrived before step(): gradients went to p.grad instead
        if levels is None:
            hashed = [(l, 1) for l in range(L) if grid_cfg.hashed[l]]
            dense = [l for l in range(L) if not grid_cfg.hashed[l]]
            # the small dense levels (LDS-privatised scatter, tiny Adam slices) travel as one group; the big dense
            # levels are groups of their own
            small = [l for l in dense if grid_cfg.entries[l] * n_features <= 16 * 32768]
            big = [(l, 1) for l in reversed(dense) if l not in small]
            if small and small == list(range(small[0], small[0] + len(small))):
                # one group -- or two, when the binned update is on: the chunked levels (three LDS chunks and more) and the
                # one- or two-chunk levels then scatter on two streams side by side (see _FieldFn.backward)
                few = [l for l in small if grid_cfg.entries[l] * n_features <= 2 * 32768]
                if self.binned_levels and few and len(few) < len(small) and few == small[:len(few)] and SPLIT_SMALL_GROUP:
                    small = [(small[len(few)], len(small) - len(few)), (few[0], len(few))]
                else:
                    small = [(small[0], len(small))]
            else:
                small = [(l, 1) for l in small]
            # hashed levels first (most Adam work released per scatter), small levels last: measured best of the
            # four obvious orders
            groups = hashed + big + small
            groups = [(l0, cnt) for (l0, cnt) in groups if l0 not in self.binned_levels]
        else:       # owned levels are not contiguous: one group each, big hashed levels first
            groups = [(l, 1) for l in sorted(levels, key=lambda l: (-int(grid_cfg.entries[l]), l))]
        self.groups = groups
        self.reduce = reduce
        self.slices = {}
        for (l0, cnt) in groups:
            a = int(grid_cfg.offset[l0]) * n_features
            b = (int(grid_cfg.offset[l0 + cnt]) if l0 + cnt < L else int(grid_cfg.total_entries)) * n_features
            self.slices[l0] = (a, b)
        self.events = {}
        self.early_events = {}
        self.works = {}
        self.side = None
        self.scatter_stream = None      # carries the scatter of the first backward chunk (see _FieldFn.backward)
        self.count_stream = None        # carries the early count + scan passes of the binned update (count_early)
        self.count_tag = None
        self.count_done = None
        self.early_fraction = float(os.environ.get('NERF_BWD_SPLIT', '0.35'))
        self.pending = False
        self.dtable = None
        self.keep = []

    def begin(self, dtable):
        self.dtable = dtable
        self.pending = True
        self.works = {}
        self.early_events = {}
        self.keep = []

    # ---- binned levels ---------------------------------------------------------------------------------
    def _binned_setup(self, cfg, n_points, device):
        if self.binned_set is None:
            ls = NerfLevelSet()
            ls.count = len(self.binned_levels)
            for i, l in enumerate(self.binned_levels):
                ls.level[i] = l
            self.binned_set = ls
        nbytes = lib().nerf_table_bin_workspace_bytes(C.byref(cfg), C.byref(self.binned_set), n_points)
        if nbytes < 0:
            check(int(nbytes), 'nerf_table_bin_workspace_bytes')
        if self.binned_ws is None or self.binned_ws.numel() < nbytes:
            if self.count_done is not None:
                self.count_done.synchronize()       # an early count pass may still be writing the old workspace
            self.binned_ws = torch.empty(int(nbytes), dtype=torch.uint8, device=device)

    def count_early(self, cfg, src, tag, device, on=None):
        """The count + scan passes of bin() need only the points: issued from the FORWARD pass on a side stream they run
        under the field forward (which is bound by random gathers, while the count pass is index arithmetic + an LDS
        histogram) instead of between the backward kernel and the fill.  `tag` identifies the forward whose backward may
        use the counts; any other backward (a second forward in between, gradient accumulation) counts again.  `on`: the side
        stream to use instead of the pipeline's own count stream."""
        self.count_tag = None
        if not self.binned_levels or self.binned_pending is not None or self.binned_fallback or src.n_points == 0:
            return
        self._binned_setup(cfg, src.n_points, device)
        if on is None:
            if self.count_stream is None:
                self.count_stream = torch.cuda.Stream()
            on = self.count_stream
        ready = torch.cuda.Event()
        ready.record()      # the points exist, and every earlier reader of the workspace (the last update pass) was issued
        with torch.cuda.stream(on):
            on.wait_event(ready)
            check(lib().nerf_table_bin_count(C.byref(cfg), C.byref(src), C.byref(self.binned_set), ptr(self.binned_ws),
                                             self.binned_ws.numel(), stream()), 'nerf_table_bin_count')
            self.count_done = torch.cuda.Event()
            self.count_done.record()
        self.count_tag = (tag, src.n_points, self.binned_ws.data_ptr())

    def bin(self, cfg, src, dfeat, stride, cols, dtable, keepalive=None, tag=None):
        """Sort the corner updates of this backward pass into table partitions (no optimizer hyper-parameters needed
        yet; FusedAdam.step applies them with nerf_table_update_binned).  `cols`: {level: first gradient column in a
        row of dfeat}; `keepalive`: whatever owns the memory `src` points into (it must outlive this pass when a
        later backward has to replay it).  One backward per step is the fast path: a second backward before step() -- gradient
        accumulation -- falls back to the atomic scatter into p.grad.  `tag`: see count_early."""
        if not self.binned_levels:
            return
        counted = getattr(self, 'count_tag', None)
        self.count_tag = None
        if self.binned_pending is not None or self.binned_fallback:
            if self.binned_pending is not None:         # move the earlier pass's updates into the gradient
                src0, dfeat0, stride0, cols0, _alive = self.binned_keep
                self._scatter_classic(cfg, src0, dfeat0, stride0, cols0, dtable)
                self.binned_pending = None
            self.binned_fallback = True
            self._scatter_classic(cfg, src, dfeat, stride, cols, dtable)
            return
        self._binned_setup(cfg, src.n_points, dfeat.device)
        col_arr = (C.c_int32 * len(self.binned_levels))(*[int(cols[l]) for l in self.binned_levels])
        if self.count_done is not None:      # whatever count pass is still in flight writes this workspace
            torch.cuda.current_stream().wait_event(self.count_done)
        if tag is not None and counted == (tag, src.n_points, self.binned_ws.data_ptr()):
            check(lib().nerf_table_bin_fill(C.byref(cfg), C.byref(src), C.byref(self.binned_set), ptr(dfeat), stride, col_arr,
                                            ptr(self.binned_ws), self.binned_ws.numel(), stream()), 'nerf_table_bin_fill')
        else:
            check(lib().nerf_table_bin_levels(C.byref(cfg), C.byref(src), C.byref(self.binned_set), ptr(dfeat), stride,
                                              col_arr, ptr(self.binned_ws), self.binned_ws.numel(), stream()),
                  'nerf_table_bin_levels')
        self.binned_pending = (src.n_points, self.binned_ws.numel())
        self.binned_keep = (src, dfeat, stride, cols, keepalive)

    # ---- the same, with the fill issued range by range (level sharding: the gradient rows arrive group by group) ----
    def bin_ranges_start(self, cfg, src, dfeat, stride, cols, dtable, keepalive=None, tag=None):
        """Prepare a bin pass over all points of `src` whose fill will be issued with bin_range() for whole-chunk point
        ranges, as their rows of `dfeat` become complete; the count + scan passes run here unless count_early() already ran
        them for this forward (`tag`).  Returns False when the fast path is not available (no binned levels, or a second
        backward before step()): the caller then calls bin() once everything has arrived."""
        if not self.binned_levels or self.binned_pending is not None or self.binned_fallback:
            return False
        counted = getattr(self, 'count_tag', None)
        self.count_tag = None
        self._binned_setup(cfg, src.n_points, dfeat.device)
        if self.count_done is not None:
            torch.cuda.current_stream().wait_event(self.count_done)
        if not (tag is not None and counted == (tag, src.n_points, self.binned_ws.data_ptr())):
            check(lib().nerf_table_bin_count(C.byref(cfg), C.byref(src), C.byref(self.binned_set), ptr(self.binned_ws),
                                             self.binned_ws.numel(), stream()), 'nerf_table_bin_count')
        col_arr = (C.c_int32 * len(self.binned_levels))(*[int(cols[l]) for l in self.binned_levels])
        self._ranged = (cfg, src, dfeat, stride, col_arr, cols, keepalive)
        return True

    def bin_range(self, p_begin, p_end):
        cfg, src, dfeat, stride, col_arr, _cols, _alive = self._ranged
        check(lib().nerf_table_bin_fill_range(C.byref(cfg), C.byref(src), C.byref(self.binned_set), ptr(dfeat), stride, col_arr,
                                              ptr(self.binned_ws), self.binned_ws.numel(), int(p_begin), int(p_end), stream()),
              'nerf_table_bin_fill_range')

    def bin_ranges_finish(self):
        cfg, src, dfeat, stride, _col_arr, cols, keepalive = self._ranged
        self._ranged = None
        self.binned_pending = (src.n_points, self.binned_ws.numel())
        self.binned_keep = (src, dfeat, stride, cols, keepalive)

    def _scatter_classic(self, cfg, src, dfeat, stride, cols, dtable):
        for l in self.binned_levels:        # one level at a time: its columns start at cols[l] of every row
            one = NerfLevelSet()
            one.count = 1
            one.level[0] = l
            check(lib().nerf_hashgrid_scatter_levels(C.byref(cfg), C.byref(src), C.byref(one), ptr(dfeat) + 4 * int(cols[l]),
                                                     stride, ptr(dtable), stream()), 'nerf_hashgrid_scatter_levels')

    def drop_binned(self):
        self.binned_pending = None
        self.binned_keep = None
        self.binned_fallback = False

    def level_partial(self, l0):
        """A first part of the level group's scatter has been issued on the CURRENT stream (atomics commute, so
        the parts may run concurrently); Adam on the group waits for this event too."""
        ev = torch.cuda.Event()
        ev.record()
        self.early_events.setdefault(l0, []).append(ev)

    def level_done(self, l0, cnt):
        if self.reduce and dist.is_initialized() and dist.get_world_size() > 1:
            for ev in self.early_events.get(l0, ()):     # the slice must be complete before it is reduced
                torch.cuda.current_stream().wait_event(ev)
            a, b = self.slices[l0]
            self.works[l0] = parallel.all_reduce_sum_(self.dtable[a:b], async_op=True)
        ev = self.events.get(l0)
        if ev is None:
            ev = self.events[l0] = torch.cuda.Event()
        ev.record()


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.grad_scale = 1.0
        self._cleared_at = None       # model.backward_count at the end of the last step() (attached models only)
        # Non-finite guard: a one-element fp64 device tensor (helpers.fused_train_loss / NerfRenderer.render_loss ->
        # loss.nonfinite_flag); when it is non-zero the NEXT step() leaves parameters and moments untouched and only clears
        # the gradients -- decided on the device, no host sync (GradScaler's role at trainer.py:134-136).  Consumed by one
        # step().  What raises it: a NaN / Inf loss (both routes) and, on the render_loss route, a NaN / Inf in the
        # compositing gradients d(sigma), d(rgb) of a finite loss (nerf_composite_loss_bwd).  Underneath, every Adam kernel
        # leaves an ENTRY whose own gradient is not finite untouched (adam_common.h), whatever the flag says.
        self.skip_flag = None
        self.skipped_steps = 0        # steps found dropped so far (read back lazily, see _collect_flags)
        self._flag_reads = []         # (event, pinned host copy) of the flags of earlier steps
        self._image_written = False   # this step's table passes wrote the half-precision image of the table themselves
        self._table_param = None
        self._pipeline = None
        self._sharding = None

    def attach(self, model, pipeline=True, shard_levels=None):
        """Give every parameter a persistent .grad and let the model's backward kernel accumulate into
        it directly; the step kernel then clears the gradient in the same pass (zero_grad is free).
        With `pipeline`, the hash-table update runs as a per-level scatter -> (all-reduce) -> Adam pipeline.
        `shard_levels` (default: on when a process group with more than one rank exists, NERF_TABLE_PARALLEL=0
        turns it off) shards the hash table by level over the ranks (level_parallel.py): the table then needs
        no gradient all-reduce and this rank's Adam pass covers only the levels it owns."""
        model.enable_direct_grads(True)
        self._table_param = model.encoder.params
        self._model = model
        if shard_levels is None:
            shard_levels = (dist.is_initialized() and dist.get_world_size() > 1
                            and os.environ.get('NERF_TABLE_PARALLEL', '1') != '0')
        if shard_levels:
            sh = model.shard_levels(None if shard_levels is True else shard_levels)
            sh.force = True         # explicitly requested (or world > 1): also valid, if pointless, for one rank
            self._sharding = sh
            # (an owner sees the points of ALL ranks on a few levels: the clustered points of a DENSE level then pile
            # into a handful of partitions and the binned update gets slower than the atomic scatter, measured with
            # tools/shard_step_emulate.py -- so only the hashed levels, whose partitions fill evenly, are binned here;
            # NERF_TABLE_BINNED_SHARDED=0 / 1 turns it off / on for every large level)
            from .level_parallel import sharded_binned_levels
            chosen = sharded_binned_levels(model.encoder.cfg, sh.world)
            model.table_pipeline = TablePipeline(model.encoder.cfg, model.encoder.cfg.n_features, levels=sh.mine,
                                                 reduce=False, binned=chosen if chosen else False)
            self._pipeline = model.table_pipeline
        elif pipeline:
            model.table_pipeline = TablePipeline(model.encoder.cfg, model.encoder.cfg.n_features)
            self._pipeline = model.table_pipeline
        return self

    def zero_grad(self, set_to_none=False):
        model = getattr(self, '_model', None)
        if model is not None and self._cleared_at is not None and self._cleared_at == model.backward_count:
            return      # no backward has reached the model since the last step(), which cleared every gradient in its own pass
        self._cleared_at = None
        if self._pipeline is not None:
            # a backward is being discarded (e.g. a non-finite loss): its binned corner updates and the pending
            # per-level pipeline state go with it
            pipe = self._pipeline
            pipe.drop_binned()
            if pipe.scatter_stream is not None:     # scatters of the discarded pass may still be running beside us
                torch.cuda.current_stream().wait_stream(pipe.scatter_stream)
            if pipe.count_stream is not None:
                torch.cuda.current_stream().wait_stream(pipe.count_stream)
            for w in pipe.works.values():
                if w is not None:
                    w.wait()
            pipe.works = {}
            pipe.pending = False
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is not None:
                    p.grad.zero_()

    def _collect_flags(self, wait=False):
        """Flags of earlier steps whose host copy has arrived: a dropped step must not count in Adam's bias correction
        (GradScaler does not call optimizer.step at all), so the step counters are taken back -- a few steps late, which
        only touches the bias-correction factor of those few steps, but without ever blocking the host."""
        keep = []
        for ev, host, stepped in self._flag_reads:
            if wait:
                ev.synchronize()
            if ev.query():
                if float(host[0]) != 0.0:
                    self.skipped_steps += 1
                    for p in stepped:       # only the parameters that took part in the dropped step
                        st = self.state.get(p)
                        if st and st.get('step', 0) > 0:
                            st['step'] -= 1
            else:
                keep.append((ev, host, stepped))
        self._flag_reads = keep

    def state_dict(self):
        """(A checkpoint written right after a dropped step must not count it: the pending flags are read first.)"""
        self._collect_flags(wait=True)
        return super().state_dict()

    @torch.no_grad()
    def step(self, closure=None):
        flag = self.skip_flag
        self._skip_ptr = ptr(flag) if flag is not None else None
        self._stepped = []
        self._image_written = False
        if self._flag_reads:
            self._collect_flags()
        # One GPU with the binned table update: the small tensors (MLPs, latent table) take their Adam step on the side
        # stream BESIDE the table passes instead of after them (three 7 us launches at the very end of the step otherwise)
        pipe = self._pipeline
        early = pipe is not None and pipe.pending and SMALL_TENSORS_ON_SIDE_STREAM
        if early:
            if pipe.side is None:
                pipe.side = torch.cuda.Stream()
            ready = torch.cuda.Event()
            ready.record()          # their gradients (and, under data parallelism, the all-reduce of them) are complete
            with torch.cuda.stream(pipe.side):
                pipe.side.wait_event(ready)
                for group in self.param_groups:
                    for p in group['params']:
                        if p is not self._table_param:
                            self._step_param(group, p)
        for group in self.param_groups:
            for p in group['params']:
                if early and p is not self._table_param:
                    continue
                self._step_param(group, p)
        for p in self._stepped:     # (parameters with a derived copy, e.g. the hash table's half-precision image)
            cb = getattr(p, '_nerf_after_update', None)
            if cb is not None and not (p is self._table_param and self._image_written):
                cb()
        self._image_written = False
        model = getattr(self, '_model', None)
        self._cleared_at = model.backward_count if model is not None else None
        if flag is not None:
            host = torch.empty(1, dtype=torch.float64, pin_memory=True)
            host.copy_(flag, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._flag_reads.append((ev, host, self._stepped))
            self.skip_flag = None

    def _step_param(self, group, p):
        b1, b2 = group['betas']
        if p.grad is None or p.numel() == 0:
            return
        st = self.state[p]
        if not st:
            st['step'] = 0
            st['exp_avg'] = torch.zeros_like(p)
            st['exp_avg_sq'] = torch.zeros_like(p)
            # The moments are first used on the pipeline's side stream, which only waits for backward events
            # recorded BEFORE these fills were issued: without this one-time wait the fills can land after the
            # first Adam pass and wipe its moments (v = 0 with m != 0 then divides by eps on the next step).
            torch.cuda.current_stream().synchronize()
        st['step'] += 1
        if getattr(self, '_stepped', None) is not None:
            self._stepped.append(p)
        pipe = self._pipeline
        if p is self._table_param and pipe is not None and pipe.binned_levels:
            self._binned_table_step(p, st, group, b1, b2, pipe)
        if p is self._table_param and self._sharding is not None:
            # level-sharded table: only the owned levels have (complete) gradients and moments here
            self._sharding.dirty = True
            if pipe.pending:
                self._pipelined_table_step(p, st, group, b1, b2, pipe)
            return
        if pipe is not None and p is self._table_param and pipe.pending:
            self._pipelined_table_step(p, st, group, b1, b2, pipe)
            return
        if pipe is not None and p is self._table_param and pipe.binned_levels:
            # no backward reached the table this step: the remaining (small) levels still take their Adam step
            flat_p, flat_g = p.view(-1), p.grad.view(-1)
            flat_m, flat_v = st['exp_avg'].view(-1), st['exp_avg_sq'].view(-1)
            img = self._table_image(p)
            for (l0, cnt) in pipe.groups:
                a, b = pipe.slices[l0]
                self._adam(flat_p[a:b], flat_g[a:b], flat_m[a:b], flat_v[a:b], group, b1, b2, st['step'],
                           image=None if img is None else img[a:b])
            return
        self._adam(p, p.grad, st['exp_avg'], st['exp_avg_sq'], group, b1, b2, st['step'],
                   image=self._table_image(p) if p is self._table_param else None)

    def _adam(self, p, g, m, v, group, b1, b2, step, image=None):
        """`image`: the slice of the hash table's half-precision gather image that mirrors `p` (precision float16): the pass
        then writes it too."""
        if image is not None:
            check(lib().nerf_adam_step_f16(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), group['lr'], b1, b2, group['eps'],
                                           group['weight_decay'], step, self.grad_scale, 1, getattr(self, '_skip_ptr', None),
                                           ptr(image), stream()), 'nerf_adam_step')
            return
        check(lib().nerf_adam_step_guarded(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), group['lr'], b1, b2, group['eps'],
                                           group['weight_decay'], step, self.grad_scale, 1, getattr(self, '_skip_ptr', None),
                                           stream()), 'nerf_adam_step')

    def _table_image(self, p):
        """The half-precision image of the table when the model gathers from one AND it mirrors the master right now: the
        table passes of this step then keep it current themselves (26 B per parameter) instead of leaving it to a
        conversion pass before the next forward.  None otherwise."""
        enc = getattr(getattr(self, '_model', None), 'encoder', None)
        if (enc is None or not getattr(enc, 'half', False) or p is not enc.params or enc.params_half is None
                or enc._half_state != (p._version, p.data_ptr()) or not FUSED_HALF_IMAGE):
            return None
        self._image_written = True
        return enc.params_half

    @torch.no_grad()
    def sync_sharded_state(self):
        """Level-sharded table: collect the Adam moments of every level from its owner (collective; needed before
        the optimizer state is saved, so that a checkpoint can be resumed with any number of ranks)."""
        self._collect_flags(wait=True)
        if self._sharding is not None and self._table_param in self.state and self.state[self._table_param]:
            st = self.state[self._table_param]
            self._sharding.sync_levels(st['exp_avg'])
            self._sharding.sync_levels(st['exp_avg_sq'])

    def table_reduced_by_pipeline(self):
        """True when the pending table gradient is being all-reduced level by level by the pipeline
        (the trainer must then leave the table out of its own gradient all-reduce)."""
        if self._sharding is not None:
            return True     # level-sharded: every owner already holds the complete gradient of its levels
        return self._pipeline is not None and self._pipeline.pending and bool(self._pipeline.works)

    def _binned_table_step(self, p, st, group, b1, b2, pipe):
        """The levels that go through the binned update: one pass that sums the step's corner updates per table
        partition in LDS and applies Adam (main stream; the small levels' scatter -> Adam runs beside it)."""
        cfg = self._model.encoder.cfg
        img = self._table_image(p)
        if pipe.binned_pending is not None and img is not None:
            n_points, nbytes = pipe.binned_pending
            check(lib().nerf_table_update_binned_f16(C.byref(cfg), C.byref(pipe.binned_set), n_points, ptr(pipe.binned_ws),
                                                     nbytes, ptr(p), ptr(st['exp_avg']), ptr(st['exp_avg_sq']), group['lr'],
                                                     b1, b2, group['eps'], group['weight_decay'], st['step'],
                                                     self.grad_scale, getattr(self, '_skip_ptr', None), ptr(img), stream()),
                  'nerf_table_update_binned')
        elif pipe.binned_pending is not None:
            n_points, nbytes = pipe.binned_pending
            check(lib().nerf_table_update_binned_guarded(C.byref(cfg), C.byref(pipe.binned_set), n_points, ptr(pipe.binned_ws),
                                                         nbytes, ptr(p), ptr(st['exp_avg']), ptr(st['exp_avg_sq']), group['lr'],
                                                         b1, b2, group['eps'], group['weight_decay'], st['step'],
                                                         self.grad_scale, getattr(self, '_skip_ptr', None), stream()),
                  'nerf_table_update_binned')
        else:
            # gradient accumulation (several backward passes) or no backward at all: the gradients of these levels
            # are in p.grad (possibly all zero) -- the plain streaming Adam pass, level by level
            flat_p, flat_g = p.view(-1), p.grad.view(-1)
            flat_m, flat_v = st['exp_avg'].view(-1), st['exp_avg_sq'].view(-1)
            F = cfg.n_features
            for l in pipe.binned_levels:
                a = int(cfg.offset[l]) * F
                b = a + int(cfg.entries[l]) * F
                self._adam(flat_p[a:b], flat_g[a:b], flat_m[a:b], flat_v[a:b], group, b1, b2, st['step'],
                           image=None if img is None else img[a:b])
        pipe.drop_binned()

    def _pipelined_table_step(self, p, st, group, b1, b2, pipe):
        main = torch.cuda.current_stream()
        if pipe.side is None:
            pipe.side = torch.cuda.Stream()
        flat_p, flat_g = p.view(-1), p.grad.view(-1)
        flat_m, flat_v = st['exp_avg'].view(-1), st['exp_avg_sq'].view(-1)
        img = self._table_image(p)
        with torch.cuda.stream(pipe.side):
            for (l0, cnt) in pipe.groups:
                a, b = pipe.slices[l0]
                pipe.side.wait_event(pipe.events[l0])
                for ev in pipe.early_events.get(l0, ()):
                    pipe.side.wait_event(ev)
                w = pipe.works.get(l0)
                if w is not None:
                    w.wait()                      # side stream waits for the slice's all-reduce
                self._adam(flat_p[a:b], flat_g[a:b], flat_m[a:b], flat_v[a:b], group, b1, b2, st['step'],
                           image=None if img is None else img[a:b])
        main.wait_stream(pipe.side)
        pipe.pending = False
