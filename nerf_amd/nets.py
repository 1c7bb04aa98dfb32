"""Field network with the reference's surface (nets.py:12-164) on hand-written gfx950 kernels.

``NeRFNetwork`` keeps the constructor arguments, the attribute names used to build optimizer
groups (``encoder``, ``sigma_net``, ``encoder_dir``, ``color_net``, ``latent_emb``; train.py:85-87)
and the ``forward`` / ``density`` signatures.  The tinycudann objects of the reference are
replaced by small modules holding one flat fp32 ``params`` tensor each; the arithmetic runs in
libnerf_hip.so (fused hash-grid gather + MFMA MLP chains), in fp32 throughout.
"""
import ctypes as C
import os
import math

import torch
import torch.nn as nn

from . import _lib
from ._lib import NerfGridCfg, NerfNetCfg, NerfPointSrc, check, f32, i64, lib, ptr, stream


def make_point_src(outer_bound, x=None, d=None, n=None, rays_o=None, rays_d=None, z=None, z_stride=0, n_samples=0,
                   n_points=0, inner_bound=0.0, ray_block=0):
    s = NerfPointSrc()
    s.ray_block = int(ray_block)
    s.mode = 0 if x is not None else 1
    s.n_samples = int(n_samples)
    s.n_points = int(n_points)
    s.x, s.d, s.n = ptr(x), ptr(d), ptr(n)
    s.rays_o, s.rays_d, s.z = ptr(rays_o), ptr(rays_d), ptr(z)
    s.z_stride = int(z_stride)
    s.inner_bound = float(inner_bound)
    s.outer_bound = float(outer_bound)
    return s


class HashGridEncoding(nn.Module):
    """tcnn.Encoding{HashGrid} (nets.py:39-50): flat fp32 table, base_resolution 16, scale 1.3819.

    precision 'float16' (nets.py:31-36,49): the fp32 table stays the master copy that the optimizer updates -- as in
    tinycudann's torch module -- and every kernel that READS the table gathers from a half-precision image of it
    (`params_half`, refreshed by nerf_table_to_half whenever the master has changed since the last read)."""

    def __init__(self, n_levels, n_features_per_level, log2_hashmap_size, base_resolution=16, per_level_scale=1.3819,
                 device='cuda', seed=None, precision='float32'):
        super().__init__()
        self.half = precision == 'float16'
        self.params_half = None
        self._half_state = None         # (version counter, data pointer) of the master the image was made from
        self.cfg = NerfGridCfg()
        total = lib().nerf_grid_cfg_init(C.byref(self.cfg), n_levels, n_features_per_level, log2_hashmap_size,
                                         base_resolution, per_level_scale)
        if total < 0:
            raise ValueError(lib().nerf_last_error().decode())
        self.n_input_dims = 3
        self.n_output_dims = n_levels * n_features_per_level
        self.level_cols = {l: l * n_features_per_level for l in range(n_levels)}    # columns of a level in a feature row
        g = None
        if seed is not None:
            g = torch.Generator(device='cpu').manual_seed(seed)
        init = (torch.rand(total * n_features_per_level, generator=g) * 2 - 1) * 1e-4
        self.params = nn.Parameter(init.to(device))
        # optimizers that write through raw pointers (FusedAdam) cannot bump the tensor's version counter: they call this
        self.params._nerf_after_update = self.mark_updated

    def mark_updated(self):
        self._half_state = None

    def gather_table(self):
        """(entry-point suffix, tensor) of what the forward kernels gather from: ('', fp32 table) or ('_f16', its image)."""
        if not self.half:
            return '', self.params
        state = (self.params._version, self.params.data_ptr())
        if self.params_half is None or self.params_half.device != self.params.device or self._half_state != state:
            if self.params_half is None or self.params_half.device != self.params.device:
                self.params_half = torch.empty(self.params.numel(), dtype=torch.float16, device=self.params.device)
            check(lib().nerf_table_to_half(ptr(self.params), ptr(self.params_half), self.params.numel(), stream()),
                  'nerf_table_to_half')
            self._half_state = state
        return '_f16', self.params_half

    def forward(self, x01):
        return _HashGridFn.apply(x01, self.params, self)


class _HashGridFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x01, params, enc):
        x = f32(x01).reshape(-1, 3)
        out = torch.empty(x.shape[0], enc.n_output_dims, device=x.device, dtype=torch.float32)
        sfx, tab = enc.gather_table() if params is enc.params else ('', params)
        check(getattr(lib(), 'nerf_hashgrid_fwd' + sfx)(C.byref(enc.cfg), ptr(tab), ptr(x), x.shape[0], ptr(out), stream()),
              'nerf_hashgrid_fwd')
        ctx.save_for_backward(x)
        ctx.enc = enc
        ctx.n_params = params.numel()
        return out.reshape(*x01.shape[:-1], enc.n_output_dims)

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        dtable = torch.zeros(ctx.n_params, device=x.device, dtype=torch.float32)
        d = f32(dout).reshape(-1, ctx.enc.n_output_dims)
        check(lib().nerf_hashgrid_bwd(C.byref(ctx.enc.cfg), ptr(x), ptr(d), x.shape[0], ptr(dtable), stream()),
              'nerf_hashgrid_bwd')
        return None, dtable, None


class SHEncoding(nn.Module):
    """tcnn.Encoding{SphericalHarmonics} (nets.py:74-81): no parameters, input in [0,1]^3."""

    def __init__(self, degree):
        super().__init__()
        self.degree = degree
        self.n_output_dims = degree * degree
        self.params = nn.Parameter(torch.zeros(0), requires_grad=False)

    def forward(self, d01):
        d = f32(d01).reshape(-1, 3)
        out = torch.empty(d.shape[0], self.n_output_dims, device=d.device, dtype=torch.float32)
        check(lib().nerf_sh_fwd(ptr(d), d.shape[0], self.degree, ptr(out), stream()), 'nerf_sh_fwd')
        return out.reshape(*d01.shape[:-1], self.n_output_dims)


class FullyFusedMLP(nn.Module):
    """tcnn.Network{FullyFusedMLP} (nets.py:53-63, 96-106): bias-free, ReLU hidden, linear output,
    one flat fp32 ``params`` ([out,in] row-major, layer after layer), Xavier-uniform init."""

    def __init__(self, n_input_dims, n_output_dims, n_neurons, n_hidden_layers, device='cuda', seed=None):
        super().__init__()
        widths = [n_input_dims] + [n_neurons] * n_hidden_layers + [n_output_dims]
        self.widths = widths
        self.n_input_dims, self.n_output_dims = n_input_dims, n_output_dims
        g = None
        if seed is not None:
            g = torch.Generator(device='cpu').manual_seed(seed)
        chunks = []
        for i in range(len(widths) - 1):
            o, n = widths[i + 1], widths[i]
            bound = math.sqrt(6.0 / (o + n))
            chunks.append((torch.rand(o * n, generator=g) * 2 - 1) * bound)
        self.params = nn.Parameter(torch.cat(chunks).to(device))

    def forward(self, x):
        """Stand-alone call with autograd (the training hot path goes through the fused field kernels instead)."""
        return _MLPFn.apply(x, self.params, self)


class _MLPFn(torch.autograd.Function):
    """tcnn.Network as a module of its own: nerf_mlp_fwd / nerf_mlp_bwd."""

    @staticmethod
    def forward(ctx, x, params, mod):
        xx = f32(x).reshape(-1, mod.n_input_dims)
        out = torch.empty(xx.shape[0], mod.n_output_dims, device=xx.device, dtype=torch.float32)
        dims = (C.c_int32 * len(mod.widths))(*mod.widths)
        check(lib().nerf_mlp_fwd(ptr(params), dims, len(mod.widths) - 1, ptr(xx), xx.shape[0], ptr(out), stream()),
              'nerf_mlp_fwd')
        ctx.save_for_backward(xx, params)
        ctx.mod, ctx.in_shape, ctx.need_dx = mod, x.shape, x.requires_grad
        return out.reshape(*x.shape[:-1], mod.n_output_dims)

    @staticmethod
    def backward(ctx, dout):
        xx, params = ctx.saved_tensors
        mod = ctx.mod
        rows = xx.shape[0]
        dims = (C.c_int32 * len(mod.widths))(*mod.widths)
        n_ws = lib().nerf_mlp_bwd_workspace_floats(dims, len(mod.widths) - 1, rows)
        ws = torch.empty(max(int(n_ws), 1), device=xx.device, dtype=torch.float32)
        dparams = torch.zeros_like(params)
        dx = torch.empty_like(xx) if ctx.need_dx else None
        d = f32(dout).reshape(rows, mod.n_output_dims)
        check(lib().nerf_mlp_bwd(ptr(params), dims, len(mod.widths) - 1, ptr(xx), ptr(d), rows, ptr(dx), ptr(dparams), ptr(ws),
                                 ws.numel(), stream()), 'nerf_mlp_bwd')
        return (dx.reshape(ctx.in_shape) if dx is not None else None), dparams, None


def _slice_src_kwargs(kw, a, b, unit):
    """make_point_src kwargs restricted to rays [a, b) (mode 1) or points [a, b) (mode 0, unit = 1)."""
    out = dict(kw)
    if kw.get('x') is not None:
        out.update(x=kw['x'][a:b], d=kw['d'][a:b] if kw.get('d') is not None else None,
                   n=kw['n'][a:b] if kw.get('n') is not None else None, n_points=b - a)
        return out
    zs = kw.get('z_stride', 0)
    out.update(rays_o=kw['rays_o'][a:b], rays_d=kw['rays_d'][a:b], z=kw['z'] if zs == 0 else kw['z'][a:b],
               n=kw['n'][a:b] if kw.get('n') is not None else None, n_points=(b - a) * unit)
    return out


EARLY_BIN_COUNT = os.environ.get('NERF_EARLY_BIN_COUNT', '1') != '0'
SHARD_COUNT_STREAM = os.environ.get('NERF_SHARD_COUNT_STREAM', 'scatter')
SHARD_COUNT_AT = os.environ.get('NERF_SHARD_COUNT_AT', 'last')      # level sharding: 'first' = before the first encode


class _FieldFn(torch.autograd.Function):
    """sigma, rgb = field(points); backward = nerf_field_bwd (gradients for table, MLPs, latent)."""

    @staticmethod
    def forward(ctx, table, sigma_w, color_w, latent, net, src_kwargs, want_xyz):
        src = make_point_src(**src_kwargs)
        M = src.n_points
        dev = table.device
        sigma = torch.empty(M, device=dev, dtype=torch.float32)
        rgb = torch.empty(M, 3, device=dev, dtype=torch.float32)
        xh = torch.empty(M, net.encoder.n_output_dims, device=dev, dtype=torch.float32)
        xyz = torch.empty(M, 3, device=dev, dtype=torch.float32) if want_xyz else None
        lat = latent if latent is not None else None
        pipe = net.table_pipeline
        ctx.count_tag = None
        if pipe is not None and net.direct_grads and pipe.binned_levels and ctx.needs_input_grad[0] and EARLY_BIN_COUNT:
            # the count + scan passes of the binned table update, issued BEFORE the field kernel so that they run beside it
            ctx.count_tag = net.forward_count = getattr(net, 'forward_count', 0) + 1
            pipe.count_early(net.encoder.cfg, src, ctx.count_tag, dev)
        sfx, tab = net.encoder.gather_table()
        check(getattr(lib(), 'nerf_field_fwd' + sfx)(C.byref(net.encoder.cfg), C.byref(net.net_cfg), ptr(tab), ptr(sigma_w),
                                                     ptr(color_w), ptr(lat), net.N, C.byref(src), ptr(sigma), ptr(rgb), ptr(xh),
                                                     ptr(xyz), stream()), 'nerf_field_fwd')
        ctx.net, ctx.src_kwargs = net, src_kwargs
        ctx.save_for_backward(table, sigma_w, color_w, latent, xh)
        ctx.mark_non_differentiable(xh)
        ctx.set_materialize_grads(False)
        if xyz is not None:
            ctx.mark_non_differentiable(xyz)
            return sigma, rgb, xh, xyz
        return sigma, rgb, xh, None

    @staticmethod
    def backward(ctx, dsigma, drgb, _dxh, _dxyz):
        table, sigma_w, color_w, latent, xh = ctx.saved_tensors
        net = ctx.net
        net.backward_count += 1
        src = make_point_src(**ctx.src_kwargs)
        dev = table.device
        if dsigma is None:
            dsigma = torch.zeros(src.n_points, device=dev)
        if drgb is None:
            drgb = torch.zeros(src.n_points, 3, device=dev)
        # With direct gradients (FusedAdam.attach) the kernel accumulates straight into the tensors that
        # param.grad aliases and autograd gets None; otherwise fresh zero-filled gradients are returned.
        direct = net.direct_grads
        if direct:
            dtable, dsw, dcw = table.grad, sigma_w.grad, color_w.grad
            dlat = latent.grad if latent is not None else None
        else:
            dtable, dsw, dcw = torch.zeros_like(table), torch.zeros_like(sigma_w), torch.zeros_like(color_w)
            dlat = torch.zeros_like(latent) if latent is not None else None
        dfeat = torch.empty_like(xh)
        dsigma, drgb = f32(dsigma), f32(drgb)
        pipe = net.table_pipeline
        unit = src.n_samples if src.mode == 1 else 1            # points per ray (chunks are cut between rays)
        rows = src.n_points // unit
        cut = int(rows * pipe.early_fraction) if pipe is not None else 0
        if pipe is not None and direct and pipe.binned_levels:
            # binned table update: one backward launch; the corner updates of the large levels are sorted into table
            # partitions on this stream (FusedAdam.step sums and applies them in one pass), the small dense levels
            # scatter on their own stream beside it
            check(lib().nerf_field_bwd(C.byref(net.encoder.cfg), C.byref(net.net_cfg), ptr(table), ptr(sigma_w), ptr(color_w),
                                       ptr(latent), net.N, C.byref(src), ptr(xh), ptr(dsigma), ptr(drgb),
                                       ptr(dfeat), None, ptr(dsw), ptr(dcw), ptr(dlat), stream()), 'nerf_field_bwd')
            pipe.begin(dtable)
            pipe.keep = [dfeat, dsigma, drgb, xh, ctx.src_kwargs]
            if pipe.groups:
                if pipe.scatter_stream is None:
                    pipe.scatter_stream = torch.cuda.Stream()
                done = torch.cuda.Event()
                done.record()
                # the small levels' LDS scatter beside the bin passes: the first group (the chunked levels) on the scatter
                # stream, a second group (the one- and two-chunk levels, one launch) on the stream the early count pass used
                streams = [pipe.scatter_stream]
                if len(pipe.groups) > 1:
                    if pipe.count_stream is None:
                        pipe.count_stream = torch.cuda.Stream()
                    streams.append(pipe.count_stream)
                for gi, (l0, cnt) in enumerate(pipe.groups):
                    st_ = streams[min(gi, len(streams) - 1)]
                    with torch.cuda.stream(st_):
                        st_.wait_event(done)
                        check(lib().nerf_hashgrid_scatter(C.byref(net.encoder.cfg), C.byref(src), ptr(dfeat), ptr(dtable), l0, cnt,
                                                          stream()), 'nerf_hashgrid_scatter')
                        pipe.level_done(l0, cnt)
            pipe.bin(net.encoder.cfg, src, dfeat, dfeat.shape[1], net.encoder.level_cols, dtable, keepalive=ctx.src_kwargs,
                     tag=ctx.count_tag)
        elif pipe is None or not direct or cut * unit < 65536 or (rows - cut) * unit < 65536:
            check(lib().nerf_field_bwd(C.byref(net.encoder.cfg), C.byref(net.net_cfg), ptr(table), ptr(sigma_w), ptr(color_w),
                                       ptr(latent), net.N, C.byref(src), ptr(xh), ptr(dsigma), ptr(drgb),
                                       ptr(dfeat), None, ptr(dsw), ptr(dcw), ptr(dlat), stream()), 'nerf_field_bwd')
            net.scatter_table(src, dfeat, dtable, keepalive=ctx.src_kwargs)
        else:
            # Two chunks of rays: the table scatter of the first (bound by memory-side atomics) runs on its own
            # stream underneath the MLP backward of the second (bound by the matrix cores).
            pipe.begin(dtable)
            pipe.keep = [dfeat, dsigma, drgb, xh]
            if pipe.scatter_stream is None:
                pipe.scatter_stream = torch.cuda.Stream()
            main = torch.cuda.current_stream()
            for k, (a, b) in enumerate(((0, cut), (cut, rows))):
                kw = _slice_src_kwargs(ctx.src_kwargs, a, b, unit)
                sub = make_point_src(**kw)
                pa, pb = a * unit, b * unit
                check(lib().nerf_field_bwd(C.byref(net.encoder.cfg), C.byref(net.net_cfg), None, ptr(sigma_w), ptr(color_w),
                                           ptr(latent), net.N, C.byref(sub), ptr(xh[pa:pb]), ptr(dsigma[pa:pb]),
                                           ptr(drgb[pa:pb]), ptr(dfeat[pa:pb]), None, ptr(dsw), ptr(dcw), ptr(dlat),
                                           stream()), 'nerf_field_bwd')
                if k == 0:
                    done = torch.cuda.Event()
                    done.record(main)
                    with torch.cuda.stream(pipe.scatter_stream):
                        pipe.scatter_stream.wait_event(done)
                        net.scatter_levels_pipelined(sub, dfeat[pa:pb], dtable, final=False)
                else:
                    net.scatter_levels_pipelined(sub, dfeat[pa:pb], dtable, final=True)
                pipe.keep.append(kw)
        if direct:
            return None, None, None, None, None, None, None
        return dtable, dsw, dcw, dlat, None, None, None


class _DensityInputGradFn(torch.autograd.Function):
    """sigma = density(x) with the gradient w.r.t. the positions only (nerf_density_bwd_input): what the pose
    alignment needs (allign/trainer.py:174-208: two frozen fields, the rigid transform is optimised).  Parameter
    gradients flow through forward(), not through density()."""

    @staticmethod
    def forward(ctx, x, net, outer_bound):
        xx = f32(x).reshape(-1, 3)
        sigma, xh = net.density_src(make_point_src(outer_bound, x=xx, n_points=xx.shape[0]), want_features=True)
        ctx.save_for_backward(xx)
        ctx.net, ctx.outer_bound, ctx.shape = net, outer_bound, x.shape
        ctx.mark_non_differentiable(xh)
        ctx.set_materialize_grads(False)
        return sigma, xh

    @staticmethod
    def backward(ctx, dsigma, _dxh):
        if dsigma is None:
            return None, None, None
        (xx,) = ctx.saved_tensors
        net = ctx.net
        src = make_point_src(ctx.outer_bound, x=xx, n_points=xx.shape[0])
        dx = torch.empty_like(xx)
        sfx, tab = net.encoder.gather_table()
        check(getattr(lib(), 'nerf_density_bwd_input' + sfx)(C.byref(net.encoder.cfg), C.byref(net.net_cfg), ptr(tab),
                                                             ptr(net.sigma_net.params), C.byref(src),
                                                             ptr(f32(dsigma).reshape(-1)), ptr(dx), stream()),
              'nerf_density_bwd_input')
        return dx.reshape(ctx.shape), None, None


class _ShardedFieldFn(torch.autograd.Function):
    """_FieldFn under a level-sharded table (level_parallel.py): features arrive by all-to-all from the level
    owners, d(features) returns the same way; only this rank's levels receive table gradient."""

    @staticmethod
    def forward(ctx, table, sigma_w, color_w, latent, net, src_kwargs, want_xyz):
        sh = net.level_sharding
        src = make_point_src(**src_kwargs)
        M = src.n_points
        dev = table.device
        plan = sh.plan(src_kwargs)
        pipe = net.table_pipeline
        count_now = None
        if (pipe is not None and net.direct_grads and pipe.binned_levels and ctx.needs_input_grad[0] and EARLY_BIN_COUNT
                and sh.width[sh.rank]):
            # count + scan passes of the owner's binned levels over the points of ALL ranks: under this forward pass, once the
            # last group's features are on the wire (before that they would delay an exchange by sharing the CUs with an encode)
            def count_now():
                plan['count_tag'] = net.forward_count = getattr(net, 'forward_count', 0) + 1
                # (on the scatter stream, idle in a forward pass: with the collectives' stream a stream of its own would be the
                # fifth of this process, and the hardware has four queues: it would share one with the exchange it runs beside)
                if pipe.scatter_stream is None:
                    pipe.scatter_stream = torch.cuda.Stream()
                pipe.count_early(net.encoder.cfg, make_point_src(**plan['kw_all']), plan['count_tag'], dev,
                                 on=pipe.scatter_stream if SHARD_COUNT_STREAM == 'scatter' else None)
            if SHARD_COUNT_AT == 'first':
                for g_ in plan['groups']:       # (pre-gathered depths: the count pass reads the points of every group)
                    if g_.get('ready') is not None:
                        g_['ready']()
                        g_['ready'] = None
                count_now()
                count_now = None
        sigma = torch.empty(M, device=dev, dtype=torch.float32)
        rgb = torch.empty(M, 3, device=dev, dtype=torch.float32)
        xyz = torch.empty(M, 3, device=dev, dtype=torch.float32) if want_xyz else None
        xh = torch.empty(M, net.encoder.n_output_dims, device=dev, dtype=torch.float32)
        # group by group: the MLPs of a group run while the next group's features are on the wire
        for grp in sh.encode_exchange_groups(table, plan, xh, after_last_start=count_now):
            a, b = grp['a'], grp['b']
            sub = src if len(plan['groups']) == 1 else make_point_src(**grp['kw_local'])
            check(lib().nerf_field_from_features(C.byref(net.encoder.cfg), C.byref(net.net_cfg), ptr(sigma_w), ptr(color_w),
                                                 ptr(latent), net.N, C.byref(sub), ptr(xh[a:b]), ptr(sigma[a:b]),
                                                 ptr(rgb[a:b]), ptr(xyz[a:b]) if xyz is not None else None,
                                                 stream()), 'nerf_field_from_features')
        ctx.net, ctx.src_kwargs, ctx.plan = net, src_kwargs, plan
        ctx.save_for_backward(table, sigma_w, color_w, latent, xh)
        ctx.mark_non_differentiable(xh)
        ctx.set_materialize_grads(False)
        if xyz is not None:
            ctx.mark_non_differentiable(xyz)
        return sigma, rgb, xh, xyz

    @staticmethod
    def backward(ctx, dsigma, drgb, _dxh, _dxyz):
        table, sigma_w, color_w, latent, xh = ctx.saved_tensors
        net, sh = ctx.net, ctx.net.level_sharding
        net.backward_count += 1
        src = make_point_src(**ctx.src_kwargs)
        dev = table.device
        if dsigma is None:
            dsigma = torch.zeros(src.n_points, device=dev)
        if drgb is None:
            drgb = torch.zeros(src.n_points, 3, device=dev)
        direct = net.direct_grads
        if direct:
            dtable, dsw, dcw = table.grad, sigma_w.grad, color_w.grad
            dlat = latent.grad if latent is not None else None
        else:
            dtable, dsw, dcw = torch.zeros_like(table), torch.zeros_like(sigma_w), torch.zeros_like(color_w)
            dlat = torch.zeros_like(latent) if latent is not None else None
        dfeat = torch.empty_like(xh)
        dsigma, drgb = f32(dsigma), f32(drgb)
        plan = ctx.plan
        if len(plan['groups']) == 1:
            check(lib().nerf_field_bwd(C.byref(net.encoder.cfg), C.byref(net.net_cfg), None, ptr(sigma_w), ptr(color_w),
                                       ptr(latent), net.N, C.byref(src), ptr(xh), ptr(dsigma), ptr(drgb),
                                       ptr(dfeat), None, ptr(dsw), ptr(dcw), ptr(dlat), stream()), 'nerf_field_bwd')
            sh.exchange_scatter(dfeat, plan['kw_all'], dtable, net.table_pipeline)
        else:
            # group by group: d(features) of a group travels to the level owners under the backward of the next one
            state = None
            for grp in plan['groups']:
                a, b = grp['a'], grp['b']
                sub = make_point_src(**grp['kw_local'])
                check(lib().nerf_field_bwd(C.byref(net.encoder.cfg), C.byref(net.net_cfg), None, ptr(sigma_w), ptr(color_w),
                                           ptr(latent), net.N, C.byref(sub), ptr(xh[a:b]), ptr(dsigma[a:b]), ptr(drgb[a:b]),
                                           ptr(dfeat[a:b]), None, ptr(dsw), ptr(dcw), ptr(dlat), stream()),
                      'nerf_field_bwd')
                state = sh.dfeat_exchange_start(plan, grp, dfeat[a:b], state, dtable, net.table_pipeline)
            sh.dfeat_exchange_finish(plan, state, dtable, net.table_pipeline)
        if direct:
            return None, None, None, None, None, None, None
        return dtable, dsw, dcw, dlat, None, None, None


class NeRFNetwork(nn.Module):
    """Same constructor and call surface as the reference NeRFNetwork (nets.py:12-164)."""

    def __init__(self, N, encoding_precision, encoding_n_levels, encoding_n_features_per_level,
                 encoding_log2_hashmap_size, geo_feat_dim, sigma_hidden_dim, sigma_num_layers, encoding_dir_precision,
                 encoding_dir_encoding, encoding_dir_degree, latent_embedding_dim, color_hidden_dim, color_num_layers,
                 device='cuda', seed=None):
        super().__init__()
        lib()  # fail loudly, before any allocation, when the HIP library is missing
        if encoding_precision not in ('float16', 'float32'):
            raise ValueError('encoding_precision must be float16 or float32')   # nets.py:31-36
        if encoding_dir_degree != 0 and encoding_dir_precision not in ('float16', 'float32'):
            raise ValueError('encoding_dir_precision must be float16 or float32')   # nets.py:67-72
        if encoding_dir_degree != 0 and encoding_dir_encoding != 'SphericalHarmonics':
            raise NotImplementedError('only the SphericalHarmonics direction encoding is implemented')
        # the fused kernels cover exactly this shape (csrc/field_fwd.hip nerf_check_cfg): reject anything else here,
        # before the 1.6 GB table is allocated, instead of at the first forward
        if (sigma_hidden_dim != 64 or color_hidden_dim != 64 or sigma_num_layers != 2 or color_num_layers != 3
                or geo_feat_dim != 15):
            raise NotImplementedError('fused kernels need sigma/color hidden_dim 64, sigma_num_layers 2, color_num_layers 3, '
                                      'geo_feat_dim 15 (got %s/%s, %s, %s, %s)' % (sigma_hidden_dim, color_hidden_dim,
                                                                                   sigma_num_layers, color_num_layers, geo_feat_dim))
        if encoding_dir_degree not in (0, 2, 4) or latent_embedding_dim % 2 or not 0 <= latent_embedding_dim <= 48:
            raise NotImplementedError('fused kernels need SH degree 0, 2 or 4 and an even latent_embedding_dim <= 48')
        # precision (nets.py:31-36): 'float16' for the hash grid = gathers from a half-precision image of the fp32 master
        # table (HashGridEncoding); blends, MLPs, activations and optimizer state stay fp32 (the CPU/fp32 semantics are the
        # parity spec).  The direction encoding has no table: its 'float16' only rounds tinycudann's output,
        # here it is evaluated in fp32 -- said out loud once, not silently.
        if encoding_dir_degree != 0 and encoding_dir_precision == 'float16':
            import warnings
            warnings.warn("nerf_amd: encoding_dir_precision 'float16' is evaluated in float32", stacklevel=2)
        self.encoding_precision = torch.float16 if encoding_precision == 'float16' else torch.float32
        sd = (lambda k: None if seed is None else seed + k)
        self.encoder = HashGridEncoding(encoding_n_levels, encoding_n_features_per_level, encoding_log2_hashmap_size,
                                        device=device, seed=sd(0), precision=encoding_precision)
        self.sigma_net = FullyFusedMLP(self.encoder.n_output_dims, 1 + geo_feat_dim, sigma_hidden_dim,
                                       sigma_num_layers - 1, device=device, seed=sd(1))
        if encoding_dir_degree != 0:
            self.encoder_dir = SHEncoding(encoding_dir_degree)
            self.directional_dim = self.encoder_dir.n_output_dims
        else:
            self.directional_dim = 0
        self.N = N
        self.latent_emb_dim = latent_embedding_dim          # (the reference's `latend_emb_dim` typo, nets.py:92, fixed)
        if latent_embedding_dim != 0:
            self.latent_emb = nn.Parameter(torch.zeros(N, latent_embedding_dim, device=device), requires_grad=True)
        else:
            self.latent_emb = None
        self.in_dim_color = geo_feat_dim + self.directional_dim + self.latent_emb_dim
        self.color_net = FullyFusedMLP(self.in_dim_color, 3, color_hidden_dim, color_num_layers - 1, device=device,
                                       seed=sd(2))
        if sigma_hidden_dim != color_hidden_dim:
            raise NotImplementedError('fused kernels need sigma_hidden_dim == color_hidden_dim')
        self.net_cfg = NerfNetCfg(sigma_hidden_dim, sigma_num_layers, geo_feat_dim, color_num_layers, encoding_dir_degree,
                                  latent_embedding_dim)
        self.direct_grads = False
        self.backward_count = 0         # backward passes that reached the field (FusedAdam.zero_grad keys its fast path on it)
        self.table_pipeline = None
        self.level_sharding = None      # level_parallel.LevelSharding under multi-GPU training
        self._local_table = 0           # >0: inside `with model.local_table()` (replicated-table kernels)

    # ---- gradient plumbing --------------------------------------------------------------
    def enable_direct_grads(self, on=True):
        """When on, every parameter must already own a persistent ``.grad`` tensor (FusedAdam.attach does
        that); the backward kernel then accumulates into it in place and skips autograd's copies."""
        if on:
            for p in (self.encoder.params, self.sigma_net.params, self.color_net.params, self.latent_emb):
                if p is not None and p.grad is None:
                    p.grad = torch.zeros_like(p)
        self.direct_grads = bool(on)

    def scatter_table(self, src, dfeat, dtable, keepalive=None):
        """dtable += scatter(dfeat).  With a table pipeline attached (FusedAdam.attach) the scatter is issued
        level by level, big hashed levels first, with an event after each level so that the optimizer (and,
        under data parallelism, the gradient all-reduce) of a level can start while later levels still scatter."""
        cfg = self.encoder.cfg
        pipe = self.table_pipeline
        if pipe is None or not self.direct_grads:       # (the pipeline works on param.grad itself)
            check(lib().nerf_hashgrid_scatter(C.byref(cfg), C.byref(src), ptr(dfeat), ptr(dtable), 0, cfg.n_levels, stream()),
                  'nerf_hashgrid_scatter')
            return
        pipe.begin(dtable)
        self.scatter_levels_pipelined(src, dfeat, dtable, final=True)
        pipe.bin(cfg, src, dfeat, dfeat.shape[1], self.encoder.level_cols, dtable, keepalive=keepalive)

    def scatter_levels_pipelined(self, src, dfeat, dtable, final):
        """One pass over the pipeline's level groups on the current stream; `final` marks the last part of the
        step's scatter (its events release the all-reduce / Adam of each group)."""
        cfg, pipe = self.encoder.cfg, self.table_pipeline
        for (l0, cnt) in pipe.groups:
            check(lib().nerf_hashgrid_scatter(C.byref(cfg), C.byref(src), ptr(dfeat), ptr(dtable), l0, cnt, stream()),
                  'nerf_hashgrid_scatter')
            if final:
                pipe.level_done(l0, cnt)
            else:
                pipe.level_partial(l0)

    # ---- level-sharded table ----------------------------------------------------------------
    def shard_levels(self, owner=None):
        """Shard the hash table by level over the process group (level_parallel.py)."""
        from .level_parallel import LevelSharding
        self.level_sharding = LevelSharding(self.encoder.cfg, self.encoder.params.device, owner=owner)
        self.level_sharding.encoder = self.encoder      # (its encode passes gather from the encoder's half-precision image)
        return self.level_sharding

    def _sharded(self):
        sh = self.level_sharding
        return sh is not None and (sh.world > 1 or sh.force) and self._local_table == 0

    def sync_table(self):
        if self.level_sharding is not None:
            self.level_sharding.sync_table(self.encoder.params.data)
            self.encoder.mark_updated()     # (broadcasts into .data do not bump the parameter's version counter)

    def local_table(self):
        """Context: make the local table whole (broadcast of the levels other ranks own, if stale) and run the
        single-GPU kernels on it -- for inference, extraction and checkpoints, whose work is sharded by pixel."""
        net = self

        class _Ctx(object):
            def __enter__(self_):
                net.sync_table()
                net._local_table += 1

            def __exit__(self_, *a):
                net._local_table -= 1
        return _Ctx()

    # ---- reference surface ----------------------------------------------------------------
    def _field(self, src_kwargs, want_xyz=False):
        if self._sharded():
            return _ShardedFieldFn.apply(self.encoder.params, self.sigma_net.params, self.color_net.params,
                                         self.latent_emb, self, src_kwargs, want_xyz)
        return _FieldFn.apply(self.encoder.params, self.sigma_net.params, self.color_net.params, self.latent_emb, self,
                              src_kwargs, want_xyz)

    @torch.no_grad()
    def field_nograd(self, src_kwargs):
        """sigma, rgb for inference: no autograd node, and neither x_hashtable (160 B/point) nor xyz is written."""
        with self.local_table():
            src = make_point_src(**src_kwargs)
            dev = self.encoder.params.device
            sigma = torch.empty(src.n_points, device=dev, dtype=torch.float32)
            rgb = torch.empty(src.n_points, 3, device=dev, dtype=torch.float32)
            sfx, tab = self.encoder.gather_table()
            check(getattr(lib(), 'nerf_field_fwd' + sfx)(C.byref(self.encoder.cfg), C.byref(self.net_cfg), ptr(tab),
                                                         ptr(self.sigma_net.params), ptr(self.color_net.params),
                                                         ptr(self.latent_emb), self.N, C.byref(src), ptr(sigma), ptr(rgb),
                                                         None, None, stream()), 'nerf_field_fwd')
        return sigma, rgb

    def forward(self, x, d, n, outer_bound):
        """nets.py:109-146.  x: contracted points [...,3], d: unit dirs [...,3], n: image index [...]."""
        prefix = x.shape[:-1]
        xx, dd, nn_ = f32(x).reshape(-1, 3), f32(d).reshape(-1, 3), i64(n).reshape(-1)
        sigma, rgb, xh, _ = self._field(dict(outer_bound=outer_bound, x=xx, d=dd, n=nn_, n_points=xx.shape[0]))
        return sigma.reshape(*prefix), rgb.reshape(*prefix, 3), {'x_hashtable': xh.reshape(*prefix, -1).detach()}

    def density(self, x, outer_bound):
        """nets.py:148-164.  The renderer calls it without grad (renderer.py:145,152); when x requires grad (pose
        alignment, allign/trainer.py:188) sigma carries the gradient w.r.t. x (not w.r.t. the parameters)."""
        prefix = x.shape[:-1]
        if torch.is_grad_enabled() and x.requires_grad:
            with self.local_table():
                sigma, xh = _DensityInputGradFn.apply(x, self, outer_bound)
            return sigma.reshape(*prefix), {'x_hashtable': xh.reshape(*prefix, -1)}
        with torch.no_grad():
            return self._density_nograd(x, outer_bound)

    @torch.no_grad()
    def _density_nograd(self, x, outer_bound):
        prefix = x.shape[:-1]
        xx = f32(x).reshape(-1, 3)
        sigma, xh = self.density_kw(dict(outer_bound=outer_bound, x=xx, n_points=xx.shape[0]), want_features=True)
        return sigma.reshape(*prefix), {'x_hashtable': xh.reshape(*prefix, -1)}

    @torch.no_grad()
    def density_kw(self, src_kwargs, want_features=False, per_group=None):
        """density() on a point source given as make_point_src keyword arguments.
        per_group(i, group, sigma_rows): level-sharded table with ray groups only -- called as soon as the densities of
        group i (points [group['a'], group['b']), rays [group['ra'], group['rb'])) exist, while later groups are still on
        the wire (the renderer resamples the group and starts its depth all-gather there)."""
        if not self._sharded():
            return self.density_src(make_point_src(**src_kwargs), want_features)
        sh = self.level_sharding
        M = int(src_kwargs['n_points'])
        plan = sh.plan(src_kwargs)
        dev = self.encoder.params.device
        xh = torch.empty(M, self.encoder.n_output_dims, device=dev, dtype=torch.float32)
        sigma = torch.empty(M, device=dev, dtype=torch.float32)
        for gi, grp in enumerate(sh.encode_exchange_groups(self.encoder.params, plan, xh)):
            a, b = grp['a'], grp['b']
            check(lib().nerf_density_from_features(C.byref(self.encoder.cfg), C.byref(self.net_cfg), ptr(self.sigma_net.params),
                                                   ptr(xh[a:b]), b - a, ptr(sigma[a:b]), stream()),
                  'nerf_density_from_features')
            if per_group is not None:
                per_group(gi, grp, sigma[a:b])
        return sigma, (xh if want_features else None)

    @torch.no_grad()
    def density_src(self, src, want_features=False):
        dev = self.encoder.params.device
        sigma = torch.empty(src.n_points, device=dev, dtype=torch.float32)
        xh = torch.empty(src.n_points, self.encoder.n_output_dims, device=dev) if want_features else None
        sfx, tab = self.encoder.gather_table()
        check(getattr(lib(), 'nerf_density_fwd' + sfx)(C.byref(self.encoder.cfg), C.byref(self.net_cfg), ptr(tab),
                                                       ptr(self.sigma_net.params), C.byref(src), ptr(sigma), ptr(xh), stream()),
              'nerf_density_fwd')
        return sigma, xh
