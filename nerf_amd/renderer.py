"""Renderer with the reference's surface (renderer.py:10-194) on gfx950 kernels.

Every method of the reference ``NerfRenderer`` is here with the same name, argument order and
return values; each one is a thin host wrapper around a kernel of libnerf_hip.so.  ``render``
and ``efficient_sampling`` take the fused route when the model is this package's
``NeRFNetwork`` (ray -> sample point -> contraction -> hash grid -> MLP in one launch, no
[N,S,3] intermediates); with any other model object (e.g. an analytic stub) they fall back to
the method-by-method route, which still runs every renderer stage in HIP.

Additive, optional arguments only: ``noise`` (the stratification noise of renderer.py:102,
injected for bit-exact index parity) and ``generator``.
"""
import ctypes as C
import math as m
import os

import torch
import torch.nn as nn

from ._lib import check, f32, i64, lib, ptr, stream
from .nets import NeRFNetwork, make_point_src

# level-sharded table with ray groups: resample and all-gather the fine depths group by group from inside the coarse pass
# (NerfRenderer._efficient_sampling_by_groups); '0': one all-gather of all depths between the passes
PIPELINED_DEPTH_GATHER = os.environ.get('NERF_PIPELINED_DEPTH_GATHER', '1') != '0'


class _CompositeFn(torch.autograd.Function):
    """renderer.py:71-85 + :181-182 as one kernel, with its hand-written backward."""

    @staticmethod
    def forward(ctx, sigmas, rgbs, z_vals, z_vals_log, bg, lo, rng):
        sig, rgb, z, zl, bgc = f32(sigmas), f32(rgbs), f32(z_vals), f32(z_vals_log), f32(bg)
        N, S = sig.shape
        dev = sig.device
        image = torch.empty(N, 3, device=dev)
        invdepth = torch.empty(N, device=dev)
        weights = torch.empty(N, S, device=dev)
        zlog_s = torch.empty(N, S, device=dev)
        check(lib().nerf_composite_fwd(ptr(sig), ptr(rgb), ptr(z), ptr(zl), ptr(bgc), N, S, lo, rng, ptr(image),
                                       ptr(invdepth), ptr(weights), ptr(zlog_s), stream()), 'nerf_composite_fwd')
        ctx.save_for_backward(sig, rgb, zl, bgc)
        ctx.mark_non_differentiable(invdepth, zlog_s)
        ctx.set_materialize_grads(False)
        return image, invdepth, weights, zlog_s

    @staticmethod
    def backward(ctx, dimage, _dinv, dweights, _dzs):
        sig, rgb, zl, bgc = ctx.saved_tensors
        N, S = sig.shape
        if dimage is None and dweights is None:
            return None, None, None, None, None, None, None
        if dimage is None:
            dimage = torch.zeros(N, 3, device=sig.device)
        dsig = torch.empty_like(sig)
        drgb = torch.empty_like(rgb)
        dw = f32(dweights) if dweights is not None else None
        check(lib().nerf_composite_bwd(ptr(sig), ptr(rgb), ptr(zl), ptr(bgc), ptr(f32(dimage)), ptr(dw), N, S, ptr(dsig),
                                       ptr(drgb), stream()), 'nerf_composite_bwd')
        return dsig, drgb, None, None, None, None, None


class _CompositeLossFn(torch.autograd.Function):
    """_CompositeFn followed by helpers._LossFn ('total') as ONE kernel, and the backward of both as one more
    (nerf_composite_loss_fwd / _bwd): the training step's tail without the weights / zlog_s round trip, without dimage /
    dweights and without five small launches.  Only the loss is differentiable; image, weights, ... come out detached."""

    @staticmethod
    def forward(ctx, sigmas, rgbs, z_vals, z_vals_log, bg, gt, lo, rng, lam_dist, lam_depth):
        sig, rgb, z, zl, bgc, g = f32(sigmas), f32(rgbs), f32(z_vals), f32(z_vals_log), f32(bg), f32(gt)
        N, S = sig.shape
        dev = sig.device
        image = torch.empty(N, 3, device=dev)
        invdepth = torch.empty(N, device=dev)
        weights = torch.empty(N, S, device=dev)
        zlog_s = torch.empty(N, S, device=dev)
        sums = torch.empty(8 + 4 * 4096, device=dev, dtype=torch.float64)     # NERF_LOSS_SUMS_DOUBLES
        out = torch.empty(6, device=dev)
        check(lib().nerf_composite_loss_fwd(ptr(sig), ptr(rgb), ptr(z), ptr(zl), ptr(bgc), ptr(g), N, S, lo, rng, lam_dist,
                                            lam_depth, ptr(image), ptr(invdepth), ptr(weights), ptr(zlog_s), ptr(sums),
                                            ptr(out), stream()), 'nerf_composite_loss_fwd')
        loss, parts, flag = out[4], out[:4], sums[5:6]
        ctx.save_for_backward(sig, rgb, zl, bgc, g, image, flag)
        ctx.consts = (lo, rng, lam_dist, lam_depth)
        ctx.mark_non_differentiable(parts, flag, image, invdepth, weights, zlog_s)
        ctx.set_materialize_grads(False)
        return loss, parts, flag, image, invdepth, weights, zlog_s

    @staticmethod
    def backward(ctx, dout, *_unused):
        if dout is None:
            return (None,) * 10
        sig, rgb, zl, bgc, g, image, flag = ctx.saved_tensors
        lo, rng, lam_dist, lam_depth = ctx.consts
        N, S = sig.shape
        dsig = torch.empty_like(sig)
        drgb = torch.empty_like(rgb)
        # (the flag: a finite loss whose gradients are not -- an Inf upstream gradient, an overflow -- drops the step too)
        check(lib().nerf_composite_loss_bwd(ptr(sig), ptr(rgb), ptr(zl), ptr(bgc), ptr(g), ptr(image), ptr(f32(dout)), N, S,
                                            lo, rng, lam_dist, lam_depth, ptr(dsig), ptr(drgb), flag.data_ptr(), stream()),
              'nerf_composite_loss_bwd')
        return (dsig, drgb) + (None,) * 8


RAY_BLOCK = int(os.environ.get('NERF_RAY_BLOCK', '32'))


class NerfRenderer(nn.Module):
    def __init__(self, model, inner_bound, outer_bound, z_bounds, steps_firstpass, steps_importance, alpha_importance):
        super().__init__()
        lib()
        self.model = model
        self.inner_bound = inner_bound
        self.outer_bound = outer_bound
        self.z_bounds = list(z_bounds)
        self.steps_firstpass = list(steps_firstpass)
        self.steps_importance = steps_importance
        self.alpha_importance = alpha_importance
        if len(self.z_bounds) != len(self.steps_firstpass) + 1:
            raise ValueError('len(z_bounds) must be len(steps) + 1')
        self._tables = {}
        # set by the Inferencer while it renders pixel-ordered rays: consecutive rays are neighbouring pixels, so the field
        # kernels walk the samples ray-block-wise (NerfPointSrc.ray_block: same results, far fewer distinct cache lines)
        self.coherent_rays = False

    def _ray_block(self, n_rays):
        if not self.coherent_rays or n_rays % 32 or torch.is_grad_enabled():
            return {}
        if not isinstance(self.model, NeRFNetwork) or self.model._sharded():
            return {}
        return {'ray_block': RAY_BLOCK}

    # ---- host-side constant tables (ray independent; built with CPU torch so they are bit-equal
    #      to what the reference's linspace / pow produce) -------------------------------------
    def _coarse_table(self, device):
        key = ('coarse', str(device))
        if key not in self._tables:
            parts = []
            for i, s in enumerate(self.steps_firstpass):
                lo, hi = m.log10(self.z_bounds[i]), m.log10(self.z_bounds[i + 1])
                parts.append(torch.linspace(lo, hi - (hi - lo) / s, s))
            zlog = torch.cat(parts)
            self._tables[key] = (zlog.to(device), torch.pow(10, zlog).to(device))
        return self._tables[key]

    def _lin_table(self, n_importance, device):
        key = ('lin', n_importance, str(device))
        if key not in self._tables:
            self._tables[key] = torch.linspace(0, 1 - 1 / n_importance, n_importance).to(device)
        return self._tables[key]

    def _steps_array(self, steps):
        return (C.c_int32 * len(steps))(*[int(s) for s in steps])

    ## Rendering Pipeline

    def prefetch_batch(self, n, h, w, K, E):
        """Prepare the NEXT batch while the current step's optimizer pass is still to come: generate its rays and, under a
        level-sharded table, start the ray all-gather of its coarse pass (LevelSharding.prefetch_rays) -- neither depends on
        the parameters.  render() / render_loss() called with these very tensors picks both up; no counterpart in the
        reference, whose step is single-GPU (trainer.py:143-150)."""
        rays_o, rays_d = self.get_rays(h, w, K, E)
        self._prefetched_rays = (h, w, K, E, rays_o, rays_d)
        if isinstance(self.model, NeRFNetwork) and self.model._sharded():
            self.model.level_sharding.prefetch_rays(rays_o, rays_d)

    def _rays_for(self, h, w, K, E):
        pf, self._prefetched_rays = getattr(self, '_prefetched_rays', None), None
        if pf is not None and pf[0] is h and pf[1] is w and pf[2] is K and pf[3] is E:
            return pf[4], pf[5]
        return self.get_rays(h, w, K, E)

    def get_rays(self, h, w, K, E):
        """renderer.py:38-48."""
        hh, ww, KK, EE = i64(h).reshape(-1), i64(w).reshape(-1), f32(K).reshape(-1, 3, 3), f32(E).reshape(-1, 4, 4)
        N = hh.shape[0]
        rays_o = torch.empty(N, 3, device=KK.device)
        rays_d = torch.empty(N, 3, device=KK.device)
        check(lib().nerf_raygen(ptr(hh), ptr(ww), ptr(KK), ptr(EE), N, ptr(rays_o), ptr(rays_d), stream()), 'nerf_raygen')
        return rays_o, rays_d

    def _z_arg(self, z_vals):
        """(tensor to keep alive, row stride) for a [N,S] depth tensor that may be a stride-0 expand."""
        if z_vals.dim() == 2 and z_vals.shape[0] > 0 and z_vals.stride(0) == 0 and z_vals.stride(1) == 1:
            return z_vals[0].contiguous(), 0
        zc = f32(z_vals)
        return zc, zc.shape[-1]

    def get_sample_points(self, rays_o, rays_d, z_vals):
        """renderer.py:51-61."""
        N, S = z_vals.shape
        ro, rd = f32(rays_o), f32(rays_d)
        z, zs = self._z_arg(z_vals)
        xyz = torch.empty(N, S, 3, device=ro.device)
        dirs = torch.empty(N, 3, device=ro.device)
        check(lib().nerf_sample_points(ptr(ro), ptr(rd), ptr(z), zs, N, S, 0.0, 1.0, ptr(xyz), None, ptr(dirs), stream()),
              'nerf_sample_points')
        return xyz, dirs[:, None, :].expand(-1, S, -1)

    def mipnerf360_scale(self, xyzs, bound_inner, bound_outer):
        """renderer.py:64-68."""
        x = f32(xyzs)
        out = torch.empty_like(x)
        check(lib().nerf_contract(ptr(x), x.numel() // 3, bound_inner, bound_outer, ptr(out), stream()), 'nerf_contract')
        return out

    def render_rays_log(self, sigmas, rgbs, z_vals, z_vals_log):
        """renderer.py:71-85."""
        zero_bg = torch.zeros(3, device=sigmas.device)
        rgb, invdepth, weights, _ = _CompositeFn.apply(sigmas, rgbs, z_vals, z_vals_log, zero_bg, 0.0, 1.0)
        return rgb, invdepth, weights

    @torch.no_grad()
    def sample_pdf(self, z_vals, weights, N_importance, noise=None, generator=None, return_inds=False):
        """renderer.py:87-118.  ``noise`` [N, N_importance] in [0,1) replaces torch.rand of :102."""
        N, S = weights.shape
        pdf = f32(weights)
        z, zs = self._z_arg(z_vals)
        if noise is None:
            noise = torch.rand([N, N_importance], device=pdf.device, generator=generator)
        noise = f32(noise)
        fine = torch.empty(N, N_importance, device=pdf.device)
        inds = torch.empty(N, N_importance, device=pdf.device, dtype=torch.int64) if return_inds else None
        if N == 0:
            return (fine, inds) if return_inds else fine
        check(lib().nerf_sample_pdf(ptr(z), zs, ptr(pdf), ptr(noise), ptr(self._lin_table(N_importance, pdf.device)), N, S,
                                    N_importance, ptr(fine), None, ptr(inds), stream()), 'nerf_sample_pdf')
        if return_inds:
            return fine, inds
        return fine

    ## Calculate density distribiution along ray using forward only first pass to inform second pass
    @torch.no_grad()
    def get_uniform_z_vals(self, rays_o, rays_d, n_samples):
        """renderer.py:122-129 (``n_samples`` is ignored there too)."""
        zlog, z = self._coarse_table(rays_o.device)
        return zlog.expand(rays_o.shape[0], -1), z.expand(rays_o.shape[0], -1)

    @torch.no_grad()
    def coarse_weights(self, sigmas, z_vals_log):
        """renderer.py:154-160 (inline in the reference's efficient_sampling)."""
        sig = f32(sigmas)
        N, S = sig.shape
        z, zs = self._z_arg(z_vals_log)
        w = torch.empty_like(sig)
        check(lib().nerf_coarse_weights(ptr(sig), ptr(z), zs, N, S, ptr(w), stream()), 'nerf_coarse_weights')
        return w

    @torch.no_grad()
    def reweight_the_weights(self, weights, alpha, steps):
        """renderer.py:132-142."""
        w = f32(weights)
        N, S = w.shape
        pdf = torch.empty_like(w)
        check(lib().nerf_reweight(ptr(w), N, S, alpha, self._steps_array(steps), len(steps), ptr(pdf), stream()),
              'nerf_reweight')
        return pdf

    @torch.no_grad()
    def efficient_sampling(self, rays_o, rays_d, n_samples, alpha_const, noise=None, generator=None):
        """renderer.py:145-167."""
        ro, rd = f32(rays_o), f32(rays_d)
        N = ro.shape[0]
        zlog_c, z_c = self._coarse_table(ro.device)
        Sc = zlog_c.shape[0]
        if isinstance(self.model, NeRFNetwork) and PIPELINED_DEPTH_GATHER and self.model._sharded():
            piped = self._efficient_sampling_by_groups(ro, rd, zlog_c, z_c, n_samples, alpha_const, noise, generator)
            if piped is not None:
                return piped
        if isinstance(self.model, NeRFNetwork):
            sigmas, _ = self.model.density_kw(dict(outer_bound=self.outer_bound, rays_o=ro, rays_d=rd, z=z_c, z_stride=0,
                                                   n_samples=Sc, n_points=N * Sc, inner_bound=self.inner_bound,
                                                   **self._ray_block(N)))
            sigmas = sigmas.view(N, Sc)
        else:
            z_vals_log, z_vals = self.get_uniform_z_vals(ro, rd, self.steps_firstpass)
            xyzs, _ = self.get_sample_points(ro, rd, z_vals)
            s_xyzs = self.mipnerf360_scale(xyzs, self.inner_bound, self.outer_bound)
            sigmas, _ = self.model.density(s_xyzs, self.outer_bound)
            sigmas = f32(sigmas)
        zlog_f = torch.empty(N, n_samples, device=ro.device)
        z_f = torch.empty(N, n_samples, device=ro.device)
        if noise is None:
            # renderer.py:102's torch.rand drawn inside the kernel (Philox, a pure function of seed / call counter / element)
            seed, counter = self._noise_stream(generator)
            check(lib().nerf_coarse_to_fine_rng(ptr(sigmas), ptr(zlog_c), 0, seed, counter,
                                                ptr(self._lin_table(n_samples, ro.device)), N, Sc, n_samples, alpha_const,
                                                self._steps_array(self.steps_firstpass), len(self.steps_firstpass),
                                                ptr(zlog_f), ptr(z_f), stream()), 'nerf_coarse_to_fine_rng')
            return zlog_f, z_f
        noise = f32(noise)
        check(lib().nerf_coarse_to_fine(ptr(sigmas), ptr(zlog_c), 0, ptr(noise), ptr(self._lin_table(n_samples, ro.device)),
                                        N, Sc, n_samples, alpha_const, self._steps_array(self.steps_firstpass),
                                        len(self.steps_firstpass), ptr(zlog_f), ptr(z_f), stream()), 'nerf_coarse_to_fine')
        return zlog_f, z_f

    def _efficient_sampling_by_groups(self, ro, rd, zlog_c, z_c, n_samples, alpha_const, noise, generator):
        """efficient_sampling under a level-sharded table with ray groups: group g is resampled as soon as its coarse
        densities exist and the all-gather of its rays + fine depths (which the fine pass's level owners need) starts at
        once, under the coarse pass of group g + 1 -- instead of ONE all-gather of all depths standing between the two
        passes for its whole latency (level_parallel.LevelSharding.pregather_*).  Same depths as the ungrouped form: the
        in-kernel noise is indexed by the ray's position in the batch (nerf_coarse_to_fine_rng_at).  None when the batch is
        not cut into the same groups in both passes."""
        sh = self.model.level_sharding
        N, Sc = ro.shape[0], zlog_c.shape[0]
        cuts = sh.cuts_for(N, N * n_samples)
        if len(cuts) < 2 or cuts != sh.cuts_for(N, N * Sc):
            return None
        zlog_f = torch.empty(N, n_samples, device=ro.device)
        z_f = torch.empty(N, n_samples, device=ro.device)
        lin = self._lin_table(n_samples, ro.device)
        steps, n_seg = self._steps_array(self.steps_firstpass), len(self.steps_firstpass)
        if noise is None:
            seed, counter = self._noise_stream(generator)
        else:
            noise = f32(noise)
        st = sh.pregather_start(N, cuts)

        def per_group(gi, grp, sig):
            ra, rb = grp['ra'], grp['rb']
            assert (ra, rb) == tuple(cuts[gi])
            if noise is None:
                check(lib().nerf_coarse_to_fine_rng_at(ptr(sig), ptr(zlog_c), 0, seed, counter, ra, ptr(lin), rb - ra, Sc,
                                                       n_samples, alpha_const, steps, n_seg, ptr(zlog_f[ra:rb]), ptr(z_f[ra:rb]),
                                                       stream()), 'nerf_coarse_to_fine_rng_at')
            else:
                check(lib().nerf_coarse_to_fine(ptr(sig), ptr(zlog_c), 0, ptr(noise[ra:rb]), ptr(lin), rb - ra, Sc, n_samples,
                                                alpha_const, steps, n_seg, ptr(zlog_f[ra:rb]), ptr(z_f[ra:rb]), stream()),
                      'nerf_coarse_to_fine')
            sh.pregather_group(st, gi, ro[ra:rb], rd[ra:rb], z_f[ra:rb])

        self.model.density_kw(dict(outer_bound=self.outer_bound, rays_o=ro, rays_d=rd, z=z_c, z_stride=0, n_samples=Sc,
                                   n_points=N * Sc, inner_bound=self.inner_bound, **self._ray_block(N)), per_group=per_group)
        sh.pregather_finish(st, z_f)
        return zlog_f, z_f

    def _noise_stream(self, generator=None):
        """(seed, counter) of the next in-kernel noise draw.  The counter is one draw from torch's HOST generator (no GPU
        work), so the noise follows torch's seeding like the reference's torch.rand does: torch.manual_seed(s) replays the
        same sequence of draws.  The seed is the generator's initial seed offset by the rank (data-parallel ranks draw
        different noise).  A device generator cannot be read without a sync: its initial seed + this renderer's own call
        count is used instead."""
        from . import parallel
        if generator is not None and generator.device.type != 'cpu':
            seed = int(generator.initial_seed())
            if getattr(self, '_noise_seed', None) != seed:
                self._noise_seed, self._noise_counter = seed, 0
            self._noise_counter += 1
            counter = self._noise_counter
        else:
            seed = int(generator.initial_seed()) if generator is not None else int(torch.initial_seed())
            counter = int(torch.randint(0, 2 ** 62, (1,), generator=generator))
        seed = (seed + parallel.rank() * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        self.last_noise_stream = (seed, counter)        # render(..., noise=uniform_noise(*last_noise_stream, N, S)) replays it
        return seed, counter

    def uniform_noise(self, seed, counter, n_rays, n_samples):
        """The [n_rays, n_samples] noise array the in-kernel draw (seed, counter) uses: render(..., noise=this) replays it."""
        out = torch.empty(n_rays, n_samples, device='cuda')
        check(lib().nerf_uniform_noise(seed, counter, n_rays * n_samples, ptr(out), stream()), 'nerf_uniform_noise')
        return out

    def render(self, n, h, w, K, E, bg_color, noise=None, generator=None, aux=True, train_loss=None):
        """renderer.py:170-194.  ``aux=False`` (inference only, no grad): the per-sample aux outputs ``xyzs`` and
        ``x_hashtable`` are not produced (172 B per sample that render_image never reads)."""
        rays_o, rays_d = self._rays_for(h, w, K, E)
        z_vals_log, z_vals = self.efficient_sampling(rays_o, rays_d, self.steps_importance, self.alpha_importance,
                                                     noise=noise, generator=generator)
        N, Sf = z_vals.shape
        lean = (not aux) and (not torch.is_grad_enabled()) and isinstance(self.model, NeRFNetwork)
        if lean:
            sigmas, rgbs = self.model.field_nograd(
                dict(outer_bound=self.outer_bound, rays_o=rays_o, rays_d=rays_d, z=z_vals, z_stride=Sf, n_samples=Sf,
                     n_points=N * Sf, n=i64(n).reshape(-1), inner_bound=self.inner_bound, **self._ray_block(N)))
            sigmas, rgbs, xyzs, x_hashtable = sigmas.view(N, Sf), rgbs.view(N, Sf, 3), None, None
        elif isinstance(self.model, NeRFNetwork):
            nn_ = i64(n).reshape(-1)
            sigmas, rgbs, xh, xyzs = self.model._field(
                dict(outer_bound=self.outer_bound, rays_o=rays_o, rays_d=rays_d, z=z_vals, z_stride=Sf, n_samples=Sf,
                     n_points=N * Sf, n=nn_, inner_bound=self.inner_bound), want_xyz=True)
            sigmas, rgbs, xyzs = sigmas.view(N, Sf), rgbs.view(N, Sf, 3), xyzs.view(N, Sf, 3)
            x_hashtable = xh.view(N, Sf, -1)
        else:
            xyzs, dirs = self.get_sample_points(rays_o, rays_d, z_vals)
            s_xyzs = self.mipnerf360_scale(xyzs, self.inner_bound, self.outer_bound)
            n_expand = n[:, None].expand(-1, Sf)
            sigmas, rgbs, aux_outputs_net = self.model(s_xyzs, dirs, n_expand, self.outer_bound)
            x_hashtable = aux_outputs_net['x_hashtable']
        lo = m.log10(self.z_bounds[0])
        rng = m.log10(self.z_bounds[-1]) - lo
        loss = None
        if train_loss is not None:
            rgb_gt, lam_dist, lam_depth = train_loss
            total, parts, flag, image, invdepth, weights, z_vals_log_s = _CompositeLossFn.apply(
                sigmas, rgbs, z_vals, z_vals_log, bg_color, rgb_gt, lo, rng, float(lam_dist), float(lam_depth))
            total.nonfinite_flag = flag
            loss = (total, parts)
        else:
            image, invdepth, weights, z_vals_log_s = _CompositeFn.apply(sigmas, rgbs, z_vals, z_vals_log, bg_color, lo, rng)

        # auxilary outputs that may be useful for inference of analysis, but not used in training
        aux_outputs = {}
        aux_outputs['invdepth'] = invdepth.detach()
        aux_outputs['z_vals'] = z_vals.detach()
        aux_outputs['z_vals_log'] = z_vals_log.detach()
        aux_outputs['sigmas'] = sigmas.detach()
        aux_outputs['rgbs'] = rgbs.detach()
        aux_outputs['xyzs'] = xyzs.detach() if xyzs is not None else None
        aux_outputs['x_hashtable'] = x_hashtable
        if loss is not None:
            return loss[0], loss[1], image, weights, z_vals_log_s, aux_outputs
        return image, weights, z_vals_log_s, aux_outputs

    def render_loss(self, n, h, w, K, E, bg_color, rgb_gt, lam_dist, lam_depth, noise=None, generator=None):
        """render() + helpers.fused_train_loss() (trainer.py:149-169) with compositing and the losses in one kernel and
        their backward in one more.  Returns (loss, parts[4] = loss_rgb, loss_dist, loss_depth, mse, image, weights,
        z_vals_log_s, aux); only `loss` carries a gradient (use render() + the helpers for anything else);
        ``loss.nonfinite_flag`` as in helpers.fused_train_loss."""
        return self.render(n, h, w, K, E, bg_color, noise=noise, generator=generator, train_loss=(rgb_gt, lam_dist, lam_depth))
