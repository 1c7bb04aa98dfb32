"""fp32 CPU restatement of the reference render-and-train hot path (TEST INFRASTRUCTURE).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  It is the checker and the timed CPU baseline, never the product.

Pinning status
--------------
* Renderer / losses / inference helpers (``OracleRenderer``, ``criterion_*``, ``psnr``,
  ``cumulative_weights_thresh``): PINNED.  ``tests/golden/make_golden.py`` imports the
  reference's own ``renderer.py`` / ``helpers.py`` / ``inference.py`` from
  ``/root/reference`` and dumps inputs + outputs of every method; ``tests/test_oracle_golden.py``
  asserts this restatement reproduces them (indices bit-exact).
* Field network (hash grid, SH, bias-free MLPs): PARITY UNPINNED.  The reference delegates
  that arithmetic to the third-party CUDA-only package ``tinycudann`` (un-vendored, no
  version pin, ``nets.py:5,39-106``) which cannot be imported here and for which the
  reference holds no tests or golden vectors.  The arithmetic below restates tiny-cuda-nn's
  published algorithm (multiresolution hash encoding, Mueller et al. 2022; SH degree 4;
  FullyFusedMLP without biases) in fp32 and is this repo's normative definition.
* Dense geometry extraction (``valid_positions`` / ``extract_dense_geometry``): restated from
  ``allign/extract_dense_geometry.py:28-101`` (not importable: needs open3d); pinned only
  through the pure-torch formulas it shares with this file -> "parity unpinned" for the
  wrapper, pinned arithmetic for the field call.

Every function cites the reference lines it follows.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# Multiresolution hash grid (tiny-cuda-nn "HashGrid", called from nets.py:39-50,118,154)
# --------------------------------------------------------------------------------------

PRIME_Y = 2654435761
PRIME_Z = 805459861
MASK32 = 0xFFFFFFFF


class GridLevel(object):
    __slots__ = ("scale", "res", "entries", "offset", "hashed")

    def __init__(self, scale, res, entries, offset, hashed):
        self.scale = scale      # python float (rounded to fp32)
        self.res = res          # grid resolution of this level
        self.entries = entries  # number of table entries (each holds F floats)
        self.offset = offset    # first entry of this level in the flat table
        self.hashed = hashed    # True: spatial hash, False: dense index


def grid_levels(n_levels, log2_hashmap_size, base_resolution=16, per_level_scale=1.3819):
    """Per-level geometry.  base_resolution / per_level_scale are hard-coded at nets.py:46-47."""
    levels = []
    offset = 0
    # per_level_scale is an fp32 config value (tcnn reads it as float); log2 / exp2 are evaluated in
    # double on fp32-rounded arguments and rounded once, so host libm differences cannot leak in
    log2_pls = np.float32(math.log2(float(np.float32(per_level_scale))))
    for l in range(n_levels):
        expo = float(np.float32(l) * log2_pls)
        scale = float(np.float32(float(np.exp2(np.float64(expo))) * base_resolution - 1.0))
        res = int(math.ceil(scale)) + 1
        full = res ** 3
        entries = min(((full + 7) // 8) * 8, 1 << log2_hashmap_size)
        hashed = full > entries
        levels.append(GridLevel(scale, res, entries, offset, hashed))
        offset += entries
    return levels, offset


def _level_corners(xd, lv):
    """Corner indices [M,8] (int64, into the flat entry table) and trilinear weights [M,8] (fp32) of one level."""
    pos = (xd * lv.scale + 0.5).float()          # == fmaf(x, scale, 0.5) (product exact in fp64)
    cell = torch.floor(pos)
    w = pos - cell                                # exact in fp32
    ig = cell.to(torch.int64)
    idxs, wcs = [], []
    for c in range(8):
        wc = torch.ones(xd.shape[0], dtype=torch.float32)
        corner = []
        for d in range(3):
            if (c >> d) & 1:
                wc = wc * w[:, d]
                corner.append((ig[:, d] + 1) & MASK32)
            else:
                wc = wc * (1 - w[:, d])
                corner.append(ig[:, d] & MASK32)
        if lv.hashed:
            idx = corner[0] ^ ((corner[1] * PRIME_Y) & MASK32) ^ ((corner[2] * PRIME_Z) & MASK32)
        else:
            idx = (corner[0] + ((corner[1] * lv.res) & MASK32) + ((corner[2] * ((lv.res * lv.res) & MASK32)) & MASK32)) & MASK32
        idxs.append(idx % lv.entries + lv.offset)
        wcs.append(wc)
    return torch.stack(idxs, 1), torch.stack(wcs, 1)


class _EncodeFn(torch.autograd.Function):
    """hashgrid_encode for inputs that carry no gradient (the training path: the sample points are constants,
    renderer.py:145).  Same forward arithmetic as the plain-autograd form below; the backward builds the table gradient
    ONCE (one zero fill + one index_add_ per level) -- plain autograd would materialise one dense table-sized gradient
    per corner gather, 160 x 1.6 GB per step at config_base, which is what a CPU timing would then measure."""

    @staticmethod
    def forward(ctx, x01, table, levels, n_features):
        tab = table.view(-1, n_features)
        xd = x01.double()
        outs, saved = [], []
        for lv in levels:
            idx, wc = _level_corners(xd, lv)
            vals = tab[idx.reshape(-1)].view(idx.shape[0], 8, n_features)      # ONE gather per level
            acc = torch.zeros(x01.shape[0], n_features, dtype=torch.float32)
            for c in range(8):
                acc = acc + wc[:, c, None] * vals[:, c]
            outs.append(acc)
            saved += [idx, wc]
        ctx.save_for_backward(*saved)
        ctx.shape, ctx.F = table.shape, n_features
        return torch.cat(outs, dim=-1)

    @staticmethod
    def backward(ctx, dout):
        F_ = ctx.F
        grad = torch.zeros(ctx.shape, dtype=torch.float32)
        g2 = grad.view(-1, F_)
        saved = ctx.saved_tensors
        for l in range(len(saved) // 2):
            idx, wc = saved[2 * l], saved[2 * l + 1]
            contrib = wc[:, :, None] * dout[:, None, l * F_:(l + 1) * F_]
            g2.index_add_(0, idx.reshape(-1), contrib.reshape(-1, F_))
        return None, grad, None, None


def hashgrid_encode(x01, table, levels, n_features):
    """x01 [M,3] fp32 in [0,1]; table flat fp32 [total_entries * F] -> [M, L*F].

    pos = fma(x, scale, 0.5); cell = floor(pos); trilinear blend of the 8 corners, corner c
    takes +1 in dimension d iff bit d of c is set; dense levels index x + y*res + z*res^2,
    hashed levels x ^ y*2654435761 ^ z*805459861 (uint32 wrap), both modulo the level size.
    """
    if not x01.requires_grad:
        return _EncodeFn.apply(x01, table, levels, n_features)
    # positions carry a gradient (pose alignment, allign/trainer.py:188): plain autograd through the weights
    tab = table.view(-1, n_features)
    outs = []
    xd = x01.double()
    for lv in levels:
        pos = (xd * lv.scale + 0.5).float()          # == fmaf(x, scale, 0.5) (product exact in fp64)
        cell = torch.floor(pos)
        w = pos - cell                                # exact in fp32
        ig = cell.to(torch.int64)
        acc = torch.zeros(x01.shape[0], n_features, dtype=torch.float32)
        for c in range(8):
            wc = torch.ones(x01.shape[0], dtype=torch.float32)
            corner = []
            for d in range(3):
                if (c >> d) & 1:
                    wc = wc * w[:, d]
                    corner.append((ig[:, d] + 1) & MASK32)
                else:
                    wc = wc * (1 - w[:, d])
                    corner.append(ig[:, d] & MASK32)
            if lv.hashed:
                idx = corner[0] ^ ((corner[1] * PRIME_Y) & MASK32) ^ ((corner[2] * PRIME_Z) & MASK32)
            else:
                idx = (corner[0] + ((corner[1] * lv.res) & MASK32) + ((corner[2] * ((lv.res * lv.res) & MASK32)) & MASK32)) & MASK32
            idx = idx % lv.entries + lv.offset
            acc = acc + wc[:, None] * tab[idx]
        outs.append(acc)
    return torch.cat(outs, dim=-1)


# --------------------------------------------------------------------------------------
# Spherical harmonics, degree <= 4 (tiny-cuda-nn "SphericalHarmonics", nets.py:74-81,128-129)
# --------------------------------------------------------------------------------------

def sh_encode(d01, degree=4):
    """d01 in [0,1]^3 (nets.py:128 maps unit dirs there); returns degree^2 real SH values."""
    x = d01[:, 0] * 2 - 1
    y = d01[:, 1] * 2 - 1
    z = d01[:, 2] * 2 - 1
    xy, xz, yz = x * y, x * z, y * z
    x2, y2, z2 = x * x, y * y, z * z
    out = [torch.full_like(x, 0.28209479177387814)]
    if degree > 1:
        out += [-0.48860251190291987 * y, 0.48860251190291987 * z, -0.48860251190291987 * x]
    if degree > 2:
        out += [1.0925484305920792 * xy, -1.0925484305920792 * yz,
                0.94617469575755997 * z2 - 0.31539156525251999,
                -1.0925484305920792 * xz, 0.54627421529603959 * x2 - 0.54627421529603959 * y2]
    if degree > 3:
        out += [0.59004358992664352 * y * (-3.0 * x2 + y2), 2.8906114426405538 * xy * z,
                0.45704579946446572 * y * (1.0 - 5.0 * z2), 0.3731763325901154 * z * (5.0 * z2 - 3.0),
                0.45704579946446572 * x * (1.0 - 5.0 * z2), 1.4453057213202769 * z * (x2 - y2),
                0.59004358992664352 * x * (-x2 + 3.0 * y2)]
    return torch.stack(out, dim=-1)


# --------------------------------------------------------------------------------------
# Bias-free MLP (tiny-cuda-nn "FullyFusedMLP", nets.py:53-63,96-106), fp32 here
# --------------------------------------------------------------------------------------

def mlp_dims(n_in, n_out, hidden, n_layers):
    """n_layers matmuls: in->hidden, (n_layers-2) x hidden->hidden, hidden->out."""
    dims = [n_in] + [hidden] * (n_layers - 1) + [n_out]
    return [(dims[i + 1], dims[i]) for i in range(n_layers)]


def mlp_forward(x, params, dims):
    """params: flat fp32, layer after layer, each [out,in] row-major.  ReLU hidden, linear out."""
    off = 0
    h = x
    for i, (o, n) in enumerate(dims):
        W = params[off:off + o * n].view(o, n)
        off += o * n
        h = h @ W.t()
        if i + 1 < len(dims):
            h = torch.relu(h)
    return h


def xavier_uniform_(params, dims, generator):
    off = 0
    with torch.no_grad():
        for (o, n) in dims:
            bound = math.sqrt(6.0 / (o + n))
            params[off:off + o * n] = (torch.rand(o * n, generator=generator) * 2 - 1) * bound
            off += o * n


class OracleNetwork(torch.nn.Module):
    """Same surface as nets.py:12-164 (NeRFNetwork), fp32 CPU, autograd-capable."""

    def __init__(self, N, encoding_precision='float32', encoding_n_levels=20, encoding_n_features_per_level=2,
                 encoding_log2_hashmap_size=24, geo_feat_dim=15, sigma_hidden_dim=64, sigma_num_layers=2,
                 encoding_dir_precision='float32', encoding_dir_encoding='SphericalHarmonics', encoding_dir_degree=4,
                 latent_embedding_dim=48, color_hidden_dim=64, color_num_layers=3, seed=0):
        super().__init__()
        self.N = N
        # precision float16 (nets.py:31-36,49): tinycudann evaluates the encoding from the fp16 image of its fp32 master
        # parameters; restated here as "the table is rounded to fp16 (nearest even) where it is gathered", gradients passing
        # straight through to the master; the blend and everything after it stay fp32 (tinycudann's own fp16 blend
        # arithmetic is absent from /root/reference and therefore unpinned)
        self.half_table = encoding_precision == 'float16'
        self.n_levels = encoding_n_levels
        self.n_features = encoding_n_features_per_level
        self.levels, total = grid_levels(encoding_n_levels, encoding_log2_hashmap_size)
        self.geo_feat_dim = geo_feat_dim
        self.sh_degree = encoding_dir_degree
        self.directional_dim = encoding_dir_degree ** 2
        self.latent_emb_dim = latent_embedding_dim
        g = torch.Generator().manual_seed(seed)
        self.encoder_params = torch.nn.Parameter((torch.rand(total * self.n_features, generator=g) * 2 - 1) * 1e-4)
        self.sigma_dims = mlp_dims(self.n_levels * self.n_features, 1 + geo_feat_dim, sigma_hidden_dim, sigma_num_layers)
        self.sigma_params = torch.nn.Parameter(torch.zeros(sum(o * n for o, n in self.sigma_dims)))
        xavier_uniform_(self.sigma_params, self.sigma_dims, g)
        self.in_dim_color = geo_feat_dim + self.directional_dim + self.latent_emb_dim
        self.color_dims = mlp_dims(self.in_dim_color, 3, color_hidden_dim, color_num_layers)
        self.color_params = torch.nn.Parameter(torch.zeros(sum(o * n for o, n in self.color_dims)))
        xavier_uniform_(self.color_params, self.color_dims, g)
        self.latent_emb = torch.nn.Parameter(torch.zeros(N, latent_embedding_dim))  # nets.py:90 zero init

    def encode(self, x01):
        table = self.encoder_params
        if self.half_table:
            table = table + (table.detach().half().float() - table.detach())
        return hashgrid_encode(x01, table, self.levels, self.n_features)

    def forward(self, x, d, n, outer_bound):
        """nets.py:109-146."""
        prefix = x.shape[:-1]
        x = x.reshape(-1, 3)
        d = d.reshape(-1, 3)
        n = n.reshape(-1)
        x = (x + outer_bound) / (2 * outer_bound)                      # nets.py:117
        x_hashtable = self.encode(x)                                   # :118
        out = mlp_forward(x_hashtable, self.sigma_params, self.sigma_dims)  # :120
        sigma = F.relu(out[..., 0])                                    # :122
        geo = out[..., 1:]                                             # :123
        cin = geo
        if self.directional_dim != 0:
            cin = torch.cat([cin, sh_encode((d + 1) / 2, self.sh_degree)], dim=-1)   # :128-130
        if self.latent_emb_dim != 0:
            cin = torch.cat([cin, self.latent_emb[n]], dim=1)          # :132-133
        color = torch.sigmoid(mlp_forward(cin, self.color_params, self.color_dims))  # :135-138
        return sigma.reshape(*prefix), color.reshape(*prefix, -1), {'x_hashtable': x_hashtable.reshape(*prefix, -1).detach()}

    def density(self, x, outer_bound):
        """nets.py:148-164."""
        prefix = x.shape[:-1]
        x = x.reshape(-1, 3)
        x = (x + outer_bound) / (2 * outer_bound)
        x_hashtable = self.encode(x)
        out = mlp_forward(x_hashtable, self.sigma_params, self.sigma_dims)
        sigma = F.relu(out[..., 0])
        return sigma.reshape(*prefix), {'x_hashtable': x_hashtable.reshape(*prefix, -1).detach()}


class AnalyticField(object):
    """Closed-form stand-in for the field network (sigma, rgb analytic in x, d, n).

    Used to drive the reference renderer when capturing golden vectors (the real network
    needs tinycudann) and to drive both sides of renderer-only parity tests.
    """

    def __init__(self, centers=((0.3, 0.1, -0.2), (-0.4, -0.3, 0.3), (0.0, 0.5, 0.1)),
                 widths=(0.35, 0.25, 0.45), peaks=(60.0, 120.0, 25.0), floor=0.02):
        self.centers = torch.tensor(centers, dtype=torch.float32)
        self.widths = torch.tensor(widths, dtype=torch.float32)
        self.peaks = torch.tensor(peaks, dtype=torch.float32)
        self.floor = floor

    def _sigma(self, x):
        d2 = ((x[:, None, :] - self.centers[None].to(x.device)) ** 2).sum(-1)
        return (self.peaks.to(x.device) * torch.exp(-d2 / (2 * self.widths.to(x.device) ** 2))).sum(-1) + self.floor

    def density(self, x, outer_bound):
        prefix = x.shape[:-1]
        s = self._sigma(x.reshape(-1, 3))
        return s.reshape(*prefix), {'x_hashtable': torch.zeros(*prefix, 1)}

    def __call__(self, x, d, n, outer_bound):
        prefix = x.shape[:-1]
        xf = x.reshape(-1, 3)
        df = d.reshape(-1, 3)
        nf = n.reshape(-1).to(torch.float32)
        s = self._sigma(xf)
        rgb = torch.sigmoid(torch.stack([
            1.5 * xf[:, 0] + 0.7 * df[:, 1] + 0.01 * nf,
            -1.2 * xf[:, 1] + 0.9 * df[:, 2],
            0.8 * xf[:, 2] * xf[:, 0] - 0.5 * df[:, 0]], dim=-1))
        return s.reshape(*prefix), rgb.reshape(*prefix, 3), {'x_hashtable': torch.zeros(*prefix, 1)}


# --------------------------------------------------------------------------------------
# Renderer (renderer.py:10-194)
# --------------------------------------------------------------------------------------

class OracleRenderer(object):
    def __init__(self, model, inner_bound, outer_bound, z_bounds, steps_firstpass, steps_importance, alpha_importance):
        self.model = model
        self.inner_bound = inner_bound
        self.outer_bound = outer_bound
        self.z_bounds = list(z_bounds)
        self.steps_firstpass = list(steps_firstpass)
        self.steps_importance = steps_importance
        self.alpha_importance = alpha_importance

    # renderer.py:38-48
    def get_rays(self, h, w, K, E):
        pix = torch.stack([w + 0.5, h + 0.5, torch.ones_like(w)], -1)
        cam = torch.inverse(K) @ pix[:, :, None]
        rays_d = (E[:, :3, :3] @ cam).squeeze(2)
        rays_o = E[:, :3, -1].expand(rays_d.shape[0], -1)
        return rays_o, rays_d

    # renderer.py:51-61
    def get_sample_points(self, rays_o, rays_d, z_vals):
        S = z_vals.shape[1]
        pts = rays_o[:, None, :].expand(-1, S, -1) + rays_d[:, None, :].expand(-1, S, -1) * z_vals[..., None]
        dirs = rays_d / torch.linalg.norm(rays_d, dim=-1, keepdim=True)
        return pts, dirs[:, None, :].expand(-1, S, -1)

    # renderer.py:64-68 (discontinuous at |x| == inner when inner != 1; kept bug-for-bug)
    def mipnerf360_scale(self, xyzs, bound_inner, bound_outer):
        d = torch.linalg.norm(xyzs, dim=-1)[..., None].expand(-1, -1, 3)
        far = d > bound_inner
        d_safe = torch.where(far, d, torch.ones_like(d))
        return torch.where(far, xyzs * ((bound_outer - (bound_outer - bound_inner) / d_safe) / d_safe), xyzs)

    # renderer.py:71-85
    def render_rays_log(self, sigmas, rgbs, z_vals, z_vals_log):
        delta = torch.zeros_like(z_vals_log)
        delta[:, :-1] = z_vals_log[:, 1:] - z_vals_log[:, :-1]
        alpha = 1 - torch.exp(-sigmas * delta)
        shifted = torch.cat([torch.zeros_like(alpha[:, :1]), alpha], dim=-1)[:, :-1]
        weights = alpha * torch.cumprod(1 - shifted, dim=-1)
        rgb = torch.sum(weights[..., None] * rgbs, -2)
        invdepth = torch.sum(weights / z_vals, -1)
        return rgb, invdepth, weights

    # renderer.py:87-118 ; `noise` replaces the torch.rand draw of :102 when given
    @torch.no_grad()
    def sample_pdf(self, z_vals, weights, N_importance, noise=None, return_all=False):
        N_rays = z_vals.shape[0]
        cdf = torch.cumsum(weights, dim=-1)
        mid = (z_vals[:, :-1] + z_vals[:, 1:]) / 2
        u = torch.linspace(0, 1 - 1 / N_importance, N_importance)[None, :].expand(N_rays, -1)
        if noise is None:
            noise = torch.rand([N_rays, N_importance])
        u = u + noise / N_importance
        u = u * (cdf[:, -2, None] - cdf[:, 1, None])
        u = u + cdf[:, 1, None]
        inds = torch.searchsorted(cdf[:, 1:-2].contiguous(), u.contiguous(), right=True) + 1
        below = torch.gather(cdf, 1, inds - 1)
        above = torch.gather(cdf, 1, inds)
        t = (u - below) / (above - below)
        mid_below = torch.gather(mid, 1, inds - 1)
        mid_above = torch.gather(mid, 1, inds)
        fine = mid_below + (mid_above - mid_below) * t
        if return_all:
            return fine, {'cdf': cdf, 'u': u, 'inds': inds}
        return fine

    # renderer.py:122-129
    @torch.no_grad()
    def get_uniform_z_vals(self, n_rays):
        parts = []
        for i, s in enumerate(self.steps_firstpass):
            lo, hi = math.log10(self.z_bounds[i]), math.log10(self.z_bounds[i + 1])
            parts.append(torch.linspace(lo, hi - (hi - lo) / s, s))
        z_vals_log = torch.cat(parts).expand(n_rays, -1)
        return z_vals_log, torch.pow(10, z_vals_log)

    # renderer.py:132-142
    @torch.no_grad()
    def reweight_the_weights(self, weights, alpha, steps):
        zero = torch.zeros_like(weights[:, :1])
        w = torch.cat([zero, weights, zero], dim=-1)
        w = 1 / 2 * (torch.maximum(w[..., :-2], w[..., 1:-1]) + torch.maximum(w[..., 1:-1], w[..., 2:]))
        w = w + alpha / w.shape[-1]
        c = 0
        for s in steps:
            w[:, c:c + s] *= s / sum(steps)
            c += s
        return w / torch.sum(w, dim=-1, keepdim=True)

    # renderer.py:145-167
    @torch.no_grad()
    def efficient_sampling(self, rays_o, rays_d, n_samples, alpha_const, noise=None, return_all=False):
        z_vals_log, z_vals = self.get_uniform_z_vals(rays_o.shape[0])
        xyzs, _ = self.get_sample_points(rays_o, rays_d, z_vals)
        s_xyzs = self.mipnerf360_scale(xyzs, self.inner_bound, self.outer_bound)
        sigmas, _ = self.model.density(s_xyzs, self.outer_bound)
        delta = torch.zeros_like(sigmas)
        delta[:, :-1] = z_vals_log[:, 1:] - z_vals_log[:, :-1]
        alpha = 1 - torch.exp(-sigmas * delta)
        shifted = torch.cat([torch.zeros_like(alpha[:, :1]), alpha], dim=-1)[:, :-1]
        weights = alpha * torch.cumprod(1 - shifted, dim=-1)
        weights_re = self.reweight_the_weights(weights, alpha_const, self.steps_firstpass)
        if return_all:
            fine, dbg = self.sample_pdf(z_vals_log, weights_re, n_samples, noise, return_all=True)
            dbg.update({'z_vals_log_coarse': z_vals_log, 'z_vals_coarse': z_vals, 'xyzs_coarse': xyzs,
                        's_xyzs_coarse': s_xyzs, 'sigmas_coarse': sigmas, 'weights_coarse': weights,
                        'weights_re': weights_re})
            return fine, torch.pow(10, fine), dbg
        fine = self.sample_pdf(z_vals_log, weights_re, n_samples, noise)
        return fine, torch.pow(10, fine)

    # renderer.py:170-194
    def render(self, n, h, w, K, E, bg_color, noise=None):
        rays_o, rays_d = self.get_rays(h, w, K, E)
        z_vals_log, z_vals = self.efficient_sampling(rays_o, rays_d, self.steps_importance, self.alpha_importance, noise)
        xyzs, dirs = self.get_sample_points(rays_o, rays_d, z_vals)
        s_xyzs = self.mipnerf360_scale(xyzs, self.inner_bound, self.outer_bound)
        n_expand = n[:, None].expand(-1, z_vals.shape[-1])
        sigmas, rgbs, aux_net = self.model(s_xyzs, dirs, n_expand, self.outer_bound)
        image, invdepth, weights = self.render_rays_log(sigmas, rgbs, z_vals, z_vals_log)
        image = image + (1 - torch.sum(weights, dim=-1)[..., None]) * bg_color
        lo, hi = math.log10(self.z_bounds[0]), math.log10(self.z_bounds[-1])
        z_vals_log_s = (z_vals_log - lo) / (hi - lo)
        aux = {'invdepth': invdepth.detach(), 'z_vals': z_vals.detach(), 'z_vals_log': z_vals_log.detach(),
               'sigmas': sigmas.detach(), 'rgbs': rgbs.detach(), 'xyzs': xyzs.detach(),
               'x_hashtable': aux_net['x_hashtable']}
        return image, weights, z_vals_log_s, aux


# --------------------------------------------------------------------------------------
# Losses / metrics (helpers.py:11-32, trainer.py:153-169)
# --------------------------------------------------------------------------------------

def psnr(rgb_, rgb):
    return -10.0 * math.log10(F.mse_loss(rgb_, rgb))            # helpers.py:11-13


def criterion_rgb(rgb_, rgb):
    return F.huber_loss(rgb_, rgb, delta=0.1)                    # helpers.py:29-32


def criterion_dist(weights, z_vals):
    """helpers.py:21-27 -- the O(S^2) pairwise form, kept literally as the spec."""
    w = torch.bmm(weights[:, :, None], weights[:, None, :])
    s = torch.abs(z_vals[:, :, None] - z_vals[:, None, :])
    return torch.mean(torch.sum(w * s, dim=[1, 2]))


def dist_scalar(it, num_iters, lambda1, lambda2):
    """trainer.py:160."""
    return 10 ** (it / num_iters * (math.log10(lambda2) - math.log10(lambda1)) + math.log10(lambda1))


def depth_scalar(it, num_iters, lambda1, lambda2):
    """trainer.py:167 (bug-for-bug: the schedule subtracts lambda2 from lambda2 -> constant lambda2)."""
    return 10 ** (it / num_iters * (math.log10(lambda2) - math.log10(lambda2)) + math.log10(lambda2))


def train_loss(rgb, rgb_gt, weights, z_vals_log_s, it, num_iters, dist_l1, dist_l2, depth_l1, depth_l2):
    """trainer.py:153-169.  Returns (loss, dict of parts)."""
    loss_rgb = criterion_rgb(rgb, rgb_gt)
    loss = loss_rgb
    parts = {'loss_rgb': loss_rgb, 'loss_dist': 0, 'dist_scalar': 0, 'loss_depth': 0, 'depth_scalar': 0}
    if not (dist_l1 == 0 or dist_l2 == 0):
        parts['loss_dist'] = criterion_dist(weights, z_vals_log_s)
        parts['dist_scalar'] = dist_scalar(it, num_iters, dist_l1, dist_l2)
        loss = loss + parts['dist_scalar'] * parts['loss_dist']
    if not (depth_l1 == 0 or depth_l2 == 0):
        parts['depth_scalar'] = depth_scalar(it, num_iters, depth_l1, depth_l2)
        parts['loss_depth'] = torch.mean(weights * (1 - z_vals_log_s))
        loss = loss + parts['depth_scalar'] * parts['loss_depth']
    return loss, parts


def make_optimizer(model, lr_enc=2e-2, lr_latent=2e-2, lr_net=1e-3, wd_net=1e-6, betas=(0.9, 0.99), eps=1e-15):
    """train.py:84-88 applied to OracleNetwork's flat parameter tensors."""
    return torch.optim.Adam([
        {'name': 'encoding', 'params': [model.encoder_params], 'lr': lr_enc},
        {'name': 'latent_emb', 'params': [model.latent_emb], 'lr': lr_latent},
        {'name': 'net', 'params': [model.sigma_params, model.color_params], 'weight_decay': wd_net, 'lr': lr_net},
    ], betas=betas, eps=eps)


# --------------------------------------------------------------------------------------
# Inference helpers (inference.py:97-157)
# --------------------------------------------------------------------------------------

@torch.no_grad()
def cumulative_weights_thresh(weights, thresh):
    """inference.py:97-105 -- one-hot of the first sample whose running weight sum reaches thresh."""
    cdf = torch.cumsum(torch.clone(weights), dim=-1)
    hit = (cdf >= thresh).to(weights.dtype)
    before = 1 - torch.cumsum(hit[..., :-1], dim=-1)
    before = torch.clamp(before, min=0)
    out = hit.clone()
    out[..., 1:] = hit[..., 1:] * before
    return out


@torch.no_grad()
def invdepth_thresh(weights, z_vals, distribution_area, max_variance_pcd):
    """inference.py:134-148."""
    start = cumulative_weights_thresh(weights, 0.5 - distribution_area / 2)
    mid = cumulative_weights_thresh(weights, 0.5)
    end = cumulative_weights_thresh(weights, 0.5 + distribution_area / 2)
    d_start = torch.sum(start * z_vals, dim=-1)
    d_mid = torch.sum(mid * z_vals, dim=-1)
    d_end = torch.sum(end * z_vals, dim=-1)
    out = 1 / d_mid
    out[torch.abs(d_start - d_end) > max_variance_pcd] = 0
    out[d_mid == 0] = 0
    return out


@torch.no_grad()
def surface_geometry(weights, z_vals, xyzs, rgbs, distribution_area, max_variance_npy, inner_bound):
    """inference.py:187-214 for one chunk: the sample at which the running weight sum crosses 0.5, kept when the
    depths of the (0.5 -/+ area/2) crossings lie within max_variance_npy of each other and the point is inside the
    inner bound.  Returns (points [P,3], colors [P,3], depth_variance [P,1])."""
    start = cumulative_weights_thresh(weights, 0.5 - distribution_area / 2)
    mid = cumulative_weights_thresh(weights, 0.5)
    end = cumulative_weights_thresh(weights, 0.5 + distribution_area / 2)
    d_start = torch.sum(start * z_vals, dim=-1)
    d_end = torch.sum(end * z_vals, dim=-1)
    var = torch.abs(d_start - d_end)[:, None, None].expand(-1, weights.shape[1], 1)
    mask = (var < max_variance_npy) & mid.to(bool)[..., None] & \
           (torch.linalg.norm(xyzs, dim=-1, keepdim=True) < inner_bound)
    points = xyzs[mask.expand(*xyzs.shape)].reshape(-1, 3)
    colors = rgbs[mask.expand(*rgbs.shape)].reshape(-1, 3)
    return points, colors, var[mask].reshape(-1, 1)


@torch.no_grad()
def threshold_margin(weights, distribution_area):
    """Per ray: the smallest distance of the running weight sum (inference.py:99) from any of the three thresholds --
    rays with a margin well above the GPU/CPU difference of the weights have the same first-crossing samples."""
    cdf = torch.cumsum(weights, dim=-1)
    m = None
    for t in (0.5 - distribution_area / 2, 0.5, 0.5 + distribution_area / 2):
        d = (cdf - t).abs().min(dim=-1).values
        m = d if m is None else torch.minimum(m, d)
    return m


# --------------------------------------------------------------------------------------
# Dense geometry extraction (allign/extract_dense_geometry.py:28-101)
# --------------------------------------------------------------------------------------

@torch.no_grad()
def valid_positions(N, H, W, K, E, res):
    """allign/extract_dense_geometry.py:28-55: inside unit sphere AND seen by any camera."""
    mask_full = torch.zeros((res, res, res), dtype=torch.bool)
    d = torch.linspace(-1, 1, res)
    for i in range(res):
        D = torch.stack(torch.meshgrid(d[i:i + 1], d, d, indexing='ij'), dim=-1)[0]     # [res,res,3]
        inside = torch.linalg.norm(D, dim=-1) < 1
        rel = D[None] - E[:, None, None, :3, -1]                                         # [N,res,res,3]
        cam = torch.inverse(E[:, None, None, :3, :3]) @ rel[..., None]
        pix_ = K[:, None, None, ...] @ cam
        pix = pix_ / pix_[:, :, :, 2, None, :]
        seen = ((pix[..., 0, 0] > 0) & (pix[..., 0, 0] < H) & (pix[..., 1, 0] > 0) & (pix[..., 1, 0] < W)
                & (pix_[..., 2, 0] > 0))
        mask_full[i] = inside & seen.any(dim=0)
    return mask_full


@torch.no_grad()
def extract_dense_geometry(model, mask, voxel_res, batch_size, thresh, outer_bound, start=0, stop=None):
    """allign/extract_dense_geometry.py:58-101 over voxel indices [start, stop)."""
    voxels = torch.linspace(-1 + 1 / voxel_res, 1 - 1 / voxel_res, voxel_res)
    stop = voxel_res ** 3 if stop is None else stop
    points, colors = [], []
    for a in range(start, stop, batch_size):
        b = min(stop, a + batch_size)
        n = torch.arange(a, b)
        x = voxels[torch.div(n, voxel_res ** 2, rounding_mode='floor')]
        y = voxels[torch.div(n, voxel_res, rounding_mode='floor') % voxel_res]
        z = voxels[n % voxel_res]
        xyz = torch.stack((x, y, z), dim=-1)
        keep = mask[((x + 1) / 2 * mask.shape[0]).to(int), ((y + 1) / 2 * mask.shape[1]).to(int),
                    ((z + 1) / 2 * mask.shape[2]).to(int)]
        xyz = xyz[keep].view(-1, 3)
        if xyz.shape[0] > 0:
            dirs = torch.tensor([0.0, 0.0, 1.0])[None].expand(xyz.shape[0], 3)
            sigmas, rgbs, _ = model(xyz, dirs, torch.zeros(xyz.shape[0], dtype=torch.int64), outer_bound)
            sel = sigmas > thresh
            points.append(xyz[sel])
            colors.append(rgbs[sel])
    if not points:
        return torch.zeros(0, 3), torch.zeros(0, 3)
    return torch.cat(points), torch.cat(colors)


# --------------------------------------------------------------------------------------
# Pose-alignment consistency step (allign/trainer.py:174-208, allign/allign2.py:35-56, allign/rotation.py:8-27)
# --------------------------------------------------------------------------------------

def so3_exp(r):
    """allign/rotation.py:18-27 with vec2skew of :8-15."""
    z = torch.zeros(1)
    skew = torch.stack([torch.cat([z, -r[2:3], r[1:2]]), torch.cat([r[2:3], z, -r[0:1]]), torch.cat([-r[1:2], r[0:1], z])])
    n = r.norm() + 1e-15
    return torch.eye(3) + (torch.sin(n) / n) * skew + ((1 - torch.cos(n)) / n ** 2) * (skew @ skew)


def rigid_transform(xyzs_a, init_transform, R, T):
    """allign/allign2.py:44-56 (homogeneous products written out)."""
    tr = torch.eye(4)
    tr[:3, :3] = so3_exp(R)
    tr[:3, 3] = T
    b = xyzs_a.reshape(-1, 3)
    b1 = torch.cat([b, b.new_ones((b.shape[0], 1))], dim=1)
    b3 = (tr @ (init_transform @ b1.T)).T
    return b3[:, :3].reshape(*xyzs_a.shape)


def scene_consistency(renderer_a, model_a, model_b, init_transform, R, T, h, w, K, E, noise=None):
    """allign/trainer.py:174-208, line by line (the [N,S,S] outer product included)."""
    rays_o, rays_d = renderer_a.get_rays(h, w, K, E)
    z_vals_log, z_vals = renderer_a.efficient_sampling(rays_o, rays_d, renderer_a.steps_importance,
                                                       renderer_a.alpha_importance, noise=noise)
    xyzs_a, _ = renderer_a.get_sample_points(rays_o, rays_d, z_vals)
    s_xyzs_a = renderer_a.mipnerf360_scale(xyzs_a, renderer_a.inner_bound, renderer_a.outer_bound)
    sigmas_a, _ = model_a.density(s_xyzs_a, renderer_a.outer_bound)
    delta = z_vals_log.new_zeros(z_vals_log.shape)
    delta[:, :-1] = (z_vals_log[:, 1:] - z_vals_log[:, :-1])
    alpha_a = 1 - torch.exp(-sigmas_a * delta)
    alpha_a_shift = torch.cat([alpha_a.new_zeros((alpha_a.shape[0], 1)), alpha_a], dim=-1)[:, :-1]
    weights_a = alpha_a * torch.cumprod(1 - alpha_a_shift, dim=-1)
    xyzs_b = rigid_transform(xyzs_a, init_transform, R, T)
    s_xyzs_b = renderer_a.mipnerf360_scale(xyzs_b, renderer_a.inner_bound, renderer_a.outer_bound)
    sigmas_b, _ = model_b.density(s_xyzs_b, renderer_a.outer_bound)
    sigmas_ab = sigmas_a + sigmas_b
    alpha_ab = 1 - torch.exp(-sigmas_ab * delta)
    alpha_ab_shift = torch.cat([alpha_ab.new_zeros((alpha_ab.shape[0], 1)), alpha_ab], dim=-1)[:, :-1]
    weights_ab = alpha_ab * torch.cumprod(1 - alpha_ab_shift, dim=-1)
    weights_a = torch.cat([weights_a[:, :-1], (1 - torch.sum(weights_a[:, :-1], dim=-1))[:, None]], -1)
    weights_ab = torch.cat([weights_ab[:, :-1], (1 - torch.sum(weights_ab[:, :-1], dim=-1))[:, None]], -1)
    wgt = torch.bmm(weights_a[:, :, None], weights_ab[:, None, :])
    s = torch.abs(z_vals_log[:, :, None] - z_vals_log[:, None, :])
    return torch.mean(torch.sum(wgt * s, dim=[1, 2]))
