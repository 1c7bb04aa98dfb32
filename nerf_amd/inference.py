"""Inferencer with the reference's surface (inference.py:27-223).  ``render_image`` keeps the
chunk loop over ``n_rays`` but leaves every chunk on the GPU and copies the finished image to
the host ONCE (the reference does a blocking .cpu() per chunk, inference.py:84-85).  Under
torch.distributed each rank renders a contiguous slice of the flattened ray list and the slices
are all_gathered once."""
import os

import numpy as np
import torch

from . import parallel
from ._lib import check, f32, lib, ptr, stream

COHERENT_RAYS = os.environ.get('NERF_COHERENT_RAYS', '1') != '0'
PIXEL_TILES = os.environ.get('NERF_PIXEL_TILES', '1') != '0'
SUPER_TILES = os.environ.get('NERF_PIXEL_SUPERTILES', '1') != '0'


def _whole_table(fn):
    """Run a pixel-sharded inference method on the whole local table (a level-sharded model first
    broadcasts the levels other ranks own; see NeRFNetwork.local_table)."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *args, **kwargs):
        model = getattr(self.renderer, 'model', None)
        # the rays of these methods are the pixels of an image in row order: tell the renderer (NerfRenderer.coherent_rays)
        before = getattr(self.renderer, 'coherent_rays', None)
        if before is not None and COHERENT_RAYS:
            self.renderer.coherent_rays = True
        try:
            if hasattr(model, 'local_table'):
                with model.local_table():
                    return fn(self, *args, **kwargs)
            return fn(self, *args, **kwargs)
        finally:
            if before is not None:
                self.renderer.coherent_rays = before
    return wrapped


class Inferencer(object):
    def __init__(self, renderer, n_rays, image_num, rotate, max_variance_npy, max_variance_pcd, distribution_area, cams,
                 freq):
        self.renderer = renderer
        self.n_rays = n_rays
        self.image_num = image_num
        self.rotate = rotate
        self.max_variance_npy = max_variance_npy
        self.max_variance_pcd = max_variance_pcd
        self.distribution_area = distribution_area
        self.cams = cams
        self.freq = freq

    def _flat(self, n, h, w, K, E):
        n_f, h_f, w_f = torch.reshape(n, (-1,)), torch.reshape(h, (-1,)), torch.reshape(w, (-1,))
        return n_f, h_f, w_f, torch.reshape(K, (-1, 3, 3)), torch.reshape(E, (-1, 4, 4))

    def _pixel_tiles(self, shape, device):
        """Permutation of the flattened pixels of an [..., H, W] batch that makes every 32 consecutive rays a 4 x 8 pixel
        tile (tiles in row order), or None: the field kernels take 32 consecutive rays per wavefront under
        NerfRenderer.coherent_rays, and a compact tile shares more hash-grid cells than a 1 x 32 strip."""
        if not (COHERENT_RAYS and PIXEL_TILES) or len(shape) < 2 or shape[-2] % 4 or shape[-1] % 8:
            return None
        key = (tuple(shape), str(device))
        cache = self.__dict__.setdefault('_tile_cache', {})
        if key not in cache:
            H, W = shape[-2], shape[-1]
            lead = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
            if SUPER_TILES and H % 32 == 0 and W % 32 == 0:
                # tiles in 32 x 32 pixel squares (8 x 4 tiles): the ray blocks in flight at one time cover a compact region
                idx = torch.arange(lead * H * W, device=device).reshape(lead, H // 32, 8, 4, W // 32, 4, 8)
                cache[key] = idx.permute(0, 1, 4, 2, 5, 3, 6).reshape(-1)
            else:
                idx = torch.arange(lead * H * W, device=device).reshape(lead, H // 4, 4, W // 8, 8)
                cache[key] = idx.permute(0, 1, 3, 2, 4).reshape(-1)
        return cache[key]

    def _lean(self):
        """render() keyword for the methods that never read x_hashtable (160 B per sample): only NerfRenderer knows it."""
        return {'aux': False} if hasattr(self.renderer, 'coherent_rays') else {}

    def _chunks(self, total):
        a0, b0 = parallel.shard_range(total, parallel.rank(), parallel.world_size())
        for a in range(a0, b0, self.n_rays):
            yield a, min(b0, a + self.n_rays)

    def _white(self, device):
        return torch.ones(3, device=device)   # fixed white background (inference.py:80)

    @_whole_table
    @torch.no_grad()
    def render_image(self, n, h, w, K, E, noise=None):
        """inference.py:54-94."""
        n_f, h_f, w_f, K_f, E_f = self._flat(n, h, w, K, E)
        total = len(n_f)
        dev = K_f.device
        tiles = self._pixel_tiles(tuple(h.shape), dev)
        if tiles is not None:       # rays in pixel-tile order (undone below: the image comes back in the caller's order)
            n_f, h_f, w_f, K_f, E_f = (t.to(dev).index_select(0, tiles) for t in (n_f, h_f, w_f, K_f, E_f))
            noise = None if noise is None else noise.to(dev).index_select(0, tiles)
        a0, b0 = parallel.shard_range(total, parallel.rank(), parallel.world_size())
        image_l = torch.zeros((b0 - a0, 3), device=dev)
        invdepth_l = torch.zeros((b0 - a0,), device=dev)
        for a, b in self._chunks(total):
            nz = None if noise is None else noise[a:b]
            image_fb, _, _, aux = self.renderer.render(n_f[a:b], h_f[a:b], w_f[a:b], K_f[a:b], E_f[a:b],
                                                       bg_color=self._white(dev), noise=nz, aux=False)
            image_l[a - a0:b - a0] = image_fb
            invdepth_l[a - a0:b - a0] = aux['invdepth']
        image_f = parallel.gather_slices(image_l, total)
        invdepth_f = parallel.gather_slices(invdepth_l, total)
        if tiles is not None:
            image_f = torch.empty_like(image_f).index_copy_(0, tiles, image_f)
            invdepth_f = torch.empty_like(invdepth_f).index_copy_(0, tiles, invdepth_f)
        image = torch.reshape(image_f, (*h.shape, 3)).cpu().numpy()
        invdepth = torch.reshape(invdepth_f, h.shape).cpu().numpy()
        return image, invdepth

    @torch.no_grad()
    def calculate_cumlative_weights_thresh(self, weights, thresh):
        """inference.py:97-105: one-hot of the first sample whose running weight sum reaches thresh."""
        w = f32(weights)
        out = torch.empty_like(w)
        flat = w.reshape(-1, w.shape[-1])
        check(lib().nerf_cum_weights_thresh(ptr(flat), flat.shape[0], flat.shape[1], thresh, ptr(out), stream()),
              'nerf_cum_weights_thresh')
        return out

    def _thresh_depths(self, weights, z_vals):
        ws = self.calculate_cumlative_weights_thresh(weights, 0.5 - self.distribution_area / 2)
        wm = self.calculate_cumlative_weights_thresh(weights, 0.5)
        we = self.calculate_cumlative_weights_thresh(weights, 0.5 + self.distribution_area / 2)
        return (torch.sum(ws * z_vals, dim=-1), torch.sum(wm * z_vals, dim=-1), torch.sum(we * z_vals, dim=-1)), wm

    @_whole_table
    @torch.no_grad()
    def render_invdepth_thresh(self, n, h, w, K, E):
        """inference.py:108-157."""
        n_f, h_f, w_f, K_f, E_f = self._flat(n, h, w, K, E)
        total = len(n_f)
        dev = K_f.device
        a0, b0 = parallel.shard_range(total, parallel.rank(), parallel.world_size())
        out_l = torch.zeros((b0 - a0,), device=dev)
        tiles = self._pixel_tiles(tuple(h.shape), dev)
        if tiles is not None:
            n_f, h_f, w_f, K_f, E_f = (t.to(dev).index_select(0, tiles) for t in (n_f, h_f, w_f, K_f, E_f))
        for a, b in self._chunks(total):
            # (weights and depths only: no per-sample aux outputs, as in render_image)
            _, weights_fb, _, aux = self.renderer.render(n_f[a:b], h_f[a:b], w_f[a:b], K_f[a:b], E_f[a:b],
                                                         bg_color=self._white(dev), **self._lean())
            (d_start, d_mid, d_end), _ = self._thresh_depths(weights_fb, aux['z_vals'])
            inv = 1 / d_mid
            inv[torch.abs(d_start - d_end) > self.max_variance_pcd] = 0
            inv[d_mid == 0] = 0
            out_l[a - a0:b - a0] = inv
        out_f = parallel.gather_slices(out_l, total)
        if tiles is not None:
            out_f = torch.empty_like(out_f).index_copy_(0, tiles, out_f)
        return torch.reshape(out_f, h.shape).cpu().numpy()

    @_whole_table
    @torch.no_grad()
    def extract_surface_geometry(self, n, h, w, K, E):
        """inference.py:160-223."""
        n_f, h_f, w_f, K_f, E_f = self._flat(n, h, w, K, E)
        total = len(n_f)
        dev = K_f.device
        points, colors, variance = [], [], []
        for a, b in self._chunks(total):
            _, weights_fb, _, aux = self.renderer.render(n_f[a:b].to(dev), h_f[a:b].to(dev), w_f[a:b].to(dev), K_f[a:b],
                                                         E_f[a:b], bg_color=self._white(dev), **self._lean())
            xyzs_fb, rgbs_fb, z_vals_fb = aux['xyzs'], aux['rgbs'], aux['z_vals']
            if xyzs_fb is None:     # lean render: the sample positions (renderer.py:51-61) from the rays, 12 B per sample
                rays_o, rays_d = self.renderer.get_rays(h_f[a:b].to(dev), w_f[a:b].to(dev), K_f[a:b], E_f[a:b])
                xyzs_fb, _ = self.renderer.get_sample_points(rays_o, rays_d, z_vals_fb)
            (d_start, _, d_end), w_mid = self._thresh_depths(weights_fb, z_vals_fb)
            dvar = torch.abs(d_start - d_end)[:, None].expand(-1, weights_fb.shape[1])
            mask = (dvar < self.max_variance_npy) & w_mid.to(bool) & \
                   (torch.linalg.norm(xyzs_fb, dim=-1) < self.renderer.inner_bound)
            points.append(xyzs_fb[mask])
            colors.append(rgbs_fb[mask])
            variance.append(dvar[mask][:, None])
        cat = lambda xs, c: torch.cat(xs, dim=0) if xs else torch.zeros((0, c), device=dev)
        pts = parallel.gather_varlen(cat(points, 3))
        col = parallel.gather_varlen(cat(colors, 3))
        var = parallel.gather_varlen(cat(variance, 1))
        return {'points': pts.cpu().numpy(), 'colors': col.cpu().numpy(), 'depth_variance': var.cpu().numpy()}
