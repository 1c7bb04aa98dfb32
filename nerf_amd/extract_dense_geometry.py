"""Dense geometry extraction with the reference's surface (allign/extract_dense_geometry.py:7-101):
camera-frustum occupancy mask, then bulk field queries over voxel_res^3 voxel centres, keeping the
points with sigma > thresh.  Differences, on purpose:

* voxel coordinates are generated on the GPU (the reference builds them on the CPU and copies, :73-79), the mask
  is applied to all voxels at once, the field runs over batches of VALID voxels in 4 x 4 x 2 brick order, and the
  threshold is applied once per GROUP of batches (the reference masks, queries, thresholds and concatenates on the CPU
  per batch, :66-95): same points, same order, one host round trip per 16 batches instead of three per batch.  What is
  held at a time beyond the table: the R^3 boolean mask (twice), the indices of the valid voxels, and sigma / rgb of one
  group of batches -- ``batch_size`` bounds the transients as it does in the reference;
* under torch.distributed each rank takes a contiguous slab of the voxel index range and the
  variable-length results are all_gathered once;
* the open3d wrapper ``generate_dense_pointcloud`` is out of scope (open3d is not part of the hot path).
The field evaluation itself is ``NeRFNetwork.field_nograd`` -> nerf_field_fwd (HIP) without the feature rows."""
import os

import torch

from . import parallel

BRICK_ORDER = os.environ.get('NERF_EXTRACT_BRICKS', '1') != '0'


class ExtractDenseGeometry():
    def __init__(self, model, dataloader, mask_res, voxel_res, batch_size, thresh, outer_bound):
        self.model = model
        self.dataloader = dataloader
        self.mask_res = mask_res
        self.voxel_res = voxel_res
        self.batch_size = batch_size
        self.thresh = thresh
        self.outer_bound = outer_bound

    @torch.no_grad()
    def get_valid_positions(self, N, H, W, K, E, res):
        """allign/extract_dense_geometry.py:28-55: inside the unit sphere AND seen by at least one camera."""
        dev = K.device
        mask_full = torch.zeros((res, res, res), dtype=torch.bool, device=dev)
        d = torch.linspace(-1, 1, res, device=dev)
        Rinv = torch.inverse(E[:, :3, :3])[:, None, None]                # [N,1,1,3,3]
        Kb = K[:, None, None]
        # several x-slices per pass (the reference takes one, :36): the same element-wise arithmetic on a taller block, sized
        # to ~8 M camera-voxel pairs: the two [pairs, 3, 3] products below are the largest temporaries, 36 B per pair each
        # (0.3 GB)
        n_cams = max(1, int(K.shape[0]))
        step = max(1, min(res, (1 << 23) // max(1, n_cams * res * res)))
        for i in range(0, res, step):
            j = min(res, i + step)
            D = torch.stack(torch.meshgrid(d[i:j], d, d, indexing='ij'), dim=-1).reshape((j - i) * res, res, 3)
            inside = torch.linalg.norm(D, dim=-1) < 1
            rel = D[None] - E[:, None, None, :3, -1]
            # the reference's two batched 3x3 mat-vecs (:41-43) as broadcast multiply + sum: a million tiny matmuls per
            # slice took 46 ms in rocBLAS (5.9 s for the 128^3 mask, 23 x the 512^3 extraction that follows)
            cam = (Rinv * rel[..., None, :]).sum(-1)                     # [N,(j-i)*res,res,3]
            pix_ = (Kb * cam[..., None, :]).sum(-1)
            pix = pix_ / pix_[..., 2:3]
            seen = ((pix[..., 0] > 0) & (pix[..., 0] < H) & (pix[..., 1] > 0) & (pix[..., 1] < W) & (pix_[..., 2] > 0))
            mask_full[i:j] = (inside & seen.any(dim=0)).view(j - i, res, res)
        return mask_full

    @torch.no_grad()
    def extract_dense_geometry(self, N, H, W, K, E, mask=None):
        """allign/extract_dense_geometry.py:58-101.  Returns (points [P,3], colors [P,3]) on the CPU."""
        if hasattr(self.model, 'local_table') and not getattr(self, '_in_local', False):
            self._in_local = True       # voxels are sharded by slab: run on the whole local table
            try:
                with self.model.local_table():
                    return self.extract_dense_geometry(N, H, W, K, E, mask)
            finally:
                self._in_local = False
        if mask is None:
            mask = self.get_valid_positions(N, H, W, K, E, res=self.mask_res)
        dev = mask.device
        R = self.voxel_res
        voxels = torch.linspace(-1 + 1 / R, 1 - 1 / R, R).to(dev)     # built on the CPU like the reference, then moved
        num_samples = R ** 3
        start, stop = parallel.shard_range(num_samples, parallel.rank(), parallel.world_size())
        # The reference masks, queries and thresholds batch by batch (:66-95), three host round trips per batch.  Here the mask
        # lookup of :80-82 is made per axis (the same float arithmetic on the same R values) and expanded to all voxels at
        # once, ONE nonzero() lists the valid voxels of this rank's slab, the field runs over batches of valid voxels only,
        # and one selection per group of 16 batches applies the threshold: the same points in the same (ascending voxel) order.
        ax = [((voxels + 1) / 2 * mask.shape[k]).to(torch.int64) for k in range(3)]
        keep = mask[ax[0][:, None, None], ax[1][None, :, None], ax[2][None, None, :]]          # [R, R, R] bool
        if start > 0 or stop < num_samples:
            flat = keep.reshape(-1).clone()
            flat[:start] = False
            flat[stop:] = False
            keep = flat.view(R, R, R)
        # valid voxels in 4 x 4 x 2 BRICK order (a wavefront of the field kernel takes 32 consecutive points: a brick shares
        # far more hash-grid cells than a 32-voxel line); un-done by the final sort of the few selected points
        bricks = BRICK_ORDER and R % 4 == 0
        if bricks:
            # positions of the valid voxels in the brick-ordered array [R/4, R/4, R/2, 4, 4, 2], turned back into canonical
            # voxel indices by arithmetic (no R^3 index array is ever built)
            q = keep.view(R // 4, 4, R // 4, 4, R // 2, 2).permute(0, 2, 4, 1, 3, 5).reshape(-1).nonzero().squeeze(1)
            iz, q = q % 2, torch.div(q, 2, rounding_mode='floor')
            iy, q = q % 4, torch.div(q, 4, rounding_mode='floor')
            ix, q = q % 4, torch.div(q, 4, rounding_mode='floor')
            bz, q = q % (R // 2), torch.div(q, R // 2, rounding_mode='floor')
            by, bx = q % (R // 4), torch.div(q, R // 4, rounding_mode='floor')
            idx = ((bx * 4 + ix) * R + (by * 4 + iy)) * R + (bz * 2 + iz)
            del q, ix, iy, iz, bx, by, bz
        else:
            idx = keep.reshape(-1).nonzero().squeeze(1)
        del keep
        P = int(idx.numel())
        dir_z = torch.tensor([0.0, 0.0, 1.0], device=dev)
        coords = lambda n: torch.stack((voxels[torch.div(n, R ** 2, rounding_mode='floor')],
                                        voxels[torch.div(n, R, rounding_mode='floor') % R], voxels[n % R]), dim=-1)
        group = 16 * self.batch_size            # sigma / rgb of one group of batches are held; one selection per group
        chosen_parts, color_parts = [], []
        for g0 in range(0, P, group):
            g1 = min(P, g0 + group)
            sig = torch.empty(g1 - g0, device=dev)
            rgb = torch.empty(g1 - g0, 3, device=dev)
            for a in range(g0, g1, self.batch_size):
                b = min(g1, a + self.batch_size)
                xyz = coords(idx[a:b]).contiguous()
                dirs = dir_z[None].expand(b - a, 3).contiguous()
                n_i = torch.zeros(b - a, dtype=torch.int64, device=dev)
                # field_nograd: no autograd node and no x_hashtable rows (160 B/point the reference's forward
                # returns at :88 and extraction discards)
                sigmas, rgbs = self.model.field_nograd(dict(outer_bound=self.outer_bound, x=xyz, d=dirs, n=n_i, n_points=b - a))
                sig[a - g0:b - g0] = sigmas
                rgb[a - g0:b - g0] = rgbs
            sel = sig > self.thresh
            chosen_parts.append(idx[g0:g1][sel])
            color_parts.append(rgb[sel])
        chosen = torch.cat(chosen_parts) if chosen_parts else idx[:0]
        colors = torch.cat(color_parts) if color_parts else torch.empty(0, 3, device=dev)
        if bricks:
            chosen, order = torch.sort(chosen)
            colors = colors[order]
        pts = parallel.gather_varlen(coords(chosen).reshape(-1, 3))
        col = parallel.gather_varlen(colors)
        return pts.cpu(), col.cpu()
