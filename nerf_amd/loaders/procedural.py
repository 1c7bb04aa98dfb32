"""Procedural synthetic scene exposing the batch interface of the reference's
CameraGeometryLoader (loaders/camera_geometry_loader.py:91-152): ``get_random_batch``,
``get_custom_batch``, ``get_image_batch``, ``get_pointcloud_batch``, ``get_calibration`` and the
attributes ``images, N, H, W, depths_bool, translation_center``.

Neither the Blender dataset of loaders/synthetic.py nor cv2 exists offline, so the scene is
generated: V cameras on a sphere of radius 4.0311*0.8 (the Blender-synthetic convention after
nerf_matrix_to_ngp's 0.8 scale, loaders/synthetic.py:24-32,60), 800x800, camera_angle_x 0.6911
(loaders/synthetic.py:54-57), OpenCV axes (+z forward, as get_rays assumes).  Ground-truth RGBA
images are optional (``render_images``): they are rendered once by the caller-provided field.

Images and calibration live ON THE GPU and batches are sampled there: the reference's CPU
randint + fancy-index + H2D copy per step (camera_geometry_loader.py:112-117) would dominate a
sub-millisecond training step.
"""
import math

import torch


def sphere_cameras(n_img, seed=0, radius=4.0311 * 0.8, H=800, W=800, camera_angle_x=0.6911):
    g = torch.Generator().manual_seed(seed)
    az = torch.rand(n_img, generator=g) * 2 * math.pi
    el = torch.rand(n_img, generator=g) * (math.pi / 2 - 0.2) + 0.1          # upper hemisphere
    pos = radius * torch.stack([torch.cos(el) * torch.cos(az), torch.cos(el) * torch.sin(az), torch.sin(el)], -1)
    fwd = -pos / pos.norm(dim=-1, keepdim=True)
    up = torch.tensor([0.0, 0.0, 1.0]).expand_as(fwd)
    right = torch.cross(fwd, up, dim=-1)
    right = right / right.norm(dim=-1, keepdim=True)
    down = torch.cross(fwd, right, dim=-1)
    E = torch.zeros(n_img, 4, 4)
    E[:, :3, 0], E[:, :3, 1], E[:, :3, 2], E[:, :3, 3] = right, down, fwd, pos
    E[:, 3, 3] = 1
    focal = 0.5 * W / math.tan(0.5 * camera_angle_x)
    K = torch.zeros(n_img, 3, 3)
    K[:, 0, 0] = focal
    K[:, 1, 1] = focal
    K[:, 0, 2] = W / 2
    K[:, 1, 2] = H / 2
    K[:, 2, 2] = 1
    return K, E


class DeviceBatchLoader(object):
    """The batch interface of the reference's CameraGeometryLoader (camera_geometry_loader.py:68-152) over
    device-resident stores: ``images`` uint8 [N,H,W,4], optional ``depths`` [N,H,W], ``intrinsics`` [N,3,3],
    ``extrinsics`` [N,4,4], ``translation_center`` [1,3].  Subclasses fill the stores."""

    images = None
    depths = None
    depths_bool = False

    def set_images(self, images_u8):
        assert images_u8.dtype == torch.uint8 and tuple(images_u8.shape) == (self.N, self.H, self.W, 4)
        self.images = images_u8.to(self.device)

    def format_groundtruth(self, gt, background=None):
        """camera_geometry_loader.py:68-88."""
        if background is None:
            color_bg = torch.rand(3, device=gt.device, generator=self.generator)
        else:
            color_bg = torch.tensor(background, dtype=torch.float32, device=gt.device)
        rgba = gt.to(torch.float32) / 255 if gt.dtype == torch.uint8 else gt
        if rgba.shape[-1] == 4:
            rgb_gt = rgba[..., :3] * rgba[..., 3:] + color_bg * (1 - rgba[..., 3:])
        else:
            rgb_gt = rgba
        return rgb_gt, color_bg

    def get_custom_batch(self, n, h, w, background=None, device='cuda'):
        """camera_geometry_loader.py:91-108."""
        n, h, w = n.to(self.device), h.to(self.device), w.to(self.device)
        K = self.intrinsics[n]
        E = self._centred_extrinsics()[n]
        depth = self.depths[n, h, w] if (self.depths_bool and self.depths is not None) else None
        if self.images is not None:
            rgb_gt, color_bg = self.format_groundtruth(self.images[n, h, w, :], background)
        else:   # no rendered ground truth: deterministic pseudo-colours (throughput runs)
            color_bg = (torch.rand(3, device=self.device, generator=self.generator) if background is None
                        else torch.tensor(background, dtype=torch.float32, device=self.device))
            hw = torch.stack([h, w, h + w], -1)
            rgb_gt = (hw % 256).to(torch.float32) / 255
        return n, h, w, K, E, rgb_gt, color_bg, depth

    def _centred_extrinsics(self):
        """extrinsics with translation_center subtracted (camera_geometry_loader.py:95), cached per version of the
        two tensors so that a batch costs one gather instead of gather + clone + subtract + copy."""
        key = (self.extrinsics.data_ptr(), self.extrinsics._version, self.translation_center.data_ptr(),
               self.translation_center._version)
        if getattr(self, '_ec_key', None) != key:
            E = self.extrinsics.clone()
            E[..., :3, 3] = E[..., :3, 3] - self.translation_center
            self._ec, self._ec_key = E, key
        return self._ec

    def get_random_batch(self, batch_size, device='cuda'):
        """camera_geometry_loader.py:112-117, sampled on the GPU: one nerf_sample_batch launch for stores that live on
        the GPU (indices, calibration gather, uint8 -> float ground truth over the batch's random background colour);
        the torch formulation below serves host-resident stores (the CPU tests of the loader logic)."""
        if torch.device(self.device).type == 'cuda':
            return self._sample_batch_hip(batch_size)
        n = torch.randint(0, self.N, (batch_size,), device=self.device, generator=self.generator)
        h = torch.randint(0, self.H, (batch_size,), device=self.device, generator=self.generator)
        w = torch.randint(0, self.W, (batch_size,), device=self.device, generator=self.generator)
        return self.get_custom_batch(n, h, w)

    def _sample_batch_hip(self, B):
        import ctypes as C
        from .._lib import check, lib, ptr, stream
        dev = self.intrinsics.device
        i64 = lambda: torch.empty(B, dtype=torch.int64, device=dev)
        n, h, w = i64(), i64(), i64()
        K = torch.empty(B, 3, 3, device=dev)
        E = torch.empty(B, 4, 4, device=dev)
        gt = torch.empty(B, 3, device=dev)
        bg = torch.empty(3, device=dev)
        with_depth = self.depths_bool and self.depths is not None
        depth = torch.empty(B, device=dev) if with_depth else None
        img = self.images
        if img is not None and (img.dtype != torch.uint8 or not img.is_contiguous()):
            raise ValueError('the image store must be a contiguous uint8 tensor [N,H,W,3|4]')
        # a batch is a pure function of (generator seed, batches drawn since that seed was set): re-seeding the loader's
        # generator (generator.manual_seed) restarts the sequence, as it does for the torch formulation
        seed = self.generator.initial_seed()
        if getattr(self, '_seed_seen', None) != seed:
            self._seed_seen, self._batch_counter = seed, 0
        self._batch_counter += 1
        check(lib().nerf_sample_batch(ptr(img), ptr(self.depths) if with_depth else None, ptr(self.intrinsics),
                                      ptr(self._centred_extrinsics()), self.N, self.H, self.W,
                                      int(img.shape[-1]) if img is not None else 4, B,
                                      C.c_uint64(seed & (2 ** 64 - 1)), C.c_uint64(self._batch_counter),
                                      ptr(n), ptr(h), ptr(w), ptr(K), ptr(E), ptr(gt), ptr(bg), ptr(depth), stream()),
              'nerf_sample_batch')
        return n, h, w, K, E, gt, bg, depth

    def get_image_batch(self, image_num, device='cpu'):
        """camera_geometry_loader.py:120-126."""
        h = torch.arange(0, self.H, device=self.device)
        w = torch.arange(0, self.W, device=self.device)
        h, w = torch.meshgrid(h, w, indexing='ij')
        n = torch.full(h.shape, fill_value=image_num, device=self.device)
        return self.get_custom_batch(n, h, w, background=(1, 1, 1), device=device)

    def get_pointcloud_batch(self, cams, freq, device='cpu'):
        """camera_geometry_loader.py:129-143."""
        idx = [i for i in range(self.N) if (i % 6 in cams) and (i // 6 % freq == 0)]
        n = torch.tensor(idx, dtype=torch.int64, device=self.device)
        h = torch.arange(0, self.H, device=self.device)
        w = torch.arange(0, self.W, device=self.device)
        n, h, w = torch.meshgrid(n, h, w, indexing='ij')
        return self.get_custom_batch(n, h, w, background=(1, 1, 1), device=device)

    def get_calibration(self, device='cuda'):
        """camera_geometry_loader.py:146-152."""
        return self.N, self.H, self.W, self.intrinsics, self._centred_extrinsics().clone()


class ProceduralSceneLoader(DeviceBatchLoader):
    def __init__(self, n_images=100, H=800, W=800, seed=0, device='cuda', images=None, batch_seed=0):
        self.N, self.H, self.W = n_images, H, W
        self.device = device
        K, E = sphere_cameras(n_images, seed, H=H, W=W)
        self.intrinsics = K.to(device)
        self.extrinsics = E.to(device)
        self.translation_center = torch.zeros(1, 3, device=device)   # the sphere is centred on the origin
        self.depths_bool = False
        # uint8 RGBA like the reference store (camera_geometry_loader.py:41); white opaque until rendered
        self.images = images if images is not None else None
        self.generator = torch.Generator(device=device).manual_seed(batch_seed)

