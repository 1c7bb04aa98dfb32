"""Multi-camera scan loader with the reference's surface (loaders/camera_geometry_loader.py:10-152), stores on the
GPU and batches sampled there.

The reference reads scans with the third-party ``camera_geometry`` package, which is not available offline; the
same constructor signature is kept and raises a clear error without it.  ``from_arrays`` builds the identical
stores from plain arrays (what ``load_scan`` / ``load_frames`` deliver frame by frame), including several scans
with a rigid ``transform`` each (configs/config_bothsides.yaml:1-16):

    images      uint8 [N,H,W,4], alpha filled with 255 when the frames are RGB       (:41, :52)
    depths      float32 [N,H,W] when every scan has depth                             (:42-43, :53-54)
    intrinsics  [N,3,3];  extrinsics [N,4,4] = transform @ frame extrinsic            (:57-58)
    translation_center = mean camera position, subtracted from E in every batch       (:61, :95)
"""
import numpy as np
import torch

from .procedural import DeviceBatchLoader


class CameraGeometryLoader(DeviceBatchLoader):
    def __init__(self, scene_paths, frame_ranges, transforms, image_scale, device='cuda', batch_seed=0):
        try:
            from camera_geometry.scan import load_scan
            from camera_geometry.scan.views import load_frames
        except ImportError as e:      # the package is not part of this image: say so instead of half-working
            raise ImportError('CameraGeometryLoader(scene_paths, ...) needs the camera_geometry package; build the '
                              'loader from arrays with CameraGeometryLoader.from_arrays(scans)') from e
        scans = []
        for scene_path, frame_range, transform_path in zip(scene_paths, frame_ranges, transforms):
            scene = load_scan(scene_path, frame_range=frame_range, image_scale=image_scale)
            frames = [f for rig in load_frames(scene) for f in rig]
            scans.append(dict(rgb=np.stack([f.rgb for f in frames]),
                              depth=None if frames[0].depth is None else np.stack([f.depth for f in frames]),
                              intrinsics=np.stack([f.camera.intrinsic for f in frames]),
                              extrinsics=np.stack([f.camera.extrinsic for f in frames]),
                              transform=None if transform_path is None else np.load(transform_path)))
        self._fill(scans, device, batch_seed)

    @classmethod
    def from_arrays(cls, scans, device='cuda', batch_seed=0):
        """scans: list of dicts with ``rgb`` uint8 [n,H,W,3|4], ``intrinsics`` [n,3,3], ``extrinsics`` [n,4,4]
        (camera-to-world) and optionally ``depth`` [n,H,W], ``transform`` [4,4]."""
        self = cls.__new__(cls)
        self._fill(scans, device, batch_seed)
        return self

    def _fill(self, scans, device, batch_seed):
        t = lambda a, dt: torch.as_tensor(np.asarray(a)).to(dt)
        self.device = device
        self.N = sum(int(np.asarray(s['rgb']).shape[0]) for s in scans)
        self.H, self.W = (int(v) for v in np.asarray(scans[0]['rgb']).shape[1:3])
        self.depths_bool = all(s.get('depth') is not None for s in scans)
        images = torch.full([self.N, self.H, self.W, 4], 255, dtype=torch.uint8)
        depths = torch.full([self.N, self.H, self.W], 1, dtype=torch.float32) if self.depths_bool else None
        K = torch.zeros([self.N, 3, 3], dtype=torch.float32)
        E = torch.zeros([self.N, 4, 4], dtype=torch.float32)
        i = 0
        for s in scans:
            rgb = t(s['rgb'], torch.uint8)
            n = rgb.shape[0]
            if tuple(rgb.shape[1:3]) != (self.H, self.W):
                raise ValueError('all scans must share one image size')
            images[i:i + n, :, :, :rgb.shape[-1]] = rgb
            if self.depths_bool:
                depths[i:i + n] = t(s['depth'], torch.float32)
            K[i:i + n] = t(s['intrinsics'], torch.float32)
            T = torch.eye(4) if s.get('transform') is None else t(s['transform'], torch.float32)
            E[i:i + n] = T @ t(s['extrinsics'], torch.float32)
            i += n
        self.images = images.to(device)
        self.depths = depths.to(device) if depths is not None else None
        self.intrinsics = K.to(device)
        self.extrinsics = E.to(device)
        self.translation_center = torch.mean(self.extrinsics[..., :3, 3], dim=0, keepdim=True)
        self.generator = torch.Generator(device=device).manual_seed(batch_seed)
