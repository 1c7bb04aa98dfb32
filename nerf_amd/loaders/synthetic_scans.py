"""Generated multi-camera scan stores in the layout the reference's CameraGeometryLoader assembles from
``camera_geometry`` scans (loaders/camera_geometry_loader.py:36-66): per scan ``rgb`` uint8 [n,H,W,3|4], ``depth``
[n,H,W], ``intrinsics`` [n,3,3], ``extrinsics`` [n,4,4] (camera-to-world) and an optional rigid ``transform`` that
registers the scan onto the first one (configs/config_bothsides.yaml:9-11).

The reference's scans are rigs of 6 cameras moving along a vine row (get_pointcloud_batch indexes ``i % 6`` /
``i // 6``, camera_geometry_loader.py:129-137); the generated ones keep that structure: ``frames`` positions along a
line, 6 cameras per position looking at the row from one side.  No real data exists offline; pixel values are a smooth
pseudo-texture of (scan, frame, row, column) so that every pixel is distinguishable in tests.
"""
import math

import numpy as np


def make_scan(n_positions, H, W, seed=0, side=+1.0, rgba=True, depth=True, focal=None):
    """One scan: 6 * n_positions frames.  ``side`` = +1 / -1: which side of the row (x = 0 plane) the rig travels on."""
    g = np.random.default_rng(seed)
    n = 6 * n_positions
    f = float(focal if focal is not None else 1.2 * W)
    K = np.tile(np.array([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]], dtype=np.float32), (n, 1, 1))
    E = np.zeros((n, 4, 4), dtype=np.float32)
    for i in range(n):
        pos_i, cam = divmod(i, 6)
        y = -1.0 + 2.0 * (pos_i + 0.5) / n_positions                      # travel along the row
        z = -0.5 + 0.2 * cam                                              # cameras stacked on the rig's mast
        eye = np.array([side * 1.6, y, z], dtype=np.float64)
        target = np.array([0.0, y + 0.1 * (cam - 2.5), 0.0])
        fwd = target - eye
        fwd /= np.linalg.norm(fwd)
        right = np.cross(fwd, np.array([0.0, 0.0, 1.0]))
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        E[i, :3, 0], E[i, :3, 1], E[i, :3, 2], E[i, :3, 3] = right, down, fwd, eye      # OpenCV axes: +z forward
        E[i, 3, 3] = 1
    hh, ww = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
    frame = np.arange(n)[:, None, None]
    chans = [(37 * frame + 3 * hh + 5 * ww + 11 * seed) % 256, (91 * frame + 7 * hh + 2 * ww + 29) % 256,
             (53 * frame + hh * ww + 17 * seed) % 256]
    if rgba:      # a band of transparent pixels so that the background blend (camera_geometry_loader.py:81) matters
        chans.append(np.where((hh + ww + frame) % 7 == 0, 64, 255))
    rgb = np.stack(chans, -1).astype(np.uint8)
    out = dict(rgb=rgb, intrinsics=K, extrinsics=E, depth=None, transform=None)
    if depth:
        out['depth'] = (1.0 + 0.5 * np.sin(0.1 * hh + 0.07 * ww + 0.3 * frame) + 0.01 * g.random((n, H, W))).astype(np.float32)
    return out


def make_scans(scene_paths, transforms, image_scale, seed=0, full_size=(512, 640)):
    """The scan list for ``CameraGeometryLoader.from_arrays`` from config values: every ``synthetic_scan:<frames>`` entry
    of scene.scene_paths makes one scan of that many frames (rounded up to whole 6-camera rigs); the scans alternate
    sides of the row; ``transforms[i]`` (None, a .npy path, or a 4x4 nested list) registers scan i, as at
    camera_geometry_loader.py:22-25,59.  ``image_scale`` scales the nominal full-size frames like load_scan's
    image_scale does."""
    H, W = (max(8, int(round(s * image_scale))) for s in full_size)
    scans = []
    for i, (path, tr) in enumerate(zip(scene_paths, transforms)):
        frames = int(str(path).split(':', 1)[1]) if ':' in str(path) else 96
        sc = make_scan(int(math.ceil(frames / 6)), H, W, seed=seed + i, side=+1.0 if i % 2 == 0 else -1.0, rgba=(i % 2 == 1))
        if tr is not None:
            sc['transform'] = np.load(tr).astype(np.float32) if isinstance(tr, str) else np.asarray(tr, dtype=np.float32)
        scans.append(sc)
    return scans
