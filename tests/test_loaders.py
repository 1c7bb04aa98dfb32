"""Loader contract (camera_geometry_loader.py:68-152) on CPU stores: multi-scan assembly and batch arithmetic."""
import numpy as np
import torch

from nerf_amd.loaders.camera_geometry_loader import CameraGeometryLoader


def _scan(n, H, W, seed, channels=3, depth=True):
    g = np.random.default_rng(seed)
    E = np.tile(np.eye(4, dtype=np.float32), (n, 1, 1))
    E[:, :3, 3] = g.normal(size=(n, 3))
    K = np.tile(np.array([[50.0, 0, W / 2], [0, 50.0, H / 2], [0, 0, 1]], dtype=np.float32), (n, 1, 1))
    return dict(rgb=g.integers(0, 256, size=(n, H, W, channels), dtype=np.uint8), intrinsics=K, extrinsics=E,
                depth=g.random(size=(n, H, W), dtype=np.float32) if depth else None)


def test_multi_scan_stores_and_batches():
    a, b = _scan(6, 8, 10, 0), _scan(12, 8, 10, 1, channels=4)
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1]], dtype=np.float32)
    T[:3, 3] = [0.5, -1.0, 2.0]
    b['transform'] = T
    L = CameraGeometryLoader.from_arrays([a, b], device='cpu', batch_seed=3)
    assert (L.N, L.H, L.W) == (18, 8, 10) and L.depths_bool
    assert L.images.dtype == torch.uint8 and tuple(L.images.shape) == (18, 8, 10, 4)
    assert bool((L.images[:6, ..., 3] == 255).all())                                  # RGB frames: opaque alpha
    E_all = np.concatenate([a['extrinsics'], T @ b['extrinsics']])
    np.testing.assert_allclose(L.extrinsics.numpy(), E_all, atol=1e-6)
    tc = E_all[:, :3, 3].mean(0, keepdims=True)
    np.testing.assert_allclose(L.translation_center.numpy(), tc, atol=1e-6)

    n = torch.tensor([0, 5, 6, 17])
    h = torch.tensor([0, 7, 3, 2])
    w = torch.tensor([9, 0, 4, 4])
    bn, bh, bw, K, E, rgb, bg, depth = L.get_custom_batch(n, h, w, background=(0.25, 0.5, 0.75), device='cpu')
    np.testing.assert_allclose(E[:, :3, 3].numpy(), E_all[n.numpy(), :3, 3] - tc, atol=1e-6)
    np.testing.assert_allclose(E[:, :3, :3].numpy(), E_all[n.numpy(), :3, :3], atol=1e-6)
    rgba = np.concatenate([np.concatenate([a['rgb'], np.full((6, 8, 10, 1), 255, np.uint8)], -1), b['rgb']])
    px = rgba[n.numpy(), h.numpy(), w.numpy()].astype(np.float32) / 255
    want = px[:, :3] * px[:, 3:] + np.array([0.25, 0.5, 0.75], np.float32) * (1 - px[:, 3:])
    np.testing.assert_allclose(rgb.numpy(), want, atol=1e-6)
    dd = np.concatenate([a['depth'], b['depth']])
    np.testing.assert_allclose(depth.numpy(), dd[n.numpy(), h.numpy(), w.numpy()])
    assert L.extrinsics[0, 0, 3] == float(E_all[0, 0, 3])                              # the store itself is not shifted

    rn, rh, rw, rK, rE, rrgb, rbg, rdepth = L.get_random_batch(64, device='cpu')
    assert rn.shape == (64,) and int(rn.max()) < 18 and int(rh.max()) < 8 and int(rw.max()) < 10
    assert rrgb.shape == (64, 3) and rbg.shape == (3,) and rdepth.shape == (64,)

    n_i, h_i, w_i, K_i, E_i, rgb_i, bg_i, _ = L.get_image_batch(7, device='cpu')
    assert tuple(h_i.shape) == (8, 10) and bool((n_i == 7).all()) and bool((bg_i == 1).all())
    assert int(h_i[3, 4]) == 3 and int(w_i[3, 4]) == 4
    pn, ph, pw, *_ = L.get_pointcloud_batch(cams=[1, 2], freq=2, device='cpu')
    idx = [i for i in range(18) if i % 6 in (1, 2) and (i // 6) % 2 == 0]
    assert sorted(set(pn.reshape(-1).tolist())) == idx and tuple(pn.shape) == (len(idx), 8, 10)
    N_, H_, W_, Kc, Ec = L.get_calibration(device='cpu')
    np.testing.assert_allclose(Ec[:, :3, 3].numpy(), E_all[:, :3, 3] - tc, atol=1e-6)


def test_scans_without_depth_and_missing_package():
    L = CameraGeometryLoader.from_arrays([_scan(3, 4, 4, 2, depth=False)], device='cpu')
    assert not L.depths_bool and L.get_random_batch(5, device='cpu')[-1] is None
    try:
        CameraGeometryLoader(['x'], [None], [None], 0.25, device='cpu')
    except ImportError as e:
        assert 'from_arrays' in str(e)
    else:
        raise AssertionError('expected ImportError without camera_geometry')
