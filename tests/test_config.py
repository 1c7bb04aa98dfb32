"""Config loader keeps the reference's keys / CLI (misc.py:10-35, configs/config_base.yaml)."""
import os

from nerf_amd.config import configurator

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_base_config_keys():
    """config_base.yaml carries the reference's own values (configs/config_base.yaml:25-76)."""
    cfg = configurator(os.path.join(ROOT, 'configs', 'config_base.yaml'), argv=[])
    assert cfg.trainer.n_rays == 16384 and cfg.trainer.dist_loss_lambda1 == 1e-5
    assert cfg.optimizer.eps == 1e-15 and cfg.optimizer.betas == [0.9, 0.99]
    assert cfg.renderer.steps == [256, 128, 64, 32] and cfg.renderer.z_bounds == [0.1, 1, 10, 100, 1000]
    assert cfg.scene.image_scale == 0.5 and cfg.inference.image.rotate == 270 and cfg.log.eval_pointcloud_freq == 'end'
    assert cfg.nets.encoding.n_levels == 20 and cfg.nets.encoding.log2_hashmap_size == 24
    assert cfg.nets.sigma.geo_feat_dim == 15 and cfg.nets.color.num_layers == 3
    assert cfg.log.eval_image_freq == 'end' and cfg.inference.image.image_num == 'middle'


def test_merge_and_dotlist():
    base = os.path.join(ROOT, 'configs', 'config_base.yaml')
    over = os.path.join(ROOT, 'configs', 'config_metric.yaml')
    cfg = configurator(None, argv=['--configs', base, over, '--params', 'trainer.num_epochs=2', 'optimizer.net.lr=5e-4',
                                   'scheduler=exp_decay'])
    assert cfg.renderer.steps == [64] and cfg.renderer.z_bounds == [1.6, 4.8]
    assert cfg.trainer.n_rays == 4096 and cfg.trainer.num_epochs == 2
    assert cfg.optimizer.net.lr == 5e-4 and cfg.optimizer.net.weight_decay == 1e-6
    assert cfg.scheduler == 'exp_decay' and cfg.renderer.importance_steps == 128


def test_bothsides_overlay_and_synthetic_scans():
    """configs/config_bothsides.yaml (:1-16) merged over the base: two scans, the second with a rigid transform,
    image_scale 0.25, log.root_dir nested under scene as in the reference's file; the generated scans have the store
    layout camera_geometry_loader.py:36-66 assembles."""
    import numpy as np
    from nerf_amd.loaders.camera_geometry_loader import CameraGeometryLoader
    from nerf_amd.loaders.synthetic_scans import make_scans
    base = os.path.join(ROOT, 'configs', 'config_base.yaml')
    over = os.path.join(ROOT, 'configs', 'config_bothsides.yaml')
    cfg = configurator(None, argv=['--configs', base, over])
    assert len(cfg.scene.scene_paths) == 2 and cfg.scene.transforms[0] is None and cfg.scene.image_scale == 0.25
    assert cfg.scene.log.root_dir == './logs/synthetic_bothsides' and cfg.trainer.n_rays == 16384
    assert cfg.renderer.steps == [256, 128, 64, 32]
    scans = make_scans(['synthetic_scan:12', 'synthetic_scan:12'], cfg.scene.transforms, 0.05)
    L = CameraGeometryLoader.from_arrays(scans, device='cpu')
    assert L.N == 24 and L.depths_bool and tuple(L.images.shape) == (24, L.H, L.W, 4)
    T = np.asarray(cfg.scene.transforms[1], dtype=np.float32)
    np.testing.assert_allclose(L.extrinsics[12:].numpy(), T @ scans[1]['extrinsics'], atol=1e-6)
    np.testing.assert_allclose(L.extrinsics[:12].numpy(), scans[0]['extrinsics'], atol=1e-6)
    assert bool((L.images[:12, ..., 3] == 255).all()) and int(L.images[12:, ..., 3].min()) == 64


def test_logger_scalars_mix_of_floats_and_tensors_read_twice():
    """Logger.scalars is read once per epoch (trainer.py:57): values already converted to floats by an earlier read
    and tensors logged since must mix (regression: torch.stack of CPU floats with device tensors)."""
    import torch
    from nerf_amd.logger import Logger
    lg = Logger(None, verbose=False)
    lg.scalar('loss', torch.tensor(1.5), 0)
    lg.scalar('loss', 2.5, 1)
    assert lg.scalars['loss'] == [1.5, 2.5]
    lg.scalar('loss', torch.tensor(3.5), 2)
    lg.scalar('lam', 0, 2)
    assert lg.scalars['loss'] == [1.5, 2.5, 3.5] and lg.scalars['lam'] == [0.0]
