"""Config loader keeps the reference's keys / CLI (misc.py:10-35, configs/config_base.yaml)."""
import os

from nerf_amd.config import configurator

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_base_config_keys():
    cfg = configurator(os.path.join(ROOT, 'configs', 'config_base.yaml'), argv=[])
    assert cfg.trainer.n_rays == 4096 and cfg.trainer.dist_loss_lambda1 == 1e-5
    assert cfg.optimizer.eps == 1e-15 and cfg.optimizer.betas == [0.9, 0.99]
    assert cfg.renderer.steps == [64] and len(cfg.renderer.z_bounds) == len(cfg.renderer.steps) + 1
    assert cfg.nets.encoding.n_levels == 20 and cfg.nets.encoding.log2_hashmap_size == 24
    assert cfg.nets.sigma.geo_feat_dim == 15 and cfg.nets.color.num_layers == 3
    assert cfg.log.eval_image_freq == 'end' and cfg.inference.image.image_num == 'middle'


def test_merge_and_dotlist():
    base = os.path.join(ROOT, 'configs', 'config_base.yaml')
    over = os.path.join(ROOT, 'configs', 'config_reference_sampling.yaml')
    cfg = configurator(None, argv=['--configs', base, over, '--params', 'trainer.num_epochs=2', 'optimizer.net.lr=5e-4',
                                   'scheduler=exp_decay'])
    assert cfg.renderer.steps == [256, 128, 64, 32] and cfg.renderer.z_bounds[-1] == 1000
    assert cfg.trainer.n_rays == 16384 and cfg.trainer.num_epochs == 2
    assert cfg.optimizer.net.lr == 5e-4 and cfg.optimizer.net.weight_decay == 1e-6
    assert cfg.scheduler == 'exp_decay' and cfg.renderer.importance_steps == 128


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
