#!/usr/bin/env python3
"""Example wiring, mirroring the reference's train.py:20-125 on the procedural scene:
loader -> NeRFNetwork -> NerfRenderer -> Inferencer -> Adam (3 groups) -> LambdaLR -> Trainer.

    python train.py [--configs configs/config_base.yaml more.yaml] [--params trainer.num_epochs=2 ...]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 train.py ...   (data parallel)
"""
import torch

from nerf_amd import parallel
from nerf_amd.config import configurator
from nerf_amd.inference import Inferencer
from nerf_amd.loaders.camera_geometry_loader import CameraGeometryLoader
from nerf_amd.loaders.procedural import ProceduralSceneLoader
from nerf_amd.loaders.synthetic_scans import make_scans
from nerf_amd.logger import Logger
from nerf_amd.nets import NeRFNetwork
from nerf_amd.optim import FusedAdam
from nerf_amd.renderer import NerfRenderer
from nerf_amd.trainer import Trainer

if __name__ == '__main__':
    cfg = configurator('./configs/config_base.yaml')
    rank, world, local = parallel.init_from_env()
    torch.cuda.set_device(local)
    # (config_bothsides.yaml nests log.root_dir under `scene`, as the reference's file does: honour it)
    root_dir = cfg.scene.get('log', {}).get('root_dir', cfg.log.root_dir)
    logger = Logger(root_dir=root_dir if rank == 0 else None, cfg=cfg, verbose=rank == 0)

    logger.log('Initiating Dataloader...')
    paths = [str(p) for p in cfg.scene.scene_paths]
    if paths == ['procedural']:
        size = cfg.scene.get('image_size', int(1600 * cfg.scene.image_scale))
        dataloader = ProceduralSceneLoader(cfg.scene.get('n_images', 100), size, size, batch_seed=1000 + rank)
    elif all(p.startswith('synthetic_scan') for p in paths):
        dataloader = CameraGeometryLoader.from_arrays(make_scans(paths, cfg.scene.transforms, cfg.scene.image_scale),
                                                      batch_seed=1000 + rank)
    else:       # the reference's wiring (train.py:30-35): scan files read by the camera_geometry package
        dataloader = CameraGeometryLoader(scene_paths=cfg.scene.scene_paths, frame_ranges=cfg.scene.frame_ranges,
                                          transforms=cfg.scene.transforms, image_scale=cfg.scene.image_scale,
                                          batch_seed=1000 + rank)

    logger.log('Initilising Model...')
    model = NeRFNetwork(
        N=dataloader.N,
        encoding_precision=cfg.nets.encoding.precision,
        encoding_n_levels=cfg.nets.encoding.n_levels,
        encoding_n_features_per_level=cfg.nets.encoding.n_features_per_level,
        encoding_log2_hashmap_size=cfg.nets.encoding.log2_hashmap_size,
        geo_feat_dim=cfg.nets.sigma.geo_feat_dim,
        sigma_hidden_dim=cfg.nets.sigma.hidden_dim,
        sigma_num_layers=cfg.nets.sigma.num_layers,
        encoding_dir_precision=cfg.nets.encoding_dir.precision,
        encoding_dir_encoding=cfg.nets.encoding_dir.encoding,
        encoding_dir_degree=cfg.nets.encoding_dir.degree,
        latent_embedding_dim=cfg.nets.latent_embedding.features,
        color_hidden_dim=cfg.nets.color.hidden_dim,
        color_num_layers=cfg.nets.color.num_layers,
        seed=0)

    renderer = NerfRenderer(model=model, inner_bound=cfg.scene.inner_bound, outer_bound=cfg.scene.outer_bound,
                            z_bounds=cfg.renderer.z_bounds, steps_firstpass=cfg.renderer.steps,
                            steps_importance=cfg.renderer.importance_steps, alpha_importance=cfg.renderer.alpha)

    image_num = int(dataloader.N / 2) + 3 if cfg.inference.image.image_num == 'middle' else cfg.inference.image.image_num
    inferencer = Inferencer(renderer=renderer, n_rays=cfg.trainer.n_rays, image_num=image_num,
                            rotate=cfg.inference.image.rotate,
                            max_variance_npy=cfg.inference.pointcloud.max_variance_npy,
                            max_variance_pcd=cfg.inference.pointcloud.max_variance_pcd,
                            distribution_area=cfg.inference.pointcloud.distribution_area,
                            cams=cfg.inference.pointcloud.cams, freq=cfg.inference.pointcloud.freq)

    optimizer = FusedAdam([
        {'name': 'encoding', 'params': list(model.encoder.parameters()), 'lr': cfg.optimizer.encoding.lr},
        {'name': 'latent_emb', 'params': [model.latent_emb], 'lr': cfg.optimizer.latent_emb.lr},
        {'name': 'net', 'params': list(model.sigma_net.parameters()) + list(model.color_net.parameters()),
         'weight_decay': cfg.optimizer.net.weight_decay, 'lr': cfg.optimizer.net.lr},
    ], betas=tuple(cfg.optimizer.betas), eps=cfg.optimizer.eps).attach(model)

    if cfg.scheduler == 'step':
        lmbda = lambda x: 1
    elif cfg.scheduler == 'exp_decay':
        lmbda = lambda x: 0.1 ** (x / (cfg.trainer.num_epochs * cfg.trainer.iters_per_epoch))
    else:
        raise ValueError
    scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=lmbda, last_epoch=-1)

    trainer = Trainer(model=model, dataloader=dataloader, logger=logger, inferencer=inferencer, renderer=renderer,
                      optimizer=optimizer, scheduler=scheduler, n_rays=cfg.trainer.n_rays,
                      num_epochs=cfg.trainer.num_epochs, iters_per_epoch=cfg.trainer.iters_per_epoch,
                      eval_image_freq=cfg.log.eval_image_freq, eval_pointcloud_freq=cfg.log.eval_pointcloud_freq,
                      save_weights_freq=cfg.log.save_weights_freq,
                      dist_loss_lambda1=cfg.trainer.dist_loss_lambda1, dist_loss_lambda2=cfg.trainer.dist_loss_lambda2,
                      depth_loss_lambda1=cfg.trainer.depth_loss_lambda1, depth_loss_lambda2=cfg.trainer.depth_loss_lambda2)
    logger.log('Beginning Training...\n')
    trainer.train()
