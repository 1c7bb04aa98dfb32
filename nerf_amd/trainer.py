"""Trainer with the reference's surface (trainer.py:15-183): same constructor arguments,
``train`` / ``train_epoch`` / ``train_step``.  Differences, all on purpose:

* no torch.cuda.amp: the fp32 semantics are the parity spec;
* the three losses and their gradients come from ONE kernel (helpers.fused_train_loss);
* scalars are logged as device tensors (no .item() sync per iteration);
* LPIPS / SSIM (torchmetrics + downloaded VGG weights, trainer.py:68-69) are out of scope;
* under torch.distributed every rank trains on its own ray batch and the gradients are
  all-reduced (RCCL) before the fused Adam step, which applies the 1/world factor.
"""
import math as m
import time

import numpy as np
import torch

from . import helpers, parallel


class Trainer(object):
    def __init__(self, model, dataloader, logger, inferencer, renderer, optimizer, scheduler, n_rays, num_epochs,
                 iters_per_epoch, eval_image_freq, eval_pointcloud_freq, save_weights_freq, dist_loss_lambda1,
                 dist_loss_lambda2, depth_loss_lambda1, depth_loss_lambda2):
        self.model = model
        self.dataloader = dataloader
        self.logger = logger
        self.inferencer = inferencer
        self.renderer = renderer
        self.optimizer = optimizer
        self.scheduler = scheduler

        self.n_rays = n_rays
        self.num_epochs = num_epochs
        self.iters_per_epoch = iters_per_epoch
        self.num_iters = self.num_epochs * self.iters_per_epoch

        self.eval_image_freq = eval_image_freq if eval_image_freq != 'end' else num_epochs
        self.eval_pointcloud_freq = eval_pointcloud_freq if eval_pointcloud_freq != 'end' else num_epochs
        self.save_weights_freq = save_weights_freq if save_weights_freq != 'end' else num_epochs

        self.dist_loss_lambda1 = dist_loss_lambda1
        self.dist_loss_lambda2 = dist_loss_lambda2
        self.depth_loss_lambda1 = depth_loss_lambda1
        self.depth_loss_lambda2 = depth_loss_lambda2

        self.iter = 0
        self.t0_train = time.time()
        self.noise = None   # optional injected stratification noise (parity tests)

    def train(self):
        self.t0_train = time.time()
        for epoch in range(self.num_epochs):
            self.train_epoch(self.iters_per_epoch)

            if self.logger is not None:
                self.logger.log(f'Iteration: {self.iter}')
                for key, val in self.logger.scalars.items():
                    self.logger.log(f'Scalar: {key} - Value: {np.mean(np.array(val[-self.iters_per_epoch:])).item():.6f}')
                self.logger.log('')

            if self.eval_image_freq is not None and self.inferencer is not None:
                if (epoch + 1) % self.eval_image_freq == 0:
                    self.logger.log('Rending Image...')
                    n, h, w, K, E, rgb_gt, _, _ = self.dataloader.get_image_batch(self.inferencer.image_num, device='cuda')
                    image, invdepth = self.inferencer.render_image(n, h, w, K, E)
                    self.logger.image_color('image', image, self.iter)
                    self.logger.image_grey('invdepth', invdepth, self.iter)
                    pred = torch.clamp(torch.as_tensor(image.copy()).to(rgb_gt.device), min=0, max=1)
                    self.logger.eval_scalar('eval_psnr', helpers.psnr(pred.reshape(-1, 3), rgb_gt.reshape(-1, 3)), self.iter)
                    invdepth_thresh = self.inferencer.render_invdepth_thresh(n, h, w, K, E)
                    self.logger.image_grey('invdepth_thresh', invdepth_thresh, self.iter)

            if self.eval_pointcloud_freq is not None and self.inferencer is not None:
                if (epoch + 1) % self.eval_pointcloud_freq == 0:
                    self.logger.log('Generating Pointcloud...')
                    n, h, w, K, E, _, _, _ = self.dataloader.get_pointcloud_batch(cams=self.inferencer.cams, freq=self.inferencer.freq)
                    pointcloud = self.inferencer.extract_surface_geometry(n, h, w, K, E)
                    self.logger.pointcloud(pointcloud, self.iter, self.inferencer.max_variance_pcd)

            if self.save_weights_freq is not None:
                if (epoch + 1) % self.save_weights_freq == 0:
                    if hasattr(self.model, 'sync_table'):
                        self.model.sync_table()      # level-sharded table: collect every level from its owner (all ranks)
                    if parallel.rank() == 0:
                        self.logger.log('Saving Model...')
                        self.logger.model(self.model, self.iter)

    # ---- checkpoint / resume (the reference only saves model weights, logger.py:101-103) ----
    def save_checkpoint(self, path):
        """Model, optimizer moments, LR schedule and iteration count.  Collective under a level-sharded table
        (every rank must call it); rank 0 writes the file."""
        if hasattr(self.model, 'sync_table'):
            self.model.sync_table()
        if hasattr(self.optimizer, 'sync_sharded_state'):
            self.optimizer.sync_sharded_state()
        if parallel.rank() == 0:
            torch.save({'iter': self.iter, 'model': self.model.state_dict(), 'optimizer': self.optimizer.state_dict(),
                        'scheduler': self.scheduler.state_dict() if self.scheduler is not None else None}, path)

    def load_checkpoint(self, path, map_location='cuda'):
        """Resume from a file written by save_checkpoint.  NOT a loader for .pth files trained by the reference: the keys
        match (encoder.params, sigma_net.params, color_net.params, latent_emb) but tinycudann's FullyFusedMLP stores its
        layers padded to multiples of 16 inputs / outputs, so load_state_dict reports a size mismatch for them.
        nerf_amd.tcnn_compat.from_reference_state_dict converts tiny-cuda-nn's published layout (UNPINNED: tinycudann is
        absent here and the reference ships no checkpoint to check it against); nothing is converted silently."""
        ck = torch.load(path, map_location=map_location, weights_only=True)
        self.model.load_state_dict(ck['model'])
        self.optimizer.load_state_dict(ck['optimizer'])
        if self.scheduler is not None and ck.get('scheduler') is not None:
            self.scheduler.load_state_dict(ck['scheduler'])
        self.iter = int(ck['iter'])
        if getattr(self.model, 'direct_grads', False):     # load_state_dict keeps the tensors; make sure grads exist
            self.model.enable_direct_grads(True)
        return self

    def train_epoch(self, iters_per_epoch):
        for i in range(iters_per_epoch):
            self.optimizer.zero_grad()
            loss = self.train_step()
            world = parallel.world_size()
            flag = getattr(loss, 'nonfinite_flag', None)
            guarded = flag is not None and hasattr(self.optimizer, 'skip_flag')
            if guarded:
                self.optimizer.skip_flag = flag      # a non-finite loss drops this step on the device (no host sync)
            if world > 1:
                skip = lambda: ([self.optimizer._table_param]
                                if getattr(self.optimizer, 'table_reduced_by_pipeline', lambda: False)() else [])
                # (the flag is summed with the small gradients: every rank must take the same decision)
                parallel.backward_and_reduce(loss, self.model, [p for g in self.optimizer.param_groups for p in g['params']],
                                             skip=skip, flags=[flag] if guarded else [])
                if hasattr(self.optimizer, 'grad_scale'):
                    self.optimizer.grad_scale = 1.0 / world
            else:
                loss.backward()
            if world > 1 and hasattr(self.renderer, 'prefetch_batch'):
                # the next iteration's batch, its rays and (level-sharded table) the ray all-gather of its coarse pass are
                # issued BEFORE the optimizer step: they depend on no parameter, and the collective then completes under
                # the table update instead of opening the next step.  Same batches in the same order.
                self._next_batch = self.dataloader.get_random_batch(self.n_rays)
                self.renderer.prefetch_batch(*self._next_batch[:5])
            self.optimizer.step()
            if self.scheduler is not None:
                self.scheduler.step()
            self.iter += 1

    def loss_lambdas(self):
        """trainer.py:154-167 (the depth schedule is constant in the reference: lambda2 - lambda2)."""
        if self.dist_loss_lambda1 == 0 or self.dist_loss_lambda2 == 0:
            dist_scalar = 0
        else:
            dist_scalar = 10 ** (self.iter / self.num_iters * (m.log10(self.dist_loss_lambda2) - m.log10(self.dist_loss_lambda1))
                                 + m.log10(self.dist_loss_lambda1))
        if self.depth_loss_lambda1 == 0 or self.depth_loss_lambda2 == 0:
            depth_scalar = 0
        else:
            depth_scalar = 10 ** (self.iter / self.num_iters * (m.log10(self.depth_loss_lambda2) - m.log10(self.depth_loss_lambda2))
                                  + m.log10(self.depth_loss_lambda2))
        return dist_scalar, depth_scalar

    def train_step(self):
        self.model.train()
        batch, self._next_batch = getattr(self, '_next_batch', None), None
        if batch is None:
            batch = self.dataloader.get_random_batch(self.n_rays)
        n, h, w, K, E, rgb_gt, color_bg, _ = batch
        dist_scalar, depth_scalar = self.loss_lambdas()
        if hasattr(self.renderer, 'render_loss'):
            # trainer.py:149-169 with compositing and the three losses in one kernel (and one for their backward)
            loss, parts, rgb, weights, z_vals_log_s, aux_outputs = self.renderer.render_loss(
                n, h, w, K, E, color_bg, rgb_gt, dist_scalar, depth_scalar, noise=self.noise)
        else:
            rgb, weights, z_vals_log_s, aux_outputs = self.renderer.render(n, h, w, K, E, color_bg, noise=self.noise)
            loss, parts = helpers.fused_train_loss(rgb, rgb_gt, weights, z_vals_log_s, dist_scalar, depth_scalar)

        if self.logger is not None:
            loss_rgb, loss_dist, loss_depth, mse = parts[0], parts[1], parts[2], parts[3]
            psnr = -10.0 * torch.log10(mse)
            self.logger.scalar('loss', loss, self.iter)
            self.logger.scalar('loss_rgb', loss_rgb, self.iter)
            self.logger.scalar('psnr_rgb', psnr, self.iter)
            self.logger.scalar('dist_scalar', dist_scalar, self.iter)
            self.logger.scalar('loss_dist', loss_dist if dist_scalar != 0 else 0, self.iter)
            self.logger.scalar('depth_scalar', depth_scalar, self.iter)
            self.logger.scalar('loss_depth', loss_depth if depth_scalar != 0 else 0, self.iter)
            secs = int(time.time() - self.t0_train)
            self.logger.scalar('loss (seconds)', loss, secs)
            self.logger.scalar('loss_rgb (seconds)', loss_rgb, secs)
            self.logger.scalar('psnr_rgb (seconds)', psnr, secs)
        return loss
