"""Checkpoint / resume: Trainer.save_checkpoint after k steps + load into fresh objects must
continue exactly like the uninterrupted run (model, Adam moments, step count, LR schedule, iteration)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_gpu_parity import NET_CFGS, _make_pair  # noqa: E402


def _make_trainer(cfg='small'):
    from nerf_amd.loaders.procedural import ProceduralSceneLoader
    from nerf_amd.optim import FusedAdam
    from nerf_amd.renderer import NerfRenderer
    from nerf_amd.trainer import Trainer
    from nerf_amd.logger import Logger

    class SeededLoader(ProceduralSceneLoader):
        counter = 0

        def get_random_batch(self, batch_size, device='cuda'):
            self.generator.manual_seed(1000 + self.counter)      # batch k is the same in every run
            self.counter += 1
            return super().get_random_batch(batch_size, device)

    net, _ = _make_pair(NET_CFGS[cfg])
    R = NerfRenderer(net, 2, 2.5, [1.6, 4.8], [64], 128, 0.2)
    loader = SeededLoader(7, 64, 64, seed=0, batch_seed=0)
    opt = FusedAdam([{'params': list(net.encoder.parameters()), 'lr': 2e-2}, {'params': [net.latent_emb], 'lr': 2e-2},
                     {'params': list(net.sigma_net.parameters()) + list(net.color_net.parameters()), 'weight_decay': 1e-6,
                      'lr': 1e-3}], betas=(0.9, 0.99), eps=1e-15).attach(net)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda it: 0.5 ** (it / 4))
    tr = Trainer(net, loader, Logger(None, verbose=False), None, R, opt, sched, 96, 1, 8, None, None, None, 1e-5, 1e-2, 0.0, 1e-1)
    tr.noise = torch.rand(96, 128, generator=torch.Generator().manual_seed(3)).cuda()
    return tr


def _params(tr):
    m = tr.model
    return [p.detach().clone() for p in (m.encoder.params, m.sigma_net.params, m.color_net.params, m.latent_emb)]


@pytest.mark.parametrize('cfg', ['small', 'base19'])
def test_resume_continues_like_uninterrupted_run(tmp_path, cfg):
    """'base19': levels large enough for the binned table update (no dense table gradient exists in that mode)."""
    a = _make_trainer(cfg)
    assert bool(a.model.table_pipeline.binned_levels) == (cfg == 'base19')
    a.train_epoch(3)
    path = str(tmp_path / 'ck.pth')
    a.save_checkpoint(path)
    a.train_epoch(2)
    want = _params(a)

    b = _make_trainer(cfg)
    b.load_checkpoint(path)
    assert b.iter == 3
    b.dataloader.counter = b.iter
    assert abs(b.scheduler.get_last_lr()[0] - 2e-2 * 0.5 ** (3 / 4)) < 1e-9
    b.train_epoch(2)
    got = _params(b)
    for g, w, name in zip(got, want, ('table', 'sigma_net', 'color_net', 'latent')):
        # same data, same state: differences are atomic summation order, amplified by Adam on ~0 gradients
        assert float(((g - w).abs() > 1e-5).float().mean()) < 0.02, name
        torch.testing.assert_close(g, w, rtol=0, atol=2e-2, msg=lambda m, n=name: n + ': ' + m)
    st = b.optimizer.state[b.model.encoder.params]
    assert st['step'] == 5 and float(st['exp_avg_sq'].abs().sum()) > 0
