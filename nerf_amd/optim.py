"""Fused dense Adam (nerf_adam_step) behind torch.optim's param_groups interface, so the
reference's optimizer wiring (train.py:84-96: three groups, per-group lr / weight_decay,
LambdaLR on top) works unchanged.  Update rule = torch.optim.Adam's single-tensor path."""
import torch

from ._lib import check, lib, ptr, stream


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.grad_scale = 1.0
        self._grads_cleared = False

    def attach(self, model):
        """Give every parameter a persistent .grad and let the model's backward kernel accumulate into
        it directly; the step kernel then clears the gradient in the same pass (zero_grad is free)."""
        model.enable_direct_grads(True)
        return self

    def zero_grad(self, set_to_none=False):
        if self._grads_cleared:       # the last step() already cleared them in its own pass
            self._grads_cleared = False
            return
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is not None:
                    p.grad.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        for group in self.param_groups:
            b1, b2 = group['betas']
            for p in group['params']:
                if p.grad is None or p.numel() == 0:
                    continue
                st = self.state[p]
                if not st:
                    st['step'] = 0
                    st['exp_avg'] = torch.zeros_like(p)
                    st['exp_avg_sq'] = torch.zeros_like(p)
                st['step'] += 1
                check(lib().nerf_adam_step(ptr(p), ptr(p.grad), ptr(st['exp_avg']), ptr(st['exp_avg_sq']), p.numel(),
                                           group['lr'], b1, b2, group['eps'], group['weight_decay'], st['step'],
                                           self.grad_scale, 1, stream()), 'nerf_adam_step')
        self._grads_cleared = True
