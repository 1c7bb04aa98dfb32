"""Losses and metrics with the reference's names (helpers.py:11-32), computed by one fused HIP
kernel (nerf_losses) that also produces the gradients.  ``criterion_dist`` is evaluated in its
O(S) prefix-sum form for sorted samples instead of the reference's [N,S,S] outer product."""
import math as m

import torch

from ._lib import check, f32, lib, ptr, stream


class _LossFn(torch.autograd.Function):
    """loss_rgb + lam_dist*loss_dist + lam_depth*loss_depth and its parts (trainer.py:153-169)."""

    @staticmethod
    def forward(ctx, image, gt, weights, zlog_s, lam_dist, lam_depth, sorted_s, want):
        img, g = f32(image), f32(gt)
        N = img.shape[0]
        use_w = weights is not None
        w = f32(weights) if use_w else None
        s = f32(zlog_s) if use_w else None
        S = w.shape[1] if use_w else 1
        sums = torch.empty(6, device=img.device, dtype=torch.float64)
        dimage = torch.empty_like(img)
        dweights = torch.empty_like(w) if use_w else None
        check(lib().nerf_losses(ptr(img), ptr(g), ptr(w), ptr(s), N, S, lam_dist, lam_depth, 1.0, 1 if sorted_s else 0,
                                ptr(sums), ptr(dimage), ptr(dweights), stream()), 'nerf_losses')
        ctx.save_for_backward(dimage, dweights)
        ctx.use_w = use_w
        ctx.want = want
        both = sums.to(torch.float32)
        parts = both[:4]
        out = both[{'rgb': 0, 'dist': 1, 'depth': 2, 'mse': 3, 'total': 4}[want]]
        flag = sums[5:6]            # fp64, != 0 iff the total loss is NaN / Inf (nerf_losses)
        ctx.mark_non_differentiable(parts, flag)
        ctx.set_materialize_grads(False)
        return out, parts, flag

    @staticmethod
    def backward(ctx, dout, _dparts, _dflag):
        dimage, dweights = ctx.saved_tensors
        if dout is None:
            return None, None, None, None, None, None, None, None
        gi = dimage * dout
        gw = dweights * dout if ctx.use_w else None
        return gi, None, gw, None, None, None, None, None


def fused_train_loss(image, gt, weights, zlog_s, lam_dist, lam_depth, sorted_s=True):
    """Returns (loss, parts[4] = loss_rgb, loss_dist, loss_depth, mse); one kernel, fwd + grads.
    ``loss.nonfinite_flag`` is a one-element fp64 device tensor, != 0 iff the loss is NaN / Inf: hand it to
    ``FusedAdam.skip_flag`` and the step is dropped on the device, without a host sync (the role GradScaler plays in
    the reference, trainer.py:134-136)."""
    loss, parts, flag = _LossFn.apply(image, gt, weights, zlog_s, float(lam_dist), float(lam_depth), sorted_s, 'total')
    loss.nonfinite_flag = flag
    return loss, parts


## Losses and Metrics
def psnr(rgb_, rgb):
    """helpers.py:11-13 (returns a Python float, like the reference)."""
    with torch.no_grad():
        _, parts, _ = _LossFn.apply(rgb_, rgb, None, None, 0.0, 0.0, True, 'rgb')
    return -10.0 * m.log10(float(parts[3]))


def mse2psnr(mse):
    """helpers.py:15-19."""
    if mse < 1e-5:
        mse = 1e-5
    return -10.0 * m.log10(mse)


def criterion_dist(weights, z_vals, assume_sorted=True):
    """helpers.py:21-27: mean_r sum_ij w_i w_j |s_i - s_j|.  ``assume_sorted`` (true for what
    ``render`` returns) selects the O(S) form; pass False for arbitrary sample order."""
    N = weights.shape[0]
    dummy = torch.zeros(N, 3, device=weights.device)
    out, _, _ = _LossFn.apply(dummy, dummy, weights, z_vals.detach(), 1.0, 0.0, assume_sorted, 'dist')
    return out


def criterion_rgb(rgb_, rgb):
    """helpers.py:29-32: Huber loss, delta = 0.1, mean reduction."""
    out, _, _ = _LossFn.apply(rgb_, rgb, None, None, 0.0, 0.0, True, 'rgb')
    return out
