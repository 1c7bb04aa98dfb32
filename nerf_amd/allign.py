"""Scene-consistency step of the reference's pose alignment (allign/trainer.py:174-208, allign/allign2.py:35-56,
allign/rotation.py:8-27): two trained fields A and B, a rigid transform T optimised so that the
density of B at T(x) agrees with the density of A along A's rays.

Only what sits on the render path is here -- the third caller of ``efficient_sampling`` + ``density`` -- not the
RANSAC / ICP initialisation (open3d) or the measurement scripts.  The gradient reaches T through
d(sigma_B)/d(position) (nerf_density_bwd_input); both fields stay frozen.  The [N,S,S] outer product of
trainer.py:203-206 is evaluated in its O(S) prefix-sum form (the fine depths are sorted).
"""
import torch
import torch.nn as nn


def vec2skew(v):
    """allign/rotation.py:8-15."""
    zero = torch.zeros(1, dtype=torch.float32, device=v.device)
    return torch.stack([torch.cat([zero, -v[2:3], v[1:2]]), torch.cat([v[2:3], zero, -v[0:1]]),
                        torch.cat([-v[1:2], v[0:1], zero])], dim=0)


def Exp(r):
    """so(3) vector -> SO(3) matrix (allign/rotation.py:18-27)."""
    skew_r = vec2skew(r)
    norm_r = r.norm() + 1e-15
    eye = torch.eye(3, dtype=torch.float32, device=r.device)
    return eye + (torch.sin(norm_r) / norm_r) * skew_r + ((1 - torch.cos(norm_r)) / norm_r ** 2) * (skew_r @ skew_r)


class Transform(nn.Module):
    """allign/allign2.py:35-56: x_b = [Exp(R) | T] @ init_transform @ x_a."""

    def __init__(self, init_transform):
        super().__init__()
        self.init_transform = nn.Parameter(init_transform.to(torch.float32), requires_grad=False)
        self.R = nn.Parameter(torch.zeros(3, device=init_transform.device), requires_grad=True)
        self.T = nn.Parameter(torch.zeros(3, device=init_transform.device), requires_grad=True)

    def forward(self, xyzs_a):
        b = xyzs_a.reshape(-1, 3)
        b = b @ self.init_transform[:3, :3].T + self.init_transform[:3, 3]
        b = b @ Exp(self.R).T + self.T
        return b.reshape(*xyzs_a.shape)


def contract(xyzs, inner_bound, outer_bound):
    """renderer.py:64-68 in differentiable torch ops."""
    r = torch.linalg.norm(xyzs, dim=-1, keepdim=True)
    far = r > inner_bound
    r_safe = torch.where(far, r, torch.ones_like(r))           # keeps 0/0 out of the unselected branch's gradient
    return torch.where(far, xyzs * ((outer_bound - (outer_bound - inner_bound) / r_safe) / r_safe), xyzs)


def _weights(sigmas, delta):
    """trainer.py:186-188 / :196-198 and the last-bin completion of :200-201."""
    alpha = 1 - torch.exp(-sigmas * delta)
    shift = torch.cat([alpha.new_zeros((alpha.shape[0], 1)), alpha], dim=-1)[:, :-1]
    w = alpha * torch.cumprod(1 - shift, dim=-1)
    return torch.cat([w[:, :-1], (1 - torch.sum(w[:, :-1], dim=-1))[:, None]], dim=-1)


def pair_distance_sorted(wa, wb, s):
    """mean_r sum_ij wa_i wb_j |s_i - s_j| for rows of s sorted ascending, without the [N,S,S] product."""
    cb, csb = torch.cumsum(wb, -1), torch.cumsum(wb * s, -1)
    below = s * (cb - wb) - (csb - wb * s)                          # sum_{j<i} wb_j (s_i - s_j)
    above = (csb[:, -1:] - csb) - s * (cb[:, -1:] - cb)            # sum_{j>i} wb_j (s_j - s_i)
    return torch.mean(torch.sum(wa * (below + above), dim=-1))


def scene_consistency(renderer, model_a, model_b, transform, h, w, K, E, noise=None):
    """allign/trainer.py:174-208.  Returns the scalar loss; differentiable w.r.t. ``transform``'s parameters."""
    rays_o, rays_d = renderer.get_rays(h, w, K, E)
    with torch.no_grad():
        saved = renderer.model
        renderer.model = model_a                      # the sampling pass of field A (trainer.py:177)
        try:
            z_vals_log, z_vals = renderer.efficient_sampling(rays_o, rays_d, renderer.steps_importance,
                                                             renderer.alpha_importance, noise=noise)
        finally:
            renderer.model = saved
        xyzs_a = rays_o[:, None, :] + rays_d[:, None, :] * z_vals[..., None]
        sigmas_a, _ = model_a.density(contract(xyzs_a, renderer.inner_bound, renderer.outer_bound), renderer.outer_bound)
        delta = torch.zeros_like(z_vals_log)
        delta[:, :-1] = z_vals_log[:, 1:] - z_vals_log[:, :-1]
        weights_a = _weights(sigmas_a, delta)
    xyzs_b = transform(xyzs_a)
    sigmas_b, _ = model_b.density(contract(xyzs_b, renderer.inner_bound, renderer.outer_bound), renderer.outer_bound)
    weights_ab = _weights(sigmas_a + sigmas_b, delta)
    return pair_distance_sorted(weights_a, weights_ab, z_vals_log)
