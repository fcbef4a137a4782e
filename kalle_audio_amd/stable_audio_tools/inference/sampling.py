"""Drop-in for stable_audio_tools/inference/sampling.py:8-86: get_alphas_sigmas, sample (v-diffusion DDIM),
sample_discrete_euler (rectified flow).  The k-diffusion samplers (111-193) are third-party and not carried over.
The per-step latent updates are [B, C, T]-sized host-orchestrated glue around the DiT forward, as in the reference."""
import math

import torch


def get_alphas_sigmas(t):
    return torch.cos(t * math.pi / 2), torch.sin(t * math.pi / 2)


def alpha_sigma_to_t(alpha, sigma):
    return torch.atan2(sigma, alpha) / math.pi * 2


def t_to_alpha_sigma(t):
    return torch.cos(t * math.pi / 2), torch.sin(t * math.pi / 2)


@torch.no_grad()
def sample_discrete_euler(model, x, steps, sigma_max=1, **extra_args):
    t = torch.linspace(sigma_max, 0, steps + 1)
    for t_curr, t_prev in zip(t[:-1], t[1:]):
        t_curr_tensor = t_curr * torch.ones((x.shape[0],), dtype=x.dtype, device=x.device)
        dt = t_prev - t_curr
        x = x + dt * model(x, t_curr_tensor, **extra_args)
    return x


@torch.no_grad()
def sample_rf(model_fn, noise, init_data=None, steps=100, sigma_max=1, device="cuda", callback=None, cond_fn=None,
              **extra_args):
    """sampling.py:200-232: discrete Euler for rectified-flow models; with init_data (a variation) the start point is the
    interpolation of the init latents and the noise at sigma_max"""
    if sigma_max > 1:
        sigma_max = 1
    if cond_fn is not None:
        raise NotImplementedError("sample_rf(cond_fn): guidance functions wrap third-party k-diffusion utilities")
    x = init_data * (1 - sigma_max) + noise * sigma_max if init_data is not None else noise
    return sample_discrete_euler(model_fn, x, steps, sigma_max, **extra_args)


@torch.no_grad()
def sample(model, x, steps, eta, **extra_args):
    ts = x.new_ones([x.shape[0]])
    t = torch.linspace(1, 0, steps + 1)[:-1]
    alphas, sigmas = get_alphas_sigmas(t)
    pred = x
    for i in range(steps):
        v = model(x, ts * t[i], **extra_args).float()
        pred = x * alphas[i] - v * sigmas[i]
        eps = x * sigmas[i] + v * alphas[i]
        if i < steps - 1:
            ddim_sigma = eta * (sigmas[i + 1] ** 2 / sigmas[i] ** 2).sqrt() * \
                (1 - alphas[i] ** 2 / alphas[i + 1] ** 2).sqrt()
            adjusted_sigma = (sigmas[i + 1] ** 2 - ddim_sigma ** 2).sqrt()
            x = pred * alphas[i + 1] + eps * adjusted_sigma
            if eta:
                x += torch.randn_like(x) * ddim_sigma
    return pred
