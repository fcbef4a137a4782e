"""Drop-in for the parts of stable_audio_tools/models/blocks.py on the hot path: FourierFeatures (84-93),
rms_norm / RMSNorm / AdaRMSNorm (203-221, 268-299), snake_beta / SnakeBeta (301-339).  HIP kernels only."""
import torch
from torch import nn

from ... import conv_ops
from ... import functional as KF


class FourierFeatures(nn.Module):
    """blocks.py:84-93"""

    def __init__(self, in_features, out_features, std=1.):
        super().__init__()
        assert out_features % 2 == 0
        if in_features != 1:
            raise NotImplementedError("FourierFeatures kernel handles the timestep case in_features == 1 (dit.py:37)")
        self.weight = nn.Parameter(torch.randn([out_features // 2, in_features]) * std)

    def forward(self, input):
        return KF.FourierFeaturesFn.apply(input, self.weight)


def rms_norm(x, scale, eps):
    """blocks.py:268-272 (scale: [D] or per-batch [B, D] broadcast over tokens is handled by AdaRMSNorm)"""
    return KF.RMSNormFn.apply(x, scale, eps)


class RMSNorm(nn.Module):
    """blocks.py:285-299"""

    def __init__(self, shape, fix_scale=False, eps=1e-6):
        super().__init__()
        self.eps = eps
        if fix_scale:
            self.register_buffer("scale", torch.ones(shape))
        else:
            self.scale = nn.Parameter(torch.ones(shape))

    def extra_repr(self):
        return f"shape={tuple(self.scale.shape)}, eps={self.eps}"

    def forward(self, x):
        return rms_norm(x, self.scale, self.eps)


def snake_beta(x, alpha, beta):
    """blocks.py:301-302 with alpha/beta already exponentiated ([1,C,1] or [C])"""
    return conv_ops.snake_beta(x, alpha.reshape(-1).float().contiguous(), beta.reshape(-1).float().contiguous(),
                               logscale=False)


class SnakeBeta(nn.Module):
    """blocks.py:306-339 (inference-only here: the VAE is frozen in every reference script, factory.py:77-80)"""

    def __init__(self, in_features, alpha=1.0, alpha_trainable=True, alpha_logscale=True):
        super().__init__()
        self.in_features = in_features
        self.alpha_logscale = alpha_logscale
        if self.alpha_logscale:
            self.alpha = nn.Parameter(torch.zeros(in_features) * alpha)
            self.beta = nn.Parameter(torch.zeros(in_features) * alpha)
        else:
            self.alpha = nn.Parameter(torch.ones(in_features) * alpha)
            self.beta = nn.Parameter(torch.ones(in_features) * alpha)
        self.alpha.requires_grad = alpha_trainable
        self.beta.requires_grad = alpha_trainable
        self.no_div_by_zero = 0.000000001

    def forward(self, x):
        if torch.is_grad_enabled() and (x.requires_grad or self.alpha.requires_grad and self.training):
            pass  # forward-only kernel; gradients do not flow (VAE frozen)
        x = x if x.dtype in (torch.float32, torch.bfloat16) else x.float()
        return conv_ops.snake_beta(x, self.alpha.detach().float(), self.beta.detach().float(),
                                   logscale=self.alpha_logscale)
