"""Tensor-level wrappers for the VAE conv kernels (csrc/conv1d.hip). GPU tensors only, no fallback."""
import ctypes

import torch

from . import _lib
from ._lib import check
from .ops import _dt, _p, _stream


def weight_norm_fold(v, g, transposed=False):
    """v: Conv1d [Cout,Cin,K] (or ConvTranspose1d [Cin,Cout,K] when transposed), g: [dim0] or None.
    Returns the packed fp32 weight [Cin][K][Cout]."""
    lib = _lib.load()
    v = v.detach().float().contiguous()
    d0, d1, K = v.shape
    cin, cout = (d0, d1) if transposed else (d1, d0)
    gg = g.detach().float().contiguous().view(-1) if g is not None else None
    w = torch.empty((cin, K, cout), device=v.device, dtype=torch.float32)
    check(lib.kalle_weight_norm_fold(_p(v), _p(gg), _p(w), d0, d1, K, int(transposed), _stream()),
          "kalle_weight_norm_fold")
    return w


def conv1d(x, w_packed, bias, *, Cout, K, stride=1, padding=0, dilation=1, act=0, alpha=None, beta=None,
           logscale=True, residual=None, post=0, out_dtype=None):
    lib = _lib.load()
    x = x.contiguous()
    B, Cin, Lin = x.shape
    Lout = (Lin + 2 * padding - dilation * (K - 1) - 1) // stride + 1
    y = torch.empty((B, Cout, Lout), device=x.device, dtype=out_dtype or x.dtype)
    if residual is not None:
        residual = residual.contiguous()
        assert residual.dtype == x.dtype and residual.shape == y.shape
    check(lib.kalle_conv1d_fwd(_p(x), _dt(x), _p(w_packed), _p(bias), _p(residual), _p(y), _dt(y), B, Cin, Lin, Cout,
                               Lout, K, stride, padding, dilation, act, _p(alpha), _p(beta), int(logscale), post,
                               _stream()), "kalle_conv1d_fwd")
    return y


def conv_transpose1d(x, w_packed, bias, *, Cout, K, stride, padding, act=0, alpha=None, beta=None, logscale=True,
                     out_dtype=None):
    lib = _lib.load()
    x = x.contiguous()
    B, Cin, Lin = x.shape
    Lout = (Lin - 1) * stride - 2 * padding + K
    y = torch.empty((B, Cout, Lout), device=x.device, dtype=out_dtype or x.dtype)
    check(lib.kalle_conv_transpose1d_fwd(_p(x), _dt(x), _p(w_packed), _p(bias), _p(y), _dt(y), B, Cin, Lin, Cout,
                                         Lout, K, stride, padding, act, _p(alpha), _p(beta), int(logscale),
                                         _stream()), "kalle_conv_transpose1d_fwd")
    return y


def snake_beta(x, alpha, beta, logscale=True):
    lib = _lib.load()
    x = x.contiguous()
    B, C, L = x.shape
    y = torch.empty_like(x)
    check(lib.kalle_snake_beta_fwd(_p(x), _p(y), _dt(x), _p(alpha), _p(beta), int(logscale), B, C, L, _stream()),
          "kalle_snake_beta_fwd")
    return y
