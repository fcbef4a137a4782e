"""Autograd for the VAE conv stacks: one node per fused forward unit  y = conv(act(x)) (+ residual) (-> tanh).

Used when the pretransform is trained (`enable_grad`, stable_audio_tools/models/factory.py:77-80 - the reference's scripts keep
the VAE frozen, but its training wrapper supports it, training/diffusion.py:343-346).  Forward = the inference kernels of
csrc/conv1d.hip (activation applied while the conv stages its input, residual and tanh at the store); backward = the kernels
of csrc/conv1d_bwd.hip plus the forward kernels over dy with re-packed weights (see include/kalle_hip.h).  fp32 throughout.
"""
import ctypes

import torch

from . import _lib, conv_ops
from ._lib import check
from .ops import _p, _stream

F32 = torch.float32


def _act(code, alpha, beta, logscale):
    return _lib.Act(int(code), int(bool(logscale)), _p(alpha), _p(beta), 0.0)


def conv_wgrad(U, V, dW, *, K, stride, padding, dilation, act_on, act=0, alpha=None, beta=None, logscale=True):
    """dW[cu, cv, k] += sum_{b, m} U[b, cu, m] * V[b, cv, m*stride - padding + k*dilation]; `act` on V (act_on 0) or U (1)"""
    lib = _lib.load()
    U, V = U.contiguous(), V.contiguous()
    assert U.dtype == F32 and V.dtype == F32 and dW.dtype == F32 and dW.is_contiguous()
    B, CU, MU = U.shape
    _, CV, LV = V.shape
    a = _act(act, alpha, beta, logscale)
    check(lib.kalle_conv_wgrad(_p(U), _p(V), _p(dW), B, CU, CV, MU, LV, K, stride, padding, dilation, act_on,
                               ctypes.addressof(a), _stream()), "kalle_conv_wgrad")
    return dW


def act_bwd(x, g, act, alpha=None, beta=None, logscale=True, want_params=True):
    """returns (dx, dalpha, dbeta) for y = act(x), upstream g"""
    lib = _lib.load()
    x, g = x.contiguous(), g.contiguous()
    B, C, L = x.shape
    dx = torch.empty_like(x)
    da = db = None
    if act == 1 and want_params:
        da = torch.zeros(C, device=x.device, dtype=F32)
        db = torch.zeros(C, device=x.device, dtype=F32)
    a = _act(act, alpha, beta, logscale)
    check(lib.kalle_act_bwd(_p(x), _p(g), _p(dx), ctypes.addressof(a), _p(da), _p(db), B, C, L, _stream()), "kalle_act_bwd")
    return dx, da, db


def tanh_bwd(dy, y):
    lib = _lib.load()
    g = torch.empty_like(dy)
    check(lib.kalle_tanh_bwd(_p(dy), _p(y), _p(g), dy.numel(), _stream()), "kalle_tanh_bwd")
    return g


def upsample_nearest(x, scale, backward=False):
    """nn.Upsample(scale_factor=scale, mode='nearest') on [B, C, L] fp32, or (backward) its adjoint on [B, C, L*scale]"""
    lib = _lib.load()
    x = x.contiguous().float()
    B, C, L = x.shape
    if backward:
        assert L % scale == 0
        L //= scale
    y = torch.empty((B, C, L if backward else L * scale), device=x.device, dtype=F32)
    check(lib.kalle_upsample_nearest(_p(x), _p(y), B * C, L, int(scale), int(backward), _stream()), "kalle_upsample_nearest")
    return y


class UpsampleNearestFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale):
        ctx.scale = scale
        return upsample_nearest(x, scale)

    @staticmethod
    def backward(ctx, dy):
        return upsample_nearest(dy, ctx.scale, backward=True), None


def channel_sum(x):
    lib = _lib.load()
    B, C, L = x.shape
    out = torch.zeros(C, device=x.device, dtype=F32)
    check(lib.kalle_channel_sum(_p(x), _p(out), B, C, L, _stream()), "kalle_channel_sum")
    return out


def weight_norm_bwd(dw, v, g):
    lib = _lib.load()
    v = v.detach().float().contiguous()
    gg = g.detach().float().contiguous().view(-1)
    dv = torch.empty_like(v)
    dg = torch.empty_like(gg)
    check(lib.kalle_weight_norm_bwd(_p(dw), _p(v), _p(gg), _p(dv), _p(dg), v.shape[0], v[0].numel(), 0, _stream()),
          "kalle_weight_norm_bwd")
    return dv, dg.view(g.shape)


def _fold(v, g, flags):
    lib = _lib.load()
    v = v.detach().float().contiguous()
    d0, d1, K = v.shape
    cin, cout = (d0, d1) if flags & 1 else (d1, d0)
    gg = g.detach().float().contiguous().view(-1)
    w = torch.empty((cin, K, (cout + 7) // 8 * 8), device=v.device, dtype=F32)
    check(lib.kalle_weight_norm_fold(_p(v), _p(gg), _p(w), d0, d1, K, flags, _stream()), "kalle_weight_norm_fold")
    return w


class ActConvFn(torch.autograd.Function):
    """y = conv(act(x)) (+ residual) (-> tanh);  kind 'conv' (WNConv1d: v [Cout, Cin, K]) or 'convT' (WNConvTranspose1d:
    v [Cin, Cout, K]).  act: 0 none, 1 SnakeBeta(alpha, beta), 2 ELU."""

    @staticmethod
    def forward(ctx, x, v, g, bias, alpha, beta, residual, cfg):
        x = x.contiguous().float()
        kind, K, stride, pad, dil = cfg["kind"], cfg["K"], cfg["stride"], cfg["padding"], cfg["dilation"]
        act, ls, tanh = cfg["act"], cfg["logscale"], cfg["tanh"]
        pr = cfg.get("pad_right")       # 'same' with an even kernel: one more zero on the right than on the left
        b32 = bias.detach().float() if bias is not None else None
        a32 = alpha.detach().float() if alpha is not None else None
        be32 = beta.detach().float() if beta is not None else None
        if kind == "conv":
            y = conv_ops.conv1d(x, _fold(v, g, 0), b32, Cout=v.shape[0], K=K, stride=stride, padding=pad, dilation=dil, act=act,
                                alpha=a32, beta=be32, logscale=ls, pad_right=pr,
                                residual=residual.contiguous().float() if residual is not None else None, post=int(tanh))
        else:
            assert residual is None and not tanh
            y = conv_ops.conv_transpose1d(x, _fold(v, g, 1), b32, Cout=v.shape[1], K=K, stride=stride, padding=pad, act=act,
                                          alpha=a32, beta=be32, logscale=ls)
        ctx.cfg = cfg
        ctx.has = (bias is not None, residual is not None)
        ctx.save_for_backward(x, v, g, alpha, beta, y if tanh else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, v, g, alpha, beta, y = ctx.saved_tensors
        cfg = ctx.cfg
        kind, K, stride, pad, dil = cfg["kind"], cfg["K"], cfg["stride"], cfg["padding"], cfg["dilation"]
        act, ls = cfg["act"], cfg["logscale"]
        has_bias, has_res = ctx.has
        gy = dy.contiguous().float()
        if cfg["tanh"]:
            gy = tanh_bwd(gy, y)
        B, Cx, Lx = x.shape
        a32 = alpha.detach().float() if alpha is not None else None
        be32 = beta.detach().float() if beta is not None else None
        # ---- gradient w.r.t. the activated input: a forward kernel over gy with re-packed weights
        dxa = None
        if ctx.needs_input_grad[0] or (act == 1 and (ctx.needs_input_grad[4] or ctx.needs_input_grad[5])):
            if kind == "conv" and stride == 1:
                pr = cfg.get("pad_right")
                dxa = conv_ops.conv1d(gy, _fold(v, g, 1 | 2), None, Cout=Cx, K=K, stride=1, padding=(K - 1) * dil - pad,
                                      dilation=dil, pad_right=None if pr is None else (K - 1) * dil - pr)
            elif kind == "conv":
                # transposed conv over gy; the symmetric right trim of `pad` outputs is given back as far as x reaches
                nat = (gy.shape[2] - 1) * stride - 2 * pad + K
                extra = max(0, min(Lx - nat, pad))
                dxa = conv_ops.conv_transpose1d(gy, _fold(v, g, 1), None, Cout=Cx, K=K, stride=stride, padding=pad, trim=-extra)
                if dxa.shape[2] < Lx:       # trailing inputs the strided conv never read: zero gradient
                    dxa = torch.nn.functional.pad(dxa, (0, Lx - dxa.shape[2]))
                elif dxa.shape[2] > Lx:
                    dxa = dxa[:, :, :Lx].contiguous()
            else:
                dxa = conv_ops.conv1d(gy, _fold(v, g, 0), None, Cout=Cx, K=K, stride=stride, padding=pad, dilation=1)
            assert dxa.shape == x.shape, (dxa.shape, x.shape)
        # ---- weight gradient (module layout) and weight-norm backward
        dv = dg = None
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dw = torch.zeros(v.shape, device=x.device, dtype=F32)
            if kind == "conv":
                conv_wgrad(gy, x, dw, K=K, stride=stride, padding=pad, dilation=dil, act_on=0, act=act, alpha=a32, beta=be32,
                           logscale=ls)
            else:
                # (the activated input is the SCALAR operand of the LDS-staged kernel: materialised once instead of re-applying
                # the activation at every use)
                xa = conv_ops.activate(x, act, a32, be32, ls)
                conv_wgrad(xa, gy, dw, K=K, stride=stride, padding=pad, dilation=1, act_on=1, act=0)
            dv, dg = weight_norm_bwd(dw, v, g)
        db = channel_sum(gy) if has_bias and ctx.needs_input_grad[3] else None
        # ---- through the input activation
        dx = dalpha = dbeta = None
        if dxa is not None:
            if act:
                dx, dalpha, dbeta = act_bwd(x, dxa, act, a32, be32, ls)
                if dalpha is not None:
                    dalpha, dbeta = dalpha.view(alpha.shape), dbeta.view(beta.shape)
            else:
                dx = dxa
        return dx, dv, dg, db, dalpha, dbeta, (gy if has_res else None), None


def act_conv(x, conv, act_module=None, residual=None, tanh=False):
    """`conv`: a WNConv1d / WNConvTranspose1d module of stable_audio_tools/models/autoencoders.py; act_module: its input
    activation (SnakeBeta / nn.ELU / None)"""
    from torch import nn
    from .stable_audio_tools.models.blocks import SnakeBeta
    code, alpha, beta, ls = 0, None, None, True
    if isinstance(act_module, SnakeBeta):
        code, alpha, beta, ls = 1, act_module.alpha, act_module.beta, act_module.alpha_logscale
    elif isinstance(act_module, nn.ELU):
        code = 2
    elif act_module is not None and not isinstance(act_module, nn.Identity):
        raise NotImplementedError(f"activation {type(act_module).__name__}")
    cfg = dict(kind="convT" if conv.transposed else "conv", K=conv.kernel_size, stride=conv.stride, padding=conv.padding,
               dilation=getattr(conv, "dilation", 1), act=code, logscale=ls, tanh=bool(tanh),
               pad_right=getattr(conv, "pad_right", None))
    return ActConvFn.apply(x, conv.weight_v, conv.weight_g, conv.bias, alpha, beta, residual, cfg)
