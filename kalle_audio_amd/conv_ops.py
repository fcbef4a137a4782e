"""Tensor-level wrappers for the VAE conv kernels (csrc/conv1d.hip). GPU tensors only, no fallback."""
import ctypes
import os

import torch

from . import _lib
from ._lib import check
from .ops import _dt, _p, _stream


def weight_norm_fold(v, g, transposed=False):
    """v: Conv1d [Cout,Cin,K] (or ConvTranspose1d [Cin,Cout,K] when transposed), g: [dim0] or None.
    Returns the packed fp32 weight [Cin][K][roundup(Cout, 8)] (pad columns zero)."""
    lib = _lib.load()
    v = v.detach().float().contiguous()
    d0, d1, K = v.shape
    cin, cout = (d0, d1) if transposed else (d1, d0)
    gg = g.detach().float().contiguous().view(-1) if g is not None else None
    w = torch.empty((cin, K, (cout + 7) // 8 * 8), device=v.device, dtype=torch.float32)   # Cout padded to 8
    check(lib.kalle_weight_norm_fold(_p(v), _p(gg), _p(w), d0, d1, K, int(transposed), _stream()),
          "kalle_weight_norm_fold")
    return w


def _act_struct(code, alpha, beta, logscale, param):
    return _lib.Act(int(code), int(bool(logscale)), _p(alpha), _p(beta), float(param))


def _epilogue(residual, out_scale, accumulate, tanh, post_act, y_raw=None):
    """post_act: None or (code, alpha, beta, logscale, param) - the next layer's input activation, applied at the store;
    y_raw: optional tensor that also receives the un-activated value"""
    pa = _act_struct(*post_act) if post_act is not None else _lib.Act(0, 0, None, None, 0.0)
    return _lib.ConvEpilogue(_p(residual), float(out_scale), int(accumulate), int(tanh), pa, _p(y_raw))


def conv1d(x, w_packed, bias, *, Cout, K, stride=1, padding=0, dilation=1, act=0, alpha=None, beta=None,
           logscale=True, residual=None, post=0, out_dtype=None, pad_right=None, act_param=0.0, out_scale=1.0,
           accumulate_into=None, post_act=None, want_raw=False):
    """padding = left pad; pad_right defaults to the same (symmetric). act: 0 none, 1 snake(-beta), 2 ELU, 3 LeakyReLU,
    4 WaveNet gate.  Store: (conv + bias + residual) * out_scale (+= accumulate_into) -> post_act -> tanh (post=1)."""
    lib = _lib.load()
    x = x.contiguous()
    B, Cin, Lin = x.shape
    if act == 4:
        Cin //= 2
    pr = padding if pad_right is None else pad_right
    Lout = (Lin + padding + pr - dilation * (K - 1) - 1) // stride + 1
    if accumulate_into is not None:
        y = accumulate_into
        assert y.is_contiguous() and tuple(y.shape) == (B, Cout, Lout)
    else:
        y = torch.empty((B, Cout, Lout), device=x.device, dtype=out_dtype or x.dtype)
    if residual is not None:
        residual = residual.contiguous()
        assert residual.dtype == x.dtype and residual.shape == y.shape
    ia = _act_struct(act, alpha, beta, logscale, act_param)
    y_raw = torch.empty_like(y) if want_raw else None      # dual output: (post_act(y), y)
    ep = _epilogue(residual, out_scale, accumulate_into is not None, post & 1, post_act, y_raw)
    # few positions, many channels (the top of the VAE, all of a single-clip decode): channels-per-lane kernel over a padded,
    # pre-activated copy of x.  Chosen when the position-per-lane tiling would leave most CUs without a workgroup.
    want = os.environ.get("KALLE_CONV_CFIRST")
    nwg64 = ((Lout + 511) // 512) * ((Cout + 63) // 64) * B
    if stride > 1:
        # strided convs (the encoder's down-samplers): the position-per-lane kernels reach 16-25 TFLOP/s at stride 8 and 38-46 at
        # stride 4, the channels-per-lane kernel 43-57 at every batch size measured (512 -> 1024, k = 16, stride 8, 13760 inputs,
        # B = 4: 5388 against 2379 us; 1024 -> 2048 x 1720, B = 8: 7237 against 2166; 256 -> 512, k = 8, stride 4, B = 8: 5021 against 3876)
        small = Cout >= 256
    elif Cout >= 256:
        # The position-per-lane kernel runs these layers in 8-wave workgroups of 128 channels x 512 positions (256 for k = 1): one
        # round of them takes the same time whether 100 or 256 exist, while the channels-per-lane kernel scales with the work
        # (tools/cfirst_vs_v2.sh, 1024 channels x 1720 positions, k = 7: B = 4 -> 128 workgroups 2316 us against 1820; B = 5 -> 160
        # workgroups 2262 against 2560; 512 channels x 13760, k = 1, B = 1 -> 216 workgroups 162 us against 222)
        small = -(-Lout // (256 if K == 1 else 512)) * -(-Cout // 128) * B < 160
    else:
        # few output channels: while positions are few too - or the reduction is long and the position-per-lane grid a handful of
        # workgroups that each walk all of it (2048 -> 128, k = 3, 215 positions, B = 16: 1089 against 222 us)
        small = Cout >= 64 and nwg64 < 256 and (Lout * B <= 1024 or (Cin >= 1024 and nwg64 <= 64))
    if ((stride == 1 or dilation == 1) and act != 4 and x.dtype == torch.float32 and y.dtype == torch.float32
            and (want == "1" or (want is None and small))):
        Lp = lib.kalle_conv_pad_len(Lout, K, stride, padding, dilation)
        lead = padding if stride == 1 else (padding + stride - 1) // stride * stride
        xp = torch.empty((B, Cin, Lp), device=x.device, dtype=torch.float32)
        check(lib.kalle_conv_pad_act(_p(x), _p(xp), B, Cin, Lin, Lp, lead, ctypes.addressof(ia), stride, _stream()),
              "kalle_conv_pad_act")
        nws = lib.kalle_conv_cfirst_ws_floats(B, Cin, Cout, Lout, K)    # > 0: input channels split over workgroups too
        ws = torch.empty(nws, device=x.device, dtype=torch.float32) if nws > 0 else None
        check(lib.kalle_conv1d_cfirst_fwd(_p(xp), _p(w_packed), _p(bias), _p(y), B, Cin, Lp, Cout, Lout, K, stride, padding,
                                          dilation, ctypes.addressof(ep), _p(ws), _stream()), "kalle_conv1d_cfirst_fwd")
        return (y, y_raw) if want_raw else y
    check(lib.kalle_conv1d_fwd(_p(x), _dt(x), _p(w_packed), _p(bias), _p(y), _dt(y), B, Cin, Lin, Cout, Lout, K, stride,
                               padding, dilation, ctypes.addressof(ia), ctypes.addressof(ep), _stream()),
          "kalle_conv1d_fwd")
    return (y, y_raw) if want_raw else y


def conv_transpose1d(x, w_packed, bias, *, Cout, K, stride, padding, act=0, alpha=None, beta=None, logscale=True,
                     out_dtype=None, trim=0, act_param=0.0, post_act=None, want_raw=False):
    """trim: drop the last `trim` outputs (causal transposed conv); negative: keep up to `padding` outputs beyond the symmetric
    right trim (data gradient of a strided conv)"""
    lib = _lib.load()
    x = x.contiguous()
    B, Cin, Lin = x.shape
    Lout = (Lin - 1) * stride - 2 * padding + K - trim
    y = torch.empty((B, Cout, Lout), device=x.device, dtype=out_dtype or x.dtype)
    ia = _act_struct(act, alpha, beta, logscale, act_param)
    y_raw = torch.empty_like(y) if want_raw else None
    ep = _epilogue(None, 1.0, False, False, post_act, y_raw)
    want = os.environ.get("KALLE_CONV_CFIRST")
    nq = (Lout - 1 + padding) // stride + 1
    # (tools/cfirst_vs_v2.sh: at 512+ output channels the channels-per-lane kernel holds 50-70 TFLOP/s where the phase-per-workgroup
    # kernel needs far more positions to get there - 2048 -> 1024 x 215, B = 8: 2082 against 3463 us; 1024 -> 512 x 1720, B = 8: 3290
    # against 3878; at 256 channels the two cross near 1500: 512 -> 256 x 13760, B = 3: 1465 against 1551, B = 4: 1870 against 1806)
    small = ((nq + 511) // 512) * ((Cout + 63) // 64) * B * stride < (4096 if Cout >= 512 else 1536) and Cout >= 256
    if (trim >= 0 and act != 4 and x.dtype == torch.float32 and y.dtype == torch.float32
            and (want == "1" or (want is None and small))):
        Lp = lib.kalle_convT_pad_len(Lout, K, stride, padding)
        xp = torch.empty((B, Cin, Lp), device=x.device, dtype=torch.float32)
        check(lib.kalle_conv_pad_act(_p(x), _p(xp), B, Cin, Lin, Lp, (K + stride - 1) // stride - 1, ctypes.addressof(ia), 1,
                                     _stream()), "kalle_conv_pad_act")
        check(lib.kalle_conv_transpose1d_cfirst_fwd(_p(xp), _p(w_packed), _p(bias), _p(y), B, Cin, Lp, Cout, Lout, K, stride,
                                                    padding, ctypes.addressof(ep), _stream()),
              "kalle_conv_transpose1d_cfirst_fwd")
        return (y, y_raw) if want_raw else y
    check(lib.kalle_conv_transpose1d_fwd(_p(x), _dt(x), _p(w_packed), _p(bias), _p(y), _dt(y), B, Cin, Lin, Cout,
                                         Lout, K, stride, padding, ctypes.addressof(ia), ctypes.addressof(ep),
                                         _stream()), "kalle_conv_transpose1d_fwd")
    return (y, y_raw) if want_raw else y


def snake_beta(x, alpha, beta, logscale=True):
    lib = _lib.load()
    x = x.contiguous()
    B, C, L = x.shape
    y = torch.empty_like(x)
    check(lib.kalle_snake_beta_fwd(_p(x), _p(y), _dt(x), _p(alpha), _p(beta), int(logscale), B, C, L, _stream()),
          "kalle_snake_beta_fwd")
    return y


def activate(x, act, alpha=None, beta=None, logscale=True):
    """act(x) as a tensor of its own (the VAE's conv kernels apply their input activation on the fly; the weight gradient of a
    transposed conv wants the activated input as its scalar operand): codes 0 none / 1 SnakeBeta / 2 ELU, x fp32 [B, C, L]"""
    if not act:
        return x
    if act == 1:
        return snake_beta(x.contiguous(), alpha, beta, logscale)
    lib = _lib.load()
    x = x.contiguous()
    B, C, L = x.shape
    y = torch.empty_like(x)
    ia = _act_struct(act, alpha, beta, logscale, 0.0)
    check(lib.kalle_conv_pad_act(_p(x), _p(y), B, C, L, L, 0, ctypes.addressof(ia), 1, _stream()), "kalle_conv_pad_act")
    return y


def kaiser_sinc_filter12(device, cutoff=0.25, half_width=0.3, kernel_size=12):
    """the 12-tap low-pass of alias-free-torch's UpSample1d / DownSample1d at ratio 2 (cutoff 0.5/2, half-width 0.6/2):
    kaiser window (beta from the attenuation implied by the transition width) x sinc, normalised to unit sum."""
    import math
    half = kernel_size // 2
    delta_f = 4 * half_width
    A = 2.285 * (half - 1) * math.pi * delta_f + 7.95
    if A > 50.0:
        kb = 0.1102 * (A - 8.7)
    elif A >= 21.0:
        kb = 0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21.0)
    else:
        kb = 0.0
    window = torch.kaiser_window(kernel_size, beta=kb, periodic=False, dtype=torch.float64)
    time = torch.arange(-half, half, dtype=torch.float64) + 0.5
    filt = 2 * cutoff * window * torch.sinc(2 * cutoff * time)
    filt = filt / filt.sum()
    return filt.to(device=device, dtype=torch.float32).contiguous()


def act1d(x, filt, alpha, beta, logscale):
    """anti-aliased snake(-beta): up 2x FIR -> activation -> down 2x FIR, one fused kernel"""
    lib = _lib.load()
    x = x.contiguous()
    B, C, L = x.shape
    y = torch.empty_like(x)
    check(lib.kalle_act1d_fwd(_p(x), _p(y), _dt(x), _p(filt), _p(alpha), _p(beta), int(logscale), B, C, L, _stream()),
          "kalle_act1d_fwd")
    return y
