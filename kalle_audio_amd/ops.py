"""Thin tensor-level wrappers over the C-ABI (kalle_audio_amd/_lib.py).

Every function here takes CUDA (HIP) tensors, launches the hand-written gfx950 kernel on torch's current
stream and returns immediately.  Nothing here falls back to torch math: CPU tensors raise.
PyTorch is used only for device memory and streams.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import KALLE_BF16, KALLE_F32, GemmEpilogue, check


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dt(t):
    if t.dtype == torch.bfloat16:
        return KALLE_BF16
    if t.dtype == torch.float32:
        return KALLE_F32
    raise TypeError(f"unsupported dtype {t.dtype}")


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("kalle_audio_amd ops run on the GPU only (HIP kernels); got a CPU tensor")
    return ctypes.c_void_p(t.data_ptr())


def _rows2d(t):
    """(rows, ld) of a tensor viewed as a row-major matrix whose last dim is contiguous."""
    assert t.stride(-1) == 1
    return t.numel() // t.shape[-1], t.shape[-1]


# ------------------------------------------------------------------------------------------------ GEMM
def gemm(a, b, *, a_kmajor=False, b_kmajor=False, out=None, out_dtype=torch.bfloat16, M=None, N=None, K=None,
         lda=None, ldb=None, ldc=None, bias=None, gate=None, rows_per_batch=0, residual=None, accumulate=False,
         alpha=1.0, c_rows_per_batch=0, c_batch_rows=0, c_row_offset=0, row_mask=None, glu_mode=0, glu_inner=0,
         glu_aux=None, glu_dbias=None):
    """C = op(a) @ op(b) with the fused epilogue of kalle_gemm_bf16.

    a: [M,K] (or [K,M] when a_kmajor), b: [N,K] (or [K,N] when b_kmajor); both bf16, last dim contiguous.
    """
    lib = _lib.load()
    assert a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16
    ar, ac = _rows2d(a)
    br, bc = _rows2d(b)
    if (ar == 1 and not a_kmajor and not b_kmajor and out is None and (bias is None or residual is None)
            and (bias is None or (bias.dtype == torch.float32 and bias.is_contiguous())) and gate is None and not glu_mode
            and row_mask is None and not accumulate and alpha == 1.0 and M is None and N is None and K is None
            and ac % 8 == 0 and ac <= 32768 and (residual is None or residual.is_contiguous())):
        # single row (decoding against the KV cache): weight-streaming kernel instead of a 128-row MFMA tile
        y = torch.empty((1, br), device=a.device, dtype=out_dtype)
        check(lib.kalle_gemv_bf16(_p(a), _p(b), b.stride(-2) if b.dim() >= 2 else bc, _p(y), _dt(y),
                                  _p(residual if residual is not None else bias), br, ac,
                                  _stream()), "kalle_gemv_bf16")
        return y
    if M is None:
        M = ac if a_kmajor else ar
    if K is None:
        K = ar if a_kmajor else ac
    if N is None:
        N = bc if b_kmajor else br
    lda = lda if lda is not None else a.stride(-2) if a.dim() >= 2 else ac
    ldb = ldb if ldb is not None else b.stride(-2) if b.dim() >= 2 else bc
    if out is None:
        assert c_rows_per_batch == 0
        out = torch.empty((M, N), device=a.device, dtype=out_dtype)
    ldc = ldc if ldc is not None else out.stride(-2)
    ep = GemmEpilogue()
    ep.bias = bias.data_ptr() if bias is not None else None
    ep.gate = gate.data_ptr() if gate is not None else None
    ep.ldg = gate.stride(-2) if gate is not None else 0
    ep.rows_per_batch = rows_per_batch
    ep.residual = residual.data_ptr() if residual is not None else None
    ep.ldr = residual.stride(-2) if residual is not None else 0
    ep.accumulate = 1 if accumulate else 0
    ep.alpha = alpha
    ep.c_rows_per_batch, ep.c_batch_rows, ep.c_row_offset = c_rows_per_batch, c_batch_rows, c_row_offset
    ep.row_mask = row_mask.data_ptr() if row_mask is not None else None
    ep.glu_mode, ep.glu_inner = glu_mode, glu_inner
    ep.glu_aux = glu_aux.data_ptr() if glu_aux is not None else None
    ep.glu_dbias = glu_dbias.data_ptr() if glu_dbias is not None else None
    if M <= FEW_ROWS_MAX and not a_kmajor and FEW_ROWS:
        # few output rows (sampling, small training batches): lend the kernel an fp32 [M][N] scratch so it may split K
        ws = _workspace(a.device, 4 * M * N * 8)
        ep.workspace, ep.workspace_bytes = ws.data_ptr(), ws.numel()
    if bias is not None:
        assert bias.dtype == torch.float32
    if gate is not None:
        assert gate.dtype == torch.float32
    if residual is not None:
        assert residual.dtype == torch.float32
    prof = KERNEL_TIMER if KERNEL_TIMER_ONLY is None else None     # (a filter names the grouped weight-gradient launch only)
    if prof is not None:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = lib.kalle_gemm_bf16(_p(a), lda, int(a_kmajor), _p(b), ldb, int(b_kmajor), _p(out), ldc, _dt(out),
                             M, N, K, ctypes.byref(ep), _stream())
    if rc == -3 and glu_mode:
        return None          # fused SwiGLU not available for this shape: the caller runs the unfused sequence
    check(rc, "kalle_gemm_bf16")
    if prof is not None:
        e1.record()
        plan = lib.kalle_gemm_last_plan()
        kname = {1: "gemm_bf16_kernel", 2: "gemm2_kernel", 3: "gemm3_kernel", 4: "gemm2_splitk+finish",
                 5: "gemm2_ks2_kernel"}.get(plan & 255, "gemm")
        variant = "%s<%d,%d,%d%s>" % (kname, int(a_kmajor), int(b_kmajor), int(out.dtype == torch.float32),
                                      ",glu%d" % glu_mode if glu_mode else "")      # (the fused-SwiGLU kernels are their own rows)
        abytes = 2.0 * (M * K + N * K) + M * N * out.element_size() + (4.0 * M * N if residual is not None else 0.0)
        prof.append((variant, 2.0 * M * N * K, e0, e1, abytes, (M, N, K)))
    return out


FEW_ROWS = os.environ.get("KALLE_GEMM_FEW_ROWS", "1") != "0"
FEW_ROWS_MAX = int(os.environ.get("KALLE_GEMM_FEW_ROWS_MAX", "4096"))  # (the dispatcher decides per shape)
_WS = {}
_WS_KEEP = []


def _workspace(device, nbytes):
    """per (device, stream) scratch for kalle_gemm_bf16's few-rows path, grown on demand (never shrinks)"""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    t = _WS.get(key)
    # (capped at 1 GiB - the kernel is told the size and splits K only as far as the scratch reaches.  The comparison must use
    # the CAPPED size: comparing with the uncapped request re-allocated - and kept - a fresh GiB on every call once a shape asked
    # for more than the cap, 4032 rows x 12288 columns at B = 32 per GPU: 11 GB per step until the device was full)
    need = max(min(nbytes, 1 << 30), 64 << 20)
    if t is None or t.numel() < need:
        if t is not None:
            _WS_KEEP.append(t)          # a captured HIP graph may still point at the old scratch: never hand it back
        t = _WS[key] = torch.empty(need, device=device, dtype=torch.uint8)
    return t


# bench.py sets this to a list to collect (kernel variant, algorithmic flops, start event, end event) per GEMM launch:
# HIP events recorded on the launch stream, read back after the timed region (no host sync while timing).
KERNEL_TIMER = None
# Two event records per launch are not free (~4.5 us each on this stack: 300 GEMMs per step = 2.7 ms of a 222-ms step), so the timed
# region of bench.py times ONLY the kernel named here (the dominant one: its roofline figure must come from the timed region) and
# collects the table of all variants in one extra step afterwards.  None: every launch.
KERNEL_TIMER_ONLY = None


# ------------------------------------------------------------------------------------------------ norms
def layernorm_fwd(x, gamma, beta=None, scale=None, shift=None, rows_per_batch=0, eps=1e-5):
    lib = _lib.load()
    rows, D = _rows2d(x)
    x = x.contiguous()
    y = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    mean = torch.empty(rows, device=x.device, dtype=torch.float32)
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    ld_mod = scale.stride(-2) if scale is not None else (shift.stride(-2) if shift is not None else 0)
    check(lib.kalle_layernorm_fwd(_p(x), _dt(x), _p(gamma), _p(beta), _p(scale), _p(shift), ld_mod, rows_per_batch,
                                  _p(y), _p(mean), _p(rstd), rows, D, eps, _stream()), "kalle_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, scale=None, rows_per_batch=0, dres=None, dx_out=None, want_dbeta=False,
                  dgamma_out=None, accumulate=False, dx_bf16=None, dx_colsum_out=None):
    """returns (dx fp32, dgamma fp32 [D], dbeta fp32 [D] or None); dgamma_out (+accumulate) writes dgamma in place;
    dx_bf16: optional bf16 tensor that receives a rounded copy of dx; dx_colsum_out (trainer mode only: dgamma_out +
    accumulate): fp32 [D] that the kernel atomically ADDS the column sums of the bf16-rounded dx to"""
    lib = _lib.load()
    rows, D = _rows2d(x)
    assert dy.dtype == torch.bfloat16 and dy.is_contiguous() and x.is_contiguous()
    if dx_out is None:
        dx_out = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    ld_mod = scale.stride(-2) if scale is not None else 0
    if dx_colsum_out is not None:
        assert dgamma_out is not None and accumulate and not want_dbeta
        check(lib.kalle_layernorm_bwd_colsum(_p(dy), _p(x), _dt(x), _p(gamma), _p(scale), ld_mod, rows_per_batch, _p(mean),
                                             _p(rstd), _p(dres), _p(dx_out), _p(dx_bf16), _p(dgamma_out), _p(dx_colsum_out),
                                             rows, D, _stream()), "kalle_layernorm_bwd_colsum")
        return dx_out, dgamma_out, None
    if dgamma_out is not None and accumulate and not want_dbeta and os.environ.get("KALLE_LN_ATOMIC", "1") != "0":
        # trainer mode: the gradient sink already holds this step's running sum - add into it from the kernel
        check(lib.kalle_layernorm_bwd_acc(_p(dy), _p(x), _dt(x), _p(gamma), _p(scale), ld_mod, rows_per_batch, _p(mean),
                                          _p(rstd), _p(dres), _p(dx_out), _p(dx_bf16), _p(dgamma_out), None, rows, D,
                                          _stream()), "kalle_layernorm_bwd_acc")
        return dx_out, dgamma_out, None
    nparts = lib.kalle_layernorm_bwd_parts(rows)
    dgp = torch.empty((nparts, D), device=x.device, dtype=torch.float32)
    dbp = torch.empty((nparts, D), device=x.device, dtype=torch.float32) if want_dbeta else None
    check(lib.kalle_layernorm_bwd(_p(dy), _p(x), _dt(x), _p(gamma), _p(scale), ld_mod, rows_per_batch, _p(mean),
                                  _p(rstd), _p(dres), _p(dx_out), _p(dx_bf16), _p(dgp), _p(dbp), rows, D, _stream()),
          "kalle_layernorm_bwd")
    dgamma = colsum(dgp, out=dgamma_out, accumulate=accumulate)
    dbeta = colsum(dbp) if want_dbeta else None
    return dx_out, dgamma, dbeta


def adaln_mod_bwd(dy, x, gamma, beta, mean, rstd, nbatch, rows_per_batch):
    lib = _lib.load()
    D = x.shape[-1]
    dscale = torch.empty((nbatch, D), device=x.device, dtype=torch.float32)
    dshift = torch.empty((nbatch, D), device=x.device, dtype=torch.float32)
    check(lib.kalle_adaln_mod_bwd(_p(dy), _p(x), _dt(x), _p(gamma), _p(beta), _p(mean), _p(rstd), _p(dscale),
                                  _p(dshift), D, nbatch, rows_per_batch, D, _stream()), "kalle_adaln_mod_bwd")
    return dscale, dshift


def rmsnorm_fwd(x, scale, rows_per_batch=0, eps=1e-6, out_dtype=None):
    lib = _lib.load()
    rows, D = _rows2d(x)
    x = x.contiguous()
    y = torch.empty(x.shape, device=x.device, dtype=out_dtype or x.dtype)
    rrms = torch.empty(rows, device=x.device, dtype=torch.float32)
    ld = scale.stride(-2) if scale.dim() >= 2 else 0
    check(lib.kalle_rmsnorm_fwd(_p(x), _dt(x), _p(scale), ld, rows_per_batch, _p(y), _dt(y), _p(rrms), rows, D, eps,
                                _stream()), "kalle_rmsnorm_fwd")
    return y, rrms


def rmsnorm_bwd(dy, x, scale, rrms, rows_per_batch=0, dres=None, dx_bf16=None, dscale_out=None, accumulate=False):
    """returns (dx fp32 [+ dres], dscale) - dscale is written / accumulated into `dscale_out` when given"""
    lib = _lib.load()
    rows, D = _rows2d(x)
    dy = dy.contiguous()
    dx = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    ld = scale.stride(-2) if scale.dim() >= 2 else 0
    if dscale_out is not None and accumulate and ld == 0:
        # trainer mode: add into the gradient sink from the kernel (no partial rows, no reduction launch)
        check(lib.kalle_rmsnorm_bwd_acc(_p(dy), _dt(dy), _p(x), _dt(x), _p(scale), ld, rows_per_batch, _p(rrms), _p(dx),
                                        _p(dscale_out), _p(dres), _p(dx_bf16), rows, D, _stream()), "kalle_rmsnorm_bwd_acc")
        return dx, None
    nparts = lib.kalle_layernorm_bwd_parts(rows)
    dsp = torch.empty((nparts, D), device=x.device, dtype=torch.float32)
    check(lib.kalle_rmsnorm_bwd(_p(dy), _dt(dy), _p(x), _dt(x), _p(scale), ld, rows_per_batch, _p(rrms), _p(dx),
                                _p(dsp), _p(dres), _p(dx_bf16), rows, D, _stream()), "kalle_rmsnorm_bwd")
    if dscale_out is not None:
        colsum(dsp, out=dscale_out, accumulate=accumulate)
        return dx, None
    return dx, colsum(dsp)


def colsum(x, out=None, accumulate=False):
    lib = _lib.load()
    rows, cols = _rows2d(x)
    ld = x.stride(-2) if x.dim() >= 2 else cols
    if out is None:
        out = torch.empty(cols, device=x.device, dtype=torch.float32)
        accumulate = False
    check(lib.kalle_colsum(_p(x), _dt(x), ld, _p(out), rows, cols, int(accumulate), _stream()), "kalle_colsum")
    return out


# ------------------------------------------------------------------------------------------------ elementwise
def swiglu_fwd(h):
    lib = _lib.load()
    rows, two_inner = _rows2d(h)
    inner = two_inner // 2
    out = torch.empty(h.shape[:-1] + (inner,), device=h.device, dtype=torch.bfloat16)
    check(lib.kalle_swiglu_fwd(_p(h), _p(out), rows, inner, _stream()), "kalle_swiglu_fwd")
    return out


def swiglu_bwd(dout, h, dbias=None):
    """dbias: optional fp32 [2*inner] that the kernel atomically adds the column sums of dh to"""
    lib = _lib.load()
    rows, two_inner = _rows2d(h)
    dh = torch.empty_like(h)
    check(lib.kalle_swiglu_bwd(_p(dout), _p(h), _p(dh), _p(dbias), rows, two_inner // 2, _stream()),
          "kalle_swiglu_bwd")
    return dh


def silu_fwd(x):
    lib = _lib.load()
    y = torch.empty_like(x)
    check(lib.kalle_silu_fwd(_p(x), _p(y), _dt(x), x.numel(), _stream()), "kalle_silu_fwd")
    return y


def silu_bwd(dy, x):
    lib = _lib.load()
    dx = torch.empty_like(x)
    check(lib.kalle_silu_bwd(_p(dy), _p(x), _p(dx), _dt(x), x.numel(), _stream()), "kalle_silu_bwd")
    return dx


def diffuse_fwd(x, noise, t, objective="v"):
    lib = _lib.load()
    x = x.contiguous()
    noise = noise.contiguous()
    xt = torch.empty_like(x)
    target = torch.empty_like(x)
    B = x.shape[0]
    check(lib.kalle_diffuse_fwd(_p(x), _p(noise), _p(t), _p(xt), _p(target), B, x.numel() // B,
                                0 if objective == "v" else 1, _stream()), "kalle_diffuse_fwd")
    return xt, target


def mse_loss(out, target, mask=None, weight=1.0, want_grad=True):
    """returns (loss scalar tensor, dout or None). out/target fp32 [B,C,T]; mask bool/uint8 [B,T]."""
    lib = _lib.load()
    out = out.contiguous()
    target = target.contiguous()
    B, C, T = out.shape
    acc = torch.zeros(2, device=out.device, dtype=torch.float32)
    diff = torch.empty_like(out) if want_grad else None
    m8 = mask.to(torch.uint8).contiguous() if mask is not None else None
    check(lib.kalle_mse_fwd(_p(out), _p(target), _p(m8), _p(acc), _p(diff), B, C, T, _stream()), "kalle_mse_fwd")
    loss = torch.empty((), device=out.device, dtype=torch.float32)
    check(lib.kalle_mse_finish(_p(acc), _p(loss), _p(diff), out.numel(), weight, _stream()), "kalle_mse_finish")
    return loss, diff


def transpose_2d(x, out_dtype=None, out=None, R=None, Cn=None, in_batch_stride=None, in_ld=None,
                 out_batch_stride=None, out_ld=None):
    """out[b][c][r] = x[b][r][c]"""
    lib = _lib.load()
    B = x.shape[0]
    R = R if R is not None else x.shape[1]
    Cn = Cn if Cn is not None else x.shape[2]
    in_batch_stride = in_batch_stride if in_batch_stride is not None else x.stride(0)
    in_ld = in_ld if in_ld is not None else x.stride(1)
    if out is None:
        out = torch.empty((B, Cn, R), device=x.device, dtype=out_dtype or x.dtype)
    out_batch_stride = out_batch_stride if out_batch_stride is not None else out.stride(0)
    out_ld = out_ld if out_ld is not None else out.stride(1)
    check(lib.kalle_transpose_2d(_p(x), _dt(x), in_batch_stride, in_ld, _p(out), _dt(out), out_batch_stride, out_ld,
                                 B, R, Cn, _stream()), "kalle_transpose_2d")
    return out


def copy_rows(src, dst, nbatch, rows, cols, src_batch_stride, src_ld, dst_batch_stride, dst_ld, accumulate=False):
    lib = _lib.load()
    check(lib.kalle_copy_rows(_p(src), _dt(src), src_batch_stride, src_ld, _p(dst), _dt(dst), dst_batch_stride,
                              dst_ld, nbatch, rows, cols, int(accumulate), _stream()), "kalle_copy_rows")
    return dst


def cast(x, dtype):
    lib = _lib.load()
    x = x.contiguous()
    out = torch.empty(x.shape, device=x.device, dtype=dtype)
    check(lib.kalle_cast(_p(x), _dt(x), _p(out), _dt(out), x.numel(), _stream()), "kalle_cast")
    return out


def fourier_features(t, w, out_dtype=torch.float32):
    lib = _lib.load()
    B, half = t.shape[0], w.shape[0]
    out = torch.empty((B, 2 * half), device=t.device, dtype=out_dtype)
    check(lib.kalle_fourier_features(_p(t.contiguous()), _p(w.contiguous()), _p(out), _dt(out), B, half, _stream()),
          "kalle_fourier_features")
    return out


def fourier_features_bwd(dout, t, w):
    lib = _lib.load()
    dw = torch.empty_like(w)
    check(lib.kalle_fourier_features_bwd(_p(dout.contiguous()), _p(t.contiguous()), _p(w.contiguous()), _p(dw),
                                         t.shape[0], w.shape[0], _stream()), "kalle_fourier_features_bwd")
    return dw


def grad_cast(g, nbatch, rows_per_batch, gate=None, x_out=None, x_in=None, row_mask=None):
    """fp32 residual-stream gradient -> bf16 GEMM operand (+ adaLN gate backward). Returns (gb, dgate or None)."""
    lib = _lib.load()
    D = g.shape[-1]
    g = g.contiguous()
    gb = torch.empty(g.shape, device=g.device, dtype=torch.bfloat16)
    if gate is not None:
        gate = gate.contiguous()  # [B, D] slice of the adaLN modulation; dgate shares its leading dim
    dgate = torch.empty_like(gate) if gate is not None else None
    check(lib.kalle_grad_cast(_p(g), _p(x_out), _p(x_in), _p(gate), gate.stride(-2) if gate is not None else 0,
                              _p(row_mask), _p(gb), _p(dgate), nbatch, rows_per_batch, D, _stream()),
          "kalle_grad_cast")
    return gb, dgate


WEIGHTS_EPOCH = 0      # bumped by every launch that writes weights through a raw pointer (torch's `_version` does not see it):
                       # part of the key of every cache derived from weights (stacked k | v weights, captured graphs)


def adam_step(param, grad, exp_avg, exp_avg_sq, param_bf16, *, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0,
              decoupled=False, step=1, grad_scale=1.0):
    global WEIGHTS_EPOCH
    WEIGHTS_EPOCH += 1
    lib = _lib.load()
    check(lib.kalle_adam_step(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), _p(param_bf16), param.numel(), lr,
                              beta1, beta2, eps, weight_decay, int(decoupled), step, grad_scale, _stream()),
          "kalle_adam_step")


# ------------------------------------------------------------------------------------------------ attention
def attention_fwd(q, k, v, *, ldq, q_off, ldk, k_off, ldv, v_off, B, H, Hkv, Nq, Nk, rope=None, key_mask=None,
                  causal=False):
    """q/k/v: base tensors (bf16) of the projection outputs; see kalle_attention_fwd. Returns (out [B,Nq,H*64], lse)."""
    lib = _lib.load()
    out = torch.empty((B, Nq, H * 64), device=q.device, dtype=torch.bfloat16)
    lse = torch.empty((B, H, Nq), device=q.device, dtype=torch.float32)
    cos, sin, rot = (rope[0], rope[1], rope[0].shape[-1] * 2) if rope is not None else (None, None, 0)
    m8 = key_mask.to(torch.uint8).contiguous() if key_mask is not None else None
    check(lib.kalle_attention_fwd(_p(q), ldq, q_off, _p(k), ldk, k_off, _p(v), ldv, v_off, _p(out), H * 64, _p(lse),
                                  _p(cos), _p(sin), rot, _p(m8), int(causal), B, H, Hkv, Nq, Nk, _stream()),
          "kalle_attention_fwd")
    return out, lse


def attention_bwd(q, k, v, out, dout, lse, dq, dk, dv, *, ldq, q_off, ldk, k_off, ldv, v_off, B, H, Hkv, Nq, Nk,
                  rope=None, key_mask=None, causal=False):
    lib = _lib.load()
    delta = torch.empty((B, H, Nq), device=q.device, dtype=torch.float32)
    cos, sin, rot = (rope[0], rope[1], rope[0].shape[-1] * 2) if rope is not None else (None, None, 0)
    m8 = key_mask.to(torch.uint8).contiguous() if key_mask is not None else None
    check(lib.kalle_attention_bwd(_p(q), ldq, q_off, _p(k), ldk, k_off, _p(v), ldv, v_off, _p(out), _p(dout), H * 64,
                                  _p(lse), _p(delta), _p(dq), _p(dk), _p(dv), _p(cos), _p(sin), rot, _p(m8),
                                  int(causal), B, H, Hkv, Nq, Nk, _stream()), "kalle_attention_bwd")


def head_norm_fwd(x, ldx, x_off, rows, heads, mode, gamma=None, beta=None):
    """qk_norm (transformer.py:422-428) on the q or k slice of a projection output: returns (y bf16 [rows, heads*64], stat)"""
    lib = _lib.load()
    y = torch.empty((rows, heads * 64), device=x.device, dtype=torch.bfloat16)
    stat = torch.empty((rows, heads, 2), device=x.device, dtype=torch.float32)
    check(lib.kalle_head_norm_fwd(_p(x), ldx, x_off, _p(y), heads * 64, 0, _p(stat), _p(gamma), _p(beta), mode, rows, heads,
                                  _stream()), "kalle_head_norm_fwd")
    return y, stat


def head_norm_bwd(x, ldx, x_off, stat, g, dx, lddx, dx_off, rows, heads, mode, gamma=None, dgamma=None, dbeta=None):
    """g: bf16 [rows, heads*64] gradient w.r.t. the normalised values; writes the gradient w.r.t. x into dx (ld / offset)"""
    lib = _lib.load()
    check(lib.kalle_head_norm_bwd(_p(x), ldx, x_off, _p(stat), _p(g), heads * 64, 0, _p(dx), lddx, dx_off, _p(gamma),
                                  _p(dgamma), _p(dbeta), mode, rows, heads, _stream()), "kalle_head_norm_bwd")


# ------------------------------------------------------------------------------------------------ Llasa head / tail
def peak_normalize_int16(x):
    """int16(clamp(x / max|x|, -1, 1) * 32767) (infer_0723.py:293); returns (int16 tensor of x's shape, peak [1] fp32)"""
    lib = _lib.load()
    x = x.contiguous()
    peak = torch.empty(1, device=x.device, dtype=torch.float32)
    out = torch.empty(x.shape, device=x.device, dtype=torch.int16)
    check(lib.kalle_peak_normalize_int16(_p(x), _dt(x), _p(peak), _p(out), x.numel(), _stream()),
          "kalle_peak_normalize_int16")
    return out, peak


def axpby(x, y, a, b, out=None):
    """out = a x + b y (fp32); out may alias x or y"""
    lib = _lib.load()
    x, y = x.contiguous(), y.contiguous()
    assert x.dtype == torch.float32 and y.dtype == torch.float32 and x.shape == y.shape
    out = torch.empty_like(x) if out is None else out
    check(lib.kalle_axpby(_p(x), _p(y), _p(out), float(a), float(b), x.numel(), _stream()), "kalle_axpby")
    return out


def add_rows(x, table):
    """x[b] += table for every batch element b, in place (transformer.py:796-797); x fp32 [B, ...] contiguous, table fp32 with
    the element count of x[0]"""
    lib = _lib.load()
    assert x.dtype == torch.float32 and table.dtype == torch.float32 and x.is_contiguous() and table.is_contiguous()
    assert x[0].numel() == table.numel()
    check(lib.kalle_add_rows(_p(x), _p(table), x.shape[0], table.numel(), _stream()), "kalle_add_rows")
    return x


def dwconv1d(x, w, B, N, out_dtype, pad, flip=False):
    """depthwise conv along the sequence of token-major bf16 activations [B * N, D] (ConformerModule, transformer.py:564);
    w fp32 [D, K]; flip=True with pad = K - 1 - padding is the data gradient"""
    lib = _lib.load()
    D, K = w.shape
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and w.dtype == torch.float32 and w.is_contiguous()
    assert x.numel() == B * N * D
    y = torch.empty((B * N, D), device=x.device, dtype=out_dtype)
    check(lib.kalle_dwconv1d_fwd(_p(x), _p(w), _p(y), _dt(y), B, N, D, K, pad, int(flip), _stream()), "kalle_dwconv1d_fwd")
    return y


def dwconv1d_wgrad(dy, x, dw, B, N, pad):
    """dw [D, K] fp32 += sum_{b, n} dy[b, n, :, None] * x[b, n + k - pad, :, None]   (atomic adds)"""
    lib = _lib.load()
    D, K = dw.shape
    assert dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and dw.dtype == torch.float32
    assert dy.is_contiguous() and x.is_contiguous() and dw.is_contiguous() and dy.numel() == B * N * D == x.numel()
    check(lib.kalle_dwconv1d_wgrad(_p(dy), _p(x), _p(dw), B, N, D, K, pad, _stream()), "kalle_dwconv1d_wgrad")
    return dw


def embed_mix_fwd(ids, table, audio, ids_mask, audio_mask):
    """ids [rows] int64, table fp32 [V, D], audio fp32/bf16 [rows, D], masks fp32 [rows] -> fp32 [rows, D]"""
    lib = _lib.load()
    rows, D = audio.shape
    out = torch.empty((rows, D), device=audio.device, dtype=torch.float32)
    check(lib.kalle_embed_mix_fwd(_p(ids), _p(table), _p(audio), _dt(audio), _p(ids_mask), _p(audio_mask), _p(out), rows, D,
                                  table.shape[0], _stream()), "kalle_embed_mix_fwd")
    return out


def embed_mix_bwd(dout, ids, ids_mask, audio_mask, dtable=None, want_daudio=True):
    lib = _lib.load()
    rows, D = dout.shape
    daudio = torch.empty((rows, D), device=dout.device, dtype=torch.float32) if want_daudio else None
    V = dtable.shape[0] if dtable is not None else 1
    check(lib.kalle_embed_mix_bwd(_p(dout), _p(ids), _p(ids_mask), _p(audio_mask), _p(dtable), _p(daudio), rows, D, V,
                                  _stream()), "kalle_embed_mix_bwd")
    return daudio


def gelu_fwd(x):
    lib = _lib.load()
    x = x.contiguous()
    y = torch.empty_like(x)
    check(lib.kalle_gelu_fwd(_p(x), _p(y), _dt(x), x.numel(), _stream()), "kalle_gelu_fwd")
    return y


def gelu_bwd(dy, x):
    lib = _lib.load()
    dy = dy.contiguous()
    assert dy.dtype == x.dtype
    dx = torch.empty_like(x)
    check(lib.kalle_gelu_bwd(_p(dy), _p(x), _p(dx), _dt(x), x.numel(), _stream()), "kalle_gelu_bwd")
    return dx


def gauss_kl_fwd(pred, label, mask_a, mask_b, std):
    """returns sums4 = [sum kl*ma, sum ma, sum kl*mb, sum mb] (fp32, device)"""
    lib = _lib.load()
    rows, d = pred.shape
    sums = torch.zeros(4, device=pred.device, dtype=torch.float32)
    check(lib.kalle_gauss_kl_fwd(_p(pred), _p(label), _p(mask_a), _p(mask_b), _p(sums), float(std), rows, d, _stream()),
          "kalle_gauss_kl_fwd")
    return sums


def gauss_kl_bwd(pred, label, mask_a, mask_b, sums, grad_a, grad_b, std):
    lib = _lib.load()
    rows, d = pred.shape
    dpred = torch.empty_like(pred)
    check(lib.kalle_gauss_kl_bwd(_p(pred), _p(label), _p(mask_a), _p(mask_b), _p(sums), _p(grad_a), _p(grad_b), _p(dpred),
                                 float(std), rows, d, _stream()), "kalle_gauss_kl_bwd")
    return dpred


def llama_decode_plan(layer_tensors, H, Hkv, inner, device):
    """layer_tensors: per layer (input_norm fp32, wqkv bf16, wo bf16, post_norm fp32, wug bf16, wdown bf16, kv_cache bf16).
    Returns the host-side descriptor array + workspace of kalle_llama_decode_step (keeps the tensors alive)."""
    lib = _lib.load()
    arr = (_lib.LlamaLayer * len(layer_tensors))()
    for d, ts in zip(arr, layer_tensors):
        for t in ts:
            assert t.is_contiguous() and t.device == torch.device(device)
        d.input_norm, d.wqkv, d.wo, d.post_norm, d.wug, d.wdown, d.kv_cache = (t.data_ptr() for t in ts)
    ws = torch.empty(lib.kalle_llama_decode_ws_bytes(H, Hkv, inner), device=device, dtype=torch.uint8)
    return {"layers": arr, "n": len(layer_tensors), "keep": layer_tensors, "ws": ws, "H": H, "Hkv": Hkv, "inner": inner}


def llama_decode_step(plan, x, t0, cache_rows, rope, eps):
    """x fp32 [D] -> fp32 [D]: every decoder layer at position t0 against the KV caches of `plan` (one host call)"""
    lib = _lib.load()
    out = torch.empty_like(x)
    check(lib.kalle_llama_decode_step(ctypes.cast(plan["layers"], ctypes.c_void_p), plan["n"], _p(x), _p(out), plan["H"],
                                      plan["Hkv"], plan["inner"], eps, t0, cache_rows, _p(rope[0]), _p(rope[1]),
                                      _p(plan["ws"]), _stream()), "kalle_llama_decode_step")
    return out


def gauss_kl2_fwd(pred, label_mean, label_std, mask_a, mask_b, std_mult=1.25):
    """two-Gaussian KL (model.py:84-100); label_std None -> label_mean is the raw mean | scale label [rows, 2 dim]"""
    lib = _lib.load()
    rows, d2 = pred.shape
    sums = torch.zeros(4, device=pred.device, dtype=torch.float32)
    check(lib.kalle_gauss_kl2_fwd(_p(pred), _p(label_mean), _p(label_std), 0 if label_std is not None else 1, float(std_mult),
                                  _p(mask_a), _p(mask_b), _p(sums), rows, d2 // 2, _stream()), "kalle_gauss_kl2_fwd")
    return sums


def gauss_kl2_bwd(pred, label_mean, label_std, mask_a, mask_b, sums, grad_a, grad_b, std_mult=1.25):
    lib = _lib.load()
    rows, d2 = pred.shape
    dpred = torch.empty_like(pred)
    check(lib.kalle_gauss_kl2_bwd(_p(pred), _p(label_mean), _p(label_std), 0 if label_std is not None else 1, float(std_mult),
                                  _p(mask_a), _p(mask_b), _p(sums), _p(grad_a), _p(grad_b), _p(dpred), rows, d2 // 2,
                                  _stream()), "kalle_gauss_kl2_bwd")
    return dpred


def segment_copy(src, dst, src_off, dst_off, lens, nbatch, rows, *, src_strides, dst_strides):
    """kalle_segment_copy: per segment s, dst[s, b, c, dst_off[s] + j] = src[s, b, c, src_off[s] + j], j < lens[s]; strides =
    (segment, batch, row) in elements; every segment in one launch (groups of KALLE_MAX_SEGMENTS)"""
    lib = _lib.load()
    assert src.dtype == dst.dtype
    MAXS = 64
    esz = src.element_size()
    for g in range(0, len(lens), MAXS):
        n = min(MAXS, len(lens) - g)
        so = (ctypes.c_int64 * n)(*[int(v) for v in src_off[g:g + n]])
        do = (ctypes.c_int64 * n)(*[int(v) for v in dst_off[g:g + n]])
        ln = (ctypes.c_int * n)(*[int(v) for v in lens[g:g + n]])
        sp = ctypes.c_void_p(src.data_ptr() + g * src_strides[0] * esz)
        dp = ctypes.c_void_p(dst.data_ptr() + g * dst_strides[0] * esz)
        if not src.is_cuda or not dst.is_cuda:
            raise RuntimeError("kalle_audio_amd ops run on the GPU only (HIP kernels); got a CPU tensor")
        check(lib.kalle_segment_copy(sp, dp, _dt(src), n, so, do, ln, nbatch, rows, src_strides[0], src_strides[1],
                                     src_strides[2], dst_strides[0], dst_strides[1], dst_strides[2], _stream()),
              "kalle_segment_copy")
    return dst


MAX_GROUP = 8


def gemm_wgrad_group(problems, overwrite=False):
    """problems: list of (dy bf16 [tokens, N], x bf16 [tokens, K], dw fp32 [N, K]); dw += dy^T x (overwrite: dw = dy^T x) for
    all of them in one launch per 8.  Returns False (nothing launched) when the shapes are outside the grouped kernel's domain."""
    lib = _lib.load()
    for dy, x, dw in problems:
        if dy.shape[0] % 8 or dy.shape[0] != x.shape[0] or dy.shape[1] % 8 or x.shape[1] % 8:
            return False
        assert dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and dw.dtype == torch.float32
        assert dy.stride(-1) == 1 and x.stride(-1) == 1 and dw.stride(-1) == 1
    prof = KERNEL_TIMER if KERNEL_TIMER_ONLY in (None, "gemm3_wgrad_group_kernel") else None
    for g in range(0, len(problems), MAX_GROUP):
        grp = problems[g:g + MAX_GROUP]
        arr = (_lib.WgradProblem * len(grp))()
        flops = abytes = 0.0
        for w, (dy, x, dw) in zip(arr, grp):
            w.dy, w.lddy, w.x, w.ldx, w.dw, w.lddw = dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), dw.data_ptr(), dw.stride(0)
            w.N, w.K, w.tokens = dy.shape[1], x.shape[1], dy.shape[0]
            _p(dy), _p(x), _p(dw)                                     # (CPU tensors raise)
            flops += 2.0 * dy.shape[0] * dy.shape[1] * x.shape[1]
            abytes += 2.0 * dy.shape[0] * (dy.shape[1] + x.shape[1]) + 8.0 * dy.shape[1] * x.shape[1]
        if prof is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        rc = lib.kalle_gemm_wgrad_group(ctypes.cast(arr, ctypes.c_void_p), len(grp), int(bool(overwrite)), _stream())
        if rc == -3 and g == 0:
            return False
        check(rc, "kalle_gemm_wgrad_group")
        if prof is not None:
            e1.record()
            prof.append(("gemm3_wgrad_group_kernel", flops, e0, e1, abytes, (len(grp), 0, grp[0][0].shape[0])))
    return True
