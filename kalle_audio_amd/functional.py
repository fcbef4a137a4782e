"""torch.autograd.Function shims over dit_ops / ops: they own forward + backward of each drop-in module, so a
reference training loop (plain torch autograd, any torch optimizer) runs the hand-written HIP kernels end to end.
No torch math op computes anything here; torch supplies tensors, streams and the autograd tape only.
"""
import os

import torch

from . import dit_ops as D
from . import ops

BF16, F32 = torch.bfloat16, torch.float32


def _to_bf16(x):
    return x if x.dtype == BF16 else ops.cast(x.float() if x.dtype not in (F32, BF16) else x, BF16)


def _to_f32(x):
    return x if x.dtype == F32 else ops.cast(x if x.dtype == BF16 else x.to(BF16), F32)


def _like(t, dtype):
    """cast result back to the caller's dtype (fp32 or bf16; fp16 callers get fp32->half via torch plumbing)."""
    if t.dtype == dtype:
        return t
    if dtype == F32:
        return _to_f32(t)
    if dtype == BF16:
        return _to_bf16(t)
    return _to_f32(t).to(dtype)


def _mask8(m):
    if m is None:
        return None
    return m.to(torch.uint8).contiguous()


def _pad8(t, rows=False):
    """zero-pad the last dim (and optionally the first) of a 2-D tensor to a multiple of 8 - the GEMM's granule.  Only odd
    channel counts get here (io_channels 4 in tests; the reference's configs use 64 / 512 / 1024); zeros add nothing to a dot
    product, padded output columns are sliced off."""
    pc = (-t.shape[-1]) % 8
    pr = (-t.shape[0]) % 8 if rows else 0
    return torch.nn.functional.pad(t, (0, pc, 0, pr)) if (pc or pr) else t


class LinearFn(torch.autograd.Function):
    """y = x @ W^T (+ b) (+ residual); x: [..., K] (bf16 or fp32), W: [N, K] fp32 parameter."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, out_dtype):
        shp = x.shape
        N, K = weight.shape
        xb = _to_bf16(x.contiguous()).view(-1, K)
        wb = D.bf16_of(weight)
        res = residual.contiguous().view(-1, N) if residual is not None else None
        b32 = D.f32_of(bias) if bias is not None else None
        ctx.padded = bool(K % 8 or N % 8)
        if ctx.padded:
            xb, wb = _pad8(xb), _pad8(wb, rows=True)
            res = _pad8(res) if res is not None else None
            b32 = torch.nn.functional.pad(b32, (0, (-N) % 8)) if b32 is not None else None
        y = ops.gemm(xb, wb, bias=b32, residual=res, out_dtype=out_dtype)
        if ctx.padded:
            y = y[:, :N].contiguous()
        ctx.save_for_backward(xb, weight)
        ctx.has_bias = bias is not None
        ctx.has_res = residual is not None
        ctx.x_dtype = x.dtype
        ctx.shp = shp
        return y.view(*shp[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        xb, weight = ctx.saved_tensors
        N, K = weight.shape
        wb = D.bf16_of(weight)
        dyc = dy.contiguous().view(-1, N)
        gb = _to_bf16(dyc)
        if ctx.padded:
            gb, wb = _pad8(gb), _pad8(wb, rows=True)
        dx = dw = db = dres = None
        if ctx.needs_input_grad[0]:
            dx = D.dgrad(gb, wb)
            dx = _like(dx[:, :K].contiguous() if ctx.padded else dx, ctx.x_dtype).view(ctx.shp)
        if ctx.needs_input_grad[1]:
            dw = ops.gemm(gb, xb, a_kmajor=True, b_kmajor=True, out_dtype=F32)
            dw = dw[:N, :K].contiguous() if ctx.padded else dw
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = ops.colsum(gb)
            db = db[:N].contiguous() if ctx.padded else db
        if ctx.has_res and ctx.needs_input_grad[3]:
            dres = dy
        return dx, dw, db, dres, None


def linear(x, weight, bias=None, residual=None, out_dtype=None):
    if out_dtype is None:
        out_dtype = F32 if (x.dtype == F32 or residual is not None) else BF16
    return LinearFn.apply(x, weight, bias, residual, out_dtype)


class SiLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return ops.silu_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.silu_bwd(_like(dy.contiguous(), x.dtype), x)


def silu(x):
    if x.dtype not in (F32, BF16):
        x = x.float()
    return SiLUFn.apply(x)


class LayerNormFn(torch.autograd.Function):
    """transformer.py:173-192"""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        shp = x.shape
        xc = x.contiguous().view(-1, shp[-1])
        if xc.dtype not in (F32, BF16):
            xc = xc.float()
        y, mean, rstd = ops.layernorm_fwd(xc, D.f32_of(gamma), D.f32_of(beta) if beta is not None else None, eps=eps)
        ctx.save_for_backward(xc, gamma, mean, rstd)
        ctx.has_beta = beta is not None
        ctx.x_dtype = x.dtype
        return _like(y, x.dtype).view(shp)

    @staticmethod
    def backward(ctx, dy):
        xc, gamma, mean, rstd = ctx.saved_tensors
        dyb = _to_bf16(dy.contiguous().view(-1, xc.shape[-1]))
        dx, dgamma, dbeta = ops.layernorm_bwd(dyb, xc, D.f32_of(gamma), mean, rstd, want_dbeta=ctx.has_beta)
        return _like(dx, ctx.x_dtype).view(dy.shape), dgamma, dbeta, None


class RMSNormFn(torch.autograd.Function):
    """blocks.py:268-272"""

    @staticmethod
    def forward(ctx, x, scale, eps):
        shp = x.shape
        xc = x.contiguous().view(-1, shp[-1])
        if xc.dtype not in (F32, BF16):
            xc = xc.float()
        y, rr = ops.rmsnorm_fwd(xc, D.f32_of(scale), eps=eps, out_dtype=xc.dtype)
        ctx.save_for_backward(xc, scale, rr)
        return y.view(shp).to(x.dtype)

    @staticmethod
    def backward(ctx, dy):
        xc, scale, rr = ctx.saved_tensors
        dyc = dy.contiguous().view(-1, xc.shape[-1])
        if dyc.dtype not in (F32, BF16):
            dyc = dyc.float()
        dx, dscale = ops.rmsnorm_bwd(dyc, xc, D.f32_of(scale), rr)
        return _like(dx, xc.dtype).view(dy.shape).to(dy.dtype), dscale.view(scale.shape), None


class FourierFeaturesFn(torch.autograd.Function):
    """blocks.py:84-93 (in_features == 1)"""

    @staticmethod
    def forward(ctx, t, weight):
        tt = t.contiguous().view(-1).float()
        w = D.f32_of(weight).contiguous().view(-1)
        ctx.save_for_backward(tt, w)
        ctx.wshape = weight.shape
        return ops.fourier_features(tt, w)

    @staticmethod
    def backward(ctx, dout):
        tt, w = ctx.saved_tensors
        dw = ops.fourier_features_bwd(_to_f32(dout.contiguous()), tt, w)
        return None, dw.view(ctx.wshape)


class TransposeFn(torch.autograd.Function):
    """(B, R, C) -> (B, C, R) with dtype conversion; optionally dropping the first `skip` rows (dit.py:199,219)."""

    @staticmethod
    def forward(ctx, x, out_dtype, skip):
        x = x.contiguous()
        if x.dtype not in (F32, BF16):
            x = x.float()
        B, R, C = x.shape
        ctx.meta = (x.dtype, B, R, C, skip)
        src = x[:, skip:] if skip else x
        return ops.transpose_2d(src, out_dtype=out_dtype, R=R - skip, Cn=C, in_batch_stride=x.stride(0),
                                in_ld=x.stride(1))

    @staticmethod
    def backward(ctx, dy):
        dtype, B, R, C, skip = ctx.meta
        dy = dy.contiguous()
        if dy.dtype not in (F32, BF16):
            dy = dy.float()
        if skip:
            dx = torch.zeros((B, R, C), device=dy.device, dtype=dtype)
            ops.transpose_2d(dy, out=dx[:, skip:], out_batch_stride=dx.stride(0), out_ld=dx.stride(1))
        else:
            dx = ops.transpose_2d(dy, out_dtype=dtype)
        return dx, None, None


def transpose(x, out_dtype=None, skip=0):
    return TransposeFn.apply(x, out_dtype or (x.dtype if x.dtype in (F32, BF16) else F32), skip)


class SpliceFn(torch.autograd.Function):
    """residual-stream assembly: out = cat(prepend [B,P,D], tokens [B,T,D]) in fp32 (transformer.py:776-781)."""

    @staticmethod
    def forward(ctx, prepend, tokens):
        tokens = tokens.contiguous()
        B, T, Dm = tokens.shape
        P = prepend.shape[1] if prepend is not None else 0
        out = torch.empty((B, P + T, Dm), device=tokens.device, dtype=F32)
        if P:
            pre = prepend.contiguous()
            pre = pre if pre.dtype in (F32, BF16) else pre.float()
            ops.copy_rows(pre, out, B, P, Dm, pre.stride(0), pre.stride(1), out.stride(0), out.stride(1))
        tk = tokens if tokens.dtype in (F32, BF16) else tokens.float()
        ops.copy_rows(tk, out[:, P:], B, T, Dm, tk.stride(0), tk.stride(1), out.stride(0), out.stride(1))
        ctx.meta = (P, T, prepend.dtype if prepend is not None else None, tokens.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        P, T, pdt, tdt = ctx.meta
        g = g.contiguous()
        B, _, Dm = g.shape
        dpre = None
        if P and ctx.needs_input_grad[0]:
            dpre = torch.empty((B, P, Dm), device=g.device, dtype=F32)
            ops.copy_rows(g, dpre, B, P, Dm, g.stride(0), g.stride(1), dpre.stride(0), dpre.stride(1))
            dpre = dpre.to(pdt)
        odt = tdt if tdt in (F32, BF16) else F32
        dtok = torch.empty((B, T, Dm), device=g.device, dtype=odt)
        ops.copy_rows(g[:, P:], dtok, B, T, Dm, g.stride(0), g.stride(1), dtok.stride(0), dtok.stride(1))
        return dpre, dtok.to(tdt)


class AddRowsFn(torch.autograd.Function):
    """x + table for every batch element (the position embedding of a ContinuousTransformer, transformer.py:796-797).  x: the fp32
    residual stream [B, N, D] (a fresh tensor from SpliceFn: updated in place), table: [N, D]; the table's gradient is the sum
    of the incoming gradient over the batch (kalle_colsum over [B, N*D])."""

    @staticmethod
    def forward(ctx, x, table):
        tab = _to_f32(table.contiguous())
        if x.dtype == F32 and x.is_contiguous() and not x.is_leaf:
            ctx.mark_dirty(x)
            out = x
        else:
            xf = _to_f32(x.contiguous())
            out = ops.axpby(xf, xf, 1.0, 0.0)
        ops.add_rows(out, tab)
        ctx.tdt = table.dtype
        ctx.tshape = table.shape
        return out

    @staticmethod
    def backward(ctx, g):
        dt = None
        if ctx.needs_input_grad[1]:
            gf = _to_f32(g.contiguous())
            dt = ops.colsum(gf.view(gf.shape[0], -1)).view(ctx.tshape).to(ctx.tdt)
        return g, dt


class ConformerFn(torch.autograd.Function):
    """Stand-alone ConformerModule (transformer.py:550-583): no residual (the block adds it, 673-674)."""

    @staticmethod
    def forward(ctx, mod, x, *params):
        B, N, Dm = x.shape
        xin = _to_f32(x.contiguous()).view(B * N, Dm)
        y, sv = D.conformer_fwd(D.conformer_params(mod), xin, B, N, residual=False)
        ctx.mod, ctx.sv, ctx.dims, ctx.xdt = mod, sv, (B, N, Dm), x.dtype
        ctx.names = [n for n, _ in mod.named_parameters()]
        return _like(y, x.dtype if x.dtype in (F32, BF16) else F32).view(B, N, Dm)

    @staticmethod
    def backward(ctx, g):
        B, N, Dm = ctx.dims
        go = D.GradOut()
        gf = _to_f32(g.contiguous()).view(B * N, Dm)
        dx, _ = D.conformer_bwd(go, D.conformer_params(ctx.mod), ctx.sv, gf, B, N, pre="", residual=False)
        ctx.sv = None
        have = dict(ctx.mod.named_parameters())
        pg = tuple(go.grads[n].view(have[n].shape) for n in ctx.names)
        return (None, _like(dx, ctx.xdt if ctx.xdt in (F32, BF16) else F32).view(B, N, Dm)) + pg


# bf16 shadows of residual-stream gradients handed from block i+1's backward to block i's: keyed by the fp32 gradient's
# address and tagged with the producing layer.  The entry keeps a reference to the fp32 gradient itself, so (a) its address
# cannot be recycled for another tensor while the entry lives and (b) autograd can never add a second consumer's gradient
# into it IN PLACE (InputBuffer only does that to a uniquely owned buffer): with a second consumer of a block output the
# summed gradient is a new tensor at a new address, the lookup misses and the consumer block casts it itself.
_GRAD_SHADOW = {}


class ContextGateFn(torch.autograd.Function):
    """Identity at the entry of ContinuousTransformer for a TRAINABLE cross-attention context.  Its output is consumed by
    this transformer's blocks and by nothing else, which is what lets them share one gradient accumulator (the first block
    backward creates it, the others add into it from the GEMM epilogue and return None); its backward runs after every
    block has contributed and hands the sum to whatever produced the context - where autograd adds the gradients of any
    other consumer of the caller's tensor as usual."""

    @staticmethod
    def forward(ctx, context):
        return context.view_as(context)

    @staticmethod
    def backward(ctx, g):
        return g


def xdt_is_f32(ctx):
    return ctx.dtypes[0] == F32


class TransformerBlockFn(torch.autograd.Function):
    """One fused TransformerBlock (transformer.py:649-695): LN -> self-attn (+RoPE) -> [LN -> cross-attn] -> LN ->
    SwiGLU FF with adaLN modulation/gating and residual adds fused into the GEMM epilogues."""

    @staticmethod
    def forward(ctx, blk, x, context, global_cond, mask8, cmask8, rope, *params):
        B, N, Dm = x.shape
        xin = _to_f32(x.contiguous()).view(B * N, Dm)
        ctxb = None
        S = 0
        if context is not None and blk.cross_attend:
            S = context.shape[1]
            # ContinuousTransformer leaves ONE bf16 copy of a trainable fp32 context on the tensor for all its layers
            pre = getattr(context, "_kalle_bf16", None)
            ctxb = (pre if pre is not None else _to_bf16(context.contiguous())).view(B * S, context.shape[-1])
        gc = _to_f32(global_cond.contiguous()) if global_cond is not None else None
        p = D.block_params(blk)
        # k | v of the conditioning already projected for every layer by the enclosing ContinuousTransformer (dit_ops.ContextKV)
        ckv = getattr(context, "_kalle_ckv", None) if ctxb is not None else None
        y, sv = D.block_fwd(p, xin, ctxb, gc, mask8, cmask8, rope, B, N, S, ckv=ckv)
        ctx.blk, ctx.sv, ctx.ctxb = blk, sv, ctxb
        blk._kalle_last_rows = B * N                    # (the trainer picks its gradient-clearing rule from it)
        # (the accumulator dict of THIS forward pass: a second forward over the same context tensor gets a new one)
        ctx.dctx_state = getattr(context, "_kalle_dctx", None) if context is not None else None
        ctx.masks = (mask8, cmask8, rope)
        ctx.dims = (B, N, S, Dm)
        ctx.dtypes = (x.dtype, context.dtype if context is not None else None,
                      global_cond.dtype if global_cond is not None else None)
        ctx.ctx_shape = context.shape if context is not None else None
        return _like(y, x.dtype if x.dtype in (F32, BF16) else F32).view(B, N, Dm)

    @staticmethod
    def backward(ctx, g):
        blk, sv = ctx.blk, ctx.sv
        mask8, cmask8, rope = ctx.masks
        B, N, S, Dm = ctx.dims
        p = D.block_params(blk)
        gf = _to_f32(g.contiguous()).view(B * N, Dm)
        sinks = getattr(blk, "_kalle_grad_sinks", None)
        go = D.GradOut(sinks, getattr(blk, "_kalle_grad_accumulate", False)) if sinks else D.GradOut()
        go.wgrad_overwrite = bool(sinks) and getattr(blk, "_kalle_wgrad_overwrite", False)
        # bf16 copy of the incoming gradient, if the block above (layer_ix + 1) left one for exactly this tensor
        sh = _GRAD_SHADOW.pop(gf.data_ptr(), None)
        g_bf16 = None
        bias2_done = False
        if sh is not None and sh[0] == blk.layer_ix + 1 and sh[1].shape == gf.shape:
            g_bf16 = sh[1]
            bias2_done = len(sh) > 3 and sh[3]
        pend = getattr(blk, "_kalle_colsum_pending", None)
        if pend is not None:
            blk._kalle_colsum_pending = None
            if not bias2_done or pend[0] != gf.data_ptr():
                # the block above fused the column sums of ITS dx into this block's FF-out bias sink, but this block's output
                # had a second consumer (the gradient that arrived is their sum, another tensor): take that contribution out
                # again - the column-sum pass below covers the whole gradient
                pend[1].sub_(ops.colsum(pend[2]))
                bias2_done = False
        if len(_GRAD_SHADOW) > 8:
            _GRAD_SHADOW.clear()
        # The context feeds every layer: instead of 24 fp32 gradients that autograd adds up one by one, the first backward
        # to run creates the gradient, returns THAT tensor, and the later ones accumulate into it in the GEMM epilogue and
        # return None (the producer of the context runs its backward only after every layer has contributed).
        dst = ctx.dctx_state if (ctx.needs_input_grad[2] and ctx.dtypes[1] == F32) else None
        acc = dst.get("acc") if dst is not None else None
        # the FF-out bias sink of the block below, if the trainer linked the blocks and its sinks take atomic adds this step
        prev = getattr(blk, "_kalle_prev_block", None)
        cs = None
        if (sinks and prev is not None and getattr(prev, "_kalle_grad_accumulate", False)
                and getattr(blk, "_kalle_grad_accumulate", False) and os.environ.get("KALLE_FUSE_BIAS_COLSUM", "1") != "0"):
            cs = (getattr(prev, "_kalle_grad_sinks", None) or {}).get("ff.ff.2.bias")
            if cs is not None and getattr(prev, "global_cond_dim", None):
                cs = None               # (adaLN blocks form their bf16 output gradient in grad_cast, not from this dx)
        dx, dctx, dglobal, go, dxb = D.block_bwd(p, sv, gf, ctx.ctxb, mask8, cmask8, rope, B, N, S, go=go,
                                                 want_dctx=ctx.needs_input_grad[2], g_bf16=g_bf16,
                                                 want_dx_bf16=blk.layer_ix > 0 and xdt_is_f32(ctx), dctx_acc=acc,
                                                 bias2_done=bias2_done, dx_colsum_out=cs.view(-1) if cs is not None else None)
        fused = cs is not None and getattr(go, "dx_colsum_fused", False)
        if dst is not None and dctx is not None:
            if acc is None:
                dst["acc"] = dctx                     # first contribution: this tensor is the gradient
            else:
                dctx = None                           # accumulated in place
        gr = go.grads
        ctx.sv = None
        hook = getattr(blk, "_kalle_on_backward_done", None)
        if hook is not None:
            hook(blk)
        xdt, cdt, gdt = ctx.dtypes
        dx = _like(dx, xdt if xdt in (F32, BF16) else F32).view(B, N, Dm)
        if dxb is not None:
            _GRAD_SHADOW[dx.data_ptr()] = (blk.layer_ix, dxb, dx, fused)
            if fused:
                prev._kalle_colsum_pending = (dx.data_ptr(), cs.view(-1), dxb)
        if dctx is not None and ctx.needs_input_grad[2]:
            dctx = dctx.view(ctx.ctx_shape).to(cdt)
        else:
            dctx = None
        if dglobal is not None and ctx.needs_input_grad[3]:
            dglobal = dglobal.to(gdt)
        else:
            dglobal = None
        names = blk._kalle_param_names
        pg = tuple(gr.get(n) for n in names)
        if getattr(blk, "conformer", None) is not None:      # conv weights are [out, in / groups, k]: the kernels wrote matrices
            have = dict(blk.named_parameters())
            pg = tuple(g_ if g_ is None else g_.view(have[n].shape) for n, g_ in zip(names, pg))
        return (None, dx, dctx, dglobal, None, None, None) + pg


def transformer_block(blk, x, context=None, global_cond=None, mask=None, context_mask=None, rotary_pos_emb=None):
    names = getattr(blk, "_kalle_param_names", None)
    if names is None:
        have = dict(blk.named_parameters())
        names = tuple(n for n in D.BLOCK_PARAM_ORDER if n in have)
        extra = set(have) - set(names)
        if extra:
            raise NotImplementedError(f"TransformerBlock options not supported by the HIP path: {sorted(extra)}")
        blk._kalle_param_names = names
    have = dict(blk.named_parameters())
    rope = None
    if rotary_pos_emb is not None:
        freqs = rotary_pos_emb[0] if isinstance(rotary_pos_emb, (tuple, list)) else rotary_pos_emb
        rope = D.rope_tables(freqs, x.shape[1])
    return TransformerBlockFn.apply(blk, x, context, global_cond, _mask8(mask), _mask8(context_mask), rope,
                                    *[have[n] for n in names])


class AttentionFn(torch.autograd.Function):
    """Stand-alone Attention module (transformer.py:396-547): projections + fused attention + to_out."""

    @staticmethod
    def forward(ctx, mod, x, context, mask8, cmask8, rope, causal, *params):
        B, N, Dm = x.shape
        H = mod.num_heads
        ctx.causal = causal
        h = _to_bf16(x.contiguous()).view(B * N, Dm)
        cross = hasattr(mod, "to_q")
        odt = F32 if x.dtype == F32 else BF16
        if cross:
            kv_in = context if context is not None else x
            S = kv_in.shape[1]
            cb = _to_bf16(kv_in.contiguous()).view(B * S, kv_in.shape[-1])
            km = cmask8 if context is not None else mask8
            out, sv = D.cross_attn_fwd(h, cb, D.bf16_of(mod.to_q.weight), D.bf16_of(mod.to_kv.weight),
                                       D.bf16_of(mod.to_out.weight), B, N, S, H, km, out_dtype=odt, row_mask=mask8,
                                       qkn=D.qk_norm_params(mod), causal=causal)
            ctx.extra = (cb, S, km, context is not None)
        else:
            out, sv = D.self_attn_fwd(h, D.bf16_of(mod.to_qkv.weight), D.bf16_of(mod.to_out.weight), B, N, H, rope,
                                      mask8, out_dtype=odt, qkn=D.qk_norm_params(mod), causal=causal)
            ctx.extra = None
        ctx.mod, ctx.h, ctx.sv, ctx.cross = mod, h, sv, cross
        ctx.meta = (B, N, Dm, H, mask8, rope, x.dtype, context.dtype if context is not None else None,
                    context.shape if context is not None else None)
        return out.view(B, N, Dm)

    @staticmethod
    def backward(ctx, g):
        mod = ctx.mod
        B, N, Dm, H, mask8, rope, xdt, cdt, cshape = ctx.meta
        gf = g.contiguous().view(B * N, Dm)
        if mask8 is not None:
            gb, _ = ops.grad_cast(_to_f32(gf), B, N, row_mask=mask8)
        else:
            gb = _to_bf16(gf)
        if ctx.cross:
            cb, S, km, has_ctx = ctx.extra
            go = D.GradOut()
            dh, dctx = D.cross_attn_bwd(go, gb, ctx.h, cb, ctx.sv, D.bf16_of(mod.to_q.weight),
                                        D.bf16_of(mod.to_kv.weight), D.bf16_of(mod.to_out.weight), B, N, S, H, km,
                                        pre="", qkn=D.qk_norm_params(mod), causal=ctx.causal)
            dwq, dwkv, dwo = go.grads["to_q.weight"], go.grads["to_kv.weight"], go.grads["to_out.weight"]
            dx = _like(dh, xdt if xdt in (F32, BF16) else F32).view(B, N, Dm)
            dc = None
            if has_ctx:
                dc = dctx.view(cshape).to(cdt)
            else:
                dx = dx + dctx.view(B, N, Dm).to(dx.dtype)  # kv_input == x: tape-level add of two input gradients
            return (None, dx, dc, None, None, None, None, dwq, dwkv, dwo) + _qk_norm_grads(mod, go)
        go = D.GradOut()
        dh = D.self_attn_bwd(go, gb, ctx.h, ctx.sv, D.bf16_of(mod.to_qkv.weight), D.bf16_of(mod.to_out.weight),
                             B, N, H, rope, mask8, pre="", qkn=D.qk_norm_params(mod), causal=ctx.causal)
        dwqkv, dwo = go.grads["to_qkv.weight"], go.grads["to_out.weight"]
        dx = _like(dh, xdt if xdt in (F32, BF16) else F32).view(B, N, Dm)
        return (None, dx, None, None, None, None, None, dwqkv, dwo) + _qk_norm_grads(mod, go)


def _qk_norm_grads(mod, go):
    """gradients of Attention(qk_norm="ln")'s two LayerNorm(64), in the order Attention.forward appends the parameters"""
    if getattr(mod, "qk_norm", "none") != "ln":
        return ()
    return tuple(go.grads[n] for n in ("q_norm.weight", "q_norm.bias", "k_norm.weight", "k_norm.bias"))


class FeedForwardFn(torch.autograd.Function):
    """Stand-alone FeedForward (transformer.py:221-269), SwiGLU variant."""

    @staticmethod
    def forward(ctx, mod, x, w1, b1, w2, b2):
        shp = x.shape
        h = _to_bf16(x.contiguous()).view(-1, shp[-1])
        odt = F32 if x.dtype == F32 else BF16
        out, sv = D.ff_fwd(h, D.bf16_of(w1), D.f32_of(b1) if b1 is not None else None, D.bf16_of(w2),
                           D.f32_of(b2) if b2 is not None else None, 1, out_dtype=odt)
        ctx.h, ctx.sv, ctx.ws, ctx.meta = h, sv, (w1, w2), (shp, x.dtype, b1 is not None)
        return out.view(*shp[:-1], w2.shape[0])

    @staticmethod
    def backward(ctx, g):
        shp, xdt, has_bias = ctx.meta
        w1, w2 = ctx.ws
        gb = _to_bf16(g.contiguous().view(-1, w2.shape[0]))
        go = D.GradOut()
        dh = D.ff_bwd(go, gb, ctx.h, ctx.sv, D.bf16_of(w1), D.bf16_of(w2), want_bias=has_bias, pre="")
        dw1, db1 = go.grads["0.proj.weight"], go.grads.get("0.proj.bias")
        dw2, db2 = go.grads["2.weight"], go.grads.get("2.bias")
        return None, _like(dh, xdt if xdt in (F32, BF16) else F32).view(shp), dw1, db1, dw2, db2


class MSELossFn(torch.autograd.Function):
    """training/losses/losses.py:53-69"""

    @staticmethod
    def forward(ctx, output, target, mask, weight):
        loss, diff = ops.mse_loss(_to_f32(output), _to_f32(target), mask, weight=weight, want_grad=True)
        ctx.save_for_backward(diff)
        ctx.odt = output.dtype
        return loss

    @staticmethod
    def backward(ctx, gl):
        (diff,) = ctx.saved_tensors
        g = diff * gl
        # (the target carries a gradient only when the latents do: a pretransform trained with enable_grad)
        return g.to(ctx.odt), (-g if ctx.needs_input_grad[1] else None), None, None


class DiffuseFn(torch.autograd.Function):
    """x_t, target of training/diffusion.py:365-379 (kalle_diffuse_fwd); differentiable w.r.t. the latents for the
    enable_grad pretransform case: v: dx = alpha g_xt - sigma g_tgt; rectified flow: dx = (1 - t) g_xt - g_tgt  ([B, C, T]
    broadcast arithmetic, host-side glue like the reference's own)."""

    @staticmethod
    def forward(ctx, x, noise, t, objective):
        ctx.save_for_backward(t)
        ctx.objective = objective
        return ops.diffuse_fwd(x, noise, t, objective)

    @staticmethod
    def backward(ctx, g_xt, g_tgt):
        (t,) = ctx.saved_tensors
        import math
        if ctx.objective == "v":
            a, s = torch.cos(t * math.pi / 2)[:, None, None], torch.sin(t * math.pi / 2)[:, None, None]
            dx = g_xt * a - g_tgt * s
        else:
            dx = g_xt * (1 - t)[:, None, None] - g_tgt
        return dx, None, None, None
