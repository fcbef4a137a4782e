"""Forward / backward of the DiT building blocks expressed as sequences of C-ABI kernel launches.

Everything here is "manual autograd": each *_fwd returns the tensors its *_bwd needs, each *_bwd returns input
gradients and hands parameter gradients to a GradOut.  The torch.autograd.Function shims in functional.py and the
data-parallel trainer in engine.py are both thin layers over these functions, so the drop-in modules and the
benchmarked train step run exactly the same kernels.

Dtype policy: residual stream fp32, GEMM operands bf16, GEMM accumulation fp32, parameter gradients
fp32.  Reference: stable_audio_tools/models/transformer.py (line numbers per function).
"""
from types import SimpleNamespace

import torch

from . import ops

BF16 = torch.bfloat16
F32 = torch.float32
import os as _os
GROUP_WGRAD = _os.environ.get("KALLE_GROUP_WGRAD", "1") != "0"     # all weight gradients of a block in one launch


def bf16_of(p):
    """bf16 compute copy of an fp32 parameter, re-cast only when the parameter changed (or pinned by the engine)."""
    pinned = getattr(p, "_kalle_bf16_pinned", None)
    if pinned is not None:
        return pinned
    c = getattr(p, "_kalle_bf16", None)
    if c is None or c[0] != p._version or c[1].device != p.device:
        c = (p._version, ops.cast(p.detach().float(), BF16))
        p._kalle_bf16 = c
    return c[1]


def f32_of(p):
    d = p.detach()
    return d if d.dtype == F32 else d.float()


def rope_tables(freqs, n=None):
    """freqs: [N, rot] as produced by RotaryEmbedding (cat(f, f)); returns cos/sin [n, rot/2] fp32 (last n positions) for the
    kernel.  The tables ride on the freqs tensor: ContinuousTransformer hands the same tensor to all of its layers, so the two
    trigonometric passes run once per forward instead of once per layer."""
    n = freqs.shape[0] if n is None else n
    hit = getattr(freqs, "_kalle_rope", None)
    if hit is not None and hit[0] == (n, freqs._version):
        return hit[1]
    half = freqs.shape[-1] // 2
    f = freqs[-n:, :half].float()
    tab = (f.cos().contiguous(), f.sin().contiguous())
    try:
        freqs._kalle_rope = ((n, freqs._version), tab)
    except (AttributeError, RuntimeError):
        pass
    return tab


class GradOut:
    """Destination of parameter gradients.  Without sinks every gradient is a fresh fp32 tensor in `.grads`
    (autograd mode).  With `sinks` (name -> fp32 view into the trainer's flat gradient bucket) the wgrad GEMMs and
    column sums write straight into the bucket (accumulating on later micro-batches) and `.grads[name]` is None."""

    def __init__(self, sinks=None, accumulate=False, prefix=""):
        self.sinks = sinks or {}
        self.accumulate = accumulate
        self.prefix = prefix
        self.grads = {}
        self.deferred = None            # list of (name, dy, x) while a block's weight gradients are being collected
        self.wgrad_overwrite = False    # trainer, first micro-batch: weight-gradient sinks are NOT pre-cleared - write, don't add
        self.touched = set()            # sink names some kernel of this backward pass has written (or added into)
        self.colsum_fused = False       # the last ln() call added the column sums of its dx into the sink it was handed

    def _sink(self, name):
        s = self.sinks.get(self.prefix + name)
        if s is not None:
            self.touched.add(self.prefix + name)
        return s

    def _sink2d(self, name, dy, x):
        """the sink of a weight gradient as the [N, K] matrix the GEMM writes (a 1 x 1 conv weight is stored [N, K, 1])"""
        s = self._sink(name)
        if s is not None and s.dim() != 2:
            s = s.view(dy.shape[1], x.shape[1])
        return s

    def zero_unwritten(self):
        """Sinks that NO kernel of this backward pass wrote - the cross-attention of a block called with context=None
        (transformer.py:684-693 skips it), a sub-module that only runs in a later micro-batch - must read as zero when the
        pass is the one that (re)defines the buffers: in overwrite mode the matrices are not pre-cleared (engine.backward
        clears only the vectors at the head of the bucket), without accumulation nothing is.  Otherwise last step's values
        would reach the all-reduce and Adam."""
        if not self.sinks:
            return
        for full, s in self.sinks.items():
            if full in self.touched:
                continue
            if not self.accumulate or (self.wgrad_overwrite and s.dim() >= 2):
                s.zero_()

    def defer(self):
        """collect the weight gradients of a block and launch them together in flush() (kalle_gemm_wgrad_group)"""
        if GROUP_WGRAD:
            self.deferred = []
        return self

    def flush(self):
        items, self.deferred = self.deferred, None
        if not items:
            return
        probs = []
        over = self.wgrad_overwrite or not self.accumulate or not self.sinks
        # a gradient over far fewer rows than the others (the adaLN modulation weights see one row per clip) would add a round of
        # 4-K-tile workgroups to the grouped launch and, being too short to slice, switch off the tail levelling for the whole
        # group (adaLN step 255 -> 265 ms): it keeps its own GEMM
        most = max(dy.shape[0] for _, dy, _ in items)
        for name, dy, x in [it for it in items if it[1].shape[0] * 8 < most]:
            s = self._sink2d(name, dy, x)
            if s is not None:
                ops.gemm(dy, x, a_kmajor=True, b_kmajor=True, out=s, accumulate=not over)
                self.grads[name] = None
            else:
                self.grads[name] = ops.gemm(dy, x, a_kmajor=True, b_kmajor=True, out_dtype=F32)
        items = [it for it in items if it[1].shape[0] * 8 >= most]
        for name, dy, x in items:
            s = self._sink2d(name, dy, x)
            if s is not None:
                out = s
                self.grads[name] = None
            else:
                out = self.grads[name] = torch.empty((dy.shape[1], x.shape[1]), device=dy.device, dtype=F32)
            probs.append((dy, x, out))
        if not ops.gemm_wgrad_group(probs, overwrite=over):    # shapes outside the grouped kernel: one GEMM each
            for dy, x, out in probs:
                ops.gemm(dy, x, a_kmajor=True, b_kmajor=True, out=out, accumulate=not over)

    def wgrad(self, name, dy, x):
        """dW[N,K] = dy[M,N]^T @ x[M,K]"""
        if (self.deferred is not None and dy.shape[0] % 8 == 0 and dy.stride(-1) == 1 and dy.stride(0) % 8 == 0
                and dy.data_ptr() % 16 == 0 and x.is_contiguous()):        # (dy may be a column slice: the kernels take ld)
            self.deferred.append((name, dy, x))
            return
        s = self._sink2d(name, dy, x)
        if s is not None:
            ops.gemm(dy, x, a_kmajor=True, b_kmajor=True, out=s, accumulate=self.accumulate and not self.wgrad_overwrite)
            self.grads[name] = None
        else:
            self.grads[name] = ops.gemm(dy, x, a_kmajor=True, b_kmajor=True, out_dtype=F32)

    def colsum(self, name, src):
        s = self._sink(name)
        if s is not None:
            ops.colsum(src, out=s.view(-1), accumulate=self.accumulate)
            self.grads[name] = None
        else:
            self.grads[name] = ops.colsum(src)

    def ln(self, name, dy, x, gamma, mean, rstd, want_bf16=False, **kw):
        """LayerNorm backward: returns (dx fp32, dx bf16 | None); dgamma goes to the sink / grads."""
        s = self._sink(name)
        if kw.get("dx_colsum_out") is not None and not (s is not None and self.accumulate and want_bf16
                                                        and _os.environ.get("KALLE_LN_ATOMIC", "1") != "0"):
            kw = dict(kw, dx_colsum_out=None)       # (only the atomic trainer path carries the fused column sums)
        self.colsum_fused = kw.get("dx_colsum_out") is not None
        dxb = torch.empty(x.shape, device=x.device, dtype=BF16) if want_bf16 else None
        dx, dgamma, _ = ops.layernorm_bwd(dy, x, gamma, mean, rstd, dgamma_out=s.view(-1) if s is not None else None,
                                          accumulate=self.accumulate if s is not None else False, dx_bf16=dxb, **kw)
        self.grads[name] = None if s is not None else dgamma
        return dx, dxb

    def rms(self, name, dy, x, scale, rrms, dres=None, dx_bf16=None):
        """RMSNorm backward (+ residual-stream gradient, + bf16 copy): returns dx fp32; dscale to the sink / grads."""
        s = self._sink(name)
        dx, dscale = ops.rmsnorm_bwd(dy, x, scale, rrms, dres=dres, dx_bf16=dx_bf16,
                                     dscale_out=s.view(-1) if s is not None else None,
                                     accumulate=self.accumulate if s is not None else False)
        self.grads[name] = None if s is not None else dscale
        return dx

    def bias_acc(self, name, n, device):
        """fp32 [n] buffer a kernel atomically ADDS a bias gradient into (zeroed first unless accumulating)."""
        s = self._sink(name)
        if s is not None:
            if not self.accumulate:
                s.zero_()
            self.grads[name] = None
            return s.view(-1)
        t = torch.zeros(n, device=device, dtype=F32)
        self.grads[name] = t
        return t


def dgrad(dy, w, **kw):
    """dx[M,K] = dy[M,N] @ w[N,K]"""
    return ops.gemm(dy, w, b_kmajor=True, **kw)


# ------------------------------------------------------------------------------------------------ attention
def qk_norm_params(attn):
    """Attention(qk_norm=...) (transformer.py:303-307): None, or (mode, q gamma, q beta, k gamma, k beta) - mode 1 "l2" (no
    parameters), 2 "ln" (two LayerNorm(64))"""
    kind = getattr(attn, "qk_norm", "none")
    if kind == "none":
        return None
    if kind == "l2":
        return (1, None, None, None, None)
    return (2, f32_of(attn.q_norm.weight), f32_of(attn.q_norm.bias), f32_of(attn.k_norm.weight), f32_of(attn.k_norm.bias))


def _qk_norm_bwd(go, pre, qkn, which, x, ldx, x_off, stat, g, dx, lddx, dx_off, rows, heads):
    """head-norm backward of q (`which` = "q_norm") or k; LayerNorm scale / bias gradients go to the sinks (added atomically)"""
    mode = qkn[0]
    gamma = dga = dbe = None
    if mode == 2:
        gamma = qkn[1] if which == "q_norm" else qkn[3]
        dga = go.bias_acc(pre + which + ".weight", 64, x.device)
        dbe = go.bias_acc(pre + which + ".bias", 64, x.device)
    ops.head_norm_bwd(x, ldx, x_off, stat, g, dx, lddx, dx_off, rows, heads, mode, gamma, dga, dbe)


def self_attn_fwd(h, wqkv, wo, B, N, H, rope, mask8, residual=None, gate=None, out_dtype=F32, qkn=None, causal=False):
    """transformer.py:419-420, 430-444, 500/514-530, 541-545.  h: bf16 [B*N, D].  qkn: qk_norm_params()"""
    D = H * 64
    causal = bool(causal) and N > 1                      # :468-469
    qkv = ops.gemm(h, wqkv)
    nrm = None
    if qkn is not None:     # :422-428, before the rotary embedding (which the attention kernel applies)
        qn, qs = ops.head_norm_fwd(qkv, 3 * D, 0, B * N, H, qkn[0], qkn[1], qkn[2])
        kn, ks = ops.head_norm_fwd(qkv, 3 * D, D, B * N, H, qkn[0], qkn[3], qkn[4])
        nrm = (qn, qs, kn, ks)
        ao, lse = ops.attention_fwd(qn, kn, qkv, ldq=D, q_off=0, ldk=D, k_off=0, ldv=3 * D, v_off=2 * D,
                                    B=B, H=H, Hkv=H, Nq=N, Nk=N, rope=rope, key_mask=mask8, causal=causal)
    else:
        ao, lse = ops.attention_fwd(qkv, qkv, qkv, ldq=3 * D, q_off=0, ldk=3 * D, k_off=D, ldv=3 * D, v_off=2 * D,
                                    B=B, H=H, Hkv=H, Nq=N, Nk=N, rope=rope, key_mask=mask8, causal=causal)
    out = ops.gemm(ao.view(B * N, D), wo, out_dtype=out_dtype, residual=residual, gate=gate, rows_per_batch=N,
                   row_mask=mask8)
    return out, (qkv, ao, lse, nrm)


def self_attn_bwd(go, gb, h, saved, wqkv, wo, B, N, H, rope, mask8, pre="self_attn.", qkn=None, causal=False):
    """gb: bf16 [B*N, D] gradient w.r.t. the to_out GEMM result. Returns dh (bf16)."""
    qkv, ao, lse, nrm = saved
    D = H * 64
    causal = bool(causal) and N > 1
    go.wgrad(pre + "to_out.weight", gb, ao.view(B * N, D))
    dao = dgrad(gb, wo)
    dqkv = torch.empty_like(qkv)
    if nrm is not None:
        qn, qs, kn, ks = nrm
        dqn, dkn = torch.empty_like(qn), torch.empty_like(kn)
        ops.attention_bwd(qn, kn, qkv, ao, dao, lse, dqn, dkn, dqkv, ldq=D, q_off=0, ldk=D, k_off=0,
                          ldv=3 * D, v_off=2 * D, B=B, H=H, Hkv=H, Nq=N, Nk=N, rope=rope, key_mask=mask8, causal=causal)
        _qk_norm_bwd(go, pre, qkn, "q_norm", qkv, 3 * D, 0, qs, dqn, dqkv, 3 * D, 0, B * N, H)
        _qk_norm_bwd(go, pre, qkn, "k_norm", qkv, 3 * D, D, ks, dkn, dqkv, 3 * D, D, B * N, H)
    else:
        ops.attention_bwd(qkv, qkv, qkv, ao, dao, lse, dqkv, dqkv, dqkv, ldq=3 * D, q_off=0, ldk=3 * D, k_off=D,
                          ldv=3 * D, v_off=2 * D, B=B, H=H, Hkv=H, Nq=N, Nk=N, rope=rope, key_mask=mask8, causal=causal)
    go.wgrad(pre + "to_qkv.weight", dqkv, h)
    return dgrad(dqkv, wqkv)


class ContextKV:
    """The k | v projections of the conditioning for ALL layers of a ContinuousTransformer in one GEMM: the context is the same
    tensor for every layer (transformer.py:800-802), only the weights differ - `ctx @ [Wkv_0; Wkv_1; ...]^T` instead of one
    (tokens x 1536 x 768) GEMM per layer, and on the way back ONE `dctx = dkv_all @ [Wkv_0; ...]` instead of a read-add-store
    of the fp32 context gradient per layer.  Layer l reads / writes columns [l * 2 Dc, (l + 1) * 2 Dc) in place (the attention
    kernels take a leading dimension and a column offset)."""

    def __init__(self, ctxb, weights, w_all=None):
        self.L = len(weights)
        self.Dc = ctxb.shape[-1]
        # [L * 2 Dc, Dc] bf16: this forward's weights, kept for the backward (the trainer may update a layer's bf16 mirror
        # before the context gradient is formed at the end of the backward pass)
        self.w_all = w_all if w_all is not None else torch.cat(weights, dim=0)
        self.ld = self.w_all.shape[0]
        self.kv = ops.gemm(ctxb, self.w_all)                   # [B * S, L * 2 Dc]
        self.dkv = None
        self.left = self.L                                     # layers whose backward has not run yet

    @classmethod
    def preprojected(cls, kv, L):
        """forward-only view of projections computed elsewhere (a sampling loop's constant conditioning)"""
        o = cls.__new__(cls)
        o.L, o.kv, o.ld = L, kv, kv.shape[1]
        o.Dc = kv.shape[1] // (2 * L)
        o.w_all, o.dkv, o.left = None, None, L
        return o

    def grad_buffer(self):
        if self.dkv is None:
            self.dkv = torch.empty_like(self.kv)
        return self.dkv

    def layer_done(self, want_dctx):
        """called by every layer's backward after it wrote its dk | dv columns; the last one returns the context gradient"""
        self.left -= 1
        if self.left > 0 or not want_dctx:
            return None
        return ops.gemm(self.dkv, self.w_all, b_kmajor=True, out_dtype=F32)


def cross_attn_fwd(h, ctx, wq, wkv, wo, B, N, S, H, cmask8, residual=None, out_dtype=F32, row_mask=None, qkn=None,
                   ckv=None, layer_ix=0, causal=False):
    """transformer.py:411-416, 505-508 (GQA), 541.  h: bf16 [B*N, D]; ctx: bf16 [B*S, Dc].  ckv: ContextKV of the enclosing
    ContinuousTransformer (then this layer's k | v are columns of ckv.kv and no projection runs here)."""
    D = H * 64
    Dc = ctx.shape[-1]
    Hkv = Dc // 64
    causal = bool(causal) and N > 1                      # :468-469; query r sees keys c <= r + S - N (create_causal_mask, :32)
    q = ops.gemm(h, wq)
    if ckv is not None and qkn is None:
        off = layer_ix * 2 * Dc
        co, lse = ops.attention_fwd(q, ckv.kv, ckv.kv, ldq=D, q_off=0, ldk=ckv.ld, k_off=off, ldv=ckv.ld, v_off=off + Dc,
                                    B=B, H=H, Hkv=Hkv, Nq=N, Nk=S, key_mask=cmask8, causal=causal)
        out = ops.gemm(co.view(B * N, D), wo, out_dtype=out_dtype, residual=residual, row_mask=row_mask)
        return out, (q, (ckv, off), co, lse, None)
    kv = ops.gemm(ctx, wkv)
    nrm = None
    if qkn is not None:
        qn, qs = ops.head_norm_fwd(q, D, 0, B * N, H, qkn[0], qkn[1], qkn[2])
        kn, ks = ops.head_norm_fwd(kv, 2 * Dc, 0, B * S, Hkv, qkn[0], qkn[3], qkn[4])
        nrm = (qn, qs, kn, ks)
        co, lse = ops.attention_fwd(qn, kn, kv, ldq=D, q_off=0, ldk=Dc, k_off=0, ldv=2 * Dc, v_off=Dc, B=B, H=H,
                                    Hkv=Hkv, Nq=N, Nk=S, key_mask=cmask8, causal=causal)
    else:
        co, lse = ops.attention_fwd(q, kv, kv, ldq=D, q_off=0, ldk=2 * Dc, k_off=0, ldv=2 * Dc, v_off=Dc, B=B, H=H,
                                    Hkv=Hkv, Nq=N, Nk=S, key_mask=cmask8, causal=causal)
    out = ops.gemm(co.view(B * N, D), wo, out_dtype=out_dtype, residual=residual, row_mask=row_mask)
    return out, (q, kv, co, lse, nrm)


def cross_attn_bwd(go, gb, h, ctx, saved, wq, wkv, wo, B, N, S, H, cmask8, dctx_acc=None, want_dctx=True,
                   pre="cross_attn.", qkn=None, causal=False):
    """Returns dh (bf16), dctx (fp32 [B*S, Dc]; accumulated into dctx_acc when given; None if not wanted)."""
    q, kv, co, lse, nrm = saved
    D = H * 64
    Dc = ctx.shape[-1]
    Hkv = Dc // 64
    causal = bool(causal) and N > 1
    go.wgrad(pre + "to_out.weight", gb, co.view(B * N, D))
    dco = dgrad(gb, wo)
    dq = torch.empty_like(q)
    if isinstance(kv, tuple):            # k | v live in the ContextKV of the enclosing transformer
        ckv, off = kv
        dkv_all = ckv.grad_buffer()
        ops.attention_bwd(q, ckv.kv, ckv.kv, co, dco, lse, dq, dkv_all, dkv_all, ldq=D, q_off=0, ldk=ckv.ld, k_off=off,
                          ldv=ckv.ld, v_off=off + Dc, B=B, H=H, Hkv=Hkv, Nq=N, Nk=S, key_mask=cmask8, causal=causal)
        go.wgrad(pre + "to_q.weight", dq, h)
        dh = dgrad(dq, wq)
        go.wgrad(pre + "to_kv.weight", dkv_all[:, off:off + 2 * Dc], ctx)
        return dh, ckv.layer_done(want_dctx)
    dkv = torch.empty_like(kv)
    if nrm is not None:
        qn, qs, kn, ks = nrm
        dkn = torch.empty_like(kn)
        ops.attention_bwd(qn, kn, kv, co, dco, lse, dq, dkn, dkv, ldq=D, q_off=0, ldk=Dc, k_off=0, ldv=2 * Dc,
                          v_off=Dc, B=B, H=H, Hkv=Hkv, Nq=N, Nk=S, key_mask=cmask8, causal=causal)
        _qk_norm_bwd(go, pre, qkn, "q_norm", q, D, 0, qs, dq, dq, D, 0, B * N, H)          # in place over dq
        _qk_norm_bwd(go, pre, qkn, "k_norm", kv, 2 * Dc, 0, ks, dkn, dkv, 2 * Dc, 0, B * S, Hkv)
    else:
        ops.attention_bwd(q, kv, kv, co, dco, lse, dq, dkv, dkv, ldq=D, q_off=0, ldk=2 * Dc, k_off=0, ldv=2 * Dc,
                          v_off=Dc, B=B, H=H, Hkv=Hkv, Nq=N, Nk=S, key_mask=cmask8, causal=causal)
    go.wgrad(pre + "to_q.weight", dq, h)
    dh = dgrad(dq, wq)
    go.wgrad(pre + "to_kv.weight", dkv, ctx)
    dctx = None
    if want_dctx:
        if dctx_acc is not None:
            dctx = dgrad(dkv, wkv, out=dctx_acc, accumulate=True)
        else:
            dctx = dgrad(dkv, wkv, out_dtype=F32)
    return dh, dctx


# ------------------------------------------------------------------------------------------------ feed-forward
def ff_fwd(h, w1, b1, w2, b2, N, residual=None, gate=None, out_dtype=F32):
    """transformer.py:216-219 (GLU proj + x*silu(gate)), 252 (linear_out)."""
    M, inner = h.shape[0], w1.shape[0] // 2
    hf = torch.empty((M, 2 * inner), device=h.device, dtype=BF16)
    act = torch.empty((M, inner), device=h.device, dtype=BF16)
    if ops.gemm(h, w1, bias=b1, out=hf, glu_mode=1, glu_inner=inner, glu_aux=act) is None:   # fused GEMM + SwiGLU
        ops.gemm(h, w1, bias=b1, out=hf)
        act = ops.swiglu_fwd(hf)
    out = ops.gemm(act, w2, bias=b2, out_dtype=out_dtype, residual=residual, gate=gate, rows_per_batch=N)
    return out, (hf, act)


def ff_bwd(go, gb, h, saved, w1, w2, want_bias=True, pre="ff.ff.", bias2_done=False):
    """bias2_done: the column sums of gb are already in the FF-out bias sink (added by the LayerNorm backward that produced gb,
    kalle_layernorm_bwd_colsum)"""
    hf, act = saved
    go.wgrad(pre + "2.weight", gb, act)
    if want_bias and bias2_done:
        go._sink(pre + "2.bias")
        go.grads[pre + "2.bias"] = None
    elif want_bias:
        go.colsum(pre + "2.bias", gb)
    db1 = go.bias_acc(pre + "0.proj.bias", hf.shape[-1], hf.device) if want_bias else None
    dhf = torch.empty_like(hf)
    # dgrad GEMM with the SwiGLU backward and the GLU bias gradient fused into its epilogue (d(act) never hits HBM)
    if ops.gemm(gb, w2, b_kmajor=True, out=dhf, N=hf.shape[-1] // 2, glu_mode=2, glu_inner=hf.shape[-1] // 2,
                glu_aux=hf, glu_dbias=db1) is None:
        dact = dgrad(gb, w2)
        dhf = ops.swiglu_bwd(dact, hf, db1)
    go.wgrad(pre + "0.proj.weight", dhf, h)
    return dgrad(dhf, w1)


# ------------------------------------------------------------------------------------------------ conformer module
def conformer_params(mod):
    """kernel-ready views of a ConformerModule (transformer.py:550-567); the 1 x 1 convolutions are plain matrices"""
    c = SimpleNamespace()
    D = mod.dim
    c.g_in, c.beta_in = f32_of(mod.in_norm.gamma), _beta(mod.in_norm)
    c.w_pw1 = bf16_of(mod.pointwise_conv.weight).view(D, D)
    c.w_glu = bf16_of(mod.glu.proj.weight)
    c.b_glu = f32_of(mod.glu.proj.bias)
    c.w_dw = f32_of(mod.depthwise_conv.weight).view(D, -1)
    c.pad = int(mod.depthwise_conv.padding[0])
    c.g_mid, c.beta_mid = f32_of(mod.mid_norm.gamma), _beta(mod.mid_norm)
    c.w_pw2 = bf16_of(mod.pointwise_conv_2.weight).view(D, D)
    return c


def conformer_fwd(c, x, B, N, residual=True):
    """transformer.py:569-583 on token-major rows (the three rearranges b n d <-> b d n around the convolutions disappear: a
    1 x 1 conv over channels is a GEMM over rows, the depthwise conv runs along the rows of a batch element).
    x: fp32 [B*N, D].  Returns (x + conformer(x) when `residual` else conformer(x), saved)."""
    D = x.shape[-1]
    sv = SimpleNamespace(x=x)
    sv.h0, sv.mean0, sv.rstd0 = ops.layernorm_fwd(x, c.g_in, c.beta_in)
    sv.p1 = ops.gemm(sv.h0, c.w_pw1)
    sv.hf = torch.empty((B * N, 2 * D), device=x.device, dtype=BF16)
    sv.act = torch.empty((B * N, D), device=x.device, dtype=BF16)
    if ops.gemm(sv.p1, c.w_glu, bias=c.b_glu, out=sv.hf, glu_mode=1, glu_inner=D, glu_aux=sv.act) is None:
        ops.gemm(sv.p1, c.w_glu, bias=c.b_glu, out=sv.hf)
        sv.act = ops.swiglu_fwd(sv.hf)
    sv.cv = ops.dwconv1d(sv.act, c.w_dw, B, N, F32, c.pad)
    sv.m, sv.mean1, sv.rstd1 = ops.layernorm_fwd(sv.cv, c.g_mid, c.beta_mid)
    sv.s = ops.silu_fwd(sv.m)
    y = ops.gemm(sv.s, c.w_pw2, out_dtype=F32, residual=x if residual else None)
    return y, sv


def conformer_bwd(go, c, sv, g, B, N, pre="conformer.", residual=True, gb=None, want_bf16=False):
    """g: fp32 [B*N, D] gradient of the module's output (gb: the same rounded to bf16, if the caller has it).  Returns
    (dx fp32 - including g itself when `residual` - , dx bf16 | None)."""
    D = g.shape[-1]
    K = c.w_dw.shape[1]
    gb = gb if gb is not None else ops.cast(g, BF16)
    go.wgrad(pre + "pointwise_conv_2.weight", gb, sv.s)
    ds = dgrad(gb, c.w_pw2)
    dm = ops.silu_bwd(ds, sv.m)
    _, dcvb = go.ln(pre + "mid_norm.gamma", dm, sv.cv, c.g_mid, sv.mean1, sv.rstd1, want_bf16=True)
    dact = ops.dwconv1d(dcvb, c.w_dw, B, N, BF16, K - 1 - c.pad, flip=True)
    ops.dwconv1d_wgrad(dcvb, sv.act, go.bias_acc(pre + "depthwise_conv.weight", D * K, g.device).view(D, K), B, N, c.pad)
    db = go.bias_acc(pre + "glu.proj.bias", 2 * D, g.device)
    dhf = ops.swiglu_bwd(dact, sv.hf, db)
    go.wgrad(pre + "glu.proj.weight", dhf, sv.p1)
    dp1 = dgrad(dhf, c.w_glu)
    go.wgrad(pre + "pointwise_conv.weight", dp1, sv.h0)
    dh0 = dgrad(dp1, c.w_pw1)
    return go.ln(pre + "in_norm.gamma", dh0, sv.x, c.g_in, sv.mean0, sv.rstd0, dres=g if residual else None,
                 want_bf16=want_bf16)


# ------------------------------------------------------------------------------------------------ transformer block
def _beta(norm):
    b = getattr(norm, "beta", None)
    if isinstance(b, torch.nn.Parameter):
        return f32_of(b)
    return None  # zero buffer (transformer.py:188): adding it is a no-op


def block_params(blk):
    """Gather the kernel-ready views of one TransformerBlock's parameters (bf16 weights, fp32 vectors)."""
    p = SimpleNamespace()
    p.g1 = f32_of(blk.pre_norm.gamma)
    p.beta1 = _beta(blk.pre_norm)
    p.wqkv = bf16_of(blk.self_attn.to_qkv.weight)
    p.wo = bf16_of(blk.self_attn.to_out.weight)
    p.cross = blk.cross_attend
    if p.cross:
        p.g2 = f32_of(blk.cross_attend_norm.gamma)
        p.beta2 = _beta(blk.cross_attend_norm)
        p.wq = bf16_of(blk.cross_attn.to_q.weight)
        p.wkv = bf16_of(blk.cross_attn.to_kv.weight)
        p.wo2 = bf16_of(blk.cross_attn.to_out.weight)
    p.g3 = f32_of(blk.ff_norm.gamma)
    p.beta3 = _beta(blk.ff_norm)
    lin1, lin2 = blk.ff.ff[0].proj, blk.ff.ff[2]
    p.w1 = bf16_of(lin1.weight)
    p.b1 = f32_of(lin1.bias) if lin1.bias is not None else None
    p.w2 = bf16_of(lin2.weight)
    p.b2 = f32_of(lin2.bias) if lin2.bias is not None else None
    p.ada = blk.global_cond_dim is not None and blk.global_cond_dim > 0
    if p.ada:
        p.wmod = bf16_of(blk.to_scale_shift_gate[1].weight)
    p.H = blk.dim // blk.dim_heads
    p.layer_ix = getattr(blk, "layer_ix", 0)
    p.qkn_s = qk_norm_params(blk.self_attn)
    p.qkn_c = qk_norm_params(blk.cross_attn) if p.cross else None
    p.causal = bool(getattr(blk, "causal", False))
    p.conf = conformer_params(blk.conformer) if getattr(blk, "conformer", None) is not None else None
    return p


BLOCK_PARAM_ORDER = ("pre_norm.gamma", "self_attn.to_qkv.weight", "self_attn.to_out.weight",
                     "cross_attend_norm.gamma", "cross_attn.to_q.weight", "cross_attn.to_kv.weight",
                     "cross_attn.to_out.weight", "ff_norm.gamma", "ff.ff.0.proj.weight", "ff.ff.0.proj.bias",
                     "ff.ff.2.weight", "ff.ff.2.bias", "to_scale_shift_gate.1.weight",
                     "self_attn.q_norm.weight", "self_attn.q_norm.bias", "self_attn.k_norm.weight", "self_attn.k_norm.bias",
                     "cross_attn.q_norm.weight", "cross_attn.q_norm.bias", "cross_attn.k_norm.weight",
                     "cross_attn.k_norm.bias",
                     "conformer.in_norm.gamma", "conformer.pointwise_conv.weight", "conformer.glu.proj.weight",
                     "conformer.glu.proj.bias", "conformer.depthwise_conv.weight", "conformer.mid_norm.gamma",
                     "conformer.pointwise_conv_2.weight")


def block_fwd(p, x, ctx, global_cond, mask8, cmask8, rope, B, N, S, ckv=None):
    """transformer.py:649-695.  x: fp32 [B*N, D] residual stream; ctx: bf16 [B*S, Dc] or None;
    global_cond: fp32 [B, G] or None (adaLN).  Returns (y fp32 [B*N, D], saved)."""
    D = x.shape[-1]
    sv = SimpleNamespace(x=x)
    ada = p.ada and global_cond is not None
    sc_s = sh_s = g_s = sc_f = sh_f = g_f = None
    if ada:
        sv.x_global = global_cond
        sv.sg = ops.silu_fwd(global_cond)                                   # 641-644 SiLU -> Linear(no bias)
        sv.sgb = ops.cast(sv.sg, BF16)
        sv.mod = ops.gemm(sv.sgb, p.wmod, out_dtype=F32)                    # [B, 6D]
        sc_s, sh_s, g_s, sc_f, sh_f, g_f = (sv.mod[:, i * D:(i + 1) * D] for i in range(6))
    # self-attention
    sv.h1, sv.mean1, sv.rstd1 = ops.layernorm_fwd(x, p.g1, p.beta1, sc_s, sh_s, rows_per_batch=N)
    sv.x1, sv.sa = self_attn_fwd(sv.h1, p.wqkv, p.wo, B, N, p.H, rope, mask8, residual=x, gate=g_s, qkn=p.qkn_s,
                                 causal=p.causal)
    xcur = sv.x1
    # cross-attention (never modulated, 670-671)
    sv.has_cross = p.cross and ctx is not None
    if sv.has_cross:
        sv.h2, sv.mean2, sv.rstd2 = ops.layernorm_fwd(xcur, p.g2, p.beta2)
        sv.x2, sv.ca = cross_attn_fwd(sv.h2, ctx, p.wq, p.wkv, p.wo2, B, N, S, p.H, cmask8, residual=xcur, qkn=p.qkn_c,
                                      ckv=ckv, layer_ix=p.layer_ix, causal=p.causal)
        xcur = sv.x2
    # conformer module (673-674 / 691-692: x = x + conformer(x), never modulated)
    sv.cf = None
    if p.conf is not None:
        sv.xc, sv.cf = conformer_fwd(p.conf, xcur, B, N)
        xcur = sv.xc
    # feed-forward
    sv.h3, sv.mean3, sv.rstd3 = ops.layernorm_fwd(xcur, p.g3, p.beta3, sc_f, sh_f, rows_per_batch=N)
    y, sv.ff = ff_fwd(sv.h3, p.w1, p.b1, p.w2, p.b2, N, residual=xcur, gate=g_f)
    sv.y = y if ada else None
    sv.ada = ada
    return y, sv


def block_bwd(p, sv, g, ctx, mask8, cmask8, rope, B, N, S, go=None, dctx_acc=None, want_dctx=True, g_bf16=None,
              want_dx_bf16=False, bias2_done=False, dx_colsum_out=None):
    """g: fp32 [B*N, D] gradient of the block output (g_bf16: the same values already rounded to bf16, when the
    producer - the next block's LayerNorm backward - emitted them).  Returns (dx fp32, dctx fp32|None,
    dglobal fp32|None, go, dx bf16|None)."""
    D = g.shape[-1]
    go = (go or GradOut()).defer()
    ada = sv.ada
    mod = sv.mod if ada else None
    sl = (lambda i: mod[:, i * D:(i + 1) * D]) if ada else (lambda i: None)
    dmod = torch.empty_like(mod) if ada else None
    dsl = (lambda i: dmod[:, i * D:(i + 1) * D]) if ada else (lambda i: None)
    xin_ff = sv.xc if sv.cf is not None else (sv.x2 if sv.has_cross else sv.x1)

    # ---- feed-forward branch:  y = xin_ff + FF(LN(xin_ff)*(1+sc)+sh) * sigmoid(1-gate)
    if ada:
        gb, dg = ops.grad_cast(g, B, N, gate=sl(5), x_out=sv.y, x_in=xin_ff)
        dsl(5).copy_(dg)
    else:
        gb = g_bf16 if g_bf16 is not None else ops.cast(g, BF16)
    dh3 = ff_bwd(go, gb, sv.h3, sv.ff, p.w1, p.w2, want_bias=p.b1 is not None,
                 bias2_done=bias2_done and g_bf16 is not None and not ada)
    if ada:
        dsc, dsh = ops.adaln_mod_bwd(dh3, xin_ff, p.g3, p.beta3, sv.mean3, sv.rstd3, B, N)
        dsl(3).copy_(dsc)
        dsl(4).copy_(dsh)
    g2, g2b = go.ln("ff_norm.gamma", dh3, xin_ff, p.g3, sv.mean3, sv.rstd3, scale=sl(3), rows_per_batch=N, dres=g,
                    want_bf16=True)
    # ---- conformer module
    if sv.cf is not None:
        g2, g2b = conformer_bwd(go, p.conf, sv.cf, g2, B, N, gb=g2b, want_bf16=True)
    # ---- cross-attention branch
    dctx = None
    if sv.has_cross:
        dh2, dctx = cross_attn_bwd(go, g2b, sv.h2, ctx, sv.ca, p.wq, p.wkv, p.wo2, B, N, S, p.H, cmask8, dctx_acc,
                                   want_dctx, qkn=p.qkn_c, causal=p.causal)
        g1, g1bf = go.ln("cross_attend_norm.gamma", dh2, sv.x1, p.g2, sv.mean2, sv.rstd2, dres=g2,
                         want_bf16=not ada and mask8 is None)
    else:
        g1, g1bf = g2, g2b
    # ---- self-attention branch:  x1 = x + SA(LN(x)*(1+sc)+sh) * sigmoid(1-gate)   (masked rows contribute 0)
    if ada:
        g1b, dg = ops.grad_cast(g1, B, N, gate=sl(2), x_out=sv.x1, x_in=sv.x, row_mask=mask8)
        dsl(2).copy_(dg)
    elif mask8 is not None:
        g1b, _ = ops.grad_cast(g1, B, N, row_mask=mask8)
    else:
        g1b = g1bf if g1bf is not None else ops.cast(g1, BF16)
    dh1 = self_attn_bwd(go, g1b, sv.h1, sv.sa, p.wqkv, p.wo, B, N, p.H, rope, mask8, qkn=p.qkn_s, causal=p.causal)
    if ada:
        dsc, dsh = ops.adaln_mod_bwd(dh1, sv.x, p.g1, p.beta1, sv.mean1, sv.rstd1, B, N)
        dsl(0).copy_(dsc)
        dsl(1).copy_(dsh)
    # (dx_colsum_out: the FF-out bias sink of the block BELOW - its output gradient is this dx - fused into this pass)
    dx, dxb = go.ln("pre_norm.gamma", dh1, sv.x, p.g1, sv.mean1, sv.rstd1, scale=sl(0), rows_per_batch=N, dres=g1,
                    want_bf16=want_dx_bf16, dx_colsum_out=dx_colsum_out if want_dx_bf16 else None)
    go.dx_colsum_fused = go.colsum_fused
    dglobal = None
    if ada:
        dmb = ops.cast(dmod, BF16)
        go.wgrad("to_scale_shift_gate.1.weight", dmb, sv.sgb)
        dsg = dgrad(dmb, p.wmod, out_dtype=F32)
        dglobal = ops.silu_bwd(dsg, sv.x_global)
    go.flush()
    go.zero_unwritten()
    return dx, dctx, dglobal, go, dxb
