"""Forward / backward of one Llama decoder layer as a sequence of C-ABI kernel launches (manual autograd, same
conventions as dit_ops.py: residual stream fp32, GEMM operands bf16, fp32 accumulation, fp32 parameter gradients).

The layer is the third-party `transformers` LlamaDecoderLayer that the reference's task model runs under
`self.base_model.model(inputs_embeds=..., attention_mask=...)` (model_sigmaVAE.py:78-81): pre-norm, RMSNorm, rotary
over the whole 64-wide head, grouped-query causal attention with a key-padding mask, SwiGLU MLP without biases.
q/k/v and up/gate are single fused GEMMs over fused parameters (split back into the HF names only in state_dict()).
"""
import math
from types import SimpleNamespace

import torch

from . import ops
from .dit_ops import BF16, F32, GradOut, bf16_of, dgrad, f32_of


def inv_freq(head_dim, theta, scaling=None):
    """rotary frequencies of LlamaRotaryEmbedding incl. the 'llama3' scaling rule (transformers
    modeling_rope_utils._compute_llama3_parameters): long wavelengths / factor, smooth blend in the medium band"""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float64) / head_dim))
    kind = (scaling or {}).get("rope_type", (scaling or {}).get("type"))
    if kind in (None, "default"):
        return inv.float()
    if kind != "llama3":
        raise NotImplementedError(f"rope scaling {kind!r}")
    factor, lo, hi = scaling["factor"], scaling["low_freq_factor"], scaling["high_freq_factor"]
    old = scaling["original_max_position_embeddings"]
    wavelen = 2 * math.pi / inv
    scaled = torch.where(wavelen > old / lo, inv / factor, inv)
    smooth = (old / wavelen - lo) / (hi - lo)
    smoothed = (1 - smooth) * scaled / factor + smooth * scaled
    medium = ~(wavelen < old / hi) & ~(wavelen > old / lo)
    return torch.where(medium, smoothed, scaled).float()


def rope_tables(L, inv, device):
    """cos / sin [L, head_dim/2] fp32 for kalle_attention_* with rot = head_dim (positions arange(L))"""
    fr = torch.arange(L, dtype=torch.float32)[:, None] * inv[None, :].float().cpu()
    return fr.cos().contiguous().to(device), fr.sin().contiguous().to(device)


def layer_params(layer):
    p = SimpleNamespace()
    p.g1 = f32_of(layer.input_layernorm.weight)
    p.g2 = f32_of(layer.post_attention_layernorm.weight)
    p.wqkv = bf16_of(layer.self_attn.qkv_proj.weight)
    p.wo = bf16_of(layer.self_attn.o_proj.weight)
    p.wug = bf16_of(layer.mlp.up_gate_proj.weight)
    p.wdown = bf16_of(layer.mlp.down_proj.weight)
    p.H, p.Hkv, p.eps = layer.self_attn.num_heads, layer.self_attn.num_kv_heads, layer.input_layernorm.variance_epsilon
    return p


PARAM_ORDER = ("input_layernorm.weight", "self_attn.qkv_proj.weight", "self_attn.o_proj.weight",
               "post_attention_layernorm.weight", "mlp.up_gate_proj.weight", "mlp.down_proj.weight")


def layer_fwd(p, x, B, L, rope, mask8):
    """x: fp32 [B*L, D] residual stream -> (fp32 [B*L, D], saved)"""
    H, Hkv = p.H, p.Hkv
    D = H * 64
    ld = (H + 2 * Hkv) * 64
    h1, rr1 = ops.rmsnorm_fwd(x, p.g1, eps=p.eps, out_dtype=BF16)
    qkv = ops.gemm(h1, p.wqkv)
    ao, lse = ops.attention_fwd(qkv, qkv, qkv, ldq=ld, q_off=0, ldk=ld, k_off=D, ldv=ld, v_off=D + Hkv * 64, B=B, H=H,
                                Hkv=Hkv, Nq=L, Nk=L, rope=rope, key_mask=mask8, causal=True)
    x2 = ops.gemm(ao.view(B * L, D), p.wo, out_dtype=F32, residual=x)
    h2, rr2 = ops.rmsnorm_fwd(x2, p.g2, eps=p.eps, out_dtype=BF16)
    inner = p.wug.shape[0] // 2
    hf = torch.empty((B * L, 2 * inner), device=x.device, dtype=BF16)
    act = torch.empty((B * L, inner), device=x.device, dtype=BF16)
    if ops.gemm(h2, p.wug, out=hf, glu_mode=1, glu_inner=inner, glu_aux=act) is None:   # fused GEMM + SwiGLU
        ops.gemm(h2, p.wug, out=hf)
        act = ops.swiglu_fwd(hf)
    x3 = ops.gemm(act, p.wdown, out_dtype=F32, residual=x2)
    return x3, (x, h1, rr1, qkv, ao, lse, x2, h2, rr2, hf, act)


def layer_bwd(p, saved, g, B, L, rope, mask8, go=None, g_bf16=None, want_dx_bf16=False):
    """g: fp32 [B*L, D] gradient of the layer output (g_bf16: its bf16 copy if the layer above left one).
    Returns (dx fp32, dx bf16 | None, go)."""
    go = (go or GradOut()).defer()          # the four weight gradients of the layer go out in one grouped launch
    x, h1, rr1, qkv, ao, lse, x2, h2, rr2, hf, act = saved
    H, Hkv = p.H, p.Hkv
    D = H * 64
    ld = (H + 2 * Hkv) * 64
    M = B * L
    gb = g_bf16 if g_bf16 is not None else ops.cast(g, BF16)
    # ---- MLP branch: x3 = x2 + down(up * silu(gate))
    go.wgrad("mlp.down_proj.weight", gb, act)
    inner = hf.shape[-1] // 2
    dhf = torch.empty_like(hf)
    if ops.gemm(gb, p.wdown, b_kmajor=True, out=dhf, N=inner, glu_mode=2, glu_inner=inner, glu_aux=hf) is None:
        dhf = ops.swiglu_bwd(dgrad(gb, p.wdown), hf, None)
    go.wgrad("mlp.up_gate_proj.weight", dhf, h2)
    dh2 = dgrad(dhf, p.wug)
    dxb2 = torch.empty((M, D), device=g.device, dtype=BF16)
    dx2 = go.rms("post_attention_layernorm.weight", dh2, x2, p.g2, rr2, dres=g, dx_bf16=dxb2)
    # ---- attention branch: x2 = x + o(attn)
    go.wgrad("self_attn.o_proj.weight", dxb2, ao.view(M, D))
    dao = dgrad(dxb2, p.wo)
    dqkv = torch.empty_like(qkv)
    ops.attention_bwd(qkv, qkv, qkv, ao, dao, lse, dqkv, dqkv, dqkv, ldq=ld, q_off=0, ldk=ld, k_off=D, ldv=ld,
                      v_off=D + Hkv * 64, B=B, H=H, Hkv=Hkv, Nq=L, Nk=L, rope=rope, key_mask=mask8, causal=True)
    go.wgrad("self_attn.qkv_proj.weight", dqkv, h1)
    dh1 = dgrad(dqkv, p.wqkv)
    dxb = torch.empty((M, D), device=g.device, dtype=BF16) if want_dx_bf16 else None
    dx = go.rms("input_layernorm.weight", dh1, x, p.g1, rr1, dres=dx2, dx_bf16=dxb)
    go.flush()
    go.zero_unwritten()
    return dx, dxb, go


def layer_fwd_cached(p, x, kv_cache, t0, rope):
    """inference with a KV cache (batch 1): x fp32 [n, D] are positions t0 .. t0+n-1; kv_cache bf16 [Lmax, 2*Hkv*64]
    holds the un-rotated k | v rows of positions < t0 and receives the new ones (rotary is applied by the attention
    kernel from the row index, so cached keys need no re-rotation).  Returns fp32 [n, D]."""
    H, Hkv = p.H, p.Hkv
    D = H * 64
    ld = (H + 2 * Hkv) * 64
    n = x.shape[0]
    h1, _ = ops.rmsnorm_fwd(x, p.g1, eps=p.eps, out_dtype=BF16)
    qkv = ops.gemm(h1, p.wqkv)
    ops.copy_rows(qkv[:, D:], kv_cache[t0:], 1, n, 2 * Hkv * 64, 0, ld, 0, 2 * Hkv * 64)
    ao, _ = ops.attention_fwd(qkv, kv_cache, kv_cache, ldq=ld, q_off=0, ldk=2 * Hkv * 64, k_off=0, ldv=2 * Hkv * 64,
                              v_off=Hkv * 64, B=1, H=H, Hkv=Hkv, Nq=n, Nk=t0 + n, rope=rope, causal=True)
    x2 = ops.gemm(ao.view(n, D), p.wo, out_dtype=F32, residual=x)
    h2, _ = ops.rmsnorm_fwd(x2, p.g2, eps=p.eps, out_dtype=BF16)
    hf = ops.gemm(h2, p.wug)
    act = ops.swiglu_fwd(hf)
    return ops.gemm(act, p.wdown, out_dtype=F32, residual=x2)
