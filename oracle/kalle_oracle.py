"""CPU fp32 ORACLE for the kalle-audio DiT / audio-VAE hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch restatement (plain torch fp32 on the CPU, functional style over state dicts that use
the reference's parameter names) of the algorithm in /root/reference/stable_audio_tools.  Each function cites the
reference file:line it follows.  It is pinned against golden vectors produced by running the reference itself in
the build container (tests/golden/make_golden.py, make_golden_r02.py -> tests/golden/*.npz; checked in
tests/test_oracle_golden.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The product path
(kalle_audio_amd/*) never does: it runs hand-written HIP kernels and fails loudly without them.

Parity status: PINNED for every function below by the committed fixtures, except `snake`-free statements marked
otherwise.  Third-party arithmetic that enters: WNConv1d/WNConvTranspose1d == torch.nn.utils.weight_norm over
Conv1d/ConvTranspose1d (descript-audio-codec `dac.nn.layers`, version unpinned in the reference).
UNPINNED: `activation1d` (alias-free-torch is absent) and the label transform injected into `llasa_model_forward`
(twj_utils is a dangling symlink in the reference); both are restated from published / in-tree definitions and say so.
"""
import math

import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------- small pieces
def layer_norm(x, gamma, beta=None, eps=1e-5):
    """transformer.py:173-192: F.layer_norm over the last dim with gamma and an (optional) beta buffer."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    y = (x - mu) * torch.rsqrt(var + eps) * gamma
    return y + beta if beta is not None else y


def rms_norm(x, scale, eps=1e-6):
    """blocks.py:268-272 (RMSNorm 285-299): x * scale * rsqrt(mean(x^2)+eps), statistics in fp32."""
    ms = (x.float() ** 2).mean(-1, keepdim=True)
    return x * (scale.float() * torch.rsqrt(ms + eps)).to(x.dtype)


def snake_beta(x, alpha, beta, logscale=True):
    """blocks.py:301-339: x + sin^2(x*a)/(b+1e-9), a,b = exp(param) per channel; x is (B, C, L)."""
    a = alpha[None, :, None]
    b = beta[None, :, None]
    if logscale:
        a, b = a.exp(), b.exp()
    return x + (1.0 / (b + 1e-9)) * torch.sin(x * a) ** 2


def fourier_features(t, weight):
    """blocks.py:84-93: f = 2 pi t W^T ; cat(cos f, sin f).  t: [B,1], weight: [F/2, 1]."""
    f = 2 * math.pi * t @ weight.t()
    return torch.cat([f.cos(), f.sin()], -1)


def rotary_freqs(n, rot_dim=32, base=10000.0):
    """transformer.py:89-138: inv_freq over rot_dim, freqs = cat(pos x inv_freq, twice) -> [n, rot_dim]."""
    inv = 1.0 / (base ** (torch.arange(0, rot_dim, 2).float() / rot_dim))
    f = torch.arange(n).float()[:, None] * inv[None, :]
    return torch.cat([f, f], -1)


def apply_rotary(t, freqs):
    """transformer.py:146-170: partial rotary (GPT-J style) on the first freqs.shape[-1] dims, fp32."""
    rot = freqs.shape[-1]
    n = t.shape[-2]
    fr = freqs[-n:]
    tr, tu = t[..., :rot], t[..., rot:]
    half = rot // 2
    rh = torch.cat([-tr[..., half:], tr[..., :half]], -1)
    return torch.cat([tr * fr.cos() + rh * fr.sin(), tu], -1)


def linear(x, w, b=None):
    y = x @ w.t()
    return y + b if b is not None else y


def _sub(sd, prefix):
    pl = len(prefix)
    return {k[pl:]: v for k, v in sd.items() if k.startswith(prefix)}


# ---------------------------------------------------------------------------------------------- attention / FF
def attention(sd, x, context=None, mask=None, context_mask=None, rotary=None, dim_heads=64, qk_l2=False, causal=False):
    """transformer.py:396-547 (einsum / fp32-softmax branch 502-530, which is what the reference runs on CPU).
    sd keys: to_qkv.weight | to_q.weight,to_kv.weight ; to_out.weight
    causal: the mask of create_causal_mask (transformer.py:32-33: ones(i, j).triu(j - i + 1), i.e. query r sees keys
    c <= r + j - i), applied as 519-523 intend.  PARITY UNPINNED for this flag: the reference calls it as
    `self.create_causal_mask` (362, 372, 521), a name Attention does not have, so every causal call on its CPU branch raises
    AttributeError and no fixture can be produced."""
    B, N, D = x.shape
    h = D // dim_heads
    if "to_q.weight" in sd:
        kv_in = context if context is not None else x
        q = linear(x, sd["to_q.weight"])
        k, v = linear(kv_in, sd["to_kv.weight"]).chunk(2, -1)
        kv_h = k.shape[-1] // dim_heads
    else:
        q, k, v = linear(x, sd["to_qkv.weight"]).chunk(3, -1)
        kv_h = h
    q = q.view(B, N, h, dim_heads).transpose(1, 2)
    k = k.reshape(B, -1, kv_h, dim_heads).transpose(1, 2)
    v = v.reshape(B, -1, kv_h, dim_heads).transpose(1, 2)
    if "q_norm.weight" in sd:                         # 426-428 qk_norm == "ln": LayerNorm(dim_heads, eps 1e-6) per head
        q = F.layer_norm(q, (dim_heads,), sd["q_norm.weight"], sd["q_norm.bias"], 1e-6)
        k = F.layer_norm(k, (dim_heads,), sd["k_norm.weight"], sd["k_norm.bias"], 1e-6)
    elif qk_l2:                                       # 423-425 qk_norm == "l2": x / max(||x||, 1e-12)
        q = q / q.norm(dim=-1, keepdim=True).clamp_min(1e-12)
        k = k / k.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    if rotary is not None and context is None:       # 430-444: self-attention only
        q = apply_rotary(q.float(), rotary)
        k = apply_rotary(k.float(), rotary)
    in_mask = context_mask                            # 446-462
    if in_mask is None and context is None:
        in_mask = mask
    if h != kv_h:                                     # 505-508 repeat_interleave
        k = k.repeat_interleave(h // kv_h, 1)
        v = v.repeat_interleave(h // kv_h, 1)
    dots = (q @ k.transpose(-1, -2)) * (dim_heads ** -0.5)
    if in_mask is not None:
        dots = dots.masked_fill(~in_mask[:, None, None, :], -torch.finfo(dots.dtype).max)
    if causal and dots.shape[-2] > 1:                 # 468-469: a single query is never masked
        i, j = dots.shape[-2:]
        dots = dots.masked_fill(torch.ones(i, j, dtype=torch.bool).triu(j - i + 1), -torch.finfo(dots.dtype).max)
    attn = dots.softmax(-1)
    out = (attn @ v).transpose(1, 2).reshape(B, N, D)
    out = linear(out, sd["to_out.weight"])
    if mask is not None:                              # 543-545 zero padded query rows
        out = out.masked_fill(~mask[:, :, None], 0.0)
    return out


def feed_forward(sd, x):
    """transformer.py:196-269: SwiGLU: proj (bias) -> x*silu(gate) -> linear_out (bias). keys ff.0.proj.*, ff.2.*"""
    hdn = linear(x, sd["ff.0.proj.weight"], sd["ff.0.proj.bias"])
    a, g = hdn.chunk(2, -1)
    return linear(a * F.silu(g), sd["ff.2.weight"], sd["ff.2.bias"])


_attention_impl = attention


def conformer_module(sd, x):
    """transformer.py:550-583 ConformerModule: LayerNorm -> 1x1 conv -> GLU (Linear D -> 2D with bias, x * silu(gate)) ->
    depthwise Conv1d(k = 17, padding 8, no bias) along the sequence -> LayerNorm -> SiLU -> 1x1 conv.  x: [B, N, D];
    keys in_norm.gamma, pointwise_conv.weight [D, D, 1], glu.proj.{weight,bias}, depthwise_conv.weight [D, 1, K],
    mid_norm.gamma, pointwise_conv_2.weight [D, D, 1]"""
    D = x.shape[-1]
    h = layer_norm(x, sd["in_norm.gamma"], sd.get("in_norm.beta"))
    h = linear(h, sd["pointwise_conv.weight"][:, :, 0])
    a, g = linear(h, sd["glu.proj.weight"], sd["glu.proj.bias"]).chunk(2, -1)
    h = a * F.silu(g)
    w = sd["depthwise_conv.weight"]
    h = F.conv1d(h.transpose(1, 2), w, None, padding=(w.shape[-1] - 1) // 2, groups=D).transpose(1, 2)
    h = F.silu(layer_norm(h, sd["mid_norm.gamma"], sd.get("mid_norm.beta")))
    return linear(h, sd["pointwise_conv_2.weight"][:, :, 0])


def transformer_block(sd, x, context=None, global_cond=None, mask=None, context_mask=None, rotary=None, dim_heads=64,
                      qk_l2=False, causal=False):
    """transformer.py:649-695.  adaLN branch when the block has to_scale_shift_gate and global_cond is given; the
    ConformerModule (673-674 / 691-692) when the block has one; causal: see attention()."""
    ln = lambda pre, t: layer_norm(t, sd[pre + ".gamma"], sd.get(pre + ".beta"))
    sa = _sub(sd, "self_attn.")
    ff = _sub(sd, "ff.")
    conf = _sub(sd, "conformer.")
    attention = lambda *a, **k: _attention_impl(*a, causal=causal, **k)  # (Attention(causal=causal), 613-633)
    if "to_scale_shift_gate.1.weight" in sd and global_cond is not None:
        mod = linear(F.silu(global_cond), sd["to_scale_shift_gate.1.weight"]).unsqueeze(1)
        sc_s, sh_s, g_s, sc_f, sh_f, g_f = mod.chunk(6, -1)
        res = x
        hx = ln("pre_norm", x) * (1 + sc_s) + sh_s
        hx = attention(sa, hx, mask=mask, rotary=rotary, dim_heads=dim_heads, qk_l2=qk_l2)
        x = hx * torch.sigmoid(1 - g_s) + res
        if context is not None:
            x = x + attention(_sub(sd, "cross_attn."), ln("cross_attend_norm", x), context=context,
                              context_mask=context_mask, dim_heads=dim_heads, qk_l2=qk_l2)
        if conf:
            x = x + conformer_module(conf, x)
        res = x
        hx = ln("ff_norm", x) * (1 + sc_f) + sh_f
        x = feed_forward(ff, hx) * torch.sigmoid(1 - g_f) + res
    else:
        x = x + attention(sa, ln("pre_norm", x), mask=mask, rotary=rotary, dim_heads=dim_heads, qk_l2=qk_l2)
        if context is not None:
            x = x + attention(_sub(sd, "cross_attn."), ln("cross_attend_norm", x), context=context,
                              context_mask=context_mask, dim_heads=dim_heads, qk_l2=qk_l2)
        if conf:
            x = x + conformer_module(conf, x)
        x = x + feed_forward(ff, ln("ff_norm", x))
    return x


def scaled_sinusoidal_embedding(scale, n, dim, theta=10000.0):
    """transformer.py:67-87 ScaledSinusoidalEmbedding: cat(sin, cos)(pos * theta^(-arange(dim/2) / (dim/2))) * scale"""
    half = dim // 2
    inv_freq = theta ** -(torch.arange(half).float() / half)
    emb = torch.arange(n).float()[:, None] * inv_freq[None, :]
    return torch.cat((emb.sin(), emb.cos()), -1) * scale


def continuous_transformer(sd, x, depth, mask=None, prepend_embeds=None, prepend_mask=None, global_cond=None,
                           context=None, context_mask=None, dim_heads=64, rotary=True, causal=False):
    """transformer.py:758-812: project_in, prepend (+mask cat 776-787), rotary over the full length, the optional position
    embedding added to the sequence (796-797: ScaledSinusoidalEmbedding when the state has pos_emb.scale,
    AbsolutePositionalEmbedding 45-65 = emb.weight[arange(n)] * dim^-0.5 when it has pos_emb.emb.weight), blocks,
    project_out."""
    B, T = x.shape[:2]
    if "project_in.weight" in sd:
        x = linear(x, sd["project_in.weight"])
    if prepend_embeds is not None:
        P = prepend_embeds.shape[1]
        x = torch.cat([prepend_embeds, x], 1)
        if prepend_mask is not None or mask is not None:
            mask = mask if mask is not None else torch.ones(B, T, dtype=torch.bool)
            prepend_mask = prepend_mask if prepend_mask is not None else torch.ones(B, P, dtype=torch.bool)
            mask = torch.cat([prepend_mask, mask], -1)
    rot = rotary_freqs(x.shape[1], max(dim_heads // 2, 32)) if rotary else None
    if "pos_emb.scale" in sd:
        x = x + scaled_sinusoidal_embedding(sd["pos_emb.scale"], x.shape[1], x.shape[2])
    elif "pos_emb.emb.weight" in sd:
        x = x + sd["pos_emb.emb.weight"][:x.shape[1]] * x.shape[2] ** -0.5
    # transformer.py:800-802: the layers are called with rotary_pos_emb, global_cond and **kwargs (context, context_mask) -
    # the assembled `mask` is NOT handed on, so a padding mask never reaches the self-attention of a ContinuousTransformer
    # (pinned by tests/golden/training_step.npz, whose steps run with mask_padding=True)
    for i in range(depth):
        x = transformer_block(_sub(sd, f"layers.{i}."), x, context=context, global_cond=global_cond, mask=None,
                              context_mask=context_mask, rotary=rot, dim_heads=dim_heads, causal=causal)
    if "project_out.weight" in sd:
        x = linear(x, sd["project_out.weight"])
    return x


# ---------------------------------------------------------------------------------------------- DiT
def _mlp2(sd, pre, x, bias):
    x = linear(x, sd[pre + ".0.weight"], sd.get(pre + ".0.bias") if bias else None)
    return linear(F.silu(x), sd[pre + ".2.weight"], sd.get(pre + ".2.bias") if bias else None)


def dit_inner(sd, cfg, x, t, mask=None, cross_attn_cond=None, cross_attn_cond_mask=None, global_embed=None,
              prepend_cond=None, prepend_cond_mask=None):
    """dit.py:135-229 (_forward), continuous_transformer branch. cfg: dict(depth, num_heads, embed_dim,
    global_cond_type)."""
    D = cfg["embed_dim"]
    dh = D // cfg["num_heads"]
    if cross_attn_cond is not None:
        cross_attn_cond = _mlp2(sd, "to_cond_embed", cross_attn_cond, False)
    if global_embed is not None:
        global_embed = _mlp2(sd, "to_global_embed", global_embed, False)
    prepend_inputs, prepend_mask, plen = None, None, 0
    if prepend_cond is not None:
        prepend_inputs = _mlp2(sd, "to_prepend_embed", prepend_cond, False)
        prepend_mask = prepend_cond_mask
    temb = _mlp2(sd, "to_timestep_embed", fourier_features(t[:, None], sd["timestep_features.weight"]), True)
    global_embed = temb if global_embed is None else global_embed + temb
    gtype = cfg.get("global_cond_type", "prepend")
    if gtype == "prepend":
        ones = torch.ones(x.shape[0], 1, dtype=torch.bool)
        if prepend_inputs is None:
            prepend_inputs, prepend_mask = global_embed.unsqueeze(1), ones
        else:
            prepend_inputs = torch.cat([prepend_inputs, global_embed.unsqueeze(1)], 1)
            prepend_mask = torch.cat([prepend_mask, ones], 1)
        plen = prepend_inputs.shape[1]
    x = F.conv1d(x, sd["preprocess_conv.weight"]) + x           # dit.py:197
    x = x.transpose(1, 2)                                        # b c t -> b t c
    out = continuous_transformer(_sub(sd, "transformer."), x, cfg["depth"], mask=mask, prepend_embeds=prepend_inputs,
                                 prepend_mask=prepend_mask, global_cond=global_embed if gtype == "adaLN" else None,
                                 context=cross_attn_cond, context_mask=cross_attn_cond_mask, dim_heads=dh)
    out = out.transpose(1, 2)[:, :, plen:]
    return F.conv1d(out, sd["postprocess_conv.weight"]) + out   # dit.py:224


def dit_forward(sd, cfg, x, t, cross_attn_cond=None, cross_attn_cond_mask=None, negative_cross_attn_cond=None,
                negative_cross_attn_mask=None, global_embed=None, prepend_cond=None, prepend_cond_mask=None,
                cfg_scale=1.0, scale_phi=0.0, mask=None):
    """dit.py:231-379 with cfg_dropout_prob = 0 (the dropout draw is the only stochastic part and is exercised
    separately).  Note dit.py:254-257: the cross-attention mask is dropped."""
    cross_attn_cond_mask = None
    if cfg_scale != 1.0 and (cross_attn_cond is not None or prepend_cond is not None):
        bx = torch.cat([x, x], 0)
        bt = torch.cat([t, t], 0)
        bg = torch.cat([global_embed, global_embed], 0) if global_embed is not None else None
        bc = None
        if cross_attn_cond is not None:
            null = torch.zeros_like(cross_attn_cond)
            if negative_cross_attn_cond is not None:
                neg = negative_cross_attn_cond
                if negative_cross_attn_mask is not None:
                    neg = torch.where(negative_cross_attn_mask.bool().unsqueeze(2), neg, null)
                bc = torch.cat([cross_attn_cond, neg], 0)
            else:
                bc = torch.cat([cross_attn_cond, null], 0)
        bp, bpm = None, None
        if prepend_cond is not None:
            bp = torch.cat([prepend_cond, torch.zeros_like(prepend_cond)], 0)
            if prepend_cond_mask is not None:
                bpm = torch.cat([prepend_cond_mask, prepend_cond_mask], 0)
        bm = torch.cat([mask, mask], 0) if mask is not None else None
        bo = dit_inner(sd, cfg, bx, bt, mask=bm, cross_attn_cond=bc, global_embed=bg, prepend_cond=bp,
                       prepend_cond_mask=bpm)
        cond, uncond = bo.chunk(2, 0)
        out = uncond + (cond - uncond) * cfg_scale
        if scale_phi != 0.0:                                      # CFG rescale, dit.py:354-357
            out = scale_phi * (out * (cond.std(1, keepdim=True) / out.std(1, keepdim=True))) + (1 - scale_phi) * out
        return out
    return dit_inner(sd, cfg, x, t, mask=mask, cross_attn_cond=cross_attn_cond,
                     cross_attn_cond_mask=cross_attn_cond_mask, global_embed=global_embed, prepend_cond=prepend_cond,
                     prepend_cond_mask=prepend_cond_mask)


# ---------------------------------------------------------------------------------------------- train step / samplers
def alphas_sigmas(t, objective="v"):
    """inference/sampling.py:8-11 ("v") ; training/diffusion.py:367-368 (rectified flow)."""
    if objective == "v":
        return torch.cos(t * math.pi / 2), torch.sin(t * math.pi / 2)
    return 1 - t, t


def diffuse(x, noise, t, objective="v"):
    """training/diffusion.py:371-379: x_t = x*alpha + n*sigma ; target = n*alpha - x*sigma | n - x."""
    a, s = alphas_sigmas(t, objective)
    a, s = a[:, None, None], s[:, None, None]
    xt = x * a + noise * s
    target = noise * a - x * s if objective == "v" else noise - x
    return xt, target


def mse_loss(output, target, mask=None, weight=1.0):
    """training/losses/losses.py:53-69: elementwise MSE, optional [B,T] bool mask broadcast over channels, mean."""
    l = (output - target) ** 2
    if mask is not None:
        m = mask.unsqueeze(1) if mask.dim() == 2 else mask
        m = m.expand(-1, l.shape[1], -1) if m.shape[1] != l.shape[1] else m
        l = l[m]
    return weight * l.mean()


def train_step_loss(sd, cfg, latents, noise, t, objective="v", padding_mask=None, **cond):
    """training/diffusion.py:365-399 with explicit t and noise (that is what makes parity seed-free)."""
    xt, target = diffuse(latents, noise, t, objective)
    out = dit_forward(sd, cfg, xt, t, **cond)
    return mse_loss(out, target, padding_mask), out, xt, target


def sample_ddim(model_fn, x, steps, eta=0.0):
    """inference/sampling.py:47-86 (v-diffusion DDIM; eta = 0 path is deterministic)."""
    ts = x.new_ones([x.shape[0]])
    t = torch.linspace(1, 0, steps + 1)[:-1]
    al, si = alphas_sigmas(t)
    pred = x
    for i in range(steps):
        v = model_fn(x, ts * t[i]).float()
        pred = x * al[i] - v * si[i]
        eps = x * si[i] + v * al[i]
        if i < steps - 1:
            dd = eta * (si[i + 1] ** 2 / si[i] ** 2).sqrt() * (1 - al[i] ** 2 / al[i + 1] ** 2).sqrt()
            adj = (si[i + 1] ** 2 - dd ** 2).sqrt()
            x = pred * al[i + 1] + eps * adj
            if eta:
                x = x + torch.randn_like(x) * dd
    return pred


def sample_euler(model_fn, x, steps, sigma_max=1.0):
    """inference/sampling.py:24-45 (rectified-flow Euler)."""
    t = torch.linspace(sigma_max, 0, steps + 1)
    for tc, tp in zip(t[:-1], t[1:]):
        x = x + (tp - tc) * model_fn(x, tc * torch.ones(x.shape[0], dtype=x.dtype))
    return x


# ---------------------------------------------------------------------------------------------- Oobleck VAE
def wn_weight(sd, pre):
    """torch weight_norm (dim=0): w = g * v / ||v|| with the norm over every dim but 0 (dac.nn.layers.WNConv1d)."""
    v, g = sd[pre + ".weight_v"], sd[pre + ".weight_g"]
    return g * v / v.flatten(1).norm(dim=1).view(-1, 1, 1)


def _act(sd, pre, x, use_snake):
    return snake_beta(x, sd[pre + ".alpha"], sd[pre + ".beta"]) if use_snake else F.elu(x)


def residual_unit(sd, x, dilation, use_snake):
    """autoencoders.py:39-62: x + conv1(act(conv7_dilated(act(x))))  keys layers.{0,1,2,3}"""
    h = _act(sd, "layers.0", x, use_snake)
    h = F.conv1d(h, wn_weight(sd, "layers.1"), sd["layers.1.bias"], dilation=dilation, padding=3 * dilation)
    h = _act(sd, "layers.2", h, use_snake)
    h = F.conv1d(h, wn_weight(sd, "layers.3"), sd["layers.3.bias"])
    return x + h


def encoder_block(sd, x, stride, use_snake):
    """autoencoders.py:64-81: RU(1,3,9) -> act -> strided conv k=2s pad=ceil(s/2). keys layers.{0..4}"""
    for i, d in enumerate((1, 3, 9)):
        x = residual_unit(_sub(sd, f"layers.{i}."), x, d, use_snake)
    x = _act(sd, "layers.3", x, use_snake)
    return F.conv1d(x, wn_weight(sd, "layers.4"), sd["layers.4.bias"], stride=stride, padding=math.ceil(stride / 2))


def decoder_block(sd, x, stride, use_snake, nearest=False):
    """autoencoders.py:83-114: act -> transposed conv k=2s+s%2 -> RU(1,3,9). keys layers.{0..4}
    nearest (use_nearest_upsample, :87-96): act -> repeat each sample `stride` times -> conv k=2s, no bias, padding 'same'
    (2s-1 zeros in all: s-1 on the left, s on the right), keys layers.1.1.*"""
    x = _act(sd, "layers.0", x, use_snake)
    if nearest:
        x = torch.repeat_interleave(x, stride, dim=2)
        x = F.conv1d(F.pad(x, (stride - 1, stride)), wn_weight(sd, "layers.1.1"), None)
    else:
        x = F.conv_transpose1d(x, wn_weight(sd, "layers.1"), sd["layers.1.bias"], stride=stride,
                               padding=math.ceil(stride / 2))
    for i, d in enumerate((1, 3, 9)):
        x = residual_unit(_sub(sd, f"layers.{i + 2}."), x, d, use_snake)
    return x


def oobleck_encoder(sd, x, strides, use_snake):
    """autoencoders.py:116-147"""
    x = F.conv1d(x, wn_weight(sd, "layers.0"), sd["layers.0.bias"], padding=3)
    for i, s in enumerate(strides):
        x = encoder_block(_sub(sd, f"layers.{i + 1}."), x, s, use_snake)
    n = len(strides)
    x = _act(sd, f"layers.{n + 1}", x, use_snake)
    return F.conv1d(x, wn_weight(sd, f"layers.{n + 2}"), sd[f"layers.{n + 2}.bias"], padding=1)


def oobleck_decoder(sd, z, strides, use_snake, final_tanh=True, nearest=False):
    """autoencoders.py:150-191 (blocks run over reversed strides; last conv has no bias)"""
    x = F.conv1d(z, wn_weight(sd, "layers.0"), sd["layers.0.bias"], padding=3)
    n = len(strides)
    for j, s in enumerate(reversed(strides)):
        x = decoder_block(_sub(sd, f"layers.{j + 1}."), x, s, use_snake, nearest=nearest)
    x = _act(sd, f"layers.{n + 1}", x, use_snake)
    x = F.conv1d(x, wn_weight(sd, f"layers.{n + 2}"), None, padding=3)
    return torch.tanh(x) if final_tanh else x


def pretransform_encode(sd, wav, strides, use_snake, scale=1.0):
    """pretransforms.py:50-61 + autoencoders.py:275-318 + bottleneck.py:85-107 (VAE bottleneck is a pass-through in
    this reference: the encoder output mean||scale is returned as is), then / scale."""
    return oobleck_encoder(_sub(sd, "encoder."), wav, strides, use_snake) / scale


def pretransform_decode(sd, z, strides, use_snake, scale=1.0, final_tanh=True):
    """pretransforms.py:63-75 + autoencoders.py:320-361: z * scale -> decoder."""
    return oobleck_decoder(_sub(sd, "decoder."), z * scale, strides, use_snake, final_tanh)


# ---------------------------------------------------------------------------------------------- parameter shapes
def conformer_shapes(D, prefix="", K=17):
    return [(prefix + "in_norm.gamma", (D,)), (prefix + "pointwise_conv.weight", (D, D, 1)),
            (prefix + "glu.proj.weight", (2 * D, D)), (prefix + "glu.proj.bias", (2 * D,)),
            (prefix + "depthwise_conv.weight", (D, 1, K)), (prefix + "mid_norm.gamma", (D,)),
            (prefix + "pointwise_conv_2.weight", (D, D, 1))]


def block_shapes(D, dim_heads=64, dim_context=None, global_cond_dim=None, prefix="", qk_ln=False, conformer=False):
    """(name, shape) list of one TransformerBlock (transformer.py:585-647); qk_ln: Attention(qk_norm="ln")'s LayerNorms"""
    qkn = lambda a: [(f"{prefix}{a}.{n}.{w}", (dim_heads,)) for n in ("q_norm", "k_norm") for w in ("weight", "bias")] if qk_ln else []
    s = [(prefix + "pre_norm.gamma", (D,)), (prefix + "self_attn.to_qkv.weight", (3 * D, D)),
         (prefix + "self_attn.to_out.weight", (D, D))] + qkn("self_attn")
    if dim_context is not None:
        s += [(prefix + "cross_attend_norm.gamma", (D,)), (prefix + "cross_attn.to_q.weight", (D, D)),
              (prefix + "cross_attn.to_kv.weight", (2 * dim_context, dim_context)),
              (prefix + "cross_attn.to_out.weight", (D, D))] + qkn("cross_attn")
    s += [(prefix + "ff_norm.gamma", (D,)), (prefix + "ff.ff.0.proj.weight", (8 * D, D)),
          (prefix + "ff.ff.0.proj.bias", (8 * D,)), (prefix + "ff.ff.2.weight", (D, 4 * D)),
          (prefix + "ff.ff.2.bias", (D,))]
    if conformer:
        s += conformer_shapes(D, prefix + "conformer.")
    if global_cond_dim:
        s += [(prefix + "to_scale_shift_gate.1.weight", (6 * D, global_cond_dim))]
    return s


def continuous_transformer_shapes(D, depth, dim_in=None, dim_out=None, dim_context=None, global_cond_dim=None,
                                  prefix=""):
    s = []
    if dim_in is not None:
        s.append((prefix + "project_in.weight", (D, dim_in)))
    if dim_out is not None:
        s.append((prefix + "project_out.weight", (dim_out, D)))
    for i in range(depth):
        s += block_shapes(D, dim_context=dim_context, global_cond_dim=global_cond_dim, prefix=f"{prefix}layers.{i}.")
    return s


def dit_shapes(io_channels, embed_dim, depth, cond_token_dim=0, global_cond_dim=0, prepend_cond_dim=0,
               global_cond_type="prepend", project_cond_tokens=True):
    """dit.py:14-133 parameter inventory (continuous_transformer branch)."""
    D = embed_dim
    s = [("timestep_features.weight", (128, 1)), ("to_timestep_embed.0.weight", (D, 256)),
         ("to_timestep_embed.0.bias", (D,)), ("to_timestep_embed.2.weight", (D, D)), ("to_timestep_embed.2.bias", (D,))]
    ce = 0
    if cond_token_dim > 0:
        ce = D if project_cond_tokens else cond_token_dim
        s += [("to_cond_embed.0.weight", (ce, cond_token_dim)), ("to_cond_embed.2.weight", (ce, ce))]
    if global_cond_dim > 0:
        s += [("to_global_embed.0.weight", (D, global_cond_dim)), ("to_global_embed.2.weight", (D, D))]
    if prepend_cond_dim > 0:
        s += [("to_prepend_embed.0.weight", (D, prepend_cond_dim)), ("to_prepend_embed.2.weight", (D, D))]
    s += continuous_transformer_shapes(D, depth, dim_in=io_channels, dim_out=io_channels,
                                       dim_context=ce if cond_token_dim > 0 else None,
                                       global_cond_dim=D if global_cond_type == "adaLN" else None,
                                       prefix="transformer.")
    s += [("preprocess_conv.weight", (io_channels, io_channels, 1)),
          ("postprocess_conv.weight", (io_channels, io_channels, 1))]
    return s


def _wnconv_shapes(pre, cout, cin, k, bias=True, transposed=False):
    wshape = (cin, cout, k) if transposed else (cout, cin, k)
    s = [(pre + ".weight_g", (wshape[0], 1, 1)), (pre + ".weight_v", wshape)]
    if bias:
        s.insert(0, (pre + ".bias", (cout,)))
    return s


def _act_shapes(pre, c, use_snake):
    return [(pre + ".alpha", (c,)), (pre + ".beta", (c,))] if use_snake else []


def residual_unit_shapes(c, use_snake, prefix=""):
    return (_act_shapes(prefix + "layers.0", c, use_snake) + _wnconv_shapes(prefix + "layers.1", c, c, 7) +
            _act_shapes(prefix + "layers.2", c, use_snake) + _wnconv_shapes(prefix + "layers.3", c, c, 1))


def encoder_block_shapes(cin, cout, stride, use_snake, prefix=""):
    s = []
    for i in range(3):
        s += residual_unit_shapes(cin, use_snake, f"{prefix}layers.{i}.")
    return s + _act_shapes(prefix + "layers.3", cin, use_snake) + _wnconv_shapes(prefix + "layers.4", cout, cin, 2 * stride)


def decoder_block_shapes(cin, cout, stride, use_snake, prefix="", nearest=False):
    s = _act_shapes(prefix + "layers.0", cin, use_snake)
    if nearest:
        s += _wnconv_shapes(prefix + "layers.1.1", cout, cin, 2 * stride, bias=False)
    else:
        s += _wnconv_shapes(prefix + "layers.1", cout, cin, 2 * stride + stride % 2, transposed=True)
    for i in range(3):
        s += residual_unit_shapes(cout, use_snake, f"{prefix}layers.{i + 2}.")
    return s


def oobleck_encoder_shapes(in_channels, channels, latent_dim, c_mults, strides, use_snake, prefix=""):
    cm = [1] + list(c_mults)
    s = _wnconv_shapes(prefix + "layers.0", cm[0] * channels, in_channels, 7)
    for i, st in enumerate(strides):
        s += encoder_block_shapes(cm[i] * channels, cm[i + 1] * channels, st, use_snake, f"{prefix}layers.{i + 1}.")
    n = len(strides)
    s += _act_shapes(f"{prefix}layers.{n + 1}", cm[-1] * channels, use_snake)
    return s + _wnconv_shapes(f"{prefix}layers.{n + 2}", latent_dim, cm[-1] * channels, 3)


def oobleck_decoder_shapes(out_channels, channels, latent_dim, c_mults, strides, use_snake, prefix="", nearest=False):
    cm = [1] + list(c_mults)
    s = _wnconv_shapes(prefix + "layers.0", cm[-1] * channels, latent_dim, 7)
    n = len(strides)
    for j, i in enumerate(range(n, 0, -1)):
        s += decoder_block_shapes(cm[i] * channels, cm[i - 1] * channels, strides[i - 1], use_snake,
                                  f"{prefix}layers.{j + 1}.", nearest=nearest)
    s += _act_shapes(f"{prefix}layers.{n + 1}", cm[0] * channels, use_snake)
    return s + _wnconv_shapes(f"{prefix}layers.{n + 2}", out_channels, cm[0] * channels, 7, bias=False)


# ---------------------------------------------------------------------------------------------- mel-VAE (backup/flows.py)
# Parity status: encoder, ResStack, causal Conv1d / ConvTranspose1d, AMP block wiring, flow: PINNED by
# tests/golden/melvae.npz.  `activation1d` restates the third-party alias-free-torch package (imported with `*` at
# backup/flows.py:5, not vendored, version unpinned, absent from this container): PARITY UNPINNED for that one
# function - the decoder fixtures were generated by running the reference's own classes with this restatement
# standing in for the missing import, so they pin the wiring around it but not its FIR taps.
def kaiser_sinc_filter1d(cutoff, half_width, kernel_size):
    """alias-free-torch filter.py: kaiser-windowed sinc low-pass, unit DC gain (published algorithm; the
    StyleGAN3-style design: attenuation from the transition width, kaiser beta from the attenuation)."""
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
    if kernel_size % 2 == 0:
        time = torch.arange(-half, half, dtype=torch.float64) + 0.5
    else:
        time = torch.arange(kernel_size, dtype=torch.float64) - half
    filt = 2 * cutoff * window * torch.sinc(2 * cutoff * time)
    return (filt / filt.sum()).float()


def snake(x, alpha, beta=None, logscale=False):
    """backup/flows.py:51-62 (Snake) / 113-126 (SnakeBeta): x + sin^2(a x) / (b + 1e-9), b = a for Snake."""
    a = alpha.view(1, -1, 1)
    b = a if beta is None else beta.view(1, -1, 1)
    if logscale:
        a, b = torch.exp(a), torch.exp(b)
    return x + (1.0 / (b + 1e-9)) * torch.sin(x * a) ** 2


def upsample1d_2x(x, filt):
    """alias-free-torch resample.py UpSample1d(ratio=2): replicate pad 5, depthwise transposed conv stride 2 with the
    12-tap filter, x2 gain, crop 15 each side."""
    C = x.shape[1]
    x = F.pad(x, (5, 5), mode="replicate")
    x = 2 * F.conv_transpose1d(x, filt.view(1, 1, -1).expand(C, -1, -1), stride=2, groups=C)
    return x[..., 15:-15]


def downsample1d_2x(x, filt):
    """alias-free-torch resample.py DownSample1d(ratio=2) = LowPassFilter1d(stride=2): replicate pad (5, 6), depthwise
    conv stride 2."""
    C = x.shape[1]
    x = F.pad(x, (5, 6), mode="replicate")
    return F.conv1d(x, filt.view(1, 1, -1).expand(C, -1, -1), stride=2, groups=C)


def activation1d(x, alpha, beta=None, logscale=False):
    """alias-free-torch act.py Activation1d(up_ratio=2, down_ratio=2, kernel 12): up -> act -> down."""
    filt = kaiser_sinc_filter1d(0.25, 0.3, 12).to(x.dtype)
    return downsample1d_2x(snake(upsample1d_2x(x, filt), alpha, beta, logscale), filt)


def res_stack(sd, x, nums=6, base=2):
    """backup/flows.py:172-191: x += conv3(lrelu(conv3_dil(lrelu(x)))), LeakyReLU slope 0.01, dilation base**i"""
    for i in range(nums):
        h = F.leaky_relu(x, 0.01)
        h = F.conv1d(h, wn_weight(sd, f"layers.{i}.1"), sd[f"layers.{i}.1.bias"], dilation=base ** i, padding=base ** i)
        h = F.leaky_relu(h, 0.01)
        h = F.conv1d(h, wn_weight(sd, f"layers.{i}.3"), sd[f"layers.{i}.3.bias"], padding=1)
        x = x + h
    return x


def melvae_encoder(sd, x, down_factors, stacks=6, base=2):
    """backup/flows.py:194-241 (keys generator.N[.layer]): conv k3 -> lrelu(.2) -> [conv k=2f stride f pad (2f-1)//2 ->
    ResStack -> lrelu(.2)]* -> conv k3"""
    x = F.leaky_relu(F.conv1d(x, wn_weight(sd, "generator.0.layer"), sd["generator.0.layer.bias"], padding=1), 0.2)
    i = 2
    for f in down_factors:
        x = F.conv1d(x, wn_weight(sd, f"generator.{i}.layer"), sd[f"generator.{i}.layer.bias"], stride=f,
                     padding=(2 * f - 1) // 2)
        x = res_stack(_sub(sd, f"generator.{i + 1}."), x, stacks, base)
        x = F.leaky_relu(x, 0.2)
        i += 3
    return F.conv1d(x, wn_weight(sd, f"generator.{i}.layer"), sd[f"generator.{i}.layer.bias"], padding=1)


def flows_conv1d(x, w, b, dilation=1, causal=False):
    """backup/flows.py:548-605: 'same' padding (k d - d)/2, or causal left padding d (k-1)"""
    k = w.shape[-1]
    if causal:
        return F.conv1d(F.pad(x, (dilation * (k - 1), 0)), w, b, dilation=dilation)
    return F.conv1d(x, w, b, dilation=dilation, padding=(k * dilation - dilation) // 2)


def flows_conv_transpose1d(x, w, b, stride, causal=False):
    """backup/flows.py:337-387: k == 2*stride; causal: padding 0 and the last `stride` outputs trimmed"""
    k = w.shape[-1]
    if causal:
        return F.conv_transpose1d(x, w, b, stride=stride)[:, :, :-stride]
    return F.conv_transpose1d(x, w, b, stride=stride, padding=(k - stride) // 2)


def _amp_act(sd, pre, x, h):
    beta = sd[pre + ".act.beta"] if h["activation"] == "snakebeta" else None
    return activation1d(x, sd[pre + ".act.alpha"], beta, h["snake_logscale"])


def amp_block(sd, x, h, dilations):
    """AMPBlock1 (backup/flows.py:279-288) when h.resblock == '1', else AMPBlock2 (325-331)"""
    c = h["causal"]
    if h["resblock"] == "1":
        for j, d in enumerate(dilations):
            xt = _amp_act(sd, f"activations.{2 * j}", x, h)
            xt = flows_conv1d(xt, wn_weight(sd, f"convs1.{j}"), sd[f"convs1.{j}.bias"], d, c)
            xt = _amp_act(sd, f"activations.{2 * j + 1}", xt, h)
            x = flows_conv1d(xt, wn_weight(sd, f"convs2.{j}"), sd[f"convs2.{j}.bias"], 1, c) + x
        return x
    for j, d in enumerate(dilations):
        xt = _amp_act(sd, f"activations.{j}", x, h)
        x = flows_conv1d(xt, wn_weight(sd, f"convs.{j}"), sd[f"convs.{j}.bias"], d, c) + x
    return x


def melvae_decode(sd, z, h):
    """BigVGANFlowVAE.inference_from_latents after sampling (backup/flows.py:509-529)"""
    x = flows_conv1d(z, wn_weight(sd, "conv_pre"), sd["conv_pre.bias"], 1, False)
    nk = len(h["resblock_kernel_sizes"])
    for i, u in enumerate(h["upsample_rates"]):
        x = flows_conv_transpose1d(x, wn_weight(sd, f"ups.{i}.0"), sd[f"ups.{i}.0.bias"], u, h["causal"])
        xs = None
        for j in range(nk):
            y = amp_block(_sub(sd, f"resblocks.{i * nk + j}."), x, h, h["resblock_dilation_sizes"][j])
            xs = y if xs is None else xs + y
        x = xs / nk
    x = _amp_act(sd, "activation_post", x, h)
    x = flows_conv1d(x, wn_weight(sd, "conv_post"), sd["conv_post.bias"], 1, h["causal"])
    return torch.tanh(x)


def wn_stack(sd, x, n_layers=4, dilation_rate=1, causal=True):
    """WN (backup/flows.py:659-687) with g=None, mask of ones, no dropout"""
    H = x.shape[1]
    out = torch.zeros_like(x)
    for i in range(n_layers):
        x_in = flows_conv1d(x, wn_weight(sd, f"in_layers.{i}"), sd[f"in_layers.{i}.bias"], dilation_rate ** i, causal)
        acts = torch.tanh(x_in[:, :H]) * torch.sigmoid(x_in[:, H:])
        rs = flows_conv1d(acts, wn_weight(sd, f"res_skip_layers.{i}"), sd[f"res_skip_layers.{i}.bias"], 1, causal)
        if i < n_layers - 1:
            x = x + rs[:, :H]
            out = out + rs[:, H:]
        else:
            out = out + rs
    return out


def residual_coupling_block(sd, x, n_flows=4, n_layers=4, causal=True):
    """ResidualCouplingBlock forward direction (backup/flows.py:736-751,783-786), mean_only, followed by Flip each"""
    half = x.shape[1] // 2
    for f in range(n_flows):
        p = f"flows.{2 * f}."
        x0, x1 = x[:, :half], x[:, half:]
        hdn = flows_conv1d(x0, sd[p + "pre.weight"], sd[p + "pre.bias"], 1, causal)
        hdn = wn_stack(_sub(sd, p + "enc."), hdn, n_layers, 1, causal)
        m = flows_conv1d(hdn, sd[p + "post.weight"], sd[p + "post.bias"], 1, causal)
        x = torch.cat([x0, m + x1], 1)
        x = torch.flip(x, [1])
    return x


def melvae_forward(sd, wav, eps, h):
    """BigVGANFlowVAE.forward (backup/flows.py:457-493) with the reparameterisation noise passed in"""
    enc = melvae_encoder(_sub(sd, "audio_encoder."), wav, h["downsample_rates"])
    m_q, logs_q = torch.split(enc, h["latent_dim"], dim=1)
    z = m_q + eps * torch.exp(logs_q)
    z_p = residual_coupling_block(_sub(sd, "flow."), z, causal=h["causal"])
    return melvae_decode(sd, z, h), z_p, logs_q


# ---------------------------------------------------------------------------------------------- Llasa (model_sigmaVAE.py)
# The decoder under Llasa is third-party: `transformers` LlamaModel (version unpinned by the reference; 5.15.0 here),
# restated below from its published algorithm (pre-norm decoder: RMSNorm, rotary over the whole head, grouped-query causal
# attention, SwiGLU MLP, final RMSNorm; llama3 rope scaling).  PINNED by tests/golden/llasa.npz (reference Llasa run over
# a tiny locally-built Llama).
def llama_inv_freq(head_dim, theta, scaling=None):
    """rope frequencies incl. the 'llama3' scaling rule (long wavelengths / factor, smooth blend in the medium band)"""
    inv = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.float32) / head_dim))
    if not scaling or scaling.get("rope_type", scaling.get("type")) in (None, "default"):
        return inv
    assert scaling.get("rope_type", scaling.get("type")) == "llama3"
    factor, lo, hi = scaling["factor"], scaling["low_freq_factor"], scaling["high_freq_factor"]
    old = scaling["original_max_position_embeddings"]
    wavelen = 2 * math.pi / inv
    scaled = torch.where(wavelen > old / lo, inv / factor, inv)
    smooth = (old / wavelen - lo) / (hi - lo)
    smoothed = (1 - smooth) * scaled / factor + smooth * scaled
    medium = ~(wavelen < old / hi) & ~(wavelen > old / lo)
    return torch.where(medium, smoothed, scaled)


def llama_rms_norm(x, w, eps):
    return w * (x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps))


def llama_layer(sd, x, cos, sin, allow, H, Hkv, eps):
    """one LlamaDecoderLayer: x + o(attn(rope(q), rope(k), v)) ; x + down(silu(gate) * up).  allow: [B,1,L,L] bool"""
    B, L, D = x.shape
    hd = D // H
    h = llama_rms_norm(x, sd["input_layernorm.weight"], eps)
    q = (h @ sd["self_attn.q_proj.weight"].T).view(B, L, H, hd).transpose(1, 2)
    k = (h @ sd["self_attn.k_proj.weight"].T).view(B, L, Hkv, hd).transpose(1, 2)
    v = (h @ sd["self_attn.v_proj.weight"].T).view(B, L, Hkv, hd).transpose(1, 2)

    def rope(t):
        return t * cos + torch.cat([-t[..., hd // 2:], t[..., :hd // 2]], -1) * sin
    q, k = rope(q), rope(k)
    k = k.repeat_interleave(H // Hkv, 1)
    v = v.repeat_interleave(H // Hkv, 1)
    dots = (q @ k.transpose(-1, -2)) * hd ** -0.5
    dots = dots.masked_fill(~allow, torch.finfo(dots.dtype).min)
    a = (dots.softmax(-1) @ v).transpose(1, 2).reshape(B, L, D)
    x = x + a @ sd["self_attn.o_proj.weight"].T
    h = llama_rms_norm(x, sd["post_attention_layernorm.weight"], eps)
    m = F.silu(h @ sd["mlp.gate_proj.weight"].T) * (h @ sd["mlp.up_proj.weight"].T)
    return x + m @ sd["mlp.down_proj.weight"].T


def llama_model(sd, cfg, inputs_embeds, attention_mask):
    """LlamaModel.forward(inputs_embeds=, attention_mask=) -> last hidden state (positions = arange(L))"""
    B, L, D = inputs_embeds.shape
    H, Hkv = cfg["num_attention_heads"], cfg["num_key_value_heads"]
    inv = llama_inv_freq(D // H, cfg["rope_theta"], cfg.get("rope_scaling"))
    fr = torch.arange(L, dtype=torch.float32)[:, None] * inv[None, :]
    emb = torch.cat([fr, fr], -1)
    cos, sin = emb.cos()[None, None], emb.sin()[None, None]
    allow = torch.ones(L, L, dtype=torch.bool).tril()[None, None] & (attention_mask > 0)[:, None, None, :]
    x = inputs_embeds
    for i in range(cfg["num_hidden_layers"]):
        x = llama_layer(_sub(sd, f"layers.{i}."), x, cos, sin, allow, H, Hkv, cfg["rms_norm_eps"])
    return llama_rms_norm(x, sd["norm.weight"], cfg["rms_norm_eps"])


def llasa_forward(sd, cfg, batch, eps, std=0.5):
    """Llasa.forward (model_sigmaVAE.py:53-104) with the sampling noise passed in. Returns the reference's dict."""
    m = _sub(sd, "base_model.model.")
    text = m["embed_tokens.weight"][batch["input_ids"]]
    lat = batch["audio_latents"] + std * eps                                        # sample(), dist_type 'fix'
    audio = lat @ sd["audio_linear.weight"].T + sd["audio_linear.bias"]
    x = audio * batch["audio_mask"].unsqueeze(-1) + text * batch["ids_mask"].unsqueeze(-1)
    hidden = llama_model(m, cfg["llama"], x, batch["ids_mask"] + batch["audio_mask"])
    h = hidden @ sd["distribution_linear.0.weight"].T + sd["distribution_linear.0.bias"]
    pred = F.gelu(h) @ sd["distribution_linear.2.weight"].T + sd["distribution_linear.2.bias"]
    kl = ((pred - batch["audio_distribution_l"]) ** 2 / (2 * std * std)).sum(2) / lat.shape[-1]
    tm, em = batch["target_mask"], batch["end_mask"]
    return {"audio_loss": (kl * tm).sum() / tm.sum(), "end_loss": (kl * em).sum() / em.sum(), "pre_mean": pred,
            "ground_truth_audio_latents": lat}


# ---------------------------------------------------------------------------------------------- round 2: wrappers
def llasa_model_forward(sd, cfg, batch, mean_stdev_fn):
    """model.py:52-107 (the Stable-Audio-VAE `Llasa`): no sampling of the inputs; the head predicts mean || log-scale;
    loss = KL(N(label_mean, 1.25 label_std) || N(pred_mean, exp(pred_log_scale))) summed over the latent dim / dim, masked
    means.  `mean_stdev_fn` stands for twj_utils.get_mean_stdev_from_stableaudio2_latents, which the reference tree does not
    contain (dangling symlink): [B, 2*lat, L] -> (mean, stdev) [B, lat, L]; parity of that callable is UNPINNED."""
    m = _sub(sd, "base_model.model.")
    text = m["embed_tokens.weight"][batch["input_ids"]]
    lat = batch["audio_latents"]
    audio = lat @ sd["audio_linear.weight"].T + sd["audio_linear.bias"]
    x = audio * batch["audio_mask"].unsqueeze(-1) + text * batch["ids_mask"].unsqueeze(-1)
    hidden = llama_model(m, cfg["llama"], x, batch["ids_mask"] + batch["audio_mask"])
    h = hidden @ sd["distribution_linear.0.weight"].T + sd["distribution_linear.0.bias"]
    pred = F.gelu(h) @ sd["distribution_linear.2.weight"].T + sd["distribution_linear.2.bias"]
    mean1, std1 = mean_stdev_fn(batch["audio_distribution_l"].transpose(1, 2))
    mean1, std1 = mean1.transpose(1, 2), std1.transpose(1, 2) * 1.25                   # model.py:85-87
    mean2, logs2 = pred.chunk(2, dim=2)
    std2 = torch.exp(logs2)
    # KL(N(m1, s1) || N(m2, s2)) = log(s2 / s1) + (s1^2 + (m1 - m2)^2) / (2 s2^2) - 1/2   (torch.distributions, model.py:93-96)
    kl = torch.log(std2 / std1) + (std1 ** 2 + (mean1 - mean2) ** 2) / (2 * std2 ** 2) - 0.5
    kl = kl.sum(2) / lat.shape[-1]
    tm, em = batch["target_mask"], batch["end_mask"]
    return {"audio_loss": (kl * tm).sum() / tm.sum(), "end_loss": (kl * em).sum() / em.sum(), "pre_mean": mean2,
            "pre_log_scale": logs2}


def conditioning_inputs(cond, cross_ids, global_ids):
    """models/diffusion.py:131-208 for cross-attention + global ids: cat over the sequence / channel dimension."""
    out = {}
    if cross_ids:
        out["cross_attn_cond"] = torch.cat([cond[k][0] for k in cross_ids], 1)
        out["cross_attn_cond_mask"] = torch.cat([cond[k][1] for k in cross_ids], 1)
    if global_ids:
        g = torch.cat([cond[k][0] for k in global_ids], -1)
        out["global_embed"] = g.squeeze(1) if g.dim() == 3 else g
    return out


def training_step(sd_dit, cfg, sd_vae, vae_strides, reals, cond_inputs, t, noise, objective="v", padding_mask=None,
                  pre_encoded=False, scale=1.0):
    """DiffusionCondTrainingWrapper.training_step (training/diffusion.py:311-437) given the step's t and noise draws:
    pretransform.encode under no_grad (340-351) or the pre-encoded / scale rule (353-356), the padding mask resized with
    nearest interpolation to the latent length (349-351), noising + target (365-379), model (390), masked MSE (393-399)."""
    x = reals
    if not pre_encoded:
        with torch.no_grad():
            x = pretransform_encode(sd_vae, reals, vae_strides, True, scale)
        if padding_mask is not None:
            padding_mask = F.interpolate(padding_mask.unsqueeze(1).float(), size=x.shape[2], mode="nearest").squeeze(1).bool()
    elif scale != 1.0:
        x = x / scale
    loss, out, xt, target = train_step_loss(sd_dit, cfg, x, noise, t, objective, padding_mask, **cond_inputs)
    return loss


def generate(sd_dit, cfg, sd_vae, vae_strides, noise, cond_inputs, steps, cfg_scale, objective, scale=1.0, neg=None,
             return_latents=False):
    """generate_diffusion_cond (inference/generation.py:90-250) after the seeded noise draw (138-142): rectified flow ->
    sample_rf -> sample_discrete_euler (sampling.py:200-232, 24-45); v -> the in-tree DDIM sampler (sampling.py:47-86; the
    reference itself routes v to third-party k-diffusion); then pretransform.decode (247)."""
    kw = dict(cond_inputs)
    if neg is not None:
        kw.update(negative_cross_attn_cond=neg["cross_attn_cond"], negative_cross_attn_mask=neg["cross_attn_cond_mask"])
    fn = lambda x_, t_: dit_forward(sd_dit, cfg, x_, t_, cfg_scale=cfg_scale, **kw)
    lat = sample_euler(fn, noise, steps) if objective == "rectified_flow" else sample_ddim(fn, noise, steps, 0.0)
    if return_latents:
        return lat
    return pretransform_decode(sd_vae, lat, vae_strides, True, scale, True)


def prepare_audio(audio, target_length, target_channels):
    """inference/utils.py:20-40 for in_sr == target_sr (the resampler is third-party torchaudio): PadCrop(randomize=False)
    (data/utils.py:8-20: crop or zero-pad on the right), batch dimension, mono <-> stereo (utils.py:5-18)"""
    n, s_ = audio.shape
    out = audio.new_zeros([n, target_length])
    out[:, :min(s_, target_length)] = audio[:, :target_length]
    out = out.unsqueeze(0)
    if target_channels == 1:
        out = out.mean(1, keepdim=True)
    elif target_channels == 2:
        if out.shape[1] == 1:
            out = out.repeat(1, 2, 1)
        elif out.shape[1] > 2:
            out = out[:, :2, :]
    return out


def build_mask(sample_size, mask_args):
    """inference/generation.py:254-274: soft inpainting mask with Hann ramps (1 = keep the input)"""
    maskstart = math.floor(mask_args["maskstart"] / 100.0 * sample_size)
    maskend = math.ceil(mask_args["maskend"] / 100.0 * sample_size)
    sl = round(mask_args["softnessL"] / 100.0 * sample_size)
    sr = round(mask_args["softnessR"] / 100.0 * sample_size)
    hl = torch.hann_window(sl * 2, periodic=False)[:sl]
    hr = torch.hann_window(sr * 2, periodic=False)[sr:]
    mask = torch.zeros((sample_size))
    mask[maskstart:maskend] = 1
    mask[maskstart:maskstart + sl] = hl
    mask[maskend - sr:maskend] = hr
    if mask_args["marination"] > 0:
        mask = mask * (1 - mask_args["marination"])
    return mask


def generate_variation(sd_dit, cfg, sd_vae, vae_strides, noise, init_audio, cond_inputs, steps, cfg_scale, init_noise_level,
                       scale=1.0, mask_args=None, return_latents=False, audio_channels=2):
    """generate_diffusion_cond(init_audio=...) for rectified flow (generation.py:164-228 -> sampling.py:200-232): prepare the init
    audio, encode it if the model is latent, x = init (1 - sigma_max) + noise sigma_max with sigma_max = init_noise_level, Euler
    from sigma_max.  With mask_args the reference cuts / pastes and builds the mask (186-214) but passes neither the mask nor a
    sigma_max on: sigma_max stays 1 and the init data drops out of x."""
    B, _, T_ = noise.shape
    if sd_vae is not None:
        ratio = 1
        for s_ in vae_strides:
            ratio *= s_
        init = prepare_audio(init_audio, T_ * ratio, audio_channels)
        init = pretransform_encode(sd_vae, init, vae_strides, True, scale)
    else:
        init = prepare_audio(init_audio, T_, noise.shape[1])
    init = init.repeat(B, 1, 1)
    sigma_max = min(init_noise_level, 1.0)
    if mask_args is not None:
        build_mask(T_, mask_args)                    # (built and dropped, as the reference's rectified-flow branch does)
        sigma_max = 1.0
    x = init * (1 - sigma_max) + noise * sigma_max
    fn = lambda x_, t_: dit_forward(sd_dit, cfg, x_, t_, cfg_scale=cfg_scale, **cond_inputs)
    lat = sample_euler(fn, x, steps, sigma_max)
    if return_latents or sd_vae is None:
        return lat
    return pretransform_decode(sd_vae, lat, vae_strides, True, scale, True)


def export_int16(audio):
    """infer_0723.py:292-293: "b d n -> d (b n)", divide by the peak, clamp, * 32767, int16"""
    o = audio.permute(1, 0, 2).reshape(audio.shape[1], -1).float()
    return (o / o.abs().max()).clamp(-1, 1).mul(32767).to(torch.int16)
