"""Drop-in for the reference's task model `model_sigmaVAE.Llasa` (model_sigmaVAE.py:8-183, imported by
train_offline.py:19): same constructor, `forward` / `infer` / `sample` / `kl` signatures, returned dict and state-dict
keys (`base_model.model.layers.N.self_attn.q_proj.weight`, ..., `audio_linear.*`, `distribution_linear.{0,2}.*`).

`base_model` replaces `transformers.AutoModelForCausalLM.from_pretrained(path)` for Llama checkpoints: it reads the
local HF directory (config.json + safetensors / .bin) itself and runs the decoder layers on this package's HIP kernels
(llama_ops.py).  q/k/v and up/gate are held as fused parameters (one GEMM each); state_dict() / load_state_dict() split
and merge them under the HF names.  Parameters are fp32 masters with bf16 compute copies (the reference's
use_flash_attention=True path is bf16 weights + fp16 heads; =False is fp32) - the flag is accepted and ignored.
Every arithmetic step runs in a HIP kernel; torch supplies tensors, RNG and the autograd tape.
"""
import json
import os

import torch
from torch import nn

from . import functional as Fn
from . import llama_ops as LO
from . import ops
from .dit_ops import BF16, F32, GradOut, bf16_of, f32_of


def _need_gpu(t):
    if not t.is_cuda:
        raise RuntimeError("kalle_audio_amd Llasa modules run on an MI355X GPU only (no CPU fallback)")


# ------------------------------------------------------------------------------------------------ autograd shims
class LlamaLayerFn(torch.autograd.Function):
    """one decoder layer, forward + backward in llama_ops; parameter gradients go straight into the trainer's flat
    buckets when the layer carries `_kalle_grad_sinks` (engine.DataParallelTrainer), else back through autograd"""

    @staticmethod
    def forward(ctx, layer, x, mask8, rope, *params):
        B, L, Dm = x.shape
        p = LO.layer_params(layer)
        y, sv = LO.layer_fwd(p, x.contiguous().view(B * L, Dm), B, L, rope, mask8)
        ctx.layer, ctx.sv, ctx.aux, ctx.dims = layer, sv, (mask8, rope), (B, L, Dm)
        layer._kalle_last_rows = B * L                  # (the trainer picks its gradient-clearing rule from it)
        return y.view(B, L, Dm)

    @staticmethod
    def backward(ctx, g):
        layer, sv = ctx.layer, ctx.sv
        mask8, rope = ctx.aux
        B, L, Dm = ctx.dims
        p = LO.layer_params(layer)
        gf = g.contiguous().view(B * L, Dm)
        sinks = getattr(layer, "_kalle_grad_sinks", None)
        go = GradOut(sinks, getattr(layer, "_kalle_grad_accumulate", False)) if sinks else GradOut()
        go.wgrad_overwrite = bool(sinks) and getattr(layer, "_kalle_wgrad_overwrite", False)
        sh = Fn._GRAD_SHADOW.pop(gf.data_ptr(), None)
        g_bf16 = sh[1] if sh is not None and sh[0] == ("llama", layer.layer_idx + 1) and sh[1].shape == gf.shape else None
        if len(Fn._GRAD_SHADOW) > 8:
            Fn._GRAD_SHADOW.clear()
        dx, dxb, go = LO.layer_bwd(p, sv, gf, B, L, rope, mask8, go=go, g_bf16=g_bf16, want_dx_bf16=layer.layer_idx > 0)
        if dxb is not None:
            Fn._GRAD_SHADOW[dx.data_ptr()] = (("llama", layer.layer_idx), dxb)
        ctx.sv = None
        hook = getattr(layer, "_kalle_on_backward_done", None)
        if hook is not None:
            hook(layer)
        return (None, dx.view(B, L, Dm), None, None) + tuple(go.grads.get(n) for n in LO.PARAM_ORDER)


class EmbedMixFn(torch.autograd.Function):
    """input_embed = audio_embed * audio_mask + embed_tokens(input_ids) * ids_mask   (model_sigmaVAE.py:66, 73)"""

    @staticmethod
    def forward(ctx, ids, table, audio, ids_mask, audio_mask):
        B, L = ids.shape
        D = table.shape[1]
        a = audio.contiguous().view(B * L, D)
        out = ops.embed_mix_fwd(ids.contiguous().view(-1), f32_of(table), a, ids_mask.view(-1), audio_mask.view(-1))
        ctx.save_for_backward(ids, ids_mask, audio_mask)
        ctx.table, ctx.adt, ctx.shape = table, audio.dtype, (B, L, D)
        return out.view(B, L, D)

    @staticmethod
    def backward(ctx, g):
        ids, im, am = ctx.saved_tensors
        B, L, D = ctx.shape
        table = ctx.table
        sink = getattr(table, "_kalle_grad_sink", None)
        dt = sink if sink is not None else (torch.zeros_like(table, dtype=F32) if ctx.needs_input_grad[1] else None)
        da = ops.embed_mix_bwd(g.contiguous().view(B * L, D), ids.contiguous().view(-1), im.view(-1), am.view(-1), dtable=dt,
                               want_daudio=ctx.needs_input_grad[2])
        if da is not None:
            da = Fn._like(da, ctx.adt).view(B, L, D)
        return None, (None if sink is not None else dt), da, None, None


class GELUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return ops.gelu_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.gelu_bwd(Fn._like(dy.contiguous(), x.dtype), x)


class GaussKLFn(torch.autograd.Function):
    """(audio_loss, end_loss) of model_sigmaVAE.py:85-95: masked means of KL(N(pred, s) || N(label, s)) / latent_dim"""

    @staticmethod
    def forward(ctx, pred, label, target_mask, end_mask, std):
        d = pred.shape[-1]
        p2 = Fn._to_f32(pred.contiguous()).view(-1, d)
        l2 = Fn._to_f32(label.contiguous()).view(-1, d)
        sums = ops.gauss_kl_fwd(p2, l2, target_mask.view(-1), end_mask.view(-1), std)
        ctx.save_for_backward(p2, l2, target_mask, end_mask, sums)
        ctx.std, ctx.pdt, ctx.shape = std, pred.dtype, pred.shape
        # the two ratios are 1-element reductions of four numbers the kernel produced (plumbing, like a .view())
        return sums[0] / sums[1], sums[2] / sums[3]

    @staticmethod
    def backward(ctx, ga, gb):
        p2, l2, tm, em, sums = ctx.saved_tensors
        dp = ops.gauss_kl_bwd(p2, l2, tm.view(-1), em.view(-1), sums, ga.float().reshape(1).contiguous(),
                              gb.float().reshape(1).contiguous(), ctx.std)
        return Fn._like(dp, ctx.pdt).view(ctx.shape), None, None, None, None


# ------------------------------------------------------------------------------------------------ modules
class Linear(nn.Linear):
    """nn.Linear whose forward / backward are kalle_gemm_bf16 launches"""

    def forward(self, x):
        _need_gpu(x)
        return Fn.linear(x, self.weight, self.bias, out_dtype=F32)


class GELU(nn.Module):
    def forward(self, x):
        return GELUFn.apply(x)


class _FusedLinear(nn.Module):
    """several bias-free nn.Linear that share their input, stored as ONE weight [sum(out_i), in] so they run as one
    GEMM; `parts` = ((hf_name, out_features), ...) are the names they carry in state_dict()"""

    def __init__(self, in_features, parts):
        super().__init__()
        self.parts = tuple(parts)
        self.weight = nn.Parameter(torch.empty(sum(n for _, n in parts), in_features))
        nn.init.normal_(self.weight, std=0.02)


class _Holder(nn.Module):
    def __init__(self, out_features, in_features):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        nn.init.normal_(self.weight, std=0.02)


def _split_hooks(owner, fused_attr):
    """state_dict(): fused weight -> HF names; load_state_dict(): HF names -> fused weight"""
    def save_hook(module, sd, prefix, local_metadata):
        key = prefix + fused_attr + ".weight"
        if key in sd:
            w = sd.pop(key)
            o = 0
            for name, n in getattr(module, fused_attr).parts:
                sd[prefix + name + ".weight"] = w[o:o + n]
                o += n

    def load_hook(module, sd, prefix, local_metadata, strict, missing, unexpected, errors):
        parts = getattr(module, fused_attr).parts
        keys = [prefix + name + ".weight" for name, _ in parts]
        if all(k in sd for k in keys):
            sd[prefix + fused_attr + ".weight"] = torch.cat([sd.pop(k) for k in keys], 0)

    owner._register_state_dict_hook(save_hook)
    owner._register_load_state_dict_pre_hook(load_hook, with_module=True)


class LlamaRMSNorm(nn.Module):
    def __init__(self, hidden_size, eps=1e-6):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(hidden_size))
        self.variance_epsilon = eps

    def forward(self, x):
        return Fn.RMSNormFn.apply(x, self.weight, self.variance_epsilon)


class LlamaAttention(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.num_heads, self.num_kv_heads = cfg["num_attention_heads"], cfg["num_key_value_heads"]
        hd = cfg.get("head_dim") or cfg["hidden_size"] // self.num_heads
        if hd != 64 or cfg.get("attention_bias"):
            raise NotImplementedError("the attention kernels are built for head_dim 64 without projection biases")
        D = cfg["hidden_size"]
        self.qkv_proj = _FusedLinear(D, (("q_proj", self.num_heads * 64), ("k_proj", self.num_kv_heads * 64),
                                         ("v_proj", self.num_kv_heads * 64)))
        self.o_proj = _Holder(D, self.num_heads * 64)
        _split_hooks(self, "qkv_proj")


class LlamaMLP(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        if cfg.get("mlp_bias") or cfg.get("hidden_act", "silu") != "silu":
            raise NotImplementedError("LlamaMLP: silu without biases only")
        D, I = cfg["hidden_size"], cfg["intermediate_size"]
        # value half first, gate half second: the layout of the fused GEMM + SwiGLU epilogue
        self.up_gate_proj = _FusedLinear(D, (("up_proj", I), ("gate_proj", I)))
        self.down_proj = _Holder(D, I)
        _split_hooks(self, "up_gate_proj")


class LlamaDecoderLayer(nn.Module):
    _kalle_bucket_unit = True      # engine.DataParallelTrainer: one gradient bucket / all-reduce per layer

    def __init__(self, cfg, layer_idx):
        super().__init__()
        self.layer_idx = layer_idx
        self.self_attn = LlamaAttention(cfg)
        self.mlp = LlamaMLP(cfg)
        self.input_layernorm = LlamaRMSNorm(cfg["hidden_size"], cfg.get("rms_norm_eps", 1e-6))
        self.post_attention_layernorm = LlamaRMSNorm(cfg["hidden_size"], cfg.get("rms_norm_eps", 1e-6))

    def forward(self, x, mask8, rope):
        have = dict(self.named_parameters())
        return LlamaLayerFn.apply(self, x, mask8, rope, *[have[n] for n in LO.PARAM_ORDER])


class _Embedding(nn.Module):
    def __init__(self, num_embeddings, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(num_embeddings, dim))
        nn.init.normal_(self.weight, std=0.02)
        self.weight._kalle_wants_sink = True     # the trainer lets the backward scatter-add into its flat gradient

    def forward(self, ids):
        _need_gpu(ids)
        z = torch.zeros(ids.shape + (self.weight.shape[1],), device=ids.device, dtype=BF16)
        one = torch.ones(ids.shape, device=ids.device, dtype=F32)
        return EmbedMixFn.apply(ids, self.weight, z, one, torch.zeros_like(one))


class LlamaModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        self.embed_tokens = _Embedding(cfg["vocab_size"], cfg["hidden_size"])
        self.layers = nn.ModuleList([LlamaDecoderLayer(cfg, i) for i in range(cfg["num_hidden_layers"])])
        self.norm = LlamaRMSNorm(cfg["hidden_size"], cfg.get("rms_norm_eps", 1e-6))
        hd = cfg.get("head_dim") or cfg["hidden_size"] // cfg["num_attention_heads"]
        self._inv_freq = LO.inv_freq(hd, cfg.get("rope_theta", 10000.0), cfg.get("rope_scaling"))
        self._rope_cache = {}

    def _rope(self, L, device):
        key = (L, str(device))
        if key not in self._rope_cache:
            self._rope_cache = {key: LO.rope_tables(L, self._inv_freq, device)}
        return self._rope_cache[key]

    def forward(self, input_ids=None, attention_mask=None, inputs_embeds=None, **kwargs):
        """returns (last_hidden_state,) - the reference indexes [0] (model_sigmaVAE.py:78-81, 123-125)"""
        x = self.embed_tokens(input_ids) if inputs_embeds is None else inputs_embeds
        _need_gpu(x)
        B, L, _ = x.shape
        x = Fn._to_f32(x.contiguous())
        mask8 = Fn._mask8(attention_mask > 0) if attention_mask is not None else None
        rope = self._rope(L, x.device)
        for layer in self.layers:
            x = layer(x, mask8, rope)
        return (self.norm(x),)


    # ---- inference with a KV cache (batch 1) ---------------------------------------------------------------------
    def init_cache(self, max_len, device):
        """one bf16 [max_len, 2 * kv_heads * 64] buffer per layer (un-rotated k | v rows)"""
        w = 2 * self.cfg["num_key_value_heads"] * 64
        return {"kv": [torch.zeros((max_len, w), device=device, dtype=BF16) for _ in self.layers], "len": 0,
                "rope": LO.rope_tables(max_len, self._inv_freq, device)}

    @torch.no_grad()
    def forward_cached(self, inputs_embeds, cache):
        """appends the positions of `inputs_embeds` [1, n, D] to the cache and returns their last hidden states"""
        _need_gpu(inputs_embeds)
        _, n, Dm = inputs_embeds.shape
        t0 = cache["len"]
        if t0 + n > cache["kv"][0].shape[0]:
            raise ValueError("KV cache too short")
        x = Fn._to_f32(inputs_embeds.contiguous()).view(n, Dm)
        if n == 1:
            # one generated frame: the whole stack is sequenced by kalle_llama_decode_step (one host call)
            plan = cache.get("plan")
            if plan is None:
                ps = [LO.layer_params(layer) for layer in self.layers]
                plan = cache["plan"] = ops.llama_decode_plan(
                    [(p.g1, p.wqkv, p.wo, p.g2, p.wug, p.wdown, kv) for p, kv in zip(ps, cache["kv"])],
                    ps[0].H, ps[0].Hkv, ps[0].wug.shape[0] // 2, x.device)
                plan["eps"] = ps[0].eps
            x = ops.llama_decode_step(plan, x.view(Dm), t0, cache["kv"][0].shape[0], cache["rope"], plan["eps"])
        else:
            for layer, kv in zip(self.layers, cache["kv"]):
                x = LO.layer_fwd_cached(LO.layer_params(layer), x, kv, t0, cache["rope"])
        cache["len"] = t0 + n
        return self.norm(x.view(1, n, Dm))


class LlamaForCausalLM(nn.Module):
    """the parts of transformers' LlamaForCausalLM the task model touches: `.model`, `.config`, `.vocab_size`,
    `resize_token_embeddings`, tied `lm_head.weight` in the state dict"""

    def __init__(self, cfg):
        super().__init__()
        self.config = _Cfg(cfg)
        self.model = LlamaModel(cfg)
        self.lm_head = _Holder(cfg["vocab_size"], cfg["hidden_size"])
        self.vocab_size = cfg["vocab_size"]
        self.tied = bool(cfg.get("tie_word_embeddings", False))
        if self.tied:
            self.lm_head.weight = self.model.embed_tokens.weight

    @classmethod
    def from_pretrained(cls, path, **kwargs):
        with open(os.path.join(path, "config.json")) as f:
            cfg = json.load(f)
        if cfg.get("model_type", "llama") != "llama":
            raise NotImplementedError(f"only Llama checkpoints are supported, got model_type={cfg.get('model_type')!r}")
        model = cls(cfg)
        sd = _read_hf_weights(path)
        if sd is not None:
            if model.tied and "lm_head.weight" not in sd:
                sd["lm_head.weight"] = sd["model.embed_tokens.weight"]
            missing, unexpected = model.load_state_dict(sd, strict=False)
            if missing:
                raise RuntimeError(f"checkpoint {path} lacks {missing}")
        return model

    def resize_token_embeddings(self, n):
        """new rows start at the mean of the old ones (transformers draws them around that mean)"""
        old = self.model.embed_tokens.weight.data
        if n != old.shape[0]:
            new = old.mean(0, keepdim=True).repeat(n, 1)
            new[:min(n, old.shape[0])] = old[:n]
            self.model.embed_tokens.weight = nn.Parameter(new)
            self.model.embed_tokens.weight._kalle_wants_sink = True
            if self.tied:
                self.lm_head.weight = self.model.embed_tokens.weight
            else:
                oh = self.lm_head.weight.data
                nh = oh.mean(0, keepdim=True).repeat(n, 1)
                nh[:min(n, oh.shape[0])] = oh[:n]
                self.lm_head.weight = nn.Parameter(nh)
        self.config.vocab_size = n
        self.vocab_size = n
        return self.model.embed_tokens


class _Cfg(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def _read_hf_weights(path):
    from safetensors.torch import load_file
    idx = os.path.join(path, "model.safetensors.index.json")
    if os.path.exists(idx):
        with open(idx) as f:
            files = sorted(set(json.load(f)["weight_map"].values()))
        sd = {}
        for fn in files:
            sd.update(load_file(os.path.join(path, fn)))
        return {k: v.float() for k, v in sd.items()}
    one = os.path.join(path, "model.safetensors")
    if os.path.exists(one):
        return {k: v.float() for k, v in load_file(one).items()}
    b = os.path.join(path, "pytorch_model.bin")
    if os.path.exists(b):
        return {k: v.float() for k, v in torch.load(b, map_location="cpu").items()}
    return None


class Llasa(nn.Module):
    """model_sigmaVAE.py:8-183"""

    def __init__(self, config, tokenizer, use_flash_attention=True):
        super().__init__()
        self.use_fa = use_flash_attention
        self.base_model = LlamaForCausalLM.from_pretrained(config['llm_model_name_or_path'])
        self.base_model.resize_token_embeddings(len(tokenizer))
        self.base_model.vocab_size = len(tokenizer)
        self.vocab_size = self.base_model.config.vocab_size
        self.hidden_size = self.base_model.config.hidden_size
        self.audio_linear = Linear(config['latent_dim'], config['audio_proj_dim'])
        self.distribution_linear = nn.Sequential(
            Linear(config['audio_proj_dim'], config['latent_dim']),
            GELU(),
            Linear(config['latent_dim'], config['latent_dim']))
        self.init_sigmaVAE()

    def forward(self, input_ids, audio_latents, audio_distribution_l, ids_mask, audio_mask, target_mask, end_mask,
                noise=None):
        """`noise` (extension, for seed-free parity tests): the N(0,1) draw of sample(); default torch.randn_like"""
        _need_gpu(input_ids)
        f = lambda m: m.to(F32).contiguous()
        ids_mask, audio_mask, target_mask, end_mask = f(ids_mask), f(audio_mask), f(target_mask), f(end_mask)
        audio_latents = self.sample(mean=audio_latents, noise=noise)
        audio_embed = self.audio_linear(audio_latents)                                   # b,t,d
        audio_latents_dim = audio_latents.shape[-1]
        input_embed = EmbedMixFn.apply(input_ids, self.base_model.model.embed_tokens.weight, audio_embed, ids_mask,
                                       audio_mask)
        attention_mask = ids_mask + audio_mask     # two 0/1 row masks; >0 is all the decoder reads from it
        hidden = self.base_model.model(inputs_embeds=input_embed, attention_mask=attention_mask)[0]
        x = self.distribution_linear(hidden)                                             # b,t,d2
        audio_loss, end_loss = GaussKLFn.apply(x, audio_distribution_l, target_mask, end_mask, float(self.std))
        return {"audio_loss": audio_loss, "end_loss": end_loss, "pre_mean": x, "pre_log_scale": self.std,
                "ground_truth_audio_latents": audio_latents}

    @torch.no_grad()
    def infer(self, input_ids, audio_latents, end_disp_kl_thres=0.5, max_length=200, sample=False, use_cfg=None,
              flow=None, use_cache=True):
        """model_sigmaVAE.py:106-148: frame-by-frame generation, stopping when KL(N(mean, std) || N(1, e)) / dim drops
        below the threshold.  The reference re-runs the decoder over the whole prefix for every frame (O(T^2) GEMM work);
        with use_cache (default) the prompt is prefilled once and every frame is one single-position pass against a KV
        cache - same arithmetic per position.  use_cache=False reproduces the reference's schedule."""
        ids = input_ids.unsqueeze(0)
        text_embed = self.base_model.model.embed_tokens(ids)
        parts = [text_embed]
        if audio_latents is not None:
            parts.append(self.audio_linear(audio_latents))
        input_embed = torch.cat(parts, dim=1)
        final = []
        model = self.base_model.model
        cache = model.init_cache(input_embed.shape[1] + max_length, input_embed.device) if use_cache else None
        step_in = input_embed
        for i in range(max_length):
            if use_cache:
                hidden = model.forward_cached(step_in, cache)
            else:
                hidden = model(inputs_embeds=input_embed)[0]
            mean2 = self.distribution_linear(hidden[:, -1:, :].contiguous())
            audio_latent = self.sample(mean2)
            final.append(audio_latent)
            # KL(N(m, s) || N(1, e)) = log(e/s) + (s^2 + (m-1)^2) / (2 e^2) - 1/2, summed over the latent dim / dim
            s, e = float(self.std), float(torch.e)
            kl = (torch.log(torch.tensor(e / s)) + (s * s + (mean2.float() - 1.0) ** 2) / (2 * e * e) - 0.5).sum(2)
            kl = kl / mean2.shape[2]
            if kl.item() < end_disp_kl_thres and i > 3:
                break
            step_in = self.audio_linear(audio_latent)
            if not use_cache:
                input_embed = torch.cat((input_embed, step_in), dim=1)
        out = torch.stack(final[:-1], dim=1).squeeze(1).squeeze(2)
        return out.transpose(1, 2)

    def init_sigmaVAE(self):
        self.std = torch.tensor(0.5)

    def sample(self, mean, dist_type='fix', noise=None):
        """model_sigmaVAE.py:153-178"""
        if dist_type == 'fix':
            n = torch.randn_like(mean, dtype=F32) if noise is None else noise.to(F32)
            return ops.axpby(mean.to(F32), n, 1.0, float(self.std))
        if dist_type == 'gaussian':
            value = float(self.std) / 0.8
            std = torch.randn(mean.size(0), device=mean.device, dtype=F32) * value
            while std.dim() < mean.dim():
                std = std.unsqueeze(-1)
            return mean + std * torch.randn_like(mean)
        return mean

    def kl(self, mean):
        """model_sigmaVAE.py:180-183: squared distance to zero"""
        return mean * mean


def sample(mean, dist_type='fix'):
    """module-level twin of Llasa.sample (model_sigmaVAE.py:187-215)"""
    if dist_type == 'fix':
        return ops.axpby(mean.to(F32), torch.randn_like(mean, dtype=F32), 1.0, 0.5)
    if dist_type == 'gaussian':
        std = torch.randn(mean.size(0), device=mean.device, dtype=mean.dtype) * (0.5 / 0.8)
        while std.dim() < mean.dim():
            std = std.unsqueeze(-1)
        return mean + std * torch.randn_like(mean)
    return mean
