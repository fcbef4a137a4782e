"""Drop-in for the reference's Stable-Audio-VAE task model `model.Llasa` (model.py:9-150; imported by train.py:24 and
training_version/melvae/train_melvae.py:24; the model of configs/twj_0828.yaml and configs/vae_12_5_dim1024-sft.yaml):
same constructor, `forward` / `infer` signatures, returned dict and state-dict keys as the sigma-VAE variant
(model_sigmaVAE.py) except that the head predicts mean || log-scale (2 * latent_dim) and the loss is the KL between the
label distribution N(mean, 1.25 stdev) and the predicted one (model.py:84-100).

The label transform `get_mean_stdev_from_stableaudio2_latents` comes from `twj_utils`, which the reference tree does not
contain (dangling symlink).  It is injectable: assign a callable ([B, 2*lat, L] -> (mean, stdev) [B, lat, L])
to `kalle_audio_amd.model.get_mean_stdev_from_stableaudio2_latents`.  The default (None) runs fused inside the KL kernel:
mean, scale = chunk(2); stdev = softplus(scale) + 1e-4, the rule stable_audio_tools/models/bottleneck.py:51-54 applies to
the same encoder output - parity of that default is UNPINNED; everything around it is pinned by tests/golden/model_llasa.npz.

The decoder stack, embedding mix, heads and KV-cached generation are the kernels of model_sigmaVAE.py / llama_ops.py.
"""
import torch
from torch import nn

from . import functional as Fn
from . import ops
from .dit_ops import F32
from .model_sigmaVAE import GELU, EmbedMixFn, Linear, LlamaForCausalLM, _need_gpu

get_mean_stdev_from_stableaudio2_latents = None      # injectable stand-in for the missing twj_utils function (see above)
LABEL_STD_MULT = 1.25                                # model.py:87


class GaussKL2Fn(torch.autograd.Function):
    """(audio_loss, end_loss) of model.py:93-100: masked means of KL(N(label) || N(pred)) summed over the latent / dim"""

    @staticmethod
    def forward(ctx, pred, label_mean, label_std, target_mask, end_mask):
        d2 = pred.shape[-1]
        p2 = Fn._to_f32(pred.contiguous()).view(-1, d2)
        lm = Fn._to_f32(label_mean.contiguous()).view(p2.shape[0], -1)
        ls = Fn._to_f32(label_std.contiguous()).view(p2.shape[0], -1) if label_std is not None else None
        sums = ops.gauss_kl2_fwd(p2, lm, ls, target_mask.view(-1), end_mask.view(-1), LABEL_STD_MULT)
        ctx.save_for_backward(p2, lm, target_mask, end_mask, sums)
        ctx.ls, ctx.pdt, ctx.shape = ls, pred.dtype, pred.shape
        return sums[0] / sums[1], sums[2] / sums[3]

    @staticmethod
    def backward(ctx, ga, gb):
        p2, lm, tm, em, sums = ctx.saved_tensors
        dp = ops.gauss_kl2_bwd(p2, lm, ctx.ls, tm.view(-1), em.view(-1), sums, ga.float().reshape(1).contiguous(),
                               gb.float().reshape(1).contiguous(), LABEL_STD_MULT)
        return Fn._like(dp, ctx.pdt).view(ctx.shape), None, None, None, None


class Llasa(nn.Module):
    """model.py:9-150"""

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
            Linear(config['audio_proj_dim'], config['latent_dim'] * 2),
            GELU(),
            Linear(config['latent_dim'] * 2, config['latent_dim'] * 2))

    def forward(self, input_ids, audio_latents, audio_distribution_l, ids_mask, audio_mask, target_mask, end_mask):
        _need_gpu(input_ids)
        f = lambda m: m.to(F32).contiguous()
        ids_mask, audio_mask, target_mask, end_mask = f(ids_mask), f(audio_mask), f(target_mask), f(end_mask)
        audio_embed = self.audio_linear(audio_latents)                                   # b,t,d   (model.py:66)
        input_embed = EmbedMixFn.apply(input_ids, self.base_model.model.embed_tokens.weight, audio_embed, ids_mask,
                                       audio_mask)                                       # (model.py:65, 70)
        attention_mask = ids_mask + audio_mask
        hidden = self.base_model.model(inputs_embeds=input_embed, attention_mask=attention_mask)[0]
        distribution_p = self.distribution_linear(hidden)                                # b,t,2*lat
        fn = get_mean_stdev_from_stableaudio2_latents
        if fn is None:
            label_mean, label_std = audio_distribution_l, None       # mean | scale rows: transformed inside the kernel
        else:
            mean1, std1 = fn(audio_distribution_l.transpose(1, 2))                       # model.py:84-86 (the caller's code)
            label_mean, label_std = mean1.transpose(1, 2), std1.transpose(1, 2)
        audio_loss, end_loss = GaussKL2Fn.apply(distribution_p, label_mean, label_std, target_mask, end_mask)
        lat = distribution_p.shape[-1] // 2
        return {"audio_loss": audio_loss, "end_loss": end_loss, "pre_mean": distribution_p[..., :lat],
                "pre_log_scale": distribution_p[..., lat:]}

    @torch.no_grad()
    def infer(self, input_ids, audio_latents, end_disp_kl_thres=0.5, max_length=1000, sample=False, use_cfg=None,
              flow=None, use_cache=True):
        """model.py:109-150: frame-by-frame generation; every frame is drawn from N(mean, exp(log_scale)) and generation
        stops when KL(N(mean, s) || N(1, e)) / dim < threshold (and i > 3).  Returns [1, 2*lat, T'] (mean || log-scale rows,
        as the reference stacks `last_disp`).  use_cache: prefill once, then one position per frame against a KV cache
        (same arithmetic per position); use_cache=False re-runs the whole prefix per frame as the reference does."""
        ids = input_ids.unsqueeze(0)
        parts = [self.base_model.model.embed_tokens(ids)]
        if audio_latents is not None:
            parts.append(self.audio_linear(audio_latents))
        input_embed = torch.cat(parts, dim=1)
        final = []
        model = self.base_model.model
        cache = model.init_cache(input_embed.shape[1] + max_length, input_embed.device) if use_cache else None
        step_in = input_embed
        e = float(torch.e)
        for i in range(max_length):
            hidden = model.forward_cached(step_in, cache) if use_cache else model(inputs_embeds=input_embed)[0]
            last_disp = self.distribution_linear(hidden[:, -1:, :].contiguous())
            lat = last_disp.shape[-1] // 2
            mean, logs = last_disp[..., :lat].float(), last_disp[..., lat:].float()
            s = torch.exp(logs)
            audio_latent = ops.axpby(mean.contiguous(), (torch.randn_like(mean) * s).contiguous(), 1.0, 1.0)
            final.append(last_disp)
            # KL(N(m, s) || N(1, e)) = log(e / s) + (s^2 + (m - 1)^2) / (2 e^2) - 1/2    (model.py:133-136)
            kl = ((1.0 - logs) + (s * s + (mean - 1.0) ** 2) / (2 * e * e) - 0.5).sum(2) / lat
            if kl.item() < end_disp_kl_thres and i > 3:
                break
            step_in = self.audio_linear(audio_latent)
            if not use_cache:
                input_embed = torch.cat((input_embed, step_in), dim=1)
        out = torch.stack(final[:-1], dim=1).squeeze(1).squeeze(2)
        return out.transpose(1, 2)
