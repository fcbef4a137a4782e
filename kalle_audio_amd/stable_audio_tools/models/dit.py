"""Drop-in for stable_audio_tools/models/dit.py:13-379 (DiffusionTransformer, "continuous_transformer" branch).
Same constructor kwargs, forward signature, state-dict keys.  The default transformer_type "x-transformers"
(dit.py:26,86-105) delegates to a third-party package that is not part of the reference tree and is not built here.
"""
import typing as tp

import torch
from torch import nn

from ... import functional as KF
from .blocks import FourierFeatures
from .transformer import ContinuousTransformer

BF16, F32 = torch.bfloat16, torch.float32


class _Mlp(nn.Sequential):
    """Linear - SiLU - Linear (dit.py:39-43, 49-53, 60-64, 68-72) run as GEMM + SiLU kernels."""

    def forward(self, x):
        l0, l2 = self[0], self[2]
        h = KF.linear(x, l0.weight, l0.bias, out_dtype=F32)
        return KF.linear(KF.silu(h), l2.weight, l2.bias, out_dtype=F32)


class DiffusionTransformer(nn.Module):
    def __init__(self, io_channels=32, patch_size=1, embed_dim=768, cond_token_dim=0, project_cond_tokens=True,
                 global_cond_dim=0, project_global_cond=True, input_concat_dim=0, prepend_cond_dim=0, depth=12,
                 num_heads=8, transformer_type: tp.Literal["x-transformers", "continuous_transformer"] = "x-transformers",
                 global_cond_type: tp.Literal["prepend", "adaLN"] = "prepend", **kwargs):
        super().__init__()
        self.cond_token_dim = cond_token_dim
        timestep_features_dim = 256
        self.timestep_features = FourierFeatures(1, timestep_features_dim)
        self.to_timestep_embed = _Mlp(nn.Linear(timestep_features_dim, embed_dim, bias=True), nn.SiLU(),
                                      nn.Linear(embed_dim, embed_dim, bias=True))
        if cond_token_dim > 0:
            cond_embed_dim = cond_token_dim if not project_cond_tokens else embed_dim
            self.to_cond_embed = _Mlp(nn.Linear(cond_token_dim, cond_embed_dim, bias=False), nn.SiLU(),
                                      nn.Linear(cond_embed_dim, cond_embed_dim, bias=False))
        else:
            cond_embed_dim = 0
        if global_cond_dim > 0:
            global_embed_dim = global_cond_dim if not project_global_cond else embed_dim
            self.to_global_embed = _Mlp(nn.Linear(global_cond_dim, global_embed_dim, bias=False), nn.SiLU(),
                                        nn.Linear(global_embed_dim, global_embed_dim, bias=False))
        if prepend_cond_dim > 0:
            self.to_prepend_embed = _Mlp(nn.Linear(prepend_cond_dim, embed_dim, bias=False), nn.SiLU(),
                                         nn.Linear(embed_dim, embed_dim, bias=False))
        self.input_concat_dim = input_concat_dim
        dim_in = io_channels + self.input_concat_dim
        self.patch_size = patch_size
        self.transformer_type = transformer_type
        self.global_cond_type = global_cond_type
        if self.transformer_type == "continuous_transformer":
            global_dim = embed_dim if self.global_cond_type == "adaLN" else None
            self.transformer = ContinuousTransformer(
                dim=embed_dim, depth=depth, dim_heads=embed_dim // num_heads, dim_in=dim_in * patch_size,
                dim_out=io_channels * patch_size, cross_attend=cond_token_dim > 0, cond_token_dim=cond_embed_dim,
                global_cond_dim=global_dim, **kwargs)
        elif self.transformer_type == "x-transformers":
            raise NotImplementedError(
                "transformer_type='x-transformers' delegates to the third-party x-transformers package "
                "(dit.py:86-105), which is outside the reference tree; use 'continuous_transformer'")
        else:
            raise ValueError(f"Unknown transformer type: {self.transformer_type}")
        self.preprocess_conv = nn.Conv1d(dim_in, dim_in, 1, bias=False)
        nn.init.zeros_(self.preprocess_conv.weight)
        self.postprocess_conv = nn.Conv1d(io_channels, io_channels, 1, bias=False)
        nn.init.zeros_(self.postprocess_conv.weight)

    def _forward(self, x, t, mask=None, cross_attn_cond=None, cross_attn_cond_mask=None, input_concat_cond=None,
                 global_embed=None, prepend_cond=None, prepend_cond_mask=None, return_info=False, kalle_ctx_kv=None,
                 kalle_cond_ready=False, kalle_global_ready=False, **kwargs):
        """kalle_cond_ready / kalle_ctx_kv: the conditioning was projected once for the whole sampling loop
        (precompute_conditioning): cross_attn_cond / global_embed are already the outputs of to_cond_embed / to_global_embed and
        kalle_ctx_kv holds the k | v projections of all layers"""
        if not x.is_cuda:
            raise RuntimeError("kalle_audio_amd modules run on an MI355X GPU only (no CPU fallback)")
        in_dtype = x.dtype
        if cross_attn_cond is not None and not kalle_cond_ready:
            cross_attn_cond = self.to_cond_embed(cross_attn_cond)
        if global_embed is not None and not kalle_global_ready:
            global_embed = self.to_global_embed(global_embed)
        if kalle_ctx_kv is not None:
            kwargs = dict(kwargs, kalle_ctx_kv=kalle_ctx_kv)
        prepend_inputs, prepend_mask, prepend_length = None, None, 0
        if prepend_cond is not None:
            prepend_inputs = self.to_prepend_embed(prepend_cond)
            if prepend_cond_mask is not None:
                prepend_mask = prepend_cond_mask
        if input_concat_cond is not None:
            if input_concat_cond.shape[2] != x.shape[2]:
                input_concat_cond = torch.nn.functional.interpolate(input_concat_cond, (x.shape[2],), mode='nearest')
            x = torch.cat([x, input_concat_cond], dim=1)  # host-side glue, as the reference (dit.py:167-173)
        timestep_embed = self.to_timestep_embed(self.timestep_features(t[:, None]))
        if global_embed is not None:
            global_embed = global_embed + timestep_embed  # [B, D] glue add (dit.py:179-180)
        else:
            global_embed = timestep_embed
        if self.global_cond_type == "prepend":
            if prepend_inputs is None:
                prepend_inputs = global_embed.unsqueeze(1)
                # the reference builds an all-true mask here (dit.py:189); all-true == no mask, so none is built
                # unless the caller supplied one
                if mask is not None:
                    prepend_mask = torch.ones((x.shape[0], 1), device=x.device, dtype=torch.bool)
            else:
                prepend_inputs = KF.SpliceFn.apply(prepend_inputs, global_embed.unsqueeze(1))
                ones = torch.ones((x.shape[0], 1), device=x.device, dtype=torch.bool)
                if prepend_mask is not None:
                    prepend_mask = torch.cat([prepend_mask, ones], dim=1)
            prepend_length = prepend_inputs.shape[1]
        # (b c t) -> (b t c); preprocess_conv(x) + x as a GEMM with the residual fused (dit.py:197-199)
        xt = KF.transpose(x, out_dtype=F32)
        wpre = self.preprocess_conv.weight
        xt = KF.linear(xt, wpre.view(wpre.shape[0], wpre.shape[1]), residual=xt, out_dtype=F32)
        extra_args = {}
        if self.global_cond_type == "adaLN":
            extra_args["global_cond"] = global_embed
        if self.patch_size > 1:
            b, tt, c = xt.shape
            xt = xt.view(b, tt // self.patch_size, self.patch_size, c).transpose(2, 3).reshape(
                b, tt // self.patch_size, c * self.patch_size)
        output = self.transformer(xt, prepend_embeds=prepend_inputs, context=cross_attn_cond,
                                  context_mask=cross_attn_cond_mask, mask=mask, prepend_mask=prepend_mask,
                                  return_info=return_info, **extra_args, **kwargs)
        if return_info:
            output, info = output
        if self.patch_size > 1:
            output = KF.transpose(output, out_dtype=F32, skip=prepend_length)
            b, cp, tt = output.shape
            output = output.view(b, cp // self.patch_size, self.patch_size, tt).transpose(2, 3).reshape(
                b, cp // self.patch_size, tt * self.patch_size)
            wpost = self.postprocess_conv.weight
            ot = KF.transpose(output, out_dtype=F32)
            ot = KF.linear(ot, wpost.view(wpost.shape[0], wpost.shape[1]), residual=ot, out_dtype=F32)
            output = KF.transpose(ot, out_dtype=F32)
        else:
            # postprocess_conv(out) + out in (b t c), then back to (b c t) dropping the prepended tokens (219-224)
            wpost = self.postprocess_conv.weight
            ot = KF.linear(output, wpost.view(wpost.shape[0], wpost.shape[1]), residual=output, out_dtype=F32)
            output = KF.transpose(ot, out_dtype=F32, skip=prepend_length)
        if in_dtype in (torch.float16,):
            output = output.to(in_dtype)
        if return_info:
            return output, info
        return output

    # ---- helpers of forward(): classifier-free guidance as cond | uncond halves of one doubled batch ------------------------
    @staticmethod
    def _twice(v):
        return None if v is None else torch.cat((v, v), dim=0)

    @staticmethod
    def _dropped(cond, prob):
        """per-sample conditioning dropout (dit.py:263-272): a Bernoulli(prob) draw per clip zeroes that clip's conditioning;
        host RNG glue, drawn exactly as the reference draws it (shape (B, 1, 1) on the tensor's device)"""
        if cond is None:
            return None
        drop = torch.bernoulli(torch.full((cond.shape[0], 1, 1), prob, device=cond.device)).to(torch.bool)
        return torch.where(drop, torch.zeros_like(cond), cond)

    @staticmethod
    def _with_unconditional(cond, negative=None, negative_mask=None):
        """[cond | what the unconditional half sees]: zeros, or the negative prompt with its masked-out tokens zeroed
        (dit.py:288-307)"""
        if cond is None:
            return None
        other = torch.zeros_like(cond)
        if negative is not None:
            other = negative if negative_mask is None else torch.where(negative_mask.to(torch.bool).unsqueeze(2), negative, other)
        return torch.cat((cond, other), dim=0)

    @staticmethod
    def _guided(both, scale, phi):
        """uncond + scale * (cond - uncond), optionally rescaled towards the conditional output's per-position std over
        channels (dit.py:345-360)"""
        cond_out, uncond_out = both.chunk(2, dim=0)
        guided = uncond_out + (cond_out - uncond_out) * scale
        if phi == 0.0:
            return guided
        ratio = cond_out.std(dim=1, keepdim=True) / guided.std(dim=1, keepdim=True)
        return phi * (guided * ratio) + (1 - phi) * guided

    @torch.no_grad()
    def precompute_conditioning(self, cross_attn_cond=None, negative_cross_attn_cond=None, negative_cross_attn_mask=None,
                                global_embed=None, prepend_cond=None, cfg_scale=1.0):
        """The samplers call the model with the SAME conditioning at every step (inference/sampling.py:24-86: `model(x, t,
        **extra_args)`), so everything that depends on it alone is constant over the loop: the cond | uncond halves of the CFG
        batch, to_cond_embed, to_global_embed and the cross-attention k | v projections of all layers.  Returns keyword
        arguments for forward() that carry those results (or {} where the fast path does not apply: trainable weights, a
        prepended conditioning, qk-norm, mixed layer shapes) - generate_diffusion_cond computes them once per call."""
        if cross_attn_cond is None or prepend_cond is not None or not cross_attn_cond.is_cuda:
            return {}
        if any(p.requires_grad for p in self.parameters()):
            return {}
        tr = self.transformer
        layers = list(tr.layers)
        if len(layers) < 2 or not all(l.cross_attend and l.cross_attn.qk_norm == "none" for l in layers):
            return {}
        if any(l.cross_attn.to_kv.weight.shape != layers[0].cross_attn.to_kv.weight.shape for l in layers):
            return {}
        guided = cfg_scale != 1.0
        cond = self._with_unconditional(cross_attn_cond, negative_cross_attn_cond, negative_cross_attn_mask) if guided \
            else cross_attn_cond
        emb = KF._to_bf16(self.to_cond_embed(cond).contiguous())
        if emb.shape[-1] != layers[0].cross_attn.to_kv.weight.shape[1]:
            return {}
        kv = KF.D.ops.gemm(emb.view(-1, emb.shape[-1]), torch.cat([KF.D.bf16_of(l.cross_attn.to_kv.weight) for l in layers], dim=0))
        out = {"kalle_ctx_embed": emb, "kalle_ctx_kv": kv}
        if global_embed is not None:
            g = self._twice(global_embed) if guided else global_embed
            out["kalle_global"] = self.to_global_embed(g)
        return out

    def forward(self, x, t, cross_attn_cond=None, cross_attn_cond_mask=None, negative_cross_attn_cond=None,
                negative_cross_attn_mask=None, input_concat_cond=None, global_embed=None,
                negative_global_embed=None, prepend_cond=None, prepend_cond_mask=None, cfg_scale=1.0,
                cfg_dropout_prob=0.0, causal=False, scale_phi=0.0, mask=None, return_info=False, kalle_ctx_embed=None,
                kalle_ctx_kv=None, kalle_global=None, **kwargs):
        assert not causal, "Causal mode is not supported for DiffusionTransformer"
        if kalle_ctx_embed is not None:
            # conditioning projected once for the whole sampling loop (precompute_conditioning): same arithmetic, hoisted
            guided = cfg_scale != 1.0
            two = self._twice if guided else (lambda v: v)
            glob = kalle_global if kalle_global is not None else two(global_embed)
            result = self._forward(two(x), two(t), cross_attn_cond=kalle_ctx_embed, cross_attn_cond_mask=None, mask=two(mask),
                                   input_concat_cond=two(input_concat_cond), global_embed=glob, prepend_cond=None,
                                   prepend_cond_mask=None, return_info=return_info, kalle_ctx_kv=kalle_ctx_kv,
                                   kalle_cond_ready=True, kalle_global_ready=kalle_global is not None, **kwargs)
            if not guided:
                return result
            if return_info:
                return self._guided(result[0], cfg_scale, scale_phi), result[1]
            return self._guided(result, cfg_scale, scale_phi)
        cross_attn_cond_mask = None             # the reference disables conditioning masks (dit.py:254-257)
        if prepend_cond_mask is not None:
            prepend_cond_mask = prepend_cond_mask.bool()
        if cfg_dropout_prob > 0.0:              # training: cross-attention draw first, prepend draw second (RNG order of :263-272)
            cross_attn_cond = self._dropped(cross_attn_cond, cfg_dropout_prob)
            prepend_cond = self._dropped(prepend_cond, cfg_dropout_prob)
        guided = cfg_scale != 1.0 and (cross_attn_cond is not None or prepend_cond is not None)
        if not guided:
            return self._forward(x, t, cross_attn_cond=cross_attn_cond, cross_attn_cond_mask=cross_attn_cond_mask,
                                 input_concat_cond=input_concat_cond, global_embed=global_embed,
                                 prepend_cond=prepend_cond, prepend_cond_mask=prepend_cond_mask, mask=mask,
                                 return_info=return_info, **kwargs)
        # one pass over [conditional | unconditional] (dit.py:275-364); the global embedding is shared by both halves
        result = self._forward(self._twice(x), self._twice(t),
                               cross_attn_cond=self._with_unconditional(cross_attn_cond, negative_cross_attn_cond,
                                                                        negative_cross_attn_mask),
                               cross_attn_cond_mask=None, mask=self._twice(mask),
                               input_concat_cond=self._twice(input_concat_cond), global_embed=self._twice(global_embed),
                               prepend_cond=self._with_unconditional(prepend_cond),
                               prepend_cond_mask=self._twice(prepend_cond_mask) if prepend_cond is not None else None,
                               return_info=return_info, **kwargs)
        if return_info:
            both, info = result
            return self._guided(both, cfg_scale, scale_phi), info
        return self._guided(result, cfg_scale, scale_phi)
