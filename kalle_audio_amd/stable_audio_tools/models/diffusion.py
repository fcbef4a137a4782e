"""Drop-in for the DiT wrappers of stable_audio_tools/models/diffusion.py: ConditionedDiffusionModel (base),
ConditionedDiffusionModelWrapper (99-214), DiTWrapper (495-546), create_diffusion_cond_from_config (618-701).
The UNet / DAU wrappers of that file belong to other model families and are out of scope."""
import typing as tp

import numpy as np
import torch
from torch import nn

from .dit import DiffusionTransformer


class ConditionedDiffusionModel(nn.Module):
    def __init__(self, *args, supports_cross_attention: bool = False, supports_input_concat: bool = False,
                 supports_global_cond: bool = False, supports_prepend_cond: bool = False, **kwargs):
        super().__init__(*args, **kwargs)
        self.supports_cross_attention = supports_cross_attention
        self.supports_input_concat = supports_input_concat
        self.supports_global_cond = supports_global_cond
        self.supports_prepend_cond = supports_prepend_cond


class ConditionedDiffusionModelWrapper(nn.Module):
    """models/diffusion.py:99-214: routes conditioner outputs to the model's keyword arguments."""

    def __init__(self, model, conditioner, io_channels, sample_rate, min_input_length: int,
                 diffusion_objective: tp.Literal["v", "rectified_flow"] = "v", pretransform=None,
                 cross_attn_cond_ids: tp.List[str] = [], global_cond_ids: tp.List[str] = [],
                 input_concat_ids: tp.List[str] = [], prepend_cond_ids: tp.List[str] = []):
        super().__init__()
        self.model = model
        self.conditioner = conditioner
        self.io_channels = io_channels
        self.sample_rate = sample_rate
        self.diffusion_objective = diffusion_objective
        self.pretransform = pretransform
        self.cross_attn_cond_ids = cross_attn_cond_ids
        self.global_cond_ids = global_cond_ids
        self.input_concat_ids = input_concat_ids
        self.prepend_cond_ids = prepend_cond_ids
        self.min_input_length = min_input_length

    def get_conditioning_inputs(self, conditioning_tensors: tp.Dict[str, tp.Any], negative=False):
        cross_attention_input = cross_attention_masks = global_cond = input_concat_cond = None
        prepend_cond = prepend_cond_mask = None
        if len(self.cross_attn_cond_ids) > 0:
            ins, masks = [], []
            for key in self.cross_attn_cond_ids:
                c, m = conditioning_tensors[key]
                if len(c.shape) == 2:
                    c, m = c.unsqueeze(1), m.unsqueeze(1)
                ins.append(c)
                masks.append(m)
            cross_attention_input = torch.cat(ins, dim=1)
            cross_attention_masks = torch.cat(masks, dim=1)
        if len(self.global_cond_ids) > 0:
            global_cond = torch.cat([conditioning_tensors[key][0] for key in self.global_cond_ids], dim=-1)
            if len(global_cond.shape) == 3:
                global_cond = global_cond.squeeze(1)
        if len(self.input_concat_ids) > 0:
            input_concat_cond = torch.cat([conditioning_tensors[key][0] for key in self.input_concat_ids], dim=1)
        if len(self.prepend_cond_ids) > 0:
            pcs, pms = [], []
            for key in self.prepend_cond_ids:
                c, m = conditioning_tensors[key]
                pcs.append(c)
                pms.append(m)
            prepend_cond = torch.cat(pcs, dim=1)
            prepend_cond_mask = torch.cat(pms, dim=1)
        if negative:
            return {"negative_cross_attn_cond": cross_attention_input,
                    "negative_cross_attn_mask": cross_attention_masks,
                    "negative_global_cond": global_cond, "negative_input_concat_cond": input_concat_cond}
        return {"cross_attn_cond": cross_attention_input, "cross_attn_mask": cross_attention_masks,
                "global_cond": global_cond, "input_concat_cond": input_concat_cond, "prepend_cond": prepend_cond,
                "prepend_cond_mask": prepend_cond_mask}

    def forward(self, x: torch.Tensor, t: torch.Tensor, cond: tp.Dict[str, tp.Any], **kwargs):
        return self.model(x, t, **self.get_conditioning_inputs(cond), **kwargs)

    def generate(self, *args, **kwargs):
        from ..inference.generation import generate_diffusion_cond
        return generate_diffusion_cond(self, *args, **kwargs)


class DiTWrapper(ConditionedDiffusionModel):
    """models/diffusion.py:495-546 (halves every parameter at construction, 505-507)."""

    def __init__(self, *args, **kwargs):
        super().__init__(supports_cross_attention=True, supports_global_cond=False, supports_input_concat=False)
        self.model = DiffusionTransformer(*args, **kwargs)
        with torch.no_grad():
            for param in self.model.parameters():
                param *= 0.5

    def forward(self, x, t, cross_attn_cond=None, cross_attn_mask=None, negative_cross_attn_cond=None,
                negative_cross_attn_mask=None, input_concat_cond=None, negative_input_concat_cond=None,
                global_cond=None, negative_global_cond=None, prepend_cond=None, prepend_cond_mask=None,
                cfg_scale=1.0, cfg_dropout_prob: float = 0.0, batch_cfg: bool = True, rescale_cfg: bool = False,
                scale_phi: float = 0.0, **kwargs):
        assert batch_cfg, "batch_cfg must be True for DiTWrapper"
        return self.model(x, t, cross_attn_cond=cross_attn_cond, cross_attn_cond_mask=cross_attn_mask,
                          negative_cross_attn_cond=negative_cross_attn_cond,
                          negative_cross_attn_mask=negative_cross_attn_mask, input_concat_cond=input_concat_cond,
                          prepend_cond=prepend_cond, prepend_cond_mask=prepend_cond_mask, cfg_scale=cfg_scale,
                          cfg_dropout_prob=cfg_dropout_prob, scale_phi=scale_phi, global_embed=global_cond, **kwargs)


class _TensorConditioner(nn.Module):
    """Stand-in for MultiConditioner (conditioners.py:469-510, out of scope: frozen text encoders fetched by name):
    passes through pre-computed conditioning tensors given in the metadata dicts as {id: (tensor, mask)}."""

    def forward(self, batch_metadata, device):
        if isinstance(batch_metadata, dict):
            return batch_metadata
        keys = batch_metadata[0].keys()
        out = {}
        for k in keys:
            ts = [md[k][0] for md in batch_metadata]
            ms = [md[k][1] for md in batch_metadata]
            out[k] = (torch.cat(ts, 0).to(device), torch.cat(ms, 0).to(device))
        return out


def create_diffusion_cond_from_config(config: tp.Dict[str, tp.Any]):
    """models/diffusion.py:618-701 for diffusion type 'dit'."""
    from .factory import create_pretransform_from_config
    model_config = config["model"]
    model_type = config["model_type"]
    diffusion_config = model_config.get('diffusion', None)
    assert diffusion_config is not None, "Must specify diffusion config"
    diffusion_model_type = diffusion_config.get('type', None)
    assert diffusion_model_type is not None, "Must specify diffusion model type"
    diffusion_model_config = diffusion_config.get('config', None)
    assert diffusion_model_config is not None, "Must specify diffusion model config"
    if diffusion_model_type == 'dit':
        diffusion_model = DiTWrapper(**diffusion_model_config)
    else:
        raise NotImplementedError(f"diffusion type {diffusion_model_type!r}: only 'dit' is on the accelerated path")
    io_channels = model_config.get('io_channels', None)
    assert io_channels is not None, "Must specify io_channels in model config"
    sample_rate = config.get('sample_rate', None)
    assert sample_rate is not None, "Must specify sample_rate in config"
    diffusion_objective = diffusion_config.get('diffusion_objective', 'v')
    conditioning_config = model_config.get('conditioning', None)
    conditioner = _TensorConditioner() if conditioning_config is not None else None
    pretransform = model_config.get("pretransform", None)
    if pretransform is not None:
        pretransform = create_pretransform_from_config(pretransform, sample_rate)
        min_input_length = pretransform.downsampling_ratio
    else:
        min_input_length = 1
    min_input_length *= diffusion_model.model.patch_size
    if model_type not in ("diffusion_cond", "diffusion_cond_inpaint"):
        raise NotImplementedError(f"model_type {model_type!r}")
    return ConditionedDiffusionModelWrapper(
        diffusion_model, conditioner, min_input_length=min_input_length, sample_rate=sample_rate,
        cross_attn_cond_ids=diffusion_config.get('cross_attention_cond_ids', []),
        global_cond_ids=diffusion_config.get('global_cond_ids', []),
        input_concat_ids=diffusion_config.get('input_concat_ids', []),
        prepend_cond_ids=diffusion_config.get('prepend_cond_ids', []), pretransform=pretransform,
        io_channels=io_channels, diffusion_objective=diffusion_objective)
