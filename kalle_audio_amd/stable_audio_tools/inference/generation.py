"""generate_diffusion_cond (stable_audio_tools/inference/generation.py:90-250), reduced to the DiT + v/rectified-flow
samplers this build carries: seed -> noise (138-142 convention: torch.manual_seed(seed); randn on the device) ->
sampler with batched CFG -> pretransform.decode."""
import numpy as np
import torch

from .sampling import sample, sample_discrete_euler


def generate_diffusion_cond(model, steps: int = 250, cfg_scale=6, conditioning=None, conditioning_tensors=None,
                            negative_conditioning_tensors=None, batch_size: int = 1, sample_size: int = 2097152,
                            seed: int = -1, device: str = "cuda", return_latents=False, eta=0.0, **sampler_kwargs):
    audio_sample_size = sample_size
    if model.pretransform is not None:
        sample_size = sample_size // model.pretransform.downsampling_ratio
    seed = seed if seed != -1 else np.random.randint(0, 2 ** 32 - 1, dtype=np.uint32)
    torch.manual_seed(int(seed))
    noise = torch.randn([batch_size, model.io_channels, sample_size], device=device)
    if conditioning_tensors is None:
        conditioning_tensors = model.conditioner(conditioning, device)
    cond_inputs = model.get_conditioning_inputs(conditioning_tensors)
    neg = model.get_conditioning_inputs(negative_conditioning_tensors, negative=True) \
        if negative_conditioning_tensors is not None else {}
    if model.diffusion_objective == "v":
        sampled = sample(model.model, noise, steps, eta, **cond_inputs, **neg, cfg_scale=cfg_scale, batch_cfg=True,
                         **sampler_kwargs)
    else:
        sampled = sample_discrete_euler(model.model, noise, steps, **cond_inputs, **neg, cfg_scale=cfg_scale,
                                        batch_cfg=True, **sampler_kwargs)
    if model.pretransform is not None and not return_latents:
        sampled = model.pretransform.decode(sampled)
    return sampled
