"""Drop-in for stable_audio_tools/models/pretransforms.py:5-90 (Pretransform base, AutoencoderPretransform)."""
import torch
from torch import nn


class Pretransform(nn.Module):
    def __init__(self, enable_grad, io_channels, is_discrete):
        super().__init__()
        self.is_discrete = is_discrete
        self.io_channels = io_channels
        self.encoded_channels = None
        self.downsampling_ratio = None
        self.enable_grad = enable_grad

    def encode(self, x):
        raise NotImplementedError

    def decode(self, z):
        raise NotImplementedError


class AutoencoderPretransform(Pretransform):
    """pretransforms.py:28-90: encode -> latents / scale ; decode(z * scale).  `model_half` runs the conv stack with
    bf16 activations (the reference uses fp16) and returns fp32."""

    def __init__(self, model, scale=1.0, model_half=False, iterate_batch=False, chunked=False):
        super().__init__(enable_grad=False, io_channels=model.io_channels,
                         is_discrete=model.bottleneck is not None and model.bottleneck.is_discrete)
        self.model = model
        self.model.requires_grad_(False).eval()
        self.scale = scale
        self.downsampling_ratio = model.downsampling_ratio
        self.io_channels = model.io_channels
        self.sample_rate = model.sample_rate
        self.model_half = model_half
        self.iterate_batch = iterate_batch
        self.encoded_channels = model.latent_dim
        self.chunked = chunked
        self.num_quantizers = None
        self.codebook_size = None

    # No torch.no_grad() here, as in the reference (pretransforms.py:50-75): the caller decides - the training wrapper runs
    # encode under torch.set_grad_enabled(self.enable_grad) (training/diffusion.py:343-346).  A frozen VAE with an input that
    # needs no gradient takes the fused inference kernels; otherwise the conv stacks run as autograd units (conv_train.py).
    def encode(self, x, **kwargs):
        if self.model_half:
            x = x.to(torch.bfloat16)
        encoded = self.model.encode_audio(x, chunked=self.chunked, iterate_batch=self.iterate_batch, **kwargs)
        return encoded.float() / self.scale

    def decode(self, z, **kwargs):
        z = z * self.scale
        if self.model_half:
            z = z.to(torch.bfloat16)
        decoded = self.model.decode_audio(z, chunked=self.chunked, iterate_batch=self.iterate_batch, **kwargs)
        return decoded.float()

    def load_state_dict(self, state_dict, strict=True):
        self.model.load_state_dict(state_dict, strict=strict)
