"""Drop-in for the continuous bottlenecks of stable_audio_tools/models/bottleneck.py: Bottleneck (10-21),
TanhBottleneck (33-49), vae_sample (51-62, as modified in this reference) and VAEBottleneck (85-107, which this
reference turned into a pass-through: the encoder's mean||scale channels are returned untouched and sampling happens
in the dataset / inference scripts).  Quantised bottlenecks (RVQ/FSQ/DAC) are discrete-codec code, out of scope."""
import torch
from torch import nn


class Bottleneck(nn.Module):
    def __init__(self, is_discrete: bool = False):
        super().__init__()
        self.is_discrete = is_discrete

    def encode(self, x, return_info=False, **kwargs):
        raise NotImplementedError

    def decode(self, x):
        raise NotImplementedError


class TanhBottleneck(Bottleneck):
    def __init__(self):
        super().__init__(is_discrete=False)
        self.tanh = nn.Tanh()

    def encode(self, x, return_info=False):
        x = torch.tanh(x)
        return (x, {}) if return_info else x

    def decode(self, x):
        return x


def vae_sample(mean, scale):
    """bottleneck.py:51-62 as modified in this reference: latents = randn * scale + mean (raw scale, not the
    softplus stdev, which only enters the KL term).  The reference's debug print (58) is not reproduced."""
    stdev = nn.functional.softplus(scale) + 1e-4
    var = stdev * stdev
    logvar = torch.log(var)
    latents = torch.randn_like(mean) * scale + mean
    kl = (mean * mean + var - logvar - 1).sum(1).mean()
    return latents, kl


class VAEBottleneck(Bottleneck):
    """bottleneck.py:85-107: identity in both directions."""

    def __init__(self):
        super().__init__(is_discrete=False)

    def encode(self, x, return_info=False, **kwargs):
        return (x, {}) if return_info else x

    def decode(self, x):
        return x
