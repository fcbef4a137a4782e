"""Drop-in for the Oobleck VAE of stable_audio_tools/models/autoencoders.py: WNConv1d / WNConvTranspose1d (the
dac.nn.layers wrappers, line 9), get_activation (24-37), ResidualUnit (39-62), EncoderBlock (64-81), DecoderBlock
(83-114), OobleckEncoder (116-147), OobleckDecoder (150-191), AudioAutoencoder (230-560, encode/decode and the
chunked overlap-and-paste variants), create_{encoder,decoder,autoencoder}_from_config (611-731).
State-dict keys match torch's old-style weight_norm (`weight_g`, `weight_v`, `bias`) and SnakeBeta (`alpha`, `beta`),
i.e. Stable-Audio-Open checkpoints load unchanged.  Inference (the VAE is frozen in every reference script,
factory.py:77-80) runs the fused kernels of kalle_audio_amd/csrc/conv1d.hip; when gradients are wanted (`enable_grad`) the
same modules run as autograd units of kalle_audio_amd/conv_train.py (csrc/conv1d_bwd.hip).  DAC / SEANet / local-attention / diffusion
autoencoders of the same file are other model families and are not built.
"""
import math
from typing import Any, Dict, Literal

import torch
from torch import nn

from ... import conv_ops
from .blocks import SnakeBeta
from .bottleneck import Bottleneck


def _prep(x):
    if not x.is_cuda:
        raise RuntimeError("kalle_audio_amd VAE modules run on an MI355X GPU only (no CPU fallback)")
    return x if x.dtype in (torch.float32, torch.bfloat16) else x.float()


def _wants_grad(module, x):
    """the training path (autograd nodes of kalle_audio_amd/conv_train.py) instead of the fused inference path: gradients are
    being recorded and either the input or one of the module's parameters wants one"""
    return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in module.parameters()))


def _act_args(act):
    """(code, alpha, beta, logscale) for the conv kernels' fused input activation."""
    if act is None or isinstance(act, nn.Identity):
        return 0, None, None, True
    if isinstance(act, SnakeBeta):
        return 1, act.alpha.detach().float(), act.beta.detach().float(), act.alpha_logscale
    if isinstance(act, nn.ELU):
        return 2, None, None, True
    raise NotImplementedError(f"activation {type(act).__name__}")


def _post_act(act):
    """descriptor for the conv kernels' OUTPUT activation: the consumer's input activation applied once at the
    producer's store (instead of once per consumer tile) whenever the raw value has no other reader"""
    code, a, b, ls = _act_args(act)
    return None if code == 0 else (code, a, b, ls, 0.0)


class _WNBase(nn.Module):
    transposed = False

    def _packed(self):
        key = (self.weight_g._version, self.weight_v._version, self.weight_v.device)
        c = getattr(self, "_kalle_packed", None)
        if c is None or c[0] != key:
            c = (key, conv_ops.weight_norm_fold(self.weight_v, self.weight_g, transposed=self.transposed))
            self._kalle_packed = c
        return c[1]

    def _bias(self):
        return self.bias.detach().float() if self.bias is not None else None


class WNConv1d(_WNBase):
    """weight_norm(nn.Conv1d) (dac.nn.layers.WNConv1d): parameters weight_g [Cout,1,1], weight_v [Cout,Cin,K], bias."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, bias=True):
        super().__init__()
        ref = nn.Conv1d(in_channels, out_channels, kernel_size, stride=stride,
                        padding=0 if isinstance(padding, str) else padding, dilation=dilation, bias=bias)
        self.pad_right = None
        if padding == 'same':
            # PyTorch 'same' (stride 1): total = dilation*(k-1) zeros, left = total // 2, the odd one goes to the right
            assert stride == 1, "padding='same' is not supported for strided convolutions"
            total = dilation * (kernel_size - 1)
            padding = total // 2
            if total % 2:
                self.pad_right = total - padding
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.weight_v = nn.Parameter(ref.weight.detach().clone())
        self.weight_g = nn.Parameter(ref.weight.detach().flatten(1).norm(dim=1).view(-1, 1, 1).clone())
        self.bias = nn.Parameter(ref.bias.detach().clone()) if bias else None

    def forward(self, x, act=None, residual=None, post=0, post_act=None, want_raw=False):
        """want_raw: also return the value before `post_act` -> (activated, raw)"""
        x = _prep(x)
        if _wants_grad(self, x) or (act is not None and _wants_grad(act, x)):
            assert post_act is None and not want_raw, "fused output activations belong to the inference path"
            from ... import conv_train
            return conv_train.act_conv(x, self, act, residual=residual, tanh=bool(post & 1))
        code, a, b, ls = _act_args(act)
        return conv_ops.conv1d(x, self._packed(), self._bias(), Cout=self.out_channels, K=self.kernel_size,
                               stride=self.stride, padding=self.padding, dilation=self.dilation, act=code, alpha=a,
                               beta=b, logscale=ls, residual=residual, post=post, post_act=_post_act(post_act),
                               want_raw=want_raw, pad_right=self.pad_right)


class WNConvTranspose1d(_WNBase):
    """weight_norm(nn.ConvTranspose1d): weight_g [Cin,1,1], weight_v [Cin,Cout,K], bias [Cout]."""
    transposed = True

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, bias=True):
        super().__init__()
        ref = nn.ConvTranspose1d(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=bias)
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.stride, self.padding = stride, padding
        self.weight_v = nn.Parameter(ref.weight.detach().clone())
        self.weight_g = nn.Parameter(ref.weight.detach().flatten(1).norm(dim=1).view(-1, 1, 1).clone())
        self.bias = nn.Parameter(ref.bias.detach().clone()) if bias else None

    def forward(self, x, act=None, post_act=None, want_raw=False):
        x = _prep(x)
        if _wants_grad(self, x) or (act is not None and _wants_grad(act, x)):
            assert post_act is None and not want_raw, "fused output activations belong to the inference path"
            from ... import conv_train
            return conv_train.act_conv(x, self, act)
        code, a, b, ls = _act_args(act)
        return conv_ops.conv_transpose1d(x, self._packed(), self._bias(), Cout=self.out_channels,
                                         K=self.kernel_size, stride=self.stride, padding=self.padding, act=code,
                                         alpha=a, beta=b, logscale=ls, post_act=_post_act(post_act), want_raw=want_raw)


class Activation1d(nn.Module):
    """alias_free_torch.Activation1d(act) as autoencoders.py:34-35 wraps it around every activation when
    `antialias_activation=True`: 2x kaiser-sinc up-sampling -> activation -> 2x low-pass down-sampling (12 taps each).  The
    package is third-party and absent from the reference tree: restated from its published algorithm (oracle.activation1d),
    PARITY UNPINNED - as for the mel-VAE's AMP blocks (kalle_audio_amd/flows.py), whose fused kernel this is.  It runs as its
    own pass (one read + one write), so the convs around it see plain inputs; inference only."""

    def __init__(self, activation):
        super().__init__()
        from ... import conv_ops
        self.act = activation
        self.upsample = nn.Module()                 # the package's state-dict names: upsample.filter, downsample.lowpass.filter
        self.upsample.register_buffer("filter", conv_ops.kaiser_sinc_filter12("cpu").view(1, 1, -1))
        self.downsample = nn.Module()
        self.downsample.lowpass = nn.Module()
        self.downsample.lowpass.register_buffer("filter", conv_ops.kaiser_sinc_filter12("cpu").view(1, 1, -1))

    def forward(self, x):
        from ... import conv_ops
        x = _prep(x)
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
            raise NotImplementedError("antialias_activation has no backward kernel (the reference freezes the VAE in every script)")
        f = self.upsample.filter.view(-1)
        if isinstance(self.act, SnakeBeta):
            return conv_ops.act1d(x, f, self.act.alpha.detach().float(), self.act.beta.detach().float(), self.act.alpha_logscale)
        if isinstance(self.act, nn.ELU):
            return conv_ops.act1d(x, f, None, None, True)
        raise NotImplementedError(f"Activation1d({type(self.act).__name__})")


def get_activation(activation: Literal["elu", "snake", "none"], antialias=False, channels=None) -> nn.Module:
    """autoencoders.py:24-37"""
    if activation == "elu":
        act = nn.ELU()
    elif activation == "snake":
        act = SnakeBeta(channels)
    elif activation == "none":
        act = nn.Identity()
    else:
        raise ValueError(f"Unknown activation {activation}")
    if antialias:
        act = Activation1d(act)
    return act


def _ac(conv, x, act, **kw):
    """conv(act(x)): an anti-aliased activation cannot ride in a conv's input staging - it runs first, as its own pass, and the
    conv sees a plain input; a plain activation is fused into the staging as usual"""
    if isinstance(act, Activation1d):
        return conv(act(x), **kw)
    return conv(x, act=act, **kw)


class ResidualUnit(nn.Module):
    """autoencoders.py:39-62: x + conv1(act(conv7_dilated(act(x)))).  The first activation is applied while the k=7 conv
    stages its input (x is also needed raw for the skip), the second by the k=7 conv's store (its only reader is the
    k=1 conv), the residual add by the k=1 conv's store.  `post_act`: the consumer's input activation, folded into the
    last store when the raw output has no other reader."""

    def __init__(self, in_channels, out_channels, dilation, use_snake=False, antialias_activation=False):
        super().__init__()
        self.dilation = dilation
        self.antialias = antialias_activation
        padding = (dilation * (7 - 1)) // 2
        self.layers = nn.Sequential(
            get_activation("snake" if use_snake else "elu", antialias=antialias_activation, channels=out_channels),
            WNConv1d(in_channels=in_channels, out_channels=out_channels, kernel_size=7, dilation=dilation,
                     padding=padding),
            get_activation("snake" if use_snake else "elu", antialias=antialias_activation, channels=out_channels),
            WNConv1d(in_channels=out_channels, out_channels=out_channels, kernel_size=1))

    def forward(self, x, post_act=None, x_act=None, dual=False):
        """x_act: layers[0](x) if the producer already stored it (then the k=7 conv stages its input without any activation
        work); dual: return (post_act(y), y) - the pair the next unit wants"""
        x = _prep(x)
        if self.antialias:              # plain composition (the producer-side activation fusion needs pointwise activations)
            assert post_act is None and x_act is None and not dual
            return _ac(self.layers[3], _ac(self.layers[1], x, self.layers[0]), self.layers[2], residual=x)
        if _wants_grad(self, x):        # training: two autograd units, raw tensors in between
            h = self.layers[1](x, act=self.layers[0])
            return self.layers[3](h, act=self.layers[2], residual=x)
        if x_act is not None:
            h = self.layers[1](x_act, post_act=self.layers[2])
        else:
            h = self.layers[1](x, act=self.layers[0], post_act=self.layers[2])
        return self.layers[3](h, residual=x, post_act=post_act, want_raw=dual)


class EncoderBlock(nn.Module):
    """autoencoders.py:64-81"""

    def __init__(self, in_channels, out_channels, stride, use_snake=False, antialias_activation=False):
        super().__init__()
        self.antialias = antialias_activation       # (the reference hands it to the block's own activation only, not to the units)
        self.layers = nn.Sequential(
            ResidualUnit(in_channels=in_channels, out_channels=in_channels, dilation=1, use_snake=use_snake),
            ResidualUnit(in_channels=in_channels, out_channels=in_channels, dilation=3, use_snake=use_snake),
            ResidualUnit(in_channels=in_channels, out_channels=in_channels, dilation=9, use_snake=use_snake),
            get_activation("snake" if use_snake else "elu", antialias=antialias_activation, channels=in_channels),
            WNConv1d(in_channels=in_channels, out_channels=out_channels, kernel_size=2 * stride, stride=stride,
                     padding=math.ceil(stride / 2)))

    def forward(self, x, post_act=None, x_act=None, dual=False):
        """x_act = layers[0].layers[0](x) if the producer stored it; dual: the strided conv returns (post_act(y), y)"""
        ru = self.layers
        if self.antialias:
            assert post_act is None and x_act is None and not dual
            return _ac(ru[4], ru[2](ru[1](ru[0](x))), ru[3])
        if _wants_grad(self, x):
            return ru[4](ru[2](ru[1](ru[0](x))), act=ru[3])
        xa, x = ru[0](x, post_act=ru[1].layers[0], x_act=x_act, dual=True)
        xa, x = ru[1](x, post_act=ru[2].layers[0], x_act=xa, dual=True)
        x = ru[2](x, post_act=self.layers[3], x_act=xa)
        return self.layers[4](x, post_act=post_act, want_raw=dual)


class DecoderBlock(nn.Module):
    """autoencoders.py:83-114"""

    def __init__(self, in_channels, out_channels, stride, use_snake=False, antialias_activation=False,
                 use_nearest_upsample=False):
        super().__init__()
        self.antialias = antialias_activation
        self.nearest = stride if use_nearest_upsample else 0
        if use_nearest_upsample:        # autoencoders.py:87-96; same Sequential nesting, so the state-dict keys match
            upsample_layer = nn.Sequential(
                nn.Upsample(scale_factor=stride, mode="nearest"),
                WNConv1d(in_channels=in_channels, out_channels=out_channels, kernel_size=2 * stride, stride=1, bias=False,
                         padding='same'))
        else:
            upsample_layer = WNConvTranspose1d(in_channels=in_channels, out_channels=out_channels,
                                               kernel_size=2 * stride + stride % 2, stride=stride,
                                               padding=math.ceil(stride / 2))
        self.layers = nn.Sequential(
            get_activation("snake" if use_snake else "elu", antialias=antialias_activation, channels=in_channels),
            upsample_layer,
            ResidualUnit(in_channels=out_channels, out_channels=out_channels, dilation=1, use_snake=use_snake),
            ResidualUnit(in_channels=out_channels, out_channels=out_channels, dilation=3, use_snake=use_snake),
            ResidualUnit(in_channels=out_channels, out_channels=out_channels, dilation=9, use_snake=use_snake))

    def forward(self, x, pre_activated=False, post_act=None):
        # every unit's output is needed raw (skip path of the next unit) and activated (its first conv): the producers store
        # both, so no k=7 conv spends VALU time re-activating its input tile once per output-channel tile
        ru = self.layers
        up = ru[1]
        if self.antialias:              # act -> up-sample -> units, one pass each (low-pass filtering does not commute with repetition)
            assert not pre_activated and post_act is None
            x = ru[0](x)
            if self.nearest:
                from ... import conv_train
                x, up = conv_train.UpsampleNearestFn.apply(_prep(x).float(), self.nearest), ru[1][1]
            return ru[4](ru[3](ru[2](up(x))))
        if self.nearest:
            # a pointwise activation commutes with sample repetition: repeat x, then the conv activates while it stages
            from ... import conv_train
            x, up = conv_train.UpsampleNearestFn.apply(_prep(x).float(), self.nearest), ru[1][1]
        if _wants_grad(self, x):
            assert not pre_activated
            return ru[4](ru[3](ru[2](up(x, act=ru[0]))))
        xa, x = up(x, act=None if pre_activated else ru[0], post_act=ru[2].layers[0], want_raw=True)
        xa, x = ru[2](x, post_act=ru[3].layers[0], x_act=xa, dual=True)
        xa, x = ru[3](x, post_act=ru[4].layers[0], x_act=xa, dual=True)
        return ru[4](x, post_act=post_act, x_act=xa)


class OobleckEncoder(nn.Module):
    """autoencoders.py:116-147"""

    def __init__(self, in_channels=2, channels=128, latent_dim=32, c_mults=[1, 2, 4, 8], strides=[2, 4, 8, 8],
                 use_snake=False, antialias_activation=False):
        super().__init__()
        c_mults = [1] + c_mults
        self.depth = len(c_mults)
        self.antialias = antialias_activation       # (autoencoders.py:136-137 does not hand it to the encoder blocks)
        layers = [WNConv1d(in_channels=in_channels, out_channels=c_mults[0] * channels, kernel_size=7, padding=3)]
        for i in range(self.depth - 1):
            layers += [EncoderBlock(in_channels=c_mults[i] * channels, out_channels=c_mults[i + 1] * channels,
                                    stride=strides[i], use_snake=use_snake)]
        layers += [get_activation("snake" if use_snake else "elu", antialias=antialias_activation,
                                  channels=c_mults[-1] * channels),
                   WNConv1d(in_channels=c_mults[-1] * channels, out_channels=latent_dim, kernel_size=3, padding=1)]
        self.layers = nn.Sequential(*layers)

    def forward(self, x):
        n = len(self.layers)
        if n == 3:
            return _ac(self.layers[2], self.layers[0](x), self.layers[1])
        if self.antialias:
            x = self.layers[0](x)
            for i in range(1, n - 2):
                x = self.layers[i](x)
            return _ac(self.layers[n - 1], x, self.layers[n - 2])
        if _wants_grad(self, x):
            x = self.layers[0](x)
            for i in range(1, n - 2):
                x = self.layers[i](x)
            return self.layers[n - 1](x, act=self.layers[n - 2])
        first_act = lambda i: self.layers[i].layers[0].layers[0]      # first activation of encoder block i
        xa, x = self.layers[0](x, post_act=first_act(1), want_raw=True)
        for i in range(1, n - 2):
            if i == n - 3:
                x = self.layers[i](x, post_act=self.layers[n - 2], x_act=xa)
            else:
                xa, x = self.layers[i](x, post_act=first_act(i + 1), x_act=xa, dual=True)
        return self.layers[n - 1](x)


class OobleckDecoder(nn.Module):
    """autoencoders.py:150-191"""

    def __init__(self, out_channels=2, channels=128, latent_dim=32, c_mults=[1, 2, 4, 8], strides=[2, 4, 8, 8],
                 use_snake=False, antialias_activation=False, use_nearest_upsample=False, final_tanh=True):
        super().__init__()
        c_mults = [1] + c_mults
        self.depth = len(c_mults)
        self.antialias = antialias_activation
        layers = [WNConv1d(in_channels=latent_dim, out_channels=c_mults[-1] * channels, kernel_size=7, padding=3)]
        for i in range(self.depth - 1, 0, -1):
            layers += [DecoderBlock(in_channels=c_mults[i] * channels, out_channels=c_mults[i - 1] * channels,
                                    stride=strides[i - 1], use_snake=use_snake,
                                    antialias_activation=antialias_activation,
                                    use_nearest_upsample=use_nearest_upsample)]
        layers += [get_activation("snake" if use_snake else "elu", antialias=antialias_activation,
                                  channels=c_mults[0] * channels),
                   WNConv1d(in_channels=c_mults[0] * channels, out_channels=out_channels, kernel_size=7, padding=3,
                            bias=False),
                   nn.Tanh() if final_tanh else nn.Identity()]
        self.layers = nn.Sequential(*layers)

    def forward(self, x):
        n = len(self.layers)
        if self.antialias:
            x = self.layers[0](x)
            for i in range(1, n - 3):
                x = self.layers[i](x)
            return _ac(self.layers[n - 2], x, self.layers[n - 3], post=1 if isinstance(self.layers[n - 1], nn.Tanh) else 0)
        if _wants_grad(self, x):
            x = self.layers[0](x)
            for i in range(1, n - 3):
                x = self.layers[i](x)
            return self.layers[n - 2](x, act=self.layers[n - 3], post=1 if isinstance(self.layers[n - 1], nn.Tanh) else 0)
        # every block starts with an activation whose only reader is that block's transposed conv: the producer's
        # store applies it (layers[i + 1].layers[0] for the next block, layers[n - 3] before the last conv)
        nxt = lambda i: self.layers[i + 1].layers[0] if i + 1 < n - 3 else self.layers[n - 3]
        x = self.layers[0](x, post_act=nxt(0))
        for i in range(1, n - 3):
            x = self.layers[i](x, pre_activated=True, post_act=nxt(i))
        post = 1 if isinstance(self.layers[n - 1], nn.Tanh) else 0
        return self.layers[n - 2](x, post=post)   # tanh fused into the store


class AudioAutoencoder(nn.Module):
    """autoencoders.py:230-560 (encode / decode / chunked encode_audio / decode_audio)."""

    def __init__(self, encoder, decoder, latent_dim, downsampling_ratio, sample_rate, io_channels=2,
                 bottleneck: Bottleneck = None, pretransform=None, in_channels=None, out_channels=None,
                 soft_clip=False):
        super().__init__()
        self.downsampling_ratio = downsampling_ratio
        self.sample_rate = sample_rate
        self.latent_dim = latent_dim
        self.io_channels = io_channels
        self.in_channels = io_channels if in_channels is None else in_channels
        self.out_channels = io_channels if out_channels is None else out_channels
        self.min_length = self.downsampling_ratio
        self.bottleneck = bottleneck
        self.encoder = encoder
        self.decoder = decoder
        if pretransform is not None:
            raise NotImplementedError("nested pretransforms inside the autoencoder")
        self.pretransform = None
        self.soft_clip = soft_clip
        self.is_discrete = self.bottleneck is not None and self.bottleneck.is_discrete

    def encode(self, audio, return_info=False, skip_pretransform=False, iterate_batch=False, **kwargs):
        info = {}
        if self.encoder is not None:
            if iterate_batch:
                latents = torch.cat([self.encoder(audio[i:i + 1]) for i in range(audio.shape[0])], dim=0)
            else:
                latents = self.encoder(audio)
        else:
            latents = audio
        if self.bottleneck is not None:
            latents, bottleneck_info = self.bottleneck.encode(latents, return_info=True, **kwargs)
            info.update(bottleneck_info)
        if return_info:
            return latents, info
        return latents

    def decode(self, latents, iterate_batch=False, **kwargs):
        if self.bottleneck is not None:
            latents = self.bottleneck.decode(latents)
        if iterate_batch:
            decoded = torch.cat([self.decoder(latents[i:i + 1]) for i in range(latents.shape[0])], dim=0)
        else:
            decoded = self.decoder(latents, **kwargs)
        if self.soft_clip:
            decoded = torch.tanh(decoded)
        return decoded

    # ---- chunked encode / decode (autoencoders.py:429-560) as a batched pipeline ------------------------------------------
    # The reference walks the chunks one by one (one encoder / decoder pass each, paste on the way).  Here the chunk layout
    # is planned once on the host, all chunks of all batch items ride on the batch axis of ONE pass (kalle_segment_copy
    # gathers them), and a second kalle_segment_copy pastes the trimmed centres - the same samples land in the same places.
    chunk_batch = 64          # most (chunk, item) rows per encoder / decoder pass

    @staticmethod
    def _chunk_plan(total_in, chunk_in, hop_in, to_out, chunk_out, total_out, trim):
        """Where each chunk is read and which part of its result is kept.
        Chunks start every `hop_in` input positions; when they do not end exactly at the signal's end, one more chunk is
        anchored there (autoencoders.py:462-466, 524-528).  A chunk's result covers `chunk_out` output positions from
        to_out(start); `trim` positions are dropped on every edge that touches a neighbour (480-490, 545-555); the anchored
        last chunk ends at total_out.  The reference pastes in order, so where two kept regions overlap the later chunk wins:
        the earlier region is cut back to the later one's start.  Returns [(src_start, keep_from, dst_start, length)]."""
        if chunk_in > total_in:
            raise ValueError(f"chunked processing needs at least one full chunk: {total_in} < chunk size {chunk_in} "
                             "(the reference fails here too: autoencoders.py:462-466); use chunked=False")
        starts = list(range(0, total_in - chunk_in + 1, hop_in))
        anchored = starts[-1] + chunk_in != total_in
        if anchored:
            starts.append(total_in - chunk_in)
        n = len(starts)
        spans = []
        for i, s0 in enumerate(starts):
            lo = total_out - chunk_out if (i == n - 1) else to_out(s0)
            hi = lo + chunk_out
            keep = 0
            if i > 0:
                lo, keep = lo + trim, trim
            if i < n - 1:
                hi -= trim
            spans.append([s0, keep, lo, hi])
        for i in range(n - 2, -1, -1):
            spans[i][3] = min(spans[i][3], spans[i + 1][2])
        return [(s0, keep, lo, max(hi - lo, 0)) for s0, keep, lo, hi in spans]

    def _run_chunked(self, fn, x, plan, chunk_in, total_out):
        from ... import ops
        B, C, _ = x.shape
        x = x.contiguous()
        y_final = None
        per_pass = max(1, self.chunk_batch // B)
        differentiable = _wants_grad(self, x)     # enable_grad pretransform (models/factory.py:77-80) with chunked=True
        for g0 in range(0, len(plan), per_pass):
            grp = plan[g0:g0 + per_pass]
            n = len(grp)
            if differentiable:
                # the raw segment-copy kernels leave no autograd nodes: gather and paste with torch slicing, which - like the
                # reference's slice-assign loop (autoencoders.py:468-494, 530-558) - is differentiable; same plan, same samples
                y = fn(torch.cat([x[:, :, p[0]:p[0] + chunk_in] for p in grp], dim=0))
                if y_final is None:
                    y_final = torch.zeros((B, y.shape[1], total_out), device=x.device, dtype=y.dtype)
                for i, (_, keep, dst, ln) in enumerate(grp):
                    if ln > 0:
                        y_final[:, :, dst:dst + ln] = y[i * B:(i + 1) * B, :, keep:keep + ln]
                continue
            stacked = torch.empty((n * B, C, chunk_in), device=x.device, dtype=x.dtype)
            ops.segment_copy(x, stacked, [p[0] for p in grp], [0] * n, [chunk_in] * n, B, C,
                             src_strides=(0, x.stride(0), x.stride(1)), dst_strides=(B * C * chunk_in, C * chunk_in, chunk_in))
            y = fn(stacked).contiguous()                                   # ONE encoder / decoder pass for n chunks
            Cy, Ly = y.shape[1], y.shape[2]
            if y_final is None:
                y_final = torch.zeros((B, Cy, total_out), device=x.device, dtype=y.dtype)
            ops.segment_copy(y, y_final, [p[1] for p in grp], [p[2] for p in grp], [p[3] for p in grp], B, Cy,
                             src_strides=(B * Cy * Ly, Cy * Ly, Ly), dst_strides=(0, y_final.stride(0), y_final.stride(1)))
        return y_final

    def encode_audio(self, audio, chunked=False, overlap=32, chunk_size=128, **kwargs):
        """autoencoders.py:429-497; chunk_size / overlap in latents"""
        if not chunked:
            return self.encode(audio, **kwargs)
        r = self.downsampling_ratio
        chunk_in, hop_in = chunk_size * r, (chunk_size - overlap) * r
        total_out = audio.shape[2] // r
        plan = self._chunk_plan(audio.shape[2], chunk_in, hop_in, lambda s0: s0 // r, chunk_size, total_out, overlap // 2)
        return self._run_chunked(lambda c: self.encode(c, **kwargs), audio, plan, chunk_in, total_out)

    def decode_audio(self, latents, chunked=False, overlap=32, chunk_size=128, **kwargs):
        """autoencoders.py:499-560"""
        if not chunked:
            return self.decode(latents, **kwargs)
        r = self.downsampling_ratio
        total_out = latents.shape[2] * r
        plan = self._chunk_plan(latents.shape[2], chunk_size, chunk_size - overlap, lambda s0: s0 * r, chunk_size * r,
                                total_out, (overlap // 2) * r)
        return self._run_chunked(lambda c: self.decode(c, **kwargs), latents, plan, chunk_size, total_out)


def create_encoder_from_config(encoder_config: Dict[str, Any]):
    """autoencoders.py:611-650 (oobleck only)"""
    encoder_type = encoder_config.get("type", None)
    assert encoder_type is not None, "Encoder type must be specified"
    if encoder_type != "oobleck":
        raise NotImplementedError(f"encoder type {encoder_type!r}: only 'oobleck' is on the accelerated path")
    encoder = OobleckEncoder(**encoder_config["config"])
    if not encoder_config.get("requires_grad", True):
        for param in encoder.parameters():
            param.requires_grad = False
    return encoder


def create_decoder_from_config(decoder_config: Dict[str, Any]):
    """autoencoders.py:652-685 (oobleck only)"""
    decoder_type = decoder_config.get("type", None)
    assert decoder_type is not None, "Decoder type must be specified"
    if decoder_type != "oobleck":
        raise NotImplementedError(f"decoder type {decoder_type!r}: only 'oobleck' is on the accelerated path")
    decoder = OobleckDecoder(**decoder_config["config"])
    if not decoder_config.get("requires_grad", True):
        for param in decoder.parameters():
            param.requires_grad = False
    return decoder


def create_autoencoder_from_config(config: Dict[str, Any]):
    """autoencoders.py:687-731"""
    from .factory import create_bottleneck_from_config
    ae_config = config["model"]
    encoder = create_encoder_from_config(ae_config["encoder"])
    decoder = create_decoder_from_config(ae_config["decoder"])
    bottleneck = ae_config.get("bottleneck", None)
    latent_dim = ae_config.get("latent_dim", None)
    assert latent_dim is not None, "latent_dim must be specified in model config"
    downsampling_ratio = ae_config.get("downsampling_ratio", None)
    assert downsampling_ratio is not None, "downsampling_ratio must be specified in model config"
    io_channels = ae_config.get("io_channels", None)
    assert io_channels is not None, "io_channels must be specified in model config"
    sample_rate = config.get("sample_rate", None)
    assert sample_rate is not None, "sample_rate must be specified in model config"
    if ae_config.get("pretransform", None) is not None:
        raise NotImplementedError("nested pretransforms inside the autoencoder")
    if bottleneck is not None:
        bottleneck = create_bottleneck_from_config(bottleneck)
    return AudioAutoencoder(encoder, decoder, io_channels=io_channels, latent_dim=latent_dim,
                            downsampling_ratio=downsampling_ratio, sample_rate=sample_rate, bottleneck=bottleneck,
                            in_channels=ae_config.get("in_channels", None),
                            out_channels=ae_config.get("out_channels", None),
                            soft_clip=ae_config["decoder"].get("soft_clip", False))
