"""Drop-in for the mel-VAE of the reference's backup/flows.py (imported by its inference scripts as
`from flows import BigVGANFlowVAE as Generator`, infer_0828_sigma.py:18): Snake (9-62), SnakeBeta (65-126), Conv1d_S
(134-170), ResStack (172-191), Encoder (194-241), AMPBlock1/2 (243-335), causal ConvTranspose1d (337-391),
BigVGANFlowVAE (396-541), causal Conv1d (548-609), WN (623-695), Flip, ResidualCouplingLayer/Block (698-790).

Same class names, constructor arguments and state-dict keys (old-style weight_norm `weight_g`/`weight_v`, the
Activation1d sub-keys `act.alpha`, `upsample.filter`, `downsample.lowpass.filter`), so a checkpoint of the reference
loads unchanged.  Forward only.  Every convolution, activation and FIR runs in kalle_audio_amd/csrc/conv1d.hip:
LeakyReLU / Snake / the WaveNet gate are fused into the consuming conv's input staging, residual adds, the mean over the
parallel AMP blocks and the final tanh into the producing conv's store, and the anti-aliased activation
(alias-free-torch `Activation1d`, third-party, absent from the reference tree: restated, parity unpinned) is one kernel.
torch only slices / concatenates / flips channels in the coupling flow.
"""
import torch
from torch import nn
from torch.nn import Parameter

from . import conv_ops


def _prep(x):
    if not x.is_cuda:
        raise RuntimeError("kalle_audio_amd mel-VAE modules run on an MI355X GPU only (no CPU fallback)")
    return x if x.dtype in (torch.float32, torch.bfloat16) else x.float()


def _h(h, key):
    return h[key] if isinstance(h, dict) else getattr(h, key)


def get_padding(kernel_size, dilation=1):
    """backup/flows.py:545-546"""
    return int((kernel_size * dilation - dilation) / 2)


class Snake(nn.Module):
    """backup/flows.py:9-62: x + sin^2(a x)/(a + 1e-9), a = exp(alpha) when alpha_logscale"""

    def __init__(self, in_features, alpha=1.0, alpha_trainable=True, alpha_logscale=False):
        super().__init__()
        self.in_features = in_features
        self.alpha_logscale = alpha_logscale
        init = torch.zeros(in_features) if alpha_logscale else torch.ones(in_features)
        self.alpha = Parameter(init * alpha, requires_grad=alpha_trainable)
        self.no_div_by_zero = 0.000000001

    def _ab(self):
        a = self.alpha.detach().float()
        return a, a

    def forward(self, x):
        a, b = self._ab()
        return conv_ops.snake_beta(_prep(x), a, b, self.alpha_logscale)


class SnakeBeta(Snake):
    """backup/flows.py:65-126: x + sin^2(a x)/(b + 1e-9)"""

    def __init__(self, in_features, alpha=1.0, alpha_trainable=True, alpha_logscale=False):
        super().__init__(in_features, alpha, alpha_trainable, alpha_logscale)
        init = torch.zeros(in_features) if alpha_logscale else torch.ones(in_features)
        self.beta = Parameter(init * alpha, requires_grad=alpha_trainable)

    def _ab(self):
        return self.alpha.detach().float(), self.beta.detach().float()


class _Resample(nn.Module):
    """holder of the 12-tap kaiser-sinc filter buffer under alias-free-torch's names"""

    def __init__(self, nested):
        super().__init__()
        if nested:
            self.lowpass = _Resample(False)
        else:
            self.register_buffer("filter", conv_ops.kaiser_sinc_filter12("cpu").view(1, 1, -1))


class Activation1d(nn.Module):
    """alias-free-torch Activation1d(up_ratio=2, down_ratio=2, kernel 12) around a Snake/SnakeBeta: one fused kernel"""

    def __init__(self, activation, up_ratio=2, down_ratio=2, up_kernel_size=12, down_kernel_size=12):
        super().__init__()
        if (up_ratio, down_ratio, up_kernel_size, down_kernel_size) != (2, 2, 12, 12):
            raise NotImplementedError("Activation1d: only the 2x / 12-tap configuration the mel-VAE uses is built")
        self.act = activation
        self.upsample = _Resample(False)
        self.downsample = _Resample(True)

    def forward(self, x):
        a, b = self.act._ab()
        return conv_ops.act1d(_prep(x), self.upsample.filter.view(-1), a, b, self.act.alpha_logscale)


class _ConvBase(nn.Module):
    """Conv1d / ConvTranspose1d parameter holder; weight_norm (old style, dim 0) applied by `weight_norm()` below"""
    transposed = False

    def _make(self, cin, cout, k, bias):
        ref = (nn.ConvTranspose1d if self.transposed else nn.Conv1d)(cin, cout, k, bias=bias)
        self.in_channels, self.out_channels, self.ksize = cin, cout, k
        self.weight = Parameter(ref.weight.detach().clone())
        self.bias = Parameter(ref.bias.detach().clone()) if bias else None
        self._wn = False

    def _apply_weight_norm(self):
        w = self.weight.detach()
        del self.weight
        self.weight_g = Parameter(w.flatten(1).norm(dim=1).view(-1, 1, 1).clone())
        self.weight_v = Parameter(w.clone())
        self._wn = True
        return self

    def _remove_weight_norm(self):
        if self._wn:
            v, g = self.weight_v.detach(), self.weight_g.detach()
            w = g * v / v.flatten(1).norm(dim=1).view(-1, 1, 1)
            del self.weight_g, self.weight_v
            self.weight = Parameter(w)
            self._wn = False
        return self

    def _packed(self):
        ps = (self.weight_g, self.weight_v) if self._wn else (self.weight,)
        key = tuple(p._version for p in ps) + (ps[-1].device, self._wn)
        c = getattr(self, "_kalle_packed", None)
        if c is None or c[0] != key:
            c = (key, conv_ops.weight_norm_fold(ps[-1], ps[0] if self._wn else None, transposed=self.transposed))
            self._kalle_packed = c
        return c[1]

    def _b(self):
        return self.bias.detach().float() if self.bias is not None else None


def weight_norm(m):
    """torch.nn.utils.weight_norm for the conv modules of this file"""
    return m._apply_weight_norm()


def remove_weight_norm(m):
    return m._remove_weight_norm()


def init_weights(m, mean=0.0, std=0.01):
    """backup/flows.py:128-131"""
    if isinstance(m, _ConvBase):
        (m.weight_v if m._wn else m.weight).data.normal_(mean, std)


class Conv1d(_ConvBase):
    """backup/flows.py:548-609: 'same' padding or causal (left-only) padding; bn / activation / input_transpose of the
    reference signature are unused by the VAE and not built.  Extra keyword arguments of forward() select what the
    kernel fuses (input activation, residual, output scale, accumulate, tanh)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, dilation=1, groups=1, padding_mode='zeros',
                 bias=True, padding=None, causal=False, bn=False, activation=None, w_init_gain=None,
                 input_transpose=False, **kwargs):
        super().__init__()
        if groups != 1 or bn or activation is not None or input_transpose or padding_mode != 'zeros':
            raise NotImplementedError("Conv1d: groups / bn / activation / input_transpose are not used by the mel-VAE")
        self.causal = causal
        if padding is None:
            if causal:
                self.pad_left, self.pad_right = dilation * (kernel_size - 1), 0
            else:
                self.pad_left = self.pad_right = get_padding(kernel_size, dilation)
        else:
            self.pad_left = self.pad_right = padding
            if causal:
                self.pad_left += dilation * (kernel_size - 1)
        self.stride, self.dilation = stride, dilation
        self._make(in_channels, out_channels, kernel_size, bias)
        if w_init_gain is not None:
            nn.init.xavier_uniform_(self.weight, gain=nn.init.calculate_gain(w_init_gain))

    def forward(self, x, act=0, act_param=0.0, alpha=None, beta=None, logscale=True, residual=None, post=0,
                out_scale=1.0, accumulate_into=None):
        return conv_ops.conv1d(_prep(x), self._packed(), self._b(), Cout=self.out_channels, K=self.ksize,
                               stride=self.stride, padding=self.pad_left, pad_right=self.pad_right,
                               dilation=self.dilation, act=act, act_param=act_param, alpha=alpha, beta=beta,
                               logscale=logscale, residual=residual, post=post, out_scale=out_scale,
                               accumulate_into=accumulate_into)


class ConvTranspose1d(_ConvBase):
    """backup/flows.py:337-391: causal => padding 0, kernel == 2*stride, last `stride` outputs dropped"""
    transposed = True

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, output_padding=0, groups=1, bias=True,
                 dilation=1, padding=None, padding_mode='zeros', causal=False, input_transpose=False, **kwargs):
        super().__init__()
        if groups != 1 or dilation != 1 or output_padding != 0 or input_transpose:
            raise NotImplementedError("ConvTranspose1d: groups / dilation / output_padding / input_transpose")
        if padding is None:
            padding = 0 if causal else (kernel_size - stride) // 2
        if causal:
            assert padding == 0, "padding is not allowed in causal ConvTranspose1d."
            assert kernel_size == 2 * stride, "kernel_size must be equal to 2*stride in Causal ConvTranspose1d."
        self.causal, self.stride, self.padding = causal, stride, padding
        self._make(in_channels, out_channels, kernel_size, bias)

    def forward(self, x):
        return conv_ops.conv_transpose1d(_prep(x), self._packed(), self._b(), Cout=self.out_channels, K=self.ksize,
                                         stride=self.stride, padding=self.padding,
                                         trim=self.stride if self.causal else 0)


class Conv1d_S(nn.Module):
    """backup/flows.py:134-170 (weight_norm only; spectral norm / init types are training-time options not built)"""

    def __init__(self, in_channels, out_channels, kernel_size=1, stride=1, dilation=1, groups=1,
                 norm_type="weight_norm", init_type=None):
        super().__init__()
        if norm_type != "weight_norm" or groups != 1:
            raise NotImplementedError("Conv1d_S: only norm_type='weight_norm', groups=1")
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, kernel_size
        self.stride, self.dilation, self.groups = stride, dilation, groups
        pad = dilation * (kernel_size - 1) // 2
        self.layer = weight_norm(Conv1d(in_channels, out_channels, kernel_size, stride=stride, dilation=dilation,
                                        padding=pad))

    def forward(self, inputs, **fuse):
        return self.layer(inputs, **fuse)


class ResStack(nn.Module):
    """backup/flows.py:172-191; LeakyReLU (slope 0.01) fused into both convs, the residual add into the second"""

    def __init__(self, channel, kernel_size=3, base=3, nums=4):
        super().__init__()
        self.layers = nn.ModuleList([
            nn.Sequential(
                nn.LeakyReLU(),
                weight_norm(Conv1d(channel, channel, kernel_size, dilation=base ** i, padding=base ** i)),
                nn.LeakyReLU(),
                weight_norm(Conv1d(channel, channel, kernel_size, dilation=1, padding=1)))
            for i in range(nums)])

    def forward(self, x):
        x = _prep(x)
        for layer in self.layers:
            h = layer[1](x, act=3, act_param=layer[0].negative_slope)
            x = layer[3](h, act=3, act_param=layer[2].negative_slope, residual=x)
        return x


class Encoder(nn.Module):
    """backup/flows.py:194-241"""

    def __init__(self, in_channels=1, out_channels=100, base_channels=12, proj_kernel_size=3, stack_kernel_size=3,
                 stack_dilation_base=2, stacks=6, channels=[12, 24, 48, 96, 192, 384, 768],
                 down_sample_factors=[2, 2, 2, 2, 4, 4], use_vae=False):
        super().__init__()
        act_slope = 0.2
        if use_vae:
            out_channels = out_channels * 2
        layers = [Conv1d_S(in_channels, base_channels, kernel_size=proj_kernel_size, stride=1),
                  nn.LeakyReLU(act_slope, True)]
        for (in_c, out_c), down_f in zip(zip(channels[:-1], channels[1:]), down_sample_factors):
            layers += [Conv1d_S(in_c, out_c, kernel_size=down_f * 2, stride=down_f),
                       ResStack(out_c, stack_kernel_size, stack_dilation_base, stacks),
                       nn.LeakyReLU(act_slope, True)]
        layers += [Conv1d_S(channels[-1], out_channels, proj_kernel_size, stride=1)]
        self.generator = nn.Sequential(*layers)

    def forward(self, conditions, z_inputs=None):
        x = _prep(conditions)
        slope = None     # a pending LeakyReLU is applied by the next conv while it stages its input
        for m in self.generator:
            if isinstance(m, nn.LeakyReLU):
                slope = m.negative_slope
            elif isinstance(m, Conv1d_S):
                x = m(x) if slope is None else m(x, act=3, act_param=slope)
                slope = None
            else:
                assert slope is None
                x = m(x)
        return x


class _AMPBase(nn.Module):
    def _make_acts(self, h, channels, activation):
        if activation == 'snake':
            mk = Snake
        elif activation == 'snakebeta':
            mk = SnakeBeta
        else:
            raise NotImplementedError(
                "activation incorrectly specified. check the config file and look for 'activation'.")
        self.activations = nn.ModuleList([
            Activation1d(activation=mk(channels, alpha_logscale=_h(h, "snake_logscale")))
            for _ in range(self.num_layers)])


class AMPBlock1(_AMPBase):
    """backup/flows.py:243-295; `out_scale` / `accumulate_into` fold the mean over parallel blocks into the last conv"""

    def __init__(self, h, channels, kernel_size=3, dilation=(1, 3, 5), activation=None, causal=True):
        super().__init__()
        self.h = h
        self.convs1 = nn.ModuleList([weight_norm(Conv1d(channels, channels, kernel_size, 1, dilation=d, causal=causal))
                                     for d in dilation[:3]])
        self.convs1.apply(init_weights)
        self.convs2 = nn.ModuleList([weight_norm(Conv1d(channels, channels, kernel_size, 1, dilation=1, causal=causal))
                                     for _ in range(3)])
        self.convs2.apply(init_weights)
        self.num_layers = len(self.convs1) + len(self.convs2)
        self._make_acts(h, channels, activation)

    def forward(self, x, out_scale=1.0, accumulate_into=None):
        acts1, acts2 = self.activations[::2], self.activations[1::2]
        n = len(self.convs1)
        for j, (c1, c2, a1, a2) in enumerate(zip(self.convs1, self.convs2, acts1, acts2)):
            xt = c1(a1(x))
            xt = a2(xt)
            if j == n - 1:
                x = c2(xt, residual=x, out_scale=out_scale, accumulate_into=accumulate_into)
            else:
                x = c2(xt, residual=x)
        return x

    def remove_weight_norm(self):
        for l in list(self.convs1) + list(self.convs2):
            remove_weight_norm(l)


class AMPBlock2(_AMPBase):
    """backup/flows.py:297-335"""

    def __init__(self, h, channels, kernel_size=3, dilation=(1, 3), activation=None, causal=True):
        super().__init__()
        self.h = h
        self.convs = nn.ModuleList([weight_norm(Conv1d(channels, channels, kernel_size, 1, dilation=d, causal=causal))
                                    for d in dilation[:2]])
        self.convs.apply(init_weights)
        self.num_layers = len(self.convs)
        self._make_acts(h, channels, activation)

    def forward(self, x, out_scale=1.0, accumulate_into=None):
        n = len(self.convs)
        for j, (c, a) in enumerate(zip(self.convs, self.activations)):
            if j == n - 1:
                x = c(a(x), residual=x, out_scale=out_scale, accumulate_into=accumulate_into)
            else:
                x = c(a(x), residual=x)
        return x

    def remove_weight_norm(self):
        for l in self.convs:
            remove_weight_norm(l)


class WN(nn.Module):
    """backup/flows.py:623-695 (g=None path; the tanh*sigmoid gate is fused into the 1x1 res/skip conv's input)"""

    def __init__(self, hidden_channels, kernel_size, dilation_rate, n_layers, gin_channels=0, p_dropout=0, causal=False):
        super().__init__()
        assert kernel_size % 2 == 1
        if gin_channels != 0:
            raise NotImplementedError("WN: conditioning input (gin_channels) is not used by BigVGANFlowVAE")
        self.hidden_channels, self.kernel_size = hidden_channels, (kernel_size,)
        self.dilation_rate, self.n_layers, self.gin_channels, self.p_dropout = dilation_rate, n_layers, 0, p_dropout
        self.in_layers = nn.ModuleList()
        self.res_skip_layers = nn.ModuleList()
        for i in range(n_layers):
            self.in_layers.append(weight_norm(Conv1d(hidden_channels, 2 * hidden_channels, kernel_size,
                                                     dilation=dilation_rate ** i, causal=causal)))
            rs = 2 * hidden_channels if i < n_layers - 1 else hidden_channels
            self.res_skip_layers.append(weight_norm(Conv1d(hidden_channels, rs, 1, causal=causal)))

    def forward(self, x, x_mask=None, g=None, **kwargs):
        H = self.hidden_channels
        output = None
        for i in range(self.n_layers):
            x_in = self.in_layers[i](x)
            rs = self.res_skip_layers[i](x_in, act=4)
            if i < self.n_layers - 1:
                x = x + rs[:, :H]
                skip = rs[:, H:]
            else:
                skip = rs
            output = skip if output is None else output + skip
            if x_mask is not None:
                x = x * x_mask
        return output if x_mask is None else output * x_mask

    def remove_weight_norm(self):
        for l in list(self.in_layers) + list(self.res_skip_layers):
            remove_weight_norm(l)


class Flip(nn.Module):
    """backup/flows.py:698-705"""

    def forward(self, x, *args, reverse=False, **kwargs):
        x = torch.flip(x, [1])
        if not reverse:
            return x, torch.zeros(x.size(0), dtype=x.dtype, device=x.device)
        return x


class ResidualCouplingLayer(nn.Module):
    """backup/flows.py:709-755"""

    def __init__(self, channels, hidden_channels, kernel_size, dilation_rate, n_layers, p_dropout=0, gin_channels=0,
                 mean_only=False, causal=True):
        assert channels % 2 == 0, "channels should be divisible by 2"
        super().__init__()
        self.channels, self.hidden_channels, self.kernel_size = channels, hidden_channels, kernel_size
        self.dilation_rate, self.n_layers, self.half_channels = dilation_rate, n_layers, channels // 2
        self.mean_only = mean_only
        self.pre = Conv1d(self.half_channels, hidden_channels, 1, causal=causal)
        self.enc = WN(hidden_channels, kernel_size, dilation_rate, n_layers, p_dropout=p_dropout,
                      gin_channels=gin_channels, causal=causal)
        self.post = Conv1d(hidden_channels, self.half_channels * (2 - mean_only), 1, causal=causal)
        self.post.weight.data.zero_()
        self.post.bias.data.zero_()

    def forward(self, x, x_mask=None, g=None, reverse=False):
        x = _prep(x)
        x0, x1 = torch.split(x, [self.half_channels] * 2, 1)
        h = self.pre(x0.contiguous())
        if x_mask is not None:
            h = h * x_mask
        h = self.enc(h, x_mask, g=g)
        stats = self.post(h)
        if x_mask is not None:
            stats = stats * x_mask
        if not self.mean_only:
            m, logs = torch.split(stats, [self.half_channels] * 2, 1)
        else:
            m, logs = stats, torch.zeros_like(stats)
        if not reverse:
            x1 = m + x1 * torch.exp(logs) if not self.mean_only else m + x1
            if x_mask is not None:
                x1 = m + (x1 - m) * x_mask
            return torch.cat([x0, x1], 1), torch.sum(logs, [1, 2])
        x1 = (x1 - m) * torch.exp(-logs) if not self.mean_only else x1 - m
        if x_mask is not None:
            x1 = x1 * x_mask
        return torch.cat([x0, x1], 1)


class ResidualCouplingBlock(nn.Module):
    """backup/flows.py:759-790"""

    def __init__(self, channels, hidden_channels, kernel_size, dilation_rate, n_layers, n_flows=4, gin_channels=0,
                 causal=True):
        super().__init__()
        self.channels, self.hidden_channels, self.kernel_size = channels, hidden_channels, kernel_size
        self.dilation_rate, self.n_layers, self.n_flows, self.gin_channels = dilation_rate, n_layers, n_flows, gin_channels
        self.flows = nn.ModuleList()
        for _ in range(n_flows):
            self.flows.append(ResidualCouplingLayer(channels, hidden_channels, kernel_size, dilation_rate, n_layers,
                                                    gin_channels=gin_channels, mean_only=True, causal=causal))
            self.flows.append(Flip())

    def forward(self, x, x_mask=None, g=None, reverse=False):
        if not reverse:
            for flow in self.flows:
                x, _ = flow(x, x_mask, g=g, reverse=reverse)
        else:
            for flow in reversed(self.flows):
                x = flow(x, x_mask, g=g, reverse=reverse)
        return x


class BigVGANFlowVAE(nn.Module):
    """backup/flows.py:396-541.  `h` is the reference's attribute-style hyper-parameter object (a dict works too)."""

    def __init__(self, h):
        super().__init__()
        self.h = h
        causal = _h(h, "causal")
        latent = _h(h, "latent_dim")
        self.latent_dim, self.use_vae = latent, _h(h, "use_vae")
        self.audio_encoder = Encoder(out_channels=latent, use_vae=self.use_vae, channels=_h(h, "downsample_channels"),
                                     down_sample_factors=_h(h, "downsample_rates"))
        self.flow = ResidualCouplingBlock(latent, _h(h, "flow_hidden_channels"), 5, 1, 4, gin_channels=0, causal=causal)
        ks, ds = _h(h, "resblock_kernel_sizes"), _h(h, "resblock_dilation_sizes")
        ur, uk, uc = _h(h, "upsample_rates"), _h(h, "upsample_kernel_sizes"), _h(h, "upsample_initial_channel")
        self.num_kernels, self.num_upsamples = len(ks), len(ur)
        self.conv_pre = weight_norm(Conv1d(latent, uc, 7, 1, causal=False))
        resblock = AMPBlock1 if _h(h, "resblock") == '1' else AMPBlock2
        self.ups = nn.ModuleList()
        for i, (u, k) in enumerate(zip(ur, uk)):
            self.ups.append(nn.ModuleList([
                weight_norm(ConvTranspose1d(uc // (2 ** i), uc // (2 ** (i + 1)), k, u, causal=causal))]))
        self.resblocks = nn.ModuleList()
        for i in range(len(self.ups)):
            ch = uc // (2 ** (i + 1))
            for k, d in zip(ks, ds):
                self.resblocks.append(resblock(h, ch, k, d, activation=_h(h, "activation"), causal=causal))
        act = _h(h, "activation")
        if act == "snake":
            self.activation_post = Activation1d(activation=Snake(ch, alpha_logscale=_h(h, "snake_logscale")))
        elif act == "snakebeta":
            self.activation_post = Activation1d(activation=SnakeBeta(ch, alpha_logscale=_h(h, "snake_logscale")))
        else:
            raise NotImplementedError(
                "activation incorrectly specified. check the config file and look for 'activation'.")
        self.conv_post = weight_norm(Conv1d(ch, 1, 7, 1, causal=causal))
        for i in range(len(self.ups)):
            self.ups[i].apply(init_weights)
        self.conv_post.apply(init_weights)

    def _decode(self, z):
        x = self.conv_pre(z)
        nk = self.num_kernels
        for i in range(self.num_upsamples):
            for up in self.ups[i]:
                x = up(x)
            xs = None
            for j in range(nk):
                # mean over the parallel AMP blocks: each block's last conv writes (conv + skip)/nk, accumulating
                xs = self.resblocks[i * nk + j](x, out_scale=1.0 / nk, accumulate_into=xs)
            x = xs
        x = self.activation_post(x)
        return self.conv_post(x, post=1)

    def _sample(self, x, noise=None):
        m_q, logs_q = torch.split(x, self.latent_dim, dim=1)
        eps = torch.randn_like(m_q) if noise is None else noise
        return (m_q + eps * torch.exp(logs_q)).contiguous(), logs_q

    def forward(self, x, noise=None):
        x = self.audio_encoder(x)
        assert self.use_vae
        z, logs_q = self._sample(x, noise)
        z_p = self.flow(z, None)
        x = self._decode(z)
        return x, (z_p, logs_q, None, None)

    def extract_latents(self, x):
        return self.audio_encoder(x)

    def inference_from_latents(self, x, do_sample=True, noise=None):
        x = _prep(x)
        if self.use_vae and do_sample:
            assert x.size(1) == self.latent_dim * 2, "Input must be like [B, D, H]"
            x, _ = self._sample(x, noise)
        else:
            assert x.size(1) == self.latent_dim, "Input must be like [B, D, H]"
        return self._decode(x.contiguous())

    def remove_weight_norm(self):
        for l in self.ups:
            for l_i in l:
                remove_weight_norm(l_i)
        for l in self.resblocks:
            l.remove_weight_norm()
        remove_weight_norm(self.conv_pre)
        remove_weight_norm(self.conv_post)
