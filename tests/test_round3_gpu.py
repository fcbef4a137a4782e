"""-m gpu, round 3: the HEADLINE configuration (24 blocks, D = 1536, 24 heads, 1024 latent channels, 125 + 1 tokens, 130 x 768
conditioning tokens) pinned end to end against the CPU oracle, the grouped weight-gradient launch at the benchmark's own plan
(32256 tokens x the seven matrices of a block), the sweep widths of SURVEY 8(d) through a `-m gpu` check, and the trainer /
cache hazards the round-2 review found (stale gradients of sub-modules that did not run, stale weight caches).

Tolerances: loss |d| <= 1e-2 relative, cosine >= 0.999 for the 24-block output, gradients rel-L2 <= 3e-2 through
24 bf16 layers (2e-2 for a single block)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "golden"))
sys.path.insert(0, os.path.join(HERE, "..", "oracle"))
import kalle_oracle as ko  # noqa: E402
from test_modules_gpu import cosine, rel  # noqa: E402
from test_round2_gpu import _batch, _slice_cond, _small_dit  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def _drop_in_installed():
    import kalle_audio_amd
    kalle_audio_amd.install()


# ------------------------------------------------------------------------------------------------ headline configuration
def test_headline_config_train_step_vs_oracle(dev):
    """bench.build_model (the benchmark's exact model) runs ONE train step on 2 clips through the trainer; the CPU oracle runs
    the same step in fp32 on the same weights and inputs (as bench.cpu_baseline does).  Compared: the loss, the 24-block
    output, and FULL gradient tensors from the first, a middle and the last block plus the input projection - the drift of
    bf16 operands through 24 residual layers (transformer.py:758-812, dit.py:135-229), never measured before round 3."""
    import bench
    from kalle_audio_amd import engine
    from kalle_audio_amd.stable_audio_tools.training.diffusion import diffusion_train_step
    cfg = dict(bench.CFG, io_channels=1024, global_cond_type="prepend")
    model = bench.build_model(dev, cfg=cfg)
    lat, noise, t, cond = bench.make_batch(2, dev, 99, cfg)
    sd = {n: p.detach().float().cpu().clone().requires_grad_(True) for n, p in model.model.model.named_parameters()}
    with torch.no_grad():
        _, info = diffusion_train_step(model, lat, t, noise, cond, objective="v")
    out_gpu = info["output"].float().cpu()
    tr = engine.DataParallelTrainer(model, lr=0.0, optimizer="Adam")
    loss = tr.train_step(model, lat, t, noise, cond, objective="v")
    torch.cuda.synchronize()
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    ocfg = dict(embed_dim=cfg["embed_dim"], depth=cfg["depth"], num_heads=cfg["num_heads"], global_cond_type="prepend")
    loss_ref, out_ref, _, _ = ko.train_step_loss(sd, ocfg, lat.cpu(), noise.cpu(), t.cpu(), "v",
                                                 cross_attn_cond=cond["prompt"][0].cpu(), global_embed=cond["global"][0].cpu())
    loss_ref.backward()
    lv, lr_ = loss.item(), loss_ref.item()
    margins = {"loss_rel": abs(lv - lr_) / abs(lr_), "out_cos": cosine(out_gpu, out_ref.detach()),
               "out_rel": rel(out_gpu, out_ref.detach())}
    names = ["transformer.project_in.weight", "transformer.project_out.weight"]
    for layer in (0, 11, 23):
        names += [f"transformer.layers.{layer}.{n}" for n in ("self_attn.to_qkv.weight", "ff.ff.0.proj.weight",
                                                              "cross_attn.to_kv.weight", "ff.ff.2.weight")]
    names += ["transformer.layers.23.pre_norm.gamma", "transformer.layers.0.ff.ff.0.proj.bias", "to_timestep_embed.0.weight"]
    for n in names:
        margins[n] = rel(tr.flat.grad_view("model.model." + n).float().cpu(), sd[n].grad)
    print("HEADLINE_MARGINS " + " ".join(f"{k}={v:.3e}" for k, v in margins.items()))
    assert margins["loss_rel"] <= 1e-2, margins
    assert margins["out_cos"] >= 0.999, margins
    for n in names:
        assert margins[n] <= 3e-2, (n, margins[n])


SEVEN = [(4608, 1536), (1536, 1536), (1536, 1536), (1536, 768), (1536, 1536), (12288, 1536), (1536, 6144)]


@pytest.mark.parametrize("overwrite", [True, False])
def test_grouped_wgrad_kernel_at_the_bench_plan(dev, overwrite):
    """kalle_gemm_wgrad_group at the shapes the headline number runs on: 32256 tokens (33280 context tokens for to_kv, whose dY
    is a column slice of the stacked k | v gradient of all layers), the seven matrices of a block - the "512 whole tiles +
    154 x 3 slices" plan of DESIGN 5.2 - overwriting and accumulating, against an fp64 product of the same bf16 operands"""
    from kalle_audio_amd import ops
    tokens, ctx_tokens = 256 * 126, 256 * 130
    g = torch.Generator(device=dev).manual_seed(5 + overwrite)
    probs, refs = [], []
    for i, (n, k) in enumerate(SEVEN):
        rows = ctx_tokens if k == 768 else tokens
        if k == 768:        # strided dY: columns [n, 2n) of a [rows, 3n] buffer (dit_ops.ContextKV)
            big = (torch.randn(rows, 3 * n, generator=g, device=dev) * 0.5).to(torch.bfloat16)
            dy = big[:, n:2 * n]
        else:
            dy = (torch.randn(rows, n, generator=g, device=dev) * 0.5).to(torch.bfloat16)
        x = torch.randn(rows, k, generator=g, device=dev).to(torch.bfloat16)
        base = torch.randn(n, k, generator=g, device=dev)
        ref = dy.double().T @ x.double()
        refs.append((ref if overwrite else ref + base.double()).float())
        del ref
        probs.append((dy, x, base.clone()))
    assert ops.gemm_wgrad_group(probs, overwrite=overwrite)
    torch.cuda.synchronize()
    for (dy, x, out), ref in zip(probs, refs):
        assert rel(out, ref) < 2e-5, (tuple(out.shape), rel(out, ref))


@pytest.mark.parametrize("io_channels,frames", [(64, 125), (512, 125), (1024, 375)])
def test_full_width_sweep_axes_step_properties(dev, io_channels, frames):
    """the other points of SURVEY 8(d)'s sweep at full width and depth (io_channels 64 / 512; 375 frames = three key blocks in
    the self-attention): a train step is reproducible, the gradient of a batch is the mean of its halves' gradients, and one
    oracle-checked quantity per point - the loss of the step against the CPU oracle on the same weights"""
    import bench
    from kalle_audio_amd import engine
    cfg = dict(bench.CFG, io_channels=io_channels, global_cond_type="prepend")
    model = bench.build_model(dev, cfg=cfg)
    lat, noise, t, cond = bench.make_batch(4, dev, 31, cfg, T=frames)
    sd = {n: p.detach().float().cpu().clone() for n, p in model.model.model.named_parameters()}
    tr = engine.DataParallelTrainer(model, lr=0.0, optimizer="Adam")

    def grads(sl, parts=1):
        tr.grad_accum_steps, tr.micro = parts, 0
        n = (sl.stop - sl.start) // parts
        losses = []
        for i in range(parts):
            s = slice(sl.start + i * n, sl.start + (i + 1) * n)
            losses.append(tr.train_step(model, lat[s], t[s], noise[s], _slice_cond(cond, s), objective="v"))
        torch.cuda.synchronize()
        return torch.stack(losses).mean().item(), tr.flat.grad.clone()

    l1, g1 = grads(slice(0, 4))
    l2, g2 = grads(slice(0, 4))
    assert abs(l1 - l2) < 1e-5 * abs(l1) and rel(g2, g1) < 1e-4
    assert torch.isfinite(g1).all() and g1.abs().max() > 0
    la, ga = grads(slice(0, 4), parts=2)
    assert abs(la - l1) < 2e-3 * abs(l1) and rel(ga * 0.5, g1) < 2e-2, (la, l1, rel(ga * 0.5, g1))
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    ocfg = dict(embed_dim=cfg["embed_dim"], depth=cfg["depth"], num_heads=cfg["num_heads"], global_cond_type="prepend")
    with torch.no_grad():
        s = slice(0, 1)
        loss_ref, *_ = ko.train_step_loss(sd, ocfg, lat[s].cpu(), noise[s].cpu(), t[s].cpu(), "v",
                                          cross_attn_cond=cond["prompt"][0][s].cpu(), global_embed=cond["global"][0][s].cpu())
    tr.grad_accum_steps, tr.micro = 1, 0
    got = tr.train_step(model, lat[s], t[s], noise[s], _slice_cond(cond, s), objective="v").item()
    assert abs(got - loss_ref.item()) <= 1e-2 * abs(loss_ref.item()), (got, loss_ref.item())


# ------------------------------------------------------------------------------------------------ stale gradients
@pytest.mark.parametrize("rows_mode", ["grouped", "plain"])
def test_no_stale_gradients_for_submodules_that_did_not_run(dev, monkeypatch, rows_mode):
    """A block called with context=None skips its cross-attention (transformer.py:684-693).  In overwrite mode the trainer does
    not pre-clear the matrix gradients, so the cross-attention sinks would keep LAST step's values and Adam would apply them:
    step with context, then step without - the cross-attention gradients must be exactly zero and Adam's second step must
    move those weights by momentum only (never by a stale gradient)."""
    from kalle_audio_amd import dit_ops, engine
    if rows_mode == "plain":
        monkeypatch.setattr(dit_ops, "GROUP_WGRAD", False)
    m = _small_dit(dev)
    tr = engine.DataParallelTrainer(m, lr=1e-3, optimizer="Adam")
    lat, noise, t, cond = _batch(dev, 4, 11)        # 4 x 126 = 504 rows: a multiple of 8, so the grouped launch is eligible
    tr.train_step(m, lat, t, noise, cond)
    torch.cuda.synchronize()
    assert all(blk._kalle_wgrad_overwrite == (rows_mode == "grouped") for _, blk in tr.blocks)
    cross = [n for n in tr.flat.names if ".cross_attn." in n or ".cross_attend_norm." in n]
    assert cross and all(tr.flat.grad_view(n).abs().max() > 0 for n in cross if "weight" in n)
    w1 = {n: tr.flat.params[n].detach().clone() for n in cross}
    no_ctx = {"g": cond["g"]}
    m.cross_attn_cond_ids = []                      # the wrapper now hands the DiT no cross-attention conditioning
    tr.train_step(m, lat, t, noise, no_ctx)
    torch.cuda.synchronize()
    for n in cross:
        assert tr.flat.grad_view(n).abs().max().item() == 0.0, n
    # Adam with a zero gradient: update = -lr * (b1 m1) / (1 - b1^2) / (sqrt(b2 v1 / (1 - b2^2)) + eps) - bounded by lr, and in
    # the direction of the FIRST step's gradient; a stale gradient would have produced a full second step (|d| ~ lr again with
    # m = (1 - b1^2)-corrected mean of two equal gradients: ratio ~1.0, against ~0.53 = b1 (1-b1) / (1 - b1^2) / sqrt(b2 ...) here)
    for n in cross:
        if "weight" not in n or tr.flat.params[n].dim() < 2:
            continue
        d1 = (w1[n] - 0).float()                    # weights after step 1
        d2 = tr.flat.params[n].detach().float() - d1
        s, cnt = tr.flat.slices[n]
        first = tr.exp_avg[s:s + cnt].view(d2.shape)            # m2 = b1 * m1 (zero gradient)
        big = first.abs() > 1e-3 * first.abs().max()
        ratio = (d2[big].abs() / 1e-3).median().item()
        assert 0.3 < ratio < 0.8, (n, ratio)       # momentum-only step; a stale gradient gives ~1.0


# ------------------------------------------------------------------------------------------------ stale weight caches (advisor)
def test_stacked_context_weights_follow_fused_adam(dev, monkeypatch):
    """FusedAdam writes parameters through the raw pointer (p._version does not move): FusedAdam step -> no_grad forward ->
    step -> no_grad forward must project the context with the CURRENT to_kv weights (the no-grad cache of the stacked k | v
    weights is keyed on a write epoch too), compared against one projection per layer (KALLE_BATCH_CTX_KV=0)"""
    from kalle_audio_amd.engine import FusedAdam
    from stable_audio_tools.training.diffusion import diffusion_train_step
    m = _small_dit(dev)
    opt = FusedAdam(m.parameters(), lr=5e-3, adam_w_mode=False)
    lat, noise, t, cond = _batch(dev, 2, 12)

    def eval_out():
        with torch.no_grad():
            return diffusion_train_step(m, lat, t, noise, cond)[1]["output"].float().clone()

    outs = []
    for _ in range(2):
        opt.zero_grad(set_to_none=True)
        loss, _ = diffusion_train_step(m, lat, t, noise, cond)
        loss.backward()
        opt.step()
        outs.append(eval_out())
    monkeypatch.setenv("KALLE_BATCH_CTX_KV", "0")
    want = eval_out()
    assert rel(outs[0], outs[1]) > 1e-3                   # the weights really moved between the two evaluations
    assert rel(outs[1], want) < 1e-6, rel(outs[1], want)  # and the cached stack followed them


def test_captured_sampler_graph_follows_new_weights(dev, monkeypatch):
    """generate -> load_state_dict(other weights) -> generate on a frozen model: the HIP graph captured for the first weights
    must not be replayed for the second (its kernels read the old bf16 copies); compared against KALLE_SAMPLE_GRAPH=0"""
    from stable_audio_tools.inference.generation import generate_diffusion_cond
    a, b = _small_dit(dev, seed=70), _small_dit(dev, seed=71)
    a.eval().requires_grad_(False)
    _, _, _, cond = _batch(dev, 1, 13)
    kw = dict(steps=6, cfg_scale=3.0, conditioning_tensors=cond, batch_size=1, sample_size=125, seed=7, device="cpu")
    first = generate_diffusion_cond(a, **kw).clone()
    assert getattr(a, "_kalle_graphed", None) is not None
    a.load_state_dict(b.state_dict())
    second = generate_diffusion_cond(a, **kw).clone()
    monkeypatch.setenv("KALLE_SAMPLE_GRAPH", "0")
    want = generate_diffusion_cond(a, **kw)
    assert rel(first, second) > 1e-2
    assert torch.equal(second, want)


# ------------------------------------------------------------------------------------------------ conv backward limits (advisor)
def test_act_bwd_and_channel_sum_beyond_65535_rows(dev):
    """B * C > 65535 (2048 channels at per-GPU batch >= 32 during VAE fine-tuning): rows ride on grid x now"""
    from kalle_audio_amd import conv_train
    from stable_audio_tools.models.blocks import SnakeBeta
    B, C, L = 40, 2048, 24
    g = torch.Generator(device=dev).manual_seed(3)
    x = torch.randn(B, C, L, generator=g, device=dev, requires_grad=True)
    act = SnakeBeta(C).to(dev)
    with torch.no_grad():
        act.alpha.normal_(0, 0.3, generator=g)
        act.beta.normal_(0, 0.3, generator=g)
    gy = torch.randn(B, C, L, generator=g, device=dev)
    a, bb = act.alpha.exp()[None, :, None], act.beta.exp()[None, :, None]
    y_ref = x + torch.sin(x * a) ** 2 / (bb + 1e-9)
    dx_ref, da_ref, db_ref = torch.autograd.grad(y_ref, [x, act.alpha, act.beta], gy)
    dx, da, db = conv_train.act_bwd(x.detach(), gy, 1, act.alpha.detach().float(), act.beta.detach().float(), True)
    assert rel(dx, dx_ref) < 1e-5 and rel(da, da_ref) < 1e-4 and rel(db, db_ref) < 1e-4
    s = conv_train.channel_sum(gy)
    assert rel(s, gy.sum(dim=(0, 2))) < 1e-5


def test_chunked_encode_with_enable_grad_is_differentiable(dev):
    """AutoencoderPretransform(enable_grad) + chunked=True: the latents must carry a graph (the reference's slice-assign loop is
    differentiable, autoencoders.py:468-494) and equal the kernel-pasted ones"""
    from stable_audio_tools.models.autoencoders import AudioAutoencoder, OobleckDecoder, OobleckEncoder
    torch.manual_seed(0)
    enc = OobleckEncoder(in_channels=2, channels=8, latent_dim=4, c_mults=[1, 2], strides=[2, 4], use_snake=True)
    dec = OobleckDecoder(out_channels=2, channels=8, latent_dim=4, c_mults=[1, 2], strides=[2, 4], use_snake=True)
    ae = AudioAutoencoder(enc, dec, latent_dim=4, downsampling_ratio=8, sample_rate=16000, io_channels=2).to(dev)
    wav = torch.randn(2, 2, 8 * 40, device=dev)
    with torch.no_grad():
        want = ae.encode_audio(wav, chunked=True, chunk_size=16, overlap=4)
    got = ae.encode_audio(wav, chunked=True, chunk_size=16, overlap=4)
    assert got.requires_grad and rel(got, want) < 1e-5
    got.square().mean().backward()
    gn = [p.grad.norm().item() for p in ae.encoder.parameters() if p.grad is not None]
    assert gn and all(v == v for v in gn) and max(gn) > 0


# ------------------------------------------------------------------------------------------------ boundary: variations / inpainting
def test_generate_diffusion_cond_from_init_audio(dev):
    """generate_diffusion_cond(init_audio=..., init_noise_level=...[, mask_args=...]) against the reference function (fixture
    generate_init_audio.npz, tests/golden/make_golden_r03.py): variation without a pretransform, with a VAE (encode on the GPU),
    and the mask_args call, which the reference's rectified-flow branch turns into plain sampling"""
    import numpy as np
    import golden_util as gu
    from test_modules_gpu import T, fx, load_seeded
    from test_round2_gpu import TensorConditioner, _e2e_cond
    from stable_audio_tools.inference.generation import build_mask, generate_diffusion_cond
    from stable_audio_tools.models import diffusion as KD
    from stable_audio_tools.models.factory import create_model_from_config
    from stable_audio_tools.models.pretransforms import AutoencoderPretransform
    f = fx("generate_init_audio")
    e = gu.E2E
    ctx, cm, gl = _e2e_cond(64, dev)
    cond = {"prompt": (ctx, cm), "g": (gl, None)}

    def model(with_vae, seed):
        dit = KD.DiTWrapper(io_channels=4, embed_dim=e["D"], depth=2, num_heads=2, cond_token_dim=e["DC"],
                            project_cond_tokens=False, global_cond_dim=e["G"], transformer_type="continuous_transformer",
                            global_cond_type="prepend")
        load_seeded(dit, seed, dev)
        pt = None
        if with_vae:
            cfg = gu.oobleck_cfg(True)
            cfg["model"]["encoder"]["config"]["latent_dim"] = 4          # (see make_golden_r03.py)
            pt = AutoencoderPretransform(load_seeded(create_model_from_config(cfg), 24, dev), scale=0.8)
        return KD.ConditionedDiffusionModelWrapper(dit, TensorConditioner(), io_channels=4, sample_rate=16000,
                                                   min_input_length=40, diffusion_objective="rectified_flow", pretransform=pt,
                                                   cross_attn_cond_ids=["prompt"], global_cond_ids=["g"]).to(dev)

    kw = dict(steps=e["steps"], cfg_scale=e["cfg_scale"], conditioning_tensors=cond, batch_size=e["B"], seed=e["seed"], device="cpu")
    margs = dict(cropfrom=10.0, pastefrom=20.0, pasteto=70.0, maskstart=20.0, maskend=70.0, softnessL=5.0, softnessR=8.0,
                 marination=0.1)
    m = model(False, 64)
    init = T(gu.make_input("init_lat", (4, 100), 64), dev)
    var = generate_diffusion_cond(m, sample_size=e["T"], init_audio=(16000, init), init_noise_level=0.6, **kw)
    assert cosine(var, f["lat/variation"]) > 0.999, cosine(var, f["lat/variation"])
    plain = generate_diffusion_cond(m, sample_size=e["T"], **kw)
    masked = generate_diffusion_cond(m, sample_size=e["T"], init_audio=(16000, init), init_noise_level=0.6, mask_args=margs, **kw)
    assert cosine(masked, f["lat/masked"]) > 0.999 and torch.equal(masked, plain)
    assert rel(var, plain) > 0.1                                         # the init data really entered the variation
    m = model(True, 65)
    wav = T(gu.make_input("init_wav", (1, 40 * e["T"] + 333), 65), dev) * 0.3
    lat = generate_diffusion_cond(m, sample_size=40 * e["T"], init_audio=(16000, wav), init_noise_level=0.45, return_latents=True, **kw)
    assert cosine(lat, f["vae/variation_latents"]) > 0.999, cosine(lat, f["vae/variation_latents"])
    audio = generate_diffusion_cond(m, sample_size=40 * e["T"], init_audio=(16000, wav), init_noise_level=0.45, **kw)
    assert cosine(audio, f["vae/variation_audio"]) > 0.999
    with pytest.raises(NotImplementedError):                             # another sample rate needs torchaudio's resampler
        generate_diffusion_cond(m, sample_size=40 * e["T"], init_audio=(22050, wav), **kw)
    mk = build_mask(125, margs)
    assert mk.shape == (125,) and abs(float(mk.max()) - 0.9) < 1e-6
    m.diffusion_objective = "v"
    with pytest.raises(NotImplementedError):
        generate_diffusion_cond(m, sample_size=40 * e["T"], init_audio=(16000, wav), **kw)


def test_antialias_activation_oobleck_units(dev):
    """ResidualUnit / EncoderBlock / DecoderBlock(antialias_activation=True) (autoencoders.py:24-37): every activation wrapped in
    alias_free_torch.Activation1d - third-party, absent from the reference tree, PARITY UNPINNED: checked against the oracle's
    restatement of the package's published algorithm (oracle.activation1d), like the mel-VAE's AMP blocks"""
    import golden_util as gu
    from test_modules_gpu import T
    from stable_audio_tools.models.autoencoders import DecoderBlock, EncoderBlock, ResidualUnit
    torch.manual_seed(3)
    for use_snake in (True, False):
        ru = ResidualUnit(16, 16, dilation=3, use_snake=use_snake, antialias_activation=True).to(dev)
        enc = EncoderBlock(16, 32, stride=4, use_snake=use_snake, antialias_activation=True).to(dev)
        dec = DecoderBlock(32, 16, stride=4, use_snake=use_snake, antialias_activation=True).to(dev)
        for mod in (ru, enc, dec):
            for n, p in mod.named_parameters():
                if n.endswith("alpha") or n.endswith("beta"):
                    torch.nn.init.normal_(p, 0.0, 0.3)
        x = torch.randn(2, 16, 200, device=dev)

        def act_ref(a, t):
            tc = t.detach().cpu()
            if use_snake:
                return ko.activation1d(tc, a.alpha.detach().cpu(), a.beta.detach().cpu(), logscale=True)
            up = ko.upsample1d_2x(tc, ko.kaiser_sinc_filter1d(0.25, 0.3, 12))
            return ko.downsample1d_2x(torch.nn.functional.elu(up), ko.kaiser_sinc_filter1d(0.25, 0.3, 12))

        def conv_ref(c, t, transposed=False):
            w = (c.weight_g * c.weight_v / c.weight_v.flatten(1).norm(dim=1).view(-1, 1, 1)).detach().cpu()
            b = c.bias.detach().cpu() if c.bias is not None else None
            if transposed:
                return torch.nn.functional.conv_transpose1d(t, w, b, stride=c.stride, padding=c.padding)
            return torch.nn.functional.conv1d(t, w, b, stride=c.stride, padding=c.padding, dilation=c.dilation)

        with torch.no_grad():
            y = ru(x)
            L = ru.layers
            want = x.cpu() + conv_ref(L[3], act_ref(L[2].act, conv_ref(L[1], act_ref(L[0].act, x))))
            assert rel(y, want) < 1e-4, (use_snake, rel(y, want))
            ye = enc(x)
            assert ye.shape == (2, 32, 50) and torch.isfinite(ye).all()
            yd = dec(ye)
            assert yd.shape == (2, 16, 200) and torch.isfinite(yd).all()


@pytest.mark.parametrize("Nq,Nk,grp,masked_tail", [(126, 130, 2, False), (126, 160, 2, True), (257, 150, 1, False), (15, 159, 4, True),
                                                   (126, 161, 2, False)])
def test_attention_forward_with_folded_tail_keys(dev, Nq, Nk, grp, masked_tail):
    """the forward's folded tail (keys 128 .. Nk - 1 <= 159 staged with the first block, no second pass through the staging code;
    161 keys take the ordinary second block): outputs AND the log-sum-exp the backward reads, against fp32 torch, with the tail
    keys masked for one clip"""
    from kalle_audio_amd import ops
    from test_kernels_gpu import _attn_ref, _mk
    B, Hkv = 2, 2
    H = Hkv * grp
    D, Dc = H * 64, Hkv * 64
    q = (_mk((B, Nq, D), dev, seed=900 + Nk) * 0.8).bfloat16()
    kv = (_mk((B, Nk, 2 * Dc), dev, seed=901 + Nk) * 0.8).bfloat16()
    mask = torch.rand(B, Nk, device=dev) > 0.3
    mask[:, 0] = True
    if masked_tail:
        mask[0, 128:] = False
    k, v = kv.float().chunk(2, -1)
    ref = _attn_ref(q.float(), k, v, mask, None, H, Hkv)
    out, lse = ops.attention_fwd(q, kv, kv, ldq=D, q_off=0, ldk=2 * Dc, k_off=0, ldv=2 * Dc, v_off=Dc, B=B, H=H, Hkv=Hkv,
                                 Nq=Nq, Nk=Nk, key_mask=mask)
    assert rel(out, ref) < 1e-2, rel(out, ref)
    qh = q.float().view(B, Nq, H, 64).transpose(1, 2)
    kh = k.reshape(B, Nk, Hkv, 64).transpose(1, 2).repeat_interleave(grp, dim=1)
    sc = (qh @ kh.transpose(-1, -2)) * 0.125
    sc = sc.masked_fill(~mask[:, None, None, :], -torch.finfo(torch.float32).max)
    assert (lse - torch.logsumexp(sc, dim=-1)).abs().max().item() < 2e-2


def test_sampler_conditioning_hoisted_out_of_the_loop_is_bit_identical(dev, monkeypatch):
    """generate_diffusion_cond projects the (constant) conditioning once per call - cond | uncond halves, to_cond_embed,
    to_global_embed, the cross-attention k | v of all layers - instead of once per sampler step: same arithmetic, so the samples
    must equal the in-loop path bit for bit, with and without negative conditioning, graph replay and eager"""
    from stable_audio_tools.inference.generation import generate_diffusion_cond
    m = _small_dit(dev, seed=72)
    m.eval().requires_grad_(False)
    _, _, _, cond = _batch(dev, 2, 14)
    neg = {"prompt": (cond["prompt"][0].flip(0), cond["prompt"][1]), "g": cond["g"]}
    for graph in ("1", "0"):
        monkeypatch.setenv("KALLE_SAMPLE_GRAPH", graph)
        for extra in ({}, {"negative_conditioning_tensors": neg}):
            kw = dict(steps=5, cfg_scale=4.0, conditioning_tensors=cond, batch_size=2, sample_size=125, seed=11, device="cpu", **extra)
            monkeypatch.setenv("KALLE_SAMPLE_PRECOND", "1")
            a = generate_diffusion_cond(m, **kw).clone()
            monkeypatch.setenv("KALLE_SAMPLE_PRECOND", "0")
            b = generate_diffusion_cond(m, **kw).clone()
            assert torch.equal(a, b), (graph, bool(extra), rel(a, b))
    monkeypatch.setenv("KALLE_SAMPLE_PRECOND", "1")
    c = generate_diffusion_cond(m, **dict(kw, cfg_scale=1.0))               # unguided: no batch doubling
    monkeypatch.setenv("KALLE_SAMPLE_PRECOND", "0")
    assert torch.equal(c, generate_diffusion_cond(m, **dict(kw, cfg_scale=1.0)))


# ------------------------------------------------------------------------------------------------ off-default block options
def _opt_imports():
    import golden_util as gu
    from test_modules_gpu import T, fx, load_seeded
    from test_round2_gpu import check_digests
    from stable_audio_tools.models import transformer as mods
    return gu, T, fx, load_seeded, check_digests, mods


@pytest.mark.parametrize("tag,ada,seed", [("conformer", False, 80), ("conformer_ada", True, 81)])
def test_transformer_block_conformer(dev, tag, ada, seed):
    """TransformerBlock(conformer=True) (transformer.py:550-583, 673-674, 691-692) against the reference's own outputs
    (tests/golden/block_options.npz): block output and input gradients within 1e-2 / 2e-2 (bf16 GEMM operands), every parameter
    gradient by digest, the depthwise taps / norm scales / GLU bias of the ConformerModule in full; and the module on its own"""
    gu, T, fx, load_seeded, check_digests, mods = _opt_imports()
    f = fx("block_options")
    o = gu.OPT_BLOCK
    Do, DCo, No, So, Bo = o["D"], o["DC"], o["N"], o["S"], o["B"]
    x = T(gu.make_input("x", (Bo, No, Do), seed), dev, True)
    ctx = T(gu.make_input("ctx", (Bo, So, DCo), seed), dev, True)
    dy = T(gu.make_input("dy", (Bo, No, Do), seed), dev)
    cmask = (torch.arange(So)[None, :] < torch.tensor([So, So - 9])[:, None]).to(dev)
    blk = load_seeded(mods.TransformerBlock(Do, dim_heads=64, cross_attend=True, dim_context=DCo,
                                            global_cond_dim=Do if ada else None, conformer=True), seed, dev)
    rot = mods.RotaryEmbedding(32).to(dev)
    kw = {}
    if ada:
        gc = T(gu.make_input("g", (Bo, Do), seed), dev, True)
        kw["global_cond"] = gc
    y = blk(x, context=ctx, context_mask=cmask, rotary_pos_emb=rot.forward_from_seq_len(No), **kw)
    y.backward(dy)
    assert rel(y, f[f"{tag}/y"]) < 1e-2, rel(y, f[f"{tag}/y"])
    assert rel(x.grad, f[f"{tag}/dx"]) < 2e-2, rel(x.grad, f[f"{tag}/dx"])
    assert rel(ctx.grad, f[f"{tag}/dctx"]) < 2e-2, rel(ctx.grad, f[f"{tag}/dctx"])
    if ada:
        assert rel(gc.grad, f[f"{tag}/dg"]) < 2e-2, rel(gc.grad, f[f"{tag}/dg"])
    g = {n: p.grad for n, p in blk.named_parameters()}
    assert all(v is not None and v.shape == p.shape for (n, p), v in zip(blk.named_parameters(), g.values()))
    check_digests(f, g, 32, prefix=f"{tag}/")
    for k in f.files:
        if k.startswith(f"{tag}/grad/"):
            name = k[len(tag) + 6:]
            assert rel(g[name], f[k]) < 3e-2, (name, rel(g[name], f[k]))
    xm = T(gu.make_input("xm", (Bo, No, Do), seed), dev, True)
    ym = blk.conformer(xm)
    ym.backward(dy)
    assert rel(ym, f[f"{tag}/module_y"]) < 1e-2, rel(ym, f[f"{tag}/module_y"])
    assert rel(xm.grad, f[f"{tag}/module_dx"]) < 2e-2, rel(xm.grad, f[f"{tag}/module_dx"])


@pytest.mark.parametrize("tag,seed", [("ct_sin", 82), ("ct_abs", 83)])
def test_continuous_transformer_position_embeddings(dev, tag, seed):
    """ContinuousTransformer(use_sinusoidal_emb / use_abs_pos_emb) (transformer.py:45-87, 733-739, 796-797) against the reference:
    output, input gradients, the embedding's own gradient in full (the learned scalar / the rows of the table that were used)"""
    gu, T, fx, load_seeded, check_digests, mods = _opt_imports()
    f = fx("block_options")
    c = gu.OPT_CT
    kw = dict(use_sinusoidal_emb=True) if tag == "ct_sin" else dict(use_abs_pos_emb=True, abs_pos_emb_max_length=c["max_len"])
    ct = load_seeded(mods.ContinuousTransformer(dim=c["D"], depth=c["depth"], dim_in=c["dim_in"], dim_out=c["dim_out"],
                                                dim_heads=64, **kw), seed, dev)
    x = T(gu.make_input("x", (c["B"], c["N"], c["dim_in"]), seed), dev, True)
    pe = T(gu.make_input("prepend", (c["B"], c["P"], c["D"]), seed), dev, True)
    y = ct(x, prepend_embeds=pe)
    y.backward(T(gu.make_input("dy", tuple(y.shape), seed), dev))
    assert rel(y, f[f"{tag}/y"]) < 1e-2, rel(y, f[f"{tag}/y"])
    assert rel(x.grad, f[f"{tag}/dx"]) < 2e-2, rel(x.grad, f[f"{tag}/dx"])
    assert rel(pe.grad, f[f"{tag}/dprepend"]) < 2e-2, rel(pe.grad, f[f"{tag}/dprepend"])
    g = {n: p.grad for n, p in ct.named_parameters()}
    check_digests(f, g, 32, prefix=f"{tag}/")
    for k in f.files:
        if k.startswith(f"{tag}/grad/"):
            name = k[len(tag) + 6:]
            assert rel(g[name], f[k]) < 3e-2, (name, rel(g[name], f[k]))


@pytest.mark.parametrize("S", [40, 48, 130])
def test_transformer_block_causal_vs_oracle(dev, S):
    """TransformerBlock(causal=True): self- and cross-attention under the mask of create_causal_mask (transformer.py:32-33; keys
    c <= r + S - N).  Parity unpinned (the reference's causal calls raise, see Attention): held to the CPU oracle, which applies
    that function's mask - S = N, S > N within one key block, and S = 130 (two key blocks behind 126 queries... here 40)."""
    gu, T, fx, load_seeded, check_digests, mods = _opt_imports()
    Do, DCo, No, Bo, seed = 256, 128, 40, 2, 91
    x = T(gu.make_input("x", (Bo, No, Do), seed), dev, True)
    ctx = T(gu.make_input("ctx", (Bo, S, DCo), seed), dev, True)
    dy = T(gu.make_input("dy", (Bo, No, Do), seed), dev)
    blk = load_seeded(mods.TransformerBlock(Do, dim_heads=64, cross_attend=True, dim_context=DCo, causal=True), seed, dev)
    rot = mods.RotaryEmbedding(32).to(dev)
    y = blk(x, context=ctx, rotary_pos_emb=rot.forward_from_seq_len(No))
    y.backward(dy)
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in gu.make_state(
        [(n, tuple(p.shape)) for n, p in blk.named_parameters()], seed).items()}
    xr, cr = x.detach().cpu().requires_grad_(True), ctx.detach().cpu().requires_grad_(True)
    yr = ko.transformer_block(sd, xr, context=cr, rotary=ko.rotary_freqs(No), causal=True)
    yr.backward(dy.cpu())
    # the mask matters: without it the output differs by far more than the tolerance
    y_nc = ko.transformer_block({k: v.detach() for k, v in sd.items()}, xr.detach(), context=cr.detach(),
                                rotary=ko.rotary_freqs(No), causal=False)
    assert rel(y_nc, yr.detach()) > 5e-2
    assert rel(y, yr.detach()) < 1e-2, rel(y, yr.detach())
    assert rel(x.grad, xr.grad) < 2e-2, rel(x.grad, xr.grad)
    assert rel(ctx.grad, cr.grad) < 2e-2, rel(ctx.grad, cr.grad)
    for n, p in blk.named_parameters():
        assert rel(p.grad, sd[n].grad) < 3e-2, (n, rel(p.grad, sd[n].grad))
    # the stand-alone module, a single query: causal is switched off (transformer.py:468-469)
    att = load_seeded(mods.Attention(Do, dim_heads=64, causal=True), seed + 1, dev)
    x1 = T(gu.make_input("x1", (Bo, 1, Do), seed), dev)
    sa = {k: torch.from_numpy(v) for k, v in gu.make_state([(n, tuple(p.shape)) for n, p in att.named_parameters()], seed + 1).items()}
    assert rel(att(x1), ko.attention(sa, x1.cpu(), causal=True)) < 1e-2


def test_conformer_dit_through_trainer_matches_autograd(dev):
    """a DiT whose blocks carry a ConformerModule and a sinusoidal position embedding: the trainer path (matrix gradients of the
    1 x 1 convolutions written into [N, K, 1] sinks by the grouped launch, the depthwise taps ADDED into a sink that is cleared
    with the vectors) equals plain autograd on the same model over a window of two micro-batches, twice (stale sums would show in the second)"""
    from kalle_audio_amd import engine
    from kalle_audio_amd.stable_audio_tools.training.diffusion import diffusion_train_step
    ref = _small_dit(dev, conformer=True, use_sinusoidal_emb=True)
    lat, noise, t, cond = _batch(dev, 4, 5)
    loss_ref, _ = diffusion_train_step(ref, lat, t, noise, cond, objective="v")
    loss_ref.backward()
    gref = {n: p.grad.clone() for n, p in ref.named_parameters()}
    m = _small_dit(dev, conformer=True, use_sinusoidal_emb=True)
    tr = engine.DataParallelTrainer(m, lr=0.0, optimizer="Adam", grad_accum_steps=2)
    for i in range(2):      # two optimizer steps of two identical micro-batches each: mean of equal gradients = the gradient
        for _ in range(2):
            loss = tr.train_step(m, lat, t, noise, cond, objective="v")
        torch.cuda.synchronize()
        assert abs(loss.item() - loss_ref.item()) < 1e-5 * max(1.0, abs(loss_ref.item()))
        for n, _ in m.named_parameters():      # (the flat buffer holds the SUM over the window; 1 / accum is folded into Adam)
            a, b = tr.flat.grad_view(n), 2.0 * gref[n]
            assert rel(a, b) < 2e-3, (i, n, rel(a, b))


# ------------------------------------------------------------------------------------------------ batch-size sweep
def test_batch_size_sweep_full_width(dev):
    """the bench width (D = 1536, 24 heads, 1024 latent channels, 126 tokens, 130 x 768 context; 2 blocks) at batch sizes that land
    on every row-count regime of the GEMM dispatcher - 126 ... 12600 rows: few-rows K slices, small tiles, 256-wide tiles, the
    persistent kernel, the grouped weight gradients with ragged K - checked through a size-independent property: a batch made of
    B copies of ONE clip has that clip's loss and (as a mean) its gradient.  And the device memory a train step leaves
    allocated does not grow from step to step (round 3: the few-rows scratch leaked a GiB per call at 20-32 clips)."""
    import bench
    from kalle_audio_amd import engine
    cfg = dict(bench.CFG, depth=2)
    model = bench.build_model(dev, cfg=cfg)
    lat1, noise1, t1, cond1 = bench.make_batch(1, dev, 17, cfg)
    tr = engine.DataParallelTrainer(model, lr=0.0, optimizer="Adam")

    def step(B):
        rep = lambda x: x.expand(B, *x.shape[1:]).contiguous()
        cond = {k: (rep(v[0]), None if v[1] is None else rep(v[1])) for k, v in cond1.items()}
        loss = tr.train_step(model, rep(lat1), rep(t1), rep(noise1), cond, objective="v")
        torch.cuda.synchronize()
        return loss.item(), tr.flat.grad.clone()

    l1, g1 = step(1)
    assert torch.isfinite(g1).all() and g1.abs().max() > 0
    for B in (2, 3, 5, 8, 13, 20, 24, 32, 40, 64, 100):
        lb, gb = step(B)
        assert abs(lb - l1) < 2e-3 * abs(l1), (B, lb, l1)
        assert rel(gb, g1) < 2e-2, (B, rel(gb, g1))
        del gb
        torch.cuda.empty_cache()
        m0 = torch.cuda.memory_allocated()
        step(B)
        step(B)
        torch.cuda.empty_cache()
        assert torch.cuda.memory_allocated() <= m0 + (8 << 20), (B, m0, torch.cuda.memory_allocated())


def test_dynamic_tile_handout_is_exact_under_contention_and_across_streams(dev):
    """the persistent 256 x 256 kernel draws its tiles from per-XCD counters (gemm2.hip, kalle_gemm_sched): every tile is computed
    exactly once whatever the order - results are BIT-identical (a) run to run, (b) with a kernel holding 48 CUs while the GEMM
    launches (kalle_debug_hold_cus: workgroups start late and the others take over their tiles), (c) with two persistent GEMMs
    of different shapes in flight on two streams (each launch owns its counter set), and the sets clean themselves (300
    launches reuse every one of the 256 sets)"""
    import ctypes
    from kalle_audio_amd import _lib, ops
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(3)
    mk = lambda r, c: (torch.randn(r, c, generator=g, device=dev) * 0.5).bfloat16()
    a1, b1 = mk(16100, 1536), mk(4608, 1536)         # 63 x 18 = 1134 tiles of 256 x 256 (ragged last row tile)
    a2, b2 = mk(32256, 1536), mk(1536, 1536)         # k-major B: 126 x 6 = 756 tiles (a shape of the train step)
    ref1 = ops.gemm(a1, b1)
    assert (lib.kalle_gemm_last_plan() & 255) == 3, lib.kalle_gemm_last_plan()
    ref2 = ops.gemm(a2, b2, b_kmajor=True)
    assert (lib.kalle_gemm_last_plan() & 255) == 3, lib.kalle_gemm_last_plan()
    torch.cuda.synchronize()
    for _ in range(3):
        assert torch.equal(ops.gemm(a1, b1), ref1)
    side = torch.cuda.Stream()
    for held in (16, 48, 96):
        with torch.cuda.stream(side):
            assert lib.kalle_debug_hold_cus(held, 16384, 1500, ctypes.c_void_p(side.cuda_stream)) == 0
        out = ops.gemm(a1, b1)
        torch.cuda.synchronize()
        assert torch.equal(out, ref1), held
    outs = []
    for i in range(150):
        with torch.cuda.stream(side):
            o2 = ops.gemm(a2, b2, b_kmajor=True)
        o1 = ops.gemm(a1, b1)
        if i % 50 == 49:
            outs.append((o1, o2))
    torch.cuda.synchronize()
    for o1, o2 in outs:
        assert torch.equal(o1, ref1) and torch.equal(o2, ref2)


# ------------------------------------------------------------------------------------------------ configuration fuzz
def _fuzz_cases():
    import random
    rnd = random.Random(20261004)
    cases = []
    for i in range(16):
        heads = rnd.choice([2, 3, 4, 5, 6])
        kv = rnd.choice([k for k in (1, 2, 3, 4, 5) if heads % k == 0])     # (transformer.py:505-508: heads // kv_heads query heads per kv head)
        cases.append(dict(seed=300 + i, D=64 * heads, heads=heads, depth=rnd.choice([1, 2, 3]), cio=rnd.choice([8, 16, 40, 64]),
                          dc=64 * kv, gd=rnd.choice([16, 32, 96]), gtype=rnd.choice(["prepend", "adaLN"]),
                          T=rnd.choice([7, 33, 64, 125, 130, 257]), S=rnd.choice([1, 7, 77, 130, 145, 200]),
                          B=rnd.choice([1, 2, 3, 5]), obj=rnd.choice(["v", "rectified_flow"]), proj=rnd.choice([True, False]),
                          pad=rnd.choice([True, False]), cmask=rnd.choice([True, False])))
    return cases


@pytest.mark.parametrize("c", _fuzz_cases(), ids=lambda c: f"s{c['seed']}-D{c['D']}-L{c['depth']}-T{c['T']}-S{c['S']}-B{c['B']}-{c['gtype']}")
def test_dit_train_step_configuration_fuzz(dev, c):
    """seeded random DiT configurations - widths of 2-5 heads, 1-3 blocks, odd latent / context lengths (7 ... 257 frames, 1 ... 200
    context tokens: ragged K-tiles, one to three key blocks, the folded key tail), batch 1-5, both global-conditioning types, both
    objectives, with / without conditioning projection, padding mask and context mask - one train step (loss, output, EVERY
    parameter gradient) against the CPU oracle on the same weights.  Tolerances of a 1-3 block bf16 path: loss 1e-2, output rel-L2
    2e-2 (cosine 0.999), gradients 4e-2 of the tensor's norm (tiny tensors: of the largest gradient norm in the model)."""
    import golden_util as gu
    from test_modules_gpu import load_seeded
    from stable_audio_tools.models.dit import DiffusionTransformer
    from kalle_audio_amd import functional as KF, ops
    seed = c["seed"]
    dit = load_seeded(DiffusionTransformer(io_channels=c["cio"], embed_dim=c["D"], depth=c["depth"], num_heads=c["heads"],
                                           cond_token_dim=c["dc"], project_cond_tokens=c["proj"], global_cond_dim=c["gd"],
                                           transformer_type="continuous_transformer", global_cond_type=c["gtype"]), seed, dev)
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in gu.make_state(
        [(n, tuple(p.shape)) for n, p in dit.named_parameters()], seed).items()}
    B, T, S = c["B"], c["T"], c["S"]
    lat = torch.from_numpy(gu.make_input("lat", (B, c["cio"], T), seed))
    noise = torch.from_numpy(gu.make_input("noise", (B, c["cio"], T), seed))
    t = torch.linspace(0.08, 0.93, B)
    ctx = torch.from_numpy(gu.make_input("ctx", (B, S, c["dc"]), seed))
    gl = torch.from_numpy(gu.make_input("glob", (B, c["gd"]), seed))
    pm = torch.from_numpy(gu.make_mask("pm", (B, T), seed, 0.8)) if c["pad"] else None
    cm = torch.from_numpy(gu.make_mask("cm", (B, S), seed, 0.7)) if c["cmask"] else None
    ocfg = dict(embed_dim=c["D"], depth=c["depth"], num_heads=c["heads"], global_cond_type=c["gtype"])
    loss_ref, out_ref, _, _ = ko.train_step_loss(sd, ocfg, lat, noise, t, c["obj"], padding_mask=pm, cross_attn_cond=ctx,
                                                 cross_attn_cond_mask=cm, global_embed=gl)
    loss_ref.backward()
    xt, tgt = ops.diffuse_fwd(lat.to(dev), noise.to(dev), t.to(dev), c["obj"])
    out = dit(xt, t.to(dev), cross_attn_cond=ctx.to(dev), cross_attn_cond_mask=None if cm is None else cm.to(dev),
              global_embed=gl.to(dev), cfg_dropout_prob=0.0)
    loss = KF.MSELossFn.apply(out, tgt, None if pm is None else pm.to(dev), 1.0)
    loss.backward()
    assert abs(loss.item() - loss_ref.item()) <= 1e-2 * abs(loss_ref.item()), (loss.item(), loss_ref.item())
    assert cosine(out, out_ref.detach()) > 0.999 and rel(out, out_ref.detach()) < 2e-2, rel(out, out_ref.detach())
    gmax = max(v.grad.norm().item() for v in sd.values() if v.grad is not None)
    for n, p in dit.named_parameters():
        g_ref = sd[n].grad
        if g_ref is None:
            assert p.grad is None or p.grad.abs().max().item() == 0, n
            continue
        assert p.grad is not None, n
        err = (p.grad.float().cpu() - g_ref).norm().item()
        assert err <= 4e-2 * max(g_ref.norm().item(), 2e-2 * gmax), (n, err, g_ref.norm().item(), gmax)


@pytest.mark.parametrize("B,T_,S,accum,gtype", [(1, 7, 1, 1, "prepend"), (3, 64, 77, 1, "adaLN"), (5, 33, 130, 2, "prepend"),
                                               (2, 257, 145, 1, "prepend"), (4, 125, 200, 2, "adaLN"), (7, 130, 7, 3, "prepend")])
def test_trainer_path_fuzz_matches_autograd(dev, B, T_, S, accum, gtype):
    """the trainer's gradient path (flat buckets, grouped weight gradients written in place or - row counts that are not a
    multiple of 8 - one GEMM each, vectors added atomically, accumulation windows of 1-3 micro-batches) against plain autograd on
    the same model and data, at row counts from 8 to 1806 per micro-batch and context lengths that fold / do not fold their tail"""
    from kalle_audio_amd import engine
    from kalle_audio_amd.stable_audio_tools.training.diffusion import diffusion_train_step
    ref = _small_dit(dev, seed=91, depth=2, global_cond_type=gtype)
    m = _small_dit(dev, seed=91, depth=2, global_cond_type=gtype)
    batches = [_batch(dev, B, 40 + i, T_=T_, S=S) for i in range(accum)]
    for lat, noise, t, cond in batches:
        loss_ref, _ = diffusion_train_step(ref, lat, t, noise, cond, objective="v")
        loss_ref.backward()
    gref = {n: p.grad.clone() for n, p in ref.named_parameters()}
    tr = engine.DataParallelTrainer(m, lr=0.0, optimizer="Adam", grad_accum_steps=accum)
    for rep in range(2):            # two windows: whatever the first one left behind must not leak into the second
        for lat, noise, t, cond in batches:
            tr.train_step(m, lat, t, noise, cond, objective="v")
        torch.cuda.synchronize()
        gmax = max(v.norm().item() for v in gref.values())
        for n, _ in m.named_parameters():
            a, b = tr.flat.grad_view(n), gref[n]
            err = (a - b).norm().item()
            assert err <= 5e-3 * max(b.norm().item(), 1e-2 * gmax), (rep, n, err, b.norm().item())


@pytest.mark.parametrize("B,T_,S,cfg_scale,objective,gtype", [(1, 125, 130, "6.0", "v", "prepend"), (3, 33, 1, "1.0", "v", "adaLN"),
                                                             (2, 257, 145, "3.5", "rectified_flow", "prepend"),
                                                             (5, 64, 77, "2.0", "rectified_flow", "adaLN")])
def test_generation_graph_replay_fuzz(dev, monkeypatch, B, T_, S, cfg_scale, objective, gtype):
    """generate_diffusion_cond over batch sizes, lengths (one to three key blocks, folded / separate key tail), guidance on and
    off, both objectives and conditioning types: the HIP-graph replay per sampler step (with the conditioning hoisted out of the
    loop) returns the eager path's samples bit for bit, a second call with another seed reuses the captured graph correctly,
    and the samples are finite"""
    from stable_audio_tools.inference.generation import generate_diffusion_cond
    m = _small_dit(dev, seed=75, global_cond_type=gtype)
    m.diffusion_objective = objective
    m.eval().requires_grad_(False)
    _, _, _, cond = _batch(dev, B, 21, T_=T_, S=S)
    outs = {}
    for graph in ("1", "0"):
        monkeypatch.setenv("KALLE_SAMPLE_GRAPH", graph)
        outs[graph] = [generate_diffusion_cond(m, steps=6, cfg_scale=float(cfg_scale), conditioning_tensors=cond, batch_size=B,
                                               sample_size=T_, seed=sd_, device="cpu").clone() for sd_ in (5, 6, 5)]
    for a, b in zip(outs["1"], outs["0"]):
        assert torch.isfinite(a).all() and torch.equal(a, b), rel(a, b)
    assert torch.equal(outs["1"][0], outs["1"][2]) and not torch.equal(outs["1"][0], outs["1"][1])


def _vae_fuzz_cases():
    import random
    rnd = random.Random(77)
    cases = []
    for i in range(8):
        nlev = rnd.choice([2, 3, 4])
        strides = [rnd.choice([2, 3, 4, 5, 8]) for _ in range(nlev)]
        cases.append(dict(seed=400 + i, ch=rnd.choice([4, 8, 12, 16]), c_mults=[1, 2, 4, 8][:nlev], strides=strides,
                          lat=rnd.choice([2, 4, 6, 8]), snake=rnd.choice([True, False]), audio=rnd.choice([1, 2]),
                          B=rnd.choice([1, 2, 3]), frames=rnd.choice([5, 17, 40, 63])))
    return cases


@pytest.mark.parametrize("c", _vae_fuzz_cases(), ids=lambda c: f"s{c['seed']}-ch{c['ch']}-{'x'.join(map(str, c['strides']))}-B{c['B']}-f{c['frames']}")
def test_oobleck_vae_configuration_fuzz(dev, c):
    """seeded random Oobleck autoencoders - 2-4 levels, strides 2-8 in any order, 4-16 base channels (so widths the conv kernels'
    8- and 16-channel wave tiles do not divide), mono / stereo, SnakeBeta / ELU, 5-63 latent frames, batch 1-3 - encode and decode
    against the CPU oracle on the same weights (fp32 conv path: rel-L2 1e-4)."""
    import golden_util as gu
    from test_modules_gpu import load_seeded
    from stable_audio_tools.models.factory import create_model_from_config
    ratio = 1
    for s_ in c["strides"]:
        ratio *= s_
    cfg = {"model_type": "autoencoder", "sample_rate": 16000, "sample_size": ratio * c["frames"], "audio_channels": c["audio"],
           "model": {"encoder": {"type": "oobleck", "config": {"in_channels": c["audio"], "channels": c["ch"], "c_mults": c["c_mults"],
                                                              "strides": c["strides"], "latent_dim": 2 * c["lat"],
                                                              "use_snake": c["snake"]}},
                     "decoder": {"type": "oobleck", "config": {"out_channels": c["audio"], "channels": c["ch"], "c_mults": c["c_mults"],
                                                              "strides": c["strides"], "latent_dim": c["lat"], "use_snake": c["snake"],
                                                              "final_tanh": c["snake"]}},
                     "bottleneck": {"type": "vae"}, "latent_dim": c["lat"], "downsampling_ratio": ratio, "io_channels": c["audio"]}}
    ae = load_seeded(create_model_from_config(cfg), c["seed"], dev)
    ae.eval().requires_grad_(False)
    sd = {k: torch.from_numpy(v) for k, v in gu.make_state([(n, tuple(p.shape)) for n, p in ae.named_parameters()], c["seed"]).items()}
    wav = torch.from_numpy(gu.make_input("wav", (c["B"], c["audio"], ratio * c["frames"]), c["seed"], 0.5))
    with torch.no_grad():
        z_ref = ko.oobleck_encoder(ko._sub(sd, "encoder."), wav, c["strides"], c["snake"])
        z = ae.encode(wav.to(dev))
        assert z.shape == z_ref.shape and rel(z, z_ref) < 1e-4, (tuple(z.shape), tuple(z_ref.shape), rel(z, z_ref))
        zl = z_ref[:, :c["lat"]].contiguous()
        rec_ref = ko.oobleck_decoder(ko._sub(sd, "decoder."), zl, c["strides"], c["snake"], final_tanh=c["snake"])
        rec = ae.decode(zl.to(dev))
        assert rec.shape == rec_ref.shape and rel(rec, rec_ref) < 1e-4, (tuple(rec.shape), rel(rec, rec_ref))


def test_fused_bias_column_sums_and_second_consumers(dev, monkeypatch):
    """trainer mode: the LayerNorm backward that produces a block's output gradient also adds its column sums into that block's
    FF-out bias sink (kalle_layernorm_bwd_colsum).  (a) the flat gradient equals the un-fused path's; (b) when a block's output
    has a SECOND consumer (hidden states in the loss) the gradient that reaches the block is another tensor - the fused
    contribution is taken out again and the bias gradient is still that of plain autograd"""
    import golden_util as gu
    from test_modules_gpu import load_seeded
    from kalle_audio_amd import engine
    from stable_audio_tools.models import transformer as mods
    x = torch.from_numpy(gu.make_input("x", (3, 40, 16), 95)).to(dev)
    ctx = torch.from_numpy(gu.make_input("ctx", (3, 9, 64), 95)).to(dev)

    def build():
        return load_seeded(mods.ContinuousTransformer(dim=128, depth=3, dim_in=16, dim_out=16, dim_heads=64, cross_attend=True,
                                                      cond_token_dim=64), 95, dev)

    def loss_of(ct, second):
        out, info = ct(x, context=ctx, return_info=True)
        loss = out.square().mean()
        if second:
            loss = loss + 0.3 * info["hidden_states"][0].square().mean() + 0.2 * info["hidden_states"][1].mean()
        return loss

    for second in (False, True):
        ref = build()
        loss_of(ref, second).backward()
        gref = {n: p.grad.clone() for n, p in ref.named_parameters()}
        for fuse in ("1", "0"):
            monkeypatch.setenv("KALLE_FUSE_BIAS_COLSUM", fuse)
            ct = build()
            tr = engine.DataParallelTrainer(ct, lr=0.0, optimizer="Adam")
            for rep in range(2):
                tr.backward(loss_of(ct, second))
                torch.cuda.synchronize()
                gmax = max(v.norm().item() for v in gref.values())
                for n, _ in ct.named_parameters():
                    a, b = tr.flat.grad_view(n), gref[n]
                    err = (a - b).norm().item()
                    assert err <= 5e-3 * max(b.norm().item(), 1e-2 * gmax), (second, fuse, rep, n, err, b.norm().item())


def _llasa_fuzz_cases():
    import random
    rnd = random.Random(4242)
    cases = []
    for i in range(6):
        H = rnd.choice([2, 3, 4, 6])
        Hkv = rnd.choice([k for k in (1, 2, 3) if H % k == 0])
        cases.append(dict(seed=500 + i, H=H, Hkv=Hkv, layers=rnd.choice([1, 2, 3]), inter=rnd.choice([192, 256, 448]),
                          lat=rnd.choice([8, 16, 32]), B=rnd.choice([1, 2, 4]), L=rnd.choice([24, 64, 130, 257]),
                          scaling=rnd.choice([True, False])))
    return cases


@pytest.mark.parametrize("c", _llasa_fuzz_cases(), ids=lambda c: f"s{c['seed']}-H{c['H']}k{c['Hkv']}-L{c['layers']}-B{c['B']}-n{c['L']}")
def test_llasa_configuration_fuzz(dev, tmp_path, c):
    """model_sigmaVAE.Llasa over seeded random Llama shapes (2-6 heads over 1-3 kv heads, 1-3 layers, odd MLP widths, with and
    without llama3 rope scaling) and ragged batches of 24-257 positions: both losses, the predicted means at the valid positions
    and EVERY parameter gradient against the CPU oracle on the same weights"""
    import json
    import golden_util as gu
    from test_modules_gpu import _Tok, _hf_grads
    from kalle_audio_amd.model_sigmaVAE import Llasa
    hid = 64 * c["H"]
    llama = dict(vocab_size=300, hidden_size=hid, intermediate_size=c["inter"], num_hidden_layers=c["layers"],
                 num_attention_heads=c["H"], num_key_value_heads=c["Hkv"], head_dim=64, rms_norm_eps=1e-5, rope_theta=500000.0,
                 max_position_embeddings=1024, tie_word_embeddings=True, attention_bias=False, mlp_bias=False, hidden_act="silu")
    if c["scaling"]:
        llama["rope_scaling"] = dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0,
                                     original_max_position_embeddings=64)
    lc = dict(latent_dim=c["lat"], tokenizer_len=310, llama=llama)
    d = tmp_path / "llama_fuzz"
    d.mkdir(exist_ok=True)
    (d / "config.json").write_text(json.dumps(dict(llama, model_type="llama")))
    m = Llasa({"llm_model_name_or_path": str(d), "latent_dim": c["lat"], "audio_proj_dim": hid}, _Tok(310), use_flash_attention=False)
    lat, I, H, Hkv = c["lat"], c["inter"], c["H"], c["Hkv"]
    names = [("audio_linear.bias", (hid,)), ("audio_linear.weight", (hid, lat)), ("base_model.model.embed_tokens.weight", (310, hid)),
             ("base_model.model.norm.weight", (hid,)), ("distribution_linear.0.bias", (lat,)), ("distribution_linear.0.weight", (lat, hid)),
             ("distribution_linear.2.bias", (lat,)), ("distribution_linear.2.weight", (lat, lat))]
    for i in range(c["layers"]):         # the reference's (HF) names: the drop-in fuses q | k | v and gate | up at load time
        pre = f"base_model.model.layers.{i}."
        names += [(pre + "input_layernorm.weight", (hid,)), (pre + "post_attention_layernorm.weight", (hid,)),
                  (pre + "mlp.down_proj.weight", (hid, I)), (pre + "mlp.gate_proj.weight", (I, hid)), (pre + "mlp.up_proj.weight", (I, hid)),
                  (pre + "self_attn.q_proj.weight", (H * 64, hid)), (pre + "self_attn.k_proj.weight", (Hkv * 64, hid)),
                  (pre + "self_attn.v_proj.weight", (Hkv * 64, hid)), (pre + "self_attn.o_proj.weight", (hid, H * 64))]
    st = gu.make_state(names, c["seed"])
    sd_gpu = {k: torch.from_numpy(v) for k, v in st.items()}
    sd_gpu["base_model.lm_head.weight"] = sd_gpu["base_model.model.embed_tokens.weight"]
    m.load_state_dict(sd_gpu)
    m = m.to(dev)
    sd = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in st.items()}
    bnp = gu.llasa_batch_long(lc, c["seed"], B=c["B"], L=c["L"])
    eps_np = gu.make_input("llasa_eps", tuple(bnp["audio_latents"].shape), c["seed"])
    ref = ko.llasa_forward(sd, lc, {k: torch.from_numpy(v) for k, v in bnp.items()}, torch.from_numpy(eps_np))
    (ref["audio_loss"] * 1.0 + ref["end_loss"] * 0.5).backward()
    b = {k: torch.from_numpy(v).to(dev) for k, v in bnp.items()}
    out = m(b["input_ids"], b["audio_latents"], b["audio_distribution_l"], b["ids_mask"], b["audio_mask"], b["target_mask"],
            b["end_mask"], noise=torch.from_numpy(eps_np).to(dev))
    for k in ("audio_loss", "end_loss"):
        assert abs(out[k].item() - ref[k].item()) <= 1e-2 * abs(ref[k].item()), (k, out[k].item(), ref[k].item())
    valid = ((b["ids_mask"] + b["audio_mask"]) > 0).cpu()
    assert rel(out["pre_mean"].float().cpu()[valid], ref["pre_mean"].detach()[valid]) < 1.5e-2
    (out["audio_loss"] * 1.0 + out["end_loss"] * 0.5).backward()
    g = _hf_grads(m)
    gmax = max(v.grad.norm().item() for v in sd.values() if v.grad is not None)
    for n, v in sd.items():
        if v.grad is None:
            continue
        err = (g[n].float().cpu() - v.grad).norm().item()
        assert err <= 4e-2 * max(v.grad.norm().item(), 2e-2 * gmax), (n, err, v.grad.norm().item())


def _melvae_fuzz_cases():
    import random
    rnd = random.Random(99)
    cases = []
    for i in range(6):
        n = rnd.choice([2, 3])
        up = [rnd.choice([2, 3, 4, 5]) for _ in range(n)]
        rk = [rnd.choice([3, 5, 7]) for _ in range(2)]
        r1 = rnd.choice([True, False])
        cases.append(dict(seed=600 + i, h=dict(
            latent_dim=rnd.choice([4, 8, 12]), use_vae=True, downsample_channels=[12, 16, 24, 32][:n + 1],
            downsample_rates=list(reversed(up)), upsample_rates=up, upsample_kernel_sizes=[2 * u for u in up],
            upsample_initial_channel=rnd.choice([24, 32, 48]), resblock="1" if r1 else "2", resblock_kernel_sizes=rk,
            resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5]] if r1 else [[1, 3], [1, 2]],
            activation=rnd.choice(["snakebeta", "snake"]), snake_logscale=rnd.choice([True, False]),
            causal=rnd.choice([True, False]), flow_hidden_channels=rnd.choice([8, 16])),
            frames=rnd.choice([9, 40, 173]), B=rnd.choice([1, 2, 3])))
    return cases


@pytest.mark.parametrize("c", _melvae_fuzz_cases(), ids=lambda c: f"s{c['seed']}-up{'x'.join(map(str, c['h']['upsample_rates']))}-rb{c['h']['resblock']}-f{c['frames']}")
def test_melvae_configuration_fuzz(dev, c):
    """backup/flows.py BigVGANFlowVAE over seeded random hyper-parameters (2-3 levels, up / down rates 2-5, kernel sizes 3-7, AMP
    block 1 / 2, snake / snakebeta, log-scale or not, causal or 'same' padding, odd channel widths) and lengths that fit no tile:
    encoder and decoder against the CPU oracle on the same weights (fp32 conv path: 2e-4; the anti-aliased activation is the
    published algorithm on both sides - parity of that piece is unpinned)"""
    import golden_util as gu
    from test_modules_gpu import load_seeded
    from kalle_audio_amd.flows import BigVGANFlowVAE
    h = c["h"]
    vae = load_seeded(BigVGANFlowVAE(h), c["seed"], dev)
    sd = {k: v.detach().cpu() for k, v in vae.state_dict().items()}
    ratio = 1
    for u in h["upsample_rates"]:
        ratio *= u
    z = torch.from_numpy(gu.make_input("zz", (c["B"], h["latent_dim"], c["frames"]), c["seed"]))
    want = ko.melvae_decode(sd, z, h)
    with torch.no_grad():
        got = vae.inference_from_latents(z.to(dev), do_sample=False)
    # ('same' padding with an odd rate: (k - u) // 2 leaves one sample more per level than frames x rate - as the reference's)
    assert got.shape == want.shape and abs(got.shape[-1] - c["frames"] * ratio) <= 2 * ratio, (tuple(got.shape), tuple(want.shape))
    assert rel(got, want) < 2e-4, rel(got, want)
    wav = torch.from_numpy(gu.make_input("ww", (c["B"], 1, c["frames"] * ratio + 3), c["seed"], 0.5))
    want = ko.melvae_encoder(ko._sub(sd, "audio_encoder."), wav, h["downsample_rates"])
    with torch.no_grad():
        got = vae.extract_latents(wav.to(dev))
    assert got.shape == want.shape and rel(got, want) < 1e-4, (tuple(got.shape), tuple(want.shape), rel(got, want))


def _conv_wgrad_cases():
    import random
    rnd = random.Random(31)
    cases = [dict(kind="conv", K=7, s=1, d=9, Cin=128, Cout=128, L=3000, B=2, act=1), dict(kind="conv", K=1, s=1, d=1, Cin=96, Cout=160, L=777, B=3, act=2),
             dict(kind="conv", K=4, s=2, d=1, Cin=40, Cout=72, L=1001, B=2, act=1), dict(kind="conv", K=8, s=4, d=1, Cin=64, Cout=130, L=2049, B=1, act=0),
             dict(kind="conv", K=16, s=8, d=1, Cin=48, Cout=96, L=1283, B=2, act=1), dict(kind="convT", K=16, s=8, d=1, Cin=96, Cout=40, L=97, B=2, act=1),
             dict(kind="convT", K=8, s=4, d=1, Cin=130, Cout=64, L=211, B=1, act=2), dict(kind="convT", K=4, s=2, d=1, Cin=32, Cout=32, L=500, B=3, act=0),
             dict(kind="conv", K=7, s=1, d=3, Cin=2, Cout=64, L=900, B=2, act=0), dict(kind="conv", K=5, s=1, d=2, Cin=64, Cout=64, L=300, B=2, act=1)]
    for i in range(4):
        cases.append(dict(kind="conv", K=7, s=1, d=rnd.choice([1, 3, 9]), Cin=rnd.choice([17, 64, 200]), Cout=rnd.choice([24, 65, 128]),
                          L=rnd.choice([63, 450, 5000]), B=rnd.choice([1, 2]), act=rnd.choice([0, 1, 2])))
    return cases


@pytest.mark.parametrize("c", _conv_wgrad_cases(), ids=lambda c: f"{c['kind']}-K{c['K']}s{c['s']}d{c['d']}-{c['Cin']}x{c['Cout']}-L{c['L']}-a{c['act']}")
def test_conv_weight_gradient_kernels_vs_torch(dev, c):
    """kalle_conv_wgrad (round 3: LDS-staged kernel, lane = V channel, U as scalar operands; the per-lane-load kernel remains for
    fewer than 16 V channels and odd tap counts): dW of Conv1d and ConvTranspose1d with the input activation (SnakeBeta / ELU /
    none) against torch autograd in fp64 - strides 1-8, dilations 1-9, channel counts that do not fill the 64-lane / 64-channel
    tiles, lengths that end inside a position tile"""
    import torch.nn.functional as F
    from kalle_audio_amd import conv_ops
    from kalle_audio_amd.conv_train import conv_wgrad
    g = torch.Generator().manual_seed(7)
    K, s_, d, Cin, Cout, L, B, act = c["K"], c["s"], c["d"], c["Cin"], c["Cout"], c["L"], c["B"], c["act"]
    x = torch.randn(B, Cin, L, generator=g, dtype=torch.float64)
    alpha = (torch.randn(Cin, generator=g, dtype=torch.float64) * 0.3) if act == 1 else None
    beta = (torch.randn(Cin, generator=g, dtype=torch.float64) * 0.3) if act == 1 else None

    def actf(t):
        if act == 1:
            a, b_ = alpha.exp()[None, :, None], beta.exp()[None, :, None]
            return t + torch.sin(t * a) ** 2 / (b_ + 1e-9)
        return F.elu(t) if act == 2 else t

    if c["kind"] == "conv":
        pad = (K - 1) * d // 2 if s_ == 1 else (s_ + 1) // 2
        w = torch.randn(Cout, Cin, K, generator=g, dtype=torch.float64, requires_grad=True)
        y = F.conv1d(actf(x), w, stride=s_, padding=pad, dilation=d)
    else:
        pad = (s_ + 1) // 2
        w = torch.randn(Cin, Cout, K, generator=g, dtype=torch.float64, requires_grad=True)
        y = F.conv_transpose1d(actf(x), w, stride=s_, padding=pad)
    gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(gy)
    xf, gyf = x.float().to(dev), gy.float().to(dev)
    a32 = alpha.float().to(dev) if alpha is not None else None
    b32 = beta.float().to(dev) if beta is not None else None
    dw = torch.zeros(w.shape, device=dev)
    if c["kind"] == "conv":
        conv_wgrad(gyf, xf, dw, K=K, stride=s_, padding=pad, dilation=d, act_on=0, act=act, alpha=a32, beta=b32, logscale=True)
    else:
        conv_wgrad(conv_ops.activate(xf, act, a32, b32, True), gyf, dw, K=K, stride=s_, padding=pad, dilation=1, act_on=1, act=0)
    assert rel(dw, w.grad.float()) < 2e-5, rel(dw, w.grad.float())
    if c["kind"] == "convT" and act:      # the in-kernel activation of the scalar-side operand (per-lane-load kernel) still agrees
        dw2 = torch.zeros(w.shape, device=dev)
        conv_wgrad(xf, gyf, dw2, K=K, stride=s_, padding=pad, dilation=1, act_on=1, act=act, alpha=a32, beta=b32, logscale=True)
        assert rel(dw2, w.grad.float()) < 2e-5
