"""Round-2 golden fixtures: the REFERENCE implementation (/root/reference, imported in place, CPU fp32) run on

  * one TransformerBlock at the BENCH width (D = 1536, 24 heads, 12 kv heads, 126 tokens, 130 context tokens), plain + adaLN
  * a DiT over 375 latent frames (+1 prepended token) and 130 context tokens (30 s clips, configs/twj_0828.yaml)
  * Llasa (model_sigmaVAE.py) at 4 heads / 2 kv heads over ragged sequences of 300
  * model.py's Llasa (two-Gaussian KL; its label transform is injected, see golden_util.default_mean_stdev)
  * DiffusionCondTrainingWrapper.training_step itself (training/diffusion.py:311-437), both objectives / samplers
  * generate_diffusion_cond (inference/generation.py:90-250): seed -> noise -> CFG sampler -> pretransform.decode, and the
    int16 export of infer_0723.py:292-293

Runs only in the build container.  Writes data only (inputs are re-derived from seeds, weights from parameter names).
Usage: python tests/golden/make_golden_r02.py [name ...]
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import golden_util as gu  # noqa: E402
import make_golden as mg  # noqa: E402
from make_golden import T, grads, load_seeded, save  # noqa: E402

REF = mg.REF


def install_training_stubs():
    """import-time names of stable_audio_tools/training/diffusion.py:1-25 that are absent here.  None carries arithmetic:
    LightningModule is nn.Module + the three members training_step touches (device, log_dict, trainer)."""
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Missing(nn.Module):
        def __init__(self, *a, **k):
            raise RuntimeError("third-party module not available in this container")

    class LightningModule(nn.Module):
        @property
        def device(self):
            return next(self.parameters()).device

        def log_dict(self, *a, **k):
            pass

    mod("pytorch_lightning", LightningModule=LightningModule, Callback=type("Callback", (), {}))
    mod("pytorch_lightning.utilities")
    mod("pytorch_lightning.utilities.rank_zero", rank_zero_only=lambda f: f)
    mod("wandb")
    mod("aeiou")
    mod("aeiou.viz", pca_point_cloud=None, audio_spectrogram_image=None, tokens_spectrogram_image=None)
    mod("auraloss")
    mod("ema_pytorch", EMA=_Missing)
    mod("audiotools", AudioSignal=None, STFTParams=None)
    mod("dac.model")
    mod("dac.model.discriminator", WNConv1d=_Missing, WNConv2d=_Missing)


def digests(prefix, g, n):
    return {f"{prefix}digest/{k}": gu.digest(v, n) for k, v in g.items()}


def f16(t):
    return t.detach().numpy().astype(np.float16)


# ------------------------------------------------------------------------------------------------------------------------
def wide_blocks():
    from stable_audio_tools.models import transformer as rt
    w = gu.WIDE_BLOCK
    D, DC, N, S, B = w["D"], w["DC"], w["N"], w["S"], w["B"]
    rot = rt.RotaryEmbedding(32)
    for name, gdim, seed in (("block_wide_plain", None, 50), ("block_wide_adaln", D, 51)):
        x = T(gu.make_input("x", (B, N, D), seed)).requires_grad_(True)
        ctx = T(gu.make_input("ctx", (B, S, DC), seed)).requires_grad_(True)
        dy = T(gu.make_input("dy", (B, N, D), seed))
        blk = load_seeded(rt.TransformerBlock(D, dim_heads=64, cross_attend=True, dim_context=DC, global_cond_dim=gdim), seed)
        kw, extra = {}, {}
        if gdim:
            gc = T(gu.make_input("g", (B, D), seed)).requires_grad_(True)
            kw["global_cond"] = gc
        y = blk(x, context=ctx, rotary_pos_emb=rot.forward_from_seq_len(N), **kw)
        y.backward(dy)
        if gdim:
            extra["dg"] = f16(gc.grad)
        # fp16 storage: the bf16 path is compared at 1e-2, fp16 rounding is 5e-4
        save(name, y=f16(y), dx=f16(x.grad), dctx=f16(ctx.grad), **extra, **digests("", grads(blk), 64))


def block_qk_norm():
    """TransformerBlock(attn_kwargs={"qk_norm": "l2" | "ln"}) (transformer.py:303-307, 422-428): q and k normalised per head
    before the rotary embedding, self-attention (4 heads) and GQA cross-attention (4 over 2 kv heads, ragged context mask);
    "ln" on an adaLN block.  Outputs, input gradients, parameter-gradient digests, the LayerNorm(64) gradients in full."""
    from stable_audio_tools.models import transformer as rt
    q = gu.QK_NORM_BLOCK
    D, DC, N, S, B = q["D"], q["DC"], q["N"], q["S"], q["B"]
    rot = rt.RotaryEmbedding(32)
    out = {}
    for kind, gdim, seed in (("l2", None, 70), ("ln", D, 71)):
        x = T(gu.make_input("x", (B, N, D), seed)).requires_grad_(True)
        ctx = T(gu.make_input("ctx", (B, S, DC), seed)).requires_grad_(True)
        dy = T(gu.make_input("dy", (B, N, D), seed))
        cmask = torch.arange(S)[None, :] < torch.tensor([S, S - 7])[:, None]
        blk = load_seeded(rt.TransformerBlock(D, dim_heads=64, cross_attend=True, dim_context=DC, global_cond_dim=gdim,
                                              attn_kwargs={"qk_norm": kind}), seed)
        kw = {}
        if gdim:
            gc = T(gu.make_input("g", (B, D), seed)).requires_grad_(True)
            kw["global_cond"] = gc
        y = blk(x, context=ctx, context_mask=cmask, rotary_pos_emb=rot.forward_from_seq_len(N), **kw)
        y.backward(dy)
        g = grads(blk)
        out[f"{kind}/y"], out[f"{kind}/dx"], out[f"{kind}/dctx"] = y, x.grad, ctx.grad
        if gdim:
            out[f"{kind}/dg"] = gc.grad
        out.update(digests(f"{kind}/", g, 32))
        for k in g:
            if "_norm." in k and "attn" in k:
                out[f"{kind}/grad/{k}"] = g[k]
    save("block_qk_norm", **out)


def dit_long():
    from stable_audio_tools.models.dit import DiffusionTransformer
    from stable_audio_tools.training.losses.losses import MSELoss, MultiLoss
    from stable_audio_tools.inference import sampling as rs
    B, N, D, S, DC, CIO, G, seed = 2, 375, 128, 130, 64, 16, 32, 52
    dit = load_seeded(DiffusionTransformer(io_channels=CIO, embed_dim=D, depth=2, num_heads=2, cond_token_dim=DC,
                                           project_cond_tokens=False, global_cond_dim=G,
                                           transformer_type="continuous_transformer", global_cond_type="prepend"), seed)
    lat = T(gu.make_input("lat", (B, CIO, N), seed))
    noise = T(gu.make_input("noise", (B, CIO, N), seed))
    tt = T(np.array([0.2, 0.65], dtype=np.float32))
    ctx = T(gu.make_input("ctx", (B, S, DC), seed))
    gl = T(gu.make_input("glob", (B, G), seed))
    pmask = T(gu.make_mask("pm", (B, N), seed, 0.7))
    al, si = rs.get_alphas_sigmas(tt)
    al, si = al[:, None, None], si[:, None, None]
    xt = lat * al + noise * si
    tgt = noise * al - lat * si
    out = {}
    o = dit(xt, tt, cross_attn_cond=ctx, global_embed=gl, cfg_dropout_prob=0.0)
    loss_m, _ = MultiLoss([MSELoss("output", "targets", weight=1.0, mask_key="padding_mask", name="mse_loss")])(
        {"output": o, "targets": tgt, "padding_mask": pmask})
    loss, _ = MultiLoss([MSELoss("output", "targets", weight=1.0, name="mse_loss")])({"output": o, "targets": tgt})
    loss.backward()
    out.update(output=o, loss=loss.detach(), loss_masked=loss_m.detach(), **digests("", grads(dit), 16))
    with torch.no_grad():
        # key mask through the transformer mask path (dit.py:189-193 builds it from prepend_cond_mask + mask)
        out["output_cfg"] = dit(xt, tt, cross_attn_cond=ctx, global_embed=gl, cfg_scale=2.5)
    save("dit_long", **out)


def _tiny_llama(lc):
    for stub in ("torchaudio", "torchaudio.transforms", "wandb"):   # empty stubs confuse transformers' / accelerate's probes
        sys.modules.pop(stub, None)
    from transformers import LlamaConfig, LlamaForCausalLM
    tmp = tempfile.mkdtemp(prefix="kalle_llama_")
    LlamaForCausalLM(LlamaConfig(**lc["llama"])).save_pretrained(tmp)

    class _Tok:
        def __len__(self):
            return lc["tokenizer_len"]

    return tmp, _Tok()


def llasa_wide():
    lc = gu.LLASA_WIDE_CONFIG
    tmp, tok = _tiny_llama(lc)
    import model_sigmaVAE as rl
    llasa = rl.Llasa({"llm_model_name_or_path": tmp, "latent_dim": lc["latent_dim"],
                      "audio_proj_dim": lc["llama"]["hidden_size"]}, tok, use_flash_attention=False)
    load_seeded(llasa, 53)
    batch = gu.llasa_batch_long(lc, 53)
    tb = {k: T(v) for k, v in batch.items()}
    eps = T(gu.make_input("llasa_eps", batch["audio_latents"].shape, 53))
    orig = torch.randn_like
    torch.randn_like = lambda t, **k: eps
    try:
        out = llasa(tb["input_ids"], tb["audio_latents"], tb["audio_distribution_l"], tb["ids_mask"], tb["audio_mask"],
                    tb["target_mask"], tb["end_mask"])
    finally:
        torch.randn_like = orig
    (out["audio_loss"] * 1.0 + out["end_loss"] * 0.5).backward()
    g = grads(llasa)
    save("llasa_wide", audio_loss=out["audio_loss"], end_loss=out["end_loss"], pre_mean=f16(out["pre_mean"]),
         **digests("", g, 16), **{f"grad/{k}": g[k] for k in ("audio_linear.weight", "base_model.model.norm.weight")})


def model_llasa():
    """model.py:9-107 (the Stable-Audio-VAE variant train.py:24 imports).  Its two imports that do not resolve here:
    ecapa_tdnn (lives in backup/, never used by the class) and twj_utils (dangling symlink; its one function is injected)."""
    sys.path.insert(0, os.path.join(REF, "backup"))
    tw = types.ModuleType("twj_utils")
    tw.get_mean_stdev_from_stableaudio2_latents = gu.default_mean_stdev
    sys.modules["twj_utils"] = tw
    sys.modules.pop("model", None)
    lc = gu.LLASA_CONFIG
    tmp, tok = _tiny_llama(lc)
    import model as rm
    llasa = rm.Llasa({"llm_model_name_or_path": tmp, "latent_dim": lc["latent_dim"],
                      "audio_proj_dim": lc["llama"]["hidden_size"]}, tok, use_flash_attention=False)
    load_seeded(llasa, 54)
    inv = {k: list(v.shape) for k, v in llasa.state_dict().items()}
    batch = gu.llasa_batch_long(lc, 54, B=3, L=48, label_mult=2)
    tb = {k: T(v) for k, v in batch.items()}
    out = llasa(tb["input_ids"], tb["audio_latents"], tb["audio_distribution_l"], tb["ids_mask"], tb["audio_mask"],
                tb["target_mask"], tb["end_mask"])
    (out["audio_loss"] * 1.0 + out["end_loss"] * 0.5).backward()
    g = grads(llasa)
    save("model_llasa", audio_loss=out["audio_loss"], end_loss=out["end_loss"], pre_mean=out["pre_mean"],
         pre_log_scale=out["pre_log_scale"], **digests("", g, 16),
         **{f"grad/{k}": g[k] for k in ("audio_linear.weight", "distribution_linear.2.bias", "distribution_linear.0.weight",
                                        "base_model.model.norm.weight")})
    import json
    with open(os.path.join(HERE, "state_dict_keys_r02.json"), "w") as f:
        json.dump({"model_llasa": inv}, f, indent=0, sort_keys=True)


class TensorConditioner(nn.Module):
    """the caller's conditioner: metadata already holds the conditioning tensors (the frozen text encoders are out of scope)"""

    def forward(self, metadata, device):
        ctx = torch.stack([md["prompt"] for md in metadata]).to(device)
        cm = torch.stack([md["prompt_mask"] for md in metadata]).to(device)
        gl = torch.stack([md["g"] for md in metadata]).to(device)
        return {"prompt": (ctx, cm), "g": (gl, None)}


def _cond_model(io_channels, objective, seed, with_pretransform=True):
    from stable_audio_tools.models import diffusion as rd
    from stable_audio_tools.models.factory import create_model_from_config
    from stable_audio_tools.models.pretransforms import AutoencoderPretransform
    e = gu.E2E
    dit = rd.DiTWrapper(io_channels=io_channels, embed_dim=e["D"], depth=2, num_heads=2, cond_token_dim=e["DC"],
                        project_cond_tokens=False, global_cond_dim=e["G"], transformer_type="continuous_transformer",
                        global_cond_type="prepend")
    load_seeded(dit, seed)                       # (overwrites the x0.5 init: values are what the test loads too)
    pt = None
    if with_pretransform:
        ae = load_seeded(create_model_from_config(gu.oobleck_cfg(True)), 23)
        pt = AutoencoderPretransform(ae, scale=0.8)
    return rd.ConditionedDiffusionModelWrapper(dit, TensorConditioner(), io_channels=io_channels, sample_rate=16000,
                                               min_input_length=40, diffusion_objective=objective, pretransform=pt,
                                               cross_attn_cond_ids=["prompt"], global_cond_ids=["g"])


def _e2e_cond(seed):
    e = gu.E2E
    ctx = T(gu.make_input("ctx", (e["B"], e["S"], e["DC"]), seed))
    cm = T(gu.make_mask("cm", (e["B"], e["S"]), seed))
    gl = T(gu.make_input("glob", (e["B"], e["G"]), seed))
    return ctx, cm, gl


def generate_e2e():
    from stable_audio_tools.inference import generation as rg
    from stable_audio_tools.inference import sampling as rs
    from einops import rearrange
    e = gu.E2E
    ctx, cm, gl = _e2e_cond(60)
    cond = {"prompt": (ctx, cm), "g": (gl, None)}
    neg = {"prompt": (ctx.flip(0), cm.flip(0)), "g": (gl, None)}
    out = {}

    def export(a):  # infer_0723.py:292-293
        o = rearrange(a, "b d n -> d (b n)")
        return o.to(torch.float32).div(torch.max(torch.abs(o))).clamp(-1, 1).mul(32767).to(torch.int16)

    model = _cond_model(4, "rectified_flow", 60)
    with torch.no_grad():
        audio = rg.generate_diffusion_cond(model, steps=e["steps"], cfg_scale=e["cfg_scale"], conditioning_tensors=cond,
                                           batch_size=e["B"], sample_size=40 * e["T"], seed=e["seed"], device="cpu")
        lat = rg.generate_diffusion_cond(model, steps=e["steps"], cfg_scale=e["cfg_scale"], conditioning_tensors=cond,
                                         batch_size=e["B"], sample_size=40 * e["T"], seed=e["seed"], device="cpu",
                                         return_latents=True)
        audio_neg = rg.generate_diffusion_cond(model, steps=e["steps"], cfg_scale=e["cfg_scale"], conditioning_tensors=cond,
                                               negative_conditioning_tensors=neg, batch_size=e["B"],
                                               sample_size=40 * e["T"], seed=e["seed"], device="cpu")
    out.update({"rf/audio": audio, "rf/latents": lat, "rf/int16": export(audio), "rf_neg/audio": audio_neg})
    # v objective: the reference routes it to k-diffusion (absent); its in-tree v sampler is sampling.py:47-86 (the call the
    # reference keeps commented at generation.py:236), composed here with the same seed / decode convention
    model = _cond_model(4, "v", 61)
    with torch.no_grad():
        torch.manual_seed(e["seed"])
        noise = torch.randn([e["B"], 4, e["T"]])
        ci = model.get_conditioning_inputs(cond)
        lat = rs.sample(model.model, noise, e["steps"], 0.0, **ci, cfg_scale=e["cfg_scale"], batch_cfg=True)
        audio = model.pretransform.decode(lat)
    out.update({"v/audio": audio, "v/latents": lat, "v/int16": export(audio)})
    save("generate_e2e", **out)


def training_step():
    """the reference class itself, one step per (objective, timestep sampler); conditioner / pretransform.encode / Sobol
    draw / noising / model / masked MSE all inside it"""
    install_training_stubs()
    from stable_audio_tools.training import diffusion as rtd
    e = gu.E2E
    B = e["B"]
    out = {}
    for tag, objective, sampler, pre_encoded, seed in (("v_uniform", "v", "uniform", False, 62),
                                                       ("rf_logit", "rectified_flow", "logit_normal", False, 63),
                                                       ("v_pre", "v", "uniform", True, 64)):
        model = _cond_model(8, objective, seed)
        torch.manual_seed(1000 + seed)                       # the scrambled Sobol engine seeds itself from the global RNG
        wrap = rtd.DiffusionCondTrainingWrapper(model, lr=1e-4, mask_padding=True, mask_padding_dropout=0.0, use_ema=False,
                                                pre_encoded=pre_encoded, cfg_dropout_prob=0.0, timestep_sampler=sampler)
        wrap.trainer = types.SimpleNamespace(optimizers=[types.SimpleNamespace(param_groups=[{"lr": 1e-4}])])
        ctx, cm, gl = _e2e_cond(seed)
        L = 40 * e["T"]
        if pre_encoded:
            reals = T(gu.make_input("lat8", (B, 8, e["T"]), seed))
            pm = T(gu.make_mask("pm", (B, e["T"]), seed, 0.75))
        else:
            reals = T(gu.make_input("wav", (B, 2, L), seed, 0.5))
            pm = T(gu.make_mask("pm", (B, e["T"]), seed, 0.75)).repeat_interleave(40, dim=1)
        meta = [{"prompt": ctx[b], "prompt_mask": cm[b], "g": gl[b], "padding_mask": [pm[b]]} for b in range(B)]
        rec = {}
        real_randn, real_randn_like = torch.randn, torch.randn_like

        def randn(*a, **k):
            r = real_randn(*a, **k)
            rec.setdefault("randn", []).append(r.clone())
            return r

        def randn_like(t, **k):
            r = real_randn_like(t, **k)
            rec["noise"] = r.clone()
            return r

        real_draw = wrap.rng.draw

        def draw(n):
            r = real_draw(n)
            rec["sobol"] = r.clone()
            return r

        wrap.rng.draw = draw
        torch.randn, torch.randn_like = randn, randn_like
        try:
            model.zero_grad()
            loss = wrap.training_step((reals, meta), 0)
        finally:
            torch.randn, torch.randn_like = real_randn, real_randn_like
        loss.backward()
        out[f"{tag}/loss"] = loss.detach()
        out[f"{tag}/noise"] = rec["noise"]
        if sampler == "uniform":
            out[f"{tag}/t"] = rec["sobol"][:, 0]
        else:
            out[f"{tag}/t_logit"] = rec["randn"][0]
        out.update(digests(f"{tag}/", {k: v for k, v in grads(model.model).items()}, 16))
    save("training_step", **out)


def chunked_vae():
    """AudioAutoencoder.decode_audio(chunked=True) of the reference (autoencoders.py:499-560): an anchored last chunk, and an
    odd overlap (kept regions of neighbours then overlap by one latent and the later chunk wins).  The reference's chunked
    ENCODE cannot run on this model: its paste buffer has latent_dim channels (autoencoders.py:472) while the pass-through VAE
    bottleneck of this reference returns 2 * latent_dim (bottleneck.py:89-100) - the assignment at :496 raises."""
    from stable_audio_tools.models.factory import create_model_from_config
    ae = load_seeded(create_model_from_config(gu.oobleck_cfg(True)), 23)
    z = T(gu.make_input("zc", (2, 4, 125), 65))
    out = {}
    with torch.no_grad():
        out["dec_48_16"] = ae.decode_audio(z, chunked=True, chunk_size=48, overlap=16)
        out["dec_48_15"] = ae.decode_audio(z, chunked=True, chunk_size=48, overlap=15)
        out["dec_full"] = ae.decode_audio(z, chunked=False)
        wav = T(gu.make_input("wavc", (2, 2, 5000), 65, 0.5))
        try:
            ae.encode_audio(wav, chunked=True, chunk_size=48, overlap=16)
            out["enc_chunked_runs"] = np.array(1)
        except RuntimeError:
            out["enc_chunked_runs"] = np.array(0)
        out["enc_full"] = ae.encode_audio(wav, chunked=False)
    save("chunked_vae", **out)


def vae_backward():
    """torch autograd of the reference's Oobleck pieces (what `enable_grad` pretransforms train with, models/factory.py:77-80):
    ResidualUnit (dilated k=7 + k=1, residual), EncoderBlock (strided conv), DecoderBlock (transposed conv), and the whole
    encode -> decode chain with the final tanh, SnakeBeta and ELU variants - input gradients in full, parameter gradients
    (weight_g, weight_v, bias, alpha, beta) as 16-sample digests + a few in full"""
    from stable_audio_tools.models import autoencoders as ra
    from stable_audio_tools.models.factory import create_model_from_config
    out = {}
    B = 2
    for snake in (True, False):
        tag = "snake" if snake else "elu"
        units = (("ru", ra.ResidualUnit(16, 16, dilation=3, use_snake=snake), 20, (B, 16, 200)),
                 ("eb", ra.EncoderBlock(16, 32, stride=4, use_snake=snake), 21, (B, 16, 203)),
                 ("db", ra.DecoderBlock(32, 16, stride=4, use_snake=snake), 22, (B, 32, 50)))
        for name, mod, seed, shp in units:
            load_seeded(mod, seed)
            x = T(gu.make_input("x", shp, seed + 100, 1.0)).requires_grad_(True)
            y = mod(x)
            dy = T(gu.make_input("dy", tuple(y.shape), seed + 100))
            y.backward(dy)
            g = grads(mod)
            out[f"{tag}/{name}/y"] = y
            out[f"{tag}/{name}/dx"] = x.grad
            out.update(digests(f"{tag}/{name}/", g, 16))
            for k in list(g)[:3]:
                out[f"{tag}/{name}/grad/{k}"] = g[k]
        ae = load_seeded(create_model_from_config(gu.oobleck_cfg(snake)), 23)
        ae.requires_grad_(True)
        wav = T(gu.make_input("wav", (B, 2, 1200), 66, 0.5)).requires_grad_(True)
        z = ae.encode(wav)
        rec = ae.decode(z[:, :4] + 0.3 * z[:, 4:])
        dz = T(gu.make_input("dz", tuple(z.shape), 66))
        drec = T(gu.make_input("drec", tuple(rec.shape), 66))
        ((z * dz).sum() + (rec * drec).sum()).backward()
        g = grads(ae)
        out[f"{tag}/ae/z"] = z
        out[f"{tag}/ae/rec"] = rec
        out[f"{tag}/ae/dwav"] = wav.grad
        out.update(digests(f"{tag}/ae/", g, 16))
    save("vae_backward", **out)


def vae_nearest():
    """DecoderBlock / OobleckDecoder with use_nearest_upsample=True (autoencoders.py:87-96: nn.Upsample(nearest) + a stride-1
    WNConv1d of kernel 2*stride, no bias, padding='same' - an even kernel, so torch pads stride-1 left and stride right):
    forward, input gradient, parameter-gradient digests; then the whole decoder forward + input gradient"""
    from stable_audio_tools.models import autoencoders as ra
    out = {}
    B = 2
    for snake in (True, False):
        tag = "snake" if snake else "elu"
        mod = load_seeded(ra.DecoderBlock(32, 16, stride=4, use_snake=snake, use_nearest_upsample=True), 31)
        x = T(gu.make_input("x", (B, 32, 50), 131, 1.0)).requires_grad_(True)
        y = mod(x)
        dy = T(gu.make_input("dy", tuple(y.shape), 131))
        y.backward(dy)
        g = grads(mod)
        out[f"{tag}/db/y"] = y
        out[f"{tag}/db/dx"] = x.grad
        out.update(digests(f"{tag}/db/", g, 16))
        for k in list(g)[:3]:
            out[f"{tag}/db/grad/{k}"] = g[k]
        dec = load_seeded(ra.OobleckDecoder(out_channels=2, channels=8, latent_dim=4, c_mults=[1, 2, 4], strides=[2, 4, 5],
                                            use_snake=snake, use_nearest_upsample=True, final_tanh=snake), 32)
        z = T(gu.make_input("z", (B, 4, 37), 132, 1.0)).requires_grad_(True)
        w = dec(z)
        dw = T(gu.make_input("dw", tuple(w.shape), 132))
        w.backward(dw)
        out[f"{tag}/dec/y"] = w
        out[f"{tag}/dec/dz"] = z.grad
        out.update(digests(f"{tag}/dec/", grads(dec), 16))
    save("vae_nearest", **out)


ALL = dict(block_qk_norm=block_qk_norm, vae_nearest=vae_nearest, vae_backward=vae_backward, chunked_vae=chunked_vae, wide_blocks=wide_blocks, dit_long=dit_long, llasa_wide=llasa_wide, generate_e2e=generate_e2e,
           training_step=training_step, model_llasa=model_llasa)


def main():
    mg.install_stubs()
    sys.path.insert(0, REF)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    names = sys.argv[1:] or list(ALL)
    # (the Llama fixtures drop the torchaudio stubs, which the stable_audio_tools imports need: run those last)
    for n in sorted(names, key=lambda n: n in ("llasa_wide", "model_llasa")):
        ALL[n]()
    print("done")


if __name__ == "__main__":
    main()
