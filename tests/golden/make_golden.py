"""Generate the golden fixtures in tests/golden/*.npz by running the REFERENCE implementation
(/root/reference, imported in place, CPU fp32).  Runs only in the build container; the fixtures it writes are data
(inputs / outputs / gradients), never reference source.  Usage:  python tests/golden/make_golden.py

Third-party modules the reference imports at module import time but that are absent here are stubbed with empty
modules; the only stub that carries arithmetic is dac.nn.layers.WNConv1d/WNConvTranspose1d =
torch.nn.utils.weight_norm(nn.Conv1d/ConvTranspose1d), which is exactly what descript-audio-codec defines.
"""
import math
import os
import sys
import types

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import golden_util as gu  # noqa: E402

REF = "/root/reference"


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Missing(nn.Module):
        def __init__(self, *a, **k):
            raise RuntimeError("third-party module not available in this container")

    mod("x_transformers", ContinuousTransformerWrapper=_Missing, Encoder=_Missing)
    mod("dac")
    mod("dac.nn")

    def WNConv1d(*a, **k):
        return torch.nn.utils.weight_norm(nn.Conv1d(*a, **k))

    def WNConvTranspose1d(*a, **k):
        return torch.nn.utils.weight_norm(nn.ConvTranspose1d(*a, **k))

    mod("dac.nn.layers", Snake1d=_Missing, WNConv1d=WNConv1d, WNConvTranspose1d=WNConvTranspose1d)
    mod("dac.nn.quantize", ResidualVectorQuantize=_Missing)
    mod("vector_quantize_pytorch", ResidualVQ=_Missing, FSQ=_Missing)
    # alias-free-torch is absent: Activation1d below is OUR restatement (oracle/kalle_oracle.py activation1d, parity
    # unpinned) wearing the package's module/attribute names, so that the reference's BigVGANFlowVAE can be run and the
    # wiring around it pinned.
    sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
    import kalle_oracle as ko

    class _Resample(nn.Module):
        def __init__(self, nested):
            super().__init__()
            filt = ko.kaiser_sinc_filter1d(0.25, 0.3, 12).view(1, 1, -1)
            if nested:
                self.lowpass = _Resample(False)
            else:
                self.register_buffer("filter", filt)

    class Activation1d(nn.Module):
        def __init__(self, activation, up_ratio=2, down_ratio=2, up_kernel_size=12, down_kernel_size=12):
            super().__init__()
            self.act = activation
            self.upsample = _Resample(False)
            self.downsample = _Resample(True)

        def forward(self, x):
            f = self.upsample.filter.view(-1)
            return ko.downsample1d_2x(self.act(ko.upsample1d_2x(x, f)), f)

    mod("alias_free_torch", Activation1d=Activation1d)
    ta = mod("torchaudio")
    ta.transforms = mod("torchaudio.transforms")
    mod("k_diffusion")
    mod("einops_exts", rearrange_many=None)


def load_seeded(module, seed):
    shapes = [(n, tuple(p.shape)) for n, p in module.named_parameters()]
    st = gu.make_state(shapes, seed)
    with torch.no_grad():
        for n, p in module.named_parameters():
            p.copy_(torch.from_numpy(st[n]))
    return module


def T(a):
    return torch.from_numpy(np.asarray(a))


def grads(module):
    return {n: p.grad.detach().numpy() for n, p in module.named_parameters() if p.grad is not None}


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path)/1024:.1f} KiB", flush=True)


def pack_grads(prefix, g, full=False):
    d = {}
    for n, a in g.items():
        d[f"{prefix}digest/{n}"] = gu.digest(a)
        if full:
            d[f"{prefix}grad/{n}"] = a
    return d


def main():
    install_stubs()
    sys.path.insert(0, REF)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    from stable_audio_tools.models import transformer as rt
    from stable_audio_tools.models.dit import DiffusionTransformer
    from stable_audio_tools.models.blocks import FourierFeatures, SnakeBeta, RMSNorm
    from stable_audio_tools.training.losses.losses import MSELoss, MultiLoss
    from stable_audio_tools.inference import sampling as rs
    from stable_audio_tools.models import autoencoders as ra
    from stable_audio_tools.models.factory import create_model_from_config
    from stable_audio_tools.models.pretransforms import AutoencoderPretransform

    B, N, D, S, DC = 2, 125, 128, 7, 64

    # ---- LayerNorm / RMSNorm / SnakeBeta / FourierFeatures ---------------------------------------------
    x = T(gu.make_input("x", (B, N, D), 1, 1.5)).requires_grad_(True)
    dy = T(gu.make_input("dy", (B, N, D), 1))
    ln = load_seeded(rt.LayerNorm(D), 1)
    y = ln(x)
    y.backward(dy)
    save("layernorm", y=y, dx=x.grad, **pack_grads("", grads(ln), full=True))

    x = T(gu.make_input("x", (B, N, D), 2, 1.5)).requires_grad_(True)
    rn = load_seeded(RMSNorm((D,)), 2)
    y = rn(x)
    y.backward(dy)
    save("rmsnorm", y=y, dx=x.grad, **pack_grads("", grads(rn), full=True))

    xs = T(gu.make_input("x", (B, 8, 100), 3, 2.0))
    sn = load_seeded(SnakeBeta(8), 3)
    save("snake_beta", y=sn(xs))

    t = T(np.linspace(0.05, 0.95, 6).astype(np.float32))
    ff = load_seeded(FourierFeatures(1, 256), 4)
    save("fourier_features", y=ff(t[:, None]))

    # ---- Attention (self: rotary + mask; cross: kv_heads != heads + context mask) ------------------------------------
    rot = rt.RotaryEmbedding(32)
    x = T(gu.make_input("x", (B, N, D), 5)).requires_grad_(True)
    mask = T(gu.make_mask("m", (B, N), 5))
    at = load_seeded(rt.Attention(D, dim_heads=64), 5)
    y = at(x, mask=mask, rotary_pos_emb=rot.forward_from_seq_len(N))
    y.backward(dy)
    save("attention_self", y=y, dx=x.grad, **pack_grads("", grads(at), full=True))

    x = T(gu.make_input("x", (B, N, D), 6)).requires_grad_(True)
    ctx = T(gu.make_input("ctx", (B, S, DC), 6)).requires_grad_(True)
    cmask = T(gu.make_mask("cm", (B, S), 6))
    at = load_seeded(rt.Attention(D, dim_heads=64, dim_context=DC), 6)
    y = at(x, context=ctx, context_mask=cmask)
    y.backward(dy)
    save("attention_cross", y=y, dx=x.grad, dctx=ctx.grad, **pack_grads("", grads(at), full=True))

    # ---- FeedForward ---------------------------------------------------------------------------------------------
    x = T(gu.make_input("x", (B, N, D), 7)).requires_grad_(True)
    fw = load_seeded(rt.FeedForward(D), 7)
    y = fw(x)
    y.backward(dy)
    save("feedforward", y=y, dx=x.grad, **pack_grads("", grads(fw)))

    # ---- TransformerBlock plain / adaLN ----------------------------------------------------------------------------
    for name, gdim, seed in (("block_plain", None, 8), ("block_adaln", D, 9)):
        x = T(gu.make_input("x", (B, N, D), seed)).requires_grad_(True)
        ctx = T(gu.make_input("ctx", (B, S, DC), seed)).requires_grad_(True)
        blk = load_seeded(rt.TransformerBlock(D, dim_heads=64, cross_attend=True, dim_context=DC, global_cond_dim=gdim),
                          seed)
        kw = {}
        extra = {}
        if gdim:
            gc = T(gu.make_input("g", (B, D), seed)).requires_grad_(True)
            kw["global_cond"] = gc
        y = blk(x, context=ctx, rotary_pos_emb=rot.forward_from_seq_len(N), **kw)
        y.backward(dy)
        if gdim:
            extra["dg"] = gc.grad
        save(name, y=y, dx=x.grad, dctx=ctx.grad, **extra, **pack_grads("", grads(blk), full=(name == "block_plain")))

    # ---- ContinuousTransformer with prepend + masks -----------------------------------------------------------------
    CIO = 16
    x = T(gu.make_input("x", (B, N, CIO), 10)).requires_grad_(True)
    pre = T(gu.make_input("pre", (B, 1, D), 10)).requires_grad_(True)
    ctx = T(gu.make_input("ctx", (B, S, DC), 10)).requires_grad_(True)
    ct = load_seeded(rt.ContinuousTransformer(dim=D, depth=2, dim_in=CIO, dim_out=CIO, dim_heads=64, cross_attend=True,
                                               cond_token_dim=DC), 10)
    dyo = T(gu.make_input("dyo", (B, N + 1, CIO), 10))
    y = ct(x, prepend_embeds=pre, prepend_mask=torch.ones(B, 1, dtype=torch.bool), context=ctx)
    y.backward(dyo)
    save("continuous_transformer", y=y, dx=x.grad, dpre=pre.grad, dctx=ctx.grad, **pack_grads("", grads(ct)))

    # ---- DiffusionTransformer (prepend / adaLN), train-step pieces, CFG inference --------------------------------------
    G = 32
    for gtype, seed in (("prepend", 11), ("adaLN", 12)):
        dit = load_seeded(DiffusionTransformer(io_channels=CIO, embed_dim=D, depth=2, num_heads=2, cond_token_dim=DC,
                                               project_cond_tokens=False, global_cond_dim=G,
                                               transformer_type="continuous_transformer", global_cond_type=gtype), seed)
        lat = T(gu.make_input("lat", (B, CIO, N), seed))
        noise = T(gu.make_input("noise", (B, CIO, N), seed))
        tt = T(np.array([0.3, 0.85], dtype=np.float32))
        ctx = T(gu.make_input("ctx", (B, S, DC), seed))
        cmask = T(gu.make_mask("cm", (B, S), seed))
        gl = T(gu.make_input("glob", (B, G), seed))
        pmask = T(gu.make_mask("pm", (B, N), seed, 0.7))
        out = {}
        for obj in ("v", "rectified_flow"):
            # training/diffusion.py:365-379 composed from importable pieces (lightning wrapper itself is not importable)
            if obj == "v":
                al, si = rs.get_alphas_sigmas(tt)
            else:
                al, si = 1 - tt, tt
            al, si = al[:, None, None], si[:, None, None]
            xt = lat * al + noise * si
            tgt = noise * al - lat * si if obj == "v" else noise - lat
            dit.zero_grad()
            o = dit(xt, tt, cross_attn_cond=ctx, cross_attn_cond_mask=cmask, global_embed=gl, cfg_dropout_prob=0.0)
            for mk, pm in (("", None), ("_masked", pmask)):
                ml = MultiLoss([MSELoss("output", "targets", weight=1.0, mask_key="padding_mask", name="mse_loss")])
                loss, _ = ml({"output": o, "targets": tgt, "padding_mask": pm})
                out[f"{obj}{mk}/loss"] = loss.detach()
            loss, _ = MultiLoss([MSELoss("output", "targets", weight=1.0, name="mse_loss")])({"output": o, "targets": tgt})
            loss.backward()
            out[f"{obj}/x_t"] = xt
            out[f"{obj}/target"] = tgt
            out[f"{obj}/output"] = o
            out.update(pack_grads(f"{obj}/", grads(dit)))
        with torch.no_grad():
            o_cfg = dit(xt, tt, cross_attn_cond=ctx, global_embed=gl, cfg_scale=3.0, scale_phi=0.5)
            o_neg = dit(xt, tt, cross_attn_cond=ctx, global_embed=gl, cfg_scale=2.0,
                        negative_cross_attn_cond=ctx.flip(0), negative_cross_attn_mask=cmask)
        out["cfg3_phi05/output"] = o_cfg
        out["cfg2_neg/output"] = o_neg
        # samplers (inference/sampling.py:24-86)
        if gtype == "prepend":
            x0 = T(gu.make_input("x0", (B, CIO, N), seed))
            fn = lambda x_, t_, **k: dit(x_, t_, cross_attn_cond=ctx, global_embed=gl, cfg_scale=3.0)
            with torch.no_grad():
                out["sample_ddim4"] = rs.sample(fn, x0, 4, 0.0)
                out["sample_euler4"] = rs.sample_discrete_euler(fn, x0, 4)
        save(f"dit_{gtype}", **out)

    # ---- Oobleck VAE pieces ---------------------------------------------------------------------------------------------
    for snake in (True, False):
        tag = "snake" if snake else "elu"
        xx = T(gu.make_input("x", (B, 16, 200), 20, 1.0))
        ru = load_seeded(ra.ResidualUnit(16, 16, dilation=3, use_snake=snake), 20)
        eb = load_seeded(ra.EncoderBlock(16, 32, stride=4, use_snake=snake), 21)
        db = load_seeded(ra.DecoderBlock(32, 16, stride=4, use_snake=snake), 22)
        with torch.no_grad():
            y_ru = ru(xx)
            y_eb = eb(xx)
            y_db = db(y_eb)
        save(f"oobleck_units_{tag}", y_ru=y_ru, y_eb=y_eb, y_db=y_db)

        cfg = {
            "model_type": "autoencoder", "sample_rate": 16000, "sample_size": 4096, "audio_channels": 2,
            "model": {
                "encoder": {"type": "oobleck", "config": {"in_channels": 2, "channels": 8, "c_mults": [1, 2, 4],
                                                         "strides": [2, 4, 5], "latent_dim": 8, "use_snake": snake}},
                "decoder": {"type": "oobleck", "config": {"out_channels": 2, "channels": 8, "c_mults": [1, 2, 4],
                                                         "strides": [2, 4, 5], "latent_dim": 4, "use_snake": snake,
                                                         "final_tanh": snake}},
                "bottleneck": {"type": "vae"},
                "latent_dim": 4, "downsampling_ratio": 40, "io_channels": 2,
            },
        }
        ae = load_seeded(create_model_from_config(cfg), 23)
        pt = AutoencoderPretransform(ae, scale=0.8)
        wav = T(gu.make_input("wav", (B, 2, 1200), 23, 0.5))
        with torch.no_grad():
            z = pt.encode(wav)              # mean || scale (bottleneck is a pass-through in this reference)
            rec = pt.decode(z[:, :4])
        save(f"oobleck_vae_{tag}", z=z, rec=rec)
    # ---- mel-VAE (backup/flows.py) ---------------------------------------------------------------------------------------
    sys.path.insert(0, os.path.join(REF, "backup"))
    import flows as rf

    class AttrDict(dict):
        __getattr__ = dict.__getitem__

    melvae_inv = {}
    for tag, h in gu.MELVAE_CONFIGS.items():
        h = AttrDict(h)
        vae = load_seeded(rf.BigVGANFlowVAE(h), 30)
        melvae_inv[tag] = {k: list(v.shape) for k, v in vae.state_dict().items()}
        wav = T(gu.make_input("melwav", (B, 1, 256), 30, 0.5))
        eps = T(gu.make_input("meleps", (B, h.latent_dim, 256 // int(np.prod(h.downsample_rates))), 30))
        orig = torch.randn_like
        torch.randn_like = lambda t, **k: eps
        try:
            with torch.no_grad():
                enc = vae.extract_latents(wav)
                rec, (z_p, logs_q, _, _) = vae(wav)
                rec_sampled = vae.inference_from_latents(enc, do_sample=True)
                rec_mean = vae.inference_from_latents(enc[:, :h.latent_dim], do_sample=False)
                rs_in = T(gu.make_input("rs", (B, 16, 64), 31))
                rs_out = vae.audio_encoder.generator[3](rs_in)
        finally:
            torch.randn_like = orig
        assert torch.equal(rec, rec_sampled)
        save(f"melvae_{tag}", enc=enc, rec=rec, z_p=z_p, logs_q=logs_q, rec_mean=rec_mean, rs_out=rs_out)

    # ---- Llasa task model (model_sigmaVAE.py) over a tiny locally-built Llama (third-party transformers arithmetic) -------
    import tempfile
    for stub in ("torchaudio", "torchaudio.transforms"):   # the empty stubs above confuse transformers' availability probes
        sys.modules.pop(stub, None)
    from transformers import LlamaConfig, LlamaForCausalLM
    import model_sigmaVAE as rl
    lc = gu.LLASA_CONFIG
    tmp = tempfile.mkdtemp(prefix="kalle_llama_")
    hf_cfg = LlamaConfig(**lc["llama"])
    LlamaForCausalLM(hf_cfg).save_pretrained(tmp)

    class _Tok:
        def __len__(self):
            return lc["tokenizer_len"]

    llasa = rl.Llasa({"llm_model_name_or_path": tmp, "latent_dim": lc["latent_dim"], "audio_proj_dim": lc["llama"]["hidden_size"]},
                     _Tok(), use_flash_attention=False)
    load_seeded(llasa, 40)
    llasa_inv = {k: list(v.shape) for k, v in llasa.state_dict().items()}
    batch = gu.llasa_batch(lc, 40)
    tb = {k: T(v) for k, v in batch.items()}
    eps = T(gu.make_input("llasa_eps", batch["audio_latents"].shape, 40))
    orig = torch.randn_like
    torch.randn_like = lambda t, **k: eps
    try:
        out = llasa(tb["input_ids"], tb["audio_latents"], tb["audio_distribution_l"], tb["ids_mask"], tb["audio_mask"],
                    tb["target_mask"], tb["end_mask"])
    finally:
        torch.randn_like = orig
    (out["audio_loss"] * 1.0 + out["end_loss"] * 0.5).backward()
    g = grads(llasa)
    save("llasa", audio_loss=out["audio_loss"], end_loss=out["end_loss"], pre_mean=out["pre_mean"],
         sampled=out["ground_truth_audio_latents"],
         **pack_grads("", g), **{f"grad/{k}": g[k] for k in ("audio_linear.weight", "distribution_linear.2.bias",
                                                               "base_model.model.layers.1.self_attn.k_proj.weight",
                                                               "base_model.model.norm.weight")})

    # ---- state-dict key/shape inventories (the drop-in contract) ---------------------------------
    import json
    inv = {}
    for gtype in ("prepend", "adaLN"):
        dit = DiffusionTransformer(io_channels=CIO, embed_dim=D, depth=2, num_heads=2, cond_token_dim=DC,
                                   project_cond_tokens=True, global_cond_dim=G, prepend_cond_dim=24,
                                   transformer_type="continuous_transformer", global_cond_type=gtype)
        inv[f"dit_{gtype}"] = {k: list(v.shape) for k, v in dit.state_dict().items()}
    cfg["model"]["encoder"]["config"]["use_snake"] = True
    cfg["model"]["decoder"]["config"]["use_snake"] = True
    inv["oobleck_autoencoder"] = {k: list(v.shape) for k, v in create_model_from_config(cfg).state_dict().items()}
    for tag, v in melvae_inv.items():
        inv[f"melvae_{tag}"] = v
    inv["llasa"] = llasa_inv
    with open(os.path.join(HERE, "state_dict_keys.json"), "w") as f:
        json.dump(inv, f, indent=0, sort_keys=True)
    print("done")


if __name__ == "__main__":
    main()
