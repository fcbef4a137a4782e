"""Round-3 golden fixtures: the REFERENCE implementation (/root/reference, imported in place, CPU fp32) run on

  * generate_diffusion_cond(init_audio=..., init_noise_level=...) for a rectified-flow model - the "variation" branch
    (inference/generation.py:164-183, 226-228 -> sampling.py:200-232): prepare_audio (PadCrop, channel fix-up), the optional
    pretransform.encode, `x = init (1 - sigma_max) + noise sigma_max`, discrete Euler from sigma_max;
  * TransformerBlock(conformer=True) (transformer.py:550-583, 673-674, 691-692), plain and adaLN, and
    ContinuousTransformer(use_sinusoidal_emb=True / use_abs_pos_emb=True) (transformer.py:45-87, 733-739, 796-797) - the
    off-default options of the DiT's transformer.  (`causal=True` cannot be pinned: every causal call of the reference's CPU
    branch raises AttributeError - transformer.py:521 calls `self.create_causal_mask`, which is the module-level function of
    line 32, not a method.)
  * the same call with mask_args (generation.py:186-224): the reference cuts / pastes the init audio and builds a soft mask,
    but its rectified-flow branch hands neither the mask nor a sigma_max to sample_rf - the result is plain sampling from the
    seed's noise.  Pinned as it is.

The VAE of case (b) has an encoder of latent_dim 4: this reference's VAE bottleneck is a pass-through (bottleneck.py:89-100),
so an encoder emitting mean | scale (2 x latent) cannot feed `init_data` of a latent_dim-channel DiT (the reference raises on
the shape); with 4 encoder channels the reference runs.

Runs only in the build container.  Writes data only.  Usage: python tests/golden/make_golden_r03.py
"""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import golden_util as gu  # noqa: E402
import make_golden as mg  # noqa: E402
import make_golden_r02 as m2  # noqa: E402
from make_golden import T, load_seeded, save  # noqa: E402

REF = mg.REF


def init_vae_cfg():
    c = gu.oobleck_cfg(True)
    c["model"]["encoder"]["config"]["latent_dim"] = 4
    return c


def _model(with_pretransform, seed):
    from stable_audio_tools.models import diffusion as rd
    from stable_audio_tools.models.factory import create_model_from_config
    from stable_audio_tools.models.pretransforms import AutoencoderPretransform
    e = gu.E2E
    dit = rd.DiTWrapper(io_channels=4, embed_dim=e["D"], depth=2, num_heads=2, cond_token_dim=e["DC"],
                        project_cond_tokens=False, global_cond_dim=e["G"], transformer_type="continuous_transformer",
                        global_cond_type="prepend")
    load_seeded(dit, seed)
    pt = None
    if with_pretransform:
        ae = load_seeded(create_model_from_config(init_vae_cfg()), 24)
        pt = AutoencoderPretransform(ae, scale=0.8)
    return rd.ConditionedDiffusionModelWrapper(dit, m2.TensorConditioner(), io_channels=4, sample_rate=16000,
                                               min_input_length=40, diffusion_objective="rectified_flow", pretransform=pt,
                                               cross_attn_cond_ids=["prompt"], global_cond_ids=["g"])


def generate_init_audio():
    from stable_audio_tools.inference import generation as rg
    e = gu.E2E
    ctx, cm, gl = m2._e2e_cond(64)
    cond = {"prompt": (ctx, cm), "g": (gl, None)}
    out = {}
    common = dict(steps=e["steps"], cfg_scale=e["cfg_scale"], conditioning_tensors=cond, batch_size=e["B"], seed=e["seed"],
                  device="cpu")
    with torch.no_grad():
        # (a) no pretransform: the init "audio" lives in the DiT's own 4-channel space; 100 of 125 frames given (PadCrop pads)
        model = _model(False, 64)
        init = T(gu.make_input("init_lat", (4, 100), 64))
        out["lat/variation"] = rg.generate_diffusion_cond(model, sample_size=e["T"], init_audio=(16000, init),
                                                          init_noise_level=0.6, **common)
        out["lat/plain"] = rg.generate_diffusion_cond(model, sample_size=e["T"], **common)
        margs = dict(cropfrom=10.0, pastefrom=20.0, pasteto=70.0, maskstart=20.0, maskend=70.0, softnessL=5.0, softnessR=8.0,
                     marination=0.1)
        out["lat/masked"] = rg.generate_diffusion_cond(model, sample_size=e["T"], init_audio=(16000, init), init_noise_level=0.6,
                                                       mask_args=margs, **common)
        # (b) latent diffusion: mono init audio, longer than the target (PadCrop crops, set_audio_channels repeats to stereo)
        model = _model(True, 65)
        wav = T(gu.make_input("init_wav", (1, 40 * e["T"] + 333), 65)) * 0.3
        out["vae/variation_latents"] = rg.generate_diffusion_cond(model, sample_size=40 * e["T"], init_audio=(16000, wav),
                                                                  init_noise_level=0.45, return_latents=True, **common)
        out["vae/variation_audio"] = rg.generate_diffusion_cond(model, sample_size=40 * e["T"], init_audio=(16000, wav),
                                                                init_noise_level=0.45, **common)
    save("generate_init_audio", **out)


def block_options():
    from stable_audio_tools.models import transformer as rt
    o = gu.OPT_BLOCK
    D, DC, N, S, B = o["D"], o["DC"], o["N"], o["S"], o["B"]
    rot = rt.RotaryEmbedding(32)
    out = {}
    for tag, gdim, seed in (("conformer", None, 80), ("conformer_ada", D, 81)):
        x = T(gu.make_input("x", (B, N, D), seed)).requires_grad_(True)
        ctx = T(gu.make_input("ctx", (B, S, DC), seed)).requires_grad_(True)
        dy = T(gu.make_input("dy", (B, N, D), seed))
        cmask = torch.arange(S)[None, :] < torch.tensor([S, S - 9])[:, None]
        blk = load_seeded(rt.TransformerBlock(D, dim_heads=64, cross_attend=True, dim_context=DC, global_cond_dim=gdim,
                                              conformer=True), seed)
        kw = {}
        if gdim:
            gc = T(gu.make_input("g", (B, D), seed)).requires_grad_(True)
            kw["global_cond"] = gc
        y = blk(x, context=ctx, context_mask=cmask, rotary_pos_emb=rot.forward_from_seq_len(N), **kw)
        y.backward(dy)
        g = mg.grads(blk)
        out[f"{tag}/y"], out[f"{tag}/dx"], out[f"{tag}/dctx"] = y, x.grad, ctx.grad
        if gdim:
            out[f"{tag}/dg"] = gc.grad
        out.update(m2.digests(f"{tag}/", g, 32))
        for k in g:
            if k.startswith("conformer.") and g[k].size <= 8192:     # vectors and the depthwise taps in full; matrices as digests
                out[f"{tag}/grad/{k}"] = g[k]
        # the module on its own (no residual): its output and input gradient
        xm = T(gu.make_input("xm", (B, N, D), seed)).requires_grad_(True)
        blk.zero_grad()
        ym = blk.conformer(xm)
        ym.backward(dy)
        out[f"{tag}/module_y"], out[f"{tag}/module_dx"] = ym, xm.grad
    c = gu.OPT_CT
    for tag, kw, seed in (("ct_sin", dict(use_sinusoidal_emb=True), 82),
                          ("ct_abs", dict(use_abs_pos_emb=True, abs_pos_emb_max_length=c["max_len"]), 83)):
        ct = load_seeded(rt.ContinuousTransformer(dim=c["D"], depth=c["depth"], dim_in=c["dim_in"], dim_out=c["dim_out"],
                                                  dim_heads=64, **kw), seed)
        x = T(gu.make_input("x", (c["B"], c["N"], c["dim_in"]), seed)).requires_grad_(True)
        pe = T(gu.make_input("prepend", (c["B"], c["P"], c["D"]), seed)).requires_grad_(True)
        y = ct(x, prepend_embeds=pe)
        dy = T(gu.make_input("dy", tuple(y.shape), seed))
        y.backward(dy)
        g = mg.grads(ct)
        out[f"{tag}/y"], out[f"{tag}/dx"], out[f"{tag}/dprepend"] = y, x.grad, pe.grad
        out.update(m2.digests(f"{tag}/", g, 32))
        for k in g:
            if k.startswith("pos_emb."):
                out[f"{tag}/grad/{k}"] = g[k]
    save("block_options", **out)


def main():
    mg.install_stubs()
    sys.path.insert(0, REF)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    want = sys.argv[1:]
    for fn in (generate_init_audio, block_options):
        if not want or fn.__name__ in want:
            fn()
    print("done")


if __name__ == "__main__":
    main()
