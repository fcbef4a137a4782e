"""Round-3 golden fixtures: the REFERENCE implementation (/root/reference, imported in place, CPU fp32) run on

  * generate_diffusion_cond(init_audio=..., init_noise_level=...) for a rectified-flow model - the "variation" branch
    (inference/generation.py:164-183, 226-228 -> sampling.py:200-232): prepare_audio (PadCrop, channel fix-up), the optional
    pretransform.encode, `x = init (1 - sigma_max) + noise sigma_max`, discrete Euler from sigma_max;
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


def main():
    mg.install_stubs()
    sys.path.insert(0, REF)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    generate_init_audio()
    print("done")


if __name__ == "__main__":
    main()
