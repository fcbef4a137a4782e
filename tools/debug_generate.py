"""debug: the first call of tests/test_round2_gpu.py::test_generate_diffusion_cond_end_to_end with KALLE_TRACE=1"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import golden_util as gu
import test_round2_gpu as R
dev = torch.device("cuda:0")
e = gu.E2E
from kalle_audio_amd.stable_audio_tools.inference.generation import generate_diffusion_cond
ctx, cm, gl = R._e2e_cond(60, dev)
cond = {"prompt": (ctx, cm), "g": (gl, None)}
model = R._cond_model(dev, 4, "rectified_flow", 60)
print("model built", file=sys.stderr, flush=True)
lat = generate_diffusion_cond(model, return_latents=True, steps=e["steps"], cfg_scale=e["cfg_scale"], conditioning_tensors=cond,
                              batch_size=e["B"], sample_size=40 * e["T"], seed=e["seed"], device="cpu")
torch.cuda.synchronize()
print("latents ok", lat.shape, file=sys.stderr, flush=True)
audio = model.pretransform.decode(lat.float())
torch.cuda.synchronize()
print("decode ok", audio.shape, file=sys.stderr, flush=True)
