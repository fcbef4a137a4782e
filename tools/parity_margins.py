"""print the measured parity margins (actual rel-L2 / cosine vs the reference fixtures) for DESIGN.md:  python tools/parity_margins.py"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import golden_util as gu
import kalle_audio_amd
kalle_audio_amd.install()
from test_modules_gpu import T, fx, load_seeded, rel, cosine
dev = torch.device("cuda:0")
out = {}
from stable_audio_tools.models import transformer as M
from stable_audio_tools.models import autoencoders as A
from stable_audio_tools.models.factory import create_model_from_config
w = gu.WIDE_BLOCK
for name, ada, seed in (("block_wide_plain", False, 50), ("block_wide_adaln", True, 51)):
    f = fx(name)
    x = T(gu.make_input("x", (1, w["N"], w["D"]), seed), dev, True)
    ctx = T(gu.make_input("ctx", (1, w["S"], w["DC"]), seed), dev, True)
    blk = load_seeded(M.TransformerBlock(w["D"], dim_heads=64, cross_attend=True, dim_context=w["DC"], global_cond_dim=w["D"] if ada else None), seed, dev)
    kw = {"global_cond": T(gu.make_input("g", (1, w["D"]), seed), dev, True)} if ada else {}
    y = blk(x, context=ctx, rotary_pos_emb=M.RotaryEmbedding(32).to(dev).forward_from_seq_len(w["N"]), **kw)
    y.backward(T(gu.make_input("dy", (1, w["N"], w["D"]), seed), dev))
    out[name] = {"y": rel(y, f["y"].astype(np.float32)), "dx": rel(x.grad, f["dx"].astype(np.float32)), "dctx": rel(ctx.grad, f["dctx"].astype(np.float32))}
f = fx("vae_backward")
for snake in (True, False):
    tag = "snake" if snake else "elu"
    ae = load_seeded(create_model_from_config(gu.oobleck_cfg(snake)), 23, dev)
    ae.requires_grad_(True)
    wav = T(gu.make_input("wav", (2, 2, 1200), 66, 0.5), dev, True)
    z = ae.encode(wav); rec = ae.decode(z[:, :4] + 0.3 * z[:, 4:])
    ((z * T(gu.make_input("dz", tuple(z.shape), 66), dev)).sum() + (rec * T(gu.make_input("drec", tuple(rec.shape), 66), dev)).sum()).backward()
    out[f"vae_{tag}"] = {"z": rel(z, f[f"{tag}/ae/z"]), "rec": rel(rec, f[f"{tag}/ae/rec"]), "dwav": rel(wav.grad, f[f"{tag}/ae/dwav"])}
print(json.dumps(out, indent=1))
