"""forward vs forward + backward of the Oobleck VAE conv path with gradients enabled (models/factory.py:77-80 enable_grad):
python tools/vae_train_bench.py [B] [seconds of audio]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kalle_audio_amd
kalle_audio_amd.install()
from stable_audio_tools.models.factory import create_model_from_config
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0
cfg = {"model_type": "autoencoder", "sample_rate": 44100, "sample_size": 441000, "audio_channels": 2,
       "model": {"encoder": {"type": "oobleck", "config": {"in_channels": 2, "channels": 128, "c_mults": [1, 2, 4, 8, 16],
                                                          "strides": [2, 4, 4, 8, 8], "latent_dim": 128, "use_snake": True}},
                 "decoder": {"type": "oobleck", "config": {"out_channels": 2, "channels": 128, "c_mults": [1, 2, 4, 8, 16],
                                                          "strides": [2, 4, 4, 8, 8], "latent_dim": 64, "use_snake": True,
                                                          "final_tanh": False}},
                 "bottleneck": {"type": "vae"}, "latent_dim": 64, "downsampling_ratio": 2048, "io_channels": 2}}
dev = torch.device("cuda")
torch.manual_seed(0)
with torch.device(dev):
    ae = create_model_from_config(cfg)
ae.train()
n = int(secs * 44100) // 2048 * 2048
wav = (torch.rand(B, 2, n, device=dev) * 2 - 1)


def run(backward):
    for p in ae.parameters():
        p.grad = None
    z = ae.encode(wav)
    rec = ae.decode(z[:, :64].contiguous())
    if backward:
        rec.square().mean().backward()
    return rec


for name, bw in (("forward (encode + decode, grad enabled)", False), ("forward + backward", True)):
    run(bw); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        run(bw)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms  (B={B}, {n} samples)", flush=True)
