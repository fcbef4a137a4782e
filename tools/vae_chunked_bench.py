"""Chunked encode / decode of one long stereo clip (AudioAutoencoder.encode_audio / decode_audio(chunked=True), autoencoders.py:429-560:
128-latent chunks, overlap 32) - all chunks ride on the batch axis of ONE encoder / decoder pass.  python tools/vae_chunked_bench.py [seconds]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kalle_audio_amd
kalle_audio_amd.install()
from stable_audio_tools.models.factory import create_model_from_config
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 95.0
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
ae.eval().requires_grad_(False)
n = int(secs * 44100) // 2048 * 2048
wav = torch.rand(1, 2, n, device=dev) * 2 - 1
with torch.no_grad():
    z = ae.encode_audio(wav, chunked=True, chunk_size=128, overlap=32)
    cases = (("encode", lambda: ae.encode_audio(wav, chunked=True, chunk_size=128, overlap=32)),
             ("decode", lambda: ae.decode_audio(z[:, :64] if z.shape[1] > 64 else z, chunked=True, chunk_size=128, overlap=32)))
    for name, fn in cases:
        y = fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            y = fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"chunked {name}: {n / 44100:.1f} s clip, {dt * 1e3:.1f} ms ({n / 44100 / dt:.0f} audio-s/s) out {tuple(y.shape)}")
