"""Pre-compute VAE latents on the GPU - the replacement for the CPU encode inside DataLoader workers.

The reference's online dataset encodes every clip on the CPU in a worker process (`twj_dataset.py:225-239`:
librosa.load -> normalize * 0.95 -> mono duplicated to two channels -> `generator.pretransform.encode`); its offline datasets
read the result back from `.npy` files holding the encoder output `mean || scale`, shape [2 * latent_dim, T]
(`twj_data_offline_sd2.py:279-287`).  The HIP kernels have no CPU path, so with this build the encode runs here, once, on the
GPU, and training uses the reference's own offline dataset classes unchanged.

    python examples/encode_latents_hip.py --model-config model_config.json --ckpt vae.ckpt --list clips.txt --out latents/

clips.txt: one audio path per line - 16-bit / float PCM `.wav` (read with scipy; already at the model's sample rate) or a
`.npy` float array [samples] / [channels, samples].  Output: <out>/<stem>.npy, float32 [2 * latent_dim, T].
Long clips go through the batched chunk pipeline (AudioAutoencoder.encode_audio(chunked=True))."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load_audio(path):
    if path.endswith(".npy"):
        a = np.load(path).astype(np.float32)
    else:
        from scipy.io import wavfile
        _, a = wavfile.read(path)
        a = a.astype(np.float32) / 32768.0 if a.dtype == np.int16 else a.astype(np.float32)
        a = a.T if a.ndim == 2 else a
    if a.ndim == 2:
        a = a.mean(0)                                   # librosa.load(mono=True)
    peak = np.abs(a).max()
    return a / peak * 0.95 if peak > 0 else a           # librosa.util.normalize(wav) * 0.95  (twj_dataset.py:231)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model-config", required=True)
    ap.add_argument("--ckpt", default=None)
    ap.add_argument("--list", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--chunk-latents", type=int, default=128)
    ap.add_argument("--overlap-latents", type=int, default=32)
    args = ap.parse_args()
    import kalle_audio_amd
    kalle_audio_amd.install()
    from stable_audio_tools.models.factory import create_model_from_config
    from stable_audio_tools.models.utils import load_ckpt_state_dict
    cfg = json.load(open(args.model_config))
    model = create_model_from_config(cfg)                                    # twj_dataset.py:184-187
    if args.ckpt:
        model.load_state_dict(load_ckpt_state_dict(args.ckpt), strict=False)
    vae = (model.pretransform if hasattr(model, "pretransform") and model.pretransform is not None else model)
    vae = vae.cuda().eval().requires_grad_(False)
    ae = getattr(vae, "model", vae)
    ratio = ae.downsampling_ratio
    os.makedirs(args.out, exist_ok=True)
    with torch.no_grad():
        for line in open(args.list):
            path = line.strip()
            if not path:
                continue
            wav = load_audio(path)
            wav = np.pad(wav, (0, (-len(wav)) % ratio))                      # whole latent frames
            x = torch.from_numpy(wav).cuda().view(1, 1, -1).repeat(1, 2, 1)  # mono duplicated to two channels (:233)
            chunked = x.shape[2] // ratio > 2 * args.chunk_latents
            z = ae.encode_audio(x, chunked=chunked, chunk_size=args.chunk_latents, overlap=args.overlap_latents)
            if hasattr(vae, "scale"):
                z = z / vae.scale                                             # AutoencoderPretransform.encode (pretransforms.py:61)
            out = os.path.join(args.out, os.path.splitext(os.path.basename(path))[0] + ".npy")
            np.save(out, z[0].float().cpu().numpy())                         # [2 * latent_dim, T]
            print(out, tuple(z.shape[1:]), flush=True)


if __name__ == "__main__":
    main()
