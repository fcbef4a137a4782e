"""Deterministic weights / digests shared by the fixture generator (make_golden.py, runs only where
/root/reference is mounted) and by the tests (run anywhere).  Own code; nothing here comes from the reference.

Weights are NOT stored in the fixtures: both sides rebuild them from (ordered parameter names+shapes, seed) with
numpy's PCG64 stream, so a fixture holds only inputs, outputs, input-gradients and per-parameter gradient digests.
"""
import hashlib

import numpy as np


def _seed_for(name, seed):
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return int.from_bytes(h[:8], "little")


def make_param(name, shape, seed):
    """value for one parameter; scale chosen so that no branch of the model is zero (the reference zero-inits
    several output projections, which would make parity checks vacuous - SURVEY.md section 7)."""
    rng = np.random.Generator(np.random.PCG64(_seed_for(name, seed)))
    n = rng.standard_normal(size=tuple(shape), dtype=np.float64)
    leaf = name.split(".")[-1]
    if leaf == "weight_g":
        v = 0.5 + 0.05 * n  # keeps the deep conv stacks (and the final tanh) out of saturation
    elif leaf in ("gamma", "scale"):
        v = 1.0 + 0.1 * n
    elif len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        if leaf == "weight" and "timestep_features" in name:
            v = n  # FourierFeatures: ~N(0,1) like its own init (blocks.py:88-89)
        else:
            v = n / np.sqrt(max(fan_in, 1))
    elif leaf in ("alpha", "beta"):
        v = 0.3 * n
    else:
        v = 0.1 * n
    return v.astype(np.float32)


def make_state(named_shapes, seed):
    """named_shapes: iterable of (name, shape) -> {name: float32 array}"""
    return {name: make_param(name, shape, seed) for name, shape in named_shapes}


def make_input(name, shape, seed, scale=1.0):
    rng = np.random.Generator(np.random.PCG64(_seed_for("input:" + name, seed)))
    return (scale * rng.standard_normal(size=tuple(shape))).astype(np.float32)


def make_mask(name, shape, seed, p_keep=0.8):
    rng = np.random.Generator(np.random.PCG64(_seed_for("mask:" + name, seed)))
    m = rng.random(size=tuple(shape)) < p_keep
    m[..., 0] = True
    return m


DIGEST_SAMPLES = 8


def digest_index(size, n=DIGEST_SAMPLES):
    return (np.arange(n, dtype=np.int64) * 2654435761 + 12345) % size


def digest(arr, n=DIGEST_SAMPLES):
    """[l2 norm, sum, n fixed-position samples] of a gradient tensor (8 samples in the round-1 fixtures, 64 in the wide ones)"""
    a = np.asarray(arr, dtype=np.float64).reshape(-1)
    return np.concatenate([[np.sqrt((a * a).sum()), a.sum()], a[digest_index(a.size, n)]]).astype(np.float64)


# mel-VAE hyper-parameters used by the fixtures (the reference's own JSON for backup/flows.py is absent)
MELVAE_CONFIGS = {
    "amp1_causal": dict(latent_dim=8, use_vae=True, downsample_channels=[12, 16, 24], downsample_rates=[2, 4],
                        upsample_rates=[4, 2], upsample_kernel_sizes=[8, 4], upsample_initial_channel=32, resblock="1",
                        resblock_kernel_sizes=[3, 7], resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5]],
                        activation="snakebeta", snake_logscale=True, causal=True, flow_hidden_channels=16),
    "amp2_same": dict(latent_dim=8, use_vae=True, downsample_channels=[12, 16, 24], downsample_rates=[2, 4],
                      upsample_rates=[4, 2], upsample_kernel_sizes=[8, 4], upsample_initial_channel=32, resblock="2",
                      resblock_kernel_sizes=[3, 5], resblock_dilation_sizes=[[1, 3], [1, 3]],
                      activation="snake", snake_logscale=True, causal=False, flow_hidden_channels=16),
}


# Llasa fixture: a tiny Llama-3-style decoder (head_dim 64, GQA, llama3 rope scaling that is active at these lengths)
LLASA_CONFIG = dict(
    latent_dim=16, tokenizer_len=310,
    llama=dict(vocab_size=300, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
               num_key_value_heads=1, head_dim=64, rms_norm_eps=1e-5, rope_theta=500000.0, max_position_embeddings=256,
               rope_scaling=dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0,
                                 original_max_position_embeddings=32),
               tie_word_embeddings=True, attention_bias=False, mlp_bias=False, hidden_act="silu"))


def llasa_batch(lc, seed, B=3, L=40):
    """collate()-shaped batch (twj_dataset_offline.py:371-384): text tokens, then audio frames, then right padding"""
    rng = np.random.Generator(np.random.PCG64(_seed_for("llasa_batch", seed)))
    lat = lc["latent_dim"]
    ids = rng.integers(0, lc["tokenizer_len"], size=(B, L)).astype(np.int64)
    ids_mask = np.zeros((B, L), np.float32)
    audio_mask = np.zeros((B, L), np.float32)
    target_mask = np.zeros((B, L), np.float32)
    end_mask = np.zeros((B, L), np.float32)
    for b in range(B):
        nt = int(rng.integers(4, 10))
        na = int(rng.integers(8, L - nt - 1)) if b else L - nt      # sample 0 fills the whole length (no padding)
        ids_mask[b, :nt] = 1
        audio_mask[b, nt:nt + na] = 1
        target_mask[b, nt - 1:nt + na - 1] = 1                        # positions that predict an audio frame
        end_mask[b, nt + na - 1] = 1                                  # position that predicts the end distribution
    return dict(input_ids=ids, audio_latents=make_input("llasa_lat", (B, L, lat), seed),
                audio_distribution_l=make_input("llasa_lbl", (B, L, lat), seed), ids_mask=ids_mask,
                audio_mask=audio_mask, target_mask=target_mask, end_mask=end_mask)


# round-2 fixtures --------------------------------------------------------------------------------------------------------
# bench-width TransformerBlock (D = 1536, 24 heads, context 768 = 12 kv heads, 126 tokens, 130 context tokens)
WIDE_BLOCK = dict(D=1536, DC=768, N=126, S=130, B=1)
QK_NORM_BLOCK = dict(D=256, DC=128, N=40, S=24, B=2)
# off-default block options (round 3): TransformerBlock(conformer=True) and ContinuousTransformer(use_sinusoidal_emb / use_abs_pos_emb)
OPT_BLOCK = dict(D=256, DC=128, N=40, S=48, B=2)
OPT_CT = dict(D=128, depth=2, dim_in=16, dim_out=16, N=37, P=3, B=2, max_len=64)

# Llasa at 4 heads / 2 kv heads, sequences of ~300 with ragged right padding (30 s clips of configs/twj_0828.yaml are
# 375 frames + text)
LLASA_WIDE_CONFIG = dict(
    latent_dim=32, tokenizer_len=310,
    llama=dict(vocab_size=300, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
               num_key_value_heads=2, head_dim=64, rms_norm_eps=1e-5, rope_theta=500000.0, max_position_embeddings=1024,
               rope_scaling=dict(rope_type="llama3", factor=8.0, low_freq_factor=1.0, high_freq_factor=4.0,
                                 original_max_position_embeddings=64),
               tie_word_embeddings=True, attention_bias=False, mlp_bias=False, hidden_act="silu"))


def llasa_batch_long(lc, seed, B=3, L=300, label_mult=1):
    """llasa_batch with the text / audio lengths scaled to L (sample 0 unpadded, the others ragged)"""
    rng = np.random.Generator(np.random.PCG64(_seed_for("llasa_batch_long", seed)))
    lat = lc["latent_dim"]
    ids = rng.integers(0, lc["tokenizer_len"], size=(B, L)).astype(np.int64)
    ids_mask = np.zeros((B, L), np.float32)
    audio_mask = np.zeros((B, L), np.float32)
    target_mask = np.zeros((B, L), np.float32)
    end_mask = np.zeros((B, L), np.float32)
    for b in range(B):
        nt = int(rng.integers(4, max(6, min(40, L // 4))))
        na = int(rng.integers(L // 3, L - nt - 1)) if b else L - nt
        ids_mask[b, :nt] = 1
        audio_mask[b, nt:nt + na] = 1
        target_mask[b, nt - 1:nt + na - 1] = 1
        end_mask[b, nt + na - 1] = 1
    return dict(input_ids=ids, audio_latents=make_input("llasa_lat", (B, L, lat), seed),
                audio_distribution_l=make_input("llasa_lbl", (B, L, lat * label_mult), seed), ids_mask=ids_mask,
                audio_mask=audio_mask, target_mask=target_mask, end_mask=end_mask)


def default_mean_stdev(latents):
    """stand-in for the reference's MISSING twj_utils.get_mean_stdev_from_stableaudio2_latents (model.py:7,84; the file is a
    dangling symlink): latents [B, 2*lat, L] = mean || scale of the Oobleck encoder; stdev = softplus(scale) + 1e-4 as
    stable_audio_tools/models/bottleneck.py:51-54 derives it from the same tensor.  Parity of THIS callable is unpinned;
    everything around it (model.py:84-100) is pinned with it injected on both sides.  torch in, torch out."""
    import torch
    mean, scale = latents.chunk(2, dim=1)
    return mean, torch.nn.functional.softplus(scale) + 1e-4


# tiny Oobleck autoencoder shared by the end-to-end fixtures (same layout as make_golden.py's oobleck_vae_snake)
def oobleck_cfg(snake=True):
    return {
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


E2E = dict(D=128, DC=64, G=32, S=9, T=125, B=2, seed=777, steps=4, cfg_scale=3.0)
