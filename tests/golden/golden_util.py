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


def digest(arr):
    """[l2 norm, sum, 8 fixed-position samples] of a gradient tensor"""
    a = np.asarray(arr, dtype=np.float64).reshape(-1)
    idx = (np.arange(DIGEST_SAMPLES, dtype=np.int64) * 2654435761 + 12345) % a.size
    return np.concatenate([[np.sqrt((a * a).sum()), a.sum()], a[idx]]).astype(np.float64)


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
