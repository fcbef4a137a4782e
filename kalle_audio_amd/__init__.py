"""kalle_audio_amd - MI355X (gfx950) native implementation of kalle-audio's DiT / audio-VAE hot path.

Layout: csrc/ (HIP kernels + C-ABI, built into libkalle_hip.so), _lib.py (ctypes binding generated from
include/kalle_hip.h), ops.py / conv_ops.py (tensor-level kernel launches), dit_ops.py (manual fwd/bwd of the DiT
block), functional.py (autograd shims), stable_audio_tools/ (drop-in modules with the reference's names, signatures
and state-dict keys), engine.py (data-parallel trainer: fused Adam + RCCL gradient all-reduce overlapped with backward).
"""
import sys

__version__ = "0.1.0"


def install():
    """Make `import stable_audio_tools` resolve to this package's drop-in (for the reference's entry scripts)."""
    import importlib
    pkg = importlib.import_module(__name__ + ".stable_audio_tools")
    sys.modules["stable_audio_tools"] = pkg
    for sub in ("models", "models.factory", "models.transformer", "models.dit", "models.diffusion", "models.blocks",
                "models.autoencoders", "models.bottleneck", "models.pretransforms", "models.utils", "training",
                "training.diffusion", "training.losses", "training.losses.losses", "training.utils", "inference",
                "inference.sampling", "inference.generation", "inference.utils"):
        sys.modules["stable_audio_tools." + sub] = importlib.import_module(f"{__name__}.stable_audio_tools.{sub}")
    # the reference's top-level modules on the path: the task models (train_offline.py:19 `from model_sigmaVAE import
    # Llasa`, train.py:24 `from model import Llasa`) and the mel-VAE (infer_0828_sigma.py:18 `from flows import BigVGANFlowVAE`)
    sys.modules["model_sigmaVAE"] = importlib.import_module(__name__ + ".model_sigmaVAE")
    sys.modules["model"] = importlib.import_module(__name__ + ".model")          # train.py:24 `from model import Llasa`
    sys.modules["flows"] = importlib.import_module(__name__ + ".flows")
    return pkg
