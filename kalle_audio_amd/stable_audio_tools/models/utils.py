"""stable_audio_tools/models/utils.py:6-12: checkpoint loader semantics kept."""
import torch
from safetensors.torch import load_file


def load_ckpt_state_dict(ckpt_path):
    if ckpt_path.endswith(".safetensors"):
        return load_file(ckpt_path)
    return torch.load(ckpt_path, map_location="cpu")["state_dict"]
