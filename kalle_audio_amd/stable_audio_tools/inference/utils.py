"""stable_audio_tools/inference/utils.py (prepare_audio, set_audio_channels) + the PadCrop it uses (data/utils.py:8-20).
Host-side glue on [channels, samples] tensors; the resampler is third-party torchaudio and only needed when the caller's
sample rate differs from the model's."""
import torch


def set_audio_channels(audio, target_channels):
    """utils.py:5-18"""
    if target_channels == 1:
        audio = audio.mean(1, keepdim=True)
    elif target_channels == 2:
        if audio.shape[1] == 1:
            audio = audio.repeat(1, 2, 1)
        elif audio.shape[1] > 2:
            audio = audio[:, :2, :]
    return audio


def pad_crop(signal, n_samples):
    """PadCrop(n_samples, randomize=False) (data/utils.py:8-20): the first n_samples, zero-padded on the right"""
    n, s = signal.shape
    out = signal.new_zeros([n, n_samples])
    out[:, :min(s, n_samples)] = signal[:, :n_samples]
    return out


def prepare_audio(audio, in_sr, target_sr, target_length, target_channels, device):
    """utils.py:20-40"""
    audio = audio.to(device)
    if in_sr != target_sr:
        try:
            from torchaudio import transforms as T
        except ImportError as e:
            raise NotImplementedError(f"prepare_audio: resampling {in_sr} -> {target_sr} Hz needs torchaudio (third-party, "
                                      "absent here); hand over audio at the model's sample rate") from e
        audio = T.Resample(in_sr, target_sr).to(device)(audio)
    audio = pad_crop(audio, target_length)
    if audio.dim() == 1:
        audio = audio.unsqueeze(0).unsqueeze(0)
    elif audio.dim() == 2:
        audio = audio.unsqueeze(0)
    return set_audio_channels(audio, target_channels)
