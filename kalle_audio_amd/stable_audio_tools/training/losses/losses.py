"""Loss terms of the training wrappers (reference interface: stable_audio_tools/training/losses/losses.py - LossModule 6-15,
ValueLoss 16-23, MSELoss 44-69, MultiLoss 85-101; same class names, constructor arguments and `forward(info)` contract).
MSELoss is the fused masked-MSE kernel (value and gradient from one pass over output / target)."""
import typing as tp

from torch import nn

from .... import functional as KF


class LossModule(nn.Module):
    """a named, weighted term computed from the step's `info` dict"""

    def __init__(self, name: str, weight: float = 1.0):
        super().__init__()
        self.name, self.weight = name, weight

    def forward(self, info, *args, **kwargs):
        raise NotImplementedError(f"{type(self).__name__} does not define its term")


class ValueLoss(LossModule):
    """a value some other part of the step already left in `info`"""

    def __init__(self, key: str, name, weight: float = 1.0):
        LossModule.__init__(self, name, weight)
        self.key = key

    def forward(self, info):
        return info[self.key] * self.weight


class MSELoss(LossModule):
    def __init__(self, key_a: str, key_b: str, weight: float = 1.0, mask_key: str = None, name: str = 'mse_loss'):
        LossModule.__init__(self, name, weight)
        self.key_a, self.key_b, self.mask_key = key_a, key_b, mask_key

    def _mask(self, info):
        """(B, T) boolean mask or None; the reference broadcasts a (B, 1, T) mask over channels (losses.py:57-63)"""
        mask = info.get(self.mask_key) if self.mask_key is not None else None
        if mask is not None and mask.ndim == 3:
            if mask.shape[1] != 1:
                raise NotImplementedError("per-channel loss masks")
            mask = mask[:, 0]
        return mask

    def forward(self, info):
        output, target = info[self.key_a], info[self.key_b]
        if output.ndim != 3:
            raise NotImplementedError("MSELoss kernel expects (B, C, T) tensors")
        # (the target's gradient exists only when the pretransform is trained: enable_grad)
        return KF.MSELossFn.apply(output, target, self._mask(info), float(self.weight))


class MultiLoss(nn.Module):
    """sum of the terms, in list order, plus each term by name (what the wrappers log)"""

    def __init__(self, losses: tp.List[LossModule]):
        super().__init__()
        self.losses = nn.ModuleList(losses)

    def forward(self, info):
        parts = {term.name: term(info) for term in self.losses}
        total = 0
        for value in parts.values():
            total = total + value
        return total, parts
