"""Drop-in for stable_audio_tools/training/losses/losses.py: LossModule (6-15), ValueLoss (16-23), MSELoss (44-69),
MultiLoss (85-101).  MSELoss runs the fused masked-MSE kernel (forward value + gradient in one pass)."""
import typing as tp

from torch import nn

from .... import functional as KF


class LossModule(nn.Module):
    def __init__(self, name: str, weight: float = 1.0):
        super().__init__()
        self.name = name
        self.weight = weight

    def forward(self, info, *args, **kwargs):
        raise NotImplementedError


class ValueLoss(LossModule):
    def __init__(self, key: str, name, weight: float = 1.0):
        super().__init__(name=name, weight=weight)
        self.key = key

    def forward(self, info):
        return self.weight * info[self.key]


class MSELoss(LossModule):
    def __init__(self, key_a: str, key_b: str, weight: float = 1.0, mask_key: str = None, name: str = 'mse_loss'):
        super().__init__(name=name, weight=weight)
        self.key_a = key_a
        self.key_b = key_b
        self.mask_key = mask_key

    def forward(self, info):
        a, b = info[self.key_a], info[self.key_b]
        mask = None
        if self.mask_key is not None and self.mask_key in info and info[self.mask_key] is not None:
            mask = info[self.mask_key]
            if mask.ndim == 3:
                if mask.shape[1] != 1:
                    raise NotImplementedError("per-channel loss masks")
                mask = mask[:, 0]
        if a.ndim != 3:
            raise NotImplementedError("MSELoss kernel expects (B, C, T) tensors")
        return KF.MSELossFn.apply(a, b, mask, float(self.weight))     # (the target's gradient exists only with enable_grad)


class MultiLoss(nn.Module):
    def __init__(self, losses: tp.List[LossModule]):
        super().__init__()
        self.losses = nn.ModuleList(losses)

    def forward(self, info):
        total_loss = 0
        losses = {}
        for loss_module in self.losses:
            module_loss = loss_module(info)
            total_loss = total_loss + module_loss
            losses[loss_module.name] = module_loss
        return total_loss, losses
