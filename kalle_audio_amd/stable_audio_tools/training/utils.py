"""stable_audio_tools/training/utils.py: InverseLR (17-56), optimizer / scheduler factories (76-111).
"FusedAdam" (deepspeed in the reference, 88-90) maps to this build's fused HIP Adam (kalle_audio_amd.engine)."""
import torch


class InverseLR(torch.optim.lr_scheduler._LRScheduler):
    """inverse-decay schedule with exponential warm-up: lr = warmup * max(final_lr, base * (1 + step/inv_gamma)^-power)"""

    def __init__(self, optimizer, inv_gamma=1., power=1., warmup=0., final_lr=0., last_epoch=-1):
        self.inv_gamma = inv_gamma
        self.power = power
        if not 0. <= warmup < 1:
            raise ValueError('Invalid value for warmup')
        self.warmup = warmup
        self.final_lr = final_lr
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        return self._get_closed_form_lr()

    def _get_closed_form_lr(self):
        warmup = 1 - self.warmup ** (self.last_epoch + 1)
        lr_mult = (1 + self.last_epoch / self.inv_gamma) ** -self.power
        return [warmup * max(self.final_lr, base_lr * lr_mult) for base_lr in self.base_lrs]


def create_optimizer_from_config(optimizer_config, parameters):
    optimizer_type = optimizer_config["type"]
    if optimizer_type == "FusedAdam":
        from ...engine import FusedAdam
        return FusedAdam(parameters, **optimizer_config["config"])
    return getattr(torch.optim, optimizer_type)(parameters, **optimizer_config["config"])


def create_scheduler_from_config(scheduler_config, optimizer):
    if scheduler_config["type"] == "InverseLR":
        scheduler_fn = InverseLR
    else:
        scheduler_fn = getattr(torch.optim.lr_scheduler, scheduler_config["type"])
    return scheduler_fn(optimizer, **scheduler_config["config"])
