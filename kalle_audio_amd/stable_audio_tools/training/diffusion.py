"""The DiT train step of stable_audio_tools/training/diffusion.py:219-450 without pytorch-lightning.

`DiffusionCondTrainingWrapper` keeps the reference's constructor arguments and `training_step((reals, metadata),
batch_idx) -> loss` contract (timestep draw 358-362, alpha/sigma 365-368, noising and target 371-379, model call 390,
MSE 393-399); lightning's backward / optimizer / DDP duties are taken by kalle_audio_amd.engine.DataParallelTrainer.
`diffusion_train_step` is the seed-free functional form (explicit t and noise) the parity tests use."""
import random
import typing as tp

import torch
from torch import nn

from ... import ops
from .losses.losses import MSELoss, MultiLoss


def diffusion_train_step(diffusion, latents, t, noise, cond, objective="v", cfg_dropout_prob=0.0, padding_mask=None,
                         losses=None):
    """training/diffusion.py:365-399 given t and noise. Returns (loss, info)."""
    latents = latents.float().contiguous()
    if latents.requires_grad:       # a pretransform trained with enable_grad: the loss reaches the VAE through x_t and target
        from ... import functional as KF
        x_t, targets = KF.DiffuseFn.apply(latents, noise.float().contiguous(), t.float().contiguous(), objective)
    else:
        x_t, targets = ops.diffuse_fwd(latents, noise.float().contiguous(), t.float().contiguous(), objective)
    extra = {"mask": padding_mask} if padding_mask is not None else {}
    output = diffusion(x_t, t, cond=cond, cfg_dropout_prob=cfg_dropout_prob, **extra)
    if losses is None:
        losses = MultiLoss([MSELoss("output", "targets", weight=1.0, mask_key="padding_mask", name="mse_loss")])
    info = {"output": output, "targets": targets, "padding_mask": padding_mask}
    loss, parts = losses(info)
    info["x_t"] = x_t
    info["losses"] = parts
    return loss, info


class DiffusionCondTrainingWrapper(nn.Module):
    """training/diffusion.py:219-450 (EMA via ema_pytorch and the wandb demo callbacks are not carried over)."""

    def __init__(self, model, lr: float = None, mask_padding: bool = False, mask_padding_dropout: float = 0.0,
                 use_ema: bool = False, log_loss_info: bool = False, optimizer_configs: dict = None,
                 pre_encoded: bool = False, cfg_dropout_prob=0.1,
                 timestep_sampler: tp.Literal["uniform", "logit_normal"] = "uniform"):
        super().__init__()
        self.diffusion = model
        self.diffusion_ema = None
        self.mask_padding = mask_padding
        self.mask_padding_dropout = mask_padding_dropout
        self.cfg_dropout_prob = cfg_dropout_prob
        self.rng = torch.quasirandom.SobolEngine(1, scramble=True)
        self.timestep_sampler = timestep_sampler
        self.diffusion_objective = model.diffusion_objective
        self.loss_modules = [MSELoss("output", "targets", weight=1.0,
                                     mask_key="padding_mask" if self.mask_padding else None, name="mse_loss")]
        self.losses = MultiLoss(self.loss_modules)
        self.log_loss_info = log_loss_info
        assert lr is not None or optimizer_configs is not None, \
            "Must specify either lr or optimizer_configs in training config"
        if optimizer_configs is None:
            optimizer_configs = {"diffusion": {"optimizer": {"type": "Adam", "config": {"lr": lr}}}}
        self.optimizer_configs = optimizer_configs
        self.pre_encoded = pre_encoded

    @property
    def device(self):
        return next(self.diffusion.parameters()).device

    def configure_optimizers(self):
        from .utils import create_optimizer_from_config, create_scheduler_from_config
        cfg = self.optimizer_configs['diffusion']
        opt = create_optimizer_from_config(cfg['optimizer'], self.diffusion.parameters())
        if "scheduler" in cfg:
            return [opt], [{"scheduler": create_scheduler_from_config(cfg['scheduler'], opt), "interval": "step"}]
        return [opt]

    def training_step(self, batch, batch_idx=0):
        reals, metadata = batch
        if reals.ndim == 4 and reals.shape[0] == 1:
            reals = reals[0]
        diffusion_input = reals
        conditioning = self.diffusion.conditioner(metadata, self.device)
        use_padding_mask = self.mask_padding and random.random() > self.mask_padding_dropout
        padding_masks = None
        if use_padding_mask:
            padding_masks = torch.stack([md["padding_mask"][0] for md in metadata], dim=0).to(self.device)
        if self.diffusion.pretransform is not None:
            if not self.pre_encoded:
                with torch.set_grad_enabled(bool(getattr(self.diffusion.pretransform, "enable_grad", False))):   # :343-346
                    diffusion_input = self.diffusion.pretransform.encode(diffusion_input)
                if use_padding_mask:
                    padding_masks = torch.nn.functional.interpolate(
                        padding_masks.unsqueeze(1).float(), size=diffusion_input.shape[2], mode="nearest").squeeze(1).bool()
            elif hasattr(self.diffusion.pretransform, "scale") and self.diffusion.pretransform.scale != 1.0:
                diffusion_input = diffusion_input / self.diffusion.pretransform.scale
        if self.timestep_sampler == "uniform":
            t = self.rng.draw(reals.shape[0])[:, 0].to(self.device)
        else:
            t = torch.sigmoid(torch.randn(reals.shape[0], device=self.device))
        noise = torch.randn_like(diffusion_input)
        loss, _ = diffusion_train_step(self.diffusion, diffusion_input, t, noise, conditioning,
                                       objective=self.diffusion_objective, cfg_dropout_prob=self.cfg_dropout_prob,
                                       padding_mask=padding_masks, losses=self.losses)
        return loss
