from .diffusion import DiffusionCondTrainingWrapper, diffusion_train_step  # noqa: F401
