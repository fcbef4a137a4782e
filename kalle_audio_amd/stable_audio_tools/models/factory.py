"""Drop-in for stable_audio_tools/models/factory.py: config -> model dispatch (3-24), create_pretransform_from_config
(32-82, 'autoencoder' type) and create_bottleneck_from_config (84-153, continuous bottlenecks)."""
import json


def create_model_from_config(model_config):
    model_type = model_config.get('model_type', None)
    assert model_type is not None, 'model_type must be specified in model config'
    if model_type == 'autoencoder':
        from .autoencoders import create_autoencoder_from_config
        return create_autoencoder_from_config(model_config)
    if model_type in ('diffusion_cond', 'diffusion_cond_inpaint'):
        from .diffusion import create_diffusion_cond_from_config
        return create_diffusion_cond_from_config(model_config)
    raise NotImplementedError(f'model type {model_type!r} is outside the accelerated DiT / audio-VAE path')


def create_model_from_config_path(model_config_path):
    with open(model_config_path) as f:
        model_config = json.load(f)
    return create_model_from_config(model_config)


def create_pretransform_from_config(pretransform_config, sample_rate):
    pretransform_type = pretransform_config.get('type', None)
    assert pretransform_type is not None, 'type must be specified in pretransform config'
    if pretransform_type != 'autoencoder':
        raise NotImplementedError(f'pretransform type {pretransform_type!r}: only "autoencoder" is built')
    from .autoencoders import create_autoencoder_from_config
    from .pretransforms import AutoencoderPretransform
    autoencoder_config = {"sample_rate": sample_rate, "model": pretransform_config["config"]}
    autoencoder = create_autoencoder_from_config(autoencoder_config)
    pretransform = AutoencoderPretransform(autoencoder, scale=pretransform_config.get("scale", 1.0),
                                           model_half=pretransform_config.get("model_half", False),
                                           iterate_batch=pretransform_config.get("iterate_batch", False),
                                           chunked=pretransform_config.get("chunked", False))
    enable_grad = pretransform_config.get('enable_grad', False)     # (VAE fine-tuning: conv backward in csrc/conv1d_bwd.hip)
    pretransform.enable_grad = enable_grad
    pretransform.eval().requires_grad_(pretransform.enable_grad)
    return pretransform


def create_bottleneck_from_config(bottleneck_config):
    bottleneck_type = bottleneck_config.get('type', None)
    assert bottleneck_type is not None, 'type must be specified in bottleneck config'
    if bottleneck_type == 'tanh':
        from .bottleneck import TanhBottleneck
        bottleneck = TanhBottleneck()
    elif bottleneck_type == 'vae':
        from .bottleneck import VAEBottleneck
        bottleneck = VAEBottleneck()
    else:
        raise NotImplementedError(f'bottleneck type {bottleneck_type!r}: discrete codecs are out of scope')
    if not bottleneck_config.get('requires_grad', True):
        for param in bottleneck.parameters():
            param.requires_grad = False
    return bottleneck
