"""MI355X-native drop-in for the hot path of the reference's `stable_audio_tools` package.

`kalle_audio_amd.install()` aliases this package as `stable_audio_tools` in sys.modules so that the reference's
entry scripts (`from stable_audio_tools.models.factory import create_model_from_config`, twj_dataset.py:184,
infer_0723.py:216) import it unchanged."""
from .models.factory import create_model_from_config, create_model_from_config_path  # noqa: F401
