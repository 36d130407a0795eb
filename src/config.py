"""Drop-in import path of the reference (`from src.config import Config, load_config`)."""
from pl_convlstm_gan_amd.host.config import (Config, DataConfig, ModelConfig, OutputConfig,  # noqa: F401
                                             TrainingConfig, load_config)
