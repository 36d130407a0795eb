"""Drop-in import path of the reference (`from src.training.trainer import Trainer`)."""
from pl_convlstm_gan_amd.host.trainer import Trainer  # noqa: F401
