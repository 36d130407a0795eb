"""Drop-in import path of the reference (`from src.utils.early_stopping import EarlyStopping`)."""
from pl_convlstm_gan_amd.host.early_stopping import EarlyStopping  # noqa: F401
