"""Drop-in import path of the reference (`from src.losses.combined_loss import CombinedLoss`)."""
from pl_convlstm_gan_amd.losses import CombinedLoss  # noqa: F401
