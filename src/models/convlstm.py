"""Drop-in import path of the reference (`from src.models.convlstm import ConvLSTMCell`)."""
from pl_convlstm_gan_amd.models import ConvLSTMCell  # noqa: F401
