"""Drop-in import path of the reference (`from src.models.generator import Generator`)."""
from pl_convlstm_gan_amd.models import Generator, UpsampleBlock  # noqa: F401
