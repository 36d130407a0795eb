"""Drop-in import path of the reference (`from src.models.coordconv import add_coord_channels`)."""
from pl_convlstm_gan_amd.models import add_coord_channels  # noqa: F401
