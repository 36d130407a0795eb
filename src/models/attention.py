"""Drop-in import path of the reference (`from src.models.attention import DEMAttention, LUAttention`)."""
from pl_convlstm_gan_amd.models import DEMAttention, LUAttention  # noqa: F401
