"""MI355X-native ConvLSTM rollout engine behind the reference's model interface.

csrc/      hand-written HIP kernels (gfx950) + the C ABI (include/plconvlstm.h)
_lib.py    ctypes binding of libplconvlstm.so
ops.py     pointer/stream plumbing
models.py  ConvLSTMCell / Generator with the reference's signatures and state_dict keys
"""
from .models import (ConvLSTMCell, DEMAttention, Generator, LUAttention, UpsampleBlock,  # noqa: F401
                     add_coord_channels)

__all__ = ["ConvLSTMCell", "DEMAttention", "Generator", "LUAttention", "UpsampleBlock", "add_coord_channels"]
