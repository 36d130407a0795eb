"""Thin Python wrappers over the C ABI: pointer/stream plumbing only, no arithmetic.

Every function requires CUDA(HIP) fp32 contiguous tensors and raises otherwise; there is no
CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _require_dev(t: torch.Tensor, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(
            f"{name} is on {t.device}: the ConvLSTM engine runs on MI355X (HIP) only; there is no CPU fallback")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32 (got {t.dtype}); the engine converts to bf16 internally")
    return t.contiguous()


def staging_device() -> torch.device:
    """Where CPU-resident operands are staged for the HIP kernels.  There is no CPU arithmetic path:
    without a GPU this raises."""
    if not torch.cuda.is_available():
        raise RuntimeError("the ConvLSTM engine runs on MI355X (HIP) only and no GPU is visible; "
                           "there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def workspace(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def convlstm_cell_forward(x, h, c, weight, bias):
    """ConvLSTMCell.forward on the HIP kernel.  NCHW fp32 in, (h', c') NCHW fp32 out."""
    L = _lib.lib()
    x = _require_dev(x, "x")
    h = _require_dev(h, "h_cur")
    c = _require_dev(c, "c_cur")
    weight = _require_dev(weight, "weight")
    B, Cin, H, W = x.shape
    Ch = h.shape[1]
    if weight.shape != (4 * Ch, Cin + Ch, 3, 3):
        raise ValueError(f"weight shape {tuple(weight.shape)} does not match (4*{Ch}, {Cin}+{Ch}, 3, 3)")
    if h.shape != (B, Ch, H, W) or c.shape != (B, Ch, H, W):
        raise ValueError("h_cur / c_cur shape mismatch")
    bias_p = _require_dev(bias, "bias").data_ptr() if bias is not None else None
    h2 = torch.empty_like(h)
    c2 = torch.empty_like(c)
    nbytes = L.plc_convlstm_cell_workspace_bytes(B, Cin, Ch, H, W)
    ws = workspace(nbytes, x.device)
    rc = L.plc_convlstm_cell_fwd(x.data_ptr(), h.data_ptr(), c.data_ptr(), weight.data_ptr(), bias_p,
                                 h2.data_ptr(), c2.data_ptr(), ws.data_ptr(), nbytes, B, Cin, Ch, H, W, _stream())
    _lib.check(rc, "plc_convlstm_cell_fwd")
    return h2, c2


def convlstm_cell_backward(x, h, c, weight, bias, dh_next, dc_next):
    """Gradients of ConvLSTMCell.forward: returns (dx, dh, dc, dweight, dbias)."""
    L = _lib.lib()
    x, h, c, weight = (_require_dev(t, n) for t, n in ((x, "x"), (h, "h_cur"), (c, "c_cur"), (weight, "weight")))
    dh_next = _require_dev(dh_next, "dh_next")
    dc_next = _require_dev(dc_next, "dc_next")
    B, Cin, H, W = x.shape
    Ch = h.shape[1]
    dx, dh, dc = torch.empty_like(x), torch.empty_like(h), torch.empty_like(c)
    dw = torch.empty_like(weight)
    db = torch.empty(4 * Ch, dtype=torch.float32, device=x.device)
    nbytes = L.plc_convlstm_cell_bwd_workspace_bytes(B, Cin, Ch, H, W)
    ws = workspace(nbytes, x.device)
    bias_p = _require_dev(bias, "bias").data_ptr() if bias is not None else None
    rc = L.plc_convlstm_cell_bwd(x.data_ptr(), h.data_ptr(), c.data_ptr(), weight.data_ptr(), bias_p,
                                 dh_next.data_ptr(), dc_next.data_ptr(), dx.data_ptr(), dh.data_ptr(), dc.data_ptr(),
                                 dw.data_ptr(), db.data_ptr(), ws.data_ptr(), nbytes, B, Cin, Ch, H, W, _stream())
    _lib.check(rc, "plc_convlstm_cell_bwd")
    return dx, dh, dc, dw, (db if bias is not None else None)


def to_cb8_bf16(x: torch.Tensor) -> torch.Tensor:
    """NCHW fp32 -> CB8 bf16 [N][ceil(C/8)][H][W][8] (returned as an int16 tensor)."""
    L = _lib.lib()
    x = _require_dev(x, "x")
    N, Cc, H, W = x.shape
    out = torch.empty((N, (Cc + 7) // 8, H, W, 8), dtype=torch.int16, device=x.device)
    _lib.check(L.plc_nchw_f32_to_cb8_bf16(x.data_ptr(), out.data_ptr(), N, Cc, H, W, _stream()), "to_cb8_bf16")
    return out


def from_cb8_bf16(x: torch.Tensor, channels: int) -> torch.Tensor:
    L = _lib.lib()
    N, ncb, H, W, _ = x.shape
    out = torch.empty((N, channels, H, W), dtype=torch.float32, device=x.device)
    _lib.check(L.plc_cb8_bf16_to_nchw_f32(x.data_ptr(), out.data_ptr(), N, channels, H, W, _stream()),
               "from_cb8_bf16")
    return out
