"""Host-side mirror of the reference's model interface, backed by the HIP engine.

Same class names, constructor signatures, attribute names and ``state_dict`` keys as the
reference's ``src/models`` (convlstm.py:4-14, attention.py:3-26, generator.py:10-92), so that
checkpoints and calling code carry over.  The ``nn.Conv2d`` members are parameter containers
only (identical default initialisation); all arithmetic runs in libplconvlstm on the GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import _lib, ops


def add_coord_channels(x: torch.Tensor) -> torch.Tensor:
    """[B,C,H,W] -> [B,C+2,H,W] with (row, col) = (linspace(0,1,H), linspace(0,1,W)) appended
    (coordconv.py:3-10).  Tensor plumbing only; inside the rollout the two channels are
    synthesised by the stem kernel and never materialised."""
    B, _, H, W = x.shape
    row = torch.linspace(0, 1, H, device=x.device).view(1, 1, H, 1).expand(B, 1, H, W)
    col = torch.linspace(0, 1, W, device=x.device).view(1, 1, 1, W).expand(B, 1, H, W)
    return torch.cat((x, row, col), dim=1)


class ConvLSTMCell(nn.Module):
    """convlstm.py:4-28.  state_dict: conv.weight [4Ch, Cin+Ch, 3, 3], conv.bias [4Ch]."""

    def __init__(self, input_dim, hidden_dim, kernel_size=3, bias=True):
        super().__init__()
        self.hidden_dim = hidden_dim
        self.conv = nn.Conv2d(input_dim + hidden_dim, 4 * hidden_dim, kernel_size,
                              padding=kernel_size // 2, bias=bias)

    def forward(self, x, h_cur, c_cur):
        if self.conv.kernel_size != (3, 3):
            raise NotImplementedError("only kernel_size=3 is compiled")
        weight, bias = self.conv.weight, self.conv.bias
        home = x.device
        if not x.is_cuda:
            # CPU-resident caller (the reference's tests build everything on the CPU): the operands are
            # staged onto the GPU, the HIP kernel runs there, the results are copied back.  No CPU
            # arithmetic exists; .to() is differentiable, so gradients find their way home.
            dev = ops.staging_device()
            x, h_cur, c_cur = x.to(dev), h_cur.to(dev), c_cur.to(dev)
        if weight.device != x.device:
            weight, bias = weight.to(x.device), None if bias is None else bias.to(x.device)
        if torch.is_grad_enabled() and any(t.requires_grad for t in (x, h_cur, c_cur, weight)):
            from .autograd import cell_with_grad
            h2, c2 = cell_with_grad(x, h_cur, c_cur, weight, bias)
        else:
            h2, c2 = ops.convlstm_cell_forward(x, h_cur, c_cur, weight.detach(), None if bias is None else bias.detach())
        return h2.to(home), c2.to(home)


class UpsampleBlock(nn.Module):
    """generator.py:10-28: conv3x3(C -> 4C) + PixelShuffle(2) + ReLU.  state_dict: conv.*"""

    def __init__(self, in_channels, upscale_factor=2):
        super().__init__()
        self.upscale_factor = upscale_factor
        self.conv = nn.Conv2d(in_channels, in_channels * upscale_factor ** 2, kernel_size=3, padding=1)
        self.pixel_shuffle = nn.PixelShuffle(upscale_factor)
        self.relu = nn.ReLU(inplace=True)


class _Attention(nn.Module):
    def __init__(self, in_channels, aux_channels):
        super().__init__()
        self.conv = nn.Sequential(
            nn.Conv2d(aux_channels, in_channels // 2, 3, padding=1),
            nn.ReLU(inplace=True),
            nn.Conv2d(in_channels // 2, in_channels, 1),
            nn.Sigmoid(),
        )


class DEMAttention(_Attention):
    """attention.py:3-13.  state_dict: conv.0.*, conv.2.*"""

    def __init__(self, in_channels, dem_channels=1):
        super().__init__(in_channels, dem_channels)


class LUAttention(_Attention):
    """attention.py:16-26."""

    def __init__(self, in_channels, lu_channels=1):
        super().__init__(in_channels, lu_channels)


def _f32(x) -> float:
    return float(np.float32(x))


class Generator(nn.Module):
    """generator.py:31-205 on the HIP rollout engine.

    Behaviour kept from the reference: upsample blocks are created lazily inside the first
    forward (so an optimizer built earlier does not hold them); constructing with neither
    ``target_grid_size`` nor ``scale_factor`` leaves those attributes unset.
    """

    def __init__(self, in_channels=1, dem_channels=1, lu_channels=0, hidden_dims=[32, 64],
                 target_grid_size=None, scale_factor=None):
        super().__init__()
        self.hidden_dims = hidden_dims
        self.lu_channels = lu_channels
        if target_grid_size is not None:
            self.target_grid_size = target_grid_size
            self.scale_factor = None
            self.target_size = None
        elif scale_factor is not None:
            self.scale_factor = scale_factor
            self.target_grid_size = None
            self.target_size = None
        self.init_conv = nn.Conv2d(in_channels + 2, hidden_dims[0], kernel_size=3, padding=1)
        self.cell1 = ConvLSTMCell(hidden_dims[0], hidden_dims[0])
        self.cell2 = ConvLSTMCell(hidden_dims[0], hidden_dims[1])
        self.dem_attn = DEMAttention(hidden_dims[1], dem_channels)
        self.lu_attn = LUAttention(hidden_dims[1], lu_channels)
        self.upsample_blocks = None
        self.post_process = nn.Sequential(
            nn.Conv2d(hidden_dims[1], 32, 3, padding=1),
            nn.ReLU(inplace=True),
            nn.Conv2d(32, 1, 3, padding=1),
        )
        self._in_channels = in_channels
        self._dem_channels = dem_channels
        # operand type of the two gate convolutions: "bf16", or "fp8" (e4m3 operands on the block-scaled
        # matrix instruction; hidden_dims [16,32] only).  Not a constructor argument: the reference has none.
        self.gate_dtype = "bf16"
        self._ws = {}
        self._tabs = {}

    # generator.py:73-92
    def _setup_upsample_blocks(self, scale_factor, device):
        if scale_factor <= 1:
            self.upsample_blocks = nn.ModuleList()
            return 1
        blocks = []
        cur = scale_factor
        while cur > 1:
            if cur < 2:
                break
            blocks.append(UpsampleBlock(self.hidden_dims[1], 2))
            cur //= 2
        self.upsample_blocks = nn.ModuleList(blocks).to(device)
        return scale_factor / (2 ** len(blocks))

    # ------------------------------------------------------------------------------------
    def _geometry(self, H, W, input_grid_size, device):
        """generator.py:106-140 -> (n_blocks, remaining, final_H, final_W)."""
        if self.target_grid_size is not None and input_grid_size is not None:
            gx, gy = input_grid_size
            tx, ty = self.target_grid_size
            sw, sh = gx / tx, gy / ty
            self.target_size = (int(H * sh), int(W * sw))
            scale = max(sh, sw)
        elif self.scale_factor is not None:
            scale = self.scale_factor
            self.target_size = None
        else:
            scale = 1
            self.target_size = None
        if self.upsample_blocks is None:
            remaining = self._setup_upsample_blocks(int(scale), device)
        else:
            remaining = scale / (2 ** len(self.upsample_blocks))
        if self.target_size is not None:
            fH, fW = self.target_size
        else:
            fH, fW = int(H * scale), int(W * scale)
        return len(self.upsample_blocks), remaining, fH, fW

    def _tables(self, H, W, device):
        key = (H, W, str(device))
        if key not in self._tabs:
            self._tabs[key] = (torch.linspace(0, 1, H).to(device), torch.linspace(0, 1, W).to(device))
        return self._tabs[key]

    def _describe(self, rain_lr, dem, lu, input_grid_size, save_for_backward=False):
        B, T, Cin, H, W = rain_lr.shape
        if Cin != 1 or self._in_channels != 1:
            raise NotImplementedError("the engine is built for in_channels=1 (rain only)")
        if self._dem_channels != 1 or dem.shape[1] != 1:
            raise NotImplementedError("the engine is built for dem_channels=1")
        n_blocks, remaining, fH, fW = self._geometry(H, W, input_grid_size, next(self.parameters()).device)
        if lu.shape[1] != self.lu_channels or self.lu_channels < 1:
            raise RuntimeError(
                f"lu has {lu.shape[1]} channels but the model was built with lu_channels={self.lu_channels}")
        d = _lib.RolloutDesc()
        d.B, d.T, d.H, d.W = B, T, H, W
        d.C1, d.C2 = int(self.hidden_dims[0]), int(self.hidden_dims[1])
        d.L = int(self.lu_channels)
        d.dem_H, d.dem_W = dem.shape[-2:]
        d.lu_H, d.lu_W = lu.shape[-2:]
        d.n_blocks = n_blocks
        uH, uW = H << n_blocks, W << n_blocks
        if remaining > 1:
            # F.interpolate(scale_factor=r): out = floor(in * r), coordinate scale = 1 / r
            d.mid_H, d.mid_W = int(np.floor(float(uH) * remaining)), int(np.floor(float(uW) * remaining))
            d.mid_scale_h = d.mid_scale_w = _f32(1.0 / remaining)
        else:
            d.mid_H, d.mid_W = uH, uW
            d.mid_scale_h = d.mid_scale_w = 1.0
        if self.target_size is not None:
            d.out_H, d.out_W = self.target_size
        else:
            d.out_H, d.out_W = d.mid_H, d.mid_W
        d.out_scale_h = _f32(np.float32(d.mid_H) / np.float32(d.out_H))
        d.out_scale_w = _f32(np.float32(d.mid_W) / np.float32(d.out_W))
        if (d.out_H, d.out_W) != (fH, fW):
            # the reference multiplies a (out_H, out_W) feature map with (fH, fW) gates and fails
            raise RuntimeError(
                f"The size of tensor a ({d.out_W}) must match the size of tensor b ({fW}) at non-singleton dimension 3")
        d.save_for_backward = 1 if save_for_backward else 0
        if self.gate_dtype not in ("bf16", "fp8"):
            raise ValueError(f"gate_dtype must be 'bf16' or 'fp8', not {self.gate_dtype!r}")
        d.gate_fp8 = 1 if self.gate_dtype == "fp8" else 0
        return d, (fH, fW)

    def _bind(self, d, rain_lr, dem, lu, out, params):
        """``params``: the parameter tensors in autograd._param_list order, on the inputs' device (the
        module's own parameters, or their staged copies when the module lives on the CPU)."""
        row, col = self._tables(d.H, d.W, rain_lr.device)
        d.rain, d.dem, d.lu = rain_lr.data_ptr(), dem.data_ptr(), lu.data_ptr()
        d.row_tab, d.col_tab = row.data_ptr(), col.data_ptr()
        keep = [t.detach().contiguous() for t in params]      # (parameters are contiguous already)
        it = iter(keep)
        P = lambda: next(it).data_ptr()  # noqa: E731
        d.init_w, d.init_b = P(), P()
        d.cell1_w, d.cell1_b = P(), P()
        d.cell2_w, d.cell2_b = P(), P()
        d.dem0_w, d.dem0_b = P(), P()
        d.dem2_w, d.dem2_b = P(), P()
        d.lu0_w, d.lu0_b = P(), P()
        d.lu2_w, d.lu2_b = P(), P()
        for k in range(len(self.upsample_blocks)):
            d.up_w[k], d.up_b[k] = P(), P()
        d.post0_w, d.post0_b = P(), P()
        d.post2_w, d.post2_b = P(), P()
        d.out = out.data_ptr()
        return keep

    def _workspace(self, d, device, fresh=False):
        L = _lib.lib()
        nbytes = L.plc_generator_workspace_bytes(C.byref(d))
        if nbytes == 0:
            _lib.check(-2 if ("hidden_dims" in L.plc_last_error().decode()) else -1, "Generator.forward")
        if fresh:
            # training: the workspace holds the activations saved for backward and is owned by
            # the autograd node, so a second forward cannot overwrite it
            ws = ops.workspace(nbytes, device)
        elif device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            # under stream capture the buffer comes from the graph's own memory pool and lives as long as the
            # graph: the cached one below is dropped as soon as a call of another size comes along, and a
            # replay would then write into memory that belongs to someone else
            ws = ops.workspace(nbytes, device)
        else:
            key = (str(device), nbytes)
            ws = self._ws.get(key)
            if ws is None:
                self._ws.clear()
                ws = ops.workspace(nbytes, device)
                self._ws[key] = ws
        d.workspace = ws.data_ptr()
        d.workspace_bytes = nbytes
        return ws

    def _bwd_workspace(self, nbytes, device):
        if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            return ops.workspace(nbytes, device)       # (as in _workspace: a captured graph owns its buffers)
        key = ("bwd", str(device), nbytes)
        ws = self._ws.get(key)
        if ws is None:
            for k in [k for k in self._ws if k[0] == "bwd"]:
                del self._ws[k]
            ws = ops.workspace(nbytes, device)
            self._ws[key] = ws
        return ws

    def forward(self, rain_lr, dem, lu, input_grid_size=None):
        """rain_lr [B,T,1,H,W], dem [B,1,Hd,Wd], lu [B,L,Hl,Wl] -> [B,T,1,Hh,Wh] (fp32)."""
        from .autograd import _param_list, rollout_with_grad
        home = rain_lr.device if isinstance(rain_lr, torch.Tensor) else None
        if isinstance(rain_lr, torch.Tensor) and not rain_lr.is_cuda:
            # CPU-resident caller (the reference's test/test_model.py builds model and inputs on the
            # CPU): inputs and parameters are staged onto the GPU, the HIP rollout runs there and the
            # result is copied back.  There is no CPU arithmetic path; without a GPU this raises.
            dev = ops.staging_device()
            rain_lr, dem, lu = rain_lr.to(dev), dem.to(dev), lu.to(dev)
        rain_lr = ops._require_dev(rain_lr, "rain_lr")
        dem = ops._require_dev(dem, "dem")
        lu = ops._require_dev(lu, "lu")
        dev = rain_lr.device
        # the lazy blocks are created where the module lives, from the caller's RNG, as in the reference
        module_dev = next(self.parameters()).device
        self._geometry(rain_lr.shape[-2], rain_lr.shape[-1], input_grid_size, module_dev)
        own = _param_list(self)
        # (inputs with requires_grad get their gradients too, as the reference's autograd returns them)
        needs_grad = torch.is_grad_enabled() and (any(p.requires_grad for p in own)
                                                  or any(t.requires_grad for t in (rain_lr, dem, lu)))
        params = [p if p.device == dev else p.to(dev) for p in own]   # .to() is differentiable
        if needs_grad:
            out = rollout_with_grad(self, rain_lr, dem, lu, input_grid_size, params)
        else:
            out = self._forward_impl(rain_lr, dem, lu, input_grid_size, False, [p.detach() for p in params])[0]
        return out if out.device == home else out.to(home)

    def _forward_impl(self, rain_lr, dem, lu, input_grid_size, save_for_backward, params):
        d, _ = self._describe(rain_lr, dem, lu, input_grid_size, save_for_backward)
        out = torch.empty((d.B, d.T, 1, d.out_H, d.out_W), dtype=torch.float32, device=rain_lr.device)
        ws = self._workspace(d, rain_lr.device, fresh=save_for_backward)
        keep = self._bind(d, rain_lr, dem, lu, out, params)
        rc = _lib.lib().plc_generator_rollout_fwd(C.byref(d), ops._stream())
        _lib.check(rc, "plc_generator_rollout_fwd")
        return out, d, (ws, keep)
