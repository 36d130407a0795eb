"""CombinedLoss with the reference's interface (src/losses/combined_loss.py:6-191).

Four L1-type terms on the generator output ``pred`` [B,T,1,Hh,Wh]:
  point     weighted |pred - obs| at rain-gauge pixels            (combined_loss.py:79-141)
  conserve  |area_mean(pred -> LR grid) - lr_input|               (combined_loss.py:64-74)
  smooth    mean|d/dx| + mean|d/dy|                               (combined_loss.py:146-155)
  temporal  mean|pred[t] - pred[t+1]|                             (combined_loss.py:160-168)

On the GPU all four terms and d(total)/d(pred) come from ONE pass over the prediction in
libplconvlstm (``plc_combined_loss``: three launches instead of ~60 elementwise / reduction
kernels), wrapped in an autograd function so that ``total.backward()`` feeds ``grad_out`` of the HIP
rollout backward.  There is no fallback on the GPU: a missing library raises.  CPU tensors take
the composed-operator formulation below (the same arithmetic written with torch ops), which is
what the host-side CPU tests and the gloo data-parallel tests run.
Reference behaviour that is kept on purpose: log weighting uses log1p(obs), so an observation
below -1 makes the total NaN; T == 1 makes the temporal term NaN (mean of an empty tensor);
batched station coordinates use sample 0's coordinates for the whole batch.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn
import torch.nn.functional as F

_WEIGHT_MODES = {"log": 1, "sqrt": 2, "stratified": 3}


class _FusedLossFn(torch.autograd.Function):
    """total = plc_combined_loss(pred, ...); backward returns the gradient computed in the same pass."""

    @staticmethod
    def forward(ctx, pred, lr_input, coords, values, scale_factor, lambdas, weight_mode, want_grad):
        from . import _lib, ops
        L = _lib.lib()
        B, T, _, Hh, Wh = pred.shape
        H, W = lr_input.shape[-2:]
        pred_c = pred.detach().contiguous().float()
        lr_c = lr_input.detach().contiguous().float()
        d = _lib.LossDesc()
        d.B, d.T, d.out_H, d.out_W, d.H, d.W = B, T, Hh, Wh, H, W
        d.N = 0 if coords is None else coords.shape[0]
        d.values_per_batch = 1 if (values is not None and values.dim() == 3) else 0
        d.weight_mode = weight_mode
        d.scale_factor = float(scale_factor)
        d.lambda_point, d.lambda_conserve, d.lambda_smooth, d.lambda_temporal = lambdas
        d.pred, d.lr_input = pred_c.data_ptr(), lr_c.data_ptr()
        if d.N:
            coords = coords.detach().to(device=pred.device, dtype=torch.int64).contiguous()
            values = values.detach().to(device=pred.device, dtype=torch.float32).contiguous()
            d.s_coords, d.s_values = coords.data_ptr(), values.data_ptr()
        out = torch.empty(5, dtype=torch.float32, device=pred.device)
        grad = torch.empty_like(pred_c) if want_grad else None
        d.losses = out.data_ptr()
        d.grad_pred = grad.data_ptr() if want_grad else None
        nbytes = L.plc_combined_loss_workspace_bytes(C.byref(d))
        ws = ops.workspace(max(nbytes, 256), pred.device)
        d.workspace, d.workspace_bytes = ws.data_ptr(), nbytes
        _lib.check(L.plc_combined_loss(C.byref(d), ops._stream()), "plc_combined_loss")
        ctx.grad = grad
        ctx.mark_non_differentiable(out)
        return out[0], out

    @staticmethod
    def backward(ctx, g_total, _g_parts):
        grad = ctx.grad
        if grad is None:
            return (None,) * 8
        return grad * g_total, None, None, None, None, None, None, None


class CombinedLoss(nn.Module):
    def __init__(self, lambda_point=1.0, lambda_conserve=1.0, lambda_smooth=0.1, lambda_temporal=0.05,
                 use_weighted_loss=True, weight_strategy='log'):
        super().__init__()
        self.lambda_point = lambda_point
        self.lambda_conserve = lambda_conserve
        self.lambda_smooth = lambda_smooth
        self.lambda_temporal = lambda_temporal
        self.l1 = nn.L1Loss()
        self.use_weighted_loss = use_weighted_loss
        self.weight_strategy = weight_strategy

    # combined_loss.py:23-59
    def compute_sample_weights(self, rain_values):
        if not self.use_weighted_loss:
            return torch.ones_like(rain_values)
        kind = self.weight_strategy
        if kind == 'log':
            return torch.log1p(rain_values) + 1.0
        if kind == 'sqrt':
            return torch.sqrt(rain_values) + 1.0
        if kind == 'stratified':
            w = torch.ones_like(rain_values)
            for threshold, value in ((10, 2.0), (25, 3.0), (50, 5.0)):
                w = w.masked_fill(rain_values >= threshold, value)
            return w
        return torch.ones_like(rain_values)

    # combined_loss.py:64-74
    def conservation_loss(self, pred, lr_input):
        B, T, C, H, W = pred.shape
        lr_hw = tuple(lr_input.shape[-2:])
        coarse = F.interpolate(pred.reshape(B * T, C, H, W), size=lr_hw, mode="area")
        return self.l1(coarse.reshape(B, T, C, *lr_hw), lr_input)

    # combined_loss.py:79-141
    def point_supervision_loss(self, pred, s_coords, s_values, scale_factor=1.0):
        zero = lambda: torch.tensor(0.0, device=pred.device)  # noqa: E731
        if s_values is None or s_coords.numel() == 0:
            return zero()
        B, T, _, H, W = pred.shape
        coords = s_coords[0] if s_coords.dim() == 3 else s_coords
        hr = ((coords.float() + 0.5) * scale_factor - 0.5).long()      # truncation, as the reference
        rows, cols = hr[:, 0], hr[:, 1]
        inside = (rows >= 0) & (rows < H) & (cols >= 0) & (cols < W)
        rows, cols = rows[inside], cols[inside]
        if rows.numel() == 0:
            return zero()
        at_stations = pred[:, :, 0][:, :, rows, cols]                 # [B, T, n_valid]
        if s_values.dim() == 3:
            obs = s_values[:, :, inside]
        else:
            obs = s_values[:, inside].unsqueeze(0).expand(B, -1, -1)
        seen = ~torch.isnan(obs)
        if int(seen.sum()) == 0:
            return zero()
        target = obs[seen]
        return ((at_stations[seen] - target).abs() * self.compute_sample_weights(target)).mean()

    # combined_loss.py:146-155
    def gradient_loss(self, pred):
        along_x = (pred[..., :, 1:] - pred[..., :, :-1]).abs().mean()
        along_y = (pred[..., 1:, :] - pred[..., :-1, :]).abs().mean()
        return along_x + along_y

    # combined_loss.py:160-168
    def temporal_consistency_loss(self, pred):
        return (pred[:, 1:] - pred[:, :-1]).abs().mean()

    # combined_loss.py:173-191
    def _fused(self, pred, lr_input, s_coords, s_values, scale_factor):
        """GPU path (combined_loss.py:173-191 in one pass); same return contract as forward."""
        coords = values = None
        if s_values is not None and s_coords is not None and s_coords.numel() > 0:
            coords = s_coords[0] if s_coords.dim() == 3 else s_coords
            values = s_values
        mode = _WEIGHT_MODES.get(self.weight_strategy, 0) if self.use_weighted_loss else 0
        lambdas = (float(self.lambda_point), float(self.lambda_conserve), float(self.lambda_smooth),
                   float(self.lambda_temporal))
        want_grad = torch.is_grad_enabled() and pred.requires_grad
        total, out = _FusedLossFn.apply(pred, lr_input, coords, values, float(scale_factor), lambdas, mode, want_grad)
        return total, {"point": out[1], "conserve": out[2], "smooth": out[3], "temporal": out[4]}

    def forward(self, pred, lr_input, s_coords, s_values, scale_factor=1.0):
        if pred.is_cuda:
            return self._fused(pred, lr_input, s_coords, s_values, scale_factor)
        parts = {
            "point": self.point_supervision_loss(pred, s_coords, s_values, scale_factor),
            "conserve": self.conservation_loss(pred, lr_input),
            "smooth": self.gradient_loss(pred),
            "temporal": self.temporal_consistency_loss(pred),
        }
        total = (self.lambda_point * parts["point"] + self.lambda_conserve * parts["conserve"]
                 + self.lambda_smooth * parts["smooth"] + self.lambda_temporal * parts["temporal"])
        return total, parts
