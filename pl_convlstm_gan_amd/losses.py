"""CombinedLoss with the reference's interface (src/losses/combined_loss.py:6-191).

Four L1-type terms on the generator output ``pred`` [B,T,1,Hh,Wh]:
  point     weighted |pred - obs| at rain-gauge pixels            (combined_loss.py:79-141)
  conserve  |area_mean(pred -> LR grid) - lr_input|               (combined_loss.py:64-74)
  smooth    mean|d/dx| + mean|d/dy|                               (combined_loss.py:146-155)
  temporal  mean|pred[t] - pred[t+1]|                             (combined_loss.py:160-168)

These are small memory-bound reductions over the output tensor; they run as PyTorch-ROCm
ops on the device the prediction lives on and feed ``grad_out`` of the HIP backward pass.
Reference behaviour that is kept on purpose: log weighting uses log1p(obs), so an observation
below -1 makes the total NaN; T == 1 makes the temporal term NaN (mean of an empty tensor);
batched station coordinates use sample 0's coordinates for the whole batch.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F


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
    def forward(self, pred, lr_input, s_coords, s_values, scale_factor=1.0):
        parts = {
            "point": self.point_supervision_loss(pred, s_coords, s_values, scale_factor),
            "conserve": self.conservation_loss(pred, lr_input),
            "smooth": self.gradient_loss(pred),
            "temporal": self.temporal_consistency_loss(pred),
        }
        total = (self.lambda_point * parts["point"] + self.lambda_conserve * parts["conserve"]
                 + self.lambda_smooth * parts["smooth"] + self.lambda_temporal * parts["temporal"])
        return total, parts
