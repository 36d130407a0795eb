"""Autograd glue for the rollout (training path)."""
from __future__ import annotations


def rollout_with_grad(model, rain_lr, dem, lu, input_grid_size):
    raise NotImplementedError(
        "the HIP backward pass of the rollout is not built yet: call the model under torch.no_grad() "
        "(there is no PyTorch fallback)")
