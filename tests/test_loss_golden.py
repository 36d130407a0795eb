"""CombinedLoss (the drop-in module) against the reference's golden values; runs on CPU."""
import math

import pytest
import torch

from conftest import load_golden
from src.losses.combined_loss import CombinedLoss


@pytest.mark.parametrize("name", ["loss_log.npz", "loss_strat.npz", "loss_sqrt.npz", "loss_plain.npz",
                                  "loss_negative_nan.npz"])
def test_combined_loss_module(name):
    g = load_golden(name)
    lam = g["lambdas"]
    mod = CombinedLoss(lambda_point=float(lam[0]), lambda_conserve=float(lam[1]), lambda_smooth=float(lam[2]),
                       lambda_temporal=float(lam[3]), use_weighted_loss=bool(g["weighted"]),
                       weight_strategy=str(g["strategy"]))
    total, parts = mod(g["pred"], g["lr"], g["s_coords"], g["s_values"], float(g["scale"]))
    assert set(parts) == {"point", "conserve", "smooth", "temporal"}
    for k in parts:
        ref, mine = float(g[k]), float(parts[k])
        if math.isnan(ref):
            assert math.isnan(mine), k
        else:
            assert abs(mine - ref) <= 1e-5 * max(1.0, abs(ref)), k
    ref = float(g["total"])
    if math.isnan(ref):
        assert math.isnan(float(total))
    else:
        assert abs(float(total) - ref) <= 1e-5 * max(1.0, abs(ref))


def test_reference_loss_fix_checks():
    """test/test_loss_fix.py:38-67, 110-138 of the reference (inequality pins)."""
    torch.manual_seed(0)
    mod = CombinedLoss()
    assert (mod.lambda_point, mod.lambda_conserve, mod.lambda_smooth, mod.lambda_temporal) == (1.0, 1.0, 0.1, 0.05)
    pred = torch.rand(2, 5, 1, 120, 96)
    lr = torch.rand(2, 5, 1, 15, 12)
    assert float(mod.conservation_loss(pred, lr)) >= 0
    const = torch.ones(2, 5, 1, 32, 32)
    noise = torch.randn(2, 5, 1, 32, 32)
    assert float(mod.gradient_loss(const)) == 0.0 < float(mod.gradient_loss(noise))
    # T == 1: the temporal term is the mean of an empty tensor (NaN), as in the reference
    assert math.isnan(float(mod.temporal_consistency_loss(torch.rand(2, 1, 1, 8, 8))))
