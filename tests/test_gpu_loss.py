"""Fused CombinedLoss kernel (plc_combined_loss, through the C ABI) against the reference goldens
(the five scalars) and against autograd of the CPU oracle's restatement (d total / d pred).

Tolerances: scalars 1e-5 relative (fp32 sums in a different order); the gradient is a sum of at most
eight terms of the form +-lambda/count, so 1e-6 * max|grad| + 1e-9 absolute is an exact-sign check."""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import cpu_oracle as O

pytestmark = pytest.mark.gpu


def _run_fused(gpu, pred, lr, coords, values, scale, **kw):
    from pl_convlstm_gan_amd.losses import CombinedLoss
    mod = CombinedLoss(**kw)
    p = pred.clone().to(gpu).requires_grad_(True)
    total, parts = mod(p, lr.to(gpu), coords.to(gpu), values.to(gpu) if values is not None else None, scale)
    return mod, p, total, parts


def _oracle(pred, lr, coords, values, scale, lam, weighted, strategy):
    p = pred.clone().requires_grad_(True)
    total, parts = O.combined_loss(p, lr, coords, values, scale, lambda_point=lam[0], lambda_conserve=lam[1],
                                   lambda_smooth=lam[2], lambda_temporal=lam[3], use_weighted=weighted,
                                   strategy=strategy)
    return p, total, parts


def _close(a, b, rel=1e-5):
    if math.isnan(b):
        return math.isnan(a)
    return abs(a - b) <= rel * max(1.0, abs(b))


@pytest.mark.parametrize("name", ["loss_log.npz", "loss_strat.npz", "loss_sqrt.npz", "loss_plain.npz",
                                  "loss_negative_nan.npz"])
def test_fused_loss_vs_golden_and_oracle_grad(gpu, name):
    g = load_golden(name)
    lam = [float(v) for v in g["lambdas"]]
    weighted, strategy, scale = bool(g["weighted"]), str(g["strategy"]), float(g["scale"])
    kw = dict(lambda_point=lam[0], lambda_conserve=lam[1], lambda_smooth=lam[2], lambda_temporal=lam[3],
              use_weighted_loss=weighted, weight_strategy=strategy)
    _, p, total, parts = _run_fused(gpu, g["pred"], g["lr"], g["s_coords"], g["s_values"], scale, **kw)
    for k in ("point", "conserve", "smooth", "temporal"):
        assert _close(float(parts[k]), float(g[k])), (k, float(parts[k]), float(g[k]))
    assert _close(float(total.detach()), float(g["total"]))
    if math.isnan(float(g["total"])):
        return
    total.backward()
    po, to, _ = _oracle(g["pred"], g["lr"], g["s_coords"], g["s_values"], scale, lam, weighted, strategy)
    to.backward()
    ref = po.grad
    err = (p.grad.cpu() - ref).abs().max().item()
    assert err <= 1e-6 * ref.abs().max().item() + 1e-9, err


@pytest.mark.parametrize("shape", [
    # B, T, Hh, Wh, H, W, scale, N
    (2, 3, 15, 12, 6, 5, 2.4, 7),      # non-integer area bins (overlapping), Wh % 4 == 0
    (1, 4, 9, 7, 9, 7, 1.0, 5),        # identity grid, scalar path (Wh % 4 != 0)
    (2, 2, 21, 18, 7, 6, 3.0, 6),      # integer bins, scalar path
    (1, 1, 8, 8, 4, 4, 2.0, 3),        # T == 1: temporal term NaN, as the reference
    (2, 5, 64, 96, 8, 12, 8.0, 35),    # the reference's default geometry family (s = 8)
])
def test_fused_loss_vs_oracle(gpu, shape):
    B, T, Hh, Wh, H, W, scale, N = shape
    gen = torch.Generator().manual_seed(B * 1000 + Hh)
    pred = torch.randn(B, T, 1, Hh, Wh, generator=gen) * 3
    lr = torch.rand(B, T, 1, H, W, generator=gen) * 5
    coords = torch.stack([torch.randint(0, H, (N,), generator=gen), torch.randint(0, W, (N,), generator=gen)], 1)
    coords[1] = coords[0]                               # two gauges on one pixel
    coords[2] = torch.tensor([H + 3, 0])                # one outside the grid
    values = torch.rand(B, T, N, generator=gen) * 30
    values[0, 0, 0] = float("nan")
    lam = [1.0, 1.0, 0.1, 0.05]
    _, p, total, parts = _run_fused(gpu, pred, lr, coords, values, scale)
    po, to, parts_o = _oracle(pred, lr, coords, values, scale, lam, True, "log")
    for k in parts_o:
        assert _close(float(parts[k]), float(parts_o[k])), (k, float(parts[k]), float(parts_o[k]))
    assert _close(float(total.detach()), float(to.detach()))
    if math.isnan(float(to)):
        return
    total.backward()
    to.backward()
    err = (p.grad.cpu() - po.grad).abs().max().item()
    assert err <= 1e-6 * po.grad.abs().max().item() + 1e-9, err


def test_fused_loss_shared_values_and_no_stations(gpu):
    gen = torch.Generator().manual_seed(5)
    pred = torch.randn(2, 3, 1, 8, 8, generator=gen)
    lr = torch.rand(2, 3, 1, 4, 4, generator=gen)
    coords = torch.randint(0, 4, (4, 2), generator=gen)
    vals2d = torch.rand(3, 4, generator=gen) * 10          # [T, N]: shared by the batch
    _, p, total, parts = _run_fused(gpu, pred, lr, coords, vals2d, 2.0)
    _, to, parts_o = _oracle(pred, lr, coords, vals2d, 2.0, [1.0, 1.0, 0.1, 0.05], True, "log")
    assert _close(float(parts["point"]), float(parts_o["point"])) and _close(float(total.detach()), float(to.detach()))
    # every observation missing -> the gauge term is exactly 0 and contributes no gradient
    nanv = torch.full((2, 3, 4), float("nan"))
    _, p, total, parts = _run_fused(gpu, pred, lr, coords, nanv, 2.0)
    assert float(parts["point"]) == 0.0
    # a scaled upstream gradient is honoured
    (total * 3.0).backward()
    po, to, _ = _oracle(pred, lr, coords, nanv, 2.0, [1.0, 1.0, 0.1, 0.05], True, "log")
    (to * 3.0).backward()
    assert (p.grad.cpu() - po.grad).abs().max().item() <= 1e-6 * po.grad.abs().max().item() + 1e-9


def test_fused_loss_bit_repeatable_at_bench_size(gpu):
    gen = torch.Generator().manual_seed(9)
    pred = torch.randn(4, 10, 1, 256, 256, generator=gen)
    lr = torch.rand(4, 10, 1, 256, 256, generator=gen) * 10
    coords = torch.randint(0, 256, (35, 2), generator=gen)
    values = torch.rand(4, 10, 35, generator=gen) * 20
    runs = []
    for _ in range(2):
        _, p, total, _ = _run_fused(gpu, pred, lr, coords, values, 1.0)
        total.backward()
        runs.append((float(total), p.grad.clone()))
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1])
    _, to, _ = _oracle(pred, lr, coords, values, 1.0, [1.0, 1.0, 0.1, 0.05], True, "log")
    assert _close(float(runs[0][0]), float(to))


@pytest.mark.parametrize("N", [1500, 2500])
def test_fused_loss_many_gauges_on_few_pixels(gpu, N):
    """Thousands of gauges on a 6x5 grid: every pixel carries a long chain of stations (N = 1500: the
    chains found once per call; N = 2500 is past the kernel's station cache and takes the rescan).  The
    gradient sums must match autograd of the oracle term by term."""
    gen = torch.Generator().manual_seed(N)
    B, T, H, W, s = 2, 2, 6, 5, 2
    pred = torch.randn(B, T, 1, H * s, W * s, generator=gen) * 3
    lr = torch.rand(B, T, 1, H, W, generator=gen) * 5
    coords = torch.stack([torch.randint(0, H + 1, (N,), generator=gen), torch.randint(0, W, (N,), generator=gen)], 1)
    values = torch.rand(B, T, N, generator=gen) * 30
    values[torch.rand(B, T, N, generator=gen) < 0.1] = float("nan")
    _, p, total, parts = _run_fused(gpu, pred, lr, coords, values, float(s))
    po, to, parts_o = _oracle(pred, lr, coords, values, float(s), [1.0, 1.0, 0.1, 0.05], True, "log")
    for k in parts_o:
        assert _close(float(parts[k]), float(parts_o[k]), rel=1e-4), (k, float(parts[k]), float(parts_o[k]))
    total.backward()
    to.backward()
    err = (p.grad.cpu() - po.grad).abs().max().item()
    assert err <= 2e-5 * po.grad.abs().max().item() + 1e-9, err       # sums of up to ~100 terms per pixel
