"""Pins the CPU oracle (oracle/cpu_oracle.py) to golden vectors produced by the reference.

Tolerance: fp32 restatement vs reference <= 1e-5 absolute (different op grouping only:
the attention gates are hoisted out of the T loop, the arithmetic is otherwise the same).
"""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden, master_params
from oracle import cpu_oracle as O

ATOL = 1e-5

GEN_CASES = {
    # name: (n_blocks, forward kwargs)
    "gen_s1.npz": (0, dict(scale_factor=1)),
    "gen_s1_odd.npz": (0, dict(scale_factor=1)),
    "gen_s2.npz": (1, dict(scale_factor=2)),
    "gen_s8.npz": (3, dict(scale_factor=8)),
    "gen_grid.npz": (3, dict(target_grid_size=(500, 500), input_grid_size=(4000, 4000))),
    "gen_s3.npz": (1, dict(scale_factor=3)),
    "gen_s6.npz": (2, dict(scale_factor=6)),
    "gen_s2_aux.npz": (1, dict(scale_factor=2)),
    "gen_s4_native.npz": (2, dict(scale_factor=4)),
    "gen_s3_grads.npz": (1, dict(scale_factor=3)),
    "gen_grid_grads.npz": (3, dict(target_grid_size=(500, 600), input_grid_size=(4000, 4000))),
    "gen_grid_mix_grads.npz": (1, dict(target_grid_size=(1000, 900), input_grid_size=(3000, 3000))),
}


@pytest.mark.parametrize("tag", ["16_16", "16_32"])
def test_cell_forward_and_grads(tag):
    g = load_golden(f"cell_{tag}.npz")
    x, h, c, w, b = (g[k].clone().requires_grad_(True) for k in ("x", "h", "c", "weight", "bias"))
    h2, c2 = O.convlstm_cell(x, h, c, w, b)
    assert torch.allclose(h2, g["h_next"], atol=ATOL, rtol=0)
    assert torch.allclose(c2, g["c_next"], atol=ATOL, rtol=0)
    ((h2 * g["rh"]).sum() + (c2 * g["rc"]).sum()).backward()
    for name, t in (("dx", x), ("dh", h), ("dc", c), ("dweight", w), ("dbias", b)):
        ref = g[name]
        assert torch.allclose(t.grad, ref, atol=1e-4 * max(1.0, float(ref.abs().max())), rtol=0), name


@pytest.mark.parametrize("name", sorted(GEN_CASES))
def test_generator_forward(name):
    n_blocks, kw = GEN_CASES[name]
    g = load_golden(name)
    assert int(g["n_blocks"]) == n_blocks
    out = O.generator_forward(master_params(n_blocks), g["rain"], g["dem"], g["lu"], **kw)
    assert out.shape == g["out"].shape
    assert float((out - g["out"]).abs().max()) <= ATOL


@pytest.mark.parametrize("name", ["gen_s1.npz", "gen_s2.npz", "gen_s3_grads.npz", "gen_grid_grads.npz",
                                  "gen_grid_mix_grads.npz"])
def test_generator_param_grads(name):
    n_blocks, kw = GEN_CASES[name]
    g = load_golden(name)
    loss, grads, _ = O.grads_of(master_params(n_blocks), g["rain"], g["dem"], g["lu"],
                                g["s_coords"], g["s_values"], fwd_kw=kw)
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * max(1.0, abs(float(g["loss"])))
    checked = 0
    for k, v in g.items():
        if not k.startswith("grad."):
            continue
        mine = grads[k[5:]]
        tol = 1e-4 * max(1e-3, float(v.abs().max()))
        assert float((mine - v).abs().max()) <= tol, k
        checked += 1
    assert checked >= 7


def test_bf16_emulation_is_close_to_fp32():
    """The bf16 storage-rounding emulation stays within 2e-2 of |out|max of the fp32 reference
    (the stated bf16 tolerance of the HIP pipeline)."""
    g = load_golden("gen_s1.npz")
    out = O.generator_forward(master_params(0), g["rain"], g["dem"], g["lu"], scale_factor=1, emulate="bf16")
    ref = g["out"]
    assert float((out - ref).abs().max()) <= 2e-2 * float(ref.abs().max())


@pytest.mark.parametrize("name", ["loss_log.npz", "loss_strat.npz", "loss_sqrt.npz", "loss_plain.npz",
                                  "loss_negative_nan.npz"])
def test_combined_loss(name):
    g = load_golden(name)
    lam = g["lambdas"]
    total, parts = O.combined_loss(g["pred"], g["lr"], g["s_coords"], g["s_values"], float(g["scale"]),
                                   lambda_point=float(lam[0]), lambda_conserve=float(lam[1]),
                                   lambda_smooth=float(lam[2]), lambda_temporal=float(lam[3]),
                                   use_weighted=bool(g["weighted"]), strategy=str(g["strategy"]))
    for k in ("point", "conserve", "smooth", "temporal"):
        ref = float(g[k])
        mine = float(parts[k])
        if math.isnan(ref):
            assert math.isnan(mine), k
        else:
            assert abs(mine - ref) <= 1e-5 * max(1.0, abs(ref)), k
    if math.isnan(float(g["total"])):
        assert math.isnan(float(total))          # log1p of obs < -1: reference behaviour kept
    else:
        assert abs(float(total) - float(g["total"])) <= 1e-5 * max(1.0, abs(float(g["total"])))


def test_upsample_plan():
    # generator.py:73-92 (shapes probed on the reference: s=3 -> 1 block x1.5, s=6 -> 2 blocks x1.5)
    assert O.upsample_plan(1) == (0, 1)
    assert O.upsample_plan(2) == (1, 1.0)
    assert O.upsample_plan(3) == (1, 1.5)
    assert O.upsample_plan(4) == (2, 1.0)
    assert O.upsample_plan(6) == (2, 1.5)
    assert O.upsample_plan(8) == (3, 1.0)


def test_train_step_order():
    """trainer.py:290-315 order with Adam built before the lazy blocks exist: the upsample
    weights receive gradients, count in the clip norm, and are not updated."""
    g = load_golden("step_s2.npz")
    assert bool(g["upsample_unchanged"])
    p = master_params(1)
    loss, grads, _ = O.grads_of(p, g["rain"], g["dem"], g["lu"], g["s_coords"], g["s_values"],
                                fwd_kw=dict(scale_factor=2))
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * max(1.0, abs(float(g["loss"])))
    names = list(p.keys())
    norm = math.sqrt(sum(float((grads[k].double() ** 2).sum()) for k in names))
    assert abs(norm - float(g["grad_norm"])) <= 1e-4 * float(g["grad_norm"])
    coef = O.clip_coef([grads[k] for k in names], 0.5)
    # Adam, first step: delta = -lr * g / (|g| + eps)  (bias-corrected m/sqrt(v))
    for k, after in g.items():
        if not k.startswith("after."):
            continue
        key = k[6:]
        if key.startswith("upsample_blocks."):
            assert torch.equal(after, p[key])
            continue
        gk = grads[key] * coef
        expect = p[key] - 5e-4 * gk / (gk.abs() + 1e-8)
        # entries whose gradient is ~0 are ill-conditioned under Adam's sign-like first step
        mask = gk.abs() > 1e-7
        assert float((expect - after)[mask].abs().max()) <= 2e-5, key


WIDE_CASES = {"genw_default_s2.npz": ([32, 64], 2), "genw_8_24_s1.npz": ([8, 24], 1),
              "genw_10_20_s2.npz": ([10, 20], 2), "genw_12_30_s1.npz": ([12, 30], 1)}


def wide_params(g):
    return {k[6:]: v for k, v in g.items() if k.startswith("param.")}


@pytest.mark.parametrize("name", sorted(WIDE_CASES))
def test_other_layer_widths(name):
    """Generators of other widths (the reference's class default [32,64]; [8,24]) with their own
    parameters: forward, loss and parameter gradients of the oracle against the reference's."""
    hidden, s = WIDE_CASES[name]
    g = load_golden(name)
    assert list(g["hidden"]) == hidden
    p = wide_params(g)
    out = O.generator_forward(p, g["rain"], g["dem"], g["lu"], scale_factor=s)
    assert float((out - g["out"]).abs().max()) <= ATOL
    loss, grads, _ = O.grads_of(p, g["rain"], g["dem"], g["lu"], g["s_coords"], g["s_values"], fwd_kw=dict(scale_factor=s))
    assert abs(float(loss) - float(g["loss"])) <= 1e-5 * max(1.0, abs(float(g["loss"])))
    checked = 0
    for k, v in g.items():
        if k.startswith("grad."):
            assert float((grads[k[5:]] - v).abs().max()) <= 1e-4 * max(1e-3, float(v.abs().max())), k
            checked += 1
    assert checked >= 9
