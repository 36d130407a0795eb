"""GPU parity of the HIP forward path (called through the C ABI) against the CPU oracle and the
reference's golden vectors.

Tolerances (the engine computes in bf16 with fp32 accumulation and an fp32 cell state):
  * vs the fp32 reference goldens / fp32 oracle:  max|diff| <= 2e-2 * max|ref|   (stated bf16 tolerance)
  * vs the oracle's bf16-storage emulation:       max|diff| <= 4e-3 * max|ref|   (only accumulation
    order and rare 1-ulp bf16 rounding flips differ)
"""
import numpy as np
import pytest
import torch

from conftest import load_golden, master_params
from oracle import cpu_oracle as O

pytestmark = pytest.mark.gpu

LOOSE = 2e-2
TIGHT = 4e-3

GEN_CASES = {
    "gen_s1.npz": (0, dict(scale_factor=1), {}),
    "gen_s1_odd.npz": (0, dict(scale_factor=1), {}),
    "gen_s2.npz": (1, dict(scale_factor=2), {}),
    "gen_s8.npz": (3, dict(scale_factor=8), {}),
    "gen_grid.npz": (3, dict(target_grid_size=(500, 500)), dict(input_grid_size=(4000, 4000))),
    "gen_s3.npz": (1, dict(scale_factor=3), {}),
    "gen_s6.npz": (2, dict(scale_factor=6), {}),
    "gen_s2_aux.npz": (1, dict(scale_factor=2), {}),
    "gen_s4_native.npz": (2, dict(scale_factor=4), {}),
}


def relmax(a, b):
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-12)


def build_model(gpu, n_blocks, ctor):
    from pl_convlstm_gan_amd import Generator
    from pl_convlstm_gan_amd.models import UpsampleBlock
    m = Generator(hidden_dims=[16, 32], lu_channels=5, **ctor)
    # materialise the lazy blocks the way the first forward would, then load the golden parameters
    m.upsample_blocks = torch.nn.ModuleList([UpsampleBlock(32, 2) for _ in range(n_blocks)])
    m.load_state_dict(master_params(n_blocks), strict=True)
    return m.to(gpu)


@pytest.mark.parametrize("tag", ["16_16", "16_32"])
def test_cell_forward(gpu, tag):
    from pl_convlstm_gan_amd import ConvLSTMCell
    g = load_golden(f"cell_{tag}.npz")
    cin, ch = g["x"].shape[1], g["h"].shape[1]
    cell = ConvLSTMCell(cin, ch)
    cell.load_state_dict({"conv.weight": g["weight"], "conv.bias": g["bias"]})
    cell = cell.to(gpu)
    with torch.no_grad():
        h2, c2 = cell(g["x"].to(gpu), g["h"].to(gpu), g["c"].to(gpu))
    h2, c2 = h2.cpu(), c2.cpu()
    assert relmax(h2, g["h_next"]) <= LOOSE
    assert relmax(c2, g["c_next"]) <= LOOSE
    he, ce = O.convlstm_cell(g["x"], g["h"], g["c"], g["weight"], g["bias"], emulate="bf16")
    # fp32 cell state: only accumulation order differs
    assert float((c2 - ce).abs().max()) <= 5e-5
    # h' is stored as bf16: equal to the rounded oracle value up to one bf16 ulp
    assert float(((h2 - O.bf16_round(he)).abs() - (2.0 ** -7) * he.abs()).max()) <= 1e-6


@pytest.mark.parametrize("name", sorted(GEN_CASES))
def test_generator_forward_golden(gpu, name):
    n_blocks, ctor, fkw = GEN_CASES[name]
    g = load_golden(name)
    m = build_model(gpu, n_blocks, ctor)
    with torch.no_grad():
        out = m(g["rain"].to(gpu), g["dem"].to(gpu), g["lu"].to(gpu), **fkw).cpu()
    assert out.shape == g["out"].shape
    assert relmax(out, g["out"]) <= LOOSE, relmax(out, g["out"])
    okw = dict(ctor)
    okw.update(fkw)
    emu = O.generator_forward(master_params(n_blocks), g["rain"], g["dem"], g["lu"], emulate="bf16", **okw)
    assert float((out - emu).abs().max()) <= TIGHT * float(g["out"].abs().max())


def _synthetic(B, T, H, W, s, seed):
    g = torch.Generator().manual_seed(seed)
    rain = 10 * torch.rand(B, T, 1, H, W, generator=g) * (torch.rand(B, T, 1, H, W, generator=g) > 0.7)
    dem = torch.rand(B, 1, H * s, W * s, generator=g)
    lu = torch.nn.functional.one_hot(torch.randint(0, 5, (B, H * s, W * s), generator=g), 5)
    return rain, dem, lu.permute(0, 3, 1, 2).float().contiguous()


def test_full_resolution_vs_oracle(gpu):
    """256x256 frames (the benchmark resolution), B=1, T=2, checked element-wise against the oracle."""
    m = build_model(gpu, 0, dict(scale_factor=1))
    rain, dem, lu = _synthetic(1, 2, 256, 256, 1, 5)
    with torch.no_grad():
        out = m(rain.to(gpu), dem.to(gpu), lu.to(gpu)).cpu()
    ref = O.generator_forward(master_params(0), rain, dem, lu, scale_factor=1)
    emu = O.generator_forward(master_params(0), rain, dem, lu, scale_factor=1, emulate="bf16")
    assert relmax(out, ref) <= LOOSE
    assert float((out - emu).abs().max()) <= TIGHT * float(ref.abs().max())


def test_full_size_properties(gpu):
    """Benchmark-size rollout (B=4, T=10, 256x256): size-independent properties, bit-exact.
    (a) repeatability, (b) samples are independent (batch of 4 == four batches of 1),
    (c) causality: frames < k do not change when inputs >= k change."""
    m = build_model(gpu, 0, dict(scale_factor=1))
    rain, dem, lu = _synthetic(4, 10, 256, 256, 1, 6)
    rain, dem, lu = rain.to(gpu), dem.to(gpu), lu.to(gpu)
    with torch.no_grad():
        a = m(rain, dem, lu).clone()
        b = m(rain, dem, lu).clone()
        assert torch.equal(a, b)
        for i in (0, 3):
            single = m(rain[i:i + 1].contiguous(), dem[i:i + 1].contiguous(), lu[i:i + 1].contiguous())
            assert torch.equal(single[0], a[i])
        rain2 = rain.clone()
        rain2[:, 6:] += 1.0
        c = m(rain2, dem, lu)
        assert torch.equal(c[:, :6], a[:, :6])
        assert not torch.equal(c[:, 6:], a[:, 6:])
    assert torch.isfinite(a).all()


def test_default_yaml_native_shape(gpu):
    """default.yaml's own shape family: B=8, T=5, 15x12 LR grid, scale 8 -> 120x96."""
    m = build_model(gpu, 3, dict(scale_factor=8))
    rain, dem, lu = _synthetic(8, 5, 15, 12, 8, 7)
    with torch.no_grad():
        out = m(rain.to(gpu), dem.to(gpu), lu.to(gpu)).cpu()
    assert out.shape == (8, 5, 1, 120, 96)
    ref = O.generator_forward(master_params(3), rain, dem, lu, scale_factor=8)
    assert relmax(out, ref) <= LOOSE


def test_reference_test_model_shapes(gpu):
    """test/test_model.py:12-50, 53-97, 150-179 of the reference (shape checks), on the engine."""
    from pl_convlstm_gan_amd import Generator
    torch.manual_seed(0)
    m = Generator(hidden_dims=[16, 32], lu_channels=5, scale_factor=8).to(gpu)
    x = torch.randn(2, 5, 1, 32, 32, device=gpu)
    with torch.no_grad():
        assert m(x, torch.randn(2, 1, 32, 32, device=gpu), torch.randn(2, 5, 32, 32, device=gpu)).shape == (2, 5, 1, 256, 256)
    m = Generator(hidden_dims=[16, 32], lu_channels=5, target_grid_size=(500, 500)).to(gpu)
    with torch.no_grad():
        y = m(x, torch.randn(2, 1, 32, 32, device=gpu), torch.randn(2, 5, 32, 32, device=gpu),
              input_grid_size=(4000, 4000))
    assert y.shape == (2, 5, 1, 256, 256)
    m = Generator(hidden_dims=[16, 32], lu_channels=5, scale_factor=4).to(gpu)
    with torch.no_grad():
        y = m(torch.randn(1, 3, 1, 16, 16, device=gpu), torch.randn(1, 1, 16, 16, device=gpu),
              torch.randn(1, 5, 16, 16, device=gpu))
    assert y.shape == (1, 3, 1, 64, 64)


def test_reference_test_model_calls_as_written():
    """The calls of the reference's test/test_model.py:22-41, 62-86, 158-170 exactly as written there:
    model and inputs are built on the CPU and the model is called on them.  The operands are staged
    onto the GPU, the HIP rollout runs there, the result comes back as a CPU tensor (the library has
    no CPU arithmetic).  The staged result equals the GPU-resident call bit for bit."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from src.models.generator import Generator
    from src.losses.combined_loss import CombinedLoss
    batch_size, T, H, W = 2, 5, 32, 32
    rain_lr = torch.randn(batch_size, T, 1, H, W)
    dem = torch.randn(batch_size, 1, H, W)
    lu = torch.randn(batch_size, 5, H, W)
    model = Generator(hidden_dims=[16, 32], lu_channels=5, scale_factor=8)
    with torch.no_grad():
        output = model(rain_lr, dem, lu)
    assert output.device.type == "cpu" and output.shape == (batch_size, T, 1, H * 8, W * 8)
    assert next(model.parameters()).device.type == "cpu" and len(model.upsample_blocks) == 3
    twin = Generator(hidden_dims=[16, 32], lu_channels=5, scale_factor=8)
    twin._geometry(H, W, None, "cpu")
    twin.load_state_dict(model.state_dict())
    twin = twin.cuda()
    with torch.no_grad():
        assert torch.equal(twin(rain_lr.cuda(), dem.cuda(), lu.cuda()).cpu(), output)

    model = Generator(hidden_dims=[16, 32], lu_channels=5, target_grid_size=(500, 500))
    with torch.no_grad():
        output = model(rain_lr, dem, lu, input_grid_size=(4000, 4000))
    assert output.shape == (batch_size, T, 1, 256, 256)

    model = Generator(hidden_dims=[16, 32], lu_channels=5, scale_factor=4)
    with torch.no_grad():
        output = model(torch.randn(1, 3, 1, 16, 16), torch.randn(1, 1, 16, 16), torch.randn(1, 5, 16, 16))
    assert output.shape == (1, 3, 1, 64, 64)

    # test/test_model.py:100-147: the loss on CPU tensors (total may be NaN for randn gauge values, as in the reference)
    loss_module = CombinedLoss(lambda_point=0.1, lambda_conserve=1.0)
    pred = torch.randn(2, 5, 1, 256, 256)
    total, parts = loss_module(pred, rain_lr, torch.randint(0, 32, (2, 5, 2)), torch.randn(2, 5, 5).abs(), scale_factor=8)
    assert torch.isfinite(total) and set(parts) == {"point", "conserve", "smooth", "temporal"}

    # a CPU-resident module trains through the staged path too: parameter gradients arrive on the CPU
    model = Generator(hidden_dims=[16, 32], lu_channels=5, scale_factor=2)
    out = model(torch.rand(1, 2, 1, 8, 8), torch.rand(1, 1, 16, 16), torch.rand(1, 5, 16, 16))
    out.square().sum().backward()
    g_cpu = {k: p.grad.clone() for k, p in model.named_parameters()}
    assert all(g.device.type == "cpu" and torch.isfinite(g).all() for g in g_cpu.values())
    assert float(g_cpu["cell2.conv.weight"].abs().max()) > 0
    # inputs that ask for a gradient get one, on their own device (the staging copies are differentiable)
    rain = torch.rand(1, 2, 1, 8, 8, requires_grad=True)
    model(rain, torch.rand(1, 1, 16, 16), torch.rand(1, 5, 16, 16)).sum().backward()
    assert rain.grad is not None and rain.grad.device.type == "cpu" and float(rain.grad.abs().max()) > 0


def test_errors(gpu):
    from pl_convlstm_gan_amd import Generator
    wide = Generator(hidden_dims=[16, 520], lu_channels=5, scale_factor=1).to(gpu)     # widths above 512 are refused
    with pytest.raises(NotImplementedError, match="hidden_dims"):
        with torch.no_grad():
            wide(torch.zeros(1, 1, 1, 8, 8, device=gpu), torch.zeros(1, 1, 8, 8, device=gpu),
                 torch.zeros(1, 5, 8, 8, device=gpu))
    # reference quirk: neither scale_factor nor target_grid_size -> AttributeError in forward
    q = Generator(hidden_dims=[16, 32], lu_channels=5).to(gpu)
    with pytest.raises(AttributeError):
        with torch.no_grad():
            q(torch.zeros(1, 1, 1, 8, 8, device=gpu), torch.zeros(1, 1, 8, 8, device=gpu),
              torch.zeros(1, 5, 8, 8, device=gpu))
