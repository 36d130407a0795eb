"""Layer widths other than [16,32] on the GPU: the reference's class default hidden_dims=[32,64]
(generator.py:33) and widths that are no multiple of 16.  These run on the run-time-width conv kernel
and the chunked weight-gradient launches; same rounding points as the instantiated kernels, so the
bounds are those of test_gpu_parity_fwd.py / test_gpu_parity_bwd.py: forward <= 2e-2 * max|ref| vs the
reference's fp32 goldens and <= 4e-3 vs the oracle's bf16-storage emulation; gradients <= 3e-2 (max
and L2) vs the emulation, <= 5e-2 L2 vs the reference's fp32 autograd goldens."""
import pytest
import torch

from conftest import load_golden
from oracle import cpu_oracle as O

pytestmark = pytest.mark.gpu


def relmax(a, b):
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-20)


def rel_l2(a, b):
    return float((a - b).norm()) / max(float(b.norm()), 1e-20)


def build(gpu, hidden, s, params):
    from pl_convlstm_gan_amd import Generator
    m = Generator(hidden_dims=hidden, lu_channels=5, scale_factor=s)
    m._geometry(8, 8, None, "cpu")              # materialise the lazy blocks, then load the golden parameters
    m.load_state_dict(params, strict=True)
    return m.to(gpu)


@pytest.mark.parametrize("name,hidden,s", [("genw_default_s2.npz", [32, 64], 2), ("genw_8_24_s1.npz", [8, 24], 1),
                                           # widths that are no multiples of 8 (the engine carries them as the next
                                           # multiple with zero channels): run-time-width kernels / instantiated kernels
                                           ("genw_10_20_s2.npz", [10, 20], 2), ("genw_12_30_s1.npz", [12, 30], 1)])
def test_other_widths_vs_reference_golden(gpu, name, hidden, s):
    from pl_convlstm_gan_amd.losses import CombinedLoss
    g = load_golden(name)
    params = {k[6:]: v for k, v in g.items() if k.startswith("param.")}
    m = build(gpu, hidden, s, params)
    out = m(g["rain"].to(gpu), g["dem"].to(gpu), g["lu"].to(gpu))
    assert out.shape == g["out"].shape
    assert relmax(out.detach().cpu(), g["out"]) <= 2e-2
    emu = O.generator_forward(params, g["rain"], g["dem"], g["lu"], scale_factor=s, emulate="bf16")
    assert float((out.detach().cpu() - emu).abs().max()) <= 4e-3 * float(g["out"].abs().max())
    loss, _ = CombinedLoss()(out, g["rain"].to(gpu), g["s_coords"].to(gpu), g["s_values"].to(gpu), float(s))
    assert abs(float(loss.detach()) - float(g["loss"])) <= 2e-3 * abs(float(g["loss"]))
    loss.backward()
    l_emu, g_emu, _ = O.grads_of(params, g["rain"], g["dem"], g["lu"], g["s_coords"], g["s_values"],
                                 fwd_kw=dict(scale_factor=s, emulate="bf16"))
    checked = 0
    for k, p in m.named_parameters():
        assert rel_l2(p.grad.cpu(), g_emu[k]) <= 3e-2, (k, rel_l2(p.grad.cpu(), g_emu[k]))
        if "grad." + k in g:
            ref = g["grad." + k]
            assert rel_l2(p.grad.cpu(), ref) <= 5e-2, (k, rel_l2(p.grad.cpu(), ref))
            checked += 1
    assert checked >= 9


WIDTH_CASES = [
    # hidden, B, T, H, W, scale, L
    ([32, 64], 2, 3, 40, 70, 1, 5),      # the class default widths over several tiles, ragged
    ([32, 64], 1, 2, 9, 12, 4, 3),       # ... through two upsample blocks
    ([8, 8], 2, 2, 17, 33, 1, 5),        # the narrowest: one channel block everywhere
    ([24, 40], 1, 3, 20, 36, 2, 9),      # odd block counts (3 and 5), two land-use blocks
    ([64, 128], 1, 2, 12, 40, 1, 5),     # more input blocks than one LDS chunk holds, 16 output tiles
    ([16, 64], 1, 2, 16, 16, 1, 5),      # instantiated cell 1 beside a run-time-width cell 2
    ([10, 20], 2, 3, 21, 40, 1, 5),      # no multiples of 8: carried as [16, 24] with zero channels
    ([13, 27], 1, 3, 20, 36, 4, 3),      # ... carried as [16, 32] (the instantiated kernels), two upsample blocks
    ([3, 5], 2, 2, 17, 33, 2, 2),        # narrower than one block (5 // 2 = 2 channels inside the attention branches)
    ([16, 35], 1, 2, 12, 40, 1, 5),      # an odd second width: 17 attention mid channels (three blocks)
]


@pytest.mark.parametrize("case", WIDTH_CASES)
def test_other_widths_forward_and_grads_vs_oracle(gpu, case):
    from pl_convlstm_gan_amd import Generator
    hidden, B, T, H, W, s, L = case
    torch.manual_seed(200 + hidden[0] + hidden[1])
    m = Generator(hidden_dims=hidden, lu_channels=L, scale_factor=s).to(gpu)
    gen = torch.Generator().manual_seed(9 + B + T)
    rain = 10 * torch.rand(B, T, 1, H, W, generator=gen) * (torch.rand(B, T, 1, H, W, generator=gen) > 0.5)
    dem = torch.rand(B, 1, H * s, W * s, generator=gen)
    lu = torch.rand(B, L, H * s, W * s, generator=gen)
    out = m(rain.to(gpu), dem.to(gpu), lu.to(gpu))
    params = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    ref = O.generator_forward(params, rain, dem, lu, scale_factor=s)
    leaf = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    emu = O.generator_forward(leaf, rain, dem, lu, scale_factor=s, emulate="bf16")
    assert relmax(out.detach().cpu(), ref) <= 2e-2
    assert float((out.detach().cpu() - emu.detach()).abs().max()) <= 4e-3 * float(ref.abs().max())
    wgt = torch.rand(out.shape, generator=gen) + 0.5
    (out * wgt.to(gpu)).sum().backward()
    (emu * wgt).sum().backward()
    for k, p in m.named_parameters():
        emax, el2 = relmax(p.grad.cpu(), leaf[k].grad), rel_l2(p.grad.cpu(), leaf[k].grad)
        assert emax <= 3e-2 and el2 <= 3e-2, (k, emax, el2)


@pytest.mark.parametrize("cin,ch", [(32, 32), (32, 64), (8, 24), (10, 20), (12, 30), (5, 3)])
def test_cell_other_widths(gpu, cin, ch):
    """ConvLSTMCell.forward and its gradients at other widths against the oracle."""
    from pl_convlstm_gan_amd import ConvLSTMCell
    torch.manual_seed(300 + cin + ch)
    cell = ConvLSTMCell(cin, ch).to(gpu)
    g = torch.Generator().manual_seed(5)
    B, H, W = 2, 13, 37
    x = torch.randn(B, cin, H, W, generator=g)
    h = 0.5 * torch.randn(B, ch, H, W, generator=g)
    c = torch.randn(B, ch, H, W, generator=g)
    rh, rc = torch.randn(B, ch, H, W, generator=g), torch.randn(B, ch, H, W, generator=g)
    xg, hg, cg = (t.to(gpu).requires_grad_(True) for t in (x, h, c))
    h2, c2 = cell(xg, hg, cg)
    ((h2 * rh.to(gpu)).sum() + (c2 * rc.to(gpu)).sum()).backward()
    w, b = cell.conv.weight.detach().cpu(), cell.conv.bias.detach().cpu()
    xo, ho, co, wo, bo = (t.clone().requires_grad_(True) for t in (x, h, c, w, b))
    h_ref, c_ref = O.convlstm_cell(xo, ho, co, wo, bo)
    ((h_ref * rh).sum() + (c_ref * rc).sum()).backward()
    assert relmax(h2.detach().cpu(), h_ref.detach()) <= 2e-2 and relmax(c2.detach().cpu(), c_ref.detach()) <= 2e-2
    for mine, ref in ((xg.grad, xo.grad), (hg.grad, ho.grad), (cg.grad, co.grad),
                      (cell.conv.weight.grad, wo.grad), (cell.conv.bias.grad, bo.grad)):
        assert rel_l2(mine.cpu(), ref) <= 3e-2
