#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE on CPU.

Run in the build container only (the reference never travels to the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [/root/reference]

Everything written here is data: seeded inputs, the reference's parameters and the
reference's outputs / gradients, as float32 .npz files.  torch 2.10.0 CPU, fp32.

Files
  master_params.npz   state_dict of Generator(hidden_dims=[16,32], lu_channels=5, scale_factor=8)
                      after one warm-up forward (3 lazily created upsample blocks included);
                      every other generator case loads a subset of it.
  cell_*.npz          ConvLSTMCell forward + gradients                    (convlstm.py:16-28)
  gen_*.npz           Generator forward (+ parameter gradients)           (generator.py:94-205)
  loss_*.npz          CombinedLoss components                             (combined_loss.py:23-191)
  step_s2.npz         one optimisation step in trainer.py:290-315 order, Adam built before the
                      first forward as trainer.py:153-158 does (lazy blocks are not updated)
  genw_*.npz          Generator of other widths (the class default hidden_dims=[32,64] of
                      generator.py:33, [8,24], and [10,20] / [12,30], which are no multiples of 8) with its
                      own parameters ("param.<key>"), forward and parameter gradients
"""
import os
import sys

import numpy as np
import torch

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

from src.models.convlstm import ConvLSTMCell          # noqa: E402
from src.models.generator import Generator            # noqa: E402
from src.losses.combined_loss import CombinedLoss     # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
torch.set_num_threads(4)


def npd(d):
    return {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in d.items()}


ONLY = [n for n in os.environ.get("PLC_GOLDEN_ONLY", "").split(",") if n]   # rewrite just these files


def save(name, **arrays):
    if ONLY and name not in ONLY:
        return
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **npd(arrays))
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def rain_like(gen, *shape):
    x = 10.0 * torch.rand(*shape, generator=gen)
    return x * (torch.rand(*shape, generator=gen) > 0.7)


# ------------------------------------------------------------------ master parameters
torch.manual_seed(0)
master = Generator(hidden_dims=[16, 32], lu_channels=5, scale_factor=8)
with torch.no_grad():
    master(torch.zeros(1, 1, 1, 4, 4), torch.zeros(1, 1, 4, 4), torch.zeros(1, 5, 4, 4))
MASTER = {k: v.detach().clone() for k, v in master.state_dict().items()}
save("master_params.npz", **MASTER)


def sub_params(n_blocks):
    keep = {}
    for k, v in MASTER.items():
        if k.startswith("upsample_blocks."):
            if int(k.split(".")[1]) >= n_blocks:
                continue
        keep[k] = v
    return keep


def model_with_master(n_blocks, fwd_kwargs=None, **ctor):
    m = Generator(hidden_dims=[16, 32], lu_channels=5, **ctor)
    with torch.no_grad():
        m(torch.zeros(1, 1, 1, 4, 4), torch.zeros(1, 1, 4, 4), torch.zeros(1, 5, 4, 4),
          **(fwd_kwargs or {}))
    assert len(m.upsample_blocks) == n_blocks, (len(m.upsample_blocks), n_blocks)
    m.load_state_dict(sub_params(n_blocks), strict=True)
    return m


# ------------------------------------------------------------------ G1: ConvLSTM cell
for tag, cin, ch, seed in (("16_16", 16, 16, 11), ("16_32", 16, 32, 12)):
    torch.manual_seed(seed)
    cell = ConvLSTMCell(cin, ch)
    B, H, W = 2, 12, 12
    x = torch.randn(B, cin, H, W, requires_grad=True)
    h = (0.5 * torch.randn(B, ch, H, W)).requires_grad_(True)
    c = torch.randn(B, ch, H, W, requires_grad=True)
    rh = torch.randn(B, ch, H, W)
    rc = torch.randn(B, ch, H, W)
    h2, c2 = cell(x, h, c)
    ((h2 * rh).sum() + (c2 * rc).sum()).backward()
    save(f"cell_{tag}.npz", x=x, h=h, c=c, weight=cell.conv.weight, bias=cell.conv.bias,
         rh=rh, rc=rc, h_next=h2, c_next=c2, dx=x.grad, dh=h.grad, dc=c.grad,
         dweight=cell.conv.weight.grad, dbias=cell.conv.bias.grad)


# ------------------------------------------------------------------ generator forward cases
def gen_case(name, n_blocks, B, T, H, W, seed, ctor, fwd_kwargs=None, aux_hw=None,
             with_grads=None, loss_kwargs=None):
    g = torch.Generator().manual_seed(seed)
    m = model_with_master(n_blocks, fwd_kwargs, **ctor)
    rain = rain_like(g, B, T, 1, H, W)
    with torch.no_grad():
        probe = m(rain[:1, :1], torch.zeros(1, 1, H, W), torch.zeros(1, 5, H, W), **(fwd_kwargs or {}))
    Hh, Wh = probe.shape[-2:]
    ah, aw = aux_hw if aux_hw is not None else (Hh, Wh)
    dem = torch.rand(B, 1, ah, aw, generator=g)
    lu = torch.nn.functional.one_hot(torch.randint(0, 5, (B, ah, aw), generator=g), 5)
    lu = lu.permute(0, 3, 1, 2).float().contiguous()
    out = m(rain, dem, lu, **(fwd_kwargs or {}))
    arrays = dict(rain=rain, dem=dem, lu=lu, out=out, n_blocks=np.int64(n_blocks))
    if with_grads is not None:
        N = 7
        s_coords = torch.randint(0, min(H, W), (B, N, 2), generator=g)
        s_coords[1:] = s_coords[0]
        s_values = 20.0 * torch.rand(B, T, N, generator=g)
        s_values[0, 0, 1] = float("nan")
        loss_mod = CombinedLoss(**(loss_kwargs or {}))
        sf = out.shape[-2] / H
        total, parts = loss_mod(out, rain, s_coords, s_values, sf)
        m.zero_grad()
        total.backward()
        arrays.update(s_coords=s_coords, s_values=s_values, loss=total.detach(),
                      loss_point=parts["point"].detach(), loss_conserve=parts["conserve"].detach(),
                      loss_smooth=parts["smooth"].detach(), loss_temporal=parts["temporal"].detach())
        for k, prm in m.named_parameters():
            if with_grads == "all" or any(k.startswith(pfx) for pfx in with_grads):
                arrays["grad." + k] = prm.grad
    save(name, **arrays)


gen_case("gen_s1.npz", 0, 2, 3, 16, 16, 21, dict(scale_factor=1), with_grads="all")
gen_case("gen_s1_odd.npz", 0, 1, 2, 15, 12, 22, dict(scale_factor=1))          # default.yaml-like ragged LR grid
gen_case("gen_s2.npz", 1, 1, 2, 8, 8, 23, dict(scale_factor=2),
         with_grads=("upsample_blocks.0.conv", "init_conv", "cell2.conv.bias", "cell1.conv.bias",
                     "post_process.2", "dem_attn.conv.2", "lu_attn.conv.0"))
gen_case("gen_s8.npz", 3, 1, 2, 8, 8, 24, dict(scale_factor=8))
gen_case("gen_grid.npz", 3, 1, 2, 8, 8, 25, dict(target_grid_size=(500, 500)),
         fwd_kwargs=dict(input_grid_size=(4000, 4000)))
gen_case("gen_s3.npz", 1, 1, 2, 8, 8, 26, dict(scale_factor=3))                 # 1 block + bilinear x1.5
gen_case("gen_s6.npz", 2, 1, 2, 8, 8, 27, dict(scale_factor=6))                 # 2 blocks + bilinear x1.5
gen_case("gen_s2_aux.npz", 1, 2, 2, 8, 8, 28, dict(scale_factor=2), aux_hw=(24, 20))  # aux maps resized
gen_case("gen_s4_native.npz", 2, 1, 3, 15, 12, 29, dict(scale_factor=4))        # ragged LR, 2 blocks
# gradients through the bilinear feature resizes (generator.py:179-195)
RESIZE_GRADS = ("upsample_blocks.0.conv", "init_conv", "cell2.conv.bias", "cell1.conv", "post_process.0",
                "dem_attn.conv.0", "lu_attn.conv.2")
gen_case("gen_s3_grads.npz", 1, 2, 2, 10, 7, 41, dict(scale_factor=3), with_grads=RESIZE_GRADS)
gen_case("gen_grid_grads.npz", 3, 1, 2, 8, 8, 42, dict(target_grid_size=(500, 600)),        # 64x64 -> 53x64
         fwd_kwargs=dict(input_grid_size=(4000, 4000)), with_grads=RESIZE_GRADS)
gen_case("gen_grid_mix_grads.npz", 1, 1, 2, 9, 12, 43, dict(target_grid_size=(1000, 900)),   # x2, x1.5, then 27 -> 30 rows
         fwd_kwargs=dict(input_grid_size=(3000, 3000)), with_grads=RESIZE_GRADS)


# ------------------------------------------------------------------ G8: CombinedLoss
def loss_case(name, seed, B, T, Hl, Wl, s, N, coords_batched=True, strategy="log", weighted=True,
              negative=False, nan_frac=0.1):
    g = torch.Generator().manual_seed(seed)
    pred = torch.randn(B, T, 1, Hl * s, Wl * s, generator=g)
    lr = rain_like(g, B, T, 1, Hl, Wl)
    coords = torch.randint(0, min(Hl, Wl), (N, 2), generator=g)
    s_coords = coords.unsqueeze(0).repeat(B, 1, 1) if coords_batched else coords
    s_values = 60.0 * torch.rand(B, T, N, generator=g)
    if negative:
        s_values = s_values - 30.0
    nanmask = torch.rand(B, T, N, generator=g) < nan_frac
    s_values[nanmask] = float("nan")
    mod = CombinedLoss(lambda_point=0.7, lambda_conserve=1.3, lambda_smooth=0.2, lambda_temporal=0.05,
                       use_weighted_loss=weighted, weight_strategy=strategy)
    total, parts = mod(pred, lr, s_coords, s_values, float(s))
    save(name, pred=pred, lr=lr, s_coords=s_coords, s_values=s_values, scale=np.float64(s),
         strategy=np.array(strategy), weighted=np.array(weighted), total=total,
         point=parts["point"], conserve=parts["conserve"], smooth=parts["smooth"],
         temporal=parts["temporal"], lambdas=np.array([0.7, 1.3, 0.2, 0.05]))


loss_case("loss_log.npz", 31, 2, 3, 6, 5, 4, 9)
loss_case("loss_strat.npz", 32, 2, 3, 6, 5, 2, 9, coords_batched=False, strategy="stratified")
loss_case("loss_sqrt.npz", 33, 1, 4, 5, 6, 1, 6, strategy="sqrt")
loss_case("loss_plain.npz", 34, 2, 2, 6, 6, 3, 5, weighted=False)
loss_case("loss_negative_nan.npz", 35, 2, 3, 6, 5, 4, 9, negative=True)      # log1p(<-1) -> NaN total


# ------------------------------------------------------------------ G9: one optimisation step
def step_case():
    g = torch.Generator().manual_seed(41)
    m = Generator(hidden_dims=[16, 32], lu_channels=5, scale_factor=2)
    # parameters that exist before the first forward (trainer.py:153-158 builds Adam here)
    base = {k: v for k, v in sub_params(0).items()}
    m.load_state_dict(base, strict=True)
    opt = torch.optim.Adam(m.parameters(), lr=5e-4)
    # the lazy block appears inside the first forward; pin its values to the master's block 0
    B, T, H, W = 2, 2, 8, 8
    rain = rain_like(g, B, T, 1, H, W)
    dem = torch.rand(B, 1, 16, 16, generator=g)
    lu = torch.nn.functional.one_hot(torch.randint(0, 5, (B, 16, 16), generator=g), 5)
    lu = lu.permute(0, 3, 1, 2).float().contiguous()
    N = 6
    s_coords = torch.randint(0, 8, (N, 2), generator=g).unsqueeze(0).repeat(B, 1, 1)
    s_values = 20.0 * torch.rand(B, T, N, generator=g)
    with torch.no_grad():
        m(rain, dem, lu)                                  # creates upsample_blocks (random)
    m.load_state_dict(sub_params(1), strict=True)         # ... then pin them
    loss_mod = CombinedLoss(lambda_point=1.0, lambda_conserve=1.0, lambda_smooth=0.1, lambda_temporal=0.05)
    opt.zero_grad()
    out = m(rain, dem, lu)
    total, _ = loss_mod(out, rain, s_coords, s_values, out.shape[-2] / H)
    total.backward()
    norm = torch.nn.utils.clip_grad_norm_(m.parameters(), max_norm=0.5)
    opt.step()
    after = {k: v.detach().clone() for k, v in m.state_dict().items()}
    keep = ("init_conv.weight", "init_conv.bias", "cell1.conv.bias", "cell2.conv.bias",
            "post_process.2.weight", "post_process.2.bias", "dem_attn.conv.2.weight",
            "lu_attn.conv.0.weight", "upsample_blocks.0.conv.bias")
    arrays = dict(rain=rain, dem=dem, lu=lu, s_coords=s_coords, s_values=s_values, loss=total.detach(),
                  grad_norm=norm.detach(),
                  upsample_unchanged=np.array(bool(torch.equal(after["upsample_blocks.0.conv.weight"],
                                                               MASTER["upsample_blocks.0.conv.weight"]))))
    for k in keep:
        arrays["after." + k] = after[k]
    save("step_s2.npz", **arrays)


step_case()


# ------------------------------------------------------------------ other layer widths (own parameters)
def wide_case(name, hidden, B, T, H, W, seed, ctor, with_grads):
    torch.manual_seed(seed)
    g = torch.Generator().manual_seed(seed + 1000)
    m = Generator(lu_channels=5, **ctor) if hidden is None else Generator(hidden_dims=hidden, lu_channels=5, **ctor)
    rain = rain_like(g, B, T, 1, H, W)
    with torch.no_grad():
        probe = m(rain[:1, :1], torch.zeros(1, 1, H, W), torch.zeros(1, 5, H, W))     # creates the lazy blocks
    Hh, Wh = probe.shape[-2:]
    dem = torch.rand(B, 1, Hh, Wh, generator=g)
    lu = torch.nn.functional.one_hot(torch.randint(0, 5, (B, Hh, Wh), generator=g), 5).permute(0, 3, 1, 2).float().contiguous()
    out = m(rain, dem, lu)
    arrays = dict(rain=rain, dem=dem, lu=lu, out=out, n_blocks=np.int64(len(m.upsample_blocks)),
                  hidden=np.asarray(m.hidden_dims, dtype=np.int64))
    for k, v in m.state_dict().items():
        arrays["param." + k] = v
    N = 7
    s_coords = torch.randint(0, min(H, W), (N, 2), generator=g).unsqueeze(0).repeat(B, 1, 1)
    s_values = 20.0 * torch.rand(B, T, N, generator=g)
    total, _ = CombinedLoss()(out, rain, s_coords, s_values, out.shape[-2] / H)
    total.backward()
    arrays.update(s_coords=s_coords, s_values=s_values, loss=total.detach())
    for k, prm in m.named_parameters():
        if with_grads == "all" or any(k.startswith(pfx) for pfx in with_grads):
            arrays["grad." + k] = prm.grad
    save(name, **arrays)


# the reference's class default widths (generator.py:33: hidden_dims=[32, 64]), one upsample block;
# gradients of the small tensors only (the file would otherwise double)
wide_case("genw_default_s2.npz", None, 1, 2, 8, 8, 51, dict(scale_factor=2),
          with_grads=("init_conv", "cell1.conv.bias", "cell2.conv.bias", "post_process.2", "post_process.0.bias",
                      "dem_attn.conv.0", "lu_attn.conv.0", "lu_attn.conv.2.bias", "upsample_blocks.0.conv.bias"))
# widths that are no multiple of 16: three hidden channel blocks, odd chunk sizes
wide_case("genw_8_24_s1.npz", [8, 24], 2, 3, 12, 9, 52, dict(scale_factor=1), with_grads="all")
# widths that are no multiple of 8 (the reference takes any, generator.py:32-58): the engine carries them as the
# next multiple with zero channels.  [10, 20] through an upsample block (run-time-width kernels), [12, 30] at
# scale 1 (carried as [16, 32]: the instantiated kernels)
wide_case("genw_10_20_s2.npz", [10, 20], 2, 3, 9, 12, 53, dict(scale_factor=2), with_grads="all")
wide_case("genw_12_30_s1.npz", [12, 30], 2, 3, 12, 9, 54, dict(scale_factor=1), with_grads="all")
print("done")
