"""Worst parameter-gradient error of the engine against autograd of the oracle's bf16 emulation at a few
shapes (the margin the parity tests have):  python devtools/grad_margin.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import cpu_oracle as O
from pl_convlstm_gan_amd import Generator

gpu = torch.device("cuda:0")
for (B, T, H, W, s, seed) in [(2, 10, 128, 128, 1, 21), (1, 20, 256, 256, 1, 31), (2, 3, 16, 16, 1, 5), (1, 20, 24, 40, 1, 7), (1, 2, 32, 32, 8, 9)]:
    torch.manual_seed(seed)
    m = Generator(hidden_dims=[16, 32], lu_channels=5, scale_factor=s).to(gpu)
    g = torch.Generator().manual_seed(seed + 1)
    rain = 10 * torch.rand(B, T, 1, H, W, generator=g) * (torch.rand(B, T, 1, H, W, generator=g) > 0.7)
    dem = torch.rand(B, 1, H * s, W * s, generator=g)
    lu = torch.rand(B, 5, H * s, W * s, generator=g)
    out = m(rain.to(gpu), dem.to(gpu), lu.to(gpu))
    params = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    leaf = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    emu = O.generator_forward(leaf, rain, dem, lu, scale_factor=s, emulate="bf16")
    wgt = torch.rand(out.shape, generator=g) + 0.5
    (out * wgt.to(gpu)).sum().backward()
    (emu * wgt).sum().backward()
    worst_l2 = max((float((p.grad.cpu() - leaf[k].grad).norm() / leaf[k].grad.norm()), k) for k, p in m.named_parameters())
    worst_mx = max((float((p.grad.cpu() - leaf[k].grad).abs().max() / leaf[k].grad.abs().max()), k) for k, p in m.named_parameters())
    rec = {k: float((p.grad.cpu() - leaf[k].grad).norm() / leaf[k].grad.norm()) for k, p in m.named_parameters()
           if k.startswith(("cell", "init_conv"))}
    print("   recurrent-path tensors rel-L2:", {k: f"{v:.2e}" for k, v in rec.items()}, flush=True)
    print(f"B={B} T={T} {H}x{W} s={s}: worst rel-L2 {worst_l2[0]:.3e} ({worst_l2[1]}), worst rel-max {worst_mx[0]:.3e} ({worst_mx[1]})", flush=True)
