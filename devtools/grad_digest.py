"""Digest of the engine's output and parameter gradients at a few shapes: run it under two settings of a
kernel-selection switch (e.g. PLC_NO_ROLLING=1) and compare the printed lines; kernels that promise
bit-identical results must print identical digests.   python devtools/grad_digest.py"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pl_convlstm_gan_amd import Generator

gpu = torch.device("cuda:0")
for (B, T, H, W, s, seed) in [(4, 10, 256, 256, 1, 3), (2, 10, 128, 128, 1, 21), (2, 3, 16, 16, 1, 5), (1, 4, 24, 40, 1, 7),
                              (1, 3, 37, 45, 1, 8), (3, 2, 9, 70, 1, 11), (1, 2, 32, 32, 8, 9), (2, 2, 15, 12, 8, 13), (2, 3, 32, 32, 8, 15), (8, 2, 128, 128, 1, 17)]:
    torch.manual_seed(seed)
    m = Generator(hidden_dims=[16, 32], lu_channels=5, scale_factor=s).to(gpu)
    g = torch.Generator().manual_seed(seed + 1)
    rain = 10 * torch.rand(B, T, 1, H, W, generator=g) * (torch.rand(B, T, 1, H, W, generator=g) > 0.7)
    dem = torch.rand(B, 1, H * s, W * s, generator=g)
    lu = torch.rand(B, 5, H * s, W * s, generator=g)
    out = m(rain.to(gpu), dem.to(gpu), lu.to(gpu))
    wgt = torch.rand(out.shape, generator=g) + 0.5
    (out * wgt.to(gpu)).sum().backward()
    h = hashlib.sha256()
    h.update(out.detach().cpu().numpy().tobytes())
    for k, p in sorted(m.named_parameters()):
        h.update(p.grad.cpu().numpy().tobytes())
    print(f"B={B} T={T} {H}x{W} s={s}: {h.hexdigest()[:24]}", flush=True)
