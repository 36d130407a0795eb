"""Where the lu / dem input gradients deviate most from autograd of the bf16-emulated restatement (grid-size mode)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import master_params  # noqa: E402
from oracle import cpu_oracle as O  # noqa: E402
from pl_convlstm_gan_amd import Generator  # noqa: E402
from pl_convlstm_gan_amd.models import UpsampleBlock  # noqa: E402

gpu = torch.device("cuda:0")
n_blocks, ctor, shape, dem_hw, lu_hw, fwd_kw = 1, dict(target_grid_size=(1000, 900)), (1, 2, 36, 30), (90, 93), (120, 99), dict(input_grid_size=(3000, 3000))
B, T, H, W = shape
gen = torch.Generator().manual_seed(11 + n_blocks + H)
rain = torch.rand(B, T, 1, H, W, generator=gen)
dem = torch.rand(B, 1, *dem_hw, generator=gen)
lu = torch.rand(B, 5, *lu_hw, generator=gen)
params = master_params(n_blocks)
m = Generator(hidden_dims=[16, 32], lu_channels=5, **ctor)
m.upsample_blocks = torch.nn.ModuleList([UpsampleBlock(32, 2) for _ in range(n_blocks)])
m.load_state_dict(params, strict=True)
m = m.to(gpu)
ins = [t.to(gpu).requires_grad_(True) for t in (rain, dem, lu)]
out = m(*ins, **fwd_kw)
print("out", out.shape)
R = torch.randn(out.shape, generator=gen)
(out * R.to(gpu)).sum().backward()
got = [t.grad.cpu() for t in ins]
leaf = [t.clone().requires_grad_(True) for t in (rain, dem, lu)]
pred = O.generator_forward({k: v.clone() for k, v in params.items()}, *leaf, emulate="bf16", **ctor, **fwd_kw)
(pred * R).sum().backward()
for name, g, e in zip(("rain", "dem", "lu"), got, [t.grad for t in leaf]):
    d = (g - e).abs()
    idx = torch.nonzero(d == d.max())[0].tolist()
    print(name, "max|e|", float(e.abs().max()), "max dev", float(d.max()), "at", idx, "got", float(g[tuple(idx)]), "want", float(e[tuple(idx)]))
    flat = d.flatten()
    top = torch.topk(flat, 8)
    print("   top deviations:", [round(float(v), 6) for v in top.values], "median |e|", float(e.abs().median()))
    nz_g, nz_e = (g != 0), (e != 0)
    print("   non-zero pattern differs at", int((nz_g != nz_e).sum()), "of", g.numel())
