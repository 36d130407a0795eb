"""Four training passes on the same c3-size input: every pass must reproduce the first one's output and parameter
gradients bit for bit (a race between concurrent launches shows up here as a tensor that differs).
    python devtools/grad_repeat.py"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pl_convlstm_gan_amd import Generator
gpu = torch.device("cuda:0")
B, T, H, W, s, seed = 4, 10, 256, 256, 1, 3
torch.manual_seed(seed)
m = Generator(hidden_dims=[16, 32], lu_channels=5, scale_factor=s).to(gpu)
g = torch.Generator().manual_seed(seed + 1)
rain = 10 * torch.rand(B, T, 1, H, W, generator=g) * (torch.rand(B, T, 1, H, W, generator=g) > 0.7)
dem = torch.rand(B, 1, H * s, W * s, generator=g)
lu = torch.rand(B, 5, H * s, W * s, generator=g)
wgt = (torch.rand(B, T, 1, H, W, generator=g) + 0.5).to(gpu)
ref = None
for rep in range(int(os.environ.get('REPS', '4'))):
    m.zero_grad()
    out = m(rain.to(gpu), dem.to(gpu), lu.to(gpu))
    (out * wgt).sum().backward()
    cur = {k: p.grad.clone() for k, p in m.named_parameters()}
    cur["out"] = out.detach().clone()
    if ref is None:
        ref = cur
    else:
        bad = [(k, float((cur[k] - ref[k]).abs().max()), float(ref[k].abs().max())) for k in cur if not torch.equal(cur[k], ref[k])]
        print("rep", rep, "differs:", bad)
