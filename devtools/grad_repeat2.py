"""Repeated training passes on the same input at a given shape: every pass must reproduce the first one's output and
parameter gradients bit for bit.    SHAPE=B,T,H,W,s REPS=n python devtools/grad_repeat2.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pl_convlstm_gan_amd import Generator
gpu = torch.device("cuda:0")
B, T, H, W, s = (int(v) for v in os.environ.get("SHAPE", "48,5,15,12,2").split(","))
seed = 3
torch.manual_seed(seed)
m = Generator(hidden_dims=[16, 32], lu_channels=5, scale_factor=s).to(gpu)
g = torch.Generator().manual_seed(seed + 1)
rain = (10 * torch.rand(B, T, 1, H, W, generator=g) * (torch.rand(B, T, 1, H, W, generator=g) > 0.7)).to(gpu)
dem = torch.rand(B, 1, H * s, W * s, generator=g).to(gpu)
lu = torch.nn.functional.one_hot(torch.randint(0, 5, (B, H * s, W * s), generator=g), 5).permute(0, 3, 1, 2).float().contiguous().to(gpu)
wgt = (torch.rand(B, T, 1, H * s, W * s, generator=g) - 0.5).to(gpu)
ref, nbad = None, 0
junk = []
for rep in range(int(os.environ.get('REPS', '40'))):
    m.zero_grad()
    if os.environ.get("JUNK"):          # vary what lies around the workspaces
        junk.append(torch.full((1 + 7919 * rep % 100000,), float("nan"), device=gpu))
        if len(junk) > 5:
            junk.pop(0)
    out = m(rain, dem, lu)
    (out * wgt).sum().backward()
    cur = {k: p.grad.clone() for k, p in m.named_parameters()}
    cur["out"] = out.detach().clone()
    if ref is None:
        ref = cur
    else:
        bad = [(k, int((cur[k] != ref[k]).sum()), float((cur[k] - ref[k]).abs().max()), float(ref[k].abs().max())) for k in cur if not torch.equal(cur[k], ref[k])]
        if bad:
            nbad += 1
            print("rep", rep, "differs:", bad, flush=True)
print("SHAPE", (B, T, H, W, s), "reps with a difference:", nbad)
