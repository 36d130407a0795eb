"""The same forward + CombinedLoss + backward captured into a HIP graph and replayed on fixed inputs and fixed
parameters: every replay must give the eager pass's gradients bit for bit.
    SHAPE=B,T,H,W,s REPS=n python devtools/grad_repeat_graph.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pl_convlstm_gan_amd import Generator
from pl_convlstm_gan_amd.losses import CombinedLoss
gpu = torch.device("cuda:0")
B, T, H, W, s = (int(v) for v in os.environ.get("SHAPE", "48,5,15,12,2").split(","))
seed = 3
torch.manual_seed(seed)
m = Generator(hidden_dims=[16, 32], lu_channels=5, scale_factor=s).to(gpu)
g = torch.Generator().manual_seed(seed + 1)
rain = (10 * torch.rand(B, T, 1, H, W, generator=g) * (torch.rand(B, T, 1, H, W, generator=g) > 0.7)).to(gpu)
dem = torch.rand(B, 1, H * s, W * s, generator=g).to(gpu)
lu = torch.nn.functional.one_hot(torch.randint(0, 5, (B, H * s, W * s), generator=g), 5).permute(0, 3, 1, 2).float().contiguous().to(gpu)
coords = torch.randint(0, min(H, W), (10, 2), generator=g).to(gpu)
values = (20 * torch.rand(B, T, 10, generator=g)).to(gpu)
lossm = CombinedLoss()


def one():
    for p in m.parameters():
        p.grad = None
    out = m(rain, dem, lu)
    loss, _ = lossm(out, rain, coords, values, float(s))
    loss.backward()
    return out.detach(), loss.detach()


def snap(out, loss):
    cur = {k: p.grad.clone() for k, p in m.named_parameters()}
    cur["out"] = out.clone()
    cur["loss"] = loss.clone()
    return cur


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        o, l = one()
torch.cuda.current_stream().wait_stream(side)
ref = snap(o, l)
del o, l
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    o, l = one()
nbad = 0
for rep in range(int(os.environ.get('REPS', '60'))):
    graph.replay()
    if os.environ.get("BETWEEN"):          # host activity between replays: index reads and D2H copies
        idx = o[:, :, 0][:, :, coords[:, 0], coords[:, 1]]
        v = float(idx.mean())
        cp = [p.detach().cpu() for p in m.parameters()]
    cur = snap(o, l)
    bad = [(k, int((cur[k] != ref[k]).sum()), float((cur[k] - ref[k]).abs().max()), float(ref[k].abs().max())) for k in cur if not torch.equal(cur[k], ref[k])]
    if bad:
        nbad += 1
        print("replay", rep, "differs from eager:", bad, flush=True)
print("SHAPE", (B, T, H, W, s), "replays with a difference:", nbad)
