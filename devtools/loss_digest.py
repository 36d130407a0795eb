"""sha256 of the fused CombinedLoss outputs (five scalars) and of d loss / d pred over a set of geometries:
integer scales (1, 2, 4, 8; widths that are and are not multiples of 4), non-integer adaptive bins, with and
without a gradient.  Two builds of the library give the same lines iff their loss kernels agree bit for bit:
    python3 devtools/loss_digest.py > a.txt;  PLC_LIB_PATH=other.so python3 devtools/loss_digest.py > b.txt"""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pl_convlstm_gan_amd.losses import CombinedLoss  # noqa: E402

gpu = torch.device("cuda:0")
CASES = [  # B, T, H, W, out_H, out_W, N
    (2, 3, 16, 24, 16, 24, 7), (2, 3, 15, 12, 120, 96, 35), (4, 10, 32, 32, 256, 256, 35), (1, 2, 9, 10, 18, 20, 5),
    (2, 2, 7, 9, 28, 36, 4), (1, 3, 10, 15, 25, 37, 6), (2, 2, 5, 6, 10, 18, 3), (1, 2, 6, 7, 12, 14, 9), (3, 1, 8, 8, 64, 64, 2),
]
for B, T, H, W, oh, ow, N in CASES:
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + ow)
    pred = (10 * torch.rand(B, T, 1, oh, ow, generator=g)).to(gpu).requires_grad_(True)
    lr = (10 * torch.rand(B, T, 1, H, W, generator=g)).to(gpu)
    coords = torch.stack([torch.randint(0, H, (N,), generator=g), torch.randint(0, W, (N,), generator=g)], 1)
    coords = coords.unsqueeze(0).repeat(B, 1, 1).to(gpu)
    values = (20 * torch.rand(B, T, N, generator=g)).to(gpu)
    loss, parts = CombinedLoss()(pred, lr, coords, values, oh / H)
    loss.backward()
    h = hashlib.sha256(loss.detach().cpu().numpy().tobytes())
    for k in sorted(parts):
        v = parts[k]
        h.update(torch.as_tensor(float(v)).numpy().tobytes())
    h.update(pred.grad.cpu().numpy().tobytes())
    print(B, T, H, W, oh, ow, h.hexdigest()[:24], float(loss))
