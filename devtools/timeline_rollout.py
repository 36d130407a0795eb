"""Phase timeline of selected conv instantiations inside a full forward + backward rollout
(instrumented build, devtools/libplc_tl.so).   python devtools/timeline_rollout.py 40111 160012 ..."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pl_convlstm_gan_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libplc_tl.so")
L = _lib.lib()
L.plc_debug_set_timeline.argtypes = [C.c_void_p]
L.plc_debug_select_timeline.argtypes = [C.c_int]
from pl_convlstm_gan_amd import Generator  # noqa: E402

dev = torch.device("cuda:0")
NS = 10
B, T, H, W = 4, 10, 256, 256
torch.manual_seed(0)
model = Generator(hidden_dims=[16, 32], lu_channels=5, scale_factor=1).to(dev)
rain = torch.rand(B, T, 1, H, W, device=dev)
dem = torch.rand(B, 1, H, W, device=dev)
lu = torch.rand(B, 5, H, W, device=dev)


def step():
    out = model(rain, dem, lu)
    out.sum().backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
NB = 4096
names = ["entry", "tile begin", "staged", "barrier", "pass0/loop begin", "loop", "epi", "drained"]
for sel in [int(s) for s in sys.argv[1:]]:
    tl = torch.zeros(NB * 4 * NS, dtype=torch.int64, device=dev)
    L.plc_debug_select_timeline(sel)
    L.plc_debug_set_timeline(tl.data_ptr())
    step()
    torch.cuda.synchronize()
    L.plc_debug_set_timeline(None)
    raw = tl.cpu().numpy().reshape(NB, 4, NS)
    used = raw[:, 0, 0] != 0
    t = raw[used][:, :, :8].astype(np.float64) / 100.0
    t -= t[:, :, 0].min()
    print(f"kernel id {sel}: {int(used.sum())} workgroups recorded (last launch, last tile of each)")
    for k, nm in enumerate(names):
        col = t[:, :, k]
        print(f"  {nm:18s} min {col.min():8.2f} mean {col.mean():8.2f} max {col.max():8.2f}")
    d = t[:, :, 2:] - t[:, :, 1:-1]
    print("  last-tile phases from 'tile begin' (us):", " ".join(f"{v:6.2f}" for v in d.mean(axis=(0, 1))))
