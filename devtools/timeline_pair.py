"""Timeline of one two-conv launch (instrumented build)."""
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
L.plc_debug_cell_pair.argtypes = [C.c_void_p] * 14 + [C.c_int] * 3 + [C.c_void_p]
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
NS = 10
B, H, W = 4, 256, 256
hw = H * W


def operands(Cin, Ch):
    x = torch.randn(B * (Cin // 8) * hw * 8, device=dev).to(torch.bfloat16)
    h = torch.randn(B * (Ch // 8) * hw * 8, device=dev).to(torch.bfloat16)
    c = torch.randn(B * Ch * hw, device=dev)
    w = torch.randn(4 * Ch, Cin + Ch, 3, 3, device=dev) * 0.05
    b = torch.zeros(4 * Ch, device=dev)
    wpk = torch.empty(L.plc_packed_weight_bytes(4 * Ch, Cin, Ch, 3), dtype=torch.uint8, device=dev)
    bpk = torch.empty(L.plc_packed_bias_bytes(4 * Ch), dtype=torch.uint8, device=dev)
    _lib.check(L.plc_pack_conv_weight(w.data_ptr(), b.data_ptr(), wpk.data_ptr(), bpk.data_ptr(), 4 * Ch, Cin, Ch, 3, 1, st), "pack")
    return [x, h, c, wpk, bpk, torch.empty_like(h), torch.empty_like(c)]


o2, o1 = operands(16, 32), operands(16, 16)
n_blk = 1024
tl = torch.zeros(n_blk * 4 * NS, dtype=torch.int64, device=dev)


def run():
    args = [t.data_ptr() for t in o2] + [t.data_ptr() for t in o1]
    _lib.check(L.plc_debug_cell_pair(*args, B, H, W, st), "pair")


for _ in range(5):
    run()
torch.cuda.synchronize()
L.plc_debug_set_timeline(tl.data_ptr())
run()
torch.cuda.synchronize()
L.plc_debug_set_timeline(None)
raw = tl.cpu().numpy().reshape(n_blk, 4, NS)
t = raw[:, :, :8].astype(np.float64) / 100.0
t -= t[:, :, 0].min()
names = ["entry", "weights", "staged", "barrier", "pass0", "loop", "epi", "drained"]
for q, label in enumerate(["cell2 first 256", "cell1 first 256", "cell2 second 256", "cell1 second 256"]):
    blk = t[q * 256:(q + 1) * 256]
    print(label)
    for k, nm in enumerate(names):
        col = blk[:, :, k]
        print(f"  {nm:10s} min {col.min():7.2f} mean {col.mean():7.2f} max {col.max():7.2f}")
print("span", t[:, :, 7].max())
hw_id = raw[:, :, 8]
xcc = raw[:, :, 9] & 0xF
cu_key = (((xcc * 8 + ((hw_id >> 13) & 7)) * 2 + ((hw_id >> 12) & 1)) * 16 + ((hw_id >> 8) & 0xF))[:, 0]
for key in sorted(set(cu_key.tolist()))[:6]:
    blks = [b for b in range(n_blk) if cu_key[b] == key]
    print("CU", key, [(b, b // 256, round(float(t[b, 0, 0]), 1), round(float(t[b, 0, 7]), 1)) for b in blks])
