"""Where a wave of the rolling-window conv spends its time (instrumented build, -DPLC_TIMELINE):
    python devtools/timeline_roll.py      # on the GPU box; expects devtools/libplc_tl.so"""
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
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream


def cell_bwd(B, Cin, Ch, H, W, sel=777, waves=4, names=None):
    g = torch.Generator(device="cpu").manual_seed(1)
    mk = lambda *s: torch.randn(*s, generator=g).to(dev)
    x, h, c = mk(B, Cin, H, W), mk(B, Ch, H, W), mk(B, Ch, H, W)
    w, b = mk(4 * Ch, Cin + Ch, 3, 3) * 0.05, mk(4 * Ch) * 0.1
    dhn, dcn = mk(B, Ch, H, W), mk(B, Ch, H, W)
    dx, dh, dc = torch.empty_like(x), torch.empty_like(h), torch.empty_like(c)
    dw, db = torch.empty_like(w), torch.empty_like(b)
    ws = torch.empty(L.plc_convlstm_cell_bwd_workspace_bytes(B, Cin, Ch, H, W), dtype=torch.uint8, device=dev)
    tl = torch.zeros(1024 * waves * 10, dtype=torch.int64, device=dev)

    def run():
        _lib.check(L.plc_convlstm_cell_bwd(x.data_ptr(), h.data_ptr(), c.data_ptr(), w.data_ptr(), b.data_ptr(), dhn.data_ptr(),
                                           dcn.data_ptr(), dx.data_ptr(), dh.data_ptr(), dc.data_ptr(), dw.data_ptr(), db.data_ptr(),
                                           ws.data_ptr(), ws.numel(), B, Cin, Ch, H, W, st), "cell_bwd")
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    L.plc_debug_select_timeline(sel)
    L.plc_debug_set_timeline(tl.data_ptr())
    run()
    torch.cuda.synchronize()
    L.plc_debug_set_timeline(None)
    raw = tl.cpu().numpy().reshape(-1, waves, 10).astype(np.float64) / 100.0
    raw = raw[raw[:, 0, 9] > 0]
    t0 = raw[:, :, 0].min()
    print(f"cell ({Cin},{Ch}) {'data' if sel == 777 else 'weight'} gradient, B={B} {H}x{W}: {raw.shape[0]} workgroups, kernel span {raw[:, :, 9].max() - t0:.2f} us")
    print(f"  wave lifetime mean {(raw[:, :, 9] - raw[:, :, 0]).mean():.2f} us; start spread {raw[:, :, 0].max() - t0:.2f} us")
    for k, nm in names or [(1, "prologue (weights, window 0)"), (2, "MFMA loops"), (3, "exchange barrier + partial writes"),
                  (4, "wait for rows in flight"), (5, "barrier"), (7, "partial-sum reads"), (8, "next request"), (6, "epilogue stores")]:
        col = raw[:, :, k]
        print(f"  {nm:36s} mean {col.mean():6.2f}  min {col.min():6.2f}  max {col.max():6.2f} us per wave")


cell_bwd(4, 16, 32, 256, 256)
cell_bwd(40, 16, 32, 256, 256, sel=778, waves=8,
         names=[(1, "first tile"), (2, "next tile's loads requested"), (3, "MFMA loop"), (4, "wait for the loads"),
                (5, "registers -> LDS"), (6, "barrier")])
