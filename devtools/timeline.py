"""Phase timeline of the conv kernels from an instrumented build (-DPLC_TIMELINE) of the library.

    python devtools/timeline.py            # on the GPU box; expects devtools/libplc_tl.so
"""
import collections
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
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
NS = 10


def cell(B, Cin, Ch, H, W, save, placement=False):
    hw = H * W
    x = torch.randn(B * (Cin // 8) * hw * 8, device=dev).to(torch.bfloat16)
    h = torch.randn(B * (Ch // 8) * hw * 8, device=dev).to(torch.bfloat16)
    c = torch.randn(B * Ch * hw, device=dev)
    w = torch.randn(4 * Ch, Cin + Ch, 3, 3, device=dev) * 0.05
    b = torch.zeros(4 * Ch, device=dev)
    wpk = torch.empty(L.plc_packed_weight_bytes(4 * Ch, Cin, Ch, 3), dtype=torch.uint8, device=dev)
    bpk = torch.empty(L.plc_packed_bias_bytes(4 * Ch), dtype=torch.uint8, device=dev)
    _lib.check(L.plc_pack_conv_weight(w.data_ptr(), b.data_ptr(), wpk.data_ptr(), bpk.data_ptr(), 4 * Ch, Cin, Ch, 3, 1, st), "pack")
    h2 = torch.empty_like(h)
    c2 = torch.empty_like(c)
    gates = torch.empty(B * 4 * Ch * hw, device=dev, dtype=torch.bfloat16) if save else None
    n_wg = B * ((H + 15) // 16) * ((W + 31) // 32)
    tl = torch.zeros(n_wg * 4 * NS, dtype=torch.int64, device=dev)

    def run():
        _lib.check(L.plc_convlstm_cell_fwd_cb8(x.data_ptr(), h.data_ptr(), c.data_ptr(), wpk.data_ptr(), bpk.data_ptr(),
                                               h2.data_ptr(), c2.data_ptr(), gates.data_ptr() if save else None,
                                               B, Cin, Ch, H, W, st), "cell")
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    L.plc_debug_set_timeline(tl.data_ptr())
    run()
    torch.cuda.synchronize()
    L.plc_debug_set_timeline(None)
    raw = tl.cpu().numpy().reshape(n_wg, 4, NS)
    t = raw[:, :, :8].astype(np.float64) / 100.0     # us
    t0 = t[:, :, 0].min()
    t -= t0
    names = ["entry", "weights issued", "tile staged (vmcnt 0)", "barrier", "first pass", "pass loop", "last epilogue",
             "stores drained"]
    print(f"cell ({Cin},{Ch}) B={B} {H}x{W} save={save}: {n_wg} workgroups")
    for k, nm in enumerate(names):
        col = t[:, :, k]
        print(f"  {nm:24s} min {col.min():7.2f}  mean {col.mean():7.2f}  max {col.max():7.2f} us")
    d = np.diff(t, axis=2)
    print("  phase means (us):", " ".join(f"{v:6.2f}" for v in d.mean(axis=(0, 1))))
    print("  kernel span:", t[:, :, 7].max(), "us")
    if not placement:
        return
    hw_id = raw[:, :, 8]
    xcc = raw[:, :, 9] & 0xF
    simd = (hw_id >> 4) & 3
    cu = (hw_id >> 8) & 0xF
    sh = (hw_id >> 12) & 1
    se = (hw_id >> 13) & 7
    cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    per_cu = collections.Counter(cu_key.reshape(-1).tolist())
    print("  distinct CUs:", len(per_cu), " waves per CU histogram:", sorted(collections.Counter(per_cu.values()).items()))
    by_cu = collections.defaultdict(list)
    for blk in range(n_wg):
        by_cu[int(cu_key[blk, 0])].append(blk)
    print("  blocks per CU (first 24 CUs by key):", [(k, v) for k, v in sorted(by_cu.items())[:24]])
    print("  block -> (xcc, se, sh, cu) for blocks 0..39:",
          [(int(xcc[b, 0]), int(se[b, 0]), int(sh[b, 0]), int(cu[b, 0])) for b in range(40)])
    print("  simd of waves 0..3 for blocks 0..7:", [[int(simd[b, w]) for w in range(4)] for b in range(8)])
    per_simd = collections.Counter((cu_key * 4 + simd).reshape(-1).tolist())
    print("  waves per SIMD histogram:", sorted(collections.Counter(per_simd.values()).items()))
    # pass-loop time of a wave against the number of waves sharing its SIMD / CU
    loop = t[:, :, 5] - t[:, :, 3]
    for nshare in sorted(set(per_simd.values())):
        sel = np.vectorize(lambda k: per_simd[k] == nshare)(cu_key * 4 + simd)
        print(f"    SIMD shared by {nshare}: {sel.sum()} waves, MFMA phase mean {loop[sel].mean():.2f} max {loop[sel].max():.2f} us")
    for x in sorted(set(xcc.reshape(-1).tolist())):
        sel = xcc == x
        print(f"    XCC {x}: {sel.sum()} waves, MFMA phase mean {loop[sel].mean():.2f}, end mean {t[:, :, 7][sel].mean():.2f} max {t[:, :, 7][sel].max():.2f}")


if __name__ == "__main__":
    cell(4, 16, 32, 256, 256, False, placement=True)
    cell(4, 16, 16, 256, 256, False, placement=True)
