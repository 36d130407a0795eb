"""In-tree build of libplconvlstm.so with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "plc_api.hip")
OUT = os.path.join(HERE, "libplconvlstm.so")
DEPS = sorted(glob.glob(os.path.join(HERE, "csrc", "*.h")) + glob.glob(os.path.join(HERE, "csrc", "*.hip"))) + \
       [os.path.join(os.path.dirname(HERE), "include", "plconvlstm.h"), os.path.abspath(__file__)]


def is_stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in DEPS)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not is_stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -fno-slp-vectorize: the compiler must not pair scalar fp32 arithmetic into packed instructions on its own.
    # On the MI355X a packed fp32 instruction whose SECOND source takes its low half from the high dword
    # (v_pk_fma_f32 ... op_sel:[0,1,0], v_pk_add_f32 ... op_sel:[0,1]) gives wrong results now and then while an
    # MFMA kernel runs on the chip at the same time (devtools/pk_opsel_probe.hip: 1.5e-4 of the instructions;
    # never alone; the plain, op_sel_hi-only, first- and third-source forms never), and the pairing produced
    # such encodings.  The rollouts run several kernels at once, so the library holds no packed fp32 instruction
    # except the hand-written plain ones of the one-channel weight gradient (tests/test_host_logic.py checks the
    # built code object).  Measured cost of the flag: none (c3 / p2 / c1, train and forward, same box).
    cmd = [hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-fno-slp-vectorize", "-shared", "-fPIC", "-o", OUT, SRC]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    build(force=True)
