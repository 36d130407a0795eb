"""CPU-side checks: the C-ABI library loads and exports every declared symbol, the host mirror keeps
the reference's interface (state_dict keys, lazy blocks, geometry), and the product path refuses
to run without a GPU (no CPU fallback)."""
import os
import re

import pytest
import torch

from conftest import ROOT, master_params


def test_library_exports_every_declared_symbol():
    from pl_convlstm_gan_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "plconvlstm.h")).read()
    declared = set(re.findall(r"\b(plc_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations found"
    L = _lib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/plconvlstm.h but not exported"
    assert declared == set(_lib.EXPORTS)
    # the header, the built library and the ctypes mirror are one version
    assert L.plc_version() == _lib.ABI_VERSION == int(re.search(r"#define PLC_VERSION (\d+)", hdr).group(1))


def test_size_queries_without_gpu():
    from pl_convlstm_gan_amd import _lib
    L = _lib.lib()
    # cell2 gate conv: 4 M tiles x 27 k-steps x 64 lanes x 16 B
    assert L.plc_packed_weight_bytes(128, 16, 32, 3) == 4 * 27 * 64 * 16
    assert L.plc_packed_weight_bytes(64, 16, 16, 3) == 2 * 18 * 64 * 16
    assert L.plc_packed_bias_bytes(128) == 4 * 32 * 4
    assert L.plc_convlstm_cell_workspace_bytes(2, 16, 32, 12, 12) > 0
    assert L.plc_convlstm_cell_workspace_bytes(0, 16, 32, 12, 12) == 0


def test_state_dict_matches_reference_names():
    from pl_convlstm_gan_amd import Generator
    from pl_convlstm_gan_amd.models import UpsampleBlock
    m = Generator(hidden_dims=[16, 32], lu_channels=5, scale_factor=8)
    assert m.upsample_blocks is None
    ref = master_params(0)
    sd = m.state_dict()
    assert set(sd) == set(ref)
    for k in ref:
        assert tuple(sd[k].shape) == tuple(ref[k].shape), k
    assert sum(p.numel() for p in m.parameters()) == 85889
    m.upsample_blocks = torch.nn.ModuleList([UpsampleBlock(32, 2) for _ in range(3)])
    m.load_state_dict(master_params(3), strict=True)
    assert sum(p.numel() for p in m.parameters()) == 196865


def test_geometry_and_lazy_blocks():
    from pl_convlstm_gan_amd import Generator
    for s, (nb, rem) in {1: (0, 1), 2: (1, 1.0), 3: (1, 1.5), 4: (2, 1.0), 6: (2, 1.5), 8: (3, 1.0)}.items():
        m = Generator(hidden_dims=[16, 32], lu_channels=5, scale_factor=s)
        n, r, fh, fw = m._geometry(32, 32, None, "cpu")
        assert (n, r) == (nb, rem)
        assert (fh, fw) == (32 * s, 32 * s)
    m = Generator(hidden_dims=[16, 32], lu_channels=5, target_grid_size=(500, 500))
    n, r, fh, fw = m._geometry(32, 32, (4000, 4000), "cpu")
    assert (n, fh, fw) == (3, 256, 256) and m.target_size == (256, 256)


@pytest.mark.skipif(torch.cuda.is_available(), reason="with a GPU, CPU operands are staged onto it instead")
def test_no_cpu_fallback():
    """Without a GPU nothing computes: CPU operands are only ever staged onto the GPU."""
    from pl_convlstm_gan_amd import ConvLSTMCell, Generator
    m = Generator(hidden_dims=[16, 32], lu_channels=5, scale_factor=1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 1, 1, 8, 8), torch.zeros(1, 1, 8, 8), torch.zeros(1, 5, 8, 8))
    c = ConvLSTMCell(16, 16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        c(torch.zeros(1, 16, 8, 8), torch.zeros(1, 16, 8, 8), torch.zeros(1, 16, 8, 8))


def test_drop_in_import_paths():
    from src.models.attention import DEMAttention, LUAttention  # noqa: F401
    from src.models.convlstm import ConvLSTMCell  # noqa: F401
    from src.models.coordconv import add_coord_channels
    from src.models.generator import Generator  # noqa: F401
    y = add_coord_channels(torch.zeros(2, 1, 3, 4))
    assert y.shape == (2, 3, 3, 4)
    assert torch.equal(y[0, 1, :, 0], torch.linspace(0, 1, 3)) and torch.equal(y[0, 2, 0, :], torch.linspace(0, 1, 4))


def test_grad_bucket_carries_an_extra_element():
    """GradBucket.gather(extra): the extra values ride behind the gradients through the same flat
    buffer (the device-side NaN flag of the data-parallel step) and scatter() ignores them."""
    import torch
    from pl_convlstm_gan_amd.training import GradBucket
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(3, 2)), torch.nn.Parameter(torch.randn(5))]
    for p in params:
        p.grad = torch.randn_like(p)
    want = [p.grad.clone() for p in params]
    bucket = GradBucket(params)
    flat = bucket.gather(torch.tensor([1.0]))
    assert flat.numel() == 6 + 5 + 1 and float(bucket.extra()) == 1.0
    flat.mul_(0.5)                                   # what the all-reduce mean does, in place
    bucket.scatter()
    for p, w in zip(params, want):
        assert p.grad.shape == p.shape and torch.equal(p.grad, w * 0.5)
    assert float(bucket.extra()) == 0.5
    assert GradBucket(params).gather().numel() == 11  # without an extra element nothing is appended


def test_bench_refuses_a_rank_count_it_cannot_honour():
    """`python bench.py --gpus N` with N above the box's GPU count, or inside a torchrun environment
    whose WORLD_SIZE disagrees with --gpus, exits non-zero before running anything (it must never print
    a line that says n_gpus: 1 for a --gpus 8 request)."""
    import subprocess
    import sys
    bench = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, bench, "--gpus", "64"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "n_gpus" not in r.stdout and "nothing was run" in r.stderr
    env["WORLD_SIZE"] = "1"
    r = subprocess.run([sys.executable, bench, "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "n_gpus" not in r.stdout and "WORLD_SIZE=1" in r.stderr


def test_built_library_has_no_packed_fp32_with_a_selected_second_source(tmp_path):
    """The device code of the built library is disassembled and searched for packed fp32 instructions whose second
    source takes its low half from the high dword (op_sel bit of source 1 set): that encoding is unreliable on
    the MI355X while an MFMA kernel runs beside it (pl_convlstm_gan_amd/build.py), and the rollouts run kernels
    side by side.  The build switches the compiler's own pairing off; this is the check that nothing brings such
    an instruction back."""
    import re
    import shutil
    import subprocess
    from pl_convlstm_gan_amd import _lib
    llvm = "/opt/rocm/lib/llvm/bin"
    bundler, objdump = os.path.join(llvm, "clang-offload-bundler"), os.path.join(llvm, "llvm-objdump")
    objcopy = shutil.which("objcopy")
    if not (os.path.exists(_lib.LIB_PATH) and os.path.exists(bundler) and os.path.exists(objdump) and objcopy):
        pytest.skip("library or binary tools not available")
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "dev.co")
    subprocess.run([objcopy, "-O", "binary", "--only-section=.hip_fatbin", _lib.LIB_PATH, fat], check=True)
    subprocess.run([bundler, "--unbundle", "--type=o", f"--input={fat}", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    f"--output={co}"], check=True)
    dis = subprocess.run([objdump, "-d", co], check=True, capture_output=True, text=True).stdout
    assert "v_mfma_f32_32x32x16_bf16" in dis            # (it is the library's device code that was read)
    packed = [ln for ln in dis.splitlines() if re.search(r"\bv_pk_(fma|mul|add)_f32\b", ln)]
    bad = [ln.strip() for ln in packed if re.search(r"op_sel:\[\d,1", ln)]
    assert not bad, bad[:5]
    # what remains is the hand-written plain form of the one-channel weight gradient
    assert all("op_sel" not in ln for ln in packed), [ln.strip() for ln in packed if "op_sel" in ln][:5]


def test_live_autograd_graph_detection():
    """training.params_in_a_live_autograd_graph: what GraphedTrainStep refuses to be built beside."""
    import torch
    from pl_convlstm_gan_amd.training import params_in_a_live_autograd_graph as held
    ps = [torch.nn.Parameter(torch.randn(3)), torch.nn.Parameter(torch.randn(2)), torch.nn.Parameter(torch.randn(2), requires_grad=False)]
    assert held(ps) == []
    loss = (ps[0] * 2).sum()
    assert [id(p) for p in held(ps)] == [id(ps[0])]
    loss.backward()
    assert [id(p) for p in held(ps)] == [id(ps[0])]          # a kept loss keeps the nodes, graph buffers freed or not
    detached = loss.detach()
    del loss
    assert held(ps) == [] and held(ps) == []


def test_the_library_issues_no_memset():
    """A hipMemsetAsync inside a captured training step gave wrong replays on this runtime (see
    training.GraphedTrainStep); the library clears memory with kernels and must not import the call."""
    import shutil
    import subprocess
    from pl_convlstm_gan_amd import _lib
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    syms = subprocess.run([nm, "-D", "--undefined-only", _lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    assert "hipMemset" not in syms, [l for l in syms.splitlines() if "hipMemset" in l]
