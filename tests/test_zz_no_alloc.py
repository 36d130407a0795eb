"""The allocation-free ConvLSTM cell entry point (plc_convlstm_cell_fwd_noalloc), called through the
C ABI: its staging area is a fixed array inside the library instead of a caller workspace.  It runs
the same kernels as plc_convlstm_cell_fwd, so the two must agree bit for bit -- on two calls in a
row, the second on a smaller input (the array keeps the first call's bytes) -- and a call whose
staging would not fit must be refused with nothing launched."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(B, Cin, Ch, H, W, seed, dev):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cin, H, W, generator=g).to(dev)
    h = torch.tanh(torch.randn(B, Ch, H, W, generator=g)).to(dev)
    c = torch.randn(B, Ch, H, W, generator=g).to(dev)
    w = (0.1 * torch.randn(4 * Ch, Cin + Ch, 3, 3, generator=g)).to(dev)
    b = (0.1 * torch.randn(4 * Ch, generator=g)).to(dev)
    return x, h, c, w, b


def test_noalloc_cell_matches_the_workspace_entry_point(gpu):
    from pl_convlstm_gan_amd import _lib, ops
    L = _lib.lib()
    st = ops._stream()
    limit = L.plc_convlstm_cell_noalloc_max_bytes()
    assert limit == 32 << 20
    # second call smaller than the first, other channel widths, ragged grid
    for (B, Cin, Ch, H, W, seed) in [(2, 16, 32, 96, 128, 1), (1, 16, 16, 19, 45, 2), (1, 16, 32, 256, 256, 3)]:
        x, h, c, w, b = _case(B, Cin, Ch, H, W, seed, gpu)
        need = L.plc_convlstm_cell_workspace_bytes(B, Cin, Ch, H, W)
        assert 0 < need <= limit
        ws = torch.empty(need, dtype=torch.uint8, device=gpu)
        h_ref, c_ref = torch.empty_like(h), torch.empty_like(c)
        _lib.check(L.plc_convlstm_cell_fwd(x.data_ptr(), h.data_ptr(), c.data_ptr(), w.data_ptr(), b.data_ptr(),
                                           h_ref.data_ptr(), c_ref.data_ptr(), ws.data_ptr(), need,
                                           B, Cin, Ch, H, W, st), "plc_convlstm_cell_fwd")
        h_out, c_out = torch.full_like(h, float("nan")), torch.full_like(c, float("nan"))

        def call():
            _lib.check(L.plc_convlstm_cell_fwd_noalloc(x.data_ptr(), h.data_ptr(), c.data_ptr(), w.data_ptr(), b.data_ptr(),
                                                       h_out.data_ptr(), c_out.data_ptr(), B, Cin, Ch, H, W, ops._stream()),
                       "plc_convlstm_cell_fwd_noalloc")
        # one unmeasured call per shape class: the first use on a device looks the array's address up and raises
        # the kernels' LDS limits (one-time host work that the documentation asks callers to do before a capture)
        call()
        torch.cuda.synchronize()
        # memory_allocated() only sees PyTorch's caching allocator; mem_get_info() is the driver's free memory
        # and would also show a hipMalloc made inside the library
        before, free_before = torch.cuda.memory_allocated(), torch.cuda.mem_get_info()[0]
        h_out.fill_(float("nan"))
        c_out.fill_(float("nan"))
        call()
        torch.cuda.synchronize()
        assert torch.cuda.memory_allocated() == before and torch.cuda.mem_get_info()[0] == free_before
        assert torch.isfinite(h_out).all() and torch.isfinite(c_out).all()
        assert torch.equal(h_out, h_ref) and torch.equal(c_out, c_ref)
        # and under stream capture, where any allocating or synchronising HIP call inside the library is an error
        h_out.fill_(float("nan"))
        c_out.fill_(float("nan"))
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            call()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(h_out, h_ref) and torch.equal(c_out, c_ref)


def test_noalloc_cell_refuses_what_does_not_fit(gpu):
    from pl_convlstm_gan_amd import _lib, ops
    L = _lib.lib()
    B, Cin, Ch, H, W = 4, 16, 32, 256, 256
    assert L.plc_convlstm_cell_workspace_bytes(B, Cin, Ch, H, W) > L.plc_convlstm_cell_noalloc_max_bytes()
    x, h, c, w, b = _case(B, Cin, Ch, H, W, 4, gpu)
    h_out, c_out = torch.zeros_like(h), torch.zeros_like(c)
    rc = L.plc_convlstm_cell_fwd_noalloc(x.data_ptr(), h.data_ptr(), c.data_ptr(), w.data_ptr(), b.data_ptr(),
                                         h_out.data_ptr(), c_out.data_ptr(), B, Cin, Ch, H, W, ops._stream())
    assert rc == -3 and b"built-in area holds" in L.plc_last_error()
    torch.cuda.synchronize()
    assert float(h_out.abs().max()) == 0.0 and float(c_out.abs().max()) == 0.0      # nothing ran
