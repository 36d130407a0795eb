"""Edge geometries of the rollout (forward and parameter gradients) on the engine against the CPU
oracle: single time step, tiny and one-pixel-wide grids, odd batch, 1 / 9 / 16 land-use channels,
auxiliary maps at another resolution, and the benchmark resolution (256x256 at scale 1 and through
three upsample blocks), where the persistent kernels walk several tiles per workgroup; with
randomly initialised parameters (not the golden ones).

The oracle's fp32 path is pinned to the reference by the goldens; its bf16-storage emulation rounds
where the engine rounds.  Bounds: forward <= 2e-2 * max|ref| vs fp32 and <= 4e-3 vs the emulation
(the tolerances of test_gpu_parity_fwd.py); gradients vs the emulation <= 3e-2 (max and L2; these
cases have few pixels, see test_gpu_parity_bwd.py on why the fp32 gradients are not the yardstick)."""
import pytest
import torch

from oracle import cpu_oracle as O

pytestmark = pytest.mark.gpu

CASES = [
    # B, T, H, W, scale, L, aux_hw
    (1, 1, 8, 8, 1, 5, None),        # T == 1 (the loss's temporal term is NaN there: gradients via sum())
    (3, 2, 3, 5, 1, 1, None),        # grid smaller than one tile, one land-use channel, odd batch
    (2, 2, 1, 40, 1, 9, None),       # one-pixel-high grid crossing a tile boundary in x, two LU blocks
    (1, 2, 33, 17, 1, 16, None),     # just past a tile boundary in y, the maximum of 16 LU channels
    (2, 3, 6, 4, 2, 5, (7, 9)),      # one upsample block, aux maps at an unrelated resolution
    (1, 2, 5, 6, 4, 3, None),        # two upsample blocks
    (2, 2, 20, 70, 1, 5, None),      # wider than one 64-column strip of the one-channel weight gradient, ragged in both axes
    # benchmark resolution: more tiles / strips than persistent workgroups, so every weight-gradient
    # kernel walks several tiles per workgroup (double-buffered LDS, accumulators kept across tiles)
    (2, 4, 256, 256, 1, 5, None),
    (1, 2, 32, 32, 8, 5, None),      # the x8 downscaling configuration: three upsample blocks up to 256x256
    (1, 20, 24, 40, 1, 5, None),     # the longest horizon of the benchmark configurations (T = 20)
]


@pytest.mark.parametrize("geom", [(24, 40, 2), (256, 256, 1)])
def test_inference_rollout_captures_into_a_hip_graph(gpu, geom):
    """The library calls are plain stream work (no allocation, no synchronisation), so a caller can
    capture the whole inference rollout with torch.cuda.graph; the replay is bit-identical.  The second
    geometry is large enough for the rollout to fork onto the library's auxiliary streams (two half-batch
    chains, attention branches on the side stream): the capture must follow the event forks and joins."""
    from pl_convlstm_gan_amd import Generator
    H, W, s = geom
    torch.manual_seed(1)
    m = Generator(hidden_dims=[16, 32], lu_channels=5, scale_factor=s).to(gpu)
    rain = torch.rand(2, 3, 1, H, W, device=gpu)
    dem = torch.rand(2, 1, H * s, W * s, device=gpu)
    lu = torch.rand(2, 5, H * s, W * s, device=gpu)
    with torch.no_grad():
        ref = m(rain, dem, lu).clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():
        m(rain, dem, lu)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.no_grad(), torch.cuda.graph(graph):
        out = m(rain, dem, lu)
    rain.mul_(0.5)                       # new input values in the captured buffers
    with torch.no_grad():
        ref2 = m(rain, dem, lu).clone()
        # a call of another size drops the model's cached workspace; the graph must not have been using it
        m(rain[:1].contiguous(), dem[:1].contiguous(), lu[:1].contiguous())
        junk = [torch.full((1 << 20,), float("nan"), device=gpu) for _ in range(8)]      # takes over whatever was freed
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, ref2) and not torch.equal(ref2, ref)
    del junk


@pytest.mark.parametrize("shape", [(2, 3, 16, 32, 1), (1, 2, 9, 20, 2)])
def test_training_step_does_not_depend_on_workspace_contents(gpu, monkeypatch, shape):
    """The zero initial states are never stored: no kernel may read slot 0 of the state arrays (or
    any other byte it did not write).  A forward + backward on workspaces pre-filled with 0xFF (NaN
    in bf16 and fp32) must equal, bit for bit, the same step on zero-filled workspaces."""
    from pl_convlstm_gan_amd import Generator, ops
    B, T, H, W, s = shape

    def run(fill):
        monkeypatch.setattr(ops, "workspace", lambda nbytes, device: torch.full(
            (max(int(nbytes), 256),), fill, dtype=torch.uint8, device=device))
        torch.manual_seed(3)
        m = Generator(hidden_dims=[16, 32], lu_channels=5, scale_factor=s).to(gpu)
        g = torch.Generator().manual_seed(4)
        rain = torch.rand(B, T, 1, H, W, generator=g).to(gpu)
        dem = torch.rand(B, 1, H * s, W * s, generator=g).to(gpu)
        lu = torch.rand(B, 5, H * s, W * s, generator=g).to(gpu)
        out = m(rain, dem, lu)
        (out * out).sum().backward()
        return out.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()}

    out0, g0 = run(0)
    out1, g1 = run(255)
    assert torch.isfinite(out1).all()
    assert torch.equal(out0, out1)
    for k in g0:
        assert torch.isfinite(g1[k]).all(), k
        assert torch.equal(g0[k], g1[k]), k


def relmax(a, b):
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-20)


def rel_l2(a, b):
    return float((a - b).norm()) / max(float(b.norm()), 1e-20)


@pytest.mark.parametrize("case", CASES)
def test_edge_geometry_forward_and_grads(gpu, case):
    from pl_convlstm_gan_amd import Generator
    B, T, H, W, s, L, aux_hw = case
    torch.manual_seed(100 + H * W + L)
    m = Generator(hidden_dims=[16, 32], lu_channels=L, scale_factor=s)
    gen = torch.Generator().manual_seed(7 + B + T)
    rain = 10 * torch.rand(B, T, 1, H, W, generator=gen) * (torch.rand(B, T, 1, H, W, generator=gen) > 0.5)
    ah, aw = aux_hw if aux_hw else (H * s, W * s)
    dem = torch.rand(B, 1, ah, aw, generator=gen)
    lu = torch.rand(B, L, ah, aw, generator=gen)
    m = m.to(gpu)
    out = m(rain.to(gpu), dem.to(gpu), lu.to(gpu))          # creates the lazy blocks, with autograd
    params = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    assert out.shape == (B, T, 1, H * s, W * s)

    ref = O.generator_forward(params, rain, dem, lu, scale_factor=s)
    leaf = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    emu = O.generator_forward(leaf, rain, dem, lu, scale_factor=s, emulate="bf16")
    scale = float(ref.abs().max())
    assert relmax(out.detach().cpu(), ref) <= 2e-2
    assert float((out.detach().cpu() - emu.detach()).abs().max()) <= 4e-3 * scale

    # a smooth scalar of the output: gradient = a fixed positive random field (no L1 sign flips, and
    # no cancellation in the bias gradients, whose incoming gradient the engine carries as bf16)
    wgt = torch.rand(out.shape, generator=gen) + 0.5
    (out * wgt.to(gpu)).sum().backward()
    (emu * wgt).sum().backward()
    worst = {}
    for k, p in m.named_parameters():
        g_ref = leaf[k].grad
        worst[k] = (relmax(p.grad.cpu(), g_ref), rel_l2(p.grad.cpu(), g_ref))
    print(case, {k: (round(a, 4), round(b, 4)) for k, (a, b) in worst.items()})
    for k, (emax, el2) in worst.items():
        assert emax <= 3e-2 and el2 <= 3e-2, (k, emax, el2)


@pytest.mark.parametrize("shape", [(2, 10, 64, 64, 1), (4, 10, 32, 32, 8), (1, 3, 20, 50, 2)])
def test_single_launch_recurrence_vs_oracle(gpu, monkeypatch, shape):
    """On small grids the whole recurrence (2 T cell steps) runs as ONE persistent launch whose
    workgroups meet at a device-wide barrier between steps.  Checked against the oracle (forward and
    parameter gradients, the bounds of test_edge_geometry_forward_and_grads), and bit for bit against
    the per-step launches (PLC_NO_PERSISTENT=1), which run the same tile code."""
    import os
    from pl_convlstm_gan_amd import Generator
    B, T, H, W, s = shape
    torch.manual_seed(11)
    m = Generator(hidden_dims=[16, 32], lu_channels=5, scale_factor=s).to(gpu)
    gen = torch.Generator().manual_seed(12)
    rain = 10 * torch.rand(B, T, 1, H, W, generator=gen) * (torch.rand(B, T, 1, H, W, generator=gen) > 0.5)
    dem = torch.rand(B, 1, H * s, W * s, generator=gen)
    lu = torch.rand(B, 5, H * s, W * s, generator=gen)
    wgt = torch.rand(B, T, 1, H * s, W * s, generator=gen) + 0.5

    def run():
        for p in m.parameters():
            p.grad = None
        with torch.no_grad():
            inf = m(rain.to(gpu), dem.to(gpu), lu.to(gpu)).clone()
        out = m(rain.to(gpu), dem.to(gpu), lu.to(gpu))
        (out * wgt.to(gpu)).sum().backward()
        return inf, out.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()}

    monkeypatch.delenv("PLC_NO_PERSISTENT", raising=False)
    inf_a, out_a, g_a = run()
    monkeypatch.setenv("PLC_NO_PERSISTENT", "1")
    inf_b, out_b, g_b = run()
    monkeypatch.delenv("PLC_NO_PERSISTENT", raising=False)
    assert torch.equal(inf_a, inf_b) and torch.equal(out_a, out_b) and torch.equal(inf_a, out_a)
    for k in g_a:
        assert torch.equal(g_a[k], g_b[k]), k

    params = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    ref = O.generator_forward(params, rain, dem, lu, scale_factor=s)
    leaf = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    emu = O.generator_forward(leaf, rain, dem, lu, scale_factor=s, emulate="bf16")
    assert relmax(out_a.cpu(), ref) <= 2e-2
    assert float((out_a.cpu() - emu.detach()).abs().max()) <= 4e-3 * float(ref.abs().max())
    (emu * wgt).sum().backward()
    for k, gr in g_a.items():
        assert relmax(gr.cpu(), leaf[k].grad) <= 3e-2 and rel_l2(gr.cpu(), leaf[k].grad) <= 3e-2, k


def test_rollouts_from_two_host_threads(gpu):
    """Two host threads, each with its own model, input and stream, call the rollout at the same time at a size
    where the library forks work onto its per-device auxiliary streams (whose events all calls share): every
    result must equal the one the same call gives alone."""
    import threading
    from pl_convlstm_gan_amd import Generator
    jobs = []
    for seed in (1, 2):
        torch.manual_seed(seed)
        m = Generator(hidden_dims=[16, 32], lu_channels=5, scale_factor=1).to(gpu)
        g = torch.Generator().manual_seed(seed + 10)
        B, T, H, W = 4, 3, 256, 256
        rain = (10 * torch.rand(B, T, 1, H, W, generator=g) * (torch.rand(B, T, 1, H, W, generator=g) > 0.7)).to(gpu)
        dem, lu = torch.rand(B, 1, H, W, generator=g).to(gpu), torch.rand(B, 5, H, W, generator=g).to(gpu)
        with torch.no_grad():
            alone = m(rain, dem, lu).clone()
        jobs.append((m, rain, dem, lu, alone))
    torch.cuda.synchronize()
    results, errors = [None, None], []

    def work(k):
        try:
            m, rain, dem, lu, _ = jobs[k]
            s = torch.cuda.Stream(device=gpu)
            with torch.cuda.stream(s), torch.no_grad():
                outs = [m(rain, dem, lu).clone() for _ in range(6)]
            s.synchronize()
            results[k] = outs
        except Exception as e:      # noqa: BLE001  (reported by the assertion below)
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for k in range(2):
        for out in results[k]:
            assert torch.equal(out, jobs[k][4])


def test_a_call_that_fails_after_forking_still_joins_its_streams(gpu):
    """Both rollout calls put part of their work on auxiliary streams of the library.  A call that fails after
    such a fork must join them before it returns (else a stream capture ends with un-joined work, and eagerly
    the auxiliary stream is still using the caller's workspace after the error).  PLC_DEBUG_FAIL_AFTER_FORK=1
    injects the failure; run in a fresh process because the library reads the switch once."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, torch\n"
        f"sys.path.insert(0, {root!r})\n"
        "from pl_convlstm_gan_amd import Generator\n"
        "gpu = torch.device('cuda:0')\n"
        "torch.manual_seed(0)\n"
        "m = Generator(hidden_dims=[16, 32], lu_channels=5, scale_factor=1).to(gpu)\n"
        "g = torch.Generator().manual_seed(1)\n"
        "B, T, H, W = 2, 3, 256, 256          # large enough for the two half-batch chains\n"
        "rain, dem, lu = (torch.rand(B, T, 1, H, W, generator=g).to(gpu), torch.rand(B, 1, H, W, generator=g).to(gpu),\n"
        "                 torch.rand(B, 5, H, W, generator=g).to(gpu))\n"
        "def attempt():\n"
        "    try:\n"
        "        with torch.no_grad():\n"
        "            m(rain, dem, lu)\n"
        "    except RuntimeError as e:\n"
        "        assert 'injected failure' in str(e), e\n"
        "        return True\n"
        "    return False\n"
        "assert attempt()\n"
        "torch.cuda.synchronize()\n"
        "side = torch.cuda.Stream()\n"
        "side.wait_stream(torch.cuda.current_stream())\n"
        "graph = torch.cuda.CUDAGraph()\n"
        "with torch.cuda.graph(graph):          # ends with an error if anything forked is left un-joined\n"
        "    assert attempt()\n"
        "graph.replay()\n"
        "torch.cuda.synchronize()\n"
        "print('joined')\n")
    env = dict(os.environ, PLC_DEBUG_FAIL_AFTER_FORK="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=150)
    assert r.returncode == 0 and "joined" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


def test_long_rollout_gradients_with_open_forget_gates(gpu):
    """The backward pass carries dL/dc from step to step as bf16.  That running sum passes through the product
    of the forget gates of all later steps, so its rounding compounds with T: checked here against autograd
    of the oracle at T = 32 with the forget-gate biases raised to +3 (f ~ 0.95: the cell state's gradient
    survives the whole rollout), to the bounds of the short-rollout parity tests (3e-2 max and L2)."""
    from pl_convlstm_gan_amd import Generator
    B, T, H, W = 1, 32, 24, 40
    torch.manual_seed(31)
    m = Generator(hidden_dims=[16, 32], lu_channels=5, scale_factor=1)
    with torch.no_grad():
        for cell, ch in ((m.cell1, 16), (m.cell2, 32)):
            cell.conv.bias[ch:2 * ch] += 3.0            # gate order i, f, o, g
    gen = torch.Generator().manual_seed(32)
    rain = 10 * torch.rand(B, T, 1, H, W, generator=gen) * (torch.rand(B, T, 1, H, W, generator=gen) > 0.5)
    dem = torch.rand(B, 1, H, W, generator=gen)
    lu = torch.rand(B, 5, H, W, generator=gen)
    # only the LAST frames carry gradient, so every parameter gradient reaches the early steps through the recurrence
    wgt = torch.zeros(B, T, 1, H, W)
    wgt[:, -2:] = torch.rand(B, 2, 1, H, W, generator=gen) + 0.5
    m = m.to(gpu)
    out = m(rain.to(gpu), dem.to(gpu), lu.to(gpu))
    (out * wgt.to(gpu)).sum().backward()
    params = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    leaf = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    emu = O.generator_forward(leaf, rain, dem, lu, scale_factor=1, emulate="bf16")
    assert float((out.detach().cpu() - emu.detach()).abs().max()) <= 4e-3 * float(emu.detach().abs().max())
    (emu * wgt).sum().backward()
    worst = {k: (relmax(p.grad.cpu(), leaf[k].grad), rel_l2(p.grad.cpu(), leaf[k].grad)) for k, p in m.named_parameters()}
    print({k: (round(a, 4), round(b, 4)) for k, (a, b) in worst.items()})
    for k, (emax, el2) in worst.items():
        assert emax <= 3e-2 and el2 <= 3e-2, (k, emax, el2)
