"""The data-parallel step on real hardware with the RCCL backend: a process group of ONE rank (all a
one-GPU box can host) initialises RCCL and sends the flat gradient buffer, with the NaN flag and the
sample-count element riding in it, through dist.all_reduce on the GPU.  For one rank the exchange is
the identity, so the step must equal, bit for bit, the same step without a process group; a NaN loss
must leave the parameters untouched.  (The N > 1 logic is covered by the world_size-2 gloo tests.)"""
import datetime
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu


def _batch(seed, dev, nan=False, B=2, T=3, H=24, W=40, N=6):
    g = torch.Generator().manual_seed(seed)
    rain = 10 * torch.rand(B, T, 1, H, W, generator=g)
    dem = torch.rand(B, 1, H, W, generator=g)
    lu = torch.rand(B, 5, H, W, generator=g)
    coords = torch.randint(0, min(H, W), (N, 2), generator=g).unsqueeze(0).repeat(B, 1, 1)
    values = 20 * torch.rand(B, T, N, generator=g)
    if nan:
        values = values - 100.0          # log1p(< -1) -> NaN loss (reference behaviour)
    return [t.to(dev) for t in (rain, dem, lu, coords, values)]


def _run(gpu, steps):
    from pl_convlstm_gan_amd import Generator
    from pl_convlstm_gan_amd.losses import CombinedLoss
    from pl_convlstm_gan_amd.optim import FusedAdam
    from pl_convlstm_gan_amd.training import train_step
    torch.manual_seed(0)
    model = Generator(hidden_dims=[16, 32], lu_channels=5, scale_factor=1).to(gpu)
    opt = FusedAdam(model.parameters(), lr=5e-4)
    out = []
    for seed, nan in steps:
        _, _, _, skipped = train_step(model, CombinedLoss(), opt, *_batch(seed, gpu, nan), max_norm=0.5,
                                      nan_check="device")
        out.append((bool(skipped), torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()))
    return out


def test_one_rank_rccl_group_runs_the_exchange(gpu):
    steps = [(10, False), (20, True), (30, False)]
    plain = _run(gpu, steps)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=gpu, timeout=datetime.timedelta(seconds=120))
    try:
        assert dist.get_backend() == "nccl"
        probe = torch.arange(8, dtype=torch.float32, device=gpu)
        dist.all_reduce(probe)                            # RCCL itself answers
        assert torch.equal(probe.cpu(), torch.arange(8, dtype=torch.float32))
        grouped = _run(gpu, steps)
    finally:
        dist.destroy_process_group()
    assert [s for s, _ in plain] == [False, True, False] == [s for s, _ in grouped]
    for (_, a), (_, b) in zip(plain, grouped):
        assert torch.equal(a, b)
    assert torch.equal(grouped[0][1], grouped[1][1]) and not torch.equal(grouped[1][1], grouped[2][1])


def test_pack_and_unpack_of_the_gradient_buffer(gpu):
    """The two library launches around the all-reduce against the torch operators they replace (multiply by the
    rank's sample count, SUM over ranks, divide by the total count; NaN flag and count in the two trailing
    elements).  The SUM of a second rank is played by adding a second packed buffer, as the collective would."""
    import ctypes as C
    from pl_convlstm_gan_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(5)
    shapes = [(16, 3, 3, 3), (16,), (64, 32, 3, 3), (64,), (1, 32, 3, 3), (1,)] + [(7,)] * 50     # > 48 tensors: two table chunks
    st = torch.cuda.current_stream(gpu).cuda_stream

    def rank_grads(seed):
        gg = torch.Generator().manual_seed(seed)
        return [torch.randn(s, generator=gg).to(gpu) for s in shapes]

    def table_of(grads):
        tb = (_lib.OptTensor * len(grads))()
        for k, t in enumerate(grads):
            tb[k].grad, tb[k].numel = t.data_ptr(), t.numel()
        return tb

    total = sum(t.numel() for t in rank_grads(0))
    for nan_rank in (None, 1):
        ga, gb = rank_grads(1), rank_grads(2)
        ref_a = [t.clone() for t in ga]
        ref_b = [t.clone() for t in gb]
        loss_a = torch.tensor([0.25], device=gpu)
        loss_b = torch.tensor([float("nan") if nan_rank == 1 else 0.5], device=gpu)
        fa = torch.empty(total + 2, device=gpu)
        fb = torch.empty(total + 2, device=gpu)
        ta, tb = table_of(ga), table_of(gb)
        _lib.check(L.plc_dp_pack_grads(ta, len(ga), fa.data_ptr(), total + 2, 3.0, loss_a.data_ptr(), st), "pack a")
        _lib.check(L.plc_dp_pack_grads(tb, len(gb), fb.data_ptr(), total + 2, 5.0, loss_b.data_ptr(), st), "pack b")
        assert torch.equal(fa[:total], torch.cat([t.reshape(-1) for t in ref_a]) * 3.0)
        assert fa[total:].tolist() == [0.0, 3.0]
        assert fb[total:].tolist() == [1.0 if nan_rank == 1 else 0.0, 5.0]
        summed = fa + fb                                           # what all_reduce(SUM) leaves on both ranks
        out_a = torch.empty(1, device=gpu)
        _lib.check(L.plc_dp_unpack_grads(ta, len(ga), summed.data_ptr(), total + 2, loss_a.data_ptr(), out_a.data_ptr(), st), "unpack")
        want = (torch.cat([t.reshape(-1) for t in ref_a]) * 3.0 + torch.cat([t.reshape(-1) for t in ref_b]) * 5.0) / 8.0
        got = torch.cat([t.reshape(-1) for t in ga])
        assert torch.equal(got, want)
        if nan_rank is None:
            assert float(out_a) == 0.25
        else:
            assert torch.isnan(out_a).all()                        # the other rank's NaN poisons this rank's guard
    # a wrong buffer length is refused
    with pytest.raises(ValueError):
        _lib.check(L.plc_dp_pack_grads(ta, len(ga), fa.data_ptr(), total + 1, 3.0, None, st), "pack")


def test_two_ranks_on_one_gpu_nan_on_one_rank(gpu):
    """Two ranks (gloo, sharing this box's GPU) run good / NaN-on-rank-1 / good / good with the lazy upsample
    block outside the optimizer: every rank skips the second batch, the block's accumulated gradient stays
    finite, parameters are identical on both ranks and bit-equal to the host-checked mode
    (tests/dp_shared_gpu_worker.py).  Ref: src/training/trainer.py:290-315."""
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dp_shared_gpu_worker.py")
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-u", worker], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=150)[0])
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for rank, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {rank}:\n{o[-3000:]}"
