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
