"""Worker of tests/test_gpu_dp_rccl.py::test_two_ranks_on_one_gpu_nan_on_one_rank (not a test module).

Two ranks share the one GPU of the box and exchange over gloo (RCCL needs one GPU per rank); what is
under test is the data-parallel step's logic on the real Generator + FusedAdam, in the reference
Trainer's arrangement: the optimizer is built before the first forward, so the lazily created upsample
block is outside it and its gradient accumulates.  Plan: good, NaN on rank 1 only, good, good."""
import datetime
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def batch(seed, dev, nan=False, B=2, T=3, H=12, W=20, N=6):
    g = torch.Generator().manual_seed(seed)
    rain = 10 * torch.rand(B, T, 1, H, W, generator=g)
    dem = torch.rand(B, 1, 2 * H, 2 * W, generator=g)
    lu = torch.rand(B, 5, 2 * H, 2 * W, generator=g)
    coords = torch.randint(0, min(H, W), (N, 2), generator=g).unsqueeze(0).repeat(B, 1, 1)
    values = 20 * torch.rand(B, T, N, generator=g)
    if nan:
        values = values - 100.0          # log1p(< -1) -> NaN loss
    return [t.to(dev) for t in (rain, dem, lu, coords, values)]


def run(mode, rank, dev):
    from pl_convlstm_gan_amd import Generator
    from pl_convlstm_gan_amd.losses import CombinedLoss
    from pl_convlstm_gan_amd.optim import FusedAdam
    from pl_convlstm_gan_amd.training import train_step
    torch.manual_seed(0)
    model = Generator(hidden_dims=[16, 32], lu_channels=5, scale_factor=2).to(dev)
    opt = FusedAdam(model.parameters(), lr=5e-4)
    flags, snaps = [], []
    for k, nan_rank in enumerate((None, 1, None, None)):
        _, _, _, s = train_step(model, CombinedLoss(), opt, *batch(10 * (k + 1) + rank, dev, nan=(rank == nan_rank)),
                                max_norm=0.5, nan_check=mode)
        flags.append(bool(s))
        snaps.append(torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu())
    lazy = torch.cat([p.grad.detach().reshape(-1) for p in model.upsample_blocks.parameters()]).cpu()
    return flags, snaps, lazy


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))
    dev = torch.device("cuda", 0)

    def same_everywhere(v):
        every = [torch.empty_like(v) for _ in range(world)]
        dist.all_gather(every, v)
        return all(torch.equal(every[0], b) for b in every)

    f_dev, p_dev, lazy_dev = run("device", rank, dev)
    f_sync, p_sync, lazy_sync = run("sync", rank, dev)
    checks = {
        "flags": f_dev == [False, True, False, False] == f_sync,
        "skip_keeps_parameters": torch.equal(p_dev[0], p_dev[1]) and not torch.equal(p_dev[1], p_dev[2]),
        "finite": bool(torch.isfinite(p_dev[3]).all()) and bool(torch.isfinite(lazy_dev).all()),
        "device_equals_sync": all(torch.equal(a, b) for a, b in zip(p_dev, p_sync)) and torch.equal(lazy_dev, lazy_sync),
        "ranks_agree": all(same_everywhere(p) for p in p_dev) and same_everywhere(lazy_dev),
    }
    print(f"rank {rank}: {checks}", flush=True)
    bad = torch.tensor([0.0 if all(checks.values()) else 1.0])
    dist.all_reduce(bad)
    dist.destroy_process_group()
    sys.exit(0 if float(bad) == 0 else 1)


if __name__ == "__main__":
    main()
