"""Multi-process (world_size 2, gloo, CPU) tests of the data-parallel step logic in
pl_convlstm_gan_amd/training.py: flat-gradient all-reduce before the clip, collective NaN skip,
identical parameters on every rank after a step.  The model here is a small CPU stand-in with
the Generator's calling convention (the HIP rollout itself needs a GPU); what is under test is
the exchange logic around it."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn


class TinyRollout(nn.Module):
    """[B,T,1,H,W] -> [B,T,1,H,W]; same forward signature as Generator."""

    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(1, 1, 3, padding=1)
        self.gate = nn.Conv2d(1 + 2, 1, 1)

    def forward(self, rain_lr, dem, lu, input_grid_size=None):
        B, T, C, H, W = rain_lr.shape
        y = self.conv(rain_lr.reshape(B * T, C, H, W)).reshape(B, T, 1, H, W)
        g = torch.sigmoid(self.gate(torch.cat((dem, lu), dim=1))).unsqueeze(1)
        return y * g


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _batch(seed, B=2, T=3, H=6, W=5, N=4, nan=False):
    g = torch.Generator().manual_seed(seed)
    rain = torch.rand(B, T, 1, H, W, generator=g)
    dem = torch.rand(B, 1, H, W, generator=g)
    lu = torch.rand(B, 2, H, W, generator=g)
    coords = torch.randint(0, min(H, W), (N, 2), generator=g).unsqueeze(0).repeat(B, 1, 1)
    values = 5 * torch.rand(B, T, N, generator=g)
    if nan:
        values = values - 100.0     # log1p(<-1) -> NaN loss (reference quirk)
    return rain, dem, lu, coords, values


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pl_convlstm_gan_amd.losses import CombinedLoss
    from pl_convlstm_gan_amd.training import train_step
    torch.manual_seed(0)
    model = TinyRollout()
    opt = torch.optim.Adam(model.parameters(), lr=5e-4)
    loss_mod = CombinedLoss()
    # step 1: every rank has a different shard
    l1, _, _, skipped1 = train_step(model, loss_mod, opt, *_batch(10 + rank), max_norm=0.5)
    p1 = [p.detach().clone() for p in model.parameters()]
    # step 2: rank 1's shard produces a NaN loss -> BOTH ranks must skip
    l2, _, _, skipped2 = train_step(model, loss_mod, opt, *_batch(20 + rank, nan=(rank == 1)), max_norm=0.5)
    p2 = [p.detach().clone() for p in model.parameters()]
    torch.save({"skipped1": skipped1, "skipped2": skipped2, "p1": p1, "p2": p2, "l1": float(l1)},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def test_dp_step_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"rank{i}.pt", weights_only=True) for i in range(world)]
    assert r[0]["skipped1"] is False and r[1]["skipped1"] is False
    assert r[0]["skipped2"] is True and r[1]["skipped2"] is True
    for a, b in zip(r[0]["p1"], r[1]["p1"]):
        assert torch.equal(a, b)                       # identical update on both ranks
    for a, b in zip(r[0]["p1"], r[0]["p2"]):
        assert torch.equal(a, b)                       # skipped step changed nothing

    # single-process reference: the mean of the two shard gradients, clipped, one Adam step
    from pl_convlstm_gan_amd.losses import CombinedLoss
    torch.manual_seed(0)
    model = TinyRollout()
    opt = torch.optim.Adam(model.parameters(), lr=5e-4)
    loss_mod = CombinedLoss()
    grads = []
    for rank in range(world):
        model.zero_grad()
        rain, dem, lu, coords, values = _batch(10 + rank)
        out = model(rain, dem, lu)
        loss, _ = loss_mod(out, rain, coords, values, 1.0)
        loss.backward()
        grads.append([p.grad.clone() for p in model.parameters()])
    for p, g0, g1 in zip(model.parameters(), *grads):
        p.grad = (g0 + g1) / 2
    torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=0.5)
    opt.step()
    for p, q in zip(model.parameters(), r[0]["p1"]):
        assert torch.allclose(p.detach(), q, atol=1e-7, rtol=1e-6)


def test_bucket_roundtrip_and_clip():
    from pl_convlstm_gan_amd.training import GradBucket, clip_flat_
    m = TinyRollout()
    for p in m.parameters():
        p.grad = torch.randn_like(p)
    ref = [p.grad.clone() for p in m.parameters()]
    total = torch.sqrt(sum((g ** 2).sum() for g in ref))
    b = GradBucket(m.parameters())
    flat = b.gather()
    assert flat.numel() == sum(p.numel() for p in m.parameters())
    norm = clip_flat_(flat, 0.5)
    b.scatter()
    assert torch.allclose(norm, total)
    coef = min(1.0, 0.5 / (float(total) + 1e-6))
    for p, g in zip(m.parameters(), ref):
        assert torch.allclose(p.grad, g * coef, atol=1e-7)


def _worker_uneven(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pl_convlstm_gan_amd.host.trainer import mean_over_ranks, shard_bounds
    from pl_convlstm_gan_amd.losses import CombinedLoss
    from pl_convlstm_gan_amd.training import train_step
    torch.manual_seed(0)
    model = TinyRollout()
    opt = torch.optim.Adam(model.parameters(), lr=5e-4)
    # a batch of 3 over 2 ranks: shards of 1 and 2 samples, sliced the way the Trainer slices
    batch = _batch(30, B=3)
    lo, hi = shard_bounds(3, rank, world)
    shard = tuple(t[lo:hi] for t in batch)
    train_step(model, CombinedLoss(), opt, *shard, max_norm=0.5)
    # a batch of 1 is shorter than the world: every rank must take the same "skip" decision
    short = shard_bounds(1, rank, world)
    torch.save({"p": [p.detach().clone() for p in model.parameters()], "n": hi - lo, "short": short is None,
                "mean": mean_over_ranks(float(rank + 1))}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def test_dp_uneven_shards_give_the_per_sample_mean(tmp_path):
    """Shards of unequal size: the exchanged gradient is the per-sample mean of the global batch (each
    rank weighted by its sample count), i.e. the step equals a single-process step on the whole batch;
    a batch shorter than the world is skipped by every rank; epoch scalars are averaged over ranks."""
    world = 2
    mp.spawn(_worker_uneven, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"rank{i}.pt", weights_only=True) for i in range(world)]
    assert sorted(x["n"] for x in r) == [1, 2]
    assert r[0]["short"] and r[1]["short"]
    assert r[0]["mean"] == r[1]["mean"] == 1.5
    for a, b in zip(r[0]["p"], r[1]["p"]):
        assert torch.equal(a, b)
    from pl_convlstm_gan_amd.losses import CombinedLoss
    torch.manual_seed(0)
    model = TinyRollout()
    opt = torch.optim.Adam(model.parameters(), lr=5e-4)
    rain, dem, lu, coords, values = _batch(30, B=3)
    loss, _ = CombinedLoss()(model(rain, dem, lu), rain, coords, values, 1.0)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm=0.5)
    opt.step()
    for p, q in zip(model.parameters(), r[0]["p"]):
        assert torch.allclose(p.detach(), q, atol=1e-7, rtol=1e-5)


def test_bucket_keeps_missing_gradients_missing():
    """A parameter without a gradient rides through the exchange as zeros and comes back without one,
    so the optimizer skips it exactly as torch.optim would."""
    from pl_convlstm_gan_amd.training import GradBucket
    m = TinyRollout()
    m.conv.weight.grad = torch.ones_like(m.conv.weight)
    b = GradBucket(m.parameters())
    flat = b.gather()
    assert flat.numel() == sum(p.numel() for p in m.parameters())
    b.scatter()
    assert m.conv.weight.grad is not None and m.conv.bias.grad is None and m.gate.weight.grad is None


def _worker_unowned(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pl_convlstm_gan_amd.losses import CombinedLoss
    from pl_convlstm_gan_amd.training import train_step
    torch.manual_seed(0)
    model = TinyRollout()
    # the reference's arrangement: part of the model (its lazily created upsample blocks, here `gate`) is not
    # in the optimizer, so nobody zeroes its gradient: it accumulates over the steps and counts in every clip
    opt = torch.optim.Adam(model.conv.parameters(), lr=5e-4)
    flags, params, loose = [], [], []
    for k, nan_rank in enumerate((None, 1, None, None)):
        _, _, _, s = train_step(model, CombinedLoss(), opt, *_batch(40 + 10 * k + rank, nan=(rank == nan_rank)), max_norm=0.5)
        flags.append(s)
        params.append([p.detach().clone() for p in model.parameters()])
        loose.append([p.grad.detach().clone() for p in model.gate.parameters()])
    torch.save({"flags": flags, "params": params, "loose": loose}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


def test_dp_nan_on_one_rank_with_parameters_outside_the_optimizer(tmp_path):
    """good / NaN on rank 1 only / good / good, with parameters the optimizer does not own: both ranks skip the
    second batch before its backward pass (src/training/trainer.py:306-310), so the un-zeroed gradients stay
    finite, the skipped step changes nothing, and both ranks hold identical parameters and accumulated gradients."""
    world = 2
    mp.spawn(_worker_unowned, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"rank{i}.pt", weights_only=True) for i in range(world)]
    assert r[0]["flags"] == r[1]["flags"] == [False, True, False, False]
    for step in range(4):
        for a, b in zip(r[0]["params"][step], r[1]["params"][step]):
            assert torch.equal(a, b) and torch.isfinite(a).all()
        for a, b in zip(r[0]["loose"][step], r[1]["loose"][step]):
            assert torch.equal(a, b) and torch.isfinite(a).all()
    for a, b in zip(r[0]["params"][0], r[0]["params"][1]):
        assert torch.equal(a, b)
    for a, b in zip(r[0]["loose"][0], r[0]["loose"][1]):
        assert torch.equal(a, b)                       # the skipped batch added nothing to the accumulated gradient
    assert not torch.equal(r[0]["loose"][1][0], r[0]["loose"][2][0])
    # the un-owned parameters never move
    for a, b in zip(r[0]["params"][0][2:], r[0]["params"][3][2:]):
        assert torch.equal(a, b)


def _trainer_worker(rank, world, port, cfg_dir, out_dir):
    """The host Trainer with CPU stand-ins for the GPU-only model and optimizer, over gloo."""
    os.environ["PLC_SHUFFLE_SEED"] = "3"
    if world > 1:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from pl_convlstm_gan_amd.host import config as C
    from pl_convlstm_gan_amd.host import trainer as TR
    C.CONFIG_DIR = cfg_dir

    class Upsampling(TinyRollout):
        def __init__(self, lu_channels=5, **kw):
            super().__init__()
            self.gate = nn.Conv2d(1 + lu_channels, 1, 1)
            self.upsample_blocks = None

        def forward(self, rain_lr, dem, lu, input_grid_size=None):
            B, T, C, H, W = rain_lr.shape
            y = self.conv(rain_lr.reshape(B * T, C, H, W))
            y = torch.nn.functional.interpolate(y, size=dem.shape[-2:], mode="nearest").reshape(B, T, 1, *dem.shape[-2:])
            g = torch.sigmoid(self.gate(torch.cat((dem, lu), dim=1))).unsqueeze(1)
            return y * g

    class CpuTrainer(TR.Trainer):
        def _make_model(self, **kw):
            torch.manual_seed(0)
            return Upsampling(lu_channels=kw["lu_channels"])

        def _make_optimizer(self, params, lr):
            return torch.optim.Adam(params, lr=lr)

    t = CpuTrainer("synthetic")
    t.device = torch.device("cpu")
    t.train()
    seen = sum(len(idx) for idx in t.train_loader.batch_sampler) if world > 1 else None
    torch.save({"history": {k: [float(x) for x in v] for k, v in t.history.items()}, "seen": seen,
                "params": [p.detach().clone() for p in t.model.parameters()]}, os.path.join(out_dir, f"w{world}_rank{rank}.pt"))
    if world > 1:
        dist.destroy_process_group()


def test_trainer_history_is_identical_on_every_rank(tmp_path):
    """The host Trainer under data parallelism (two ranks over gloo, CPU stand-ins for model and optimizer): every
    rank loads only its slice of each shuffled batch, validation batches are dealt out over the ranks, and both
    ranks end with the same history, the same decisions and the same parameters -- which also equal a single
    process's run over the same shuffled batches up to floating-point summation order."""
    from pl_convlstm_gan_amd.host import config as C
    from pl_convlstm_gan_amd.host.datasets import write_synthetic_dataset
    paths = write_synthetic_dataset(str(tmp_path / "basin"), start_year=2012, end_year=2013, lr_hw=(6, 5), scale=2,
                                    n_stations=6)
    cfg = C.Config()
    for k, v in paths.items():
        setattr(cfg.data, k, v)
    cfg.model.scale_factor = 2
    cfg.training.batch_size = 64
    cfg.training.epochs = 2
    cfg.training.train_years, cfg.training.val_years, cfg.training.test_years = [2012, 2012], [2013, 2013], [2013, 2013]
    cfg.output.output_dir = str(tmp_path / "out")
    cfg.output.log_interval = 1000
    cfg_dir = tmp_path / "configs"
    cfg.to_yaml(str(cfg_dir / "synthetic.yaml"))
    mp.spawn(_trainer_worker, args=(2, _free_port(), str(cfg_dir), str(tmp_path)), nprocs=2, join=True)
    mp.spawn(_trainer_worker, args=(1, 0, str(cfg_dir), str(tmp_path)), nprocs=1, join=True)
    r0, r1 = (torch.load(tmp_path / f"w2_rank{i}.pt", weights_only=True) for i in range(2))
    one = torch.load(tmp_path / "w1_rank0.pt", weights_only=True)
    assert r0["history"] == r1["history"] and r0["history"]["epoch"] == [0.0, 1.0]
    for a, b in zip(r0["params"], r1["params"]):
        assert torch.equal(a, b)
    n_train = 366 - 5
    assert r0["seen"] + r1["seen"] == n_train and abs(r0["seen"] - r1["seen"]) <= 6     # each rank loads its own half only
    # against the single process: what the ranks decide on (validation RMSE, learning rate) and where training ends.
    # (the TRAIN loss / RMSE columns are means of per-rank batch figures, i.e. means over half batches, and agree
    # only approximately)
    for a, b in zip(one["params"], r0["params"]):
        assert torch.allclose(a, b, atol=1e-6, rtol=1e-4)
    assert one["history"]["learning_rate"] == r0["history"]["learning_rate"]
