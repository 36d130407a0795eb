"""FusedAdam (plc_clip_adam_step through the C ABI) against torch.nn.utils.clip_grad_norm_ +
torch.optim.Adam on the same gradients (a floating-point kernel: the torch fp32 operators are its
reference).  Tolerance 2e-6 relative to max|param| per tensor: the two differ only in fp32
rounding order (fma contraction, sum order of the norm)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(16, 3, 3, 3), (16,), (64, 32, 3, 3), (64,), (128, 48, 3, 3), (1, 32, 3, 3), (1,), (7, 5)]


def _make(gpu, seed):
    gen = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter((torch.randn(s, generator=gen) * 0.1).to(gpu)) for s in SHAPES]


def _grads(step, scale):
    gen = torch.Generator().manual_seed(100 + step)
    return [torch.randn(s, generator=gen) * scale for s in SHAPES]


@pytest.mark.parametrize("weight_decay", [0.0, 0.01])
@pytest.mark.parametrize("scale", [1.0, 1e-3])        # clip active / inactive
def test_fused_adam_matches_torch(gpu, weight_decay, scale):
    from pl_convlstm_gan_amd.optim import FusedAdam
    mine, ref = _make(gpu, 1), _make(gpu, 1)
    # the last tensor is not owned by the optimizer (the reference's lazily created blocks): it is
    # clipped and counts in the norm, but is not stepped
    opt_m = FusedAdam(mine[:-1], lr=5e-4, weight_decay=weight_decay)
    opt_r = torch.optim.Adam(ref[:-1], lr=5e-4, weight_decay=weight_decay)
    for step in range(6):
        gs = _grads(step, scale)
        for p, q, g in zip(mine, ref, gs):
            p.grad = g.clone().to(gpu)
            q.grad = g.clone().to(gpu)
        norm_r = torch.nn.utils.clip_grad_norm_(ref, 0.5)
        opt_r.step()
        norm_m = opt_m.clip_and_step(mine, 0.5)
        assert abs(float(norm_m) - float(norm_r)) <= 1e-5 * float(norm_r)
        for p, q in zip(mine, ref):
            assert (p.grad - q.grad).abs().max().item() <= 1e-6 * q.grad.abs().max().item() + 1e-12
    for k, (p, q) in enumerate(zip(mine, ref)):
        assert (p - q).abs().max().item() <= 2e-6 * q.abs().max().item(), k
    assert torch.equal(mine[-1].detach().cpu(), _make("cpu", 1)[-1].detach())       # never stepped
    for p, q in zip(mine[:-1], ref[:-1]):
        sm, sr = opt_m.state[p], opt_r.state[q]
        assert float(sm["step"]) == float(sr["step"]) == 6.0
        assert (sm["exp_avg"] - sr["exp_avg"]).abs().max().item() <= 2e-6 * sr["exp_avg"].abs().max().item()
        assert (sm["exp_avg_sq"] - sr["exp_avg_sq"]).abs().max().item() <= 2e-6 * sr["exp_avg_sq"].abs().max().item()


def test_state_dict_interchanges_with_torch_adam(gpu):
    from pl_convlstm_gan_amd.optim import FusedAdam
    a, b = _make(gpu, 2), _make(gpu, 2)
    opt_a = FusedAdam(a, lr=1e-3)
    for step in range(3):
        for p, g in zip(a, _grads(step, 1.0)):
            p.grad = g.to(gpu)
        opt_a.step()
    # continue in torch.optim.Adam from FusedAdam's state, and in FusedAdam itself
    for p, q in zip(a, b):
        q.data.copy_(p.data)
    opt_b = torch.optim.Adam(b, lr=1e-3)
    opt_b.load_state_dict(copy.deepcopy(opt_a.state_dict()))     # (load_state_dict aliases same-device tensors)
    for step in range(3, 5):
        for p, q, g in zip(a, b, _grads(step, 1.0)):
            p.grad = g.clone().to(gpu)
            q.grad = g.clone().to(gpu)
        opt_a.step()
        opt_b.step()
    for p, q in zip(a, b):
        assert (p - q).abs().max().item() <= 2e-6 * q.abs().max().item()
    # and back: a torch.optim.Adam state_dict loads into FusedAdam
    c = _make(gpu, 2)
    opt_c = FusedAdam(c, lr=1e-3)
    opt_c.load_state_dict(opt_b.state_dict())
    assert float(opt_c.state[c[0]]["step"]) == 5.0


def test_second_param_group_and_many_tensors(gpu):
    """A group added later has its own step count (the general path), and more tensors than one
    launch's table holds are split over launches."""
    from pl_convlstm_gan_amd.optim import FusedAdam
    gen = torch.Generator().manual_seed(3)
    mine = [torch.nn.Parameter(torch.randn(5, 3, generator=gen).to(gpu)) for _ in range(60)]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in mine]
    opt_m, opt_r = FusedAdam(mine[:50], lr=1e-3), torch.optim.Adam(ref[:50], lr=1e-3)
    for step in range(4):
        if step == 2:
            opt_m.add_param_group({"params": mine[50:], "lr": 2e-3})
            opt_r.add_param_group({"params": ref[50:], "lr": 2e-3})
        for p, q in zip(mine, ref):
            g = torch.randn(5, 3, generator=gen)
            p.grad, q.grad = g.clone().to(gpu), g.clone().to(gpu)
        torch.nn.utils.clip_grad_norm_(ref, 0.5)
        opt_r.step()
        opt_m.clip_and_step(mine, 0.5)
    for k, (p, q) in enumerate(zip(mine, ref)):
        assert (p - q).abs().max().item() <= 2e-6 * q.abs().max().item(), k


def test_guarded_step_skips_nan_batches_on_the_device(gpu):
    """clip_and_step(..., guard=loss): the reference's `if torch.isnan(loss): continue`
    (src/training/trainer.py:306-308) decided by the optimizer kernel.  A sequence with NaN guards at
    steps 1 and 4 must end where torch's clip + Adam ends when those batches are simply left out:
    parameters, moments and the step count (tolerances of test_fused_adam_matches_torch)."""
    from pl_convlstm_gan_amd.optim import FusedAdam
    mine, ref = _make(gpu, 5), _make(gpu, 5)
    opt_m = FusedAdam(mine[:-1], lr=5e-4)
    opt_r = torch.optim.Adam(ref[:-1], lr=5e-4)
    nan_steps = {1, 4}
    for step in range(7):
        gs = _grads(step, 1.0)
        bad = step in nan_steps
        for p, g in zip(mine, gs):
            p.grad = (g * float("nan") if bad else g.clone()).to(gpu)      # a NaN loss comes with NaN gradients
        before = [p.detach().clone() for p in mine]
        guard = torch.tensor(float("nan") if bad else 1.25, device=gpu)
        opt_m.clip_and_step(mine, 0.5, guard=guard)
        assert int(opt_m.last_skipped.item()) == int(bad)
        if bad:
            for p, b in zip(mine, before):
                assert torch.equal(p.detach(), b)
            continue
        for q, g in zip(ref, gs):
            q.grad = g.clone().to(gpu)
        torch.nn.utils.clip_grad_norm_(ref, 0.5)
        opt_r.step()
    for k, (p, q) in enumerate(zip(mine, ref)):
        assert torch.isfinite(p).all()
        assert (p - q).abs().max().item() <= 2e-6 * q.abs().max().item(), k
    sd = opt_m.state_dict()                                   # reconciles the host-side step entries
    for p, q in zip(mine[:-1], ref[:-1]):
        sm, sr = opt_m.state[p], opt_r.state[q]
        assert float(sm["step"]) == float(sr["step"]) == 5.0
        assert (sm["exp_avg"] - sr["exp_avg"]).abs().max().item() <= 2e-6 * sr["exp_avg"].abs().max().item()
        assert (sm["exp_avg_sq"] - sr["exp_avg_sq"]).abs().max().item() <= 2e-6 * sr["exp_avg_sq"].abs().max().item()
    assert all(float(s["step"]) == 5.0 for s in sd["state"].values())


def test_guarded_and_plain_steps_interleave(gpu):
    """A plain step after guarded ones starts from the reconciled count, and a guarded step after a
    plain one re-seeds the device counter: the mixture equals six plain torch steps."""
    from pl_convlstm_gan_amd.optim import FusedAdam
    mine, ref = _make(gpu, 6), _make(gpu, 6)
    opt_m = FusedAdam(mine, lr=1e-3)
    opt_r = torch.optim.Adam(ref, lr=1e-3)
    ok = torch.tensor(0.5, device=gpu)
    for step in range(6):
        gs = _grads(step, 1.0)
        for p, q, g in zip(mine, ref, gs):
            p.grad = g.clone().to(gpu)
            q.grad = g.clone().to(gpu)
        torch.nn.utils.clip_grad_norm_(ref, 0.5)
        opt_r.step()
        opt_m.clip_and_step(mine, 0.5, guard=ok if step in (0, 1, 4, 5) else None)
    for k, (p, q) in enumerate(zip(mine, ref)):
        assert (p - q).abs().max().item() <= 2e-6 * q.abs().max().item(), k
    opt_m.sync_steps()
    assert all(float(opt_m.state[p]["step"]) == 6.0 for p in mine)


def test_train_step_device_nan_check_equals_sync(gpu):
    """training.train_step with nan_check='device' (no host sync inside the step) against the
    default 'sync' mode: bit-identical parameters after three steps, one of them on a batch whose
    gauge values make the loss NaN-free, and a skipped flag that is a GPU tensor."""
    from pl_convlstm_gan_amd import Generator
    from pl_convlstm_gan_amd.losses import CombinedLoss
    from pl_convlstm_gan_amd.optim import FusedAdam
    from pl_convlstm_gan_amd.training import train_step

    def run(mode):
        torch.manual_seed(11)
        m = Generator(hidden_dims=[16, 32], lu_channels=5, scale_factor=1).to(gpu)
        opt = FusedAdam(m.parameters(), lr=5e-4)
        g = torch.Generator().manual_seed(12)
        flags = []
        for step in range(3):
            rain = (10 * torch.rand(2, 3, 1, 16, 32, generator=g)).to(gpu)
            if step == 1:
                rain[0, 0, 0, 0, 0] = float("nan")            # poisons the output, hence the loss
            dem = torch.rand(2, 1, 16, 32, generator=g).to(gpu)
            lu = torch.rand(2, 5, 16, 32, generator=g).to(gpu)
            coords = torch.randint(0, 16, (6, 2), generator=g).to(gpu)
            values = (20 * torch.rand(2, 3, 6, generator=g)).to(gpu)
            _, _, _, skipped = train_step(m, CombinedLoss(), opt, rain, dem, lu, coords, values, nan_check=mode)
            flags.append(skipped)
        torch.cuda.synchronize()
        return [p.detach().clone() for p in m.parameters()], flags, opt

    p_sync, f_sync, _ = run("sync")
    p_dev, f_dev, opt = run("device")
    assert f_sync == [False, True, False]
    assert all(isinstance(f, torch.Tensor) and f.is_cuda for f in f_dev)
    assert [bool(f) for f in f_dev] == [False, True, False]
    for a, b in zip(p_sync, p_dev):
        assert torch.isfinite(b).all() and torch.equal(a, b)
    opt.sync_steps()
    assert {float(s["step"]) for s in opt.state.values()} == {2.0}


def _lazy_block_run(gpu, mode, plan):
    """A generator at scale 2 with the optimizer built BEFORE the first forward, as the reference's Trainer
    does (src/training/trainer.py:153-158): the lazily created upsample block is outside the optimizer, nobody
    zeroes its gradient, it accumulates and counts in every clip norm."""
    from pl_convlstm_gan_amd import Generator
    from pl_convlstm_gan_amd.losses import CombinedLoss
    from pl_convlstm_gan_amd.optim import FusedAdam
    from pl_convlstm_gan_amd.training import train_step
    torch.manual_seed(21)
    m = Generator(hidden_dims=[16, 32], lu_channels=5, scale_factor=2).to(gpu)
    opt = FusedAdam(m.parameters(), lr=5e-4)
    assert m.upsample_blocks is None
    g = torch.Generator().manual_seed(22)
    flags, lazy_grads = [], []
    for bad in plan:
        rain = (10 * torch.rand(2, 3, 1, 12, 20, generator=g)).to(gpu)
        dem = torch.rand(2, 1, 24, 40, generator=g).to(gpu)
        lu = torch.rand(2, 5, 24, 40, generator=g).to(gpu)
        coords = torch.randint(0, 12, (6, 2), generator=g).to(gpu)
        values = (20 * torch.rand(2, 3, 6, generator=g)).to(gpu)
        if bad:
            values = values - 100.0          # log1p(< -1) -> NaN loss, and NaN gradients from its backward pass
        _, _, _, skipped = train_step(m, CombinedLoss(), opt, rain, dem, lu, coords, values, max_norm=0.5, nan_check=mode)
        flags.append(bool(skipped))
        lazy_grads.append([None if p.grad is None else p.grad.detach().clone() for p in m.upsample_blocks.parameters()])
    torch.cuda.synchronize()
    owned = {id(p) for grp in opt.param_groups for p in grp["params"]}
    assert all(id(p) not in owned for p in m.upsample_blocks.parameters())
    return [p.detach().clone() for p in m.parameters()], flags, lazy_grads


@pytest.mark.parametrize("plan", [[False, True, False, False], [True, False, True, False]])
def test_device_nan_skip_with_the_lazy_block_outside_the_optimizer(gpu, plan):
    """The reference leaves a NaN batch BEFORE loss.backward() (trainer.py:306-310).  With the skip decided on
    the device the backward pass of that batch has run; its NaN gradients must not stay in the accumulating
    gradient of the block nobody zeroes, or every later clip norm is NaN.  Parameters after good / NaN / good /
    good batches are bit-equal to the host-checked mode, and so is the block's accumulated gradient."""
    p_sync, f_sync, g_sync = _lazy_block_run(gpu, "sync", plan)
    p_dev, f_dev, g_dev = _lazy_block_run(gpu, "device", plan)
    assert f_sync == plan == f_dev
    for a, b in zip(p_sync, p_dev):
        assert torch.isfinite(b).all() and torch.equal(a, b)
    for step, (gs, gd) in enumerate(zip(g_sync, g_dev)):
        for a, b in zip(gs, gd):
            if a is None:               # (NaN batch first: the host-checked mode has made no gradient yet)
                assert b is None or not b.any()
            else:
                assert torch.isfinite(b).all() and torch.equal(a, b), step


def test_non_finite_clip_norm_propagates_like_clip_grad_norm(gpu):
    """torch.nn.utils.clip_grad_norm_ multiplies by clamp(max_norm / (norm + 1e-6), max=1): a NaN norm gives a
    NaN factor and NaN gradients (and parameters after the step).  The fused step does the same instead of
    silently switching the clip off."""
    from pl_convlstm_gan_amd.optim import FusedAdam
    mine, ref = _make(gpu, 7), _make(gpu, 7)
    opt_m, opt_r = FusedAdam(mine, lr=5e-4), torch.optim.Adam(ref, lr=5e-4)
    for p, q, g in zip(mine, ref, _grads(0, 1.0)):
        p.grad, q.grad = g.clone().to(gpu), g.clone().to(gpu)
    mine[2].grad.view(-1)[5] = float("nan")
    ref[2].grad.view(-1)[5] = float("nan")
    torch.nn.utils.clip_grad_norm_(ref, 0.5)
    opt_r.step()
    norm = opt_m.clip_and_step(mine, 0.5)
    assert torch.isnan(norm).all()
    for p, q in zip(mine, ref):
        assert torch.isnan(q.grad).all() and torch.isnan(p.grad).all()
        assert torch.isnan(q).all() and torch.isnan(p).all()
