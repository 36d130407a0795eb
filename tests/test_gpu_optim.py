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
