"""The training step captured into a HIP graph (training.GraphedTrainStep) against the eager step: the
same kernels replayed, so parameters, losses and the NaN-batch skip must agree bit for bit over several
steps on different batches -- including a batch whose loss is NaN (skipped on the device) and a change of
the learning rate between steps (re-capture)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _batch(seed, dev, nan=False, B=4, T=3, H=15, W=12, s=2, N=6):
    g = torch.Generator().manual_seed(seed)
    rain = 10 * torch.rand(B, T, 1, H, W, generator=g)
    dem = torch.rand(B, 1, H * s, W * s, generator=g)
    lu = torch.rand(B, 5, H * s, W * s, generator=g)
    coords = torch.randint(0, min(H, W), (N, 2), generator=g).unsqueeze(0).repeat(B, 1, 1)
    values = 20 * torch.rand(B, T, N, generator=g)
    if nan:
        values = values - 100.0          # log1p(< -1) -> NaN loss (reference behaviour)
    return [t.to(dev) for t in (rain, dem, lu, coords, values)]


def _setup(gpu, s=2, lazy_block_outside_optimizer=False, **geom):
    from pl_convlstm_gan_amd import Generator
    from pl_convlstm_gan_amd.losses import CombinedLoss
    from pl_convlstm_gan_amd.optim import FusedAdam
    torch.manual_seed(0)
    model = Generator(hidden_dims=[16, 32], lu_channels=5, scale_factor=s).to(gpu)
    opt = FusedAdam(model.parameters(), lr=5e-4) if lazy_block_outside_optimizer else None     # as the reference's Trainer does
    with torch.no_grad():
        model(*_batch(1, gpu, s=s, **geom)[:3])      # the lazy block exists before either run starts (same RNG draw)
    return model, CombinedLoss(), opt or FusedAdam(model.parameters(), lr=5e-4)


def test_graphed_step_equals_eager_step(gpu):
    from pl_convlstm_gan_amd.training import GraphedTrainStep, train_step
    plan = [(10, False, 5e-4), (20, True, 5e-4), (30, False, 5e-4), (40, False, 2e-4), (50, False, 2e-4)]

    model, loss_mod, opt = _setup(gpu)
    eager = []
    for seed, nan, lr in plan:
        opt.param_groups[0]["lr"] = lr
        loss, parts, _, skipped = train_step(model, loss_mod, opt, *_batch(seed, gpu, nan), max_norm=0.5, nan_check="device")
        eager.append((float(loss.detach()), bool(skipped), torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()))
    steps_eager = opt.state_dict()["state"][0]["step"]

    model, loss_mod, opt = _setup(gpu)
    start = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()
    step = GraphedTrainStep(model, loss_mod, opt, _batch(99, gpu), max_norm=0.5)
    assert torch.equal(torch.cat([p.detach().reshape(-1) for p in model.parameters()]), start)    # warm-up left no trace
    for (seed, nan, lr), (l_ref, s_ref, p_ref) in zip(plan, eager):
        opt.param_groups[0]["lr"] = lr
        loss, parts, pred, skipped = step(*_batch(seed, gpu, nan))
        got = float(loss.detach())
        assert (got != got and l_ref != l_ref) or got == l_ref
        assert bool(skipped) == s_ref
        assert torch.equal(torch.cat([p.detach().reshape(-1) for p in model.parameters()]), p_ref)
    assert [s for _, s, _ in eager] == [False, True, False, False, False]
    # the Adam step count lives on the device: 4 applied updates, as in the eager run
    assert float(opt.state_dict()["state"][0]["step"]) == float(steps_eager) == 4.0
    with pytest.raises(ValueError, match="captured for shape"):
        step(*_batch(60, gpu, B=2))


def test_graphed_step_with_auxiliary_streams(gpu):
    """The same at a shape (B=2, T=3, 256x256, scale 1) where both rollout calls fork onto the library's
    auxiliary streams -- two half-batch chains in both recurrences, leaf weight gradients and the attention
    branches on the side stream: the capture follows the event forks and joins, and the replay equals the
    eager step bit for bit, a skipped NaN batch included."""
    from pl_convlstm_gan_amd.training import GraphedTrainStep, train_step
    geom = dict(B=2, T=3, H=256, W=256, s=1)
    plan = [(11, False), (21, True), (31, False)]

    model, loss_mod, opt = _setup(gpu, **geom)
    eager = []
    for seed, nan in plan:
        loss, _, _, skipped = train_step(model, loss_mod, opt, *_batch(seed, gpu, nan, **geom), max_norm=0.5, nan_check="device")
        eager.append((float(loss.detach()), bool(skipped), torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()))

    model, loss_mod, opt = _setup(gpu, **geom)
    step = GraphedTrainStep(model, loss_mod, opt, _batch(99, gpu, **geom), max_norm=0.5)
    for (seed, nan), (l_ref, s_ref, p_ref) in zip(plan, eager):
        loss, _, _, skipped = step(*_batch(seed, gpu, nan, **geom))
        got = float(loss.detach())
        assert (got != got and l_ref != l_ref) or got == l_ref
        assert bool(skipped) == s_ref
        assert torch.equal(torch.cat([p.detach().reshape(-1) for p in model.parameters()]), p_ref)
    assert [s for _, s, _ in eager] == [False, True, False]


@pytest.mark.parametrize("eager_first", [0, 2])
def test_graphed_step_with_the_lazy_block_outside_the_optimizer(gpu, eager_first):
    """The reference builds its optimizer before the lazy upsample block exists: the block's gradients are never
    zeroed, accumulate from step to step and count in the clip norm.  The captured step must continue that
    accumulation exactly where the eager steps left it (or start it, when built before any step) -- the
    warm-up's contributions taken back."""
    from pl_convlstm_gan_amd.training import GraphedTrainStep, train_step
    seeds = [12, 22, 32, 42, 52]
    model, loss_mod, opt = _setup(gpu, lazy_block_outside_optimizer=True)
    assert len(opt.param_groups[0]["params"]) == 18 and len(list(model.parameters())) == 20
    ref = []
    for seed in seeds:
        train_step(model, loss_mod, opt, *_batch(seed, gpu), max_norm=0.5, nan_check="device")
        ref.append(torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone())
    assert not torch.equal(ref[0], ref[1])

    model, loss_mod, opt = _setup(gpu, lazy_block_outside_optimizer=True)
    got = []
    for seed in seeds[:eager_first]:
        out = train_step(model, loss_mod, opt, *_batch(seed, gpu), max_norm=0.5, nan_check="device")
        got.append(torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone())
        del out                              # no autograd graph of an eager step may outlive it
    step = GraphedTrainStep(model, loss_mod, opt, _batch(99, gpu), max_norm=0.5)
    for seed in seeds[eager_first:]:
        step(*_batch(seed, gpu))
        got.append(torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone())
    for a, b in zip(ref, got):
        assert torch.equal(a, b)


def test_replays_with_index_reads_and_host_copies_between_them(gpu):
    """What a Trainer does between steps -- reads the prediction at the rain gauges through indexing kernels, and
    once per epoch copies parameters and optimizer state to the host (a checkpoint) -- with the lazy block outside
    the optimizer, on the reference's 15x12 basin grid (the single-launch persistent recurrence): 14 replays equal
    14 eager steps bit for bit.  (With hipMemsetAsync nodes in the captured step this sequence ended in NaN
    parameters; the library clears with kernels now.)"""
    from pl_convlstm_gan_amd.training import GraphedTrainStep, train_step
    geom = dict(B=48, T=5, H=15, W=12, s=2, N=10)

    def run(graph):
        model, loss_mod, opt = _setup(gpu, lazy_block_outside_optimizer=True, **geom)
        step, losses, kept = None, [], None
        for k in range(14):
            b = _batch(100 + k, gpu, **geom)
            if graph and k > 0:
                if step is None:
                    step = GraphedTrainStep(model, loss_mod, opt, b, max_norm=0.5)
                loss, parts, fake, skipped = step(*b)
            else:
                loss, parts, fake, skipped = train_step(model, loss_mod, opt, *b, max_norm=0.5, nan_check="device")
            loss, fake = loss.detach(), fake.detach()
            assert not bool(skipped)
            losses.append(loss.item())
            coords = b[3][0]
            hr = ((coords.float() + 0.5) * 2.0 - 0.5).long()
            rows, cols = hr[:, 0], hr[:, 1]
            inside = (rows >= 0) & (rows < fake.shape[-2]) & (cols >= 0) & (cols < fake.shape[-1])
            at = fake[:, :, 0][:, :, rows[inside], cols[inside]]
            truth = b[4][:, :, inside]
            seen = ~torch.isnan(truth)
            rmse = torch.sqrt(((at[seen] - truth[seen]) ** 2).mean()).item()
            kept = (at, truth, inside, rmse)                     # alive across the next step, as the Trainer keeps them
            if k % 7 == 6:
                sd = opt.state_dict()
                host = [v.cpu() for st in sd["state"].values() for v in st.values() if torch.is_tensor(v)]
                host += [p.detach().cpu() for p in model.parameters()]
        return losses, torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()

    l_eager, p_eager = run(False)
    l_graph, p_graph = run(True)
    assert l_eager == l_graph
    assert torch.isfinite(p_graph).all() and torch.equal(p_eager, p_graph)


def test_many_replays_of_forward_loss_backward_on_fixed_inputs(gpu):
    """Forward + CombinedLoss + backward captured once and replayed 80 times on fixed inputs and parameters:
    prediction, loss and every parameter gradient of every replay equal the eager pass bit for bit."""
    from pl_convlstm_gan_amd.losses import CombinedLoss
    geom = dict(B=48, T=5, H=15, W=12, s=2, N=10)
    model, loss_mod, _ = _setup(gpu, **geom)
    rain, dem, lu, coords, values = _batch(7, gpu, **geom)

    def one():
        for p in model.parameters():
            p.grad = None
        out = model(rain, dem, lu)
        loss, _ = loss_mod(out, rain, coords, values, 2.0)
        loss.backward()
        return out.detach(), loss.detach()

    def snap(out, loss):
        return [p.grad.clone() for p in model.parameters()] + [out.clone(), loss.clone()]

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            o, l = one()
    torch.cuda.current_stream().wait_stream(side)
    ref = snap(o, l)
    del o, l
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        o, l = one()
    for rep in range(80):
        graph.replay()
        for k, (a, b) in enumerate(zip(ref, snap(o, l))):
            assert torch.equal(a, b), (rep, k)


def test_building_beside_a_live_autograd_graph_is_refused(gpu):
    """A loss kept from an eager step keeps that step's gradient-accumulation nodes (and their stream) alive;
    capturing then used to end in a crash inside the HIP runtime.  The constructor says what is wrong instead."""
    from pl_convlstm_gan_amd.training import GraphedTrainStep, train_step
    model, loss_mod, opt = _setup(gpu)
    kept = train_step(model, loss_mod, opt, *_batch(10, gpu), max_norm=0.5, nan_check="device")
    with pytest.raises(RuntimeError, match="autograd graph of an earlier step"):
        GraphedTrainStep(model, loss_mod, opt, _batch(99, gpu), max_norm=0.5)
    del kept
    GraphedTrainStep(model, loss_mod, opt, _batch(99, gpu), max_norm=0.5)(*_batch(11, gpu))
