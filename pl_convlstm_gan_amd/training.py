"""One optimisation step in the reference's order (src/training/trainer.py:290-315), with
batch data parallelism over the GPUs of a node.

    zero_grad -> forward -> CombinedLoss -> NaN skip -> backward -> [all-reduce] -> clip -> Adam

The reference is single-device.  Data parallelism is new here: one process per GPU, each with
its own batch shard; the only exchange is ONE all-reduce (RCCL over xGMI) of one flat fp32
gradient buffer (85 889 floats for hidden [16,32] at scale 1), issued after backward and before
``clip_grad_norm_`` so that the clip sees the global gradient; the NaN flag that makes every rank
skip the same batches travels as one more element of that buffer (device-side check) or as a
1-element all-reduce before the backward pass (host-side check).  At 0.34 MB the gradient message is
latency-bound, so it is sent as a single bucket with no overlap machinery.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class GradBucket:
    """Flat fp32 view of all parameter gradients, rebuilt lazily when the parameter set changes
    (the generator creates its upsample blocks inside the first forward)."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.sizes = [p.numel() for p in self.params]
        self.flat: Optional[torch.Tensor] = None

    def gather(self, extra: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``extra`` (1-D fp32) rides at the end of the buffer through the same all-reduce."""
        # a parameter without a gradient contributes zeros to the exchange; scatter() leaves it
        # without one if no rank had any, so the optimizer skips it as torch.optim would
        self.had_grad = [p.grad is not None for p in self.params]
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.params]
        parts = [g.reshape(-1) for g in grads]
        if extra is not None:
            parts.append(extra.reshape(-1).to(parts[0].dtype))
        self.flat = torch.cat(parts)
        return self.flat

    def scatter(self, keep_none: bool = True):
        """Point every ``p.grad`` at its slice of the flat buffer.  ``keep_none``: parameters that had
        no gradient at gather() time get ``None`` back (only valid when every rank agrees on which
        parameters those are, i.e. the same graph on every rank -- the case here)."""
        for p, had, chunk in zip(self.params, self.had_grad, self.flat[:sum(self.sizes)].split(self.sizes)):
            p.grad = chunk.view_as(p) if (had or not keep_none) else None

    def extra(self) -> torch.Tensor:
        return self.flat[sum(self.sizes):]


def _world(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def _dp_active(group=None) -> bool:
    """A process group exists: the gradient exchange runs (also for a group of one rank, where it is
    the identity -- that is how the RCCL path is exercised on a one-GPU box)."""
    return dist.is_available() and dist.is_initialized()


def _all_reduce(t: torch.Tensor, op, group=None) -> torch.Tensor:
    """dist.all_reduce in place.  With the gloo backend and a GPU tensor the reduction goes through
    a host copy: gloo is only the rehearsal backend (several ranks sharing one GPU on a one-GPU box),
    a rehearsal with gloo collectives on GPU tensors once hung without a diagnosis, and a host
    all-reduce has no GPU dependency by construction.  RCCL reduces the GPU tensor directly."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        host = t.detach().cpu()
        dist.all_reduce(host, op=op, group=group)
        t.copy_(host)
    else:
        dist.all_reduce(t, op=op, group=group)
    return t


def all_reduce_mean_(flat: torch.Tensor, group=None) -> torch.Tensor:
    w = _world(group)
    if w > 1:
        _all_reduce(flat, dist.ReduceOp.SUM, group)
        flat.mul_(1.0 / w)
    return flat


def all_reduce_weighted_mean_(flat: torch.Tensor, n_local: int, group=None) -> torch.Tensor:
    """Sample-weighted mean over ranks, in place: ``flat`` must carry one spare LAST element; every rank
    scales its buffer by its own sample count, the spare element carries that count through the same
    SUM all-reduce, and the sum is divided by the total (a device scalar: no host sync).  With equal
    shards this equals the plain mean; with unequal ones it is the per-sample mean of the global
    batch (a mean of per-rank means would overweight the short shard)."""
    if _world(group) > 1:
        flat.mul_(float(n_local))
        flat[-1] = float(n_local)
        _all_reduce(flat, dist.ReduceOp.SUM, group)
        total = flat[-1].clone()
        flat.div_(total)
    elif _dp_active(group):
        _all_reduce(flat, dist.ReduceOp.SUM, group)     # one rank: the sum is the buffer itself
    return flat


def any_rank_true(flag: torch.Tensor, group=None) -> bool:
    """Collective OR of a 0-dim bool/float tensor (host sync, as the reference's isnan check)."""
    f = flag.to(torch.float32).reshape(1)
    if _world(group) > 1:
        _all_reduce(f, dist.ReduceOp.MAX, group)
    return bool(f.item() > 0)


def exchange_grads_device(params, n_local: int, guard: Optional[torch.Tensor] = None, group=None) -> Optional[torch.Tensor]:
    """The step's one collective for GPU gradients, as two library launches around the all-reduce
    (``plc_dp_pack_grads`` / ``plc_dp_unpack_grads``): every rank's gradients, scaled by its sample count,
    travel in one flat buffer whose last two elements are this rank's NaN flag (``guard`` is NaN) and its
    sample count; afterwards every ``p.grad`` holds the per-sample mean over the global batch.  Returns the
    guard for the optimizer's guarded step -- NaN when the loss of ANY rank was NaN, so that all ranks skip
    the same batches -- or None when no guard was given.  Parameters without a gradient are left out (every
    rank runs the same graph, so they agree on which those are)."""
    import ctypes as C

    from . import _lib
    L = _lib.lib()
    ps = [p for p in params if p.grad is not None]
    if not ps:
        return guard
    dev = ps[0].grad.device
    table = (_lib.OptTensor * len(ps))()
    total = 0
    for k, p in enumerate(ps):
        g = p.grad
        if not g.is_cuda or g.dtype != torch.float32 or not g.is_contiguous():
            raise RuntimeError("exchange_grads_device: gradients must be contiguous fp32 GPU tensors")
        table[k].grad, table[k].numel = g.data_ptr(), g.numel()
        total += g.numel()
    flat = torch.empty(total + 2, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    g_in = None
    if guard is not None:
        g_in = guard.detach().reshape(-1)[:1].float().contiguous()
    # (a group of one rank: the weights cancel, and 1 keeps the gradients bit-identical to a step without a group)
    weight = float(n_local) if _world(group) > 1 else 1.0
    _lib.check(L.plc_dp_pack_grads(table, len(ps), flat.data_ptr(), total + 2, weight,
                                   g_in.data_ptr() if g_in is not None else None, stream), "plc_dp_pack_grads")
    _all_reduce(flat, dist.ReduceOp.SUM, group)
    g_out = torch.empty_like(g_in) if g_in is not None else None
    _lib.check(L.plc_dp_unpack_grads(table, len(ps), flat.data_ptr(), total + 2,
                                     g_in.data_ptr() if g_in is not None else None,
                                     g_out.data_ptr() if g_out is not None else None, stream), "plc_dp_unpack_grads")
    return g_out


def clip_flat_(flat: torch.Tensor, max_norm: float) -> torch.Tensor:
    """torch.nn.utils.clip_grad_norm_ on the flat buffer: scale by min(1, max_norm/(||g||+1e-6)).
    Returns the pre-clip norm (0-dim tensor, no host sync)."""
    norm = flat.norm(2)
    flat.mul_(torch.clamp(max_norm / (norm + 1e-6), max=1.0))
    return norm


def _stash_unowned_grads(optimizer, params):
    """{id(p): copy of p.grad} for the parameters in ``params`` that ``optimizer`` does not hold (their
    gradients accumulate over steps: optimizer.zero_grad() never sees them).  A parameter without a
    gradient yet gets zeros and an explicit zero gradient to accumulate into, which gives the bits of the
    first assignment.  The copies live in buffers kept on the optimizer object (same addresses every step,
    as a captured step needs); None when every parameter is owned."""
    owned = {id(p) for group in optimizer.param_groups for p in group["params"]}
    loose = [p for p in params if id(p) not in owned and p.requires_grad]
    if not loose:
        return None
    bufs = getattr(optimizer, "_unowned_grad_copies", None)
    if bufs is None:
        bufs = optimizer._unowned_grad_copies = {}
    for k in [k for k in bufs if k not in {id(p) for p in loose}]:
        del bufs[k]
    src, dst = [], []
    with torch.no_grad():
        for p in loose:
            b = bufs.get(id(p))
            if b is None or b.shape != p.shape or b.device != p.device:
                b = bufs[id(p)] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            if p.grad is None:
                p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
                b.zero_()
            else:
                src.append(p.grad)
                dst.append(b)
        if dst:
            torch._foreach_copy_(dst, src)
    return bufs


def train_step(model, loss_module, optimizer, rain_lr, dem, lu, s_coords, s_values, *, max_norm: float = 0.5,
               forward_kwargs: Optional[dict] = None, group=None, clip_params: Optional[Iterable] = None,
               nan_check: str = "sync"):
    """Returns (loss, loss_dict, fake_hr, skipped).  ``clip_params`` defaults to all model
    parameters (the reference clips over model.parameters(), lazily created blocks included).

    ``nan_check`` -- how the reference's ``if torch.isnan(loss): continue`` (trainer.py:306-308) is
    carried out:
      "sync"   the host reads the flag before launching the backward pass (``skipped`` is a bool);
      "device" (needs optim.FusedAdam) the backward pass and the optimizer step are launched at once
               and the optimizer's guarded step leaves parameters and state untouched when the loss
               (of any rank) is NaN; ``skipped`` is then a 0-dim bool GPU tensor, valid once the step
               has run -- no host synchronisation inside the step.  The reference skips a NaN batch BEFORE
               ``loss.backward()`` (trainer.py:306-310); here the backward pass of such a batch has run by the
               time the skip is decided, which only matters for parameters the optimizer does not own (the
               reference builds its optimizer before the generator's lazy upsample blocks exist,
               trainer.py:153-158): nobody zeroes their ``.grad``, it accumulates from step to step, and the
               NaN gradients of a skipped batch would stay in it for good.  Their gradients are therefore
               copied before the backward pass and the skipped step puts the copies back, so that after a
               skipped batch every tensor the next step reads is what the reference would hold.
    """
    if nan_check not in ("sync", "device"):
        raise ValueError(f"nan_check must be 'sync' or 'device', not {nan_check!r}")
    fused = hasattr(optimizer, "clip_and_step")     # optim.FusedAdam: clip + Adam in two launches
    on_device = nan_check == "device" and fused and rain_lr.is_cuda
    optimizer.zero_grad()
    fake_hr = model(rain_lr, dem, lu, **(forward_kwargs or {}))
    scale_factor = fake_hr.shape[-2] / rain_lr.shape[-2]
    loss, loss_dict = loss_module(fake_hr, rain_lr, s_coords, s_values, scale_factor)
    guard = None
    restore = None
    params = list(clip_params if clip_params is not None else model.parameters())
    if on_device:
        guard = loss.detach().reshape(1)
        restore = _stash_unowned_grads(optimizer, params)
    elif any_rank_true(torch.isnan(loss.detach()), group):
        return loss, loss_dict, fake_hr, True
    loss.backward()
    if _dp_active(group) and fused and rain_lr.is_cuda:
        # device mode: this rank's NaN flag rides in the gradient buffer, so the step has ONE collective
        guard = exchange_grads_device(params, int(rain_lr.shape[0]), guard, group)
    elif _dp_active(group) or not fused:
        bucket = GradBucket(params)
        flat = bucket.gather(torch.zeros(1, dtype=torch.float32, device=rain_lr.device) if _dp_active(group) else None)
        all_reduce_weighted_mean_(flat, int(rain_lr.shape[0]), group)
        if not fused:
            clip_flat_(flat[:sum(bucket.sizes)], max_norm)
        bucket.scatter()
    if fused:
        optimizer.clip_and_step(params, max_norm, guard=guard, restore=restore)
    else:
        optimizer.step()
    if on_device:
        skipped = optimizer.last_skipped.reshape(()).to(torch.bool) if optimizer.last_skipped is not None \
            else torch.isnan(guard).reshape(())
        return loss, loss_dict, fake_hr, skipped
    return loss, loss_dict, fake_hr, False


def params_in_a_live_autograd_graph(params) -> list:
    """The parameters among ``params`` whose gradient-accumulation node is kept alive by somebody else, i.e. that
    an existing autograd graph (a loss or an output that has not been detached or dropped) still points at.
    A parameter's accumulation node is cached on the parameter only weakly: it dies with the last graph that
    uses it, and a new one is made on the next use.  So a mark left in the node's metadata is found again only
    if some graph held the node in between."""
    def node(p):
        return p.expand_as(p).grad_fn.next_functions[0][0]

    params = [p for p in params if p.requires_grad]
    mark = object()
    for p in params:
        node(p).metadata["plc_live_probe"] = mark
    held = [p for p in params if node(p).metadata.get("plc_live_probe") is mark]
    for p in held:
        node(p).metadata.pop("plc_live_probe", None)
    return held


class GraphedTrainStep:
    """``train_step(..., nan_check="device")`` captured once into a HIP graph and replayed.

    For shapes where a step is launch-bound (the reference's own 15x12 basin grid: ~100 kernels of a few
    microseconds each) the replay removes the host work of a step -- descriptor building, ~100 launches,
    PyTorch's autograd bookkeeping -- and the gaps between the kernels.  Every library call is plain
    stream work without allocation or synchronisation, and the NaN-batch skip is decided on the device,
    so the whole step (zero_grad, forward, CombinedLoss, BPTT, clip, Adam) captures as it is.

    * one instance serves ONE batch shape; inputs are copied into the capture's static buffers;
    * the returned tensors (loss, loss parts, prediction, skipped flag) live in the graph's memory pool
      and are overwritten by the next replay: read or clone them before stepping again; they carry no
      autograd graph;
    * the learning rate is a launch argument, so a changed ``param_groups[0]['lr']`` (ReduceLROnPlateau)
      triggers a re-capture; the Adam step count lives on the device and advances with every replay;
    * building the instance runs ``warmup`` real steps on the example batch (lazy upsample blocks,
      workspaces, optimizer state are created outside the capture) and then restores parameters,
      optimizer state and the accumulated gradients of parameters the optimizer does not own, in place,
      so the first replay starts from the state the caller handed in;
    * single process only (a captured RCCL all-reduce has not been exercised here: data-parallel
      runs use the eager step);
    * no autograd graph of an earlier eager step over the same parameters may still be alive when the
      instance is built (a kept ``loss`` is enough: its gradient-accumulation nodes stay bound to the stream of
      that step and pull the legacy stream into the capture, which the HIP runtime answers with a crash at the
      end of the capture).  The constructor checks for it and raises a RuntimeError that says so: detach or drop
      such tensors first;
    * checked against the eager step bit for bit: uninterrupted runs of replays, replays after eager steps, and
      replays between which the host reads the prediction through indexing kernels and copies parameters and
      optimizer state to the host, as a Trainer with a station RMSE and a checkpoint does
      (tests/test_gpu_graph_step.py).  The library issues no ``hipMemsetAsync``: with its two memset nodes in the
      captured step (the zero bias rows of the data-gradient convs, the persistent kernel's barrier counter) some
      replays computed from memory the node should have cleared, and which replay did depended on what the host
      did in between.  On fixed inputs the first wrong tensor was the output of the first data-gradient conv
      (some channels NaN: its bias rows are that memset's target) while prediction, loss and the gradient formed
      before that conv were right.  Small kernels do the clearing now (plc_zero_words_kernel); a memset node
      alone between two kernels of a minimal graph stays ordered (devtools/graph_memset_probe.hip), so what the
      runtime does with it inside the larger captured step was not established.
    """

    def __init__(self, model, loss_module, optimizer, example_batch, *, max_norm: float = 0.5,
                 forward_kwargs: Optional[dict] = None, clip_params: Optional[Iterable] = None, warmup: int = 2):
        if not hasattr(optimizer, "clip_and_step"):
            raise TypeError("GraphedTrainStep needs optim.FusedAdam (the NaN-batch skip must run on the device)")
        if _world() > 1:
            raise RuntimeError("GraphedTrainStep is single-process: use train_step under data parallelism")
        held = params_in_a_live_autograd_graph(model.parameters())
        if held:
            raise RuntimeError(
                f"GraphedTrainStep: an autograd graph of an earlier step still references {len(held)} parameter(s) of the "
                "model (a kept loss or prediction that was not detached): its gradient-accumulation nodes are bound to "
                "that step's stream and would pull it into the capture.  Detach or delete those tensors first.")
        self.model, self.loss_module, self.optimizer = model, loss_module, optimizer
        self.max_norm, self.forward_kwargs = max_norm, forward_kwargs
        self.clip_params = list(clip_params) if clip_params is not None else None
        self.static = [t.detach().clone() for t in example_batch]
        if not all(t.is_cuda for t in self.static):
            raise RuntimeError("GraphedTrainStep: the batch must be on the GPU")
        params = [p.detach().clone() for p in model.parameters()]
        had_state = optimizer.snapshot()
        # Parameters the optimizer does not own (the reference never registers the lazily created upsample
        # blocks) are not touched by its zero_grad(): their .grad ACCUMULATES from step to step and counts in
        # the clip norm, exactly as in the reference.  What the warm-up adds to it must be taken back.
        had_grad = {id(p): p.grad.detach().clone() for p in model.parameters() if p.grad is not None}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        with torch.no_grad():
            for p, s in zip(model.parameters(), params):      # (the lazy blocks created by the warm-up keep their values)
                p.copy_(s)
        optimizer.restore_(had_state)
        owned = {id(p) for group in optimizer.param_groups for p in group["params"]}
        with torch.no_grad():
            for p in model.parameters():
                if id(p) in owned or p.grad is None:
                    continue
                if id(p) in had_grad:
                    p.grad.copy_(had_grad[id(p)])
                else:
                    p.grad.zero_()      # (accumulating into zeros gives the bits of the first assignment)
        self.graph = None
        self._capture()

    def _eager(self):
        return train_step(self.model, self.loss_module, self.optimizer, *self.static, max_norm=self.max_norm,
                          forward_kwargs=self.forward_kwargs, clip_params=self.clip_params, nan_check="device")

    def _capture(self):
        self.lr = float(self.optimizer.param_groups[0]["lr"])
        self.optimizer.sync_steps()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            out = self._eager()
        # Only the values leave the capture.  Its autograd graph must not: the gradient-accumulation nodes in it
        # are bound to the capture stream, stay cached on the parameters while anything references them, and a
        # later eager step (a shorter last batch) would then accumulate on that stream.  The buffers live in
        # the graph's memory pool whether or not Python still references them.
        loss, parts, pred, skipped = out
        self.out = (loss.detach(), {k: (v.detach() if torch.is_tensor(v) else v) for k, v in parts.items()},
                    pred.detach(), skipped)
        del out, loss, parts, pred
        # the capture queued nothing; undo the host-side step bookkeeping it did
        self.optimizer.uncount_captured_step()
        # whatever cached buffers of the model the captured launches point into stay alive with this object
        # (the model drops a cached workspace as soon as a call of another size comes along)
        self._keep = list(getattr(self.model, "_ws", {}).values()) + list(getattr(self.model, "_tabs", {}).values())

    def __call__(self, rain_lr, dem, lu, s_coords, s_values):
        for dst, src in zip(self.static, (rain_lr, dem, lu, s_coords, s_values)):
            if dst.shape != src.shape:
                raise ValueError(f"GraphedTrainStep was captured for shape {tuple(dst.shape)}, got {tuple(src.shape)}")
            dst.copy_(src, non_blocking=True)
        if float(self.optimizer.param_groups[0]["lr"]) != self.lr:
            self._capture()
        self.graph.replay()
        self.optimizer.note_replayed_step()
        return self.out
