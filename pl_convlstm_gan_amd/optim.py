"""Adam with the gradient clip folded in, on the HIP engine.

``FusedAdam`` takes torch.optim.Adam's constructor arguments, keeps torch.optim.Adam's per-parameter
state (``step``, ``exp_avg``, ``exp_avg_sq``) and ``state_dict`` layout -- checkpoints written by
either load into the other -- and applies the same update rule, but ``clip_and_step`` runs
``clip_grad_norm_`` + ``step`` (src/training/trainer.py:311-313 of the reference) as two launches
of libplconvlstm (``plc_clip_adam_step``) instead of ~25 small ones.  GPU parameters only; there
is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Optional

import torch

from . import _lib


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False):
        if amsgrad:
            raise NotImplementedError("FusedAdam: amsgrad is not implemented")
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("FusedAdam: invalid hyper-parameter")
        # the same keys torch.optim.Adam keeps in its param_groups, so state_dicts interchange
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False,
                        foreach=None, capturable=False, differentiable=False, fused=None,
                        decoupled_weight_decay=False)
        super().__init__(params, defaults)
        # guarded steps (clip_and_step(..., guard=loss)): the update count lives on the device
        self._dev_step: Optional[torch.Tensor] = None     # int64[1], applied updates of the tracked parameters
        self._dev_params = []                             # the parameters that counter speaks for
        self._unsynced = False                            # their host-side `step` entries may be ahead of it
        self.last_skipped: Optional[torch.Tensor] = None  # int32[1] on the GPU: outcome of the last guarded step

    def sync_steps(self):
        """Bring the host-side ``step`` entries in line with the device counter of the guarded
        steps (one host synchronisation; a no-op when nothing is pending)."""
        if self._unsynced and self._dev_step is not None:
            v = float(self._dev_step.item())
            for p in self._dev_params:
                self.state[p]["step"].fill_(v)
        self._unsynced = False

    def state_dict(self):
        self.sync_steps()
        return super().state_dict()

    # ---- support for a captured (HIP graph) step: the state tensors must keep their addresses -----------
    def snapshot(self):
        """Copies of every state tensor, keyed by parameter identity."""
        self.sync_steps()
        return {id(p): {k: v.clone() for k, v in st.items() if torch.is_tensor(v)} for p, st in self.state.items()}

    @torch.no_grad()
    def restore_(self, snap):
        """Put the state back IN PLACE (state created since the snapshot is zeroed): a captured graph keeps
        pointing at these tensors."""
        self.sync_steps()
        for p, st in self.state.items():
            old = snap.get(id(p), {})
            for k, v in st.items():
                if torch.is_tensor(v):
                    if k in old:
                        v.copy_(old[k])
                    else:
                        v.zero_()
        if self._dev_step is not None and self._dev_params:
            self._dev_step.fill_(int(float(self.state[self._dev_params[0]]["step"])))
        self._unsynced = False

    def uncount_captured_step(self):
        """A guarded step issued under graph capture advanced the host-side ``step`` entries but ran
        nothing: take the advance back (the device counter is what the replays move)."""
        for p in self._dev_params:
            self.state[p]["step"] -= 1

    def note_replayed_step(self):
        """A replay advanced the device counter: the host-side entries are behind until sync_steps()."""
        self._unsynced = True

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._dev_step, self._dev_params, self._unsynced = None, [], False

    def _state_of(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.tensor(0.0, dtype=torch.float32)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @staticmethod
    def _check(p, what):
        t = p if what == "param" else p.grad
        if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError(f"FusedAdam: {what} must be a contiguous fp32 GPU tensor (there is no CPU path)")

    def _launch(self, entries, max_norm, hp, step, device, guard=None, restore=None):
        """entries: list of (param, update flag).  Returns the pre-clip norm (1-element tensor).
        With ``guard`` (a 1-element fp32 GPU tensor) the step is the guarded one: ``step`` is ignored,
        the device counter ``self._dev_step`` is used, and ``self.last_skipped`` (int32[1], GPU)
        tells afterwards whether the update was skipped; ``restore`` maps id(param) to the tensor its
        gradient is set back to in that case."""
        L = _lib.lib()
        n = len(entries)
        table = (_lib.OptTensor * n)()
        for k, (p, upd) in enumerate(entries):
            self._check(p, "grad")
            e = table[k]
            e.grad, e.numel, e.update = p.grad.data_ptr(), p.numel(), int(upd)
            if restore and id(p) in restore:
                r = restore[id(p)]
                if r.shape != p.grad.shape or r.dtype != torch.float32 or not r.is_cuda or not r.is_contiguous():
                    raise RuntimeError("FusedAdam: a restore tensor must match its gradient (contiguous fp32, GPU)")
                e.restore = r.data_ptr()
            if upd:
                self._check(p, "param")
                st = self.state[p]
                e.param, e.exp_avg, e.exp_avg_sq = p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
        nbytes = L.plc_clip_adam_workspace_bytes(table, n)
        ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
        norm = torch.empty(1, dtype=torch.float32, device=device)
        stream = torch.cuda.current_stream(device).cuda_stream
        if guard is None:
            rc = L.plc_clip_adam_step(table, n, float(max_norm), float(hp["lr"]), float(hp["betas"][0]),
                                      float(hp["betas"][1]), float(hp["eps"]), float(hp["weight_decay"]), int(step),
                                      norm.data_ptr(), ws.data_ptr(), nbytes, stream)
            _lib.check(rc, "plc_clip_adam_step")
        else:
            self.last_skipped = torch.empty(1, dtype=torch.int32, device=device)
            rc = L.plc_clip_adam_step_guarded(table, n, float(max_norm), float(hp["lr"]), float(hp["betas"][0]),
                                              float(hp["betas"][1]), float(hp["eps"]), float(hp["weight_decay"]),
                                              guard.data_ptr(), self._dev_step.data_ptr(), self.last_skipped.data_ptr(),
                                              norm.data_ptr(), ws.data_ptr(), nbytes, stream)
            _lib.check(rc, "plc_clip_adam_step_guarded")
        return norm

    @torch.no_grad()
    def clip_and_step(self, clip_params: Optional[Iterable[torch.nn.Parameter]] = None, max_norm: float = 0.0,
                      guard: Optional[torch.Tensor] = None, restore: Optional[dict] = None):
        """clip_grad_norm_(clip_params, max_norm) then Adam on this optimizer's parameters.
        ``clip_params`` may hold parameters the optimizer does not own (they are clipped, not
        stepped).  Returns the gradient norm before clipping as a 1-element GPU tensor, or None
        when nothing was clipped.

        ``guard`` (a scalar GPU tensor, normally the loss): the reference's ``if torch.isnan(loss):
        continue`` decided on the device.  If it is NaN nothing changes (parameters, moments, step
        count); otherwise the step is the ordinary one.  No host synchronisation takes place;
        ``self.last_skipped`` (int32[1] on the GPU) holds the outcome, and the host-side ``step``
        entries are reconciled by ``sync_steps()`` / ``state_dict()``.

        ``restore`` ({id(param): tensor}, guarded steps only): gradients that a skipped step sets back to
        the given values -- for parameters whose ``.grad`` accumulates because no optimizer zeroes it (see
        ``training.train_step``); the gradients of all other parameters are left as the backward pass made them."""
        if guard is not None:
            done = self._guarded(clip_params, max_norm, guard, restore)
            if done is not NotImplemented:
                return done
            self.last_skipped = None
            if bool(torch.isnan(guard.detach()).any()):      # layouts the guarded launch does not cover: host check
                for p in (clip_params if clip_params is not None else []):
                    if restore and id(p) in restore and p.grad is not None:
                        p.grad.copy_(restore[id(p)])
                return None
        self.sync_steps()
        self._dev_step = None            # the host counters lead from here on; a guarded call re-seeds the device one
        clip = [p for p in (clip_params if clip_params is not None else []) if p.grad is not None] if max_norm > 0 else []
        clip_ids = {id(p) for p in clip}
        batches = []            # (group, step, [params with grad])
        for group in self.param_groups:
            if group.get("amsgrad") or group.get("maximize"):
                raise NotImplementedError("FusedAdam: amsgrad / maximize are not implemented")
            by_step = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self._state_of(p)
                st["step"] += 1
                by_step.setdefault(int(st["step"].item()), []).append(p)
            batches += [(group, s, ps) for s, ps in by_step.items()]
        if not batches and not clip:
            return None
        device = (batches[0][2][0] if batches else clip[0]).device
        stepped_ids = {id(p) for _, _, ps in batches for p in ps}
        single = len(batches) == 1 and (not clip or stepped_ids <= clip_ids)
        if single:
            group, s, ps = batches[0]
            entries = [(p, True) for p in ps] + [(p, False) for p in clip if id(p) not in stepped_ids]
            return self._launch(entries, max_norm if clip else 0.0, group, s, device)
        # general case (several groups / step counts, or stepped parameters outside the clip set):
        # clip first, then one un-clipped launch per batch
        norm = None
        if clip:
            norm = self._launch([(p, False) for p in clip], max_norm, self.param_groups[0], 1, device)
        for group, s, ps in batches:
            self._launch([(p, True) for p in ps], 0.0, group, s, device)
        return norm

    def _guarded(self, clip_params, max_norm, guard, restore=None):
        """The guarded single-launch case: one parameter group, every parameter with a gradient at
        the same step count, stepped parameters inside the clip set.  NotImplemented otherwise."""
        if len(self.param_groups) != 1:
            return NotImplemented
        group = self.param_groups[0]
        if group.get("amsgrad") or group.get("maximize"):
            raise NotImplementedError("FusedAdam: amsgrad / maximize are not implemented")
        ps = [p for p in group["params"] if p.grad is not None]
        if not ps or not guard.is_cuda:
            return NotImplemented
        clip = [p for p in (clip_params if clip_params is not None else []) if p.grad is not None] if max_norm > 0 else []
        stepped_ids = {id(p) for p in ps}
        if clip and not stepped_ids <= {id(p) for p in clip}:
            return NotImplemented
        states = [self._state_of(p) for p in ps]
        if self._dev_step is not None and [id(p) for p in ps] != [id(p) for p in self._dev_params]:
            self.sync_steps()            # the set of stepped parameters changed: start over from the host counters
            self._dev_step = None
        if self._dev_step is None:
            steps = {float(st["step"]) for st in states}
            if len(steps) != 1:
                return NotImplemented
            self._dev_step = torch.tensor([int(steps.pop())], dtype=torch.int64, device=ps[0].device)
            self._dev_params = ps
        for st in states:
            st["step"] += 1              # optimistic; sync_steps() replaces it by the device count
        self._unsynced = True
        g = guard.detach().reshape(-1)[:1]
        if g.dtype != torch.float32:
            g = g.float()
        entries = [(p, True) for p in ps] + [(p, False) for p in clip if id(p) not in stepped_ids]
        if restore and not {id(p) for p, _ in entries} >= set(restore):
            return NotImplemented        # a tensor to put back that this launch does not list (no clipping): host path
        return self._launch(entries, max_norm if clip else 0.0, group, 0, ps[0].device, guard=g.contiguous(),
                            restore=restore)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.clip_and_step(None, 0.0)
        return loss
