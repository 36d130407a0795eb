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

    def _launch(self, entries, max_norm, hp, step, device):
        """entries: list of (param, update flag).  Returns the pre-clip norm (1-element tensor)."""
        L = _lib.lib()
        n = len(entries)
        table = (_lib.OptTensor * n)()
        for k, (p, upd) in enumerate(entries):
            self._check(p, "grad")
            e = table[k]
            e.grad, e.numel, e.update = p.grad.data_ptr(), p.numel(), int(upd)
            if upd:
                self._check(p, "param")
                st = self.state[p]
                e.param, e.exp_avg, e.exp_avg_sq = p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
        nbytes = L.plc_clip_adam_workspace_bytes(table, n)
        ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
        norm = torch.empty(1, dtype=torch.float32, device=device)
        rc = L.plc_clip_adam_step(table, n, float(max_norm), float(hp["lr"]), float(hp["betas"][0]),
                                  float(hp["betas"][1]), float(hp["eps"]), float(hp["weight_decay"]), int(step),
                                  norm.data_ptr(), ws.data_ptr(), nbytes, torch.cuda.current_stream(device).cuda_stream)
        _lib.check(rc, "plc_clip_adam_step")
        return norm

    @torch.no_grad()
    def clip_and_step(self, clip_params: Optional[Iterable[torch.nn.Parameter]] = None, max_norm: float = 0.0):
        """clip_grad_norm_(clip_params, max_norm) then Adam on this optimizer's parameters.
        ``clip_params`` may hold parameters the optimizer does not own (they are clipped, not
        stepped).  Returns the gradient norm before clipping as a 1-element GPU tensor, or None
        when nothing was clipped."""
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

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.clip_and_step(None, 0.0)
        return loss
