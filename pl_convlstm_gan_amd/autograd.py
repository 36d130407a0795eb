"""Autograd glue: makes the HIP forward/backward pair differentiable for PyTorch-ROCm callers
(``loss.backward()`` in the reference's training step, src/training/trainer.py:304-310)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib, ops


class _CellFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, h, c, weight, bias):
        h2, c2 = ops.convlstm_cell_forward(x, h, c, weight, bias)
        ctx.save_for_backward(x, h, c, weight, bias)
        return h2, c2

    @staticmethod
    def backward(ctx, dh2, dc2):
        x, h, c, weight, bias = ctx.saved_tensors
        dx, dh, dc, dw, db = ops.convlstm_cell_backward(x, h, c, weight, bias, dh2.contiguous(), dc2.contiguous())
        return dx, dh, dc, dw, db


def cell_with_grad(x, h, c, weight, bias):
    return _CellFn.apply(x, h, c, weight, bias)


def _param_list(model):
    """Parameters in the order of the gradient descriptor."""
    ps = [model.init_conv.weight, model.init_conv.bias,
          model.cell1.conv.weight, model.cell1.conv.bias,
          model.cell2.conv.weight, model.cell2.conv.bias,
          model.dem_attn.conv[0].weight, model.dem_attn.conv[0].bias,
          model.dem_attn.conv[2].weight, model.dem_attn.conv[2].bias,
          model.lu_attn.conv[0].weight, model.lu_attn.conv[0].bias,
          model.lu_attn.conv[2].weight, model.lu_attn.conv[2].bias]
    for blk in model.upsample_blocks:
        ps += [blk.conv.weight, blk.conv.bias]
    ps += [model.post_process[0].weight, model.post_process[0].bias,
           model.post_process[2].weight, model.post_process[2].bias]
    return ps


class _RolloutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, rain_lr, dem, lu, input_grid_size, *params):
        out, desc, ws = model._forward_impl(rain_lr, dem, lu, input_grid_size, True, params)
        ctx.model = model
        ctx.desc = desc
        ctx.ws = ws                      # saved activations live in the forward workspace (+ the bound parameters)
        ctx.keep = (rain_lr, dem, lu)    # the descriptor holds raw pointers into these
        ctx.n_blocks = len(model.upsample_blocks)
        ctx.shapes = [p.shape for p in params]
        return out

    @staticmethod
    def backward(ctx, grad_out):
        L = _lib.lib()
        d = ctx.desc
        grad_out = grad_out.contiguous()
        dev = grad_out.device
        grads = [torch.empty(s, dtype=torch.float32, device=dev) for s in ctx.shapes]
        g = _lib.RolloutGrads()
        g.grad_out = grad_out.data_ptr()
        it = iter(grads)
        nxt = lambda: next(it).data_ptr()  # noqa: E731
        g.init_w, g.init_b = nxt(), nxt()
        g.cell1_w, g.cell1_b = nxt(), nxt()
        g.cell2_w, g.cell2_b = nxt(), nxt()
        g.dem0_w, g.dem0_b = nxt(), nxt()
        g.dem2_w, g.dem2_b = nxt(), nxt()
        g.lu0_w, g.lu0_b = nxt(), nxt()
        g.lu2_w, g.lu2_b = nxt(), nxt()
        for k in range(ctx.n_blocks):
            g.up_w[k], g.up_b[k] = nxt(), nxt()
        g.post0_w, g.post0_b = nxt(), nxt()
        g.post2_w, g.post2_b = nxt(), nxt()
        nbytes = L.plc_generator_bwd_workspace_bytes(C.byref(d))
        if nbytes == 0:
            msg = L.plc_last_error().decode("utf-8", "replace")
            raise NotImplementedError(f"plc_generator_rollout_bwd: {msg}")
        wsb = ctx.model._bwd_workspace(nbytes, dev)
        g.workspace, g.workspace_bytes = wsb.data_ptr(), nbytes
        # gradients of the inputs only where autograd asks for them (rain_lr, dem, lu = arguments 1..3)
        d_in = [torch.empty_like(t, dtype=torch.float32) if ctx.needs_input_grad[1 + k] else None
                for k, t in enumerate(ctx.keep)]
        g.d_rain, g.d_dem, g.d_lu = (t.data_ptr() if t is not None else None for t in d_in)
        rc = L.plc_generator_rollout_bwd(C.byref(d), C.byref(g), ops._stream())
        _lib.check(rc, "plc_generator_rollout_bwd")
        return (None, d_in[0], d_in[1], d_in[2], None) + tuple(grads)


def rollout_with_grad(model, rain_lr, dem, lu, input_grid_size, params):
    """``params``: _param_list(model) on the inputs' device (the lazy blocks exist by now)."""
    return _RolloutFn.apply(model, rain_lr, dem, lu, input_grid_size, *params)
