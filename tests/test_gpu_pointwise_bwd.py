"""plc_lstm_pointwise_bwd (the pointwise part of the ConvLSTM cell backward, one timestep) through
the C ABI against torch autograd of the reference cell's equations (src/models/convlstm.py:22-28):
i,f,o = sigmoid, g = tanh, c' = f*c + i*g, h' = o*tanh(c').

A floating-point kernel, so the torch fp32 operators are its reference.  The engine reads the saved
gates and the incoming dh as bf16 and writes dz as bf16: the reference is evaluated on the SAME
bf16-rounded gates and dh, which leaves the output rounding (2^-9 relative per element) and the
engine's fast tanh; bounds: dz <= 1e-2 * max|ref| per element and 5e-3 relative L2, dc (fp32 out)
<= 2e-3 * max|ref|.  plc_lstm_pointwise_bwd_bf16dc (dc carried as bf16, the rollout's form) gets the
same dc_next values rounded to bf16 and its dc output is allowed bf16's 2^-8 relative rounding."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def cb8(x, dtype):
    """NCHW -> [N][C/8][H][W][8] in `dtype` (bf16 returned as raw int16)."""
    N, C, H, W = x.shape
    y = x.view(N, C // 8, 8, H, W).permute(0, 1, 3, 4, 2).contiguous().to(dtype)
    return y.view(torch.int16) if dtype == torch.bfloat16 else y


def from_cb8(y, dtype):
    if dtype == torch.bfloat16:
        y = y.view(torch.bfloat16)
    N, ncb, H, W, _ = y.shape
    return y.float().permute(0, 1, 4, 2, 3).reshape(N, ncb * 8, H, W)


def bf(x):
    return x.to(torch.bfloat16).float()


@pytest.mark.parametrize("dc_bf16", [False, True])
@pytest.mark.parametrize("B,Ch,H,W,n_dh,with_dc,null_c", [(2, 16, 9, 13, 2, True, False), (1, 32, 16, 40, 3, True, False),
                                                          (3, 32, 5, 7, 1, False, False), (2, 16, 8, 8, 0, True, False),
                                                          (2, 32, 6, 10, 2, True, True)])
def test_pointwise_bwd_matches_autograd(gpu, B, Ch, H, W, n_dh, with_dc, null_c, dc_bf16):
    """null_c: c_prev passed as NULL stands for the zero initial state (first timestep)."""
    from pl_convlstm_gan_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(B * 100 + Ch + n_dh)
    z = torch.randn(B, 4 * Ch, H, W, generator=g)
    c_prev = torch.randn(B, Ch, H, W, generator=g)
    if null_c:
        c_prev = torch.zeros_like(c_prev)
    dhs = [bf(torch.randn(B, Ch, H, W, generator=g)) for _ in range(n_dh)]
    dc_next = torch.randn(B, Ch, H, W, generator=g) if with_dc else None
    if dc_bf16 and with_dc:
        dc_next = bf(dc_next)
    zi, zf, zo, zg = z.split(Ch, dim=1)
    gates = bf(torch.cat([torch.sigmoid(zi), torch.sigmoid(zf), torch.sigmoid(zo), torch.tanh(zg)], dim=1))

    # reference: autograd through the cell equations with the (rounded) gate values as leaves,
    # then the activation derivatives expressed in the gate values, as the engine has them
    gl = gates.clone().requires_grad_(True)
    cp = c_prev.clone().requires_grad_(True)
    i_, f_, o_, g_ = gl.split(Ch, dim=1)
    c_new = f_ * cp + i_ * g_
    h_new = o_ * torch.tanh(c_new)
    dh_total = sum(dhs) if dhs else torch.zeros(B, Ch, H, W)
    loss = (h_new * dh_total).sum() + ((c_new * dc_next).sum() if with_dc else 0.0)
    loss.backward()
    di, df, do, dg = gl.grad.split(Ch, dim=1)
    gi, gf, go, gg = gates.split(Ch, dim=1)
    dz_ref = torch.cat([di * gi * (1 - gi), df * gf * (1 - gf), do * go * (1 - go), dg * (1 - gg * gg)], dim=1)
    dc_ref = cp.grad

    # a dh view with a wider image stride: the third contribution lives inside a 2*Ch-channel tensor
    dev = gpu
    views = []
    for k, d in enumerate(dhs):
        if k == 2:
            wide = cb8(torch.cat([torch.randn(B, Ch, H, W, generator=g), d], dim=1), torch.bfloat16).to(dev)
            off = (Ch // 8) * H * W * 16
            views.append((wide, wide.data_ptr() + off, 2 * (Ch // 8) * H * W * 16))
        else:
            t = cb8(d, torch.bfloat16).to(dev)
            views.append((t, t.data_ptr(), (Ch // 8) * H * W * 16))
    while len(views) < 3:
        views.append((None, None, 0))
    gates8 = cb8(gates, torch.bfloat16).to(dev)
    cp8 = cb8(c_prev, torch.float32).to(dev)
    dc_dtype = torch.bfloat16 if dc_bf16 else torch.float32
    dc8 = cb8(dc_next, dc_dtype).to(dev) if with_dc else None
    dz8 = torch.empty(B, 4 * Ch // 8, H, W, 8, dtype=torch.int16, device=dev)
    dco8 = torch.empty(B, Ch // 8, H, W, 8, dtype=torch.int16 if dc_bf16 else torch.float32, device=dev)
    entry = L.plc_lstm_pointwise_bwd_bf16dc if dc_bf16 else L.plc_lstm_pointwise_bwd
    rc = entry(views[0][1], views[0][2], views[1][1], views[1][2], views[2][1], views[2][2],
               dc8.data_ptr() if with_dc else None, gates8.data_ptr(), None if null_c else cp8.data_ptr(),
               dz8.data_ptr(), dco8.data_ptr(), B, Ch, H, W, torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "plc_lstm_pointwise_bwd")
    torch.cuda.synchronize()
    dz = from_cb8(dz8.cpu(), torch.bfloat16)
    dc = from_cb8(dco8.cpu(), dc_dtype)
    sz, sc = float(dz_ref.abs().max()) + 1e-20, float(dc_ref.abs().max()) + 1e-20
    assert float((dz - dz_ref).abs().max()) <= 1e-2 * sz
    assert float((dz - dz_ref).norm()) <= 5e-3 * float(dz_ref.norm()) + 1e-12
    assert float((dc - dc_ref).abs().max()) <= (6e-3 if dc_bf16 else 2e-3) * sc


def test_pointwise_bwd_rejects_bad_arguments(gpu):
    from pl_convlstm_gan_amd import _lib
    L = _lib.lib()
    t = torch.zeros(64, device=gpu)
    p = t.data_ptr()
    assert L.plc_lstm_pointwise_bwd(None, 0, None, 0, None, 0, None, None, p, p, p, 1, 16, 2, 2, None) != 0
    assert L.plc_lstm_pointwise_bwd(None, 0, None, 0, None, 0, None, p, p, p, p, 1, 12, 2, 2, None) != 0
    assert L.plc_lstm_pointwise_bwd(None, 0, None, 0, None, 0, None, p, p, p, p, 0, 16, 2, 2, None) != 0
