"""-m gpu: the twice-differentiable fused layers of the path-length pass (textboxgan_amd/ops2.py) against the float64 oracle
layers differentiated twice by torch autograd on the CPU (training_step.py:300-347: the loss is a function of a GRADIENT)."""
import math

import pytest
import torch

from oracle import ref_ops as R

from conftest import arith_modes

pytestmark = pytest.mark.gpu


def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64)


def l2_err(a, ref):
    a, ref = a.detach().double().cpu(), ref.detach().double().cpu()
    assert a.shape == ref.shape, (a.shape, ref.shape)
    return float((a - ref).norm() / (ref.norm() + 1e-30))


def _second_order(out, x, style, nimg, leaves):
    """a path-length-shaped objective: first-order term + squared norms of the gradients with respect to style and input"""
    g_style, g_x = torch.autograd.grad((out * nimg).sum(), (style, x), create_graph=True)
    loss = g_style.square().sum() + 0.3 * g_x.square().sum() + 0.1 * (out * out).sum()
    return (g_style, g_x), torch.autograd.grad(loss, leaves)


@arith_modes
@pytest.mark.parametrize("up", [False, True], ids=["conv_1", "conv_0_up"])
@pytest.mark.parametrize("shape", [(2, 16, 24, 6, 10), (3, 64, 32, 8, 32), (8, 128, 128, 16, 64)], ids=["small", "64ch", "128ch"])
def test_mod_layer2_first_and_second_order(dev, up, shape):
    """ops2.mod_layer2: output, the recorded gradient (d/dstyle, d/dx) and the gradient OF that gradient with respect to every
    input (x, w, modulation dense + bias, noise strength, bias, style)."""
    from textboxgan_amd import ops, ops2
    B, I, O, H, W = shape
    sd = 20
    x, style = rnd(B, I, H, W, seed=1), rnd(B, sd, seed=2)
    w, mw, mb = rnd(3, 3, I, O, seed=3), rnd(sd, I, seed=4), rnd(I, seed=5) * 0.1
    Ho, Wo = (2 * H, 2 * W) if up else (H, W)
    noise, strength, bias = rnd(B, 1, Ho, Wo, seed=6), torch.tensor(0.3, dtype=torch.float64), rnd(O, seed=7) * 0.2
    nimg = rnd(B, O, Ho, Wo, seed=8)
    leaves = [t.requires_grad_(True) for t in (x, w, mw, mb, strength, bias, style)]
    y = R.t_modulated_conv2d(x, style, w, mw, mb, up=up, demodulate=True, fused=False)
    out = R.t_bias_act(R.t_noise(y, noise, strength), bias, "lrelu")
    (gs_ref, gx_ref), grads_ref = _second_order(out, x, style, nimg, leaves)

    f = lambda t: t.detach().float().to(dev).contiguous()
    xd, wd, mwd, mbd, std, bd, styled = [f(t).requires_grad_(True) for t in (x, w, mw, mb, strength, bias, style)]
    coef_w, coef_m = 1.0 / math.sqrt(9 * I), 1.0 / math.sqrt(sd)
    s = torch.addmm(mbd + 1.0, styled, mwd * coef_m)
    wsq = wd.square().sum(dim=(0, 1)) * (coef_w ** 2)
    d = torch.rsqrt(s.square() @ wsq + 1e-8)
    if B == 3:  # strided views of wider tensors, as the batched affines of the synthesis pass hand them over (ops2.synthesis_styles)
        s, d = torch.cat([s, s], dim=1)[:, :I], torch.cat([d, d * 2.0], dim=1)[:, :O]
        assert not s.is_contiguous() and not d.is_contiguous()
    outd = ops2.mod_layer2(xd, wd, s, d, f(noise), std, bd, up=up)
    assert l2_err(outd, out) < 2e-5
    ops.FLAGS.no_filter_grads = True
    try:
        g_style, g_x = torch.autograd.grad((outd * f(nimg)).sum(), (styled, xd), create_graph=True)
    finally:
        ops.FLAGS.no_filter_grads = False
    assert l2_err(g_style, gs_ref) < 1e-4 and l2_err(g_x, gx_ref) < 1e-4
    loss = g_style.square().sum() + 0.3 * g_x.square().sum() + 0.1 * (outd * outd).sum()
    grads = torch.autograd.grad(loss, (xd, wd, mwd, mbd, std, bd, styled))
    for name, a, e in zip(("x", "w", "mod_w", "mod_b", "strength", "bias", "style"), grads, grads_ref):
        assert l2_err(a, e) < 3e-4, (name, l2_err(a, e))


@arith_modes
@pytest.mark.parametrize("shape", [(2, 16, 6, 10, True), (4, 128, 16, 64, True), (3, 40, 8, 32, False)],
                         ids=["small", "128ch", "no-skip"])
def test_torgb2_first_and_second_order(dev, shape):
    from textboxgan_amd import ops, ops2
    B, I, H, W, has_skip = shape
    sd = 20
    x, style = rnd(B, I, H, W, seed=11), rnd(B, sd, seed=12)
    w, mw, mb, bias = rnd(1, 1, I, 3, seed=13), rnd(sd, I, seed=14), rnd(I, seed=15) * 0.1, rnd(3, seed=16) * 0.2
    skip = rnd(B, 3, H, W, seed=17) if has_skip else None
    nimg = rnd(B, 3, H, W, seed=18)
    leaves = [t.requires_grad_(True) for t in (x, w, mw, mb, bias, style)]
    y = R.t_bias_act(R.t_modulated_conv2d(x, style, w, mw, mb, up=False, demodulate=False, fused=False), bias, "linear")
    out = y if skip is None else skip + y
    (gs_ref, gx_ref), grads_ref = _second_order(out, x, style, nimg, leaves)

    f = lambda t: t.detach().float().to(dev).contiguous()
    xd, wd, mwd, mbd, bd, styled = [f(t).requires_grad_(True) for t in (x, w, mw, mb, bias, style)]
    s = torch.addmm(mbd + 1.0, styled, mwd * (1.0 / math.sqrt(sd)))
    outd = ops2.torgb2(xd, wd, s, bd, None if skip is None else f(skip))
    assert l2_err(outd, out) < 2e-5
    ops.FLAGS.no_filter_grads = True
    try:
        g_style, g_x = torch.autograd.grad((outd * f(nimg)).sum(), (styled, xd), create_graph=True)
    finally:
        ops.FLAGS.no_filter_grads = False
    assert l2_err(g_style, gs_ref) < 1e-4 and l2_err(g_x, gx_ref) < 1e-4
    loss = g_style.square().sum() + 0.3 * g_x.square().sum() + 0.1 * (outd * outd).sum()
    grads = torch.autograd.grad(loss, (xd, wd, mwd, mbd, bd, styled))
    for name, a, e in zip(("x", "w", "mod_w", "mod_b", "bias", "style"), grads, grads_ref):
        assert l2_err(a, e) < 3e-4, (name, l2_err(a, e))


@pytest.mark.parametrize("shape", [(2, 3, 5, 7), (3, 8, 16, 64), (2, 4, 96, 128)], ids=["ragged", "small", "two-chunks"])
def test_second_order_elementwise_entries_c_abi(dev, shape):
    """tbg_axpby_planes_f32 / tbg_bias_act_bwd2_f32 called through the C ABI: values against their float64 definitions (tbg.h),
    scalar (HW % 4 != 0) and 16-byte paths, one and several partial-sum chunks; bad calls return negative codes, never launch."""
    import ctypes as C
    from textboxgan_amd import native as N
    L = N.lib()
    B, M, H, W = shape
    HW = H * W
    nch = L.tbg_bias_act_bwd_chunks(HW)
    f = lambda t: t.float().to(dev).contiguous()
    a, b_, c = rnd(B, M, H, W, seed=1), rnd(B, M, H, W, seed=2), rnd(B, M, H, W, seed=3)
    sa, sb = rnd(B, M, seed=4), rnd(B, M, seed=5)
    ad, bd, cd, sad, sbd = f(a), f(b_), f(c), f(sa), f(sb)
    y = torch.empty_like(ad)
    part = torch.empty(B, M, nch, device=dev)
    assert L.tbg_axpby_planes_f32(N.ptr(ad), N.ptr(sad), N.ptr(bd), N.ptr(sbd), N.ptr(cd), N.ptr(y), N.ptr(part), B * M, HW,
                                  N.stream()) == 0
    assert l2_err(y, sa[:, :, None, None] * a + sb[:, :, None, None] * b_) < 1e-6
    assert l2_err(part.sum(dim=2), (c * a).sum(dim=(2, 3))) < 1e-5
    y2 = torch.empty_like(ad)  # no scales, no second operand
    assert L.tbg_axpby_planes_f32(N.ptr(ad), None, None, None, None, N.ptr(y2), None, B * M, HW, N.stream()) == 0
    assert torch.equal(y2, ad)
    EINVAL = -1
    assert L.tbg_axpby_planes_f32(None, None, None, None, None, N.ptr(y), None, B * M, HW, N.stream()) == EINVAL
    assert L.tbg_axpby_planes_f32(N.ptr(ad), None, None, None, None, None, None, B * M, HW, N.stream()) == EINVAL       # nothing to write
    assert L.tbg_axpby_planes_f32(N.ptr(ad), None, None, None, None, N.ptr(y), N.ptr(part), B * M, HW, N.stream()) == EINVAL  # sums without c
    assert L.tbg_axpby_planes_f32(N.ptr(ad), None, None, N.ptr(sbd), None, N.ptr(y), None, B * M, HW, N.stream()) == EINVAL   # scale without b
    assert L.tbg_axpby_planes_f32(N.ptr(ad), None, None, None, None, N.ptr(y), None, 0, HW, N.stream()) == EINVAL

    # bias_act_bwd2: out = lrelu(d * yc + noise * strength + b) * sqrt2 built in float64
    yc, noise, bias = rnd(B, M, H, W, seed=6), rnd(B, 1, H, W, seed=7), rnd(M, seed=8) * 0.3
    d, gdd, strength = rnd(B, M, seed=9).abs() + 0.5, rnd(B, M, seed=10), torch.tensor(0.4, dtype=torch.float64)
    pre = d[:, :, None, None] * yc + noise * strength + bias[None, :, None, None]
    out = torch.where(pre > 0, pre, 0.2 * pre) * math.sqrt(2.0)
    m = torch.where(pre > 0, 1.0, 0.2).double() * math.sqrt(2.0)
    cc, dout = rnd(B, M, H, W, seed=11), rnd(B, M, H, W, seed=12)
    g_ref = m * (d[:, :, None, None] * cc + gdd[:, :, None, None] * yc)
    pc_ref = (dout * m * cc).sum(dim=(2, 3))
    keep = (f(d), f(bias), f(noise), f(strength))  # (the epilogue holds raw pointers)
    epi = N.epilogue(out_scale=keep[0], bias=keep[1], noise=keep[2], strength=keep[3], act=N.ACT_LRELU, slope=0.2)
    ccd, outd, doutd, gddd = f(cc), f(out), f(dout), f(gdd)
    g = torch.empty_like(ccd)
    assert L.tbg_bias_act_bwd2_f32(N.ptr(ccd), N.ptr(outd), N.ptr(doutd), N.ptr(gddd), N.ptr(g), N.ptr(part), B, M, HW,
                                   C.byref(epi), N.stream()) == 0
    assert l2_err(g, g_ref) < 2e-6 and l2_err(part.sum(dim=2), pc_ref) < 1e-5
    assert L.tbg_bias_act_bwd2_f32(None, N.ptr(outd), N.ptr(doutd), None, N.ptr(g), None, B, M, HW, C.byref(epi), N.stream()) == EINVAL
    assert L.tbg_bias_act_bwd2_f32(N.ptr(ccd), N.ptr(outd), None, None, N.ptr(g), N.ptr(part), B, M, HW, C.byref(epi),
                                   N.stream()) == EINVAL  # sums need dout
    assert L.tbg_bias_act_bwd2_f32(N.ptr(ccd), N.ptr(outd), N.ptr(doutd), None, N.ptr(g), None, B, M, HW, None, N.stream()) == EINVAL
    res = N.epilogue(residual=ccd)
    assert L.tbg_bias_act_bwd2_f32(N.ptr(ccd), N.ptr(outd), N.ptr(doutd), None, N.ptr(g), None, B, M, HW, C.byref(res),
                                   N.stream()) == EINVAL  # a fused residual is not part of this form
