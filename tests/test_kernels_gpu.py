"""-m gpu: every libtbg_hip kernel (through the C ABI) against the float64 CPU oracle."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_ops as R

from conftest import arith_modes

pytestmark = pytest.mark.gpu


def rel_err(a, ref):
    a = a.detach().double().cpu()
    ref = ref.detach().double().cpu()
    assert a.shape == ref.shape, (a.shape, ref.shape)
    return float((a - ref).abs().max() / (ref.abs().max() + 1e-30))


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64) * scale


# ---------------------------------------------------------------------------------------- upfirdn
UF_CASES = [
    # (B*C, H, W, up(x,y), down(x,y), pad(x0,x1,y0,y1), gain, note)
    (6, 16, 64, (1, 1), (1, 1), (1, 1, 1, 1), 4.0, "blur after up-conv"),
    (6, 17, 65, (1, 1), (1, 1), (1, 1, 1, 1), 4.0, "blur on 2H+1 x 2W+1"),
    (5, 32, 128, (1, 1), (1, 1), (2, 3, 2, 3), 1.0, "blur before 3x3 down"),
    (5, 32, 128, (1, 1), (1, 1), (1, 2, 1, 2), 1.0, "blur before 1x1 skip"),
    (3, 8, 32, (2, 2), (1, 1), (2, 1, 2, 1), 4.0, "rgb upsample"),
    (4, 16, 64, (1, 1), (2, 2), (1, 2, 1, 2), 1.0, "decimated skip FIR"),
    (4, 8, 16, (1, 1), (2, 1), (1, 2, 1, 2), 1.0, "decimated skip FIR, keep height"),
    (4, 8, 16, (2, 1), (1, 1), (2, 1, 2, 1), 1.0, "grad of keep-height decimation"),
    (2, 64, 256, (1, 1), (1, 1), (2, 3, 2, 3), 1.0, "full-size plane"),
    (3, 5, 7, (1, 1), (1, 1), (-1, 2, 3, -1), 1.0, "negative pads (crop)"),
    (3, 9, 13, (2, 2), (1, 1), (1, 2, 1, 2), 4.0, "up 2x2, odd pads (phase 1,1)"),
    (3, 9, 13, (2, 2), (1, 1), (3, 0, 2, 1), 4.0, "up 2x2, phase (1,0)"),
    (3, 9, 13, (2, 2), (1, 1), (2, 1, 3, 0), 4.0, "up 2x2, phase (0,1)"),
    (4, 8, 16, (2, 1), (1, 1), (1, 2, 2, 1), 1.0, "up 2x1, phase 1"),
    (2, 64, 256, (1, 1), (1, 1), (2, 2, 2, 2), 4.0, "grad of the up-conv blur: 65x257 out"),
    (3, 7, 9, (3, 1), (1, 2), (2, 3, 1, 2), 1.0, "factor 3 (direct kernel)"),
]


@pytest.mark.parametrize("case", UF_CASES, ids=[c[-1] for c in UF_CASES])
def test_upfirdn2d_matches_oracle(dev, case):
    from textboxgan_amd import ops
    major, H, W, up, down, pad, gain, _ = case
    x = rnd(1, major, H, W, seed=1)
    k = torch.from_numpy(R.setup_kernel([1, 3, 3, 1]).astype(np.float64) * gain)
    k[0, 1] += 0.05  # asymmetric: catches flip/transposition mistakes
    ref = R.t_upfirdn2d(x[0][..., None], k.numpy(), upx=up[0], upy=up[1], downx=down[0], downy=down[1],
                        padx0=pad[0], padx1=pad[1], pady0=pad[2], pady1=pad[3])[..., 0][None]
    y = ops.upfirdn2d_raw(x.float().to(dev), k.float().to(dev), up, down, pad)
    assert rel_err(y, ref) < 1e-5


@pytest.mark.parametrize("case", UF_CASES, ids=[c[-1] for c in UF_CASES])
def test_upfirdn2d_f16_matches_oracle(dev, case):
    """the op's second registered type (upfirdn_2d.cu:323-324): fp16 tensors, fp32 accumulation (.cu:101,195), one rounding
    of the result -- against the float64 oracle evaluated on the fp16-rounded operands: within half an fp16 ulp of it (+ the
    fp32 accumulation error), i.e. 1e-3 of max; minor > 1 (the op's NHWC "minor" axis) included."""
    import ctypes as C
    from textboxgan_amd import native as N
    major, H, W, up, down, pad, gain, _ = case
    for minor in (1, 3):
        x = rnd(major, H, W, minor, seed=1).half()
        k = torch.from_numpy(R.setup_kernel([1, 3, 3, 1]).astype(np.float64) * gain)
        k[0, 1] += 0.05
        k = k.half()
        ref = R.t_upfirdn2d(x.double(), k.double().numpy(), upx=up[0], upy=up[1], downx=down[0], downy=down[1],
                            padx0=pad[0], padx1=pad[1], pady0=pad[2], pady1=pad[3])
        xd, kd = x.to(dev).contiguous(), k.to(dev).contiguous()
        y = torch.empty(tuple(ref.shape), device=dev, dtype=torch.float16)
        N.check(N.lib().tbg_upfirdn2d_f16(N.ptr(xd), N.ptr(kd), N.ptr(y), major, H, W, minor, 4, 4, up[0], up[1], down[0], down[1],
                                          pad[0], pad[1], pad[2], pad[3], N.stream()), "tbg_upfirdn2d_f16")
        assert tuple(y.shape) == tuple(ref.shape)
        assert rel_err(y, ref) < 1e-3
    # argument checks as the fp32 entry (OP_REQUIRES list, upfirdn_2d.cu:241-256)
    assert N.lib().tbg_upfirdn2d_f16(N.ptr(xd), N.ptr(kd), N.ptr(y), major, H, W, 0, 4, 4, 1, 1, 1, 1, 0, 0, 0, 0, N.stream()) != 0
    assert N.lib().tbg_upfirdn2d_f16(N.ptr(xd), None, N.ptr(y), major, H, W, 1, 4, 4, 1, 1, 1, 1, 0, 0, 0, 0, N.stream()) != 0


@pytest.mark.parametrize("case", UF_CASES, ids=[c[-1] for c in UF_CASES])
@pytest.mark.parametrize("taps", [4, 3, 2], ids=["4tap", "3tap", "2tap"])
def test_upfirdn2d_separable_matches_oracle(dev, case, taps):
    """tbg_upfirdn2d_sep_f32 (1-D factors; the path the model's [1,3,3,1] filters take) == the oracle applied to
    outer(ky, kx), with the fused per-plane input scale and epilogue (demod scale, noise, bias, lrelu*sqrt2)."""
    import ctypes as C
    from textboxgan_amd import native as N
    major, H, W, up, down, pad, gain, _ = case
    kx = torch.tensor([1.0, 3.0, 3.5, 1.25][:taps], dtype=torch.float64) / 8.0   # asymmetric factors
    ky = torch.tensor([0.75, 2.5, 3.0, 1.0][:taps], dtype=torch.float64) / 8.0 * gain
    outW = (W * up[0] + pad[0] + pad[1] - taps + down[0]) // down[0]
    outH = (H * up[1] + pad[2] + pad[3] - taps + down[1]) // down[1]
    if outW < 1 or outH < 1:
        pytest.skip("empty output for this filter size")
    Bn, Mc = (2, major // 2) if major % 2 == 0 else (1, major)
    x = rnd(1, major, H, W, seed=1)
    isc = rnd(major, seed=2).abs() + 0.5
    osc = rnd(major, seed=3).abs() + 0.5
    bias = rnd(Mc, seed=4)
    noise = rnd(Bn, outH * outW, seed=5)
    strength = torch.tensor(0.3, dtype=torch.float64)
    k2 = torch.outer(ky, kx)
    ref = R.t_upfirdn2d((x[0] * isc[:, None, None])[..., None], k2.numpy(), upx=up[0], upy=up[1], downx=down[0],
                        downy=down[1], padx0=pad[0], padx1=pad[1], pady0=pad[2], pady1=pad[3])[..., 0]
    pre = ref * 0.7 * osc[:, None, None] + (noise.reshape(Bn, 1, outH, outW) * strength).expand(Bn, Mc, outH, outW).reshape(
        major, outH, outW) + bias.repeat(Bn)[:, None, None]
    exp = F.leaky_relu(pre, 0.2) * math.sqrt(2.0)
    f = lambda t: t.float().to(dev).contiguous()
    xd, kxd, kyd, iscd, oscd, bd, nd, sd = map(f, (x, kx, ky, isc, osc, bias, noise, strength))
    y = torch.empty((major, outH, outW), device=dev)
    epi = N.epilogue(out_scale=oscd, bias=bd, noise=nd, strength=sd, alpha=0.7, act=N.ACT_LRELU, slope=0.2)
    rc = N.lib().tbg_upfirdn2d_sep_f32(N.ptr(xd), N.ptr(kxd), N.ptr(kyd), N.ptr(y), major, H, W, taps, taps, up[0], up[1],
                                       down[0], down[1], pad[0], pad[1], pad[2], pad[3], N.ptr(iscd), Mc, C.byref(epi),
                                       N.stream())
    assert rc == 0
    assert rel_err(y, exp) < 2e-5
    # plain form (no scale / epilogue) as well
    y2 = torch.empty((major, outH, outW), device=dev)
    rc = N.lib().tbg_upfirdn2d_sep_f32(N.ptr(xd), N.ptr(kxd), N.ptr(kyd), N.ptr(y2), major, H, W, taps, taps, up[0], up[1],
                                       down[0], down[1], pad[0], pad[1], pad[2], pad[3], None, 1, None, N.stream())
    assert rc == 0
    ref0 = R.t_upfirdn2d(x[0][..., None], k2.numpy(), upx=up[0], upy=up[1], downx=down[0], downy=down[1], padx0=pad[0],
                         padx1=pad[1], pady0=pad[2], pady1=pad[3])[..., 0]
    assert rel_err(y2, ref0) < 1e-5


def test_upfirdn2d_hand_computed_vector(dev):
    """an answer worked out by hand (independent of the oracle): x = [[1,2],[3,4]], k = [[1,2],[3,4]] (applied flipped),
    up 2, pads (1,0): u = zero-insert -> [[1,0,2,0],[0,0,0,0],[3,0,4,0],[0,0,0,0]], left/top pad 1, true convolution with k:
    y[Y][X] = sum_{i,j} upad[Y+i][X+j] * k[1-i][1-j]."""
    from textboxgan_amd import native as N
    x = torch.tensor([[1.0, 2.0], [3.0, 4.0]])
    k = torch.tensor([[1.0, 2.0], [3.0, 4.0]])
    upad = torch.zeros(5, 5)
    upad[1::2, 1::2][:2, :2] = x
    exp = torch.zeros(4, 4)
    for Y in range(4):
        for X in range(4):
            exp[Y, X] = sum(upad[Y + i, X + j] * k[1 - i, 1 - j] for i in range(2) for j in range(2))
    assert exp[0].tolist() == [1.0, 2.0, 2.0, 4.0] and exp[1].tolist() == [3.0, 4.0, 6.0, 8.0]  # the hand values
    y = torch.empty(1, 4, 4, 1, device=dev)
    rc = N.lib().tbg_upfirdn2d_f32(N.ptr(x.to(dev)), N.ptr(k.to(dev)), N.ptr(y), 1, 2, 2, 1, 2, 2, 2, 2, 1, 1, 1, 0, 1, 0,
                                   N.stream())
    assert rc == 0 and torch.equal(y.cpu().reshape(4, 4), exp)


def test_upfirdn2d_generic_minor_and_big_filter(dev):
    from textboxgan_amd import native as N
    x = rnd(3, 9, 11, 2, seed=2)
    k = rnd(5, 6, seed=3)
    ref = R.t_upfirdn2d(x, k.numpy(), upx=2, upy=1, downx=1, downy=2, padx0=2, padx1=3, pady0=1, pady1=2)
    xd = x.float().to(dev).contiguous()
    kd = k.float().to(dev).contiguous()
    y = torch.empty(ref.shape, device=dev)
    rc = N.lib().tbg_upfirdn2d_f32(N.ptr(xd), N.ptr(kd), N.ptr(y), 3, 9, 11, 2, 5, 6, 2, 1, 1, 2, 2, 3, 1, 2, N.stream())
    assert rc == 0
    assert rel_err(y, ref) < 1e-5


def test_upfirdn2d_error_codes(dev):
    from textboxgan_amd import native as N
    x = torch.zeros(1, 4, 4, 1, device=dev)
    k = torch.ones(4, 4, device=dev)
    y = torch.zeros(1, 8, 8, 1, device=dev)
    L = N.lib()
    assert L.tbg_upfirdn2d_f32(N.ptr(x), N.ptr(k), N.ptr(y), 1, 4, 4, 1, 4, 4, 0, 1, 1, 1, 0, 0, 0, 0, N.stream()) == -1
    assert L.tbg_upfirdn2d_f32(N.ptr(x), N.ptr(k), N.ptr(y), 1, 2, 2, 1, 4, 4, 1, 1, 1, 1, 0, 0, 0, 0, N.stream()) == -1  # empty out
    assert L.tbg_upfirdn2d_f32(None, N.ptr(k), N.ptr(y), 1, 4, 4, 1, 4, 4, 1, 1, 1, 1, 0, 0, 0, 0, N.stream()) == -1
    assert b"invalid" in L.tbg_strerror(-1)


def test_conv_and_friends_error_codes(dev):
    """argument validation of the other entries: a bad call returns a negative code (never launches, never throws) --
    the C-ABI counterpart of the reference op's OP_REQUIRES checks (upfirdn_2d.cu:232-307)."""
    import ctypes as C
    from textboxgan_amd import native as N, ops
    L = N.lib()
    EINVAL, EUNSUP = -1, -4
    x = torch.zeros(1, 8, 4, 4, device=dev)
    wp = ops.pack_filter(torch.zeros(9, 8, 8, device=dev), False, False)
    y = torch.zeros(1, 8, 4, 4, device=dev)
    e = N.epilogue()
    ok = N.ConvDesc(1, 8, 8, 4, 4, 4, 4, 3, 3, 1, 1, 1, 1, 0, 0, 8, 1)
    call = lambda d, xx=x, ww=wp.data, yy=y, ep=e: L.tbg_conv2d_f32(C.byref(d), N.ptr(xx), N.ptr(ww), N.ptr(yy), None,
                                                                   C.byref(ep), N.stream())
    assert call(ok) == 0
    assert call(ok, xx=None) == EINVAL and call(ok, yy=None) == EINVAL
    assert call(N.ConvDesc(0, 8, 8, 4, 4, 4, 4, 3, 3, 1, 1, 1, 1, 0, 0, 8, 1)) == EINVAL            # B = 0
    assert call(N.ConvDesc(1, 8, 8, 4, 4, 4, 4, 5, 5, 1, 1, 2, 2, 0, 0, 8, 1)) == EUNSUP            # 25 taps
    assert call(N.ConvDesc(1, 8, 8, 4, 4, 4, 4, 3, 3, 3, 1, 1, 1, 0, 0, 8, 1)) == EUNSUP            # stride 3
    assert call(N.ConvDesc(1, 8, 8, 4, 4, 4, 4, 3, 3, 1, 1, 1, 1, 0, 0, 4, 1)) == EINVAL            # ldw < M
    assert call(N.ConvDesc(1, 8, 8, 4, 4, 4, 4, 3, 3, 1, 1, 1, 1, 0, 0, 8, 0)) == EINVAL            # ksplit 0
    assert call(N.ConvDesc(1, 8, 8, 4, 4, 4, 4, 3, 3, 1, 1, 1, 1, 0, 0, 8, 2),
                ep=N.epilogue(bias=torch.zeros(8, device=dev))) == EINVAL                            # split-K + epilogue
    assert call(N.ConvDesc(1, 8, 8, 4, 4, 4, 4, 3, 3, 2, 2, 1, 1, 1, 0, 8, 1)) == EINVAL            # transposed with padding
    wd = N.WgradDesc(1, 8, 8, 4, 4, 4, 4, 3, 3, 1, 1, 1, 1, 72, 8, 1, 1.0)
    nbytes = L.tbg_conv2d_wgrad_workspace_bytes(C.byref(wd))
    assert nbytes > 0
    ws = torch.empty(nbytes // 4, device=dev)
    dw = torch.zeros(9, 8, 8, device=dev)
    wg = lambda d, w_=ws, nb=nbytes: L.tbg_conv2d_wgrad_f32(C.byref(d), N.ptr(x), N.ptr(x), N.ptr(dw), None, None, N.ptr(w_), nb,
                                                         N.stream())
    assert wg(wd) == 0
    assert wg(wd, nb=16) == EINVAL                                                                   # workspace too small
    assert wg(N.WgradDesc(1, 8, 8, 4, 4, 4, 4, 2, 2, 1, 1, 0, 0, 32, 8, 1, 1.0)) == EUNSUP           # 2x2 filter
    assert L.tbg_conv2d_wgrad_workspace_bytes(C.byref(N.WgradDesc(1, 8, 8, 4, 4, 4, 4, 2, 2, 1, 1, 0, 0, 32, 8, 1, 1.0))) < 0
    assert L.tbg_weight_pack_f32(N.ptr(x), None, 9, 8, 8, 0, 0, N.stream()) == EINVAL
    assert L.tbg_slab_epilogue_f32(N.ptr(x), N.ptr(y), 1, 8, 16, 0, C.byref(e), N.stream()) == EINVAL   # nslab 0
    h = torch.zeros(1, 2, 4, device=dev)
    assert L.tbg_lstm_step_fwd_f32(None, None, N.ptr(h), N.ptr(h), N.ptr(h), None, 1, 3, 2, 4, 0, N.stream()) == EINVAL
    assert L.tbg_lstm_step_fwd_f32(N.ptr(h), None, N.ptr(h), N.ptr(h), N.ptr(h), None, 1, 3, 2, 4, 3, N.stream()) == EINVAL  # s >= T
    assert L.tbg_attn_ctx_fwd_f32(N.ptr(h), N.ptr(h), N.ptr(h), N.ptr(h), N.ptr(h), N.ptr(h), 1, 65, 4, 4, N.stream()) == EUNSUP
    torch.cuda.synchronize()


def test_upfirdn2d_grad_and_gradgrad(dev):
    """first and second order gradients through the recursive Function == autograd through the oracle."""
    from textboxgan_amd import ops
    k64 = torch.from_numpy(R.setup_kernel([1, 3, 3, 1]).astype(np.float64) * 4)
    for up, down, pad in (((2, 2), (1, 1), (2, 1, 2, 1)), ((1, 1), (2, 2), (1, 2, 1, 2)), ((1, 1), (1, 1), (2, 3, 2, 3))):
        x = rnd(2, 3, 8, 12, seed=4).requires_grad_(True)
        y = R.t_simple_upfirdn2d(x, k64.numpy(), up=up[0], down=down[0], pad0=pad[0], pad1=pad[1])
        gy = rnd(*y.shape, seed=5).requires_grad_(True)
        (gx,) = torch.autograd.grad(y, x, gy, create_graph=True)
        gg = rnd(*gx.shape, seed=6)
        (ggy,) = torch.autograd.grad(gx, gy, gg)

        xd = x.detach().float().to(dev).requires_grad_(True)
        gyd = gy.detach().float().to(dev).requires_grad_(True)
        yd = ops.upfirdn2d(xd, k64.float().to(dev), up, down, pad)
        (gxd,) = torch.autograd.grad(yd, xd, gyd, create_graph=True)
        (ggyd,) = torch.autograd.grad(gxd, gyd, gg.float().to(dev))
        assert rel_err(yd, y) < 1e-5
        assert rel_err(gxd, gx) < 1e-5
        assert rel_err(ggyd, ggy) < 1e-5


# ---------------------------------------------------------------------------------------- conv
def ref_conv(x, w_hwio, stride, pad, transposed=False):
    wt = w_hwio.permute(3, 2, 0, 1)
    if transposed:
        return F.conv_transpose2d(x, w_hwio.permute(2, 3, 0, 1), stride=stride)
    return F.conv2d(x, wt, stride=stride, padding=pad)


CONV_CASES = [
    # B, C, M, H, W, k, stride, pad, note
    (2, 16, 32, 16, 64, 3, (1, 1), (1, 1), "3x3 small"),
    (2, 128, 128, 16, 64, 3, (1, 1), (1, 1), "3x3 128ch"),
    (3, 24, 64, 12, 40, 3, (1, 1), (1, 1), "3x3 odd sizes M64"),
    (4, 513, 512, 4, 4, 3, (1, 1), (1, 1), "head conv 513 (split-K, 4x4)"),
    (2, 64, 128, 34, 66, 3, (2, 2), (0, 0), "3x3 stride 2 VALID"),
    (2, 32, 48, 10, 34, 3, (1, 2), (0, 0), "3x3 stride (1,2) VALID"),
    (2, 3, 64, 16, 64, 1, (1, 1), (0, 0), "fromRGB 1x1"),
    (2, 64, 128, 16, 32, 1, (1, 1), (0, 0), "skip 1x1"),
    (16, 512, 512, 4, 16, 3, (1, 1), (1, 1), "4x16 512ch batch16 (split-K)"),
]


@arith_modes
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[-1] for c in CONV_CASES])
def test_conv2d_forward(dev, case):
    from textboxgan_amd import ops
    B, C, M, H, W, k, stride, pad, _ = case
    x = rnd(B, C, H, W, seed=10)
    w = rnd(k, k, C, M, seed=11)
    ref = ref_conv(x, w, stride, pad)
    y = ops.conv2d_raw(x.float().to(dev), w.float().to(dev), M, k, k, (ref.shape[2], ref.shape[3]), stride, pad)
    assert rel_err(y, ref) < 2e-5


@arith_modes
def test_conv2d_flip(dev):
    from textboxgan_amd import ops
    x = rnd(2, 16, 8, 16, seed=12)
    w = rnd(3, 3, 16, 32, seed=13)
    ref = ref_conv(x, torch.flip(w, (0, 1)), (1, 1), (1, 1))
    y = ops.conv2d_raw(x.float().to(dev), w.float().to(dev), 32, 3, 3, (8, 16), (1, 1), (1, 1), flip=True)
    assert rel_err(y, ref) < 2e-5


TCONV_CASES = [
    (2, 16, 32, 4, 16, (2, 2), "up 4x16"),
    (2, 128, 128, 16, 64, (2, 2), "up 16x64 128ch"),
    (2, 32, 16, 8, 9, (1, 2), "transposed stride (1,2)"),
    (3, 512, 256, 4, 16, (2, 2), "up 512->256 (split-K)"),
]


@arith_modes
@pytest.mark.parametrize("case", TCONV_CASES, ids=[c[-1] for c in TCONV_CASES])
def test_conv2d_transposed(dev, case):
    from textboxgan_amd import ops
    B, C, M, H, W, stride, _ = case
    x = rnd(B, C, H, W, seed=14)
    w = rnd(3, 3, C, M, seed=15)
    ref = F.conv_transpose2d(x, w.permute(2, 3, 0, 1), stride=stride)
    y = ops.conv2d_raw(x.float().to(dev), w.float().to(dev), M, 3, 3, (ref.shape[2], ref.shape[3]), stride, (0, 0),
                       transposed=True)
    assert rel_err(y, ref) < 2e-5
    # padded output (rows/cols past the natural extent must be written as zeros)
    y2 = ops.conv2d_raw(x.float().to(dev), w.float().to(dev), M, 3, 3, (ref.shape[2] + 1, ref.shape[3] + 1), stride,
                        (0, 0), transposed=True)
    ref2 = F.pad(ref, (0, 1, 0, 1))
    assert rel_err(y2, ref2) < 2e-5


@arith_modes
def test_conv2d_fused_epilogue(dev):
    from textboxgan_amd import ops, native as N
    B, C, M, H, W = 3, 32, 48, 8, 32
    x, w = rnd(B, C, H, W, seed=16), rnd(3, 3, C, M, seed=17)
    s, d = rnd(B, C, seed=18) + 1.0, rnd(B, M, seed=19).abs() + 0.5
    noise, bias, res = rnd(B, 1, H, W, seed=20), rnd(M, seed=21), rnd(B, M, H, W, seed=22)
    strength = torch.tensor(0.37, dtype=torch.float64)
    alpha = 0.123
    pre = F.conv2d(x * s[:, :, None, None], w.permute(3, 2, 0, 1), padding=1) * alpha * d[:, :, None, None]
    pre = pre + noise * strength + bias[None, :, None, None] * 0.5
    ref = (F.leaky_relu(pre, 0.2) * math.sqrt(2) + res) * 0.7
    f = lambda t: t.float().to(dev).contiguous()
    dd, bd, nd, sd, rd = f(d), f(bias), f(noise), f(strength), f(res)  # keep alive: the epilogue holds raw pointers
    epi = N.epilogue(out_scale=dd, bias=bd, noise=nd, strength=sd, residual=rd, alpha=alpha,
                     bias_mul=0.5, act=N.ACT_LRELU, res_scale=0.7)
    y = ops.conv2d_raw(f(x), f(w), M, 3, 3, (H, W), (1, 1), (1, 1), in_scale=f(s), epi=epi)
    assert rel_err(y, ref) < 2e-5
    # fused dot product of the unscaled accumulator
    aux = rnd(B, M, H, W, seed=23)
    dot = torch.zeros(B, M, device=dev)
    epi2 = N.epilogue(out_scale=dd, alpha=alpha)
    y2 = ops.conv2d_raw(f(x), f(w), M, 3, 3, (H, W), (1, 1), (1, 1), epi=epi2, dot=(f(aux), dot))
    raw = F.conv2d(x, w.permute(3, 2, 0, 1), padding=1) * alpha
    assert rel_err(y2, raw * d[:, :, None, None]) < 2e-5
    assert rel_err(dot, (raw * aux).sum(dim=(2, 3))) < 2e-5


@arith_modes
@pytest.mark.parametrize("ksplit", [None, 4])
def test_conv2d_gate_epilogue(dev, ksplit):
    """out = gate > 0 ? acc + residual : 0 (the frozen ResNet's backward: residual sum and ReLU gate of the unit in front on
    the data-gradient launch), unsplit and through the split-K slab pass; 3x3 stride 1 and the strided 1x1 transposed form."""
    from textboxgan_amd import ops, native as N
    f = lambda t: t.float().to(dev).contiguous()
    B, C, M, H, W = 3, 64, 48, 4, 25
    x, w = rnd(B, C, H, W, seed=41), rnd(3, 3, C, M, seed=42)
    res, gate = rnd(B, M, H, W, seed=43), rnd(B, M, H, W, seed=44)
    gate = torch.where(gate.abs() < 0.3, torch.zeros_like(gate), gate)  # exact zeros gate off, as relu outputs do
    ref = (F.conv2d(x, w.permute(3, 2, 0, 1), padding=1) + res) * (gate > 0)
    rd, gd = f(res), f(gate)
    ops.TUNING.force_ksplit = ksplit
    try:
        y = ops.conv2d_raw(f(x), f(w), M, 3, 3, (H, W), (1, 1), (1, 1), epi=N.epilogue(residual=rd, res_first=1, gate=gd))
        assert rel_err(y, ref) < 2e-5
        assert torch.equal(y == 0, (ref == 0).to(dev))
        # 1x1 stride (2, 1) transposed (data gradient of a strided shortcut): three of four... here every second row has no tap
        xs, ws = rnd(B, C, 4, 25, seed=45), rnd(1, 1, M, C, seed=46)  # forward conv M -> C, stride (2, 1): dgrad maps C -> M
        res2, gate2 = rnd(B, M, 8, 25, seed=47), rnd(B, M, 8, 25, seed=48)
        ref2 = (F.conv_transpose2d(xs, ws.permute(3, 2, 0, 1), stride=(2, 1), output_padding=(1, 0)) + res2) * (gate2 > 0)
        pf = ops.pack_filter(f(ws), transpose=True, flip=False)
        r2, g2 = f(res2), f(gate2)
        y2 = ops.conv2d_raw(f(xs), pf, M, 1, 1, (8, 25), (2, 1), (0, 0), transposed=True,
                            epi=N.epilogue(residual=r2, res_first=1, gate=g2))
        assert rel_err(y2, ref2) < 2e-5
    finally:
        ops.TUNING.force_ksplit = None


WG_CASES = [
    (2, 16, 32, 16, 64, 3, (1, 1), (1, 1), "3x3 s1"),
    (2, 128, 128, 16, 64, 3, (1, 1), (1, 1), "3x3 s1 128ch"),
    (3, 24, 40, 9, 21, 3, (1, 1), (1, 1), "3x3 odd"),
    (2, 64, 96, 34, 66, 3, (2, 2), (0, 0), "3x3 s2 VALID"),
    (2, 32, 48, 10, 34, 3, (1, 2), (0, 0), "3x3 s(1,2) VALID"),
    (2, 3, 64, 16, 64, 1, (1, 1), (0, 0), "1x1 fromRGB"),
    (4, 513, 512, 4, 4, 3, (1, 1), (1, 1), "513->512 4x4"),
    (3, 40, 72, 9, 17, 1, (2, 2), (0, 0), "1x1 s2 (strided shortcut)"),
    (2, 24, 40, 5, 3, 3, (1, 1), (1, 1), "3x3 on a 3-wide map (narrow-tile path)"),
]


@pytest.mark.parametrize("case", WG_CASES, ids=[c[-1] for c in WG_CASES])
def test_conv2d_wgrad(dev, case):
    from textboxgan_amd import ops
    B, C, M, H, W, k, stride, pad, _ = case
    x = rnd(B, C, H, W, seed=30).requires_grad_(True)
    w = rnd(k, k, C, M, seed=31).requires_grad_(True)
    y = ref_conv(x, w, stride, pad)
    dy = rnd(*y.shape, seed=32)
    (dw_ref,) = torch.autograd.grad(y, w, dy)
    g = ops._Geom(stride, pad, k, k, (H, W), (y.shape[2], y.shape[3]))
    dw = ops._bwd_weight_launch(x.detach().float().to(dev), dy.float().to(dev), g, C, M, alpha=1.0)
    assert rel_err(dw, dw_ref) < 3e-5


WGV_CASES = [
    (3, 40, 72, 5, 36, "ragged: odd rows, a partial 32-wide tile, partial channel tiles"),
    (2, 64, 64, 8, 32, "exactly one tile per row"),
    (1, 130, 70, 2, 128, "three / two channel tiles, one row pair"),
    (2, 64, 64, 3, 100, "W = 100: quads end inside the last tile"),
]


@pytest.mark.parametrize("case", WGV_CASES, ids=[c[-1] for c in WGV_CASES])
def test_conv2d_wgrad_vector_staging(dev, case):
    """the float4-staged filter-gradient instance (stride-1 3x3, rows of whole 16-byte units, W >= 32) against a float64
    reference, with both per-(sample, channel) scale vectors and the fused additive term; and that it IS that instance."""
    from textboxgan_amd import ops, native as N
    B, C, M, H, W, _ = case
    x, dy = rnd(B, C, H, W, seed=40), rnd(B, M, H, W, seed=41)
    xs, ds = rnd(B, C, seed=42).abs() + 0.5, rnd(B, M, seed=43).abs() + 0.5
    addw, addq = rnd(3, 3, C, M, seed=44), rnd(C, M, seed=45)
    w = torch.zeros(3, 3, C, M, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(x * xs[:, :, None, None], w.permute(3, 2, 0, 1), padding=1)
    (ref,) = torch.autograd.grad(y, w, dy * ds[:, :, None, None])
    ref = 0.7 * ref + 0.3 * addw * addq[None, None]
    f = lambda t: t.float().to(dev).contiguous()
    g = ops._Geom((1, 1), (1, 1), 3, 3, (H, W), (H, W))
    desc = N.WgradDesc(B, M, C, H, W, H, W, 3, 3, 1, 1, 1, 1, C * M, M, 1, 0.7)
    assert N.wgrad_kernel_name(desc) == "conv_wgrad_kernel<2, 2, 9, 64, true, 1>", N.wgrad_kernel_name(desc)
    dw = ops._bwd_weight_launch(f(x), f(dy), g, C, M, alpha=0.7, x_scale=f(xs), dy_scale=f(ds), add=(f(addw), f(addq), 0.3))
    assert rel_err(dw, ref) < 3e-5
    dw_plain = ops._bwd_weight_launch(f(x), f(dy), g, C, M, alpha=1.0)
    (ref_plain,) = torch.autograd.grad(F.conv2d(x, w.permute(3, 2, 0, 1), padding=1), w, dy)
    assert rel_err(dw_plain, ref_plain) < 3e-5
    # padding is SELECTED to zero, not multiplied: the branch-free loads of out-of-range positions read element 0 of the
    # tensor, and an Inf sitting there must stay inside its own output channel
    dyi = f(dy).clone()
    dyi[0, 0, 0, 0] = float("inf")
    dwi = ops._bwd_weight_launch(f(x), dyi, g, C, M, alpha=1.0)
    assert bool(torch.isfinite(dwi[..., 1:]).all()) and rel_err(dwi[..., 1:], ref_plain[..., 1:]) < 3e-5


WGV2_CASES = [
    (2, 40, 72, 11, 65, "odd map 11x65 -> 5x32, partial channel tiles"),
    (3, 64, 64, 9, 130, "9x130 -> 4x64 (two tiles per row, Wl = 2 Ws + 2)"),
    (1, 130, 70, 5, 129, "5x129 -> 2x64, three / two channel tiles"),
]


@pytest.mark.parametrize("bf16", [False, True], ids=["f32", "bf16"])
@pytest.mark.parametrize("case", WGV2_CASES, ids=[c[-1] for c in WGV2_CASES])
def test_conv2d_wgrad_vector_staging_stride2(dev, case, bf16):
    """the float4-staged stride-2 VALID filter-gradient instances (fp32 and bf16): scales, additive term, ragged channels."""
    from textboxgan_amd import ops, native as N
    B, C, M, H, W, _ = case
    Ho, Wo = (H - 3) // 2 + 1, (W - 3) // 2 + 1
    x, dy = rnd(B, C, H, W, seed=50), rnd(B, M, Ho, Wo, seed=51)
    xs, ds = rnd(B, C, seed=52).abs() + 0.5, rnd(B, M, seed=53).abs() + 0.5
    addw, addq = rnd(3, 3, C, M, seed=54), rnd(C, M, seed=55)
    rd = (lambda t: t.float().bfloat16().double()) if bf16 else (lambda t: t)
    f32 = lambda t: t.float().double()
    xr = rd((f32(x) * f32(xs)[:, :, None, None]).float().double())
    dyr = rd((f32(dy) * f32(ds)[:, :, None, None]).float().double())
    w = torch.zeros(3, 3, C, M, dtype=torch.float64, requires_grad=True)
    (ref,) = torch.autograd.grad(F.conv2d(xr, w.permute(3, 2, 0, 1), stride=2), w, dyr)
    ref = 0.7 * ref + 0.3 * f32(addw) * f32(addq)[None, None]
    f = lambda t: t.float().to(dev).contiguous()
    g = ops._Geom((2, 2), (0, 0), 3, 3, (H, W), (Ho, Wo))
    desc = N.WgradDesc(B, M, C, Ho, Wo, H, W, 3, 3, 2, 2, 0, 0, C * M, M, 1, 0.7)
    want = "conv_wgrad_bf16_kernel<2, 2, 9, 32, 2, 2>" if bf16 else "conv_wgrad_kernel<2, 2, 9, 32, true, 2>"
    assert N.wgrad_kernel_name(desc, bf16) == want, N.wgrad_kernel_name(desc, bf16)
    with ops.compute_dtype("bf16" if bf16 else "f32"):
        dw = ops._bwd_weight_launch(f(x), f(dy), g, C, M, alpha=0.7, x_scale=f(xs), dy_scale=f(ds), add=(f(addw), f(addq), 0.3))
    assert rel_err(dw, ref) < 5e-5


@arith_modes
def test_conv_random_shapes_all_three_passes(dev):
    """seeded sweep over awkward shapes (channels not multiples of 4 / 32, odd maps, 1- and 2-pixel maps, every stride
    the kernels accept): forward, data gradient and filter gradient against float64 autograd."""
    from textboxgan_amd import ops
    rng = np.random.RandomState(1234)
    n_split = 0
    for case in range(28):
        k = int(rng.choice([1, 3]))
        stride = (1, 1) if rng.rand() < 0.6 else tuple(int(v) for v in rng.choice([1, 2], size=2))
        pad = (k // 2, k // 2) if stride == (1, 1) else (0, 0)
        B = int(rng.randint(1, 6))
        C = int(rng.choice([1, 3, 5, 8, 17, 33, 64, 100, 130]))
        M = int(rng.choice([1, 3, 7, 16, 31, 40, 64, 96, 129]))
        H = int(rng.randint(max(1, k if pad == (0, 0) else 1), 20))
        W = int(rng.randint(max(1, k if pad == (0, 0) else 1), 40))
        x = rnd(B, C, H, W, seed=100 + case).requires_grad_(True)
        w = rnd(k, k, C, M, seed=200 + case).requires_grad_(True)
        y = ref_conv(x, w, stride, pad)
        dy = rnd(*y.shape, seed=300 + case)
        gx, gw = torch.autograd.grad(y, (x, w), dy)
        g = ops._Geom(stride, pad, k, k, (H, W), (y.shape[2], y.shape[3]))
        xd, wd, dyd = x.detach().float().to(dev), w.detach().float().to(dev), dy.float().to(dev)
        tag = (case, B, C, M, H, W, k, stride)
        assert rel_err(ops._fwd_launch(xd, wd, g), y) < 3e-5, ("fwd", tag)
        if k == 3 or stride == (1, 1) or True:
            assert rel_err(ops._bwd_data_launch(dyd, wd, g), gx) < 3e-5, ("dgrad", tag)
        if (k == 3) or (k == 1):
            assert rel_err(ops._bwd_weight_launch(xd, dyd, g, C, M), gw) < 5e-5, ("wgrad", tag)
        n_split += 1
    assert n_split == 28


@arith_modes
def test_conv_primitives_double_backward(dev):
    """conv2d / bwd_data / bwd_weight close under differentiation (R1 and path-length need it)."""
    from textboxgan_amd import ops
    for stride, pad, H, W in (((1, 1), (1, 1), 8, 12), ((2, 2), (0, 0), 9, 13), ((1, 2), (0, 0), 6, 13)):
        x = rnd(2, 8, H, W, seed=40).requires_grad_(True)
        w = rnd(3, 3, 8, 16, seed=41).requires_grad_(True)
        y = ref_conv(x, w, stride, pad)
        gy = rnd(*y.shape, seed=42).requires_grad_(True)
        gx, gw = torch.autograd.grad(y, (x, w), gy, create_graph=True)
        loss = gx.square().sum() + (gw * rnd(*gw.shape, seed=43)).sum()
        r = torch.autograd.grad(loss, (x, w, gy))

        f = lambda t: t.detach().float().to(dev).requires_grad_(True)
        xd, wd, gyd = f(x), f(w), f(gy)
        yd = ops.conv2d(xd, wd, stride, pad)
        gxd, gwd = torch.autograd.grad(yd, (xd, wd), gyd, create_graph=True)
        lossd = gxd.square().sum() + (gwd * rnd(*gw.shape, seed=43).float().to(dev)).sum()
        rd = torch.autograd.grad(lossd, (xd, wd, gyd))
        assert rel_err(yd, y) < 2e-5 and rel_err(gxd, gx) < 2e-5 and rel_err(gwd, gw) < 2e-5
        for a, b in zip(rd, r):
            assert rel_err(a, b) < 5e-5


@arith_modes
def test_conv_transpose_primitive(dev):
    from textboxgan_amd import ops
    x = rnd(2, 8, 5, 7, seed=44).requires_grad_(True)
    wt = rnd(3, 3, 8, 12, seed=45).requires_grad_(True)
    y = F.conv_transpose2d(x, wt.permute(2, 3, 0, 1), stride=2)
    gy = rnd(*y.shape, seed=46)
    gx, gw = torch.autograd.grad(y, (x, wt), gy)
    f = lambda t: t.detach().float().to(dev).requires_grad_(True)
    xd, wd = f(x), f(wt)
    yd = ops.conv_transpose2d_s2(xd, wd)
    gxd, gwd = torch.autograd.grad(yd, (xd, wd), gy.float().to(dev))
    assert rel_err(yd, y) < 2e-5 and rel_err(gxd, gx) < 2e-5 and rel_err(gwd, gw) < 2e-5


# ---------------------------------------------------------------------------------------- elementwise
def test_bias_act_fwd_bwd(dev):
    from textboxgan_amd import ops, native as N
    for HW in (64, 4096 + 12, 16384):
        B, M = 2, 5
        x = rnd(B, M, HW, seed=50).requires_grad_(True)
        d = (rnd(B, M, seed=51).abs() + 0.5).requires_grad_(True)
        bias = rnd(M, seed=52).requires_grad_(True)
        noise = rnd(B, 1, HW, seed=53)
        strength = torch.tensor(0.3, dtype=torch.float64, requires_grad=True)
        pre = x * d[:, :, None] + noise * strength + bias[None, :, None]
        out = F.leaky_relu(pre, 0.2) * math.sqrt(2)
        dout = rnd(B, M, HW, seed=54)
        gx, gd, gb, gs = torch.autograd.grad(out, (x, d, bias, strength), dout)
        f = lambda t: t.detach().float().to(dev).contiguous()
        dd, bd, nd, sd = f(d), f(bias), f(noise), f(strength)  # keep alive: the epilogue holds raw pointers
        epi = N.epilogue(out_scale=dd, bias=bd, noise=nd, strength=sd, act=N.ACT_LRELU)
        y = ops.bias_act_fwd_raw(f(x), epi)
        assert rel_err(y, out) < 1e-5
        dx, dpre, pdb, pdn, pdy = ops.bias_act_bwd_raw(f(dout), y, epi, want_dx=True, want_dn=True, want_dyy=True)
        assert rel_err(dx, gx) < 1e-5
        assert rel_err(pdb.sum(dim=(0, 2)), gb) < 1e-4
        assert rel_err(pdn.sum(), gs) < 1e-4
        assert rel_err(pdy.sum(dim=2) / dd, gd) < 1e-4


def test_weight_pack(dev):
    """tbg_weight_pack_f32: Wp[t'][c/4][m][c%4], zero padded past C, for both orientations and tap orders."""
    from textboxgan_amd import ops
    w = rnd(3, 3, 13, 20, seed=60).float()
    for transpose in (False, True):
        for flip in (False, True):
            pf = ops.pack_filter(w.to(dev), transpose, flip)
            src = (torch.flip(w, (0, 1)) if flip else w).reshape(9, 13, 20)
            gemm = src.permute(0, 2, 1) if transpose else src        # [T, C, M]
            T, Cc, M = gemm.shape
            assert (pf.T, pf.C, pf.M) == (T, Cc, M)
            C4 = (Cc + 3) // 4
            ref = torch.zeros(T, C4 * 4, M)
            ref[:, :Cc] = gemm
            ref = ref.reshape(T, C4, 4, M).permute(0, 1, 3, 2).contiguous()
            assert torch.equal(pf.data.cpu().reshape(T, C4, M, 4), ref)


def test_adam_and_ema(dev):
    from textboxgan_amd import ops
    n = 10007
    th, g = rnd(n, seed=70), rnd(n, seed=71)
    m, v = rnd(n, seed=72) * 0.1, rnd(n, seed=73).abs() * 0.1
    lr, b1, b2, eps, t = 1.7778e-3, 0.0, 0.99111, 1e-8, 7
    m2 = b1 * m + (1 - b1) * g
    v2 = b2 * v + (1 - b2) * g * g
    th2 = th - lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t) * m2 / (v2.sqrt() + eps)
    f = lambda x: x.float().to(dev).contiguous()
    thd, md, vd, gd = f(th), f(m), f(v), f(g)
    step = torch.tensor([t - 1], dtype=torch.int64, device=dev)
    ops.adam_tf_(thd, md, vd, gd, step, lr, b1, b2, eps)
    assert rel_err(thd, th2) < 1e-6 and rel_err(md, m2) < 1e-6 and rel_err(vd, v2) < 1e-6
    dst, src = f(th), f(g)
    ops.ema_lerp_(dst, src, 0.99)
    assert rel_err(dst, g + (th - g) * 0.99) < 1e-6


def test_demod_coefs(dev):
    from textboxgan_amd import ops
    s = (rnd(5, 24, seed=80) + 1).requires_grad_(True)
    w = rnd(3, 3, 24, 40, seed=81).requires_grad_(True)
    coef = 1 / math.sqrt(9 * 24)
    ww = (w * coef)[None] * s[:, None, None, :, None]
    d = torch.rsqrt(ww.square().sum(dim=(1, 2, 3)) + 1e-8)
    gd = rnd(5, 40, seed=82)
    gs, gw = torch.autograd.grad(d, (s, w), gd)
    f = lambda t: t.detach().float().to(dev).requires_grad_(True)
    sd, wd = f(s), f(w)
    dd = ops.demod_coefs(sd, wd)
    gsd, gwd = torch.autograd.grad(dd, (sd, wd), gd.float().to(dev))
    assert rel_err(dd, d) < 1e-5 and rel_err(gsd, gs) < 1e-4 and rel_err(gwd, gw) < 1e-4


# ---------------------------------------------------------------------------------------- small-tensor tails
def test_modconv_bwd_smalls(dev):
    """tbg_modconv_bwd_smalls_f32 == the torch composition it replaces (modulated_conv2d.py:78-82), float64."""
    from textboxgan_amd import ops
    for B, I, O, nch in ((16, 512, 512, 1), (5, 24, 40, 3), (32, 128, 128, 4), (16, 128, 130, 16), (7, 30, 1030, 2)):
        pdb, pdn, pdy = rnd(B, O, nch, seed=90), rnd(B, O, nch, seed=91), rnd(B, O, nch, seed=92)
        d, s = rnd(B, O, seed=93).abs() + 0.5, rnd(B, I, seed=94) + 1.0
        wsq, ds_conv = rnd(I, O, seed=95).abs(), rnd(B, I, seed=96)
        t = pdy.sum(dim=2) * d.square()
        ds_ref = ds_conv - s * (t @ wsq.t())
        dwsq_ref = s.square().t() @ t
        f = lambda a: a.float().to(dev).contiguous()
        db, dstr, ds, dwsq = ops.modconv_bwd_smalls_raw(f(pdb), f(pdn), f(pdy), f(d), f(s), f(wsq), f(ds_conv))
        assert rel_err(db, pdb.sum(dim=(0, 2))) < 1e-5 and rel_err(dstr, pdn.sum()) < 1e-4
        assert rel_err(ds, ds_ref) < 2e-5 and rel_err(dwsq, dwsq_ref) < 2e-5
        # round 5: the convolution's style dot arrives as its launch's partial slots [B, I, slots], summed by the kernel
        parts = rnd(B, I, 5, seed=97)
        _, _, ds_p, _ = ops.modconv_bwd_smalls_raw(f(pdb), f(pdn), f(pdy), f(d), f(s), f(wsq), f(parts))
        assert rel_err(ds_p, parts.sum(dim=2) - s * (t @ wsq.t())) < 2e-5


def test_torgb_bwd_smalls(dev):
    from textboxgan_amd import ops
    B, Cc, O, coef = 7, 130, 3, 0.25
    G, w, s = rnd(B, Cc, O, seed=97), rnd(Cc, O, seed=98), rnd(B, Cc, seed=99)
    f = lambda a: a.float().to(dev).contiguous()
    ds, dw = ops.torgb_bwd_smalls_raw(f(G), f(w), f(s), coef)
    assert rel_err(ds, coef * (G * w[None]).sum(dim=2)) < 1e-5
    assert rel_err(dw, coef * (G * s[:, :, None]).sum(dim=0)) < 1e-5
    # round 5: per-pixel-chunk partials of the Gram and of the masked dy, summed by the kernel (+ the bias gradient)
    Gp, dys = rnd(B, Cc, 4, O, seed=100), rnd(B, 4, O, seed=101)
    ds, dw, db = ops.torgb_bwd_smalls_raw(f(Gp), f(w), f(s), coef, dysum=f(dys))
    Gs = Gp.sum(dim=2)
    assert rel_err(ds, coef * (Gs * w[None]).sum(dim=2)) < 1e-5 and rel_err(dw, coef * (Gs * s[:, :, None]).sum(dim=0)) < 1e-5
    assert rel_err(db, dys.sum(dim=(0, 1))) < 1e-5


def test_rgb_backproject_parts_and_bias_rider(dev):
    """(a) tbg_rgb_backproject_f32's dysum output = the per-chunk sums of the masked dy (ToRGB's bias gradient, to_rgb.py:28-33) and
    its chunk partials of G add up to the summed form; (b) tbg_wgrad_desc's bias rider: db = sum of the bias_act backward's partial
    sums, written by the filter gradient's reduce launch, for the NCHW and the unit-tensor filter gradients."""
    from textboxgan_amd import ops
    f = lambda a: a.float().to(dev).contiguous()
    B, Cc, H, W = 3, 16, 64, 96   # 6144 pixels: three 2048-pixel chunks
    x, dy, w, s = f(rnd(B, Cc, H, W, seed=110)), f(rnd(B, 3, H, W, seed=111)), f(rnd(Cc, 3, seed=112)), f(rnd(B, Cc, seed=113))
    mask = f((rnd(B, W // 32, seed=114) > 0).double())
    dx0, G0, dym0 = ops.rgb_backproject_raw(x, dy, w, s, 0.3, colmask=mask, mask_cw=32, want_dym=True)
    dx1, Gp, dysum, dym1 = ops.rgb_backproject_raw(x, dy, w, s, 0.3, colmask=mask, mask_cw=32, want_dym=True, parts=True)
    assert torch.equal(dx0, dx1) and torch.equal(dym0, dym1) and Gp.shape[2] == 3 and torch.equal(Gp.sum(dim=2), G0)
    assert rel_err(dysum.sum(dim=(0, 1)), dym0.double().sum(dim=(0, 2, 3)).cpu()) < 1e-5
    for arith, (B, I, O, H, W) in (("f32", (4, 24, 40, 8, 16)), ("f32x3", (2, 64, 64, 8, 32)), ("bf16", (2, 64, 128, 8, 32))):
        with ops.compute_dtype(arith):
            xx, dyy, parts = f(rnd(B, I, H, W, seed=120)), f(rnd(B, O, H, W, seed=121)), f(rnd(B, O, 3, seed=122))
            g = ops._Geom((1, 1), (1, 1), 3, 3, (H, W), (H, W))
            db = torch.full((O,), float("nan"), device=dev)
            dw_ref = ops._bwd_weight_launch(xx, dyy, g, I, O, alpha=0.5)
            dw = ops._bwd_weight_launch(xx, dyy, g, I, O, alpha=0.5, bias=(parts, db))
            assert torch.equal(dw, dw_ref) and rel_err(db, parts.double().sum(dim=(0, 2)).cpu()) < 1e-5, arith
            if arith != "f32":
                db.fill_(float("nan"))
                dwu = torch.empty(3, 3, I, O, device=dev)
                ops.wgrad_units_raw(ops.units_pack(dyy), ops.units_pack(xx), dwu, I * O, O, 1, 0.5, bias=(parts, db))
                assert rel_err(db, parts.double().sum(dim=(0, 2)).cpu()) < 1e-5, arith


@pytest.mark.parametrize("B", [16, 4, 2, 32], ids=lambda b: f"B{b}")
def test_minibatch_std_fused_fwd_bwd(dev, B):
    """mini_batch_std.py:10-35: forward and first-order gradient vs autograd through the float64 oracle."""
    from textboxgan_amd import ops
    x = rnd(B, 48, 4, 4, seed=101).requires_grad_(True)
    y = R.t_minibatch_std(x, 4)
    dy = rnd(*y.shape, seed=102)
    (gx,) = torch.autograd.grad(y, x, dy)
    xd = x.detach().float().to(dev).requires_grad_(True)
    yd = ops.minibatch_std_fused(xd, 4)
    (gxd,) = torch.autograd.grad(yd, xd, dy.float().to(dev))
    assert rel_err(yd, y) < 1e-5 and rel_err(gxd, gx) < 2e-5
    from textboxgan_amd import native as N
    bad = torch.zeros(6, 8, 4, 4, device=dev)
    out = torch.zeros(6, 9, 4, 4, device=dev)
    assert N.lib().tbg_minibatch_std_fwd_f32(N.ptr(bad), N.ptr(out), 6, 8, 16, 4, N.stream()) == -1  # 6 % 4 != 0
