"""-m gpu: PHASE unit tensors (tbg.h; csrc/conv_units_s2.hip) -- the stand-alone producer against its definition and the stride-2
convolution that reads them against float64 at the bar of the NCHW stride-2 kernels (tests/test_x3_gpu.py, test_bf16_gpu.py) and
against those kernels themselves."""
import math

import pytest
import torch

from textboxgan_amd import native as N, ops

pytestmark = pytest.mark.gpu


def _rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64)


def _rel(a, r):
    return float((a.double().cpu() - r).abs().max() / (r.abs().max() + 1e-30))


@pytest.mark.parametrize("planes", [3, 1])
@pytest.mark.parametrize("shape", [(2, 20, 17, 65), (1, 64, 18, 66), (3, 130, 5, 9)])
@pytest.mark.parametrize("scaled", [True, False])
def test_units_pack_s2_is_the_definition(dev, planes, shape, scaled):
    """P[pl][b][c/8][2 (Y&1) + (X&1)][Y>>1][X>>1][c%8] = t[b][c][Y][X]: planes = 3 sums EXACTLY to the fp32 product, planes = 1 is
    its RNE bf16; positions t does not have and the channel tail are zero; every unit is written."""
    B, C, Hin, Win = shape
    g = torch.Generator().manual_seed(7)
    x = (torch.randn(B, C, Hin, Win, generator=g) * torch.exp(3 * torch.randn(B, C, Hin, Win, generator=g))).to(dev)
    s = (torch.rand(B, C, generator=g) + 0.5).to(dev) if scaled else None
    P = ops.units_pack_s2(x, s, planes=planes)
    Hq, Wq, C8 = P.Ho + 1, P.Wo + 1, (C + 7) // 8
    assert P.data.numel() * 2 == N.lib().tbg_units_s2_bytes(B, C, P.Ho, P.Wo, planes)
    got = P.data.float().double().reshape(planes, B, C8, 2, 2, Hq, Wq, 8)
    assert torch.isfinite(got).all()
    v = (x * s[:, :, None, None]) if scaled else x
    exp = torch.zeros(B, C8 * 8, 2 * Hq, 2 * Wq, device=dev, dtype=torch.float32)
    exp[:, :C, :Hin, :Win] = v
    exp = exp.reshape(B, C8, 8, Hq, 2, Wq, 2).permute(0, 1, 4, 6, 3, 5, 2)  # [B, C8, py, px, i, j, 8]
    if planes == 3:
        assert torch.equal(got.sum(0), exp.double())
    assert torch.equal(got[0], exp.bfloat16().double())


CONV_S2 = [(2, 64, 64, 17, 65), (2, 128, 128, 18, 66), (3, 64, 192, 33, 129), (1, 256, 256, 34, 130), (2, 64, 128, 66, 258)]


@pytest.mark.parametrize("mode", ["f32x3", "bf16"])
@pytest.mark.parametrize("case", CONV_S2, ids=[str(c) for c in CONV_S2])
def test_conv_units_s2_matches_float64_and_the_nchw_kernel(dev, mode, case):
    """tbg_conv2d_units_s2 (phase tiles by LDS-DMA, two stages per channel chunk) with the full fused epilogue, and -- as the data
    gradient of the up-convolution -- with the transposed pack, out_scale and the fused dot product: against float64 on the operands
    the kernels see at the NCHW kernels' bar, and against the NCHW stride-2 kernel of the same arithmetic (same products; the
    taps are summed in another order, so fp32 rounding differs)."""
    import torch.nn.functional as F
    B, C, M, Hin, Win = case
    if mode == "bf16" and C % 16:
        pytest.skip("bf16 units need whole 16-channel chunks")
    planes = 3 if mode == "f32x3" else 1
    Ho, Wo = (Hin - 3) // 2 + 1, (Win - 3) // 2 + 1
    x, w = _rnd(B, C, Hin, Win, seed=1), _rnd(3, 3, C, M, seed=2) / math.sqrt(9 * C)
    s, dmod = _rnd(B, C, seed=3).abs() + 0.5, _rnd(B, M, seed=4).abs() + 0.5
    bias = _rnd(M, seed=6) * 0.2
    f = lambda t: t.float().to(dev).contiguous()
    xd, wd, sd, dd, bd = f(x), f(w), f(s), f(dmod), f(bias)
    xs = xd * sd[:, :, None, None]
    w_ref = wd.double().cpu()
    if mode == "bf16":
        xs, w_ref = xs.bfloat16().float(), wd.bfloat16().double().cpu()
    with ops.compute_dtype(mode):
        assert ops.conv_units_s2_ok(C, M, Hin, Win, planes)
        XP = ops.units_pack_s2(xd, sd, planes=planes)
        pf = ops.pack_filter(wd, False, False)
        epi = lambda: N.epilogue(out_scale=dd, bias=bd, alpha=0.9, act=N.ACT_LRELU)
        y = torch.full((B, M, Ho, Wo), float("nan"), device=dev)
        ops.conv2d_units_s2_raw(XP, pf, M, epi=epi(), out=y)
        y_nchw = ops.conv2d_raw(xd, pf, M, 3, 3, (Ho, Wo), (2, 2), (0, 0), in_scale=sd, epi=epi(), allow_split=False)
        pre = 0.9 * F.conv2d(xs.double().cpu(), w_ref.permute(3, 2, 0, 1), stride=2) * dd.double().cpu()[:, :, None, None]
        ref = F.leaky_relu(pre + bd.double().cpu()[None, :, None, None], 0.2) * math.sqrt(2.0)
        errs = [_rel(y, ref)]
        dns = [float((y - y_nchw).abs().max() / y_nchw.abs().max())]
        # data gradient of a transposed convolution M -> C (the up-convolution's backward): transposed + flipped pack of ITS filter
        wt = f(_rnd(3, 3, M, C, seed=9) / math.sqrt(9 * M))      # the up-convolution's filter [kh, kw, in = M, out = C]
        aux = f(_rnd(B, M, Ho, Wo, seed=8))
        pft = ops.pack_filter(wt, True, True)
        dot_u, dot_n = torch.empty(B, M, device=dev), torch.empty(B, M, device=dev)
        dx = ops.conv2d_units_s2_raw(XP, pft, M, epi=N.epilogue(out_scale=dd, alpha=0.7), dot=(aux, dot_u))
        dx_n = ops.conv2d_raw(xd, pft, M, 3, 3, (Ho, Wo), (2, 2), (0, 0), in_scale=sd, epi=N.epilogue(out_scale=dd, alpha=0.7),
                              dot=(aux, dot_n), allow_split=False)
        errs += [0.0, 0.0]
        dns += [float((dx - dx_n).abs().max() / dx_n.abs().max()), float((dot_u - dot_n).abs().max() / dot_n.abs().max())]
    print(f"\nCONVS2 {mode} {case}: vs float64 {errs[0]:.2e}   vs nchw kernel {dns[0]:.1e} {dns[1]:.1e} {dns[2]:.1e}")
    assert max(errs) < 3e-5, errs
    assert max(dns) < 5e-6, dns


def test_conv_units_s2_refuses_other_geometries(dev):
    import ctypes as C
    for d in (N.ConvDesc(2, 64, 64, 16, 64, 16, 64, 3, 3, 1, 1, 1, 1, 0, 0, 64, 1),     # stride 1
              N.ConvDesc(2, 64, 64, 17, 33, 8, 16, 3, 3, 2, 2, 0, 0, 0, 0, 64, 1),      # 16-pixel rows
              N.ConvDesc(2, 64, 72, 17, 65, 8, 32, 3, 3, 2, 2, 0, 0, 0, 0, 72, 1),      # partial channel tile
              N.ConvDesc(2, 64, 64, 17, 65, 8, 32, 3, 3, 2, 2, 0, 0, 0, 0, 64, 2)):     # split K
        assert N.lib().tbg_conv2d_units_s2_blocks(C.byref(d), 3) == -4  # TBG_EUNSUPPORTED


WG_S2 = [(2, 128, 64, 17, 65), (2, 128, 128, 18, 66), (3, 256, 64, 9, 65), (1, 128, 64, 66, 258), (4, 256, 128, 33, 129)]


@pytest.mark.parametrize("planes", [3, 1])
@pytest.mark.parametrize("case", WG_S2, ids=[str(c) for c in WG_S2])
def test_wgrad_units_s2_matches_float64(dev, planes, case):
    """tbg_conv2d_wgrad_units_s2 (8-wave 128 x 64 blocks, phase tiles by LDS-DMA, transposing operand reads) with both operands
    scaled and the fused additive term, against float64 on the operands the kernels see.  planes = 3: the fp32 bar of
    conv_wgrad_x3_kernel (3e-5) and not worse than 2x the exact fp32 kernel; planes = 1: float64 on the bf16-rounded operands."""
    import torch.nn.functional as F
    B, CS, CL, Hl, Wl = case
    Hs, Ws = (Hl - 3) // 2 + 1, (Wl - 3) // 2 + 1
    x, dy = _rnd(B, CL, Hl, Wl, seed=40), _rnd(B, CS, Hs, Ws, seed=41)
    xs, ds = _rnd(B, CL, seed=42).abs() + 0.5, _rnd(B, CS, seed=43).abs() + 0.5
    addw, addq = _rnd(3, 3, CL, CS, seed=44), _rnd(CL, CS, seed=45)
    f32 = lambda t: t.float().double()
    xr = (f32(x) * f32(xs)[:, :, None, None]).float()
    dyr = (f32(dy) * f32(ds)[:, :, None, None]).float()
    if planes == 1:
        xr, dyr = xr.bfloat16().float(), dyr.bfloat16().float()
    w = torch.zeros(3, 3, CL, CS, dtype=torch.float64, requires_grad=True)
    (ref,) = torch.autograd.grad(F.conv2d(xr.double(), w.permute(3, 2, 0, 1), stride=2), w, dyr.double())
    ref = 0.7 * ref + 0.3 * f32(addw) * f32(addq)[None, None]
    f = lambda t: t.float().to(dev).contiguous()
    assert ops.wgrad_units_s2_ok(CS, CL, Hs, Ws, Hl, Wl)
    SU, LP = ops.units_pack(f(dy), f(ds), planes=planes), ops.units_pack_s2(f(x), f(xs), planes=planes)
    dw = torch.full((3, 3, CL, CS), float("nan"), device=dev)
    ops.wgrad_units_s2_raw(SU, LP, dw, CL * CS, CS, 1, 0.7, add=(f(addw), f(addq), 0.3))
    err = _rel(dw, ref)
    if planes == 3:
        g = ops._Geom((2, 2), (0, 0), 3, 3, (Hl, Wl), (Hs, Ws))
        with ops.compute_dtype("f32"):
            dw32 = ops._bwd_weight_launch(f(x), f(dy), g, CL, CS, alpha=0.7, x_scale=f(xs), dy_scale=f(ds), add=(f(addw), f(addq), 0.3))
        e32 = _rel(dw32, ref)
        print(f"\nWGS2 {case}: f32 {e32:.3e}  units x3 {err:.3e}")
        assert err < 3e-5 and err <= max(2.0 * e32, 1e-6), (err, e32)
    else:
        print(f"\nWGS2 bf16 {case}: {err:.3e}")
        assert err < 3e-5, err


@pytest.mark.parametrize("planes", [3, 1])
@pytest.mark.parametrize("case", [(2, 20, 16, 64, (2, 3, 2, 3), False), (2, 64, 16, 64, (2, 2, 2, 2), True),
                                  (1, 128, 64, 256, (2, 3, 2, 3), False), (3, 16, 5, 9, (2, 2, 2, 2), True)], ids=str)
def test_fir_units_s2_equals_fir_then_pack(dev, planes, case):
    """the fused producer (tbg_upfirdn2d_units_s2_f32: blur -> phase unit tensor) against the two launches it replaces (to single fp32 roundings)
    -- the D blocks' blur (pad 2,3: 64x256 -> 66x258) and the up-convolution's backward blur (pad 2,2 with the demodulation scale:
    -> 65x257, whose odd size leaves the last row / column of two phases empty)."""
    B, C, H, W, pad, scaled = case
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, C, H, W, generator=g).to(dev)
    s = (torch.rand(B, C, generator=g) + 0.5).to(dev) if scaled else None
    k = ops.fir_kernel(dev, 4.0 if scaled else 1.0)
    t = ops.upfirdn2d_raw(x, k, pad=pad, in_scale=s.reshape(-1) if scaled else None)
    ref = ops.units_pack_s2(t, planes=planes)
    P = ops.upfirdn2d_units_s2(x, k, pad=pad, in_scale=s.reshape(-1) if scaled else None, planes=planes)
    assert (P.Hin, P.Win, P.Ho, P.Wo) == (ref.Hin, ref.Win, ref.Ho, ref.Wo)
    # the value behind every unit against the two-launch form: the same arithmetic (horizontal pass, vertical pass, scale) in
    # another kernel -- the compiler contracts a few multiply-adds differently, so single fp32 roundings differ (both forms sit at
    # the same 1e-7 of float64): planes = 3 reconstructs fp32 exactly, planes = 1 may land on the other side of a bf16 boundary
    Hq, Wq, C8 = P.Ho + 1, P.Wo + 1, (C + 7) // 8
    val = lambda U: U.data.float().double().reshape(planes, B, C8, 2, 2, Hq, Wq, 8).sum(0)
    got, exp = val(P), val(ref)
    assert torch.equal(got == 0, exp == 0), "padding / channel tail"
    # (a rounding of an intermediate is relative to the tensor's scale, not to a result that cancelled; one bf16 ulp is up to
    # 2^-7 of the value)
    bad = (got - exp).abs() > (0.0 if planes == 3 else 2.0 ** -7) * exp.abs() + 2.0 ** -21 * float(exp.abs().max())
    assert not bool(bad.any()), (int(bad.sum()), float((got - exp).abs().max()))
    assert float(((got != exp).double().mean())) < (0.5 if planes == 3 else 5e-3), "more than rounding-level cases differ"


@pytest.mark.parametrize("mode", ["f32x3", "bf16"])
def test_blur_conv_s2_fused_equals_the_nchw_layers(dev, mode, monkeypatch):
    """ops._BlurConvS2Fused (FIR -> phase units -> strided convolution; backward: units(dpre) -> filter gradient from unit tensors)
    against the two launches + NCHW backward it replaces (ops.upfirdn2d + ops.conv_bias_act_fused): output, input gradient,
    filter gradient and bias gradient at the kernels' own agreement (same products, other summation orders)."""
    monkeypatch.setattr(ops.TUNING, "units_min_blocks", 1)
    B, I, O, H, W = 2, 64, 128, 32, 128
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, I, H, W, generator=g).to(dev).requires_grad_(True)
    w = (torch.randn(3, 3, I, O, generator=g)).to(dev).requires_grad_(True)
    b = (0.1 * torch.randn(O, generator=g)).to(dev).requires_grad_(True)
    dout = torch.randn(B, O, H // 2, W // 2, generator=g).to(dev)
    with ops.compute_dtype(mode):
        assert ops.blur_conv_s2_units(B, I, O, H, W)
        k = ops.fir_kernel(dev, 1.0)
        y0 = ops.conv_bias_act_fused(ops.upfirdn2d(x, k, pad=(2, 3, 2, 3)), w, b, stride=(2, 2), out_mul=0.7)
        g0 = torch.autograd.grad(y0, (x, w, b), dout)
        for t2 in (False, True):  # the data gradient through the NCHW transposed kernel, then through tbg_conv2d_units_t2
            monkeypatch.setattr(ops.TUNING, "use_units_t2", t2)
            y1 = ops.blur_conv_s2_fused(x, w, b, out_mul=0.7)
            g1 = torch.autograd.grad(y1, (x, w, b), dout)
            if not t2:
                assert float((g1[0] - g0[0]).abs().max()) / float(g0[0].abs().max()) < (5e-6 if mode == "f32x3" else 3e-4)
    # bf16: the two FIR kernels differ by single fp32 roundings, which moves a few elements of the blurred tensor to the neighbouring
    # bf16 value (2^-8 relative on one of 576 products)
    tol = 5e-6 if mode == "f32x3" else 3e-4
    rel = lambda a, r: float((a - r).abs().max() / r.abs().max())
    errs = [rel(y1, y0)] + [rel(a, r) for a, r in zip(g1, g0)]
    print(f"\nBLURCONV {mode}: y {errs[0]:.1e} dx {errs[1]:.1e} dw {errs[2]:.1e} db {errs[3]:.1e}")
    assert max(errs) < tol, errs


@pytest.mark.parametrize("mode", ["f32x3", "bf16"])
def test_modconv_up_backward_through_phase_units_equals_the_nchw_path(dev, mode, monkeypatch):
    """ops._ModConvUpFused's backward with the blur^T output as a phase unit tensor (data gradient = tbg_conv2d_units_s2 with the
    fused style dot, filter gradient = tbg_conv2d_wgrad_units_s2 with the demodulation term) against its NCHW launches."""
    monkeypatch.setattr(ops.TUNING, "units_min_blocks", 1)
    B, I, O, H, W = 2, 128, 64, 16, 64
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, I, H, W, generator=g).to(dev).requires_grad_(True)
    w = torch.randn(3, 3, I, O, generator=g).to(dev).requires_grad_(True)
    s = (torch.rand(B, I, generator=g) + 0.5).to(dev).requires_grad_(True)
    noise = torch.randn(B, 1, 2 * H, 2 * W, generator=g).to(dev)
    strength = torch.tensor(0.2, device=dev, requires_grad=True)
    b = (0.1 * torch.randn(O, generator=g)).to(dev).requires_grad_(True)
    dout = torch.randn(B, O, 2 * H, 2 * W, generator=g).to(dev)
    res = []
    for on in (False, True):  # NCHW launches, then every stride-2 launch of the layer from unit tensors
        monkeypatch.setattr(ops.TUNING, "use_units_s2", on)
        monkeypatch.setattr(ops.TUNING, "use_units_t2", on)
        with ops.compute_dtype(mode):
            y = ops.modconv_up_fused(x, w, s, noise, strength, b)
            res.append((y,) + torch.autograd.grad(y, (x, w, s, strength, b), dout))
    rel = lambda a, r: float((a - r).abs().max() / (r.abs().max() + 1e-30))
    errs = [rel(a, r) for a, r in zip(res[1], res[0])]
    print(f"\nMODCONVUP {mode}: " + " ".join(f"{e:.1e}" for e in errs))
    tol = 5e-6 if mode == "f32x3" else 3e-4  # (bf16: see test_blur_conv_s2_fused_equals_the_nchw_layers)
    assert max(errs) < tol, errs


CONV_T2 = [(2, 64, 64, 8, 32, 1), (2, 128, 128, 16, 64, 2), (3, 64, 192, 5, 9, 1), (1, 256, 64, 32, 128, 2), (2, 64, 128, 33, 20, 1)]


@pytest.mark.parametrize("mode", ["f32x3", "bf16"])
@pytest.mark.parametrize("case", CONV_T2, ids=[str(c) for c in CONV_T2])
def test_conv_units_t2_matches_float64_and_the_nchw_kernel(dev, mode, case):
    """tbg_conv2d_units_t2 (merged-class transposed convolution over a FLAT tiling of the padded unit tensor) as the
    up-convolution's forward (flip, output 2H+1) and as the strided convolution's data gradient (transposed pack, output 2H+2):
    against float64 on the operands the kernels see and against the NCHW transposed kernels; every output element written."""
    import torch.nn.functional as F
    B, C, M, H, W, extra = case
    if mode == "bf16" and C % 16:
        pytest.skip("bf16 units need whole 16-channel chunks")
    planes = 3 if mode == "f32x3" else 1
    Hout, Wout = 2 * H + extra, 2 * W + extra
    x, w = _rnd(B, C, H, W, seed=1), _rnd(3, 3, C, M, seed=2) / math.sqrt(9 * C)
    s = _rnd(B, C, seed=3).abs() + 0.5
    f = lambda t: t.float().to(dev).contiguous()
    xd, wd, sd = f(x), f(w), f(s)
    xs = xd * sd[:, :, None, None]
    w_ref = wd.double().cpu()
    if mode == "bf16":
        xs, w_ref = xs.bfloat16().float(), wd.bfloat16().double().cpu()
    with ops.compute_dtype(mode):
        assert ops.conv_units_t2_ok(C, M, planes)
        XU = ops.units_pack(xd, sd, planes=planes)
        # (1) the up-convolution's forward: conv_transpose with flip(w) = correlation form; ops passes flip=True on the plain pack
        pf = ops.pack_filter(wd, False, False)
        y = torch.full((B, M, Hout, Wout), float("nan"), device=dev)
        ops.conv2d_units_t2_raw(XU, pf, M, (Hout, Wout), flip=True, alpha=0.8, out=y)
        y_n = ops.conv2d_raw(xd, pf, M, 3, 3, (Hout, Wout), (2, 2), (0, 0), transposed=True, flip=True, in_scale=sd,
                             epi=N.epilogue(alpha=0.8), allow_split=False)
        ref = 0.8 * F.conv_transpose2d(xs.double().cpu(), torch.flip(w_ref, (0, 1)).permute(2, 3, 0, 1), stride=2)
        refp = torch.zeros(B, M, Hout, Wout, dtype=torch.float64)
        refp[:, :, :2 * H + 1, :2 * W + 1] = ref
        assert torch.isfinite(y).all()
        e1, d1 = _rel(y, refp), float((y - y_n).abs().max() / y_n.abs().max())
        # (2) the data gradient of a strided convolution M' = C -> its input channels: transposed pack, no flip
        wt = f(_rnd(3, 3, M, C, seed=9) / math.sqrt(9 * M))  # the strided convolution's filter [kh, kw, in = M, out = C]
        pft = ops.pack_filter(wt, True, False)
        g = ops.conv2d_units_t2_raw(XU, pft, M, (Hout, Wout), flip=False, alpha=1.0)
        g_n = ops.conv2d_raw(xd, pft, M, 3, 3, (Hout, Wout), (2, 2), (0, 0), transposed=True, in_scale=sd, allow_split=False)
        d2 = float((g - g_n).abs().max() / g_n.abs().max())
    print(f"\nCONVT2 {mode} {case}: vs float64 {e1:.2e}   vs nchw kernel {d1:.1e} {d2:.1e}")
    assert e1 < 3e-5 and max(d1, d2) < 5e-6, (e1, d1, d2)
