"""-m gpu: convolution operands in the 8-channel-unit layout (tbg.h "unit tensors", csrc/conv_units.hip): the stand-alone
producer against its definition, and the kernels that consume unit tensors against float64 at the tolerances of the NCHW
kernels they replace (tests/test_x3_gpu.py for f32x3, tests/test_bf16_gpu.py for bf16)."""
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


def _planes(U):
    """[planes, B, C8, H+2, W+2, 8] float64 view of a unit tensor"""
    return U.data.float().double().reshape(U.planes, U.B, (U.C + 7) // 8, U.H + 2, U.W + 2, 8)


@pytest.mark.parametrize("planes", [3, 1])
@pytest.mark.parametrize("shape", [(2, 20, 5, 9), (3, 64, 8, 32), (1, 130, 2, 33)])
@pytest.mark.parametrize("scaled", [True, False])
def test_units_pack_is_the_definition(dev, planes, shape, scaled):
    """U[pl][b][c/8][1+y][1+x][c%8]: planes = 3 sums EXACTLY to the fp32 product x * s (hi = its RNE bf16), planes = 1 is
    its RNE bf16; ring and channel tail are zero; every unit is written (the buffer starts as NaN)."""
    B, C, H, W = shape
    g = torch.Generator().manual_seed(7)
    x = (torch.randn(B, C, H, W, generator=g) * torch.exp(3 * torch.randn(B, C, H, W, generator=g))).to(dev)
    s = (torch.rand(B, C, generator=g) + 0.5).to(dev) if scaled else None
    U = ops.units_pack(x, s, planes=planes)
    assert U.data.numel() * 2 == N.lib().tbg_units_bytes(B, C, H, W, planes)
    P = _planes(U)
    assert torch.isfinite(P).all()
    v = (x * s[:, :, None, None]) if scaled else x  # the fp32 product the kernels form
    C8 = (C + 7) // 8
    exp = torch.zeros(B, C8 * 8, H + 2, W + 2, device=dev, dtype=torch.float32)
    exp[:, :C, 1:-1, 1:-1] = v
    exp = exp.reshape(B, C8, 8, H + 2, W + 2).permute(0, 1, 3, 4, 2)
    if planes == 3:
        assert torch.equal(P.sum(0), exp.double()), float((P.sum(0) - exp.double()).abs().max())
        assert torch.equal(P[0], exp.bfloat16().double())
    else:
        assert torch.equal(P[0], exp.bfloat16().double())


WG_UNITS = [(2, 64, 64, 2, 32), (2, 64, 128, 8, 32), (3, 128, 64, 6, 64), (4, 128, 128, 16, 64), (1, 64, 64, 64, 256)]


@pytest.mark.parametrize("planes", [3, 1])
@pytest.mark.parametrize("case", WG_UNITS, ids=[str(c) for c in WG_UNITS])
def test_wgrad_units_matches_float64(dev, planes, case):
    """tbg_conv2d_wgrad_units (LDS-DMA staged, transposing operand reads) with both operands scaled and the fused additive
    term, against float64 on the operands the kernels see.  planes = 3: the fp32 bar of conv_wgrad_x3_kernel (3e-5) and not
    worse than 2x the exact fp32 kernel; planes = 1: float64 on the bf16-rounded operands (products exact: 3e-5)."""
    import torch.nn.functional as F
    B, C, M, H, W = case
    x, dy = _rnd(B, C, H, W, seed=40), _rnd(B, M, H, W, seed=41)
    xs, ds = _rnd(B, C, seed=42).abs() + 0.5, _rnd(B, M, seed=43).abs() + 0.5
    addw, addq = _rnd(3, 3, C, M, seed=44), _rnd(C, M, seed=45)
    f32 = lambda t: t.float().double()
    xr = (f32(x) * f32(xs)[:, :, None, None]).float()
    dyr = (f32(dy) * f32(ds)[:, :, None, None]).float()
    if planes == 1:
        xr, dyr = xr.bfloat16().float(), dyr.bfloat16().float()
    w = torch.zeros(3, 3, C, M, dtype=torch.float64, requires_grad=True)
    (ref,) = torch.autograd.grad(F.conv2d(xr.double(), w.permute(3, 2, 0, 1), padding=1), w, dyr.double())
    ref = 0.7 * ref + 0.3 * f32(addw) * f32(addq)[None, None]
    f = lambda t: t.float().to(dev).contiguous()
    assert ops.wgrad_units_ok(M, C, H, W, H, W, 3, 3, (1, 1), (1, 1))
    SU, LU = ops.units_pack(f(dy), f(ds), planes=planes), ops.units_pack(f(x), f(xs), planes=planes)
    dw = torch.full((3, 3, C, M), float("nan"), device=dev)
    ops.wgrad_units_raw(SU, LU, dw, C * M, M, 1, 0.7, add=(f(addw), f(addq), 0.3))
    err = _rel(dw, ref)
    if planes == 3:
        g = ops._Geom((1, 1), (1, 1), 3, 3, (H, W), (H, W))
        with ops.compute_dtype("f32"):
            dw32 = ops._bwd_weight_launch(f(x), f(dy), g, C, M, alpha=0.7, x_scale=f(xs), dy_scale=f(ds), add=(f(addw), f(addq), 0.3))
        e32 = _rel(dw32, ref)
        print(f"\nWGUNITS {case}: f32 {e32:.3e}  units x3 {err:.3e}")
        assert err < 3e-5 and err <= max(2.0 * e32, 1e-6), (err, e32)
    else:
        print(f"\nWGUNITS bf16 {case}: {err:.3e}")
        assert err < 3e-5, err


def test_wgrad_units_refuses_other_geometries(dev):
    import ctypes as C
    for d in (N.WgradDesc(2, 64, 64, 4, 16, 4, 16, 3, 3, 1, 1, 1, 1, 64 * 64, 64, 1, 1.0),      # 16-pixel rows
              N.WgradDesc(2, 64, 72, 8, 32, 8, 32, 3, 3, 1, 1, 1, 1, 72 * 64, 64, 1, 1.0),      # partial channel tile
              N.WgradDesc(2, 64, 64, 8, 32, 17, 65, 3, 3, 2, 2, 0, 0, 64 * 64, 64, 1, 1.0)):    # strided
        assert N.lib().tbg_conv2d_wgrad_units_workspace_bytes(C.byref(d)) == -4  # TBG_EUNSUPPORTED


CONV_UNITS = [(2, 64, 64, 8, 32), (2, 128, 64, 16, 64), (3, 128, 192, 8, 96), (1, 256, 256, 16, 64), (2, 64, 64, 64, 256)]


@pytest.mark.parametrize("mode", ["f32x3", "bf16"])
@pytest.mark.parametrize("case", CONV_UNITS, ids=[str(c) for c in CONV_UNITS])
def test_conv_units_matches_float64_and_the_nchw_kernel(dev, mode, case):
    """tbg_conv2d_units (all-DMA staging from the unit tensor, double-buffered tiles, 512-thread blocks) with the full fused
    epilogue (demodulation, noise, bias, LeakyReLU) and, as a data gradient, with flip + out_scale + the fused dot product:
    against float64 on the operands the kernels see at the NCHW kernels' bar, and against the NCHW kernel of the same
    arithmetic (same products, same accumulation order)."""
    import torch.nn.functional as F
    B, C, M, H, W = case
    if mode == "bf16" and C % 16:
        pytest.skip("bf16 units need whole 16-channel chunks")
    planes = 3 if mode == "f32x3" else 1
    x, w = _rnd(B, C, H, W, seed=1), _rnd(3, 3, C, M, seed=2) / math.sqrt(9 * C)
    s, dmod = _rnd(B, C, seed=3).abs() + 0.5, _rnd(B, M, seed=4).abs() + 0.5
    noise, bias = _rnd(B, 1, H, W, seed=5), _rnd(M, seed=6) * 0.2
    f = lambda t: t.float().to(dev).contiguous()
    xd, wd, sd, dd, nd, bd = f(x), f(w), f(s), f(dmod), f(noise), f(bias)
    strength = torch.tensor(0.3, device=dev)
    xs = (xd * sd[:, :, None, None])
    w_ref = wd.double().cpu()
    if mode == "bf16":
        xs, w_ref = xs.bfloat16().float(), wd.bfloat16().double().cpu()
    with ops.compute_dtype(mode):
        assert ops.conv_units_ok(C, M, H, W, 3, 3, (1, 1), (1, 1), False, planes)
        XU = ops.units_pack(xd, sd, planes=planes)
        pf = ops.pack_filter(wd, False, False)
        epi = lambda: N.epilogue(out_scale=dd, bias=bd, noise=nd, strength=strength, alpha=0.9, act=N.ACT_LRELU)
        y = ops.conv2d_units_raw(XU, pf, M, epi=epi())
        y_nchw = ops.conv2d_raw(xd, pf, M, 3, 3, (H, W), (1, 1), (1, 1), in_scale=sd, epi=epi(), allow_split=False)
        pre = 0.9 * F.conv2d(xs.double().cpu(), w_ref.permute(3, 2, 0, 1), padding=1) * dd.double().cpu()[:, :, None, None]
        pre = pre + nd.double().cpu() * 0.3 + bd.double().cpu()[None, :, None, None]
        ref = F.leaky_relu(pre, 0.2) * math.sqrt(2.0)
        err, dn = _rel(y, ref), float((y - y_nchw).abs().max() / y_nchw.abs().max())
        # data-gradient form: transposed + flipped pack, per-(b, channel) output scale, fused dot with a second tensor
        dy, aux = f(_rnd(B, M, H, W, seed=7)), f(_rnd(B, C, H, W, seed=8))
        DU = ops.units_pack(dy, dd, planes=planes)
        pft = ops.pack_filter(wd, True, True)
        dot_u, dot_n = torch.empty(B, C, device=dev), torch.empty(B, C, device=dev)
        dx = ops.conv2d_units_raw(DU, pft, C, epi=N.epilogue(out_scale=sd), dot=(aux, dot_u))
        dx_n = ops.conv2d_raw(dy, pft, C, 3, 3, (H, W), (1, 1), (1, 1), in_scale=dd, epi=N.epilogue(out_scale=sd), dot=(aux, dot_n),
                              allow_split=False)
        dys = dy * dd[:, :, None, None]
        if mode == "bf16":
            dys = dys.bfloat16().float()
        gref = F.conv_transpose2d(dys.double().cpu(), w_ref.permute(3, 2, 0, 1), padding=1)
        dref = (gref * aux.double().cpu()).sum((2, 3))
        gref = gref * sd.double().cpu()[:, :, None, None]
        errs = (err, _rel(dx, gref), _rel(dot_u, dref))
        dns = (dn, float((dx - dx_n).abs().max() / dx_n.abs().max()), float((dot_u - dot_n).abs().max() / dot_n.abs().max()))
    print(f"\nCONVUNITS {mode} {case}: vs float64 {errs[0]:.2e} {errs[1]:.2e} {errs[2]:.2e}   vs nchw kernel {dns[0]:.1e} {dns[1]:.1e} {dns[2]:.1e}")
    assert max(errs) < 3e-5, errs
    assert max(dns) < 2e-6, dns


@pytest.mark.parametrize("planes", [3, 1])
@pytest.mark.parametrize("shape", [(2, 20, 5, 9), (2, 64, 16, 64), (1, 128, 64, 256)])
def test_bias_act_bwd_units_equals_bias_act_bwd_then_pack(dev, planes, shape):
    """the fused producer (tbg_bias_act_bwd_units) against the two launches it replaces: the unit tensor bit for bit, dpre bit
    for bit, and the partial sums (different chunking) to fp32 summation order."""
    B, M, H, W = shape
    g = torch.Generator().manual_seed(11)
    out = torch.randn(B, M, H, W, generator=g).to(dev)
    dout = torch.randn(B, M, H, W, generator=g).to(dev)
    d, noise = (torch.rand(B, M, generator=g) + 0.5).to(dev), torch.randn(B, 1, H, W, generator=g).to(dev)
    bias, strength = torch.randn(M, generator=g).to(dev), torch.tensor(0.3, device=dev)
    epi = lambda: N.epilogue(out_scale=d, bias=bias, noise=noise, strength=strength, act=N.ACT_LRELU)
    dx_ref, dpre_ref, pdb, pdn, pdy = ops.bias_act_bwd_raw(dout, out, epi(), want_dx=True, want_dn=True, want_dyy=True)
    U_ref = ops.units_pack(dpre_ref, d, planes=planes)
    DU, dpre, qdb, qdn, qdy = ops.bias_act_bwd_units_raw(dout, out, epi(), planes=planes, want_dpre=True, want_dn=True, want_dyy=True)
    assert torch.equal(dpre, dpre_ref)
    assert torch.equal(DU.data.view(torch.int16), U_ref.data.view(torch.int16)), "unit tensor differs from pack(dpre * d)"
    for a, r in ((qdb, pdb), (qdn, pdn), (qdy, pdy)):
        a, r = a.sum(2).double().cpu(), r.sum(2).double().cpu()
        assert float((a - r).abs().max()) <= 2e-5 * float(r.abs().max() + 1e-30)
    DU2, none, _, _, _ = ops.bias_act_bwd_units_raw(dout, out, epi(), planes=planes)
    assert none is None and torch.equal(DU2.data.view(torch.int16), U_ref.data.view(torch.int16))


# ----------------------------------------------------------------------------------------------------------------
# UNIT SINKS (tbg.h, round 5): a producer launch writes units(out * scale) itself -- the bits of tbg_units_pack_f32(out, scale)
# ----------------------------------------------------------------------------------------------------------------
class _AlwaysSink(ops.UnitSink):
    """a sink that is wanted whatever the consumer's geometry (the unit tests exercise the producers, not the dispatch rule)"""

    def wanted(self, B, Cc, H, W):
        return True


def _check_sink(y, U, y_ref, scale, planes, what):
    assert U is not None and U.planes == planes, what
    assert torch.equal(y, y_ref), (what, "the fp32 output must not change with a sink", float((y - y_ref).abs().max()))
    ref = ops.units_pack(y_ref, scale, planes=planes)
    assert U.data.numel() == ref.data.numel(), what
    same = torch.equal(U.data.view(torch.int16), ref.data.view(torch.int16))
    if not same:
        a, b = _planes(U), _planes(ref)
        bad = (a != b).nonzero()
        raise AssertionError((what, "unit tensor differs from units_pack", bad.shape[0], bad[:5].tolist()))


SINK_CONVS = [  # (B, C, M, H, W, k, stride, pad, transposed, residual)
    (2, 64, 128, 16, 64, 3, (1, 1), (1, 1), False, False),    # 128x128 tiles, one image per tile
    (3, 32, 64, 8, 32, 3, (1, 1), (1, 1), False, False),      # 64-channel tile
    (5, 128, 128, 4, 16, 3, (1, 1), (1, 1), False, False),    # several images per tile (ring from every image's borders)
    (4, 64, 64, 2, 8, 1, (1, 1), (0, 0), False, True),        # 1x1 + residual (the discriminator block's output launch)
    (2, 8, 64, 16, 64, 1, (1, 1), (0, 0), False, False),      # thin input (fromRGB-like)
    (2, 64, 64, 9, 33, 3, (2, 2), (0, 0), False, False),      # strided VALID: 4 x 16 output
    (16, 512, 512, 4, 16, 3, (1, 1), (1, 1), False, False),   # split-K: the sink rides on slab_epilogue_units
]


@pytest.mark.parametrize("arith", ["f32x3", "bf16"])
@pytest.mark.parametrize("case", SINK_CONVS, ids=[str(c[:9]) for c in SINK_CONVS])
def test_conv_unit_sink_equals_units_pack(dev, arith, case):
    """tbg_conv2d_{x3,bf16} with tbg_epilogue.units_out (conv_epilogue's sink; tbg_slab_epilogue_units_f32 for split K): the unit
    tensor -- interior, ring of zero units, every plane -- is bit for bit tbg_units_pack_f32 of the launch's own fp32 output times
    the scale, and the fp32 output is bit for bit that of the launch without a sink."""
    B, C, M, H, W, k, stride, pad, transposed, has_res = case
    planes = 3 if arith == "f32x3" else 1
    f = lambda t: t.float().to(dev).contiguous()
    x, w = f(_rnd(B, C, H, W, seed=60)), f(_rnd(k, k, C, M, seed=61) / math.sqrt(k * k * C))
    Ho, Wo = (H + 2 * pad[0] - k) // stride[0] + 1, (W + 2 * pad[1] - k) // stride[1] + 1
    res = f(_rnd(B, M, Ho, Wo, seed=62)) if has_res else None
    d, bias, scale = f(_rnd(B, M, seed=63).abs() + 0.5), f(_rnd(M, seed=64)), f(_rnd(B, M, seed=65))
    with ops.compute_dtype(arith):
        mk = lambda: (N.epilogue(alpha=0.5, bias=bias, residual=res, res_scale=0.7) if has_res else
                      ops._lrelu_epi(out_scale=d, bias=bias, alpha=0.5))
        pf = ops.pack_filter(w, False, False)
        y_ref = ops.conv2d_raw(x, pf, M, k, k, (Ho, Wo), stride, pad, epi=mk())
        for sc in (scale, None):
            y, U = ops.conv2d_raw(x, pf, M, k, k, (Ho, Wo), stride, pad, epi=mk(), sink=_AlwaysSink(sc, "s1", M))
            _check_sink(y, U, y_ref, sc, planes, ("conv2d_raw", case, sc is not None))


@pytest.mark.parametrize("arith", ["f32x3", "bf16"])
def test_conv_units_and_fir_unit_sinks_equal_units_pack(dev, arith):
    """the same for tbg_conv2d_units (64- and 128-channel tiles), tbg_conv2d_units_s2 and the blur's sink form
    (tbg_upfirdn2d_sep_f32 with units_out: fir_units_kernel), which must also reproduce upfirdn2d_tile_kernel's fp32 output."""
    planes = 3 if arith == "f32x3" else 1
    f = lambda t: t.float().to(dev).contiguous()
    with ops.compute_dtype(arith):
        for (B, C, M, H, W) in ((2, 64, 128, 16, 64), (1, 32, 64, 8, 32), (3, 64, 256, 8, 64)):
            x, w = f(_rnd(B, C, H, W, seed=70)), f(_rnd(3, 3, C, M, seed=71) / math.sqrt(9 * C))
            s, d, bias, nz, scale = (f(_rnd(B, C, seed=72)), f(_rnd(B, M, seed=73).abs() + 0.5), f(_rnd(M, seed=74)),
                                     f(_rnd(B, 1, H, W, seed=75)), f(_rnd(B, M, seed=76)))
            strength = f(torch.tensor(0.3, dtype=torch.float64))
            XU = ops.units_pack(x, s)
            pf = ops.pack_filter(w, False, False)
            mk = lambda: ops._lrelu_epi(out_scale=d, bias=bias, noise=nz, strength=strength, alpha=0.25)
            y_ref = ops.conv2d_units_raw(XU, pf, M, epi=mk())
            y, U = ops.conv2d_units_raw(XU, pf, M, epi=mk(), sink=_AlwaysSink(scale, "up", M))
            _check_sink(y, U, y_ref, scale, planes, ("conv2d_units_raw", B, C, M, H, W))
        # stride-2 form: 17 x 65 -> 8 x 32
        B, C, M = 2, 64, 128
        t, w = f(_rnd(B, C, 17, 65, seed=80)), f(_rnd(3, 3, C, M, seed=81) / math.sqrt(9 * C))
        bias, scale = f(_rnd(M, seed=82)), f(_rnd(B, M, seed=83))
        TP = ops.units_pack_s2(t)
        pf = ops.pack_filter(w, False, False)
        mk = lambda: N.epilogue(alpha=0.3, bias=bias, act=N.ACT_LRELU, gain=1.1)
        y_ref = ops.conv2d_units_s2_raw(TP, pf, M, epi=mk())
        y, U = ops.conv2d_units_s2_raw(TP, pf, M, epi=mk(), sink=_AlwaysSink(None, "s1", M))
        _check_sink(y, U, y_ref, None, planes, "conv2d_units_s2_raw")
        # the blur behind the up-convolution (pad 1, gain-4 [1,3,3,1] x [1,3,3,1]) with the full epilogue, ragged and whole tiles
        for (B, C, H, W) in ((2, 16, 8, 32), (1, 64, 32, 128), (3, 8, 5, 19)):
            yu = f(_rnd(B, C, 2 * H + 1, 2 * W + 1, seed=90))
            d, bias, nz, scale = f(_rnd(B, C, seed=91).abs() + 0.5), f(_rnd(C, seed=92)), f(_rnd(B, 1, 2 * H, 2 * W, seed=93)), f(_rnd(B, C, seed=94))
            strength = f(torch.tensor(0.2, dtype=torch.float64))
            k = ops.fir_kernel(dev, gain=4.0)
            mk = lambda: ops._lrelu_epi(out_scale=d.reshape(-1), bias=bias, noise=nz, strength=strength, alpha=1.0)
            y_tile = ops.upfirdn2d_raw(yu, k, pad=(1, 1, 1, 1), epi=mk())
            y, U = ops.upfirdn2d_raw(yu, k, pad=(1, 1, 1, 1), epi=mk(), sink=_AlwaysSink(scale, "s1", C))
            # against the tile kernel: the same expression, but the compiler may contract multiply-adds differently (<= 2 ulp)
            assert float((y - y_tile).abs().max()) <= 4e-7 * float(y_tile.abs().max()), ("fir sink vs tile kernel", B, C, H, W)
            _check_sink(y, U, y, scale, planes, ("fir_units_kernel", B, C, H, W))
