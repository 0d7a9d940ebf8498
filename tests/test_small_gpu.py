"""-m gpu: tbg_conv2d_units_small (csrc/conv_small.hip, tbg.h "SMALL MAPS") -- the small-map convolutions with the K split inside
the block -- against float64 on the operands the kernel sees (f32x3: the fp32 values; bf16: their RNE roundings), at the bars of
the kernels it replaces (tests/test_units_gpu.py), over the geometries a training step launches (the recogniser trunk's 1x25 ...
8x25 maps, the networks' 4x16 ... 8x32 layers) and the edge cases of its tiling: pixel tiles that span rows and samples, a ragged
last tile, channel tails in M and C, both tile widths, the strided and transposed-strided 1x1 forms, every epilogue term, the
fused dot product and the unit sink (bit for bit tbg_units_pack_f32 of the launch's own output)."""
import math

import pytest
import torch
import torch.nn.functional as F

from textboxgan_amd import native as N, ops

pytestmark = pytest.mark.gpu


def _rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64)


def _rel(a, r):
    return float((a.double().cpu() - r).abs().max() / (r.abs().max() + 1e-30))


def _pad(k, stride, transposed):
    return 1 if (k == 3 and tuple(stride) == (1, 1) and not transposed) else 0


def _ref_conv(xs, w, k, stride, transposed, out_hw):
    """float64 definition (tbg.h): xs [B,C,H,W], w [k,k,C,M] HWIO"""
    if not transposed:
        return F.conv2d(xs, w.permute(3, 2, 0, 1), stride=stride, padding=_pad(k, stride, transposed))
    if k > 1:  # y[b, m, sy a + kh, sx b' + kw] += x[b, c, a, b'] W[kh, kw, c, m]
        y = F.conv_transpose2d(xs, w.permute(2, 3, 0, 1), stride=stride)
        return F.pad(y, (0, out_hw[1] - y.shape[3], 0, out_hw[0] - y.shape[2]))
    B, _, H, W = xs.shape
    y = torch.zeros(B, w.shape[3], *out_hw, dtype=torch.float64)
    y[:, :, :(H - 1) * stride[0] + 1:stride[0], :(W - 1) * stride[1] + 1:stride[1]] = F.conv2d(xs, w.permute(3, 2, 0, 1))
    return y


# (B, C, M, Hin, Win, k, stride, transposed)
GEOMS = [
    (16, 256, 256, 2, 25, 3, (1, 1), False),   # recogniser stage 4: 25 tiles of 32 flattened pixels, rows and samples inside a tile
    (16, 512, 512, 1, 25, 3, (1, 1), False),   # one-row maps, 24 row units per wave
    (16, 128, 128, 4, 25, 3, (1, 1), False),
    (3, 64, 40, 8, 25, 3, (1, 1), False),      # M tail (40 = 32 + 8), ragged last pixel tile (600 = 18 x 32 + 24)
    (2, 20, 64, 5, 9, 3, (1, 1), False),       # C tail (20 channels = 2 1/2 units; bf16: skipped, odd unit count)
    (16, 512, 512, 4, 16, 3, (1, 1), False),   # generator 4x16: 512 blocks of 32 pixels, two tiles per sample
    (32, 256, 256, 8, 16, 3, (1, 1), False),   # discriminator 8x16 on the joint batch: TN = 2, two tiles per sample
    (5, 48, 96, 3, 3, 3, (1, 1), False),       # 3-wide rows: eleven segments in a tile
    (16, 256, 256, 2, 25, 1, (1, 1), False),   # 1x1
    (16, 128, 256, 4, 25, 1, (2, 1), False),   # 1x1 stride (2, 1): 4x25 -> 2x25 (first unit of a recogniser stage)
    (4, 64, 128, 8, 25, 1, (2, 2), False),     # 1x1 stride (2, 2): 8x25 -> 4x13
    (16, 256, 128, 2, 25, 1, (2, 1), True),    # its data gradient: 2x25 -> 4x25, odd rows = epilogue(0)
    (4, 128, 64, 4, 13, 1, (2, 2), True),      # 4x13 -> 8x25
    (32, 64, 64, 16, 50, 1, (1, 1), False),    # many pixels, few channels: TN = 2, half the waves without a row unit
    # tap-list forms of k x k layers
    (16, 256, 512, 9, 33, 3, (2, 2), False),   # the blur's strided VALID convolution: 9x33 -> 4x16 (conv_downsample_2d)
    (32, 512, 512, 6, 10, 3, (1, 2), False),   # width-only stride (the discriminator's last block): 6x10 -> 4x4
    (3, 40, 72, 7, 12, 3, (2, 1), False),      # height-only stride, channel tails
    (16, 512, 256, 4, 16, 3, (2, 2), True),    # stride-2 transposed (upsample_conv_2d / the strided layers' data gradient): 4x16 -> 9x33
    (32, 512, 512, 4, 4, 3, (1, 2), True),     # 4x4 -> 6x10: a full correlation along y (sources two rows outside the map: zero units)
    (2, 24, 40, 3, 5, 3, (2, 2), True),        # tiny, ragged classes (7x11 outputs)
    (8, 64, 64, 4, 4, 3, (1, 2), True, (6, 10)),  # an output wider than the taps reach (the forward layer floored): zeros there
    (4, 64, 64, 5, 9, 2, (2, 2), False),       # 2x2 stride 2
]


def _out_hw(H, W, k, stride, transposed):
    if transposed and k == 1:  # the data gradient of a strided 1x1 layer whose input had an even height and an odd width (4x25, 8x25)
        return (H * stride[0], (W - 1) * stride[1] + 1)
    if transposed:
        return ((H - 1) * stride[0] + k, (W - 1) * stride[1] + k)
    p = _pad(k, stride, transposed)
    return ((H + 2 * p - k) // stride[0] + 1, (W + 2 * p - k) // stride[1] + 1)


@pytest.mark.parametrize("mode", ["f32x3", "bf16"])
@pytest.mark.parametrize("geom", GEOMS, ids=[str(g) for g in GEOMS])
def test_conv_small_matches_float64(dev, mode, geom):
    """plain (alpha only), the modulated layer's epilogue (demodulation, noise, bias, LeakyReLU), the ResNet unit's (bias, residual
    before the ReLU, gate) and the flipped form, all against float64; repeated launches are bit-identical (fixed summation order)."""
    B, C, M, H, W, k, stride, transposed = geom[:8]
    planes = 3 if mode == "f32x3" else 1
    if planes == 1 and ((C + 7) // 8) % 2:
        pytest.skip("bf16 chunks are two channel units")
    Ho, Wo = geom[8] if len(geom) > 8 else _out_hw(H, W, k, stride, transposed)
    f = lambda t: t.float().to(dev).contiguous()
    x, w, s = _rnd(B, C, H, W, seed=1), _rnd(k, k, C, M, seed=2) / math.sqrt(k * k * C), _rnd(B, C, seed=3).abs() + 0.5
    dmod, noise, bias = _rnd(B, M, seed=4).abs() + 0.5, _rnd(B, 1, Ho, Wo, seed=5), _rnd(M, seed=6) * 0.2
    res, gate = _rnd(B, M, Ho, Wo, seed=7), _rnd(B, M, Ho, Wo, seed=8)
    xd, wd, sd, dd, nd, bd, rd, gd = map(f, (x, w, s, dmod, noise, bias, res, gate))
    strength = torch.tensor(0.3, device=dev)
    xs = xd * sd[:, :, None, None]
    w_ref = wd.double().cpu()
    if mode == "bf16":
        xs, w_ref = xs.bfloat16().float(), wd.bfloat16().double().cpu()
    xs = xs.double().cpu()
    c64 = lambda t: t.double().cpu()
    with ops.compute_dtype(mode):
        pd = _pad(k, stride, transposed)
        assert ops.conv_small_ok(C, M, H, W, Ho, Wo, k, k, stride, (pd, pd), transposed, planes)
        XU = ops.units_pack(xd, sd, planes=planes)
        pf = ops.pack_filter(wd, False, False)
        run = lambda **kw: ops.conv2d_small_raw(XU, pf, M, k, (Ho, Wo), stride, transposed, **kw)
        acc = _ref_conv(xs, w_ref, k, stride, transposed, (Ho, Wo))
        errs = {}
        y0 = run(epi=N.epilogue(alpha=0.9))
        errs["plain"] = _rel(y0, 0.9 * acc)
        assert torch.equal(y0, run(epi=N.epilogue(alpha=0.9))), "repeated launches differ"
        y1 = run(epi=N.epilogue(out_scale=dd, bias=bd, noise=nd, strength=strength, alpha=0.9, act=N.ACT_LRELU, slope=0.2,
                                gain=math.sqrt(2.0)))
        pre = 0.9 * acc * c64(dd)[:, :, None, None] + c64(nd) * 0.3 + c64(bd)[None, :, None, None]
        errs["modconv"] = _rel(y1, F.leaky_relu(pre, 0.2) * math.sqrt(2.0))
        y2 = run(epi=N.epilogue(bias=bd, residual=rd, res_first=1, act=N.ACT_LRELU, slope=0.0, gain=1.0, gate=gd))
        ref2 = torch.relu(acc + c64(bd)[None, :, None, None] + c64(rd)) * (c64(gd) > 0)
        errs["resunit"] = _rel(y2, ref2)
        y3 = run(epi=N.epilogue(bias=bd, residual=rd, res_scale=0.7))
        errs["resafter"] = _rel(y3, (acc + c64(bd)[None, :, None, None] + c64(rd)) * 0.7)
        if k > 1:
            yf = ops.conv2d_small_raw(XU, ops.pack_filter(wd, False, True), M, k, (Ho, Wo), stride, transposed, flip=True,
                                      epi=N.epilogue(alpha=0.9))
            assert torch.equal(yf, y0), "flip of a flipped pack"
    print(f"\nCONVSMALL {mode} {geom}: " + "  ".join(f"{k_} {v:.2e}" for k_, v in errs.items()))
    assert max(errs.values()) < 3e-5, errs


DOTS = [(16, 512, 512, 4, 16), (4, 128, 256, 8, 32), (6, 64, 64, 4, 8), (32, 256, 256, 8, 16)]  # (the last: 64-pixel tiles)


@pytest.mark.parametrize("mode", ["f32x3", "bf16"])
@pytest.mark.parametrize("geom", DOTS, ids=[str(g) for g in DOTS])
def test_conv_small_data_gradient_with_fused_dot(dev, mode, geom):
    """the modulated layer's data gradient (modulated_conv2d.py:94-96 backward): transposed + flipped pack, out_scale = s, and the
    per-(b, channel) dot product of the UNSCALED accumulator with a second tensor from the same launch (the style gradient)."""
    B, C, M, H, W = geom
    planes = 3 if mode == "f32x3" else 1
    f = lambda t: t.float().to(dev).contiguous()
    w = f(_rnd(3, 3, C, M, seed=2) / math.sqrt(9 * C))
    dy, aux, sd, dd = f(_rnd(B, M, H, W, seed=7)), f(_rnd(B, C, H, W, seed=8)), f(_rnd(B, C, seed=3).abs() + 0.5), f(_rnd(B, M, seed=4).abs() + 0.5)
    with ops.compute_dtype(mode):
        DU = ops.units_pack(dy, dd, planes=planes)
        pft = ops.pack_filter(w, True, True)
        dot = torch.full((B, C), float("nan"), device=dev)
        dx = ops.conv2d_small_raw(DU, pft, C, 3, (H, W), epi=N.epilogue(out_scale=sd, alpha=0.5), dot=(aux, dot))
        dys, w_ref = dy * dd[:, :, None, None], w.double().cpu()
        if mode == "bf16":
            dys, w_ref = dys.bfloat16().float(), w.bfloat16().double().cpu()
        g = 0.5 * F.conv_transpose2d(dys.double().cpu(), w_ref.permute(3, 2, 0, 1), padding=1)
        e1, e2 = _rel(dx, g * sd.double().cpu()[:, :, None, None]), _rel(dot, (g * aux.double().cpu()).sum((2, 3)))
    print(f"\nCONVSMALL dot {mode} {geom}: dx {e1:.2e} dot {e2:.2e}")
    assert e1 < 3e-5 and e2 < 3e-5


def test_conv_small_dot_needs_whole_tiles(dev):
    import ctypes as C
    d = N.ConvDesc(16, 256, 256, 2, 25, 2, 25, 3, 3, 1, 1, 1, 1, 0, 0, 256, 1)
    assert N.lib().tbg_conv2d_units_small_dot_slots(C.byref(d), 3) == 0  # 50-pixel maps: a 32-pixel tile straddles two samples
    for bad in (N.ConvDesc(2, 64, 64, 8, 32, 4, 16, 3, 3, 2, 2, 0, 0, 0, 0, 64, 1),     # strided 3x3
                N.ConvDesc(2, 64, 64, 8, 32, 8, 32, 3, 3, 1, 1, 1, 1, 0, 0, 64, 2),     # split K belongs to the NCHW entry
                N.ConvDesc(2, 64, 64, 8, 2, 8, 2, 3, 3, 1, 1, 1, 1, 0, 0, 64, 1)):      # 2-wide rows
        assert N.lib().tbg_conv2d_units_small_blocks(C.byref(bad), 3) == -4  # TBG_EUNSUPPORTED
    d16 = N.ConvDesc(2, 24, 64, 8, 32, 8, 32, 3, 3, 1, 1, 1, 1, 0, 0, 64, 1)
    assert N.lib().tbg_conv2d_units_small_blocks(C.byref(d16), 1) == -4 and N.lib().tbg_conv2d_units_small_blocks(C.byref(d16), 3) > 0


class _AlwaysSink(ops.UnitSink):
    def wanted(self, B, Cc, H, W):
        return True


SINKS = [(16, 256, 256, 2, 25, 3, (1, 1), False), (3, 64, 40, 8, 25, 3, (1, 1), False), (16, 512, 512, 4, 16, 3, (1, 1), False),
         (16, 128, 256, 4, 25, 1, (2, 1), False), (16, 256, 128, 2, 25, 1, (2, 1), True), (5, 48, 96, 3, 3, 3, (1, 1), False),
         (16, 256, 512, 9, 33, 3, (2, 2), False)]


@pytest.mark.parametrize("mode", ["f32x3", "bf16"])
@pytest.mark.parametrize("geom", SINKS, ids=[str(g) for g in SINKS])
def test_conv_small_unit_sink_equals_units_pack(dev, mode, geom):
    """tbg_epilogue.units_out served by the small-map kernel: interior, ring of zero units and every plane bit for bit
    tbg_units_pack_f32(y, scale) of the launch's own fp32 output, which itself does not change with a sink; y = NULL is legal."""
    B, C, M, H, W, k, stride, transposed = geom
    planes = 3 if mode == "f32x3" else 1
    Ho, Wo = _out_hw(H, W, k, stride, transposed)
    f = lambda t: t.float().to(dev).contiguous()
    x, w = f(_rnd(B, C, H, W, seed=60)), f(_rnd(k, k, C, M, seed=61) / math.sqrt(k * k * C))
    bias, scale, res = f(_rnd(M, seed=64)), f(_rnd(B, M, seed=65)), f(_rnd(B, M, Ho, Wo, seed=66))
    with ops.compute_dtype(mode):
        XU, pf = ops.units_pack(x, None, planes=planes), ops.pack_filter(w, False, False)
        mk = lambda: N.epilogue(bias=bias, residual=res, res_first=1, act=N.ACT_LRELU, slope=0.0, gain=1.0)
        y_ref = ops.conv2d_small_raw(XU, pf, M, k, (Ho, Wo), stride, transposed, epi=mk())
        for sc in (scale, None):
            y, U = ops.conv2d_small_raw(XU, pf, M, k, (Ho, Wo), stride, transposed, epi=mk(), sink=_AlwaysSink(sc, "s1", M))
            assert U is not None and torch.equal(y, y_ref)
            ref = ops.units_pack(y_ref, sc, planes=planes)
            assert torch.equal(U.data.view(torch.int16), ref.data.view(torch.int16)), (geom, sc is not None)
