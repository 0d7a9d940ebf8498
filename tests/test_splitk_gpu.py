"""tbg_conv2d_splitk: split-K convolutions finished in the launch (the last split of a tile to arrive sums the slabs in slab
order and runs the real epilogue) against the two-launch form (slabs + tbg_slab_epilogue_f32) and an fp64 reference.

Replaces nothing of the reference's: a K split is this library's way to fill 256 CUs with the small-map layers of
layers/conv.py:51-73, models/discriminator.py:68-84 and the frozen OCR ResNet (aster/ backbone); what must hold is that the
result is the convolution's, whichever block finishes a tile (bit-identical runs), and that the tickets are left zero."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _rnd(*shape, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64)


def _mode(name):
    from textboxgan_amd import ops
    return ops.compute_dtype(name)


SHAPES = [  # B, C, M, H, W, k, ksplit
    (16, 256, 256, 2, 25, 3, 8),     # frozen-OCR stage: one image per 64-pixel tile
    (16, 128, 128, 4, 25, 3, 4),
    (16, 512, 512, 1, 25, 3, 8),
    (16, 256, 256, 2, 25, 1, 4),     # 1x1 (few-tap instantiations)
    (3, 72, 40, 5, 7, 3, 3),         # ragged: partial channel tile, several images per tile, uneven split (9 chunks / 3)
    (16, 256, 256, 8, 32, 3, 8),     # 128 x 128 tiles
    (2, 64, 96, 9, 21, 3, 8),        # more splits than... 8 chunks: one chunk per split
]


@pytest.mark.parametrize("mode", ["f32x3", "bf16"])
@pytest.mark.parametrize("shape", SHAPES, ids=[f"B{s[0]}_C{s[1]}_M{s[2]}_{s[3]}x{s[4]}_k{s[5]}_ks{s[6]}" for s in SHAPES])
def test_finish_in_launch_equals_two_launches(dev, mode, shape):
    """alpha-only epilogue: y = sum of the slabs in slab order in both forms -> the same bits; and against fp64."""
    from textboxgan_amd import ops, native as N
    B, C, M, H, W, k, ks = shape
    x, w = _rnd(B, C, H, W, seed=1), _rnd(k, k, C, M, seed=2) / math.sqrt(k * k * C)
    ref = F.conv2d(x, w.permute(3, 2, 0, 1), padding=k // 2) * 0.75
    xd = x.float().to(dev)
    with _mode(mode):
        pf = ops.pack_filter(w.float().to(dev), False, False)
        try:
            ops.TUNING.force_ksplit = ks
            ops.TUNING.fused_splitk = False
            with N.record_calls() as log0:
                y0 = ops.conv2d_raw(xd, pf, M, k, k, (H, W), (1, 1), (k // 2, k // 2), epi=N.epilogue(alpha=0.75))
            ops.TUNING.fused_splitk = True
            with N.record_calls() as log1:
                y1 = ops.conv2d_raw(xd, pf, M, k, k, (H, W), (1, 1), (k // 2, k // 2), epi=N.epilogue(alpha=0.75))
            ys = [ops.conv2d_raw(xd, pf, M, k, k, (H, W), (1, 1), (k // 2, k // 2), epi=N.epilogue(alpha=0.75)) for _ in range(20)]
        finally:
            ops.TUNING.force_ksplit = None
            ops.TUNING.fused_splitk = True
    assert "tbg_slab_epilogue_f32" in log0, log0
    assert len(log1) == 1 and all(n.startswith("conv_fprop_kernel<") for n in log1), log1  # ONE launch
    assert torch.equal(y0, y1)
    for y in ys:  # whichever block arrives last: the same bits
        assert torch.equal(y, y1)
    tol = 2e-5 if mode == "f32x3" else 2e-2
    assert float((y1.double().cpu() - ref).abs().max()) <= tol * float(ref.abs().max())
    tk = ops._splitk_tickets(dev, 1)
    assert int(tk.abs().sum()) == 0  # every ticket back at zero


@pytest.mark.parametrize("mode", ["f32x3", "bf16"])
def test_finish_in_launch_full_epilogue_and_sink(dev, mode):
    """demodulation, noise, bias, leaky ReLU, residual (both orders), gate -- conv_epilogue on the sum of the slabs -- against the
    two-launch form (last-bit differences: the slab pass evaluates v*(alpha*d) + bias, the convolution's epilogue (v*d) + n + b)
    and an fp64 reference; a unit sink written by the finishing block == units_pack of its own y."""
    from textboxgan_amd import ops, native as N
    B, C, M, H, W = 16, 256, 128, 4, 25
    x, w = _rnd(B, C, H, W, seed=3), _rnd(3, 3, C, M, seed=4) / math.sqrt(9 * C)
    d, bias, noise, strength = _rnd(B, M, seed=5).abs() + 0.5, _rnd(M, seed=6), _rnd(B, 1, H, W, seed=7), torch.tensor([0.3], dtype=torch.float64)
    res, gate = _rnd(B, M, H, W, seed=8), _rnd(B, M, H, W, seed=9)
    pre = F.conv2d(x, w.permute(3, 2, 0, 1), padding=1) * 1.25 * d[:, :, None, None] + noise * strength + bias[None, :, None, None] * 0.5
    ref_a = (F.leaky_relu(pre, 0.2) * math.sqrt(2) + res) * 0.7
    ref_b = (pre + res) * (gate > 0)
    f = lambda t: t.float().to(dev).contiguous()
    xd, dd, bd, nd, sd, rd, gd = f(x), f(d), f(bias), f(noise), f(strength), f(res), f(gate)
    tol = 3e-5 if mode == "f32x3" else 3e-2
    with _mode(mode):
        pf = ops.pack_filter(f(w), False, False)
        epi_a = lambda: N.epilogue(out_scale=dd, bias=bd, noise=nd, strength=sd, residual=rd, alpha=1.25, bias_mul=0.5,
                                   act=N.ACT_LRELU, res_scale=0.7)
        epi_b = lambda: N.epilogue(out_scale=dd, bias=bd, noise=nd, strength=sd, residual=rd, alpha=1.25, bias_mul=0.5,
                                   res_first=1, gate=gd)
        try:
            ops.TUNING.force_ksplit = 4
            outs = {}
            for fused in (False, True):
                ops.TUNING.fused_splitk = fused
                outs[fused] = (ops.conv2d_raw(xd, pf, M, 3, 3, (H, W), (1, 1), (1, 1), epi=epi_a()),
                               ops.conv2d_raw(xd, pf, M, 3, 3, (H, W), (1, 1), (1, 1), epi=epi_b()))
            # unit sink on the finishing block
            scale = f(_rnd(B, M, seed=10).abs() + 0.5)
            class _AlwaysSink(ops.UnitSink):
                def wanted(self, B, Cc, H, W):
                    return True
            sink = _AlwaysSink(scale, "s1", M)
            with N.record_calls() as log:
                ys, U = ops.conv2d_raw(xd, pf, M, 3, 3, (H, W), (1, 1), (1, 1), epi=epi_a(), sink=sink)
        finally:
            ops.TUNING.force_ksplit = None
            ops.TUNING.fused_splitk = True
        assert len(log) == 1 and all(n.startswith("conv_fprop_kernel<") for n in log), log
        assert U is not None
        Up = ops.units_pack(ys, scale, planes=U.planes)
    for got, ref in ((outs[True][0], ref_a), (outs[True][1], ref_b)):
        assert float((got.double().cpu() - ref).abs().max()) <= tol * float(ref.abs().max())
    for a, b in zip(outs[False], outs[True]):
        assert float((a - b).abs().max()) <= 1e-6 * float(a.abs().max())
    assert torch.equal(outs[True][1] == 0, (ref_b == 0).to(dev))
    assert torch.equal(ys, outs[True][0])
    assert torch.equal(U.data.view(torch.int16), Up.data.view(torch.int16))


@pytest.mark.parametrize("mode", ["f32x3", "bf16"])
def test_finish_in_launch_transposed(dev, mode):
    """stride-2 transposed 3x3 (up-convolution forward / data gradient of the strided layers), split: the per-class form with a
    real epilogue, and alpha-only (the merged store-only form where the library picks it); both == the two-launch form's bits
    for alpha-only and the fp64 reference."""
    from textboxgan_amd import ops, native as N
    B, C, M, H, W = 16, 256, 128, 8, 32
    x, w = _rnd(B, C, H, W, seed=11), _rnd(3, 3, C, M, seed=12) / math.sqrt(9 * C)
    ref = F.conv_transpose2d(x, w.permute(2, 3, 0, 1), stride=2)  # [B, M, 17, 65]: y[2a+kh, 2b+kw] += x[a, b] w[kh, kw]
    f = lambda t: t.float().to(dev).contiguous()
    tol = 3e-5 if mode == "f32x3" else 3e-2
    with _mode(mode):
        pf = ops.pack_filter(f(w), False, False)
        outs = {}
        try:
            ops.TUNING.force_ksplit = 4
            for fused in (False, True):
                ops.TUNING.fused_splitk = fused
                outs[fused] = ops.conv2d_raw(f(x), pf, M, 3, 3, (17, 65), (2, 2), (0, 0), transposed=True)
            bias = f(_rnd(M, seed=13))
            yb = ops.conv2d_raw(f(x), pf, M, 3, 3, (17, 65), (2, 2), (0, 0), transposed=True, epi=N.epilogue(bias=bias))
        finally:
            ops.TUNING.force_ksplit = None
            ops.TUNING.fused_splitk = True
    assert torch.equal(outs[False], outs[True])
    assert float((outs[True].double().cpu() - ref).abs().max()) <= tol * float(ref.abs().max())
    refb = ref + bias.double().cpu()[None, :, None, None]
    assert float((yb.double().cpu() - refb).abs().max()) <= tol * float(refb.abs().max())


def test_fp32_mode_keeps_two_launches(dev):
    """the exact-fp32 builds have no in-launch finish (TBG_EUNSUPPORTED from the entry; ops never asks): two launches, same API"""
    import ctypes as C
    from textboxgan_amd import ops, native as N
    B, Cc, M, H, W = 4, 64, 64, 4, 25
    xd = _rnd(B, Cc, H, W, seed=20).float().to(dev)
    with _mode("f32"):
        pf = ops.pack_filter((_rnd(3, 3, Cc, M, seed=21) / 24).float().to(dev), False, False)
        try:
            ops.TUNING.force_ksplit = 4
            with N.record_calls() as log:
                ops.conv2d_raw(xd, pf, M, 3, 3, (H, W), (1, 1), (1, 1))
        finally:
            ops.TUNING.force_ksplit = None
    assert "tbg_slab_epilogue_f32" in log
    d = N.ConvDesc(B, Cc, M, H, W, H, W, 3, 3, 1, 1, 1, 1, 0, 0, M, 4)
    y = torch.empty(B, M, H, W, device=dev)
    slabs = torch.empty(4, B, M, H, W, device=dev)
    tk = torch.zeros(64, device=dev, dtype=torch.int32)
    rc = N.lib().tbg_conv2d_splitk(C.byref(d), N.ptr(xd), N.ptr(pf.data), N.ptr(y), N.ptr(slabs), slabs.numel(), N.ptr(tk), 64, None,
                                   C.byref(N.epilogue()), 0, N.stream())
    assert N.lib().tbg_conv2d_splitk_slab_floats(C.byref(d), 0, 0) == N.EUNSUPPORTED
    assert rc == N.EUNSUPPORTED
