"""-m gpu: the "f32x3" arithmetic (tbg.h: fp32 operands split into three bf16 terms, six partial products on the bf16 matrix
pipe, fp32 accumulate) against float64 at the UNCHANGED fp32 tolerances of tests/test_kernels_gpu.py / test_fullwidth_gpu.py
(the split path may stand in for fp32 only if it passes the fp32 bars)."""
import math

import numpy as np
import pytest
import torch

from textboxgan_amd import native as N, ops

pytestmark = pytest.mark.gpu


def _bf16_planes(pf):
    n = pf.data.numel() // 3
    return [pf.data[i * n:(i + 1) * n].float().double() for i in range(3)]


@pytest.mark.parametrize("transpose,flip", [(False, False), (True, True), (True, False)])
def test_weight_pack_x3_is_an_exact_split(dev, transpose, flip):
    """hi + mid + lo == w bit for bit (summed in float64), in the unit layout of the bf16 pack."""
    g = torch.Generator().manual_seed(3)
    T, I, O = 9, 20, 12  # C not a multiple of 8: zero padded units
    w = (torch.randn(T, I, O, generator=g) * torch.exp(4 * torch.randn(T, I, O, generator=g))).to(dev)
    pf3 = ops.pack_filter(w, transpose, flip, bf16="f32x3")
    pf1 = ops.pack_filter(w, transpose, flip, bf16="bf16")
    assert pf3.fmt == ops.FMT_X3 and pf3.data.numel() == 3 * pf1.data.numel()
    hi, mid, lo = _bf16_planes(pf3)
    assert torch.equal(hi, pf1.data.float().double()), "plane 0 is the RNE bf16 pack"
    C, M = (O, I) if transpose else (I, O)
    C8 = (C + 7) // 8
    tot = (hi + mid + lo).reshape(T, C8, M, 8)
    wd = w.double()
    for t in range(T):
        td = T - 1 - t if flip else t
        src = wd[t].t() if transpose else wd[t]          # [C, M]
        pad = torch.zeros(C8 * 8, M, dtype=torch.float64, device=dev)
        pad[:C] = src
        exp = pad.reshape(C8, 8, M).permute(0, 2, 1)     # [C8, M, 8]
        assert torch.equal(tot[td], exp), (t, float((tot[td] - exp).abs().max()))


def _cases():
    from test_fullwidth_gpu import _conv_cases_for_coverage
    return _conv_cases_for_coverage()


@pytest.mark.parametrize("ci", range(18))
def test_conv_x3_matches_float64_at_fp32_tolerance(dev, ci):
    """every conv case of the fp32 coverage list (forward, data gradient; transposed classes; split-K; strided) in f32x3
    arithmetic: <= 3e-5 of the float64 result -- the fp32 kernels' bar -- and not worse than 2x the exact-fp32 kernel."""
    import torch.nn.functional as F
    case = _cases()[ci]
    B, Cc, Mo, H, W, k, stride, pad, transposed = case
    g = torch.Generator().manual_seed(900 + ci)
    x = torch.randn(B, Cc, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(k, k, Cc, Mo, generator=g, dtype=torch.float64) / math.sqrt(k * k * Cc)
    s = torch.rand(B, Cc, generator=g, dtype=torch.float64) + 0.5  # style modulation while staging
    rel = lambda a, r: float((a.double().cpu() - r).abs().max() / (r.abs().max() + 1e-30))
    xd, wd, sd = x.float().to(dev), w.float().to(dev), s.float().to(dev)
    x32, w32, s32 = xd.double().cpu(), wd.double().cpu(), sd.double().cpu()  # the fp32 operands the kernels see
    xs = x32 * s32[:, :, None, None]
    errs = {}
    for mode in ("f32", "f32x3"):
        with ops.compute_dtype(mode):
            if transposed:
                ref = F.conv_transpose2d(xs, w32.permute(2, 3, 0, 1), stride=stride)
                y = ops.conv2d_raw(xd, wd, Mo, k, k, (ref.shape[2], ref.shape[3]), stride, (0, 0), transposed=True,
                                   in_scale=sd)
                errs[mode] = rel(y, ref)
            else:
                ref = F.conv2d(xs, w32.permute(3, 2, 0, 1), stride=stride, padding=pad)
                geom = ops._Geom(stride, pad, k, k, (H, W), (ref.shape[2], ref.shape[3]))
                y = ops.conv2d_raw(xd, ops.pack_filter(wd, False, False), Mo, k, k, geom.yhw, stride, pad, in_scale=sd)
                dy = torch.randn(*ref.shape, generator=g, dtype=torch.float64).float()
                xr = x32.clone().requires_grad_(True)
                (gx,) = torch.autograd.grad(F.conv2d(xr, w32.permute(3, 2, 0, 1), stride=stride, padding=pad), xr, dy.double())
                dx = ops._bwd_data_launch(dy.to(dev), wd, geom)
                errs[mode] = max(rel(y, ref), rel(dx, gx))
    print(f"\nX3ERR {case}: f32 {errs['f32']:.3e}  f32x3 {errs['f32x3']:.3e}")
    assert errs["f32x3"] < 3e-5, (case, errs)
    assert errs["f32x3"] <= max(2.0 * errs["f32"], 1e-6), (case, errs)


def test_conv_x3_kernel_names_and_merged_form(dev):
    d = N.ConvDesc(16, 128, 128, 32, 128, 65, 257, 3, 3, 2, 2, 0, 0, 1, 1, 128, 1)
    assert N.conv_kernel_name(d, True, ops.FMT_X3).endswith("true, true, true>")
    d = N.ConvDesc(16, 128, 128, 64, 256, 64, 256, 3, 3, 1, 1, 1, 1, 0, 0, 128, 1)  # 2048 tiles of 128 x 128 -> 128 x 256 tiles
    assert N.conv_kernel_name(d, True, ops.FMT_X3) == "conv_fprop_kernel<2, 2, 2, 4, 8, 9, 0, 2, true, false, true>"
    d = N.ConvDesc(16, 128, 128, 32, 128, 32, 128, 3, 3, 1, 1, 1, 1, 0, 0, 128, 1)
    assert N.conv_kernel_name(d, True, ops.FMT_X3) == "conv_fprop_kernel<2, 2, 2, 2, 8, 9, 0, 2, true, false, true>"


def test_conv_x3_merged_transposed_vs_float64(dev):
    """the merged-class form (all four output-parity classes from one halo tile) in f32x3 arithmetic, forced by variant 5."""
    import ctypes as C
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(77)
    B, Cc, Mo, H, W = 3, 72, 96, 9, 33
    x = torch.randn(B, Cc, H, W, generator=g).to(dev)
    w = (torch.randn(3, 3, Cc, Mo, generator=g) / math.sqrt(9 * Cc)).to(dev)
    ref = F.conv_transpose2d(x.double().cpu(), w.double().cpu().permute(2, 3, 0, 1), stride=2)
    pf = ops.pack_filter(w, False, False, bf16="f32x3")
    d = N.ConvDesc(B, Cc, Mo, H, W, 2 * H + 1, 2 * W + 1, 3, 3, 2, 2, 0, 0, 1, 0, pf.M, 1)
    for variant in (4, 5):
        y = torch.full((B, Mo, 2 * H + 1, 2 * W + 1), float("nan"), device=dev)
        e = N.epilogue()
        N.check(N.lib().tbg_conv2d_x3_variant(C.byref(d), N.ptr(x), N.ptr(pf.data), N.ptr(y), None, C.byref(e), variant,
                                              N.stream()), "x3 variant")
        err = float((y.double().cpu() - ref).abs().max() / ref.abs().max())
        assert err < 3e-5, (variant, err)


def _rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64)


def _rel(a, r):
    return float((a.double().cpu() - r).abs().max() / (r.abs().max() + 1e-30))


X3_WG1 = [(3, 40, 72, 5, 36), (2, 64, 64, 8, 32), (1, 130, 70, 2, 128), (2, 64, 64, 3, 100), (4, 128, 128, 16, 64)]
X3_WG2 = [(2, 40, 72, 11, 65), (3, 64, 64, 9, 130), (1, 130, 70, 5, 129), (4, 64, 128, 33, 129)]


@pytest.mark.parametrize("stride,case", [(1, c) for c in X3_WG1] + [(2, c) for c in X3_WG2],
                         ids=[f"s1-{c}" for c in X3_WG1] + [f"s2-{c}" for c in X3_WG2])
def test_wgrad_x3_matches_float64_at_fp32_tolerance(dev, stride, case):
    """conv_wgrad_x3_kernel (both float4-staged geometries; ragged maps, partial channel tiles, several chunks per block) with
    both per-(sample, channel) scale vectors and the fused additive term against float64 on the fp32 operands: the fp32
    kernels' bar (3e-5 / 5e-5 for the strided form), and not worse than 2x the exact fp32 kernel."""
    import torch.nn.functional as F
    B, C, M, H, W = case
    Ho, Wo = ((H - 3) // 2 + 1, (W - 3) // 2 + 1) if stride == 2 else (H, W)
    x, dy = _rnd(B, C, H, W, seed=40), _rnd(B, M, Ho, Wo, seed=41)
    xs, ds = _rnd(B, C, seed=42).abs() + 0.5, _rnd(B, M, seed=43).abs() + 0.5
    addw, addq = _rnd(3, 3, C, M, seed=44), _rnd(C, M, seed=45)
    f32 = lambda t: t.float().double()
    xr = (f32(x) * f32(xs)[:, :, None, None]).float().double()      # the scaled operands as the kernels form them (fp32)
    dyr = (f32(dy) * f32(ds)[:, :, None, None]).float().double()
    w = torch.zeros(3, 3, C, M, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(xr, w.permute(3, 2, 0, 1), stride=stride, padding=1 if stride == 1 else 0)
    (ref,) = torch.autograd.grad(y, w, dyr)
    ref = 0.7 * ref + 0.3 * f32(addw) * f32(addq)[None, None]
    f = lambda t: t.float().to(dev).contiguous()
    g = ops._Geom((stride, stride), (1, 1) if stride == 1 else (0, 0), 3, 3, (H, W), (Ho, Wo))
    pad = 1 if stride == 1 else 0
    desc = N.WgradDesc(B, M, C, Ho, Wo, H, W, 3, 3, stride, stride, pad, pad, C * M, M, 1, 0.7)
    assert N.wgrad_kernel_name(desc, ops.FMT_X3) == f"conv_wgrad_x3_kernel<{stride}>"
    errs = {}
    for mode in ("f32", "f32x3"):
        with ops.compute_dtype(mode):
            dw = ops._bwd_weight_launch(f(x), f(dy), g, C, M, alpha=0.7, x_scale=f(xs), dy_scale=f(ds),
                                        add=(f(addw), f(addq), 0.3))
        errs[mode] = _rel(dw, ref)
    print(f"\nX3WG s{stride} {case}: f32 {errs['f32']:.3e}  f32x3 {errs['f32x3']:.3e}")
    assert errs["f32x3"] < (3e-5 if stride == 1 else 5e-5), errs
    assert errs["f32x3"] <= max(2.0 * errs["f32"], 1e-6), errs


def test_wgrad_x3_small_geometries_fall_back_to_exact_fp32(dev):
    """narrow maps / 1x1 filters are not x3 geometries: the entry runs the exact fp32 kernel (and says so by name)."""
    desc = N.WgradDesc(4, 512, 513, 4, 4, 4, 4, 3, 3, 1, 1, 1, 1, 513 * 512, 512, 1, 1.0)
    assert N.wgrad_kernel_name(desc, ops.FMT_X3) == N.wgrad_kernel_name(desc, ops.FMT_F32)
    desc = N.WgradDesc(2, 64, 3, 16, 64, 16, 64, 1, 1, 1, 1, 0, 0, 3 * 64, 64, 1, 1.0)
    assert N.wgrad_kernel_name(desc, ops.FMT_X3) == N.wgrad_kernel_name(desc, ops.FMT_F32)
