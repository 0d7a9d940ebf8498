"""isolated timings of the phase-unit stride-2 kernels against the NCHW stride-2 kernels they replace:
python tools/bench_units_s2.py [B]"""
import sys, torch
sys.path.insert(0, ".")
from textboxgan_amd import ops, native as N
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
print("--- forward 3x3 s2 (D: blur output -> conv_1 with bias + lrelu; G: data gradient of the up-convolution with out_scale + dot)")
for (C, M, Hin, Win, Bx) in [(128, 128, 65, 257, B), (64, 128, 66, 258, 2 * B), (128, 256, 33, 129, B), (128, 256, 34, 130, 2 * B),
                             (256, 512, 18, 66, 2 * B), (256, 256, 17, 65, B)]:
    Ho, Wo = (Hin - 3) // 2 + 1, (Win - 3) // 2 + 1
    x = torch.randn(Bx, C, Hin, Win, device=dev)
    w = torch.randn(3, 3, C, M, device=dev) / (9 * C) ** 0.5
    bs = torch.randn(M, device=dev)
    fl = 2.0 * Bx * C * M * Ho * Wo * 9
    row = f"B={Bx} {C}->{M} {Hin}x{Win}:"
    for mode, planes in (("f32x3", 3), ("bf16", 1)):
        with ops.compute_dtype(mode):
            pf = ops.pack_filter(w, False, False)
            epi = lambda: N.epilogue(bias=bs, act=N.ACT_LRELU)
            out = torch.empty(Bx, M, Ho, Wo, device=dev)
            t_old = timeit(lambda: ops.conv2d_raw(x, pf, M, 3, 3, (Ho, Wo), (2, 2), (0, 0), epi=epi(), out=out))
            t_pack = timeit(lambda: ops.units_pack_s2(x, planes=planes))
            XP = ops.units_pack_s2(x, planes=planes)
            d = N.ConvDesc(Bx, C, M, Hin, Win, Ho, Wo, 3, 3, 2, 2, 0, 0, 0, 0, M, 1)
            import ctypes
            nb = N.lib().tbg_conv2d_units_s2_blocks(ctypes.byref(d), planes)
            t_new = timeit(lambda: ops.conv2d_units_s2_raw(XP, pf, M, epi=epi(), out=out))
        row += (f"  [{mode}] nchw {t_old:7.1f} us ({fl / t_old / 1e6:6.1f} TF)  units {t_new:7.1f} us ({fl / t_new / 1e6:6.1f} TF, {nb} blocks)"
                f"  pack {t_pack:6.1f} us")
    print(row, flush=True)
print("--- filter gradient 3x3 s2 (S = gradient on the output grid, scaled; L = the strided convolution's input)")
for (CS, CL, Hl, Wl, Bx) in [(128, 128, 65, 257, B), (128, 64, 66, 258, 2 * B), (256, 128, 33, 129, B), (256, 128, 34, 130, 2 * B),
                             (512, 256, 18, 66, 2 * B), (256, 256, 17, 65, B)]:
    Hs, Ws = (Hl - 3) // 2 + 1, (Wl - 3) // 2 + 1
    x, dy = torch.randn(Bx, CL, Hl, Wl, device=dev), torch.randn(Bx, CS, Hs, Ws, device=dev)
    ds = torch.rand(Bx, CS, device=dev) + 0.5
    dw = torch.empty(3, 3, CL, CS, device=dev)
    g = ops._Geom((2, 2), (0, 0), 3, 3, (Hl, Wl), (Hs, Ws))
    fl = 2.0 * Bx * CS * CL * Hs * Ws * 9
    row = f"B={Bx} CS={CS} CL={CL} L={Hl}x{Wl}:"
    for mode, planes in (("f32x3", 3), ("bf16", 1)):
        with ops.compute_dtype(mode):
            t_old = timeit(lambda: ops._bwd_weight_launch(x, dy, g, CL, CS, alpha=1.0, dy_scale=ds))
        SU, LP = ops.units_pack(dy, ds, planes=planes), ops.units_pack_s2(x, planes=planes)
        t_new = timeit(lambda: ops.wgrad_units_s2_raw(SU, LP, dw, CL * CS, CS, 1, 1.0))
        row += f"  [{mode}] nchw {t_old:7.1f} us ({fl / t_old / 1e6:6.1f} TF)  units {t_new:7.1f} us ({fl / t_new / 1e6:6.1f} TF)"
    print(row, flush=True)
print("--- producers: FIR (NCHW) + stand-alone phase pack vs the fused FIR -> phase units")
for (C, H, W, Bx, pad, gain) in [(128, 64, 256, B, (2, 2, 2, 2), 4.0), (64, 64, 256, 2 * B, (2, 3, 2, 3), 1.0), (128, 32, 128, 2 * B, (2, 3, 2, 3), 1.0),
                                 (256, 16, 64, 2 * B, (2, 3, 2, 3), 1.0)]:
    x = torch.randn(Bx, C, H, W, device=dev)
    sc = torch.rand(Bx * C, device=dev) + 0.5
    k = ops.fir_kernel(dev, gain)
    row = f"B={Bx} C={C} {H}x{W} pad={pad}:"
    t_fir = timeit(lambda: ops.upfirdn2d_raw(x, k, pad=pad, in_scale=sc))
    t = ops.upfirdn2d_raw(x, k, pad=pad, in_scale=sc)
    for planes in (3, 1):
        t_pack = timeit(lambda: ops.units_pack_s2(t, planes=planes))
        t_fused = timeit(lambda: ops.upfirdn2d_units_s2(x, k, pad=pad, in_scale=sc, planes=planes))
        nb = 4.0 * x.numel() + 2.0 * planes * t.numel()
        row += f"  [planes={planes}] fir {t_fir:6.1f} + pack {t_pack:6.1f} us   fused {t_fused:6.1f} us ({nb / t_fused / 1e6:5.2f} TB/s)"
    print(row, flush=True)
print("--- transposed 3x3 s2 (G: up-convolution forward; D: data gradient of the strided convolution)")
for (C, M, H, W, Bx, extra) in [(128, 128, 32, 128, B, 1), (128, 64, 32, 128, 2 * B, 2), (128, 64, 32, 128, B, 2), (256, 128, 16, 64, B, 1),
                                (256, 128, 16, 64, 2 * B, 2), (512, 256, 8, 32, 2 * B, 2)]:
    x = torch.randn(Bx, C, H, W, device=dev)
    w = torch.randn(3, 3, C, M, device=dev) / (9 * C) ** 0.5
    xs = torch.rand(Bx, C, device=dev) + 0.5
    fl = 2.0 * Bx * C * M * H * W * 9
    hw = (2 * H + extra, 2 * W + extra)
    row = f"B={Bx} {C}->{M} {H}x{W}->{hw[0]}x{hw[1]}:"
    for mode, planes in (("f32x3", 3), ("bf16", 1)):
        with ops.compute_dtype(mode):
            pf = ops.pack_filter(w, False, False)
            out = torch.empty(Bx, M, *hw, device=dev)
            t_old = timeit(lambda: ops.conv2d_raw(x, pf, M, 3, 3, hw, (2, 2), (0, 0), transposed=True, flip=True, in_scale=xs, out=out))
            XU = ops.units_pack(x, xs, planes=planes)
            import ctypes
            d = N.ConvDesc(Bx, C, M, H, W, hw[0], hw[1], 3, 3, 2, 2, 0, 0, 1, 1, M, 1)
            nb = N.lib().tbg_conv2d_units_t2_blocks(ctypes.byref(d), planes)
            t_new = timeit(lambda: ops.conv2d_units_t2_raw(XU, pf, M, hw, flip=True, out=out))
        row += f"  [{mode}] nchw {t_old:7.1f} us ({fl / t_old / 1e6:6.1f} TF)  units {t_new:7.1f} us ({fl / t_new / 1e6:6.1f} TF, {nb} blocks)"
    print(row, flush=True)
