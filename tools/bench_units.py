"""isolated timings of the unit-tensor kernels against the NCHW kernels they replace: python tools/bench_units.py [B] [lib.so | -] [wgrad]"""
import sys, torch
sys.path.insert(0, ".")
from textboxgan_amd import ops
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
if len(sys.argv) > 2 and sys.argv[2] != "-":  # a variant build of the library (tools/build_variant.sh)
    import os
    from textboxgan_amd import native
    native.LIB_PATH = os.path.abspath(sys.argv[2])
ONLY_WGRAD = len(sys.argv) > 3 and sys.argv[3] == "wgrad"
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (C, M, H, W) in [(128, 128, 64, 256), (128, 128, 32, 128), (256, 256, 16, 64), (512, 512, 8, 32), (64, 64, 64, 256)]:
    x, dy = torch.randn(B, C, H, W, device=dev), torch.randn(B, M, H, W, device=dev)
    xs, ds = torch.rand(B, C, device=dev) + 0.5, torch.rand(B, M, device=dev) + 0.5
    dw = torch.empty(3, 3, C, M, device=dev)
    g = ops._Geom((1, 1), (1, 1), 3, 3, (H, W), (H, W))
    fl = 2.0 * B * C * M * H * W * 9
    row = f"B={B} {C}->{M} {H}x{W}:"
    for mode, planes in (("f32x3", 3), ("bf16", 1)):
        with ops.compute_dtype(mode):
            t_old = timeit(lambda: ops._bwd_weight_launch(x, dy, g, C, M, alpha=1.0, x_scale=xs, dy_scale=ds))
        t_pack = timeit(lambda: ops.units_pack(x, xs, planes=planes))
        SU, LU = ops.units_pack(dy, ds, planes=planes), ops.units_pack(x, xs, planes=planes)
        t_new = timeit(lambda: ops.wgrad_units_raw(SU, LU, dw, C * M, M, 1, 1.0))
        row += (f"  [{mode}] nchw {t_old:7.1f} us ({fl / t_old / 1e6:6.1f} TF)  units {t_new:7.1f} us ({fl / t_new / 1e6:6.1f} TF)"
                f"  pack {t_pack:6.1f} us ({(4 + 2 * planes) * x.numel() / t_pack / 1e6:5.2f} TB/s)")
    print(row, flush=True)

if ONLY_WGRAD:
    sys.exit(0)
print("--- forward 3x3 s1 (with style scale, demod/noise/bias/lrelu epilogue)")
from textboxgan_amd import native as N
for (C, M, H, W, Bx) in [(128, 128, 64, 256, B), (128, 128, 32, 128, B), (128, 128, 32, 128, 2 * B), (64, 64, 64, 256, 2 * B), (256, 256, 16, 64, B),
                         (256, 256, 16, 64, 2 * B), (512, 512, 8, 32, B)]:
    x = torch.randn(Bx, C, H, W, device=dev)
    w = torch.randn(3, 3, C, M, device=dev) / (9 * C) ** 0.5
    xs, dd = torch.rand(Bx, C, device=dev) + 0.5, torch.rand(Bx, M, device=dev) + 0.5
    nz, bs, st = torch.randn(Bx, 1, H, W, device=dev), torch.randn(M, device=dev), torch.tensor(0.1, device=dev)
    fl = 2.0 * Bx * C * M * H * W * 9
    row = f"B={Bx} {C}->{M} {H}x{W}:"
    for mode, planes in (("f32x3", 3), ("bf16", 1)):
        with ops.compute_dtype(mode):
            pf = ops.pack_filter(w, False, False)
            epi = lambda: N.epilogue(out_scale=dd, bias=bs, noise=nz, strength=st, act=N.ACT_LRELU)
            out = torch.empty(Bx, M, H, W, device=dev)
            t_old = timeit(lambda: ops.conv2d_raw(x, pf, M, 3, 3, (H, W), (1, 1), (1, 1), in_scale=xs, epi=epi(), out=out))
            XU = ops.units_pack(x, xs, planes=planes)
            t_new = timeit(lambda: ops.conv2d_units_raw(XU, pf, M, epi=epi(), out=out))
            # the data-gradient form (flipped transposed filter, per-channel output scale, fused dot with a second tensor: OPT = 2) and
            # the forward with a unit sink (the next layer's operand written by this launch)
            DU, pft, aux, dot = ops.units_pack(out, dd, planes=planes), ops.pack_filter(w, True, True), torch.randn_like(x), torch.empty(Bx, C, device=dev)
            dx = torch.empty(Bx, C, H, W, device=dev)
            t_dg = timeit(lambda: ops.conv2d_units_raw(DU, pft, C, epi=N.epilogue(out_scale=xs), dot=(aux, dot), out=dx)) if C == M else float("nan")
            class _Sink(ops.UnitSink):
                def wanted(self, *a):
                    return True
            sink = _Sink(dd, "s1", M)
            t_sk = timeit(lambda: ops.conv2d_units_raw(XU, pf, M, epi=epi(), out=out, sink=sink))
        row += (f"  [{mode}] nchw {t_old:7.1f} us ({fl / t_old / 1e6:6.1f} TF)  units {t_new:7.1f} us ({fl / t_new / 1e6:6.1f} TF)"
                f"  +sink {t_sk:7.1f}  dgrad+dot {t_dg:7.1f}")
    print(row, flush=True)
