"""Per-layer micro-benchmark of the MFMA conv kernels at the BASELINE configs[1] shapes (B=16)."""
import sys; sys.path.insert(0, '.')
import math, torch
from textboxgan_amd import ops, native as N
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ops._TLS.compute = sys.argv[2] if len(sys.argv) > 2 else "f32"  # f32 | bf16 | f32x3
print("arithmetic:", ops.compute_mode())

def timeit(fn, n=30):
    for _ in range(10): fn()  # clocks ramp from idle over the first launches: 5 timed calls under-reported by ~20%
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

# (name, C, M, H, W, k, kind)  kind: s1 = 3x3 SAME ; up = transposed s2 ; down = strided VALID (input H+2,W+2)
L = [("G 64x256 128->128", 128, 128, 64, 256, 3, "s1"), ("G 32x128 128->128", 128, 128, 32, 128, 3, "s1"),
     ("G 16x64 256->256", 256, 256, 16, 64, 3, "s1"), ("G 8x32 256->256", 256, 256, 8, 32, 3, "s1"),
     ("G 4x16 512->512", 512, 512, 4, 16, 3, "s1"),
     ("G up 32x128->64x256 128->128", 128, 128, 32, 128, 3, "up"), ("G up 16x64 256->128", 256, 128, 16, 64, 3, "up"),
     ("G up 8x32 256->256", 256, 256, 8, 32, 3, "up"), ("G up 4x16 512->256", 512, 256, 4, 16, 3, "up"),
     ("D 64x256 64->64", 64, 64, 64, 256, 3, "s1"), ("D 32x128 128->128", 128, 128, 32, 128, 3, "s1"),
     ("D down 64x256 64->128", 64, 128, 66, 258, 3, "down"), ("D down 32x128 128->128", 128, 128, 34, 130, 3, "down"),
     ("D 4x8 512->512", 512, 512, 4, 8, 3, "s1"), ("toRGB 64x256 128->3", 128, 3, 64, 256, 1, "s1"),
     ("fromRGB 64x256 3->64", 3, 64, 64, 256, 1, "s1")]
print(f"{'layer':34s} {'GFLOP':>8s} | {'fwd ms':>8s} {'TF':>6s} | {'bwdD ms':>8s} {'TF':>6s} | {'wgrad ms':>8s} {'TF':>6s}")
tot = [0, 0, 0]
for name, C, M, H, W, k, kind in L:
    x = torch.randn(B, C, H, W, device=dev)
    w = torch.randn(k, k, C, M, device=dev)
    wp = ops.pack_filter(w, False, False)   # packed once: the timings below are the conv kernels alone
    if kind == "s1":
        ohw = (H, W); g = ops._Geom((1, 1), (k // 2, k // 2), k, k, (H, W), ohw)
        fwd = lambda: ops.conv2d_raw(x, wp, M, k, k, ohw, (1, 1), (k // 2, k // 2))
        flops = 2 * B * C * M * k * k * H * W
    elif kind == "up":
        ohw = (2 * H + 1, 2 * W + 1); g = ops._Geom((2, 2), (0, 0), k, k, ohw, (H, W))
        fwd = lambda: ops.conv2d_raw(x, wp, M, k, k, ohw, (2, 2), (0, 0), transposed=True, flip=True)
        flops = 2 * B * C * M * k * k * H * W
    else:
        ohw = ((H - 3) // 2 + 1, (W - 3) // 2 + 1); g = ops._Geom((2, 2), (0, 0), k, k, (H, W), ohw)
        fwd = lambda: ops.conv2d_raw(x, wp, M, k, k, ohw, (2, 2), (0, 0))
        flops = 2 * B * C * M * k * k * ohw[0] * ohw[1]
    y = fwd()
    dy = torch.randn_like(y)
    if kind == "up":   # bwd-data of the transposed conv = strided conv of dy ; wgrad: S = x, L = dy
        wt = ops.pack_filter(w, True, True)
        bwd = lambda: ops.conv2d_raw(dy, wt, C, k, k, (H, W), (2, 2), (0, 0))
        dw = torch.empty_like(w)
        wg = lambda: ops.wgrad_raw(x, dy, k, k, (2, 2), (0, 0), dw, -C * M, 1, M, 1.0, out_offset=(k * k - 1) * C * M)
    else:
        if kind == "s1":
            wt = ops.pack_filter(w, True, True)
            bwd = lambda: ops.conv2d_raw(dy, wt, C, k, k, (H, W), (1, 1), (k - 1 - k // 2, k - 1 - k // 2))
        else:
            wt = ops.pack_filter(w, True, False)
            bwd = lambda: ops.conv2d_raw(dy, wt, C, k, k, (H, W), (2, 2), (0, 0), transposed=True)
        wg = lambda: ops._bwd_weight_launch(x, dy, g, C, M)
    t = [timeit(fwd), timeit(bwd), timeit(wg)]
    for i in range(3): tot[i] += t[i]
    print(f"{name:34s} {flops/1e9:8.2f} | {t[0]:8.3f} {flops/t[0]/1e9:6.1f} | {t[1]:8.3f} {flops/t[1]/1e9:6.1f} | {t[2]:8.3f} {flops/t[2]/1e9:6.1f}")
print("sum ms", [round(v, 2) for v in tot])
