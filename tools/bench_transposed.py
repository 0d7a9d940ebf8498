"""stride-2 transposed 3x3 convolution: merged-class kernel (variant 0) vs one launch per parity class (variant 4),
fp32 and bf16, at the step's shapes (G up-conv forward; data gradient of D's stride-2 convs)."""
import sys; sys.path.insert(0, '.')
import torch
from textboxgan_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, n=30):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
L = [("G up 32x128 128->128", 128, 128, 32, 128), ("G up 16x64 256->128", 256, 128, 16, 64), ("G up 8x32 256->256", 256, 256, 8, 32),
     ("G up 4x16 512->256", 512, 256, 4, 16), ("G up 2x8 128->512", 128, 512, 2, 8),
     ("D dgrad 32x128 128->64", 128, 64, 32, 128), ("D dgrad 16x64 128->128", 128, 128, 16, 64), ("D dgrad 8x32 256->128", 256, 128, 8, 32),
     ("D dgrad 4x16 256->256", 256, 256, 4, 16)]
MODES = ((16, "f32x3"), (32, "f32x3")) if len(sys.argv) > 1 and sys.argv[1] == "f32x3" else ((16, False), (32, True))
for B, bf in MODES:
    print(f"--- B={B} {bf if isinstance(bf, str) else 'bf16' if bf else 'fp32'}: TFLOP/s auto | per-class, full-height tiles (variant 4) | [fp32: tile height / 2 (8) | / 4 (9)] | merged (5)")
    for name, C, M, H, W in L:
        x = torch.randn(B, C, H, W, device=dev)
        wp = ops.pack_filter(torch.randn(3, 3, C, M, device=dev), False, False, bf16=bf)
        flops = 2.0 * B * C * M * 9 * H * W
        row = f"{name:28s}"
        for v in ((0, 4, 8, 9, 5) if not bf else (0, 4, 5)):
            ops.TUNING.force_variant = v
            t = timeit(lambda: ops.conv2d_raw(x, wp, M, 3, 3, (2 * H + 1, 2 * W + 1), (2, 2), (0, 0), transposed=True, flip=True))
            row += f" {flops / t / 1e9:7.1f}"
        ops.TUNING.force_variant = 0
        print(row)
