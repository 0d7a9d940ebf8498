"""fp32 conv instantiation families per layer shape (tbg_conv2d_f32_variant): TFLOP/s of auto / pipelined / plain / occ4
for the forward and the data gradient of the step's 3x3 stride-1 and strided layers at batch B."""
import sys; sys.path.insert(0, '.')
import torch
from textboxgan_amd import ops
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
BF16 = len(sys.argv) > 2 and sys.argv[2] == "bf16"
X3 = len(sys.argv) > 2 and sys.argv[2] == "f32x3"

def timeit(fn, n=30):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

L = [("64x256 128->128", 128, 128, 64, 256, (1, 1)), ("32x128 128->128", 128, 128, 32, 128, (1, 1)),
     ("16x64 256->256", 256, 256, 16, 64, (1, 1)), ("16x64 128->128", 128, 128, 16, 64, (1, 1)),
     ("8x32 256->256", 256, 256, 8, 32, (1, 1)), ("8x16 256->256", 256, 256, 8, 16, (1, 1)),
     ("4x16 512->512", 512, 512, 4, 16, (1, 1)), ("4x8 512->512", 512, 512, 4, 8, (1, 1)),
     ("64x256 64->64", 64, 64, 64, 256, (1, 1)), ("32x128 256->128 (dgrad shape)", 256, 128, 32, 128, (1, 1)),
     ("down 66x258 64->128", 64, 128, 66, 258, (2, 2)), ("down 34x130 128->128", 128, 128, 34, 130, (2, 2)),
     ("down 18x66 128->256", 128, 256, 18, 66, (2, 2)), ("down 10x34 256->256 (w only)", 256, 256, 8, 34, (1, 2))]
names = {0: "auto", 3: "regprefetch", 4: "plain-loop"} if BF16 else {0: "auto", 7: "regprefetch", 1: "pipelined", 2: "plain", 3: "occ4"}
if X3:
    names = {0: "auto", 1: "tile128x256", 2: "tile128x128"}
print(f"B={B}  TFLOP/s per variant (ksplit as the heuristic picks it)")
for name, C, M, H, W, stride in L:
    x = torch.randn(B, C, H, W, device=dev)
    wp = ops.pack_filter(torch.randn(3, 3, C, M, device=dev), False, False, bf16="f32x3" if X3 else BF16)
    pad = (1, 1) if stride == (1, 1) else (0, 0)
    ohw = ((H + 2 * pad[0] - 3) // stride[0] + 1, (W + 2 * pad[1] - 3) // stride[1] + 1)
    flops = 2.0 * B * C * M * 9 * ohw[0] * ohw[1]
    row = f"{name:34s}"
    for v in names:
        ops.TUNING.force_variant = v
        try:
            t = timeit(lambda: ops.conv2d_raw(x, wp, M, 3, 3, ohw, stride, pad))
            row += f" {names[v]}={flops / t / 1e9:6.1f}"
        except Exception:
            row += f" {names[v]}=  n/a "
    ops.TUNING.force_variant = 0
    print(row)
