"""Fixed cost (prologue + epilogue + launch) of the fprop kernel: time vs C at constant output, and a plain fill of
the same output for reference."""
import sys; sys.path.insert(0, '.')
import torch
from textboxgan_amd import ops
dev = torch.device('cuda:0')
def timeit(f, n=50):
    for _ in range(5): f()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = 16
for (M, H, W) in ((128, 64, 256), (128, 32, 128), (256, 16, 64), (256, 8, 32)):
    y = torch.empty(B, M, H, W, device=dev)
    tf = timeit(lambda: y.fill_(1.0))
    ts = {}
    for C in (8, 64, 128):
        x = torch.randn(B, C, H, W, device=dev); w = ops.pack_filter(torch.randn(9, C, M, device=dev), False, False)
        ops.TUNING.force_ksplit = 1
        ts[C] = timeit(lambda: ops.conv2d_raw(x, w, M, 3, 3, (H, W), (1, 1), (1, 1)))
    slope = (ts[128] - ts[64]) / 64
    print(f"M={M} {H}x{W}: fill {tf:6.1f} us | C=8 {ts[8]:6.1f}  C=64 {ts[64]:6.1f}  C=128 {ts[128]:6.1f} us | "
          f"marginal {2*B*M*9*H*W/slope/1e6:6.1f} TF  intercept {ts[128]-128*slope:6.1f} us")
