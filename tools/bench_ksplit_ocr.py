"""Split-K sweep for the small frozen-OCR conv shapes (B=16)."""
import sys; sys.path.insert(0, '.')
import torch
from textboxgan_amd import ops
ops._TLS.compute = sys.argv[1] if len(sys.argv) > 1 else "f32"
ops.TUNING.force_variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device('cuda:0')
shapes = [(256, 256, 2, 25, 3), (512, 512, 1, 25, 3), (128, 128, 4, 25, 3), (512, 512, 4, 8, 3), (256, 256, 8, 16, 3),
          (32, 32, 16, 50, 3), (64, 64, 8, 25, 3), (256, 256, 2, 25, 1), (128, 128, 4, 25, 1), (256, 256, 8, 16, 1), (32, 32, 32, 100, 3)]
B = 16
for C, M, H, W, k in shapes:
    x = torch.randn(B, C, H, W, device=dev); w = ops.pack_filter(torch.randn(k * k, C, M, device=dev), False, False)
    row = []
    for ks in (None, 1, 2, 4, 8, 16, 32, 64):
        ops.TUNING.force_ksplit = ks
        f = lambda: ops.conv2d_raw(x, w, M, k, k, (H, W), (1, 1), (k // 2, k // 2))
        for _ in range(3): f()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()  # graph replay: the step runs captured, eager timing of 10-us kernels is host bound
        with torch.cuda.graph(g):
            for _ in range(20): f()
        g.replay(); torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): g.replay()
        e1.record(); torch.cuda.synchronize()
        e0t = e0.elapsed_time(e1) / 5
        class _T:  # keep the row code below
            pass
        row.append(f"{'auto' if ks is None else ks}:{e0t / 20 * 1e3:6.1f}")
    print(f"C={C} M={M} {H}x{W} k={k}  " + "  ".join(row))
