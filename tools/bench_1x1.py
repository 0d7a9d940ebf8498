"""isolated time of the small-map 1x1 / few-tap launches (f32x3): python tools/bench_1x1.py [lib.so]"""
import sys; sys.path.insert(0, '.')
import torch
from textboxgan_amd import native
if len(sys.argv) > 1:
    native.LIB_PATH = sys.argv[1]
from textboxgan_amd import ops
dev = torch.device('cuda:0')
print(native.LIB_PATH)
with ops.compute_dtype("f32x3"):
    for B, C, M, H, W in ((16, 256, 256, 2, 25), (16, 128, 128, 4, 25), (16, 512, 512, 1, 25), (16, 64, 64, 8, 25), (32, 256, 512, 4, 8),
                          (32, 128, 256, 8, 32)):
        x = torch.randn(B, C, H, W, device=dev); w = ops.pack_filter(torch.randn(1, C, M, device=dev), False, False)
        f = lambda: ops.conv2d_raw(x, w, M, 1, 1, (H, W), (1, 1), (0, 0))
        for _ in range(5): f()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(50): f()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        print(f"1x1 B={B} C={C} M={M} {H}x{W}: {e0.elapsed_time(e1) / 50 * 1e3:6.1f} us per call (conv + split epilogue, graph replay of 50)")
