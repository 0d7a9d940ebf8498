"""Split-K x tile sweep for the small-map convolution shapes of a step (frozen-OCR ResNet stages, the discriminator's last
block, B = 16): conv2d_raw (convolution + split-K second half) in graph replay, us per call.
usage: python tools/bench_ksplit_ocr.py [f32|bf16|f32x3] [variants, e.g. 0,10,11,13]   (tbg_conv2d_*_variant: 10 = 32 x 256,
11 = 64 x 256, 12 = 64 x 64, 13 = 128 x 128 channel x pixel tiles)"""
import sys; sys.path.insert(0, '.')
import torch
from textboxgan_amd import ops
ops._TLS.compute = sys.argv[1] if len(sys.argv) > 1 else "f32x3"
variants = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0").split(",")]
dev = torch.device('cuda:0')
shapes = [(256, 256, 2, 25, 3), (128, 128, 4, 25, 3), (512, 512, 1, 25, 3), (512, 512, 4, 16, 3), (64, 64, 8, 25, 3),
          (32, 32, 16, 50, 3), (256, 256, 2, 25, 1), (128, 128, 4, 25, 1), (256, 256, 8, 32, 3)]
B = 16
for C, M, H, W, k in shapes:
    x = torch.randn(B, C, H, W, device=dev); w = ops.pack_filter(torch.randn(k * k, C, M, device=dev), False, False)
    for v in variants:
        ops.TUNING.force_variant = v
        row = []
        for ks in (None, 1, 2, 4, 8, 16, 32):
            ops.TUNING.force_ksplit = ks
            f = lambda: ops.conv2d_raw(x, w, M, k, k, (H, W), (1, 1), (k // 2, k // 2))
            try:
                for _ in range(3): f()
            except Exception as e:
                row.append(f"{'auto' if ks is None else ks}:   n/a"); continue
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()  # graph replay: the step runs captured, eager timing of 10-us kernels is host bound
            with torch.cuda.graph(g):
                for _ in range(20): f()
            g.replay(); torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): g.replay()
            e1.record(); torch.cuda.synchronize()
            row.append(f"{'auto' if ks is None else ks}:{e0.elapsed_time(e1) / 5 / 20 * 1e3:6.1f}")
        print(f"C={C} M={M} {H}x{W} k={k} v={v:2d}  " + "  ".join(row), flush=True)
