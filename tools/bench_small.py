"""Small-map convolutions: tbg_conv2d_units_small (K split inside the block, one launch) against the NCHW kernel's split-K pair
(convolution + tbg_slab_epilogue_f32), both with a real epilogue (bias + residual + ReLU), in graph replay, us per call.
usage: python tools/bench_small.py [f32x3|bf16] [B] [lib.so | -] [taps | rows]"""
import sys; sys.path.insert(0, '.')
import torch
from textboxgan_amd import native as N, ops
mode = sys.argv[1] if len(sys.argv) > 1 else "f32x3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
if len(sys.argv) > 3 and sys.argv[3] != "-":  # a variant build of the library (tools/build_variant.sh)
    import os
    N.LIB_PATH = os.path.abspath(sys.argv[3])
ops._TLS.compute = mode
dev = torch.device('cuda:0')
# (C, M, Hin, Win, k, stride, transposed)
shapes = [(256, 256, 2, 25, 3, (1, 1), 0), (128, 128, 4, 25, 3, (1, 1), 0), (512, 512, 1, 25, 3, (1, 1), 0), (64, 64, 8, 25, 3, (1, 1), 0),
          (32, 32, 16, 50, 3, (1, 1), 0), (512, 512, 4, 16, 3, (1, 1), 0), (256, 256, 8, 32, 3, (1, 1), 0), (256, 256, 8, 16, 3, (1, 1), 0),
          (512, 512, 4, 8, 3, (1, 1), 0), (512, 512, 4, 4, 3, (1, 1), 0),
          (256, 256, 2, 25, 1, (1, 1), 0), (128, 128, 4, 25, 1, (1, 1), 0), (512, 512, 1, 25, 1, (1, 1), 0), (64, 64, 8, 25, 1, (1, 1), 0),
          (128, 256, 4, 25, 1, (2, 1), 0), (256, 128, 2, 25, 1, (2, 1), 1), (64, 128, 8, 25, 1, (2, 1), 0), (32, 64, 16, 50, 1, (2, 2), 0),
          # tap-list forms: strided VALID 3x3 (behind a blur) and stride-2 transposed 3x3
          (256, 512, 9, 33, 3, (2, 2), 0), (512, 128, 5, 17, 3, (2, 2), 0), (256, 512, 10, 18, 3, (2, 2), 0), (512, 512, 6, 10, 3, (1, 2), 0),
          (256, 256, 17, 65, 3, (2, 2), 0), (512, 512, 4, 4, 3, (1, 2), 1), (512, 256, 4, 16, 3, (2, 2), 1), (512, 256, 4, 8, 3, (2, 2), 1),
          (128, 512, 2, 8, 3, (2, 2), 1), (256, 128, 8, 32, 3, (2, 2), 1)]
if len(sys.argv) > 4:
    shapes = [s for s in shapes if (s[4] > 1 and s[5] != (1, 1)) == (sys.argv[4] == "taps")]


def timed(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): f()
    g.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5 / n * 1e3


for C, M, H, W, k, st, tr in shapes:
    if tr:
        Ho, Wo = (H * st[0], (W - 1) * st[1] + 1) if k == 1 else ((H - 1) * st[0] + k, (W - 1) * st[1] + k)
    elif st == (1, 1) and k == 3:
        Ho, Wo = H, W
    else:
        Ho, Wo = (H - k) // st[0] + 1, (W - k) // st[1] + 1
    x = torch.randn(B, C, H, W, device=dev); w = ops.pack_filter(torch.randn(k * k, C, M, device=dev), False, False)
    bias, res = torch.randn(M, device=dev), torch.randn(B, M, Ho, Wo, device=dev)
    epi = lambda: N.epilogue(bias=bias, residual=res, res_first=1, act=N.ACT_LRELU, slope=0.0)
    XU = ops.units_pack(x)
    new = lambda: ops.conv2d_small_raw(XU, w, M, k, (Ho, Wo), st, bool(tr), epi=epi())
    pad = (1, 1) if (k == 3 and st == (1, 1)) else (0, 0)
    old = lambda: ops.conv2d_raw(x, w, M, k, k, (Ho, Wo), st, pad, transposed=bool(tr), epi=epi(), allow_small=False)
    pack = lambda: ops.units_pack(x)
    flops = 2.0 * B * M * C * k * k * (H * W if tr else Ho * Wo)
    t_new, t_old, t_pack = timed(new), timed(old), timed(pack)
    print(f"C={C:3d} M={M:3d} {H}x{W}->{Ho}x{Wo} k={k} s={st} T={tr}: small {t_new:6.1f} us ({flops / t_new * 1e-6:6.1f} TF)   "
          f"nchw+slab {t_old:6.1f} us   (units_pack of the input {t_pack:5.1f})", flush=True)
