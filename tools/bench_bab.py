"""bias_act_bwd_units (activation backward writing a unit tensor) in graph replay, us per call and TB/s:
python tools/bench_bab.py [f32x3|bf16] [B] [lib.so]"""
import sys; sys.path.insert(0, '.')
import torch
from textboxgan_amd import native as N, ops
mode = sys.argv[1] if len(sys.argv) > 1 else "f32x3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
if len(sys.argv) > 3 and sys.argv[3] != "-":
    import os
    N.LIB_PATH = os.path.abspath(sys.argv[3])
ops._TLS.compute = mode
dev = torch.device('cuda:0')


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


tot = 0.0
for (M, H, W) in [(128, 64, 256), (64, 64, 256), (128, 32, 128), (256, 32, 128), (256, 16, 64), (512, 16, 64), (512, 8, 32), (512, 4, 16)]:
    dout, act = torch.randn(B, M, H, W, device=dev), torch.randn(B, M, H, W, device=dev)
    noise = torch.randn(B, 1, H, W, device=dev)
    st = torch.ones(1, device=dev)
    bias = torch.randn(M, device=dev)
    sc = torch.rand(B, M, device=dev) + 0.5
    epi = N.epilogue(bias=bias, noise=noise, strength=st, act=N.ACT_LRELU, out_scale=sc)
    planes = 3 if mode == "f32x3" else 1
    t = timed(lambda: ops.bias_act_bwd_units_raw(dout, act, epi, planes=planes, want_db=True, want_dn=True))
    nb = 8.0 * dout.numel() + 2.0 * planes * B * M * (H + 2) * (W + 2) + 4.0 * B * H * W
    tot += t
    print(f"{M:4d} x {H:3d}x{W:3d}: {t:8.1f} us  {nb / t / 1e6:6.2f} TB/s")
print(f"sum {tot:.1f} us")
