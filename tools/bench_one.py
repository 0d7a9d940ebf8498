"""Run ONE conv shape repeatedly (for rocprofv3 --pmc). usage: bench_one.py kind [n]"""
import sys; sys.path.insert(0, '.')
import torch
from textboxgan_amd import ops, native as N
dev = torch.device('cuda:0')
kind = sys.argv[1] if len(sys.argv) > 1 else "fwd"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B, C, M, H, W = 16, 128, 128, 64, 256
x = torch.randn(B, C, H, W, device=dev); w = torch.randn(3, 3, C, M, device=dev)
wp = ops.pack_filter(w, False, False)
g = ops._Geom((1, 1), (1, 1), 3, 3, (H, W), (H, W))
if kind == "fwd":
    fn = lambda: ops.conv2d_raw(x, wp, M, 3, 3, (H, W), (1, 1), (1, 1))
elif kind == "fwdmod":
    s = torch.rand(B, C, device=dev) + 0.5; d = torch.rand(B, M, device=dev) + 0.5
    nz = torch.randn(B, 1, H, W, device=dev); st = torch.tensor(0.1, device=dev); b = torch.randn(M, device=dev)
    epi = N.epilogue(out_scale=d, bias=b, noise=nz, strength=st, act=N.ACT_LRELU, alpha=0.03)
    fn = lambda: ops.conv2d_raw(x, wp, M, 3, 3, (H, W), (1, 1), (1, 1), in_scale=s, epi=epi)
elif kind == "dgradmod":   # data gradient of a modulated conv: in_scale = d, out_scale = s, fused style-gradient dot
    s = torch.rand(B, C, device=dev) + 0.5; d = torch.rand(B, M, device=dev) + 0.5
    dy = torch.randn(B, M, H, W, device=dev); ds = torch.zeros(B, C, device=dev)
    wt = ops.pack_filter(w, True, True)
    epi = N.epilogue(alpha=0.03, out_scale=s)
    fn = lambda: ops.conv2d_raw(dy, wt, C, 3, 3, (H, W), (1, 1), (1, 1), in_scale=d, epi=epi, dot=(x, ds))
elif kind == "dgradscale":
    s = torch.rand(B, C, device=dev) + 0.5; d = torch.rand(B, M, device=dev) + 0.5
    dy = torch.randn(B, M, H, W, device=dev); wt = ops.pack_filter(w, True, True)
    epi = N.epilogue(alpha=0.03, out_scale=s)
    fn = lambda: ops.conv2d_raw(dy, wt, C, 3, 3, (H, W), (1, 1), (1, 1), in_scale=d, epi=epi)
elif kind == "dgrad":
    dy = torch.randn(B, M, H, W, device=dev); wt = ops.pack_filter(w, True, True)
    fn = lambda: ops.conv2d_raw(dy, wt, C, 3, 3, (H, W), (1, 1), (1, 1))
elif kind == "wgrad":
    dy = torch.randn(B, M, H, W, device=dev)
    fn = lambda: ops._bwd_weight_launch(x, dy, g, C, M)
for _ in range(n): fn()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n): fn()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
print(kind, "ms %.3f  TF %.1f" % (ms, 2 * B * C * M * 9 * H * W / ms / 1e9))
