"""How long do prologue + epilogue of the fprop kernel take?  C = 8 (one chunk) vs a plain fill of the same output."""
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
B, M, H, W = 16, 128, 64, 256
y = torch.empty(B, M, H, W, device=dev)
print("fill 134 MB: %.1f us" % timeit(lambda: y.fill_(1.0)))
src = torch.randn(B, M, H, W, device=dev)
print("copy 134 MB: %.1f us" % timeit(lambda: y.copy_(src)))
for C in (4, 8, 16, 32, 64, 128):
    x = torch.randn(B, C, H, W, device=dev); w = ops.pack_filter(torch.randn(9, C, M, device=dev), False, False)
    t = timeit(lambda: ops.conv2d_raw(x, w, M, 3, 3, (H, W), (1, 1), (1, 1)))
    print(f"conv C={C:3d}: {t:7.1f} us   ({2*B*C*M*9*H*W/t/1e6:6.1f} TF)")
