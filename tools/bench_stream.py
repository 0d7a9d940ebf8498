"""GB/s of the HBM-bound kernels on the step's large shapes (algorithmic bytes / time)."""
import sys; sys.path.insert(0, '.')
import torch
from textboxgan_amd import ops, native as N
dev = torch.device('cuda:0')
def timeit(f, n=30):
    for _ in range(3): f()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = 16
for (C, H, W) in ((128, 64, 256), (64, 64, 256), (128, 32, 128), (256, 16, 64), (512, 4, 16)):
    x = torch.randn(B, C, H, W, device=dev); dout = torch.randn_like(x)
    b = torch.randn(C, device=dev); nz = torch.randn(B, 1, H, W, device=dev); st = torch.tensor(0.1, device=dev)
    d = torch.rand(B, C, device=dev) + 0.5
    epi = ops._lrelu_epi(out_scale=d, bias=b, noise=nz, strength=st, alpha=1.0)
    out = ops.bias_act_fwd_raw(x, epi)
    nb = x.numel() * 4
    t = timeit(lambda: ops.bias_act_fwd_raw(x, epi))
    print(f"[{B},{C},{H},{W}] bias_act_fwd  {t:7.1f} us  {2*nb/t/1e3:6.0f} GB/s", end="   ")
    t = timeit(lambda: ops.bias_act_bwd_raw(dout, out, epi, want_dn=True, want_dyy=True))
    print(f"bias_act_bwd {t:7.1f} us  {3*nb/t/1e3:6.0f} GB/s", end="   ")
    k = ops.fir_kernel(dev, 1.0)
    t = timeit(lambda: ops.upfirdn2d_raw(x, k, pad=(2, 1, 2, 1)))
    print(f"blur {t:7.1f} us  {2*nb/t/1e3:6.0f} GB/s", end="   ")
    xe = torch.randn(B, C, H + 1, W + 1, device=dev)
    t2 = timeit(lambda: ops.upfirdn2d_raw(xe, k, pad=(1, 1, 1, 1), epi=epi))
    print(f"blur+epi {t2:7.1f} us  {2*nb/t2/1e3:6.0f} GB/s", end="   ")
    t = timeit(lambda: ops.upfirdn2d_raw(x, k, down=(2, 2), pad=(2, 1, 2, 1)))
    print(f"blur/2 {t:7.1f} us  {1.25*nb/t/1e3:6.0f} GB/s", end="   ")
    y = torch.empty_like(x)
    t = timeit(lambda: y.copy_(x))
    print(f"copy {t:7.1f} us  {2*nb/t/1e3:6.0f} GB/s")
