"""dense layer launch forms: one-launch kernels vs library GEMM chain, at the step's sizes (GPU box)."""
import sys; sys.path.insert(0, '.')
import math, torch
from textboxgan_amd import ops
dev = torch.device('cuda:0')
def timeit(fn, n=200):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for R, K, Nn, lrelu in ((32, 256, 256, True), (16, 256, 512, False), (16, 256, 128, False), (16, 512, 1, False), (64, 256, 256, True)):
    x = torch.randn(R, K, device=dev, requires_grad=True); w = torch.randn(K, Nn, device=dev, requires_grad=True)
    b = torch.randn(Nn, device=dev, requires_grad=True); dout = torch.randn(R, Nn, device=dev)
    line = f"R={R} K={K} N={Nn} lrelu={lrelu}: "
    for name, fn in (("one-launch", ops._DenseBiasAct), ("library", ops._DenseBiasActGemm)):
        g = torch.cuda.CUDAGraph()
        def step():
            out = fn.apply(x, w, b, 1.0 / math.sqrt(K), 1.0, lrelu, 0.0)
            return torch.autograd.grad(out, (x, w, b), dout)
        step(); torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(10): step()
        us = timeit(g.replay, 50) / 10
        with torch.no_grad():
            fwd = lambda: fn.apply(x, w, b, 1.0 / math.sqrt(K), 1.0, lrelu, 0.0)
            g2 = torch.cuda.CUDAGraph(); fwd(); torch.cuda.synchronize()
            with torch.cuda.graph(g2):
                for _ in range(10): fwd()
            us_f = timeit(g2.replay, 50) / 10
        line += f"{name}: fwd {us_f:6.1f} us, fwd+bwd {us:6.1f} us (graph replay)   "
    print(line)
