"""The frozen BiLSTM layer (B x 25 x 512 -> 2 x 256) forward + backward in graph replay: fused step kernels against the
library-GEMM + pointwise pair.  usage: python tools/bench_lstm.py [B]"""
import sys; sys.path.insert(0, '.')
import torch
from textboxgan_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
if len(sys.argv) > 2:
    import os
    from textboxgan_amd import native
    native.LIB_PATH = os.path.abspath(sys.argv[2])
dev = torch.device('cuda:0')
T, In, H, D = 25, 512, 256, 2
g = torch.Generator().manual_seed(0)
r = lambda *s: (torch.randn(*s, generator=g) * 0.05).to(dev)
w_ih, w_hh, b = r(D, 4 * H, In), r(D, 4 * H, H), r(D, 4 * H)
w_hhT = w_hh.transpose(1, 2).contiguous()
x = r(B, T, In).requires_grad_(True)
dy = r(B, T, D * H)
for fused in (True, False):
    ops.TUNING.fused_lstm = fused
    def f():
        y = ops.frozen_bilstm_layer(x, w_ih, w_hh, b, w_hhT)
        return torch.autograd.grad(y, x, dy)[0]
    for _ in range(3): f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(4): out = f()
    gr.replay(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): gr.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"B={B} fused={fused}: {e0.elapsed_time(e1) / 5 / 4 * 1e3:7.1f} us per layer forward + backward ({2 * T} recurrent steps)", flush=True)
