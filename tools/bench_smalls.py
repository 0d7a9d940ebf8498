"""isolated time of tbg_modconv_bwd_smalls_f32 at the step's sizes: python tools/bench_smalls.py"""
import sys; sys.path.insert(0, '.')
import torch
from textboxgan_amd import ops
dev = torch.device('cuda:0')
for B, I, O, nch in ((16, 512, 512, 1), (16, 256, 256, 1), (16, 128, 128, 3), (32, 512, 512, 1), (32, 128, 128, 3), (8, 128, 128, 3)):
    r = lambda *s: torch.randn(*s, device=dev)
    pdb, pdn, pdy, d, s, wsq, dsc = r(B, O, nch), r(B, O, nch), r(B, O, nch), r(B, O).abs() + 0.5, r(B, I), r(I, O).abs(), r(B, I)
    f = lambda: ops.modconv_bwd_smalls_raw(pdb, pdn, pdy, d, s, wsq, dsc)
    for _ in range(5): f()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(50): f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"B={B} I={I} O={O} nch={nch}: {e0.elapsed_time(e1) / 50 * 1e3:6.1f} us per launch (graph replay of 50)")
