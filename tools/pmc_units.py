"""a few launches of each unit-tensor kernel on the largest layer, for rocprofv3 --pmc (tools/pmc_units.sh)"""
import sys, torch
sys.path.insert(0, ".")
from textboxgan_amd import ops, native as N
if len(sys.argv) > 1:  # a variant build of the library (tools/build_variant.sh)
    import os
    N.LIB_PATH = os.path.abspath(sys.argv[1])
dev = torch.device("cuda:0")
B, C, M, H, W = 16, 128, 128, 64, 256
x, dy = torch.randn(B, C, H, W, device=dev), torch.randn(B, M, H, W, device=dev)
xs, ds = torch.rand(B, C, device=dev) + 0.5, torch.rand(B, M, device=dev) + 0.5
w = torch.randn(3, 3, C, M, device=dev) / (9 * C) ** 0.5
nz, bs, st = torch.randn(B, 1, H, W, device=dev), torch.randn(M, device=dev), torch.tensor(0.1, device=dev)
dw, out = torch.empty(3, 3, C, M, device=dev), torch.empty(B, M, H, W, device=dev)
for mode, planes in (("f32x3", 3), ("bf16", 1)):
    with ops.compute_dtype(mode):
        pf = ops.pack_filter(w, False, False)
        XU, DU = ops.units_pack(x, xs, planes=planes), ops.units_pack(dy, ds, planes=planes)
        for _ in range(4):
            ops.conv2d_units_raw(XU, pf, M, epi=N.epilogue(out_scale=ds, bias=bs, noise=nz, strength=st, act=N.ACT_LRELU), out=out)
            ops.wgrad_units_raw(DU, XU, dw, C * M, M, 1, 1.0)
torch.cuda.synchronize()
