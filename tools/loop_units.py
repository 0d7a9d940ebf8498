"""the two dominant unit-tensor kernels back to back for a few seconds each (clock / power sampling: tools/power_trace.sh)"""
import sys, time, torch
sys.path.insert(0, ".")
from textboxgan_amd import ops, native as N
dev = torch.device("cuda:0")
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
B, C, M, H, W = 16, 128, 128, 64, 256
x, dy = torch.randn(B, C, H, W, device=dev), torch.randn(B, M, H, W, device=dev)
w = torch.randn(3, 3, C, M, device=dev) / (9 * C) ** 0.5
with ops.compute_dtype("f32x3"):
    XU, DU = ops.units_pack(x), ops.units_pack(dy)
    pf = ops.pack_filter(w, False, False)
    dw = torch.empty(3, 3, C, M, device=dev)
    for name, fn in (("conv_units_fprop_kernel<3, 2, 0>", lambda: ops.conv2d_units_raw(XU, pf, M, epi=ops._lrelu_epi(alpha=0.1))),
                     ("conv_wgrad_units_kernel<3>", lambda: ops.wgrad_units_raw(DU, XU, dw, C * M, M, 1, 1.0))):
        torch.cuda.synchronize(); t0 = time.time(); n = 0
        while time.time() - t0 < secs:
            for _ in range(50): fn()
            torch.cuda.synchronize(); n += 50
        dt = time.time() - t0
        print(f"{name}: {n} launches in {dt:.2f} s = {dt / n * 1e6:.1f} us / launch = {2.0 * B * C * M * H * W * 9 / (dt / n) / 1e12:.1f} TFLOP/s", flush=True)
