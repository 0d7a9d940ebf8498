import sys; sys.path.insert(0, '.')
import torch
from textboxgan_amd import ops
if len(sys.argv) > 1:  # a variant build of the library (tools/build_variant.sh)
    import os
    from textboxgan_amd import native
    native.LIB_PATH = os.path.abspath(sys.argv[1])
dev = torch.device('cuda:0')
def timeit(f, n=30):
    for _ in range(5): f()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = 16
for (C, H, W) in ((128, 64, 256), (128, 32, 128), (256, 16, 64), (64, 64, 256)):
    x = torch.randn(B, C, H, W, device=dev); w = torch.randn(C, 3, device=dev); s = torch.rand(B, C, device=dev)
    b = torch.randn(3, device=dev); skip = torch.randn(B, 3, H, W, device=dev); dy = torch.randn(B, 3, H, W, device=dev)
    nb = x.numel() * 4
    t1 = timeit(lambda: ops.rgb_project_raw(x, w, 3, s, b, skip, 0.1))
    t2 = timeit(lambda: ops.rgb_backproject_raw(x, dy, w, s, 0.1, want_dx=True, want_G=True))
    t3 = timeit(lambda: ops.rgb_backproject_raw(x, dy, w, s, 0.1, want_dx=False, want_G=True))
    print(f"[{B},{C},{H},{W}] project {t1:7.1f} us {nb/t1/1e3:6.0f} GB/s | backproject dx+G {t2:7.1f} us {2*nb/t2/1e3:6.0f} GB/s | G only {t3:7.1f} us {nb/t3/1e3:6.0f} GB/s")
