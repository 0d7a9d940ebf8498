"""Split-K sweep for the transposed (parity-class) conv shapes of the B=16 step."""
import sys; sys.path.insert(0, '.')
import torch
from textboxgan_amd import ops
dev = torch.device('cuda:0')
shapes = [  # C, M, Hin, Win, Hout, Wout, stride
    (128, 128, 16, 64, 34, 130, (2, 2)), (128, 64, 32, 128, 66, 258, (2, 2)), (256, 128, 8, 32, 18, 66, (2, 2)),
    (256, 256, 8, 16, 10, 34, (1, 2)), (512, 256, 4, 8, 10, 18, (2, 2)), (128, 128, 32, 128, 65, 257, (2, 2)),
    (256, 128, 16, 64, 33, 129, (2, 2)), (256, 256, 8, 32, 17, 65, (2, 2)), (512, 512, 4, 4, 6, 10, (1, 2)),
    (512, 256, 4, 16, 9, 33, (2, 2)), (512, 512, 2, 8, 5, 17, (2, 2)),
    # non-transposed small layers (3x3 s1)
    (512, 512, 4, 16, 4, 16, None), (512, 512, 2, 8, 2, 8, None), (256, 256, 8, 32, 8, 32, None), (512, 512, 4, 8, 4, 8, None),
    # mid layers whose tile count is at or below one block per CU
    (256, 256, 16, 64, 16, 64, None), (128, 128, 16, 64, 16, 64, None), (128, 128, 32, 128, 32, 128, None),
]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
_scope = ops.compute_dtype(sys.argv[2] if len(sys.argv) > 2 else "f32"); _scope.__enter__()
for C, M, Hin, Win, Hout, Wout, st in shapes:
    x = torch.randn(B, C, Hin, Win, device=dev); w = ops.pack_filter(torch.randn(9, C, M, device=dev), False, False)
    row = []
    for ks in (None, 1, 2, 3, 4, 6, 8, 16):
        ops.TUNING.force_ksplit = ks
        f = (lambda: ops.conv2d_raw(x, w, M, 3, 3, (Hout, Wout), st, (0, 0), transposed=True)) if st else \
            (lambda: ops.conv2d_raw(x, w, M, 3, 3, (Hout, Wout), (1, 1), (1, 1)))
        for _ in range(3): f()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        row.append(f"{'auto' if ks is None else ks}:{e0.elapsed_time(e1) / 20 * 1e3:6.1f}")
    print(f"C={C} M={M} {Hin}x{Win}->{Hout}x{Wout} T={st}  " + "  ".join(row))
