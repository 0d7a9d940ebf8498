"""Experiment: f32x3 forward convs of the mid-size layers, cache-hot (the same launch repeated) vs cache-cold (a 1 GiB fill
between launches, timed by events around the conv only), by K split."""
import sys; sys.path.insert(0, '.')
import torch
from textboxgan_amd import ops
dev = torch.device('cuda:0')
B = 16
junk = torch.empty(256 * 1024 * 1024, device=dev)


def t_hot(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def t_cold(fn, n=12):
    tot = 0.0
    for i in range(n + 2):
        junk.fill_(float(i))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if i >= 2: tot += e0.elapsed_time(e1)
    return tot / n * 1e3


L = [("16x64 256->256", 256, 256, 16, 64, (1, 1)), ("32x128 128->128", 128, 128, 32, 128, (1, 1)),
     ("8x32 256->256", 256, 256, 8, 32, (1, 1)), ("16x64 128->128", 128, 128, 16, 64, (1, 1)),
     ("down 65x257 128->128", 128, 128, 65, 257, (2, 2)), ("down 33x129 128->256", 128, 256, 33, 129, (2, 2)),
     ("64x256 128->128", 128, 128, 64, 256, (1, 1))]
for name, C, M, H, W, stride in L:
    x = torch.randn(B, C, H, W, device=dev)
    wp = ops.pack_filter(torch.randn(3, 3, C, M, device=dev), False, False, bf16="f32x3")
    pad = (1, 1) if stride == (1, 1) else (0, 0)
    ohw = ((H + 2 * pad[0] - 3) // stride[0] + 1, (W + 2 * pad[1] - 3) // stride[1] + 1)
    flops = 2.0 * B * C * M * 9 * ohw[0] * ohw[1]
    row = f"{name:24s}"
    for ks in (None, 1, 2, 4):
        ops.TUNING.force_ksplit = ks
        fn = lambda: ops.conv2d_raw(x, wp, M, 3, 3, ohw, stride, pad)
        h, c = t_hot(fn), t_cold(fn)
        row += f"  ks={ks}: hot {h:6.1f}us {flops / h / 1e6:5.1f}TF cold {c:6.1f}us {flops / c / 1e6:5.1f}TF |"
    ops.TUNING.force_ksplit = None
    print(row, flush=True)
