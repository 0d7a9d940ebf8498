"""How exact is fp32 convolution emulated by bf16 operand splitting?  CPU experiment (float64 truth):
x = x1 + x2 + x3, w = w1 + w2 + w3 with bf16 planes; products kept: 6 of 9 (drop x2*w3, x3*w2, x3*w3) or all 9;
accumulation in fp32 (as the MFMA does).  usage: python tools/exp_bf16x3_accuracy.py"""
import torch, torch.nn.functional as F
torch.manual_seed(0)
def split3(t):
    a = t.bfloat16().float(); r = t - a
    b = r.bfloat16().float(); r2 = r - b
    c = r2.bfloat16().float()
    return a, b, c
B, C, M, H, W = 2, 128, 128, 32, 64
x = torch.randn(B, C, H, W); w = torch.randn(M, C, 3, 3) / (9 * C) ** 0.5
truth = F.conv2d(x.double(), w.double(), padding=1)
rel = lambda y: float((y.double() - truth).abs().max() / truth.abs().max())
rms = lambda y: float(((y.double() - truth).pow(2).mean() / truth.pow(2).mean()).sqrt())
xs, ws = split3(x), split3(w)
assert float((xs[0] + xs[1] + xs[2] - x).abs().max()) == 0.0 and float((ws[0] + ws[1] + ws[2] - w).abs().max()) == 0.0, "3 bf16 planes hold an fp32 value exactly"
conv = lambda a, b: F.conv2d(a, b, padding=1)  # fp32 accumulate (products of two bf16 values are exact in fp32)
pairs6 = [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1)]
pairs9 = [(i, j) for i in range(3) for j in range(3)]
y6 = sum(conv(xs[i], ws[j]) for i, j in sorted(pairs6, key=lambda p: -(p[0] + p[1])))  # small terms first
y9 = sum(conv(xs[i], ws[j]) for i, j in sorted(pairs9, key=lambda p: -(p[0] + p[1])))
y32 = conv(x, w)
ybf = conv(xs[0], ws[0])
print(f"conv {C}->{M} {H}x{W}, K = {9*C}: max-abs / max and relative RMS error against float64")
for name, y in (("fp32 (oneDNN)", y32), ("bf16x3, 9 products", y9), ("bf16x3, 6 products", y6), ("bf16 (1 product)", ybf)):
    print(f"  {name:22s} max {rel(y):.3e}   rms {rms(y):.3e}")
