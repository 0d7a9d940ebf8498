import sys; sys.path.insert(0, '.')
import math, torch
from oracle import ref_ops as R
from textboxgan_amd import ops, native as N
dev = torch.device('cuda:0')
def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64)
def rel(a, r): return float((a.double().cpu() - r).abs().max() / (r.abs().max() + 1e-30))
B, I, O, H, W = 2, 128, 128, 16, 64
for variant in (4, 0, 0, 4):
    ops.FORCE_VARIANT = variant
    x = rnd(B, I, H, W, seed=1); w = rnd(3, 3, I, O, seed=3); s = rnd(B, I, seed=9) + 1.0
    coef = 1 / math.sqrt(9 * I)
    f = lambda t: t.float().to(dev).contiguous()
    xd, wd, sd = f(x), f(w), f(s)
    yref = torch.nn.functional.conv_transpose2d(x * s[:, :, None, None], torch.flip(w, (0, 1)).permute(2, 3, 0, 1), stride=2) * coef
    # poison memory around: allocate guard tensors
    g0 = torch.full((1 << 20,), 7.0, device=dev)
    y = ops.conv2d_raw(xd, ops.pack_filter(wd, False, False), O, 3, 3, (2 * H + 1, 2 * W + 1), (2, 2), (0, 0), transposed=True,
                       flip=True, in_scale=sd, epi=N.epilogue(alpha=coef))
    g1 = torch.full((1 << 20,), 7.0, device=dev)
    torch.cuda.synchronize()
    print("variant", variant, "fwd err", rel(y, yref), "guards", float(g0.min()), float(g0.max()), float(g1.min()), float(g1.max()))
    dy = rnd(B, O, 2 * H + 1, 2 * W + 1, seed=5)
    dyd = f(dy)
    wt = ops.pack_filter(wd, transpose=True, flip=True)
    ds = torch.zeros_like(sd)
    dx = ops.conv2d_raw(dyd, wt, I, 3, 3, (H, W), (2, 2), (0, 0), epi=N.epilogue(alpha=coef, out_scale=sd), dot=(xd, ds))
    dxref = torch.nn.functional.conv2d(dy, torch.flip(w, (0, 1)).permute(2, 3, 0, 1).transpose(0, 1).flip(2, 3) if False else w.permute(2, 3, 0, 1).flip(2,3).flip(2,3), stride=2) if False else None
    # reference via autograd of the transposed conv
    xs = (x * s[:, :, None, None]).requires_grad_(True)
    yy = torch.nn.functional.conv_transpose2d(xs, torch.flip(w, (0, 1)).permute(2, 3, 0, 1), stride=2) * coef
    (gxs,) = torch.autograd.grad(yy, xs, dy)
    print("   dgrad err", rel(dx, gxs * s[:, :, None, None]), "ds err", rel(ds, (gxs * x).sum(dim=(2, 3))))
ops.FORCE_VARIANT = 0
