import sys; sys.path.insert(0, '.')
import math, torch
import torch.nn.functional as F
from oracle import ref_ops as R
from textboxgan_amd import ops, native as N
dev = torch.device('cuda:0')
def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64)
def rel(a, r): return float((a.double().cpu() - r.detach()).abs().max() / (r.detach().abs().max() + 1e-30))
B, I, O, H, W = 2, 128, 128, 16, 64
for variant in (4, 0):
    ops.FORCE_VARIANT = variant
    x = rnd(B, I, H, W, seed=1).requires_grad_(True); w = rnd(3, 3, I, O, seed=3); s = (rnd(B, I, seed=9) + 1.0)
    noise, strength, bias = rnd(B, 1, 2 * H, 2 * W, seed=6), torch.tensor(0.3, dtype=torch.float64), rnd(O, seed=7) * 0.2
    coef = 1 / math.sqrt(9 * I)
    wc = w * coef
    ww = wc[None] * s[:, None, None, :, None]
    d = torch.rsqrt(ww.square().sum(dim=(1, 2, 3)) + 1e-8)
    xs = x * s[:, :, None, None]
    y_up = F.conv_transpose2d(xs, torch.flip(wc, (0, 1)).permute(2, 3, 0, 1), stride=2); y_up.retain_grad()
    k = R.setup_kernel([1, 3, 3, 1]) * 4
    yb = R.t_simple_upfirdn2d(y_up, k, pad0=1, pad1=1)
    pre = yb * d[:, :, None, None] + noise * strength + bias[None, :, None, None]; pre.retain_grad()
    out = F.leaky_relu(pre, 0.2) * math.sqrt(2)
    dout = rnd(*out.shape, seed=8)
    out.backward(dout)
    f = lambda t: t.detach().float().to(dev).contiguous()
    xd, wd, sd, nd, std, bd = f(x).requires_grad_(True), f(w), f(s), f(noise), f(strength), f(bias)
    outd = ops.modconv_up_fused(xd, wd, sd, nd, std, bd)
    print("variant", variant, "out", rel(outd, out))
    (dx,) = torch.autograd.grad(outd, xd, f(dout))
    print("   dx (fused node)", rel(dx, x.grad))
    # manual pieces
    dd, wsq = ops.demod_coefs_raw(sd, wd, coef)
    epi = ops._lrelu_epi(out_scale=dd, bias=bd, noise=nd, strength=std, alpha=1.0)
    _, dpre, pdb, pdn, pdy = ops.bias_act_bwd_raw(f(dout), outd.detach(), epi, want_dn=True, want_dyy=True)
    print("   dpre", rel(dpre, pre.grad))
    kk = ops.fir_kernel(dev, gain=4.0)
    dy_up = ops.upfirdn2d_raw(dpre, kk, pad=(2, 2, 2, 2), in_scale=dd.reshape(-1))
    print("   dy_up", rel(dy_up, y_up.grad), tuple(dy_up.shape))
    wt = ops.pack_filter(wd, transpose=True, flip=True)
    ds = torch.zeros_like(sd)
    dx2 = ops.conv2d_raw(dy_up, wt, I, 3, 3, (H, W), (2, 2), (0, 0), epi=N.epilogue(alpha=coef, out_scale=sd), dot=(xd.detach(), ds))
    print("   dx (manual)", rel(dx2, x.grad))
    dx3 = ops.conv2d_raw(f(y_up.grad), wt, I, 3, 3, (H, W), (2, 2), (0, 0), epi=N.epilogue(alpha=coef, out_scale=sd))
    print("   dx from reference dy_up", rel(dx3, x.grad))
ops.FORCE_VARIANT = 0
