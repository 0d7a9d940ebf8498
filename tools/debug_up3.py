import sys; sys.path.insert(0, '.')
import math, torch
from oracle import ref_ops as R
from textboxgan_amd import ops, native as N
dev = torch.device('cuda:0')
def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64)
B, I, O, H, W = 2, 128, 128, 16, 64
sd_ = 20
for variant in (4, 0):
    ops.FORCE_VARIANT = variant
    x, style = rnd(B, I, H, W, seed=1), rnd(B, sd_, seed=2)
    w, mw, mb = rnd(3, 3, I, O, seed=3), rnd(sd_, I, seed=4), rnd(I, seed=5) * 0.1
    Ho, Wo = 2 * H, 2 * W
    noise, strength, bias = rnd(B, 1, Ho, Wo, seed=6), torch.tensor(0.3, dtype=torch.float64), rnd(O, seed=7) * 0.2
    leaves = [t.requires_grad_(True) for t in (x, w, mw, mb, strength, bias)]
    y = R.t_modulated_conv2d(x, style, w, mw, mb, up=True, demodulate=True, fused=False)
    out = R.t_bias_act(R.t_noise(y, noise, strength), bias, "lrelu")
    dout = rnd(*out.shape, seed=8)
    grads = torch.autograd.grad(out, leaves, dout, retain_graph=True)
    f = lambda t: t.detach().float().to(dev).contiguous()
    xd, wd, nd, std, bd = f(x).requires_grad_(True), f(w).requires_grad_(True), f(noise), f(strength).requires_grad_(True), f(bias).requires_grad_(True)
    mwd, mbd = f(mw).requires_grad_(True), f(mb).requires_grad_(True)
    s = ops.dense_bias_act(f(style), mwd, mbd, 1.0 / math.sqrt(sd_), 1.0, lrelu=False, offset=1.0)
    outd = ops.modconv_up_fused(xd, wd, s, nd, std, bd)
    for wanted in ((xd,), (xd, wd), (xd, wd, mwd, mbd, std, bd)):
        gd = torch.autograd.grad(outd, wanted, f(dout), retain_graph=True)
        a, b = gd[0].double().cpu(), grads[0]
        err = (a - b).abs()
        idx = torch.nonzero(err > 1e-3 * b.abs().max())
        print("variant", variant, "wanted", len(wanted), "dx max err", float(err.max() / b.abs().max()), "n bad", idx.shape[0],
              "first bad", idx[:5].tolist(), "bad b", sorted(set(idx[:, 0].tolist())), "bad ch range",
              (int(idx[:, 1].min()), int(idx[:, 1].max())) if idx.numel() else None)
ops.FORCE_VARIANT = 0
