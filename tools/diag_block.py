import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import math, torch
from oracle import ref_ops as R
from textboxgan_amd.config import small_config
from textboxgan_amd.models import ModulatedConv2D, ToRGB
from test_layers_gpu import rnd, rel_err, l2_err
dev = torch.device('cuda:0'); cfg = small_config(4)
sd = cfg.style_dim
for up, k, demod in ((True, 3, True), (False, 3, True), (False, 1, False)):
    B, I, O, H, W = 4, 16, (3 if k == 1 else 16), 8, 32
    x, style = rnd(B, I, H, W, seed=1), rnd(B, sd, seed=2)
    w, mw, mb = rnd(k, k, I, O, seed=3), rnd(sd, I, seed=4), rnd(I, seed=5) * 0.1
    leaves = [t.requires_grad_(True) for t in (x, w, mw, mb)]
    y = R.t_modulated_conv2d(x, style, w, mw, mb, up=up, demodulate=demod, fused=False)
    dout = rnd(*y.shape, seed=8)
    grads = torch.autograd.grad(y, leaves, dout)
    m = ModulatedConv2D(cfg, I, O, k, up, demod).to(dev)
    with torch.no_grad():
        m.w.copy_(w.float()); m.mod_dense.w.copy_(mw.float()); m.mod_bias.b.copy_(mb.float())
    xd = x.detach().float().to(dev).requires_grad_(True)
    s = m.style(style.float().to(dev)); d = m.demod(s, "composable")
    yd = m.conv_composable(xd, s, d)
    gd = torch.autograd.grad(yd, (xd, m.w, m.mod_dense.w, m.mod_bias.b), dout.float().to(dev))
    print('up' if up else 'k%d' % k, 'fwd', rel_err(yd, y), [('%.1e' % l2_err(a, b)) for a, b in zip(gd, grads)])
