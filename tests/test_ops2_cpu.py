"""CPU: the batched style affines / demodulation coefficients of the twice-differentiable synthesis pass (ops2.synthesis_styles:
pure tensor ops) against the per-layer formulas of modulated_conv2d.py:74-82, values and first / second derivatives."""
import math

import torch

from textboxgan_amd import ops2


def _per_layer(style, rows, mw, mb, coef, dl, cw):
    s = [torch.addmm(mb[l] + 1.0, style[:, rows[l]], mw[l] * coef) for l in range(len(mw))]
    d = {l: torch.rsqrt(s[l].square() @ (cw[j].square().sum((0, 1)) / (9 * cw[j].shape[2])) + 1e-8) for j, l in enumerate(dl)}
    return s, d


def test_synthesis_styles_matches_per_layer_formulas():
    torch.manual_seed(0)
    B, K = 3, 32
    Is = [16, 16, 24, 8]
    style = torch.randn(B, 5, K, dtype=torch.float64, requires_grad=True)
    mw = [torch.randn(K, i, dtype=torch.float64, requires_grad=True) for i in Is]
    mb = [torch.randn(i, dtype=torch.float64, requires_grad=True) for i in Is]
    cw = [torch.randn(3, 3, 16, 24, dtype=torch.float64, requires_grad=True),
          torch.randn(3, 3, 24, 8, dtype=torch.float64, requires_grad=True)]
    rows, dl, coef = [0, 0, 1, 2], [1, 2], 1 / math.sqrt(K)  # two layers share latent row 0, as the first toRGB and conv do
    leaves = [style] + mw + mb + cw
    f = lambda ss, dd: sum((s ** 3).sum() for s in ss) + sum((d ** 2).sum() for d in dd.values())
    got, ref = ops2.synthesis_styles(style, rows, mw, mb, coef, dl, cw), _per_layer(style, rows, mw, mb, coef, dl, cw)
    for a, b in zip(got[0], ref[0]):
        assert torch.allclose(a, b, atol=1e-12)
    for l in dl:
        assert torch.allclose(got[1][l], ref[1][l], atol=1e-12)
    for a, b in zip(torch.autograd.grad(f(*got), leaves), torch.autograd.grad(f(*ref), leaves)):
        assert torch.allclose(a, b, atol=1e-10)
    # a loss of the GRADIENT with respect to the styles (the path-length shape)
    hs = []
    for fn in (ops2.synthesis_styles, _per_layer):
        (gs,) = torch.autograd.grad(f(*fn(style, rows, mw, mb, coef, dl, cw)), style, create_graph=True)
        hs.append(torch.autograd.grad(gs.square().sum(), leaves))
    for a, b in zip(*hs):
        assert torch.allclose(a, b, rtol=1e-9, atol=1e-9)
