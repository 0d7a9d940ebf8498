"""-m gpu: BASELINE configs[1] sizes (B=16, 64x256 boxes, full channel widths).  The float64 oracle is too slow at these
sizes, so the kernels are pinned through size-independent properties of the operators instead:
  * adjointness   <conv(x; w), dy> = <x, conv^T(dy; w)> = <w, wgrad(x, dy)>   (one scalar, three independent kernels)
  * linearity in the input / in dy
  * split-K (slab) launches == unsplit launches
  * the FIR resampler's gradient is its adjoint (upfirdn_2d_v2.py:204-209)
  * a HIP-graph replay of the whole step == the eager step on the same injected randomness."""
import math

import pytest
import torch

from conftest import arith_modes

pytestmark = pytest.mark.gpu


def _dot(a, b):
    return float((a.double() * b.double()).sum())


def _close(a, b, tol):
    return abs(a - b) <= tol * max(abs(a), abs(b), 1e-30)


FULL_LAYERS = [
    # name, C, M, H, W, k, stride, pad
    ("G 64x256 128->128 3x3", 128, 128, 64, 256, 3, (1, 1), (1, 1)),
    ("G 16x64 256->256 3x3", 256, 256, 16, 64, 3, (1, 1), (1, 1)),
    ("G 4x16 512->512 3x3 (split-K)", 512, 512, 4, 16, 3, (1, 1), (1, 1)),
    ("D 64x256 64->128 3x3 s2", 64, 128, 67, 259, 3, (2, 2), (0, 0)),
    ("D 4x16 512->512 1x1 s2", 512, 512, 4, 16, 1, (2, 2), (0, 0)),
]


@arith_modes
@pytest.mark.parametrize("layer", FULL_LAYERS, ids=[l[0] for l in FULL_LAYERS])
def test_conv_adjoint_identities_full_size(dev, layer):
    from textboxgan_amd import ops
    _, C, M, H, W, k, stride, pad = layer
    B = 16
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn(B, C, H, W, generator=g).to(dev)
    w = (torch.randn(k, k, C, M, generator=g) / math.sqrt(k * k * C)).to(dev)
    Ho, Wo = (H + 2 * pad[0] - k) // stride[0] + 1, (W + 2 * pad[1] - k) // stride[1] + 1
    dy = torch.randn(B, M, Ho, Wo, generator=g).to(dev)
    geom = ops._Geom(stride, pad, k, k, (H, W), (Ho, Wo))
    y = ops._fwd_launch(x, w, geom)
    dx = ops._bwd_data_launch(dy, w, geom)
    dw = ops._bwd_weight_launch(x, dy, geom, C, M)
    assert y.shape == dy.shape and dx.shape == x.shape and dw.shape == w.shape
    s_fwd, s_bwd, s_wg = _dot(y, dy), _dot(x, dx), _dot(w, dw)
    # ~1e8-1e9 fp32 products per scalar: fp32 accumulation inside the MFMA chains, float64 only for the final dot
    assert _close(s_fwd, s_bwd, 2e-4), (s_fwd, s_bwd)
    assert _close(s_fwd, s_wg, 2e-4), (s_fwd, s_wg)
    # linearity in x (same filter): conv(2.5 x + x2) == 2.5 conv(x) + conv(x2)
    x2 = torch.randn(B, C, H, W, generator=g).to(dev)
    lhs = ops._fwd_launch(2.5 * x + x2, w, geom)
    rhs = 2.5 * y + ops._fwd_launch(x2, w, geom)
    assert float((lhs - rhs).abs().max()) <= 2e-4 * float(rhs.abs().max())
    # no atomics anywhere (slabs, partial tiles, fixed-order reductions): a repeated launch repeats bit for bit
    assert torch.equal(ops._fwd_launch(x, w, geom), y) and torch.equal(ops._bwd_data_launch(dy, w, geom), dx)
    assert torch.equal(ops._bwd_weight_launch(x, dy, geom, C, M), dw)


@arith_modes
def test_transposed_up_conv_adjoint_full_size(dev):
    """up-conv (stride-2 transposed 3x3, upfirdn_2d_v2.py:65-103): <convT(x), dy> == <x, strided conv(dy)>."""
    from textboxgan_amd import ops
    B, C, M, H, W = 16, 128, 128, 32, 128
    g = torch.Generator(device="cpu").manual_seed(2)
    x = torch.randn(B, C, H, W, generator=g).to(dev)
    w = (torch.randn(3, 3, C, M, generator=g) / math.sqrt(9 * C)).to(dev)
    y = ops.conv2d_raw(x, ops.pack_filter(w, False, False), M, 3, 3, (2 * H + 1, 2 * W + 1), (2, 2), (0, 0), transposed=True,
                       flip=True)
    dy = torch.randn(*y.shape, generator=g).to(dev)
    dx = ops.conv2d_raw(dy, ops.pack_filter(w, True, True), C, 3, 3, (H, W), (2, 2), (0, 0))
    assert _close(_dot(y, dy), _dot(x, dx), 2e-4)


@arith_modes
def test_split_k_slabs_equal_unsplit(dev):
    from textboxgan_amd import ops
    B, C, M, H, W = 16, 512, 512, 4, 16
    g = torch.Generator(device="cpu").manual_seed(3)
    x = torch.randn(B, C, H, W, generator=g).to(dev)
    w = ops.pack_filter((torch.randn(3, 3, C, M, generator=g) / math.sqrt(9 * C)).to(dev), False, False)
    outs = []
    try:
        for ks in (1, 2, 8, 64):  # 64 = one chunk per split (and exactly the chunk count)
            ops.TUNING.force_ksplit = ks
            outs.append(ops.conv2d_raw(x, w, M, 3, 3, (H, W), (1, 1), (1, 1)))
    finally:
        ops.TUNING.force_ksplit = None
    for o in outs[1:]:
        assert float((o - outs[0]).abs().max()) <= 2e-5 * float(outs[0].abs().max())


def test_upfirdn_gradient_is_adjoint_full_size(dev):
    from textboxgan_amd import ops
    B, C, H, W = 16, 128, 65, 257
    g = torch.Generator(device="cpu").manual_seed(4)
    x = torch.randn(B, C, H, W, generator=g).to(dev).requires_grad_(True)
    k = ops.fir_kernel(dev, 4.0)
    y = ops.upfirdn2d(x, k, pad=(1, 1, 1, 1))
    assert y.shape == (B, C, 64, 256)
    dy = torch.randn(*y.shape, generator=g).to(dev)
    (dx,) = torch.autograd.grad(y, x, dy)
    assert _close(_dot(y, dy), _dot(x.detach(), dx), 1e-4)


def test_graph_replay_equals_eager_step_full_size(dev):
    """same seeded state + the same device RNG streams (the captured graph reads seed/offset of the default generator at
    replay time): the captured step and the eager step run the same kernels on the same numbers -- the library has no
    atomics left (round 3); the OCR branch's torch grid_sample backward (atomic scatter) is the one order-dependent sum.  All seven losses and the flat weight buffers of G and
    D after three optimisation steps must therefore agree to ~1e-5 RELATIVE (the first round's 5e-3 absolute bound was
    the size of the Adam updates themselves and could not have detected a wrong gradient)."""
    from textboxgan_amd.config import Config
    from textboxgan_amd.training_step import build_trainer_state
    from bench import synthetic_batch, bench_init_
    cfg = Config(batch_size_per_gpu=16)
    batch = synthetic_batch(cfg, dev, 7)
    results = []
    for use_graphs in (False, True):
        torch.manual_seed(11)
        st = build_trainer_state(cfg, dev, seed=0, use_graphs=use_graphs)
        bench_init_(st)
        ts = st["training_step"]
        w_start = st["generator"]._flat.flat.clone()
        losses = None
        for _ in range(3):  # graph mode: step 1 eager warm-up, step 2 capture + replay, step 3 replay
            torch.manual_seed(100 + ts.g_optimizer.iterations)  # device RNG drives z, z2, noise, dropout identically
            losses = ts.dist_train_step(batch["real_images"], batch["ocr_images"], batch["input_words"], batch["ocr_labels"],
                                        False, False, 1e-4)
        flat = [float(v) for grp in losses[:2] for v in grp] + [float(losses[2])]
        results.append((flat, st["generator"]._flat.flat.clone(), st["discriminator"]._flat.flat.clone(), w_start))
    (l0, g0, d0, s0), (l1, g1, d1, _) = results
    # Measured (r02): after three steps the losses agree to ~4e-5 relative (the atomics' rounding is amplified by two
    # optimisation steps), so 5e-4 on the losses; the UPDATES (w_end - w_start) are compared by relative L2 -- with
    # beta1 = 0 an Adam step is ~lr*sign(g), so a wrong gradient anywhere shows up as a block of flipped signs
    # (relative L2 error of order 1), while a handful of noise-level elements changing sign does not.
    for a, b in zip(l0, l1):
        assert math.isfinite(a) and _close(a, b, 5e-4), (l0, l1)
    upd_e, upd_g = (g0 - s0).double(), (g1 - s0).double()
    assert float(upd_e.abs().max()) > 1e-3, "three Adam steps must have moved the weights (otherwise this is vacuous)"
    rel = float((upd_e - upd_g).norm() / upd_e.norm())
    # The bar has to clear what two EAGER runs of the same code differ by: the atomic scatter of the OCR branch's grid_sample backward
    # has a handful of discrete outcomes (its gradient set differs by 0 / 2.3e-8 / 2.8e-8 / 3.9e-8 relative between identical runs), and
    # with beta1 = 0 each outcome flips the update sign of a different set of noise-level elements: over 24 alternating eager / graph
    # triples per build, 21 - 23 agree to 1e-4 and the rest sit at one or two discrete levels -- 0.011 (rounds 2 - 5), 0.021 / 0.024
    # since the reduction orders of round 6 (tools/repeat_grads.py: 60 repeated gradient computations are bit-identical in the g and d sets).  A wrong
    # gradient in one layer is a block of flipped signs: O(0.1 - 1).
    assert rel <= 5e-2, ("G update, relative L2 graph vs eager", rel)
    reld = float((d0 - d1).double().norm() / (d0.double() - d0.double().mean()).norm())
    assert reld <= 1e-3, ("D weights, relative L2 graph vs eager", reld)
