"""-m gpu: the networks and the whole step at the REAL channel widths (Config(): 128..512 / 64..512 feature maps,
64x256 boxes) against the torch-CPU fp32 oracle on identical weights and injected randomness -- the widths the
per-kernel tests cover piecewise and test_training_step_gpu.py covers only at small_config.

Tolerances: BASELINE.json north_star asks for the generator RGB output within 1e-3 max-abs; losses 2e-4 relative;
gradient sets by relative L2 (a LeakyReLU mask can flip between two fp32 evaluation orders, so max-abs on a gradient
tensor is not a stable measure)."""
import ctypes as C
import math
import os

import numpy as np
import pytest
import torch

from oracle import ref_model as M, ref_ops as R
from textboxgan_amd.config import Config

from conftest import arith_modes

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def l2_err(a, ref):
    a = a.detach().double().cpu()
    ref = ref.detach().double().cpu()
    assert a.shape == ref.shape, (a.shape, ref.shape)
    return float((a - ref).norm() / (ref.norm() + 1e-30))


def _todev(rand, dev):
    return {k: ([t.to(dev) for t in v] if isinstance(v, list) else (v.to(dev) if torch.is_tensor(v) else v))
            for k, v in rand.items()}


@pytest.fixture(scope="module")
def full4():
    cfg = Config(batch_size_per_gpu=4)
    return cfg, M.make_batch(cfg), M.make_rand(cfg, seed=99)


@arith_modes
@pytest.mark.parametrize("training", [False, True], ids=["inference", "training"])
def test_generator_full_width_rgb_within_1e3(dev, full4, training):
    """north_star: generator RGB output within 1e-3 max-abs of the reference on identical seeds (B = 4)."""
    from textboxgan_amd.models import Generator
    cfg, batch, rand = full4
    P = M.init_generator(cfg, seed=0, bench_init=True)
    G = Generator(cfg)
    G.load_state_dict({k: v.clone() for k, v in P.items()})
    G = G.to(dev)
    with torch.no_grad():
        ref = M.generator({k: v.clone() for k, v in P.items()}, cfg, batch["input_words"], rand["z"], rand, training=training)
        got = G((batch["input_words"].to(dev), rand["z"].to(dev)), training=training, rand=_todev(rand, dev))
    err = float((got.cpu() - ref).abs().max())
    assert ref.shape == (4, 3, 64, 256) and math.isfinite(err)
    assert err <= 1e-3, f"max-abs {err} (|ref|max {float(ref.abs().max())})"


@arith_modes
def test_discriminator_full_width_forward_backward(dev, full4):
    """D forward + all parameter gradients + the image gradient at the real widths (B = 4)."""
    from textboxgan_amd.models import Discriminator
    cfg, batch, _ = full4
    P = M.init_discriminator(cfg, seed=1, bench_init=True)
    D = Discriminator(cfg)
    D.load_state_dict({k: v.clone() for k, v in P.items()})
    D = D.to(dev)
    img = batch["real_images"].clone().requires_grad_(True)
    Pr = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    sc = M.discriminator(Pr, cfg, img)
    loss = torch.nn.functional.softplus(-sc).sum()
    names = list(Pr.keys())
    gref = torch.autograd.grad(loss, [img] + [Pr[n] for n in names])
    imgd = batch["real_images"].to(dev).requires_grad_(True)
    scd = D(imgd)
    lossd = torch.nn.functional.softplus(-scd).sum()
    params = dict(D.named_parameters())
    gd = torch.autograd.grad(lossd, [imgd] + [params[n] for n in names])
    assert float((scd.detach().cpu() - sc.detach()).abs().max()) <= 2e-4 * max(1.0, float(sc.abs().max()))
    assert l2_err(gd[0], gref[0]) < 2e-3, "d/d image"
    for n, a, e in zip(names, gd[1:], gref[1:]):
        assert l2_err(a, e) < 2e-3, n


# ----------------------------------------------------------------------------------------------------------------
# one whole dist_train_step at the real Config against the CPU oracle, per arithmetic
# ----------------------------------------------------------------------------------------------------------------
# fp32 bars (unchanged since round 2; "f32x3" -- three bf16 terms per operand on the bf16 pipe -- must meet the SAME bars).
_TOL_F32 = dict(loss=5e-4, g=5e-3, ocr=2e-2, d=5e-3, flat_g=2e-3, flat_o=5e-3, flat_d=2e-3, pl_mean=2e-4,
                adam=1e-3)
# bf16 mode has no counterpart in the (fp32-only) reference; the bars are the mode-accuracy bars of tests/test_bf16_gpu.py's
# docstring: unit round-off 2^-9 per operand through 12 + 7 layers and their backward -> losses 3e-2, gradient SETS 8e-2
# relative L2; single tensors (few elements, sums with cancellation) 2.5e-1; the post-Adam comparison is dropped (with
# beta1 = 0 a step is ~lr*sign(g): a sign flip of a near-zero gradient is not an error of the mode).
_TOL_BF16 = dict(loss=3e-2, g=2.5e-1, ocr=None, d=2.5e-1, flat_g=8e-2, flat_o=8e-2, flat_d=8e-2, pl_mean=3e-2,
                 adam=None)
STEP_TOL = {"f32": _TOL_F32, "f32x3": _TOL_F32, "bf16": _TOL_BF16}
_ORACLE_STEPS, _PRODUCT_STEPS = {}, {}


def _oracle_step(B, reg):
    """the CPU oracle's step (losses, the three gradient sets, post-step state) for batch B: computed once per (B, reg) and
    shared by every arithmetic that is compared with it."""
    key = (B, reg)
    if key not in _ORACLE_STEPS:
        from conftest import ocr_oracle
        cfg = Config(batch_size_per_gpu=B)
        batch, rand = M.make_batch(cfg), M.make_rand(cfg, seed=99)
        st = M.make_state(cfg, seed=0, bench_init=True)
        init = {k: {n: v.clone() for n, v in st[k].items()} for k in ("G", "D")}
        ocr_cpu = ocr_oracle(cfg.max_char_number)
        prev = torch.get_num_threads()
        if B >= 16:  # the per-sample pieces over-subscribe oneDNN on a 256-thread host (44 s / step against ~13 s)
            torch.set_num_threads(min(prev, 16))
        try:
            losses, grads = M.training_step(st, cfg, batch["real_images"], batch["ocr_images"], batch["input_words"],
                                            batch["ocr_labels"], reg[0], reg[1], 1e-4, rand, ocr_cpu.serve, return_grads=True)
        finally:
            torch.set_num_threads(prev)
        _ORACLE_STEPS[key] = (cfg, batch, rand, init, st, losses, grads)
    return _ORACLE_STEPS[key]


def _step_vs_oracle(dev, arith, B, reg, tol_override=None):
    """run ONE product dist_train_step (arithmetic `arith`, per-GPU batch B, lazy-reg flags reg) on the oracle's weights and
    injected randomness, compare 7 losses, the three gradient sets (per tensor and as flat buffers), pl_mean and the
    post-Adam discriminator, and return the set of kernel instantiations the step launched (native.record_calls)."""
    key = (str(dev), arith, B, reg)
    if key in _PRODUCT_STEPS:
        return _PRODUCT_STEPS[key]
    from textboxgan_amd import native as N
    from textboxgan_amd.training_step import build_trainer_state
    tol = dict(STEP_TOL[arith], **(tol_override or {}))
    do_r1, do_pl = reg
    cfg, batch, rand, init, st, ref_losses, ref_grads = _oracle_step(B, reg)
    prod = build_trainer_state(cfg, dev, seed=0, compute_dtype=arith)
    prod["generator"].load_state_dict({k: v.clone() for k, v in init["G"].items()})
    prod["discriminator"].load_state_dict({k: v.clone() for k, v in init["D"].items()})
    prod["g_clone"].load_state_dict({k: v.clone() for k, v in init["G"].items()})  # model_loader.py:13-20: g_clone starts as G
    ts = prod["training_step"]
    b = {k: v.to(dev) for k, v in batch.items()}
    with N.record_calls() as log:
        losses = ts.dist_train_step(b["real_images"], b["ocr_images"], b["input_words"], b["ocr_labels"], do_r1, do_pl, 1e-4,
                                    rand=_todev(rand, dev))
        torch.cuda.synchronize()
    flat = lambda t: [float(x) for x in t] if isinstance(t, tuple) else [float(t)]
    got = flat(losses[0]) + flat(losses[1]) + flat(losses[2])
    exp = flat(ref_losses[0]) + flat(ref_losses[1]) + flat(ref_losses[2])
    for name, a, e in zip(("reg_g", "g", "pl", "reg_d", "d", "r1", "ocr"), got, exp):
        assert abs(a - e) <= tol["loss"] * max(1.0, abs(e)), (arith, name, a, e)
    gnames = [n for n in prod["generator"]._flat.names if n.startswith(("latent_encoder.", "synthesis."))]
    worst = max((l2_err(v, ref_grads["g"][n]), n) for n, v in zip(gnames, ts.g_views))
    assert worst[0] < tol["g"], (arith, "g", worst)
    onames = [n for n in prod["generator"]._flat.names if n.startswith(("synthesis.", "word_encoder."))]
    # the OCR-weighted (1e-4) gradients are sums with heavy cancellation; a SCALAR parameter (noise_strength) is one such
    # sum, so its relative error is the conditioning of that sum (measured 2.4e-2 on synth_blocks.4.apply_noise_1)
    cat = lambda names, d: torch.cat([d[n].reshape(-1) for n in names])
    catv = lambda views: torch.cat([v.reshape(-1) for v in views])  # (the flat buffers carry alignment padding)
    if tol["ocr"] is None:
        # bf16: in the COUPLED step the OCR-weighted set cannot be compared element-wise -- the recogniser's image gradient is a
        # chaotic function of the image for this random-weight stand-in (exact-fp32 kernel noise of 1e-7 already becomes 1.2e-3
        # on this set) and the bf16 generator's images differ from the oracle's by ~1e-2.  The set is therefore checked
        # DECOUPLED, element-wise, in test_bf16_ocr_gradient_set_decoupled (recogniser on the oracle's images; generator chain
        # given the oracle's d(loss)/d(image)); here: the loss (above) and finiteness.
        assert torch.isfinite(catv(ts.o_views)).all()
        onames = []
    worst = max([(l2_err(v, ref_grads["ocr"][n]), n) for n, v in zip(onames, ts.o_views) if v.numel() > 1] or [(0.0, "-")])
    assert worst[0] < (tol["ocr"] or 1.0), (arith, "ocr", worst)
    # the set's SCALAR parameters (the ten noise strengths: one heavily cancelling sum each) are compared as ONE vector at the
    # per-tensor bar -- a single criterion for both fp32 arithmetics (a lone scalar has no norm to be relative to; measured
    # absolute errors are 1e-4 .. 1.2e-3 of the set norm on every tensor of this set, profiles/r03_step_error_tables.txt).
    sc = [(v.detach().double().cpu().reshape(()), ref_grads["ocr"][n].double().reshape(())) for n, v in zip(onames, ts.o_views)
          if v.numel() == 1]
    if sc:
        a, e = torch.stack([x for x, _ in sc]), torch.stack([y for _, y in sc])
        err = float((a - e).norm() / (e.norm() + 1e-30))
        assert err < tol["ocr"], (arith, "ocr scalars as one vector", err)
    worst = max((l2_err(v, ref_grads["d"][n]), n) for n, v in zip(prod["discriminator"]._flat.names, ts.d_views))
    assert worst[0] < tol["d"], (arith, "d", worst)
    # whole flat gradient buffers (what Adam / the all-reduce consume)
    errs = dict(g=l2_err(catv(ts.g_views), cat(gnames, ref_grads["g"])),
                o=l2_err(catv(ts.o_views), cat(onames, ref_grads["ocr"])) if onames else 0.0,
                d=l2_err(catv(ts.d_views), cat(prod["discriminator"]._flat.names, ref_grads["d"])))
    assert errs["g"] < tol["flat_g"] and errs["o"] < tol["flat_o"] and errs["d"] < tol["flat_d"], (arith, errs)
    if do_pl:
        assert abs(float(prod["pl_mean"]) - float(st["pl_mean"])) <= tol["pl_mean"] * max(1.0, abs(float(st["pl_mean"])))
    if tol["adam"] is not None:
        for n, v in prod["discriminator"].state_dict().items():
            assert l2_err(v, st["D"][n]) < tol["adam"], (arith, "D after Adam", n)
    # the caller's g_clone EMA (train.py:208, generator.py:48-59) against the oracle's
    with N.record_calls() as log2:
        prod["g_clone"].set_as_moving_average_of(prod["generator"])
        torch.cuda.synchronize()
    expect = {k: v.clone() for k, v in init["G"].items()}  # g_clone before the update (= G before the step)
    M.ema_update(expect, {k: v.detach().cpu() for k, v in prod["generator"].state_dict().items()})  # the oracle's rule
    for n, v in prod["g_clone"].state_dict().items():
        d = float((v.detach().cpu().double() - expect[n].double()).abs().max())
        assert d <= 1e-6 * max(1.0, float(expect[n].abs().max())), (arith, "g_clone after EMA", n, d)
    log = set(log) | set(log2)
    _PRODUCT_STEPS[key] = (frozenset(log), errs)
    return _PRODUCT_STEPS[key]


@arith_modes
@pytest.mark.parametrize("reg", [(False, False), (True, True)], ids=["plain", "r1+pl"])
def test_training_step_full_width_matches_oracle(dev, arith, reg):
    """one dist_train_step at the real Config (B = 4): 7 losses, the three gradient sets, post-Adam weights, pl_mean --
    in exact fp32 and in f32x3 arithmetic at the same (fp32) tolerances."""
    _step_vs_oracle(dev, arith, 4, reg)


@pytest.mark.parametrize("reg", [(False, False), (True, True)], ids=["plain", "r1+pl"])
def test_training_step_full_width_matches_oracle_bf16(dev, reg):
    """BASELINE configs[2] arithmetic (bf16 MFMA operands, fp32 accumulate) at the real widths against the fp32 CPU oracle:
    losses, every gradient tensor, the three flat gradient sets, pl_mean -- at the bf16 mode-accuracy bars (_TOL_BF16).
    Round 2 compared this mode with the product's own fp32 path only."""
    _step_vs_oracle(dev, "bf16", 4, reg)


def test_bf16_ocr_gradient_set_decoupled(dev):
    """The gradient set ocr_optimizer applies to synthesis + word_encoder (reference training_step.py:201-206, 375-402) in bf16
    mode, checked ELEMENT-WISE in two decoupled halves (the coupled comparison is meaningless: see _step_vs_oracle):
      (a) the frozen recogniser + softmax-CE (fp32 grade in every mode) on the ORACLE's fake images: d(w * ocr_loss)/d(fake)
          against the oracle's, at the fp32 per-tensor bar of that set;
      (b) the bf16 generator chain GIVEN the oracle's d(w * ocr_loss)/d(fake): every tensor of the set and the flat buffer
          against the oracle's OCR set at the bf16 bars (per tensor _TOL_BF16["g"], flat _TOL_BF16["flat_o"])."""
    from textboxgan_amd import ops
    from textboxgan_amd.training_step import build_trainer_state
    cfg, batch, rand, init, st, ref_losses, ref_grads = _oracle_step(4, (False, False))
    prod = build_trainer_state(cfg, dev, seed=0, compute_dtype="bf16")
    prod["generator"].load_state_dict({k: v.clone() for k, v in init["G"].items()})
    ts, G = prod["training_step"], prod["generator"]
    b = {k: v.to(dev) for k, v in batch.items()}
    w = 1e-4
    # (a) the step's own OCR branch (TrainingStep._get_ocr_loss under the arithmetic the bf16 step gives it)
    fake_ref = ref_grads["fake"].to(dev).requires_grad_(True)
    with ops.STATE_LOCK, ops.filter_cache(), ops.compute_dtype("f32x3"):
        loss = ts._get_ocr_loss(fake_ref, b["ocr_labels"], b["ocr_images"])
        (dfake,) = torch.autograd.grad(w * loss, fake_ref)
    assert abs(float(loss) - float(ref_losses[2])) <= _TOL_F32["loss"] * max(1.0, abs(float(ref_losses[2])))
    e_a = l2_err(dfake, ref_grads["dfake_ocr"])
    assert e_a < _TOL_F32["ocr"], ("d(ocr loss)/d(fake) on the oracle's images", e_a)
    # (b) the generator's ocr-pass in bf16 mode from the oracle's image gradient
    r = _todev(rand, dev)
    with ops.STATE_LOCK, ops.filter_cache(), ops.compute_dtype("bf16"):
        fake = G((b["input_words"], r["z"]), training=True, rand=r, mask_words=b["input_words"])
        grads = torch.autograd.grad(fake, ts.o_params, grad_outputs=ref_grads["dfake_ocr"].to(dev), allow_unused=True)
    torch.cuda.synchronize()
    onames = [n for n in G._flat.names if n.startswith(("synthesis.", "word_encoder."))]
    assert len(onames) == len(grads)
    per = sorted(((l2_err(g, ref_grads["ocr"][n]), n) for n, g in zip(onames, grads) if g.numel() > 1), reverse=True)
    assert per[0][0] < _TOL_BF16["g"], ("o-set per tensor", per[:3])
    a = torch.cat([g.reshape(-1) for g in grads]).double().cpu()
    e = torch.cat([ref_grads["ocr"][n].reshape(-1) for n in onames]).double()
    flat = float((a - e).norm() / e.norm())
    print(f"bf16 o-set decoupled: dfake {e_a:.3e}; flat {flat:.3e}; worst tensors {per[:3]}")
    assert flat < _TOL_BF16["flat_o"], ("o-set flat", flat)


@pytest.mark.parametrize("arith", ["f32x3", "f32"])
def test_training_step_full_width_batch16_matches_oracle(dev, arith):
    """the BENCHMARKED geometry: per-GPU batch 16 (joint [fake; real] discriminator pass over 32 samples, the tile
    rounds and split-K choices of the B = 16 launches) -- one plain step against the CPU oracle at the fp32 tolerances."""
    _step_vs_oracle(dev, arith, 16, (False, False))


@arith_modes
def test_hello_full_size_fixture(dev):
    """BASELINE configs[0] ("Hello", B = 1, generator forward) at the real widths against the committed float64-oracle
    fixture (tests/golden/hello_fullsize.npz, made by tests/golden/make_golden_fullsize.py)."""
    from textboxgan_amd.models import Generator, generator_output_to_uint8, mask_text_box
    n = np.load(os.path.join(GOLD, "hello_fullsize.npz"))
    cfg = Config(batch_size_per_gpu=1)
    words = torch.from_numpy(n["words"])
    assert words[0].tolist() == [44, 15, 22, 22, 25, 0, 0, 0]
    G = Generator(cfg)
    G.load_state_dict(M.init_generator(cfg, seed=11, bench_init=True))
    G = G.to(dev)
    rand = dict(noises=[torch.from_numpy(n[f"noise{i}"]).to(dev) for i in range(10)])
    with torch.no_grad():
        img = G((words.to(dev), torch.from_numpy(n["z"]).to(dev)), training=False, truncation_psi=1.0, rand=rand)
    rows = img[0][:, (5, 31, 60), :].cpu().numpy()
    assert np.abs(rows - n["image_rows"]).max() <= 1e-3
    cs = np.array([float(img.double().sum()), float(img.double().abs().sum()), float(img.double().square().sum())])
    np.testing.assert_allclose(cs, n["image_checksum"], rtol=2e-4, atol=2e-2)
    u8 = generator_output_to_uint8(mask_text_box(img, words.to(dev), cfg.char_width))[0, :, : 32 * 5]
    assert tuple(u8.shape) == tuple(n["u8_shape"])
    # +-1 grey level per pixel at most (rounding of values that sit on a .5 boundary)
    assert abs(int(u8.to(torch.int64).sum()) - int(n["u8_sum"])) <= u8.numel() // 200


# ----------------------------------------------------------------------------------------------------------------
# every kernel instantiation the BASELINE step launches is compared with the float64 oracle somewhere
# ----------------------------------------------------------------------------------------------------------------
def _conv_cases_for_coverage():
    """(B, C, M, H, W, k, stride, pad, transposed) of the oracle-compared convolutions below: chosen so that their
    descriptors select every conv_fprop / conv_wgrad instantiation the full-size step uses."""
    return [
        (2, 64, 64, 64, 256, 3, (1, 1), (1, 1), False),    # 64x256 tile, software-pipelined <1,4,2,2,4,9,3,3>
        (13, 64, 64, 64, 256, 3, (1, 1), (1, 1), False),   # the same beyond 768 tiles (832): 4 blocks/CU <1,4,2,2,4,9,3,4>
        (6, 24, 256, 64, 256, 3, (1, 1), (1, 1), False),   # >= 1536 tiles: 4 waves/SIMD <2,2,2,2,4,9,0,4>
        (8, 16, 128, 64, 256, 3, (1, 1), (1, 1), False),   # >= 1024 tiles of 128x128: the f32x3 128x256 tile <2,2,2,4,8,9,0,2>
        (2, 128, 128, 16, 64, 3, (1, 1), (1, 1), False),   # <2,2,2,2,8,9,0,3>
        (4, 512, 512, 4, 16, 3, (1, 1), (1, 1), False),    # 64x64 tile <2,2,1,1,8,9,0,3>, split-K
        (2, 16, 32, 16, 64, 3, (1, 1), (1, 1), False),     # BM = 32 <1,4,1,2,8,9,0,3>
        (2, 40, 64, 16, 64, 3, (1, 1), (1, 1), False),     # M <= 64, few tiles: 64x64
        (2, 64, 128, 67, 259, 3, (2, 2), (0, 0), False),   # strided (ragged 33x129 output: scalar-staged filter gradient)
        (2, 64, 128, 65, 257, 3, (2, 2), (0, 0), False),   # strided, 32x128 output: float4-staged filter gradient <2,2,9,32,true,2>
        (2, 128, 128, 16, 64, 3, (2, 2), (0, 0), True),    # stride-2 transposed, parity classes <2,2,2,2,16,4,0,3>
        (2, 128, 128, 16, 64, 3, (1, 2), (0, 0), True),    # stride (1,2) transposed: parity classes <2,2,2,2,16,4,0,3>
        (6, 32, 64, 32, 128, 3, (1, 2), (0, 0), True),     # the same on the 64 x 256 tile <1,4,2,2,16,4,0,3>
        (6, 32, 64, 32, 128, 3, (2, 2), (0, 0), True),     # transposed, M = 64, >= 96 256-pixel tiles: <1,4,2,2,16,4,0,3>
        (2, 32, 24, 8, 32, 3, (2, 2), (0, 0), True),       # transposed BM = 32
        (4, 512, 256, 4, 16, 3, (2, 2), (0, 0), True),     # transposed 64x64
        (2, 3, 64, 16, 64, 1, (1, 1), (0, 0), False),      # 1x1
        (2, 128, 256, 16, 64, 1, (2, 2), (0, 0), False),   # 1x1 strided (skip branch geometry)
        (2, 32, 32, 16, 50, 1, (2, 2), (0, 0), True),      # 1x1 transposed stride (the recogniser's first stage, data gradient): BM = 32 few-tap
    ]


def _run_conv_case(dev, case, seed, arith="f32"):
    """forward (+ data gradient + filter gradient for the non-transposed forms) vs float64 torch, in arithmetic `arith`
    (bf16: the float64 reference is evaluated on operands pre-rounded to bf16 exactly as the kernels round them, cf.
    tests/test_bf16_gpu.py); returns the kernel instantiations these launches select."""
    import torch.nn.functional as F
    from textboxgan_amd import native as N, ops
    B, Cc, Mo, H, W, k, stride, pad, transposed = case
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cc, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(k, k, Cc, Mo, generator=g, dtype=torch.float64) / math.sqrt(k * k * Cc)
    q = (lambda t: t.float().bfloat16().double()) if arith == "bf16" else (lambda t: t.float().double())
    rel = lambda a, r: float((a.double().cpu() - r).abs().max() / (r.abs().max() + 1e-30))
    with N.record_calls() as log, ops.compute_dtype(arith):
        if transposed:
            ref = F.conv_transpose2d(q(x), q(w).permute(2, 3, 0, 1), stride=stride)
            y = ops.conv2d_raw(x.float().to(dev), w.float().to(dev), Mo, k, k, (ref.shape[2], ref.shape[3]), stride, (0, 0),
                               transposed=True)
            assert rel(y, ref) < 3e-5, ("transposed", arith, case)
        else:
            xr, wr = q(x).requires_grad_(True), q(w).requires_grad_(True)
            ref = F.conv2d(xr, wr.permute(3, 2, 0, 1), stride=stride, padding=pad)
            dy = torch.randn(*ref.shape, generator=g, dtype=torch.float64)
            gx, gw = torch.autograd.grad(ref, (xr, wr), q(dy))
            if arith == "bf16" and dy.shape[3] <= 4:  # narrow rows keep the exact fp32 filter-gradient kernel
                xe, we = x.float().double().requires_grad_(True), w.float().double().requires_grad_(True)
                (gw,) = torch.autograd.grad(F.conv2d(xe, we.permute(3, 2, 0, 1), stride=stride, padding=pad), we,
                                            dy.float().double())
            geom = ops._Geom(stride, pad, k, k, (H, W), (ref.shape[2], ref.shape[3]))
            xd, wd, dyd = x.float().to(dev), w.float().to(dev), dy.float().to(dev)
            assert rel(ops._fwd_launch(xd, wd, geom), ref.detach()) < 3e-5, ("fwd", arith, case)
            assert rel(ops._bwd_data_launch(dyd, wd, geom), gx) < 3e-5, ("dgrad", arith, case)
            assert rel(ops._bwd_weight_launch(xd, dyd, geom, Cc, Mo), gw) < 5e-5, ("wgrad", arith, case)
    return set(log)


def _run_units_case(dev, arith, M, seed):
    """the unit-tensor kernels (csrc/conv_units.hip: producer, all-DMA forward, transposing-read filter gradient) against
    float64 on the operands they see, in arithmetic `arith` (f32x3 / bf16); returns the instantiations launched."""
    import torch.nn.functional as F
    from textboxgan_amd import native as N, ops
    B, Cc, H, W = 2, 64, 16, 64
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cc, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(3, 3, Cc, M, generator=g, dtype=torch.float64) / math.sqrt(9 * Cc)
    dy = torch.randn(B, M, H, W, generator=g, dtype=torch.float64)
    q = (lambda t: t.float().bfloat16().double()) if arith == "bf16" else (lambda t: t.float().double())
    rel = lambda a, r: float((a.double().cpu() - r).abs().max() / (r.abs().max() + 1e-30))
    xr, wr = q(x).requires_grad_(True), q(w).requires_grad_(True)
    ref = F.conv2d(xr, wr.permute(3, 2, 0, 1), padding=1)
    gx, gw = torch.autograd.grad(ref, (xr, wr), q(dy))
    small_was, ops.TUNING.use_small = ops.TUNING.use_small, False  # (maps this small would take tbg_conv2d_units_small otherwise)
    with N.record_calls() as log, ops.compute_dtype(arith):
        xd, wd, dyd = x.float().to(dev), w.float().to(dev), dy.float().to(dev)
        XU, DU = ops.units_pack(xd), ops.units_pack(dyd)
        assert rel(ops.conv2d_units_raw(XU, ops.pack_filter(wd, False, False), M), ref.detach()) < 3e-5, ("units fwd", arith, M)
        assert rel(ops.conv2d_units_raw(DU, ops.pack_filter(wd, True, True), Cc), gx) < 3e-5, ("units dgrad", arith, M)
        dw = torch.empty(3, 3, Cc, M, device=dev)
        assert rel(ops.wgrad_units_raw(DU, XU, dw, Cc * M, M, 1, 1.0), gw) < 5e-5, ("units wgrad", arith, M)
        # the forward kernel's epilogue instantiations (conv_common.h OPT: residual operand, dot / gate operand, both, neither)
        res, gate, aux = (torch.randn(B, M, H, W, generator=g, dtype=torch.float64) for _ in range(3))
        resd, gated, auxd = res.float().to(dev), gate.float().to(dev), aux.float().to(dev)
        pf, y0 = ops.pack_filter(wd, False, False), ref.detach()
        y = ops.conv2d_units_raw(XU, pf, M, epi=N.epilogue(residual=resd, res_scale=0.5))
        assert rel(y, (y0 + res) * 0.5) < 3e-5, ("units fwd + residual", arith, M)
        y = ops.conv2d_units_raw(XU, pf, M, epi=N.epilogue(gate=gated))
        assert rel(y, torch.where(gate > 0, y0, torch.zeros_like(y0))) < 3e-5, ("units fwd + gate", arith, M)
        y = ops.conv2d_units_raw(XU, pf, M, epi=N.epilogue(residual=resd, res_scale=0.5, gate=gated))
        assert rel(y, torch.where(gate > 0, (y0 + res) * 0.5, torch.zeros_like(y0))) < 3e-5, ("units fwd + residual + gate", arith, M)
        dot = torch.empty(B, M, device=dev)
        y = ops.conv2d_units_raw(XU, pf, M, dot=(auxd, dot))
        assert rel(y, y0) < 3e-5 and rel(dot, (y0 * aux).sum((2, 3))) < 3e-5, ("units fwd + dot", arith, M)
    ops.TUNING.use_small = small_was
    assert any(k.startswith("conv_units_fprop_kernel") for k in log), sorted(log)
    return set(log)


def _run_units_s2_case(dev, arith, M, seed):
    """the phase-unit kernels (csrc/conv_units_s2.hip: fused FIR producer, all-DMA stride-2 convolution, 8-wave filter gradient)
    against float64 on the operands they see, in arithmetic `arith` (f32x3 / bf16); returns the instantiations launched."""
    import torch.nn.functional as F
    from textboxgan_amd import native as N, ops
    B, Cc, H, W = 2, 64, 16, 64
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cc, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(3, 3, Cc, M, generator=g, dtype=torch.float64) / math.sqrt(9 * Cc)
    dy = torch.randn(B, M, H // 2, W // 2, generator=g, dtype=torch.float64)
    q = (lambda t: t.float().bfloat16().double()) if arith == "bf16" else (lambda t: t.float().double())
    rel = lambda a, r: float((a.double().cpu() - r).abs().max() / (r.abs().max() + 1e-30))
    with N.record_calls() as log, ops.compute_dtype(arith):
        xd, wd, dyd = x.float().to(dev), w.float().to(dev), dy.float().to(dev)
        k = ops.fir_kernel(dev, 1.0)
        t = ops.upfirdn2d_raw(xd, k, pad=(2, 3, 2, 3))            # the blurred tensor (the NCHW FIR: oracle-compared elsewhere)
        TP = ops.upfirdn2d_units_s2(xd, k, pad=(2, 3, 2, 3))      # ... and its phase unit tensor from the fused producer
        dpk = (TP.data.float() - ops.units_pack_s2(t).data.float()).abs().max()  # same arithmetic, single fp32 roundings apart
        assert float(dpk) <= 2.0 ** -7 * float(t.abs().max()), ("fir units", arith, float(dpk))
        tr, wr = q(t.double().cpu()).requires_grad_(True), q(w).requires_grad_(True)
        ref = F.conv2d(tr, wr.permute(3, 2, 0, 1), stride=2)
        gw, gt = torch.autograd.grad(ref, (wr, tr), q(dy))  # filter gradient; data gradient (a transposed convolution of dy)
        assert rel(ops.conv2d_units_s2_raw(TP, ops.pack_filter(wd, False, False), M), ref.detach()) < 3e-5, ("units s2 fwd", arith, M)
        if M % 128 == 0:
            dw = torch.empty(3, 3, Cc, M, device=dev)
            assert rel(ops.wgrad_units_s2_raw(ops.units_pack(dyd), TP, dw, Cc * M, M, 1, 1.0), gw) < 5e-5, ("units s2 wgrad", arith, M)
        got = ops.conv2d_units_t2_raw(ops.units_pack(dyd), ops.pack_filter(wd, True, False), Cc, (H + 2, W + 2))
        assert rel(got, gt) < 3e-5, ("units t2", arith, M)
    return set(log)


_COVERED = {}


def _covered(dev, arith):
    """kernel instantiations launched by ORACLE-COMPARED work in arithmetic `arith`: the convolution cases above (float64
    reference per launch) plus two whole full-width steps (plain and R1 + path length, B = 4) that are compared with the
    CPU oracle on losses and every gradient -- they bring every non-convolution kernel family (FIR, RGB ends, bias/act,
    dense, minibatch-std, split-K epilogue, modulated-conv tails, filter packing, Adam, the frozen OCR's LSTM / attention
    kernels) at the real channel widths."""
    if arith not in _COVERED:
        cov = set()
        from textboxgan_amd import ops as _ops
        for small in ((True, False) if arith != "f32" else (True,)):
            # round 6: small geometries take tbg_conv2d_units_small; the NCHW instantiations they used to select still serve the
            # launches beyond its block limit (and every step in exact fp32), so each case is compared on both routes
            _ops.TUNING.use_small = small
            try:
                for i, case in enumerate(_conv_cases_for_coverage()):
                    cov |= _run_conv_case(dev, case, 500 + i, arith)
                    if arith == "bf16":  # the frozen OCR branch of a bf16 step runs on the f32x3 kernels (training_step.py)
                        cov |= _run_conv_case(dev, case, 500 + i, "f32x3")
                if arith != "f32":
                    import test_units_gpu as TU
                    from textboxgan_amd import native as N
                    with N.record_calls() as log:
                        for case in TU.SINK_CONVS:
                            TU.test_conv_unit_sink_equals_units_pack(dev, arith, case)
                    cov |= set(log)
            finally:
                _ops.TUNING.use_small = True
        if arith != "f32":  # the 3x3 stride-1 layers of these arithmetics consume unit tensors
            cov |= _run_units_case(dev, arith, 128, 700) | _run_units_case(dev, arith, 64, 701)
            cov |= _run_units_s2_case(dev, arith, 128, 702) | _run_units_s2_case(dev, arith, 64, 703)
            # the unit-sink forms (round 5): every producer against the definition of the unit tensor of its own fp32 result
            import test_units_gpu as TU
            from textboxgan_amd import native as N
            with N.record_calls() as log:
                for case in TU.SINK_CONVS:
                    TU.test_conv_unit_sink_equals_units_pack(dev, arith, case)
                TU.test_conv_units_and_fir_unit_sinks_equal_units_pack(dev, arith)
            cov |= set(log)
        for reg in ((False, False), (True, True)):
            cov |= set(_step_vs_oracle(dev, arith, 4, reg)[0])
        _COVERED[arith] = cov
    return _COVERED[arith]


STEP_VARIANTS = [(a, b, r) for a, b in (("f32x3", 16), ("f32", 16), ("bf16", 32))
                 for r in ((False, False), (False, True), (True, True))]


@pytest.mark.parametrize("arith,B,reg", STEP_VARIANTS,
                         ids=[f"{a}-B{b}-{'plain' if r == (False, False) else 'pl' if r == (False, True) else 'pl+r1'}"
                              for a, b, r in STEP_VARIANTS])
def test_every_step_instantiation_is_oracle_compared(dev, arith, B, reg):
    """run each BENCHMARKED step variant -- {plain, +PL, +PL+R1} x {configs[1]: B = 16 in f32x3 (the headline arithmetic)
    and exact fp32, configs[2]: B = 32 in bf16}, full widths, eager -- with the call recorder on, and require EVERY kernel
    instantiation it launches (convolutions, filter gradients and FIR by their template instantiation, every other
    family by its entry point) to be launched by oracle-compared work as well (_covered)."""
    from bench import bench_init_, synthetic_batch
    from textboxgan_amd import native as N
    from textboxgan_amd.training_step import build_trainer_state
    covered = _covered(dev, arith)
    cfg = Config(batch_size_per_gpu=B)
    st = build_trainer_state(cfg, dev, seed=0, use_graphs=False, compute_dtype=arith)
    bench_init_(st)
    batch = synthetic_batch(cfg, dev, 1234)
    ts = st["training_step"]
    with N.record_calls() as used:
        ts.dist_train_step(batch["real_images"], batch["ocr_images"], batch["input_words"], batch["ocr_labels"], reg[0], reg[1],
                           1e-4)
        st["g_clone"].set_as_moving_average_of(st["generator"])
        torch.cuda.synchronize()
    assert any(k.startswith("conv_fprop") for k in used) and any(k.startswith("upfirdn2d") for k in used), sorted(used)
    missing = sorted(set(used) - covered)
    assert not missing, f"launched by the {arith} B={B} step but never compared with the oracle: {missing}"
    if reg == (False, False):
        # round 5: unit tensors are written by the launch that PRODUCES the activation (tbg_epilogue.units_out), never by a
        # stand-alone pass over a finished fp32 tensor: no plain step of the benchmarked configurations launches units_pack
        # (round 6: except for the inputs of small-map launches that nothing of ours produced -- word-encoder output, pooled maps,
        # decimated skips: at most 2M elements, recorded as "[small]")
        packs = sorted(k for k in used if k.startswith("units_pack_kernel") and not k.endswith("[small]"))
        assert not packs, f"the {arith} B={B} plain step still launches the stand-alone unit producer: {packs}"
