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


@pytest.mark.parametrize("reg", [(False, False), (True, True)], ids=["plain", "r1+pl"])
def test_training_step_full_width_matches_oracle(dev, full4, reg):
    """one dist_train_step at the real Config (B = 4): 7 losses, the three gradient sets, post-Adam weights, pl_mean."""
    from textboxgan_amd.aster import AsterLikeOCR
    from textboxgan_amd.training_step import build_trainer_state
    do_r1, do_pl = reg
    cfg, batch, rand = full4
    st = M.make_state(cfg, seed=0, bench_init=True)
    prod = build_trainer_state(cfg, dev, seed=0)
    prod["generator"].load_state_dict({k: v.clone() for k, v in st["G"].items()})
    prod["discriminator"].load_state_dict({k: v.clone() for k, v in st["D"].items()})
    ts = prod["training_step"]
    ocr_cpu = AsterLikeOCR(max_steps=cfg.max_char_number)
    w = 1e-4
    ref_losses, ref_grads = M.training_step(st, cfg, batch["real_images"], batch["ocr_images"], batch["input_words"],
                                            batch["ocr_labels"], do_r1, do_pl, w, rand, ocr_cpu.serve, return_grads=True)
    b = {k: v.to(dev) for k, v in batch.items()}
    losses = ts.dist_train_step(b["real_images"], b["ocr_images"], b["input_words"], b["ocr_labels"], do_r1, do_pl, w,
                                rand=_todev(rand, dev))
    torch.cuda.synchronize()
    flat = lambda t: [float(x) for x in t] if isinstance(t, tuple) else [float(t)]
    got = flat(losses[0]) + flat(losses[1]) + flat(losses[2])
    exp = flat(ref_losses[0]) + flat(ref_losses[1]) + flat(ref_losses[2])
    for name, a, e in zip(("reg_g", "g", "pl", "reg_d", "d", "r1", "ocr"), got, exp):
        assert abs(a - e) <= 5e-4 * max(1.0, abs(e)), (name, a, e)
    gnames = [n for n in prod["generator"]._flat.names if n.startswith(("latent_encoder.", "synthesis."))]
    worst = max((l2_err(v, ref_grads["g"][n]), n) for n, v in zip(gnames, ts.g_views))
    assert worst[0] < 5e-3, ("g", worst)
    onames = [n for n in prod["generator"]._flat.names if n.startswith(("synthesis.", "word_encoder."))]
    # the OCR-weighted (1e-4) gradients are sums with heavy cancellation; a SCALAR parameter (noise_strength) is one such
    # sum, so its relative error is the conditioning of that sum (measured 2.4e-2 on synth_blocks.4.apply_noise_1)
    worst = max((l2_err(v, ref_grads["ocr"][n]), n) for n, v in zip(onames, ts.o_views) if v.numel() > 1)
    assert worst[0] < 2e-2, ("ocr", worst)
    worst = max((l2_err(v, ref_grads["ocr"][n]), n) for n, v in zip(onames, ts.o_views) if v.numel() == 1)
    assert worst[0] < 6e-2, ("ocr scalar", worst)
    worst = max((l2_err(v, ref_grads["d"][n]), n) for n, v in zip(prod["discriminator"]._flat.names, ts.d_views))
    assert worst[0] < 5e-3, ("d", worst)
    # whole flat gradient buffers (what Adam / the all-reduce consume)
    cat = lambda names, d: torch.cat([d[n].reshape(-1) for n in names])
    catv = lambda views: torch.cat([v.reshape(-1) for v in views])  # (the flat buffers carry alignment padding)
    assert l2_err(catv(ts.g_views), cat(gnames, ref_grads["g"])) < 2e-3
    assert l2_err(catv(ts.o_views), cat(onames, ref_grads["ocr"])) < 5e-3
    assert l2_err(catv(ts.d_views), cat(prod["discriminator"]._flat.names, ref_grads["d"])) < 2e-3
    if do_pl:
        assert abs(float(prod["pl_mean"]) - float(st["pl_mean"])) <= 2e-4 * max(1.0, abs(float(st["pl_mean"])))
    for n, v in prod["discriminator"].state_dict().items():
        assert l2_err(v, st["D"][n]) < 1e-3, ("D after Adam", n)


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
    ]


def _run_conv_case(dev, case, seed):
    """forward (+ data gradient + filter gradient for the non-transposed forms) vs float64 torch; returns the kernel
    names these launches select."""
    import torch.nn.functional as F
    from textboxgan_amd import native as N, ops
    B, Cc, Mo, H, W, k, stride, pad, transposed = case
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, Cc, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(k, k, Cc, Mo, generator=g, dtype=torch.float64) / math.sqrt(k * k * Cc)
    rel = lambda a, r: float((a.double().cpu() - r).abs().max() / (r.abs().max() + 1e-30))
    names = set()
    prof = ops.PROFILE
    prof.enable()
    try:
        if transposed:
            ref = F.conv_transpose2d(x, w.permute(2, 3, 0, 1), stride=stride)
            y = ops.conv2d_raw(x.float().to(dev), w.float().to(dev), Mo, k, k, (ref.shape[2], ref.shape[3]), stride, (0, 0),
                               transposed=True)
            assert rel(y, ref) < 3e-5, ("transposed", case)
        else:
            xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
            ref = F.conv2d(xr, wr.permute(3, 2, 0, 1), stride=stride, padding=pad)
            dy = torch.randn(*ref.shape, generator=g, dtype=torch.float64)
            gx, gw = torch.autograd.grad(ref, (xr, wr), dy)
            geom = ops._Geom(stride, pad, k, k, (H, W), (ref.shape[2], ref.shape[3]))
            xd, wd, dyd = x.float().to(dev), w.float().to(dev), dy.float().to(dev)
            assert rel(ops._fwd_launch(xd, wd, geom), ref.detach()) < 3e-5, ("fwd", case)
            assert rel(ops._bwd_data_launch(dyd, wd, geom), gx) < 3e-5, ("dgrad", case)
            assert rel(ops._bwd_weight_launch(xd, dyd, geom, Cc, Mo), gw) < 5e-5, ("wgrad", case)
    finally:
        recs = prof.collect()
        prof.disable()
    names.update(k for k in recs if k.startswith("conv_"))
    return names


def test_every_step_instantiation_is_oracle_compared(dev):
    """run the BASELINE configs[1] step (B = 16, full widths, eager) with the launch recorder on, collect every
    conv_fprop / conv_wgrad instantiation name (tbg_conv2d_kernel_name: a pure function of the descriptor), and require
    each one to be selected by at least one of the float64-oracle-compared cases above (which are run here)."""
    from bench import bench_init_, synthetic_batch
    from textboxgan_amd import ops
    from textboxgan_amd.training_step import build_trainer_state
    covered = set()
    for i, case in enumerate(_conv_cases_for_coverage()):
        covered |= _run_conv_case(dev, case, 500 + i)
    cfg = Config(batch_size_per_gpu=16)
    st = build_trainer_state(cfg, dev, seed=0, use_graphs=False)
    bench_init_(st)
    batch = synthetic_batch(cfg, dev, 1234)
    ts = st["training_step"]
    ops.PROFILE.enable()
    try:
        ts.dist_train_step(batch["real_images"], batch["ocr_images"], batch["input_words"], batch["ocr_labels"], False, False,
                           1e-4)
        used = {k for k in ops.PROFILE.collect() if k.startswith("conv_")}
    finally:
        ops.PROFILE.disable()
    assert used, "the recorder saw no convolution launches"
    missing = sorted(used - covered)
    assert not missing, f"instantiations launched by the step but never compared with the oracle: {missing}"
