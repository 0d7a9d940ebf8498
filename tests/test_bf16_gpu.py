"""-m gpu: the bf16-in / fp32-accumulate path (BASELINE configs[2]).

Two kinds of comparison, tolerances stated here because bf16 has no counterpart in the (fp32-only) reference:
  * EXACTNESS of the kernels: tbg_conv2d_bf16 / tbg_conv2d_wgrad_bf16 against a float64 oracle evaluated on operands
    PRE-ROUNDED to bf16 exactly as the kernels round them (x*s then RNE; the filter RNE).  Products of two bf16 values are
    exact in fp32, so only the fp32 summation order differs: 3e-5 relative, the same bar as the fp32 kernels.
  * ACCURACY of the mode against the fp32 oracle (what a user of configs[2] gives up): bf16 keeps 8 significand bits
    (unit round-off 2^-9 = 2e-3 per operand); a 1152-term dot product of independently rounded operands is off by
    ~3e-3 relative in L2; through the 12-layer generator / 7-block discriminator and their backward passes we allow
    2e-2 (forward images, losses) and 8e-2 (gradient sets), relative L2.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ref_model as M, ref_ops as R
from textboxgan_amd.config import Config, small_config

pytestmark = pytest.mark.gpu


def rel_err(a, ref):
    a = a.detach().double().cpu()
    ref = ref.detach().double().cpu()
    assert a.shape == ref.shape, (a.shape, ref.shape)
    return float((a - ref).abs().max() / (ref.abs().max() + 1e-30))


def l2_err(a, ref):
    a = a.detach().double().cpu()
    ref = ref.detach().double().cpu()
    assert a.shape == ref.shape, (a.shape, ref.shape)
    return float((a - ref).norm() / (ref.norm() + 1e-30))


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64) * scale


def bf(t):
    """round an fp64 tensor the way the kernels do: to fp32 first (what sits in HBM), then RNE to bf16."""
    return t.float().bfloat16().double()


def test_weight_pack_bf16_layout(dev):
    from textboxgan_amd import ops
    w = rnd(3, 3, 13, 20, seed=60).float()
    for transpose in (False, True):
        for flip in (False, True):
            pf = ops.pack_filter(w.to(dev), transpose, flip, bf16=True)
            assert pf.bf16 and pf.data.dtype == torch.bfloat16
            src = (torch.flip(w, (0, 1)) if flip else w).reshape(9, 13, 20)
            gemm = src.permute(0, 2, 1) if transpose else src  # [T, C, M]
            T, Cc, Mo = gemm.shape
            assert (pf.T, pf.C, pf.M) == (T, Cc, Mo)
            C8 = (Cc + 7) // 8
            ref = torch.zeros(T, C8 * 8, Mo)
            ref[:, :Cc] = gemm
            ref = ref.reshape(T, C8, 8, Mo).permute(0, 1, 3, 2).contiguous().bfloat16()
            assert torch.equal(pf.data.cpu().reshape(T, C8, Mo, 8), ref)


def _wgrad_reference(x, w, dy, stride, pad, gw_rounded):
    """tbg_conv2d_wgrad_bf16 keeps the EXACT fp32 kernel for tile rows narrower than 8 pixels (dy maps <= 4 wide: the 4x4
    head): there the reference is the un-rounded fp64 filter gradient."""
    if dy.shape[3] > 4:
        return gw_rounded
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y = F.conv2d(xr, wr.permute(3, 2, 0, 1), stride=stride, padding=pad)
    return torch.autograd.grad(y, wr, dy)[0]


BF_CASES = [
    # B, C, M, H, W, k, stride, pad, note
    (2, 16, 32, 16, 64, 3, (1, 1), (1, 1), "3x3 small"),
    (2, 128, 128, 16, 64, 3, (1, 1), (1, 1), "3x3 128ch"),
    (3, 24, 64, 12, 40, 3, (1, 1), (1, 1), "3x3 odd sizes, C % 16 != 0"),
    (2, 64, 64, 64, 256, 3, (1, 1), (1, 1), "64x256 tile"),
    (4, 512, 512, 4, 16, 3, (1, 1), (1, 1), "4x16 512ch (split-K)"),
    (4, 513, 512, 4, 4, 3, (1, 1), (1, 1), "head conv 513 on 4x4 (wgrad: fp32 fallback, narrow rows)"),
    (2, 64, 128, 34, 66, 3, (2, 2), (0, 0), "3x3 stride 2 VALID"),
    (2, 32, 48, 10, 34, 3, (1, 2), (0, 0), "3x3 stride (1,2)"),
    (2, 3, 64, 16, 64, 1, (1, 1), (0, 0), "1x1 fromRGB-like"),
    (3, 40, 72, 9, 17, 1, (2, 2), (0, 0), "1x1 stride 2"),
    (2, 100, 130, 7, 9, 3, (1, 1), (1, 1), "awkward channels / map"),
]


@pytest.mark.parametrize("case", BF_CASES, ids=[c[-1] for c in BF_CASES])
def test_conv_bf16_three_passes_vs_rounded_oracle(dev, case):
    from textboxgan_amd import ops
    B, Cc, Mo, H, W, k, stride, pad, _ = case
    x = rnd(B, Cc, H, W, seed=1)
    w = rnd(k, k, Cc, Mo, seed=2) / math.sqrt(k * k * Cc)
    xr, wr = bf(x).requires_grad_(True), bf(w).requires_grad_(True)
    y = F.conv2d(xr, wr.permute(3, 2, 0, 1), stride=stride, padding=pad)
    dy = rnd(*y.shape, seed=3)
    gx, gw = torch.autograd.grad(y, (xr, wr), bf(dy))
    gw = _wgrad_reference(x, w, dy, stride, pad, gw)
    geom = ops._Geom(stride, pad, k, k, (H, W), (y.shape[2], y.shape[3]))
    xd, wd, dyd = x.float().to(dev), w.float().to(dev), dy.float().to(dev)
    with ops.compute_dtype("bf16"):
        assert rel_err(ops._fwd_launch(xd, wd, geom), y) < 3e-5, "fwd"
        assert rel_err(ops._bwd_data_launch(dyd, wd, geom), gx) < 3e-5, "dgrad"
        assert rel_err(ops._bwd_weight_launch(xd, dyd, geom, Cc, Mo), gw) < 5e-5, "wgrad"
    # and the mode's accuracy against the un-rounded fp64 result
    y64 = F.conv2d(x, w.permute(3, 2, 0, 1), stride=stride, padding=pad)
    with ops.compute_dtype("bf16"):
        assert l2_err(ops._fwd_launch(xd, wd, geom), y64) < 1e-2


@pytest.mark.parametrize("dims", [(3, 40, 72, 5, 36), (2, 64, 64, 8, 32), (1, 130, 70, 2, 128)], ids=["ragged", "one-tile", "3x2 tiles"])
def test_wgrad_bf16_vector_staging(dev, dims):
    """the float4-staged bf16 filter-gradient instance, with both scale vectors (rounded AFTER scaling, as the kernel
    does) and the fused additive term, against the rounded-operand float64 reference."""
    from textboxgan_amd import ops, native as N
    B, C, M, H, W = dims
    x, dy = rnd(B, C, H, W, seed=40), rnd(B, M, H, W, seed=41)
    xs, ds = rnd(B, C, seed=42).abs() + 0.5, rnd(B, M, seed=43).abs() + 0.5
    addw, addq = rnd(3, 3, C, M, seed=44), rnd(C, M, seed=45)
    f32 = lambda t: t.float().double()  # the kernel multiplies in fp32 before rounding to bf16
    xr = bf((f32(x) * f32(xs)[:, :, None, None]).float().double())
    dyr = bf((f32(dy) * f32(ds)[:, :, None, None]).float().double())
    w = torch.zeros(3, 3, C, M, dtype=torch.float64, requires_grad=True)
    (ref,) = torch.autograd.grad(F.conv2d(xr, w.permute(3, 2, 0, 1), padding=1), w, dyr)
    ref = 0.7 * ref + 0.3 * f32(addw) * f32(addq)[None, None]
    f = lambda t: t.float().to(dev).contiguous()
    g = ops._Geom((1, 1), (1, 1), 3, 3, (H, W), (H, W))
    desc = N.WgradDesc(B, M, C, H, W, H, W, 3, 3, 1, 1, 1, 1, C * M, M, 1, 0.7)
    assert N.wgrad_kernel_name(desc, True) == "conv_wgrad_bf16_kernel<2, 2, 9, 64, 1, 1>"
    with ops.compute_dtype("bf16"):
        dw = ops._bwd_weight_launch(f(x), f(dy), g, C, M, alpha=0.7, x_scale=f(xs), dy_scale=f(ds), add=(f(addw), f(addq), 0.3))
    assert rel_err(dw, ref) < 5e-5


TBF_CASES = [
    (2, 16, 32, 4, 16, (2, 2), "up 4x16"),
    (2, 128, 128, 16, 64, (2, 2), "up 16x64 128ch"),
    (2, 32, 16, 8, 9, (1, 2), "transposed stride (1,2)"),
    (3, 512, 256, 4, 16, (2, 2), "up 512->256 (split-K)"),
    (2, 128, 64, 32, 128, (2, 2), "up 128->64, wide tile"),
]


@pytest.mark.parametrize("case", TBF_CASES, ids=[c[-1] for c in TBF_CASES])
def test_conv_bf16_transposed_vs_rounded_oracle(dev, case):
    from textboxgan_amd import ops
    B, Cc, Mo, H, W, stride, _ = case
    x = rnd(B, Cc, H, W, seed=14)
    w = rnd(3, 3, Cc, Mo, seed=15) / math.sqrt(9 * Cc)
    ref = F.conv_transpose2d(bf(x), bf(w).permute(2, 3, 0, 1), stride=stride)
    with ops.compute_dtype("bf16"):
        y = ops.conv2d_raw(x.float().to(dev), w.float().to(dev), Mo, 3, 3, (ref.shape[2], ref.shape[3]), stride, (0, 0),
                           transposed=True)
    assert rel_err(y, ref) < 3e-5
    # filter gradient of the up-conv form (S = x on the small grid, L = dy_up on the large one; wgrad_raw strides)
    if stride == (2, 2):
        dyu = rnd(*ref.shape, seed=16)
        xr, dyr = bf(x), bf(dyu)
        wt = torch.zeros(3, 3, Cc, Mo, dtype=torch.float64, requires_grad=True)
        (gw,) = torch.autograd.grad(F.conv_transpose2d(xr, wt.permute(2, 3, 0, 1), stride=stride), wt, dyr)
        dw = torch.empty(3, 3, Cc, Mo, device=dev)
        with ops.compute_dtype("bf16"):
            ops.wgrad_raw(x.float().to(dev), dyu.float().to(dev), 3, 3, (2, 2), (0, 0), dw, Cc * Mo, 1, Mo, 1.0)
        assert rel_err(dw, gw) < 5e-5


def test_conv_bf16_fused_epilogue_and_modulation(dev):
    """in_scale is applied BEFORE the bf16 rounding (x*s is what the matrix core sees); the epilogue stays fp32."""
    from textboxgan_amd import ops, native as N
    B, Cc, Mo, H, W = 3, 32, 48, 8, 32
    x, w = rnd(B, Cc, H, W, seed=16), rnd(3, 3, Cc, Mo, seed=17)
    s, d = rnd(B, Cc, seed=18) + 1.0, rnd(B, Mo, seed=19).abs() + 0.5
    noise, bias, res = rnd(B, 1, H, W, seed=20), rnd(Mo, seed=21), rnd(B, Mo, H, W, seed=22)
    strength = torch.tensor(0.37, dtype=torch.float64)
    alpha = 0.123
    xs = bf((x.float() * s.float()[:, :, None, None]).double())
    pre = F.conv2d(xs, bf(w).permute(3, 2, 0, 1), padding=1) * alpha * d.float().double()[:, :, None, None]
    pre = pre + noise.float().double() * strength.float().double() + bias.float().double()[None, :, None, None] * 0.5
    ref = (F.leaky_relu(pre, 0.2) * math.sqrt(2) + res.float().double()) * 0.7
    f = lambda t: t.float().to(dev).contiguous()
    dd, bd, nd, sd, rd = f(d), f(bias), f(noise), f(strength), f(res)
    epi = N.epilogue(out_scale=dd, bias=bd, noise=nd, strength=sd, residual=rd, alpha=alpha, bias_mul=0.5, act=N.ACT_LRELU,
                     res_scale=0.7)
    with ops.compute_dtype("bf16"):
        y = ops.conv2d_raw(f(x), f(w), Mo, 3, 3, (H, W), (1, 1), (1, 1), in_scale=f(s), epi=epi)
    assert rel_err(y, ref) < 3e-5


def test_conv_bf16_random_shapes(dev):
    from textboxgan_amd import ops
    rng = np.random.RandomState(4321)
    for case in range(16):
        k = int(rng.choice([1, 3]))
        stride = (1, 1) if rng.rand() < 0.6 else tuple(int(v) for v in rng.choice([1, 2], size=2))
        pad = (k // 2, k // 2) if stride == (1, 1) else (0, 0)
        B = int(rng.randint(1, 5))
        Cc = int(rng.choice([1, 3, 5, 8, 17, 33, 64, 100, 130]))
        Mo = int(rng.choice([1, 3, 7, 16, 31, 40, 64, 96, 129]))
        H = int(rng.randint(max(1, k if pad == (0, 0) else 1), 20))
        W = int(rng.randint(max(1, k if pad == (0, 0) else 1), 40))
        x = rnd(B, Cc, H, W, seed=100 + case)
        w = rnd(k, k, Cc, Mo, seed=200 + case)
        xr, wr = bf(x).requires_grad_(True), bf(w).requires_grad_(True)
        y = F.conv2d(xr, wr.permute(3, 2, 0, 1), stride=stride, padding=pad)
        dy = rnd(*y.shape, seed=300 + case)
        gx, gw = torch.autograd.grad(y, (xr, wr), bf(dy))
        gw = _wgrad_reference(x, w, dy, stride, pad, gw)
        geom = ops._Geom(stride, pad, k, k, (H, W), (y.shape[2], y.shape[3]))
        xd, wd, dyd = x.float().to(dev), w.float().to(dev), dy.float().to(dev)
        tag = (case, B, Cc, Mo, H, W, k, stride)
        with ops.compute_dtype("bf16"):
            assert rel_err(ops._fwd_launch(xd, wd, geom), y.detach()) < 3e-5, ("fwd", tag)
            assert rel_err(ops._bwd_data_launch(dyd, wd, geom), gx) < 3e-5, ("dgrad", tag)
            assert rel_err(ops._bwd_weight_launch(xd, dyd, geom, Cc, Mo), gw) < 5e-5, ("wgrad", tag)


def _todev(rand, dev):
    return {k: ([t.to(dev) for t in v] if isinstance(v, list) else (v.to(dev) if torch.is_tensor(v) else v))
            for k, v in rand.items()}


def test_training_step_bf16_vs_fp32_oracle_small(dev):
    """whole step (r1 + pl) in bf16 mode against the fp32 CPU oracle: stated bf16 tolerances (module docstring)."""
    from conftest import ocr_oracle
    from textboxgan_amd.training_step import build_trainer_state
    cfg = small_config(4)
    st = M.make_state(cfg, seed=0, bench_init=True)
    batch, rand = M.make_batch(cfg), M.make_rand(cfg, seed=99)
    prod = build_trainer_state(cfg, dev, seed=0, compute_dtype="bf16")
    prod["generator"].load_state_dict({k: v.clone() for k, v in st["G"].items()})
    prod["discriminator"].load_state_dict({k: v.clone() for k, v in st["D"].items()})
    ts = prod["training_step"]
    ocr_cpu = ocr_oracle(cfg.max_char_number)
    ref_losses, ref_grads = M.training_step(st, cfg, batch["real_images"], batch["ocr_images"], batch["input_words"],
                                            batch["ocr_labels"], True, True, 1e-4, rand, ocr_cpu.serve, return_grads=True)
    b = {k: v.to(dev) for k, v in batch.items()}
    losses = ts.dist_train_step(b["real_images"], b["ocr_images"], b["input_words"], b["ocr_labels"], True, True, 1e-4,
                                rand=_todev(rand, dev))
    torch.cuda.synchronize()
    flat = lambda t: [float(x) for x in t] if isinstance(t, tuple) else [float(t)]
    got = flat(losses[0]) + flat(losses[1]) + flat(losses[2])
    exp = flat(ref_losses[0]) + flat(ref_losses[1]) + flat(ref_losses[2])
    for name, a, e in zip(("reg_g", "g", "pl", "reg_d", "d", "r1", "ocr"), got, exp):
        assert abs(a - e) <= 3e-2 * max(1.0, abs(e)), (name, a, e)
    gnames = [n for n in prod["generator"]._flat.names if n.startswith(("latent_encoder.", "synthesis."))]
    cat = lambda names, d: torch.cat([d[n].reshape(-1) for n in names])
    catv = lambda views: torch.cat([v.reshape(-1) for v in views])  # (the flat buffers carry alignment padding)
    assert l2_err(catv(ts.g_views), cat(gnames, ref_grads["g"])) < 8e-2
    assert l2_err(catv(ts.d_views), cat(prod["discriminator"]._flat.names, ref_grads["d"])) < 8e-2


def test_generator_bf16_full_width_accuracy(dev):
    """generator forward at the real widths, bf16 mode vs the fp32 oracle (B = 2): relative L2 of the image <= 2e-2."""
    from textboxgan_amd import ops
    from textboxgan_amd.models import Generator
    cfg = Config(batch_size_per_gpu=2)
    batch, rand = M.make_batch(cfg), M.make_rand(cfg, seed=99, with_pl=False)
    P = M.init_generator(cfg, seed=0, bench_init=True)
    G = Generator(cfg)
    G.load_state_dict({k: v.clone() for k, v in P.items()})
    G = G.to(dev)
    with torch.no_grad():
        ref = M.generator(P, cfg, batch["input_words"], rand["z"], rand, training=False)
        with ops.compute_dtype("bf16"):
            got = G((batch["input_words"].to(dev), rand["z"].to(dev)), training=False, rand=_todev(rand, dev))
    err = l2_err(got, ref)
    assert err < 2e-2, err


def test_config3_step_bs32_bf16_runs_and_tracks_fp32(dev):
    """BASELINE configs[2]: per-GPU batch 32, bf16, PL + R1 variants, HIP-graph replay.  Same seeds, fp32 vs bf16 HIP
    paths: first-step losses within the bf16 tolerance; three more steps stay finite."""
    from bench import bench_init_, synthetic_batch
    from textboxgan_amd.training_step import build_trainer_state
    cfg = Config(batch_size_per_gpu=32)
    batch = synthetic_batch(cfg, dev, 1234)
    first = {}
    for dtype in ("f32", "bf16"):
        st = build_trainer_state(cfg, dev, seed=0, use_graphs=False, compute_dtype=dtype)
        bench_init_(st)
        ts = st["training_step"]
        torch.manual_seed(5)
        l = ts.dist_train_step(batch["real_images"], batch["ocr_images"], batch["input_words"], batch["ocr_labels"], True, True,
                               1e-4)
        first[dtype] = [float(v) for grp in l[:2] for v in grp] + [float(l[2])]
        if dtype == "bf16":
            ts.use_graphs = True
            for i in range(4):
                l = ts.dist_train_step(batch["real_images"], batch["ocr_images"], batch["input_words"], batch["ocr_labels"],
                                       False, i % 2 == 1, 1e-4)
            assert all(math.isfinite(float(v)) for grp in l[:2] for v in grp) and math.isfinite(float(l[2]))
            for p in list(st["generator"].parameters()) + list(st["discriminator"].parameters()):
                assert torch.isfinite(p).all()
        del st, ts
        torch.cuda.empty_cache()
    for name, a, e in zip(("reg_g", "g", "pl", "reg_d", "d", "r1", "ocr"), first["bf16"], first["f32"]):
        assert abs(a - e) <= 3e-2 * max(1.0, abs(e)), (name, a, e)


@pytest.mark.parametrize("bf16", [False, True], ids=["fp32", "bf16"])
@pytest.mark.parametrize("case", [(2, 128, 128, 16, 64), (3, 512, 256, 4, 16), (2, 40, 72, 9, 13), (8, 64, 64, 32, 128)],
                         ids=["16x64", "4x16-splitk", "odd", "bf16-auto-size"])
def test_merged_class_transposed_kernel(dev, case, bf16):
    """the merged-class form of the stride-2 transposed 3x3 convolution (all four output-parity classes from one staged halo
    tile; tbg_conv2d_*_variant 5, and the library's own choice for large bf16 launches) == conv_transpose2d, natural and
    padded output sizes, with the style modulation applied while staging."""
    from textboxgan_amd import ops
    B, Cc, Mo, H, W = case
    x = rnd(B, Cc, H, W, seed=41)
    w = rnd(3, 3, Cc, Mo, seed=42) / math.sqrt(9 * Cc)
    s = rnd(B, Cc, seed=43) + 1.0
    xs = x.float() * s.float()[:, :, None, None]
    xr, wr = (bf(xs.double()), bf(w)) if bf16 else (xs.double(), w.float().double())
    ref = F.conv_transpose2d(xr, wr.permute(2, 3, 0, 1), stride=2)
    f = lambda t: t.float().to(dev).contiguous()
    try:
        ops.TUNING.force_variant = 5
        with ops.compute_dtype("bf16" if bf16 else "f32"):
            y = ops.conv2d_raw(f(x), f(w), Mo, 3, 3, (2 * H + 1, 2 * W + 1), (2, 2), (0, 0), transposed=True, in_scale=f(s))
            y2 = ops.conv2d_raw(f(x), f(w), Mo, 3, 3, (2 * H + 2, 2 * W + 2), (2, 2), (0, 0), transposed=True, in_scale=f(s))
            yflip = ops.conv2d_raw(f(x), f(w), Mo, 3, 3, (2 * H + 1, 2 * W + 1), (2, 2), (0, 0), transposed=True, flip=True,
                                   in_scale=f(s))
    finally:
        ops.TUNING.force_variant = 0
    assert rel_err(y, ref) < 3e-5
    assert rel_err(y2, F.pad(ref, (0, 1, 0, 1))) < 3e-5
    assert rel_err(yflip, F.conv_transpose2d(xr, torch.flip(wr, (0, 1)).permute(2, 3, 0, 1), stride=2)) < 3e-5
    if bf16 and B * H * W >= 16384:  # the library picks the merged form by itself here
        from textboxgan_amd import native as N
        d = N.ConvDesc(B, Cc, Mo, H, W, 2 * H + 1, 2 * W + 1, 3, 3, 2, 2, 0, 0, 1, 0, Mo, 1)
        assert N.conv_kernel_name(d, True, True).endswith("true, true, false>")
        with ops.compute_dtype("bf16"):
            y3 = ops.conv2d_raw(f(x), f(w), Mo, 3, 3, (2 * H + 1, 2 * W + 1), (2, 2), (0, 0), transposed=True, in_scale=f(s))
        assert rel_err(y3, ref) < 3e-5
