"""-m gpu: fused HIP layers and whole networks against the oracle (float64 on CPU)."""
import math

import numpy as np
import pytest
import torch

from oracle import ref_model as M, ref_ops as R
from textboxgan_amd.config import small_config

from conftest import arith_modes

pytestmark = pytest.mark.gpu


def rel_err(a, ref):
    a = a.detach().double().cpu()
    ref = ref.detach().double().cpu()
    assert a.shape == ref.shape, (a.shape, ref.shape)
    return float((a - ref).abs().max() / (ref.abs().max() + 1e-30))


def l2_err(a, ref):
    """relative L2 error: robust to the rare LeakyReLU mask flip between fp32 and the float64 oracle (a flip is a
    discrete event that moves single elements of a whole-network gradient by O(1%) of the maximum)."""
    a = a.detach().double().cpu()
    ref = ref.detach().double().cpu()
    assert a.shape == ref.shape, (a.shape, ref.shape)
    return float((a - ref).norm() / (ref.norm() + 1e-30))


def rnd(*shape, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64)


def to64(P):
    return {k: v.double() for k, v in P.items()}


@arith_modes
@pytest.mark.parametrize("up", [False, True], ids=["conv_1", "conv_0_up"])
@pytest.mark.parametrize("shape", [(3, 16, 24, 8, 32), (2, 128, 128, 16, 64), (16, 512, 512, 4, 16)],
                         ids=["small", "128ch", "512ch-splitk"])
def test_modconv_fused_fwd_bwd(dev, up, shape):
    """fused modulated conv (+up) + noise + bias + lrelu: forward and all seven gradients."""
    from textboxgan_amd import ops
    B, I, O, H, W = shape
    if up and I == 512:
        O = 256
    sd = 20
    x, style = rnd(B, I, H, W, seed=1), rnd(B, sd, seed=2)
    w, mw, mb = rnd(3, 3, I, O, seed=3), rnd(sd, I, seed=4), rnd(I, seed=5) * 0.1
    Ho, Wo = (2 * H, 2 * W) if up else (H, W)
    noise, strength, bias = rnd(B, 1, Ho, Wo, seed=6), torch.tensor(0.3, dtype=torch.float64), rnd(O, seed=7) * 0.2
    leaves = [t.requires_grad_(True) for t in (x, w, mw, mb, strength, bias)]
    y = R.t_modulated_conv2d(x, style, w, mw, mb, up=up, demodulate=True, fused=False)
    out = R.t_bias_act(R.t_noise(y, noise, strength), bias, "lrelu")
    dout = rnd(*out.shape, seed=8)
    s_ref = R.t_bias_act(R.t_dense(style, mw), mb, "linear") + 1.0
    grads = torch.autograd.grad(out, leaves, dout, retain_graph=True)
    (gs_total,) = torch.autograd.grad(out, s_ref, dout, allow_unused=True) if False else (None,)

    f = lambda t: t.detach().float().to(dev).contiguous()
    xd, wd, nd, std, bd = f(x).requires_grad_(True), f(w).requires_grad_(True), f(noise), f(strength).requires_grad_(True), f(bias).requires_grad_(True)
    mwd, mbd = f(mw).requires_grad_(True), f(mb).requires_grad_(True)
    s = ops.dense_bias_act(f(style), mwd, mbd, 1.0 / math.sqrt(sd), 1.0, lrelu=False, offset=1.0)  # style dense (fused mode)
    fn = ops.modconv_up_fused if up else ops.modconv_fused
    outd = fn(xd, wd, s, nd, std, bd)  # demodulation inside the node
    assert rel_err(outd, out) < 3e-5
    gd = torch.autograd.grad(outd, (xd, wd, mwd, mbd, std, bd), f(dout))
    # ONE criterion, element-wise 2e-4 of max, for every gradient.  LeakyReLU is piecewise linear: a pre-activation within
    # rounding of zero can take the other branch under a different (equally valid) fp32 summation order, and the gradient of
    # THAT branch is what the product must deliver.  So the expected gradients are the oracle's float64 gradients of the
    # branch assignment the product's forward took (its output signs); with no flipped activation that is the oracle's own
    # lrelu gradient bit for bit.  The flips themselves are bounded separately: few, and only where the oracle's
    # pre-activation is within fp32 rounding of zero.
    pre = R.t_bias_act(R.t_noise(y, noise, strength), bias, "linear")
    took_pos = outd.detach().cpu() > 0
    flipped = took_pos != (pre.detach() > 0)
    flips = int(flipped.sum())
    assert flips <= 1e-5 * out.numel() + 1, flips
    if flips:
        assert float(pre.detach()[flipped].abs().max()) <= 2e-5 * float(pre.detach().abs().max()), "flip away from the kink"
    slope = torch.where(took_pos, 1.0, 0.2).double() * math.sqrt(2.0)
    grads = torch.autograd.grad(pre * slope, leaves, dout)
    for name, a, b in zip(("dx", "dw", "dmod_w", "dmod_b", "dstrength", "dbias"), gd, grads):
        err = (a.detach().double().cpu() - b.detach().double()).abs() / (b.detach().double().abs().max() + 1e-30)
        n_bad = int((err > 2e-4).sum())
        assert n_bad == 0, (name, n_bad, float(err.max()), flips)


@pytest.mark.parametrize("dims", [(5, 24, 40), (32, 256, 256), (16, 256, 512), (64, 100, 130), (16, 512, 1), (33, 96, 65),
                                  (8, 1536, 48), (4, 768, 20)],
                         ids=["small", "mapping", "style", "ragged", "head", "rows33", "library-gemm", "k768"])
def test_dense_bias_act(dev, dims):
    """equalised-LR dense + bias (+lrelu*sqrt2 / +offset): forward and the three gradients vs the oracle layers -- the
    one-launch kernels (K <= ops.TUNING.dense_small_k) at the step's sizes and ragged ones, and the library-GEMM form above."""
    from textboxgan_amd import ops
    B, I, O = dims
    lrmul = 0.01
    x, w, b = rnd(B, I, seed=21), rnd(I, O, seed=22) / lrmul, rnd(O, seed=23) / lrmul
    leaves = [t.requires_grad_(True) for t in (x, w, b)]
    f = lambda t: t.detach().float().to(dev).contiguous().requires_grad_(True)
    for lrelu, offset in ((True, 0.0), (False, 1.0), (False, 0.0), (True, 0.5)):
        ref = R.t_bias_act(R.t_dense(x, w, lrmul=lrmul), b, "lrelu" if lrelu else "linear", lrmul=lrmul) + offset
        dout = rnd(*ref.shape, seed=24)
        grads = torch.autograd.grad(ref, leaves, dout)
        xd, wd, bd = f(x), f(w), f(b)
        # both launch forms are product paths (ops.dense_bias_act routes by activation and K): test each directly
        form = ops._DenseBiasAct if I <= ops.TUNING.dense_small_k else ops._DenseBiasActGemm
        out = form.apply(xd, wd, bd, lrmul / math.sqrt(I), lrmul, lrelu, offset)
        assert rel_err(out, ref) < 1e-5
        assert rel_err(ops.dense_bias_act(xd, wd, bd, lrmul / math.sqrt(I), lrmul, lrelu=lrelu, offset=offset), ref) < 1e-5
        gd = torch.autograd.grad(out, (xd, wd, bd), dout.float().to(dev), retain_graph=True)
        for name, a, b_ in zip(("dx", "dw", "db"), gd, grads):
            assert rel_err(a, b_) < 1e-5, (lrelu, offset, name)
        # any subset of the three gradients (frozen weights: projector.py; detached input: the mapping network's z)
        (only_dx,) = torch.autograd.grad(out, (xd,), dout.float().to(dev), retain_graph=True)
        assert rel_err(only_dx, grads[0]) < 1e-5
        (only_db,) = torch.autograd.grad(out, (bd,), dout.float().to(dev))
        assert rel_err(only_db, grads[2]) < 1e-5


@pytest.mark.parametrize("dims", [(5, 100, (3, 70, 130), None), (16, 512, (512, 512, 256, 128, 64, 3), (0, 0, 1, 2, 3, 5)),
                                  (33, 64, (65,), None)], ids=["ragged", "synthesis-like (shared and unused rows)", "rows33"])
def test_style_affines_one_launch(dev, dims):
    """all style affines of the synthesis network in one launch each way (tbg_dense_multi_*) vs the float64 definition
    s_l = coef * style[:, l] @ W_l + b_l + 1: outputs, d(style) (written slot by slot), dW_l, db_l; and with frozen weights."""
    from textboxgan_amd import ops
    R_, K, Ns, rows = dims
    L = len(Ns)
    rows = tuple(range(L)) if rows is None else rows
    style = rnd(R_, max(rows) + 1, K, seed=61).requires_grad_(True)
    ws = [rnd(K, n, seed=62 + i).requires_grad_(True) for i, n in enumerate(Ns)]
    bs = [rnd(n, seed=82 + i).requires_grad_(True) for i, n in enumerate(Ns)]
    coef = 1.0 / math.sqrt(K)
    refs = [coef * style[:, rows[l]] @ ws[l] + bs[l] + 1.0 for l in range(L)]
    douts = [rnd(R_, n, seed=92 + i) for i, n in enumerate(Ns)]
    gref = torch.autograd.grad(refs, [style] + ws + bs, douts, retain_graph=True)
    f = lambda t: t.detach().float().to(dev).contiguous().requires_grad_(True)
    sd, wd, bd = f(style), [f(w) for w in ws], [f(b) for b in bs]
    outs = ops.style_affines(sd, wd, bd, coef, rows)
    for o, r in zip(outs, refs):
        assert rel_err(o, r) < 1e-5
    gd = torch.autograd.grad(outs, [sd] + wd + bd, [d.float().to(dev) for d in douts], retain_graph=True)
    for i, (a, b_) in enumerate(zip(gd, gref)):
        assert rel_err(a, b_) < 1e-5, i
    (only_style,) = torch.autograd.grad(outs, [sd], [d.float().to(dev) for d in douts], retain_graph=True)
    assert rel_err(only_style, gref[0]) < 1e-5
    # a subset of the outputs used (the others' slots of d(style) must come back as zeros)
    (part,) = torch.autograd.grad(outs[0], [sd], douts[0].float().to(dev))
    (pref,) = torch.autograd.grad(refs[0], [style], douts[0])
    assert rel_err(part, pref) < 1e-5


@pytest.mark.parametrize("dims", [(3, 7, 12, 8), (16, 25, 512, 256)], ids=["small", "ocr-encoder"])
def test_frozen_bilstm_layer(dev, dims):
    """batched-GEMM + pointwise-kernel BiLSTM layer vs torch.nn.LSTM (float64 CPU): output and input gradient."""
    from textboxgan_amd import ops
    B, T, In, H = dims
    torch.manual_seed(5)
    ref = torch.nn.LSTM(In, H, num_layers=1, bidirectional=True, batch_first=True).double()
    x = rnd(B, T, In, seed=31).requires_grad_(True)
    y = ref(x)[0]
    dy = rnd(*y.shape, seed=32)
    (gx,) = torch.autograd.grad(y, x, dy)
    g = lambda n: torch.stack([getattr(ref, f"{n}_l0"), getattr(ref, f"{n}_l0_reverse")]).detach().float().to(dev).contiguous()
    xd = x.detach().float().to(dev).requires_grad_(True)
    yd = ops.frozen_bilstm_layer(xd, g("weight_ih"), g("weight_hh"), (g("bias_ih") + g("bias_hh")).contiguous())
    assert rel_err(yd, y) < 1e-5
    (gxd,) = torch.autograd.grad(yd, xd, dy.float().to(dev))
    assert rel_err(gxd, gx) < 2e-5


def test_frozen_attn_decoder(dev):
    """one-node attention decoder (HIP attention context + LSTM-cell launches) vs the oracle's decoder loop (oracle/ref_ocr.py,
    float64, the same frozen weights): logits and d/d(encoder output) for a gradient that reaches every step."""
    from oracle.ref_ocr import OcrOracle
    from textboxgan_amd.aster import AsterLikeOCRHip
    hip = AsterLikeOCRHip().to(dev)
    ref = OcrOracle(hip.state_dict(), max_steps=hip.max_steps, dtype=torch.float64)
    enc = (rnd(5, 25, 512, seed=41) * 0.5).requires_grad_(True)
    lg = ref._decode(enc)
    dl = rnd(*lg.shape, seed=42)
    (g,) = torch.autograd.grad(lg, enc, dl)
    encd = enc.detach().float().to(dev).requires_grad_(True)
    lgd = hip._decode(encd)
    # greedy feedback: the comparison is only meaningful while both decoders pick the same symbols
    assert torch.equal(lgd.argmax(-1).cpu(), lg.argmax(-1))
    assert rel_err(lgd, lg) < 1e-4
    (gd,) = torch.autograd.grad(lgd, encd, dl.float().to(dev))
    assert rel_err(gd, g) < 1e-4


@pytest.mark.parametrize("dims,masked", [((3, 24, 8, 32), False), ((3, 24, 8, 32), True), ((4, 128, 64, 256), True),
                                         ((2, 130, 6, 40), True), ((2, 512, 2, 8), False)],
                         ids=["small", "small-masked", "last-block-masked", "ragged-masked", "first-block"])
def test_torgb_fused(dev, dims, masked):
    """ToRGB (+ skip add, + mask_text_box as the launch's epilogue on the last block) forward and all six gradients; the
    128-channel 64x256 case is the step's last block (8 pixel chunks of the deterministic channel-Gram reduction)."""
    from textboxgan_amd import ops
    B, I, H, W = dims
    sd = 16
    x, style, skip = rnd(B, I, H, W, seed=11), rnd(B, sd, seed=12), rnd(B, 3, H, W, seed=13)
    w, mw, mb, b = rnd(1, 1, I, 3, seed=14), rnd(sd, I, seed=15), rnd(I, seed=16) * 0.1, rnd(3, seed=17)
    cw = max(W // 8, 1)
    words = torch.tensor([[1] * (1 + (i * 3) % 8) + [0] * (7 - (i * 3) % 8) for i in range(B)])
    leaves = [t.requires_grad_(True) for t in (x, w, mw, mb, b, skip)]
    y = R.t_modulated_conv2d(x, style, w, mw, mb, up=False, demodulate=False, fused=False)
    out = R.t_bias_act(y, b, "linear") + skip
    if masked:
        out = R.t_mask_text_box(out, words, cw)
    dout = rnd(*out.shape, seed=18)
    grads = torch.autograd.grad(out, leaves, dout)
    f = lambda t: t.detach().float().to(dev).contiguous().requires_grad_(True)
    xd, wd, mwd, mbd, bd, skd = [f(t) for t in (x, w, mw, mb, b, skip)]
    s = torch.addmm(mbd + 1.0, style.float().to(dev), mwd / math.sqrt(sd))
    colmask = (words != 0).float().to(dev) if masked else None
    outd = ops.torgb_fused(xd, wd, s, bd, skd, colmask, cw if masked else 0)
    assert rel_err(outd, out) < 3e-5
    gd = torch.autograd.grad(outd, (xd, wd, mwd, mbd, bd, skd), dout.float().to(dev), retain_graph=True)
    for name, a, b_ in zip(("dx", "dw", "dmod_w", "dmod_b", "db", "dskip"), gd, grads):
        assert rel_err(a, b_) < 2e-4, name
    # the channel Gram is reduced without atomics: two runs are bit-identical
    gd2 = torch.autograd.grad(ops.torgb_fused(xd, wd, s, bd, skd, colmask, cw if masked else 0), (wd, mwd), dout.float().to(dev))
    assert torch.equal(gd2[0], gd[1]) and torch.equal(gd2[1], gd[2])


def _load(module, P, dev):
    sd = {k: v.detach().float() for k, v in P.items()}
    missing = module.load_state_dict(sd, strict=True)
    return module.to(dev)


@arith_modes
@pytest.mark.parametrize("mode", ["fused", "composable"])
def test_discriminator_matches_oracle(dev, mode):
    from textboxgan_amd.models import Discriminator
    cfg = small_config(4)
    P = to64(M.init_discriminator(cfg, seed=3, bench_init=True))
    for v in P.values():
        v.requires_grad_(True)
    img = rnd(4, 3, 64, 256, seed=21).requires_grad_(True)
    scores = M.discriminator(P, cfg, img)
    gs = rnd(4, 1, seed=22)
    names = list(P.keys())
    grads = torch.autograd.grad(scores, [img] + [P[n] for n in names], gs)
    D = _load(Discriminator(cfg), P, dev)
    imgd = img.detach().float().to(dev).requires_grad_(True)
    sc = D(imgd, mode=mode)
    assert rel_err(sc, scores) < 1e-4
    pd = dict(D.named_parameters())
    gd = torch.autograd.grad(sc, [imgd] + [pd[n] for n in names], gs.float().to(dev))
    assert l2_err(gd[0], grads[0]) < 5e-4, "d/dimage"
    for n, a, b in zip(names, gd[1:], grads[1:]):
        assert l2_err(a, b) < 5e-4 and rel_err(a, b) < 5e-2, n


@arith_modes
def test_discriminator_joint_pass_matches_two_oracle_calls(dev):
    """D over [fake; real] with parts=2 (the d-step's single pass) == the oracle's two separate calls: scores, the
    d-pass gradients (sum over both halves, image gradient pruned) and -- in the G-loss pass's first-half mode -- the
    gradient w.r.t. the fake images with the filter gradients pruned.  Also with the staged (cuts) form."""
    from textboxgan_amd import ops
    from textboxgan_amd.models import Discriminator
    cfg = small_config(4)
    P = to64(M.init_discriminator(cfg, seed=3, bench_init=True))
    for v in P.values():
        v.requires_grad_(True)
    names = list(P.keys())
    fake, real = rnd(4, 3, 64, 256, seed=31).requires_grad_(True), rnd(4, 3, 64, 256, seed=32)
    sf, sr = M.discriminator(P, cfg, fake), M.discriminator(P, cfg, real)
    gf, gr = rnd(4, 1, seed=33), rnd(4, 1, seed=34)
    (dfake_ref,) = torch.autograd.grad(sf, fake, gf, retain_graph=True)                     # G-loss pass
    dpar_ref = torch.autograd.grad([sf, sr], [P[n] for n in names], [gf, gr])              # D-loss pass
    D = _load(Discriminator(cfg), P, dev)
    pd = dict(D.named_parameters())
    f = lambda t: t.detach().float().to(dev)
    for cuts in (None, [1, 3]):
        faked = f(fake).requires_grad_(True)
        both = torch.cat([faked, f(real)], dim=0)
        out = D(both, parts=2, cuts=cuts)
        scores, taps = out if cuts is not None else (out, None)
        assert rel_err(scores[:4], sf) < 1e-4 and rel_err(scores[4:], sr) < 1e-4
        ops.FLAGS.skip_d_wgrad, ops.FLAGS.d_first_half = True, 4
        try:
            (dfake,) = torch.autograd.grad(scores[:4], faked, f(gf), retain_graph=True)
        finally:
            ops.FLAGS.skip_d_wgrad, ops.FLAGS.d_first_half = False, 0
        assert l2_err(dfake, dfake_ref) < 5e-4, "d/dfake in first-half mode"
        ops.FLAGS.skip_image_grad = True
        try:
            gd = torch.autograd.grad(scores, [pd[n] for n in names], torch.cat([f(gf), f(gr)], dim=0))
        finally:
            ops.FLAGS.skip_image_grad = False
        for n, a, b in zip(names, gd, dpar_ref):
            assert l2_err(a, b) < 5e-4 and rel_err(a, b) < 5e-2, (cuts, n)
        if cuts is not None:
            assert len(taps) == 2 and all(t.shape[0] == 8 for t in taps)


@arith_modes
@pytest.mark.parametrize("mode", ["fused", "composable", "fused2"])
@pytest.mark.parametrize("training", [True, False])
def test_generator_matches_oracle(dev, mode, training):
    from textboxgan_amd.models import Generator
    cfg = small_config(4)
    P = to64(M.init_generator(cfg, seed=5, bench_init=True))
    names = [k for k in P if k not in M.NON_TRAINABLE]
    for n in names:
        P[n].requires_grad_(True)
    batch = M.make_batch(cfg)
    rand = M.make_rand(cfg, seed=7)
    rand64 = {k: ([t.double() for t in v] if isinstance(v, list) else (v.double() if torch.is_tensor(v) else v))
              for k, v in rand.items()}
    G = _load(Generator(cfg), P, dev)  # before the oracle call: training=True updates w_avg in place
    img = M.generator(P, cfg, batch["input_words"], rand64["z"], rand64, training=training)
    gi = rnd(*img.shape, seed=23)
    grads = torch.autograd.grad(img, [P[n] for n in names], gi, allow_unused=True)
    randd = {k: ([t.to(dev) for t in v] if isinstance(v, list) else (v.to(dev) if torch.is_tensor(v) else v))
             for k, v in rand.items()}
    imgd = G((batch["input_words"].to(dev), randd["z"]), training=training, rand=randd, mode=mode)
    assert rel_err(imgd, img) < 1e-4
    pd = dict(G.named_parameters())
    gd = torch.autograd.grad(imgd, [pd[n] for n in names], gi.float().to(dev), allow_unused=True)
    for n, a, b in zip(names, gd, grads):
        if b is None:
            assert a is None or float(a.abs().max()) == 0.0, n
            continue
        assert l2_err(a, b) < 1e-2 and rel_err(a, b) < 5e-2, n  # fp32 through ~30 layers vs the float64 oracle
    if training:
        assert rel_err(G.latent_encoder.w_avg, P["latent_encoder.w_avg"]) < 1e-5


def test_generator_hello_known_answer_geometry(dev):
    """config 1 of BASELINE.json (plumbing): 'Hello' -> [1,3,64,256] -> uint8 crop [64,160,3]."""
    from textboxgan_amd.models import Generator, generator_output_to_uint8, mask_text_box
    from textboxgan_amd.char_tokens import string_to_main_int_sequence
    from textboxgan_amd.config import cfg
    words = torch.from_numpy(string_to_main_int_sequence(["Hello"])).to(dev)
    torch.manual_seed(0)
    G = Generator(cfg).to(dev)
    z = torch.randn(1, cfg.z_dim, device=dev)
    with torch.no_grad():
        img = G((words, z), training=False, truncation_psi=1.0)
    assert img.shape == (1, 3, 64, 256) and torch.isfinite(img).all()
    u8 = generator_output_to_uint8(mask_text_box(img, words, cfg.char_width))[0, :, : 32 * 5]
    assert u8.shape == (64, 160, 3) and u8.dtype == torch.uint8


@arith_modes
def test_ocr_hip_matches_oracle_network(dev):
    """AsterInferer(AsterLikeOCRHip) -- MFMA convolutions with folded BatchNorm, lstm_step / attn_ctx kernels, hand-written
    backward -- against the ORACLE's restatement of the wrapper (ref_model.ocr_convert_inputs / ocr_call, per sample as
    aster_inferer.py:28-37) and of the network (oracle/ref_ocr.py: no code shared with the product; float64; the product
    network's frozen weights handed over as data): logits and d(CE)/d(image)."""
    from oracle.ref_ocr import OcrOracle
    from textboxgan_amd.aster import AsterInferer, AsterLikeOCRHip
    from textboxgan_amd.config import Config
    cfg = Config()
    hip = AsterInferer(model=AsterLikeOCRHip()).to(dev)
    x = (rnd(4, 3, 64, 256, seed=31) * 0.5).float()
    labels = torch.tensor([[5, 6, 1, 1, 1, 1, 1, 1], [2, 3, 4, 5, 6, 7, 8, 9], [7, 1, 1, 1, 1, 1, 1, 1], [3, 4, 5, 6, 1, 1, 1, 1]])
    ref = {}
    for dt in (torch.float64, torch.float32):
        orc = OcrOracle(hip.model.state_dict(), max_steps=8, dtype=dt)
        xr = x.to(dt).requires_grad_(True)
        lg_ref = M.ocr_call(M.ocr_convert_inputs(xr, labels, cfg), orc.serve, 8)
        ce = torch.nn.functional.cross_entropy(lg_ref.reshape(-1, lg_ref.shape[-1]), labels.reshape(-1), reduction="sum")
        ref[dt] = (lg_ref, torch.autograd.grad(ce, xr)[0])
    xx = x.to(dev).requires_grad_(True)
    lg = hip(hip.convert_inputs(xx, labels.to(dev)))
    ce = torch.nn.functional.cross_entropy(lg.reshape(-1, lg.shape[-1]), labels.reshape(-1).to(dev), reduction="sum")
    (g,) = torch.autograd.grad(ce, xx)
    # logits: against the float64 oracle (measured 2e-5)
    assert rel_err(lg, ref[torch.float64][0]) < 1e-4
    # d(CE)/d(image): the third sample is a one-character word -- its 32-pixel crop is stretched x8 to 256 columns
    # (aster_inferer.py:166-187), which hands the localisation network's max-pools EXACT ties between neighbouring pixels.  A
    # tie's subgradient is a choice: every fp32 evaluation sees the tie and takes the first index, float64 sees the two
    # candidates 1e-17 apart and follows the rounding -- a discrete event worth 1e-2 of the maximum on 230 pixels of that
    # sample (the oracle's OWN fp32 evaluation differs from its float64 one by exactly that: 1.0e-2 max, 3.0e-3 relative L2).
    # So: element-wise against the oracle evaluated in fp32 (same ties, still no shared code) at the old bar, and relative L2
    # against float64.
    assert rel_err(g, ref[torch.float32][1]) < 5e-3
    assert l2_err(g, ref[torch.float64][1]) < 5e-3


@arith_modes
@pytest.mark.parametrize("kind", ["up3x3", "conv3x3", "torgb1x1"])
def test_modconv_composable_block_exact(dev, kind):
    """any-order path of one modulated conv (x*s -> HIP conv primitive(s) -> *d) vs the oracle: forward and
    all first-order gradients to 1e-5 -- no activation in between, so no mask-flip noise here."""
    from textboxgan_amd.models import ModulatedConv2D
    cfg = small_config(4)
    sd = cfg.style_dim
    up, k, demod = {"up3x3": (True, 3, True), "conv3x3": (False, 3, True), "torgb1x1": (False, 1, False)}[kind]
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
    s = m.style(style.float().to(dev))
    yd = m.conv_composable(xd, s, m.demod(s, "composable"))
    gd = torch.autograd.grad(yd, (xd, m.w, m.mod_dense.w, m.mod_bias.b), dout.float().to(dev))
    assert rel_err(yd, y) < 1e-5
    for a, b in zip(gd, grads):
        assert rel_err(a, b) < 1e-5


@arith_modes
def test_hip_path_reproduces_committed_golden_fixtures(dev):
    """tests/golden/*.npz (float64 oracle outputs committed as data): upfirdn2d cases and the whole
    generator -> mask -> discriminator chain at the reduced-channel config on the HIP path."""
    import os
    import numpy as np
    from textboxgan_amd import ops
    from textboxgan_amd.models import Discriminator, Generator, mask_text_box
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    z = np.load(os.path.join(gold, "upfirdn2d.npz"))
    for name in ("blur_up", "blur_down3", "blur_skip", "rgb_up", "skip_dec", "skip_dec_w"):
        ux, uy, dx, dy, p0, p1, p2, p3 = [int(v) for v in z[name + "_p"]]
        x = torch.from_numpy(z[name + "_x"])[..., 0][None].to(dev)  # [1, major, H, W]
        y = ops.upfirdn2d_raw(x.contiguous(), torch.from_numpy(z[name + "_k"]).to(dev), (ux, uy), (dx, dy), (p0, p1, p2, p3))
        assert rel_err(y[0], torch.from_numpy(z[name + "_y"])[..., 0]) < 1e-5, name
    n = np.load(os.path.join(gold, "networks_small.npz"))
    cfg = small_config(2)
    G = _load(Generator(cfg), M.init_generator(cfg, seed=11, bench_init=True), dev)
    D = _load(Discriminator(cfg), M.init_discriminator(cfg, seed=12, bench_init=True), dev)
    words = torch.from_numpy(n["words"]).to(dev)
    rand = dict(noises=[torch.from_numpy(n[f"noise{i}"]).to(dev) for i in range(10)])
    with torch.no_grad():
        img = G((words, torch.from_numpy(n["z"]).to(dev)), training=False, rand=rand)
        sc = D(mask_text_box(img, words, cfg.char_width))
    assert rel_err(img[:, :, 31, :], torch.from_numpy(n["image_row"])) < 1e-4
    chk = torch.tensor([float(img.sum()), float(img.abs().sum()), float(img.square().sum())])
    assert rel_err(chk, torch.from_numpy(n["image_checksum"])) < 1e-4
    assert rel_err(sc, torch.from_numpy(n["scores"])) < 1e-3


@arith_modes
def test_discriminator_block_fused_skip_gradient_equals_two_nodes(dev):
    """ops._ConvBiasActSkipFused (conv_0 and the skip FIR of a DiscriminatorBlock as one node: the input gradient conv^T(dt) +
    FIR^T(dxd) formed by the data-gradient launch's epilogue) against the two-node form with the autograd engine's add --
    scores, every parameter gradient and d/d(image), on the whole batch and in the G-loss pass's first-half mode
    (FLAGS.d_first_half: only the leading samples are differentiated, discriminator.py:68-84)."""
    from textboxgan_amd import ops
    from textboxgan_amd.models import Discriminator
    cfg = small_config(4)
    torch.manual_seed(3)
    D = Discriminator(cfg).to(dev)
    img = (torch.randn(8, 3, cfg.char_height, cfg.image_width, device=dev) * 0.5)
    res = {}
    for fused in (True, False):
        ops.TUNING.fuse_skip_grad = fused
        try:
            with ops.STATE_LOCK, ops.filter_cache():
                x = img.clone().requires_grad_(True)
                sc = D(x, parts=2)
                g_full = torch.autograd.grad(sc.square().sum(), [x] + list(D.parameters()), retain_graph=True)
                ops.FLAGS.skip_d_wgrad, ops.FLAGS.d_first_half = True, 4
                try:
                    (g_half,) = torch.autograd.grad(sc[:4].sum(), x)
                finally:
                    ops.FLAGS.skip_d_wgrad, ops.FLAGS.d_first_half = False, 0
            res[fused] = (sc.detach(), [g.detach() for g in g_full], g_half[:4].detach())
        finally:
            ops.TUNING.fuse_skip_grad = True
    assert torch.equal(res[True][0], res[False][0])
    for a, b in zip(res[True][1], res[False][1]):
        assert l2_err(a, b) < 1e-6
    assert l2_err(res[True][2], res[False][2]) < 1e-6
