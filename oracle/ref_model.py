"""ORACLE (test infrastructure, NOT product code) -- the whole TextBoxGAN training step.

A functional torch-CPU restatement of the path the reference runs in
``TrainingStep._train_step`` (training_step.py:138-222): generator (word encoder, latent
encoder with w_avg EMA + style mixing, synthesis), mask, discriminator, OCR wrapper, the
three losses, lazy R1 / path-length regularisers, three gradient sets taken at the
pre-update weights and three Keras-semantics Adam updates, plus the caller's g_clone EMA
(train.py:208).  Stock ``torch.nn.functional`` convolutions (oneDNN) + autograd.

All randomness (z, z2, mixing decision, dropout mask, noise maps, PL latents/noise) is
INJECTED through ``rand`` because TF's Philox streams cannot be reproduced.

PARITY UNPINNED (see oracle/ref_ops.py header): no TF here, the reference has no tests.

Parameters live in flat dicts keyed by the reference's attribute paths,
weights in the reference's own layouts ([k,k,I,O] conv filters, [in,out] dense).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import ref_ops as R

Params = Dict[str, torch.Tensor]
FIR = [1, 3, 3, 1]


# ----------------------------------------------------------------------------------------
# parameter construction (reference initialisers)
# ----------------------------------------------------------------------------------------
def _normal(g, shape, std=1.0):
    return torch.from_numpy((g.standard_normal(size=shape) * std).astype(np.float32))


def init_generator(cfg, seed=0, bench_init=False) -> Params:
    """Shapes/initialisers: word_encoder.py:28-37, dense.py:16-21, modulated_conv2d.py:55-64,
    bias_act.py:20-23, noise.py:8-10, latent_encoder.py:29-37, synthesis_block.py:91-135."""
    g = np.random.default_rng(seed)
    P: Params = {}
    fm = cfg.generator_feat_maps
    sd = cfg.style_dim
    P["word_encoder.w_embedding"] = _normal(g, (cfg.main_vocab, cfg.embedding_out_dim))
    P["word_encoder.w0_embedding"] = torch.zeros(1, cfg.embedding_out_dim)
    lim = math.sqrt(6.0 / (cfg.embedding_out_dim + cfg.word_encoder_dense_dim))  # glorot uniform
    P["word_encoder.fc.kernel"] = torch.from_numpy(
        g.uniform(-lim, lim, size=(cfg.embedding_out_dim, cfg.word_encoder_dense_dim)).astype(np.float32))
    P["word_encoder.fc.bias"] = torch.zeros(cfg.word_encoder_dense_dim)

    def modconv(prefix, k, cin, cout):
        P[prefix + ".w"] = _normal(g, (k, k, cin, cout))
        P[prefix + ".mod_dense.w"] = _normal(g, (sd, cin))
        P[prefix + ".mod_bias.b"] = torch.zeros(cin)

    def bias(name, n):
        P[name] = _normal(g, (n,), 0.1) if bench_init else torch.zeros(n)

    def torgb(prefix, cin):
        modconv(prefix + ".conv", 1, cin, 3)
        bias(prefix + ".apply_bias.b", 3)

    torgb("synthesis.initial_torgb", fm[0])
    prev = fm[0]
    for i, f in enumerate(fm[1:]):
        pre = f"synthesis.synth_blocks.{i}"
        modconv(pre + ".conv_0", 3, prev, f)
        P[pre + ".apply_noise_0.noise_strength"] = torch.tensor(0.1 if bench_init else 0.0)
        bias(pre + ".apply_bias_act_0.b", f)
        modconv(pre + ".conv_1", 3, f, f)
        P[pre + ".apply_noise_1.noise_strength"] = torch.tensor(0.1 if bench_init else 0.0)
        bias(pre + ".apply_bias_act_1.b", f)
        torgb(f"synthesis.torgbs.{i}", f)
        prev = f
    for i in range(cfg.n_mapping):
        fan_in = cfg.z_dim if i == 0 else sd
        P[f"latent_encoder.g_mapping.dense_layers.{i}.w"] = _normal(g, (fan_in, sd), 1.0 / 0.01)
        bias(f"latent_encoder.g_mapping.bias_act_layers.{i}.b", sd)
    P["latent_encoder.w_avg"] = torch.zeros(sd)
    return P


def init_discriminator(cfg, seed=1, bench_init=False) -> Params:
    """discriminator.py:157-200, conv.py:41-49, from_rgb.py:14-24."""
    g = np.random.default_rng(seed)
    P: Params = {}
    fm = cfg.discrim_feat_maps
    res = cfg.discrim_resolutions

    def bias(name, n):
        P[name] = _normal(g, (n,), 0.1) if bench_init else torch.zeros(n)

    P["initial_fromrgb.conv.w"] = _normal(g, (1, 1, 3, fm[0]))
    bias("initial_fromrgb.apply_bias_act.b", fm[0])
    for i, (f0, f1) in enumerate(zip(fm[:-1], fm[1:])):
        pre = f"blocks.{i}"
        P[pre + ".conv_0.w"] = _normal(g, (3, 3, f0, f0))
        bias(pre + ".apply_bias_act_0.b", f0)
        P[pre + ".conv_1.w"] = _normal(g, (3, 3, f0, f1))
        bias(pre + ".apply_bias_act_1.b", f1)
        P[pre + ".conv_skip.w"] = _normal(g, (1, 1, f0, f1))
    n_f0, n_f1 = fm[-2], fm[-1]
    P["last_block.conv_0.w"] = _normal(g, (3, 3, n_f0 + 1, n_f0))
    bias("last_block.apply_bias_act_0.b", n_f0)
    hf, wf = res[-1]
    P["last_block.dense_1.w"] = _normal(g, (n_f0 * hf * wf, n_f1))
    bias("last_block.apply_bias_act_1.b", n_f1)
    P["last_dense.w"] = _normal(g, (n_f1, 1))
    bias("last_bias.b", 1)
    return P


NON_TRAINABLE = ("word_encoder.w0_embedding", "latent_encoder.w_avg")


def trainable(P: Params, prefixes: Tuple[str, ...]) -> List[str]:
    return [k for k in P if k.startswith(prefixes) and k not in NON_TRAINABLE]


# ----------------------------------------------------------------------------------------
# generator
# ----------------------------------------------------------------------------------------
def word_encoder(P: Params, cfg, words: torch.Tensor, dropout_mask: Optional[torch.Tensor]):
    """word_encoder.py:39-63.  dropout_mask: [B,8,emb] of {0, 1/0.7} or None (inactive)."""
    B = words.shape[0]
    table = torch.cat([P["word_encoder.w0_embedding"], P["word_encoder.w_embedding"]], 0)
    emb = table[words.long()]  # [B, 8, emb]
    if dropout_mask is not None:
        emb = emb * dropout_mask
    x = emb.reshape(B * cfg.max_char_number, cfg.embedding_out_dim)
    x = F.relu(x @ P["word_encoder.fc.kernel"] + P["word_encoder.fc.bias"])
    h0, w0 = cfg.generator_resolutions[0]
    c0 = cfg.generator_feat_maps[0]
    return x.reshape(B, w0, c0, h0).permute(0, 2, 3, 1)  # [B, C0, h0, w0]


def mapping(P: Params, cfg, z: torch.Tensor):
    """mapping_block.py:35-45."""
    x = z * torch.rsqrt(z.square().mean(dim=1, keepdim=True) + 1e-8)
    for i in range(cfg.n_mapping):
        x = R.t_dense(x, P[f"latent_encoder.g_mapping.dense_layers.{i}.w"], 1.0, 0.01)
        x = R.t_bias_act(x, P[f"latent_encoder.g_mapping.bias_act_layers.{i}.b"], "lrelu", 0.01)
    return x


def lerp(a, b, t):
    return a + (b - a) * t


def n_style(cfg) -> int:
    return 3 * (len(cfg.generator_resolutions) - 1)


def latent_encoder(P: Params, cfg, z, training: bool, rand: Optional[dict], truncation_psi=1.0):
    """latent_encoder.py:80-99 (+ :39-78).  Mutates P['latent_encoder.w_avg'] when training."""
    ns = n_style(cfg)
    w = mapping(P, cfg, z)
    wb = w[:, None, :].expand(-1, ns, -1)
    if training:
        with torch.no_grad():
            batch_avg = wb[:, 0].mean(dim=0)
            P["latent_encoder.w_avg"].copy_(lerp(batch_avg, P["latent_encoder.w_avg"], 0.995))
        w2 = mapping(P, cfg, rand["z2"])
        wb2 = w2[:, None, :].expand(-1, ns, -1)
        cutoff = int(rand["mix_cutoff"])  # already resolved: U{1..ns-1} w.p. .9 else ns
        idx = torch.arange(ns)[None, :, None]
        wb = torch.where(idx < cutoff, wb, wb2)
    else:
        wb = lerp(P["latent_encoder.w_avg"], wb, truncation_psi)
    return wb


def to_rgb(P, pre, x, style):
    """to_rgb.py:28-33: 1x1 modconv without demod + bias."""
    y = R.t_modulated_conv2d(x, style, P[pre + ".conv.w"], P[pre + ".conv.mod_dense.w"],
                             P[pre + ".conv.mod_bias.b"], up=False, demodulate=False, fused=False)
    return R.t_bias_act(y, P[pre + ".apply_bias.b"], "linear")


def synthesis(P: Params, cfg, x, style, noises: List[torch.Tensor], fused=False):
    """synthesis_block.py:62-74,137-156."""
    k_up, p0_up, p1_up = R.compute_paddings(FIR, up=True, down=False, is_conv=False)
    y = to_rgb(P, "synthesis.initial_torgb", x, style[:, 0])
    nblocks = len(cfg.generator_resolutions) - 1
    for i in range(nblocks):
        pre = f"synthesis.synth_blocks.{i}"
        s0, s1, s2 = style[:, 3 * i], style[:, 3 * i + 1], style[:, 3 * i + 2]
        for j, (st, up) in enumerate(((s0, True), (s1, False))):
            x = R.t_modulated_conv2d(x, st, P[f"{pre}.conv_{j}.w"], P[f"{pre}.conv_{j}.mod_dense.w"],
                                     P[f"{pre}.conv_{j}.mod_bias.b"], up=up, demodulate=True, fused=fused)
            x = R.t_noise(x, noises[2 * i + j], P[f"{pre}.apply_noise_{j}.noise_strength"])
            x = R.t_bias_act(x, P[f"{pre}.apply_bias_act_{j}.b"], "lrelu")
        y = R.t_simple_upfirdn2d(y, k_up, up=2, pad0=p0_up, pad1=p1_up)
        y = y + to_rgb(P, f"synthesis.torgbs.{i}", x, s2)
    return y


def generator(P: Params, cfg, words, z, rand: dict, training: bool, ret_style=False,
              truncation_psi=1.0, noises_key="noises", fused=False):
    """generator.py:19-43.  Dropout follows the call context: active iff training."""
    we = word_encoder(P, cfg, words, rand.get("dropout_mask") if training else None)
    style = latent_encoder(P, cfg, z, training, rand, truncation_psi)
    if ret_style:
        style = style.clone()  # a distinct tensor so d/dstyle is well defined
    img = synthesis(P, cfg, we, style, rand[noises_key], fused=fused)
    return (img, style) if ret_style else img


# ----------------------------------------------------------------------------------------
# discriminator
# ----------------------------------------------------------------------------------------
def conv2d_layer(w, x, down=False, reduce_height=None):
    """conv.py:51-73."""
    kk = w.shape[0]
    _, coef = R.runtime_coef(w.shape, 1.0, 1.0)
    wc = w * coef
    if down:
        k, pad0, pad1 = R.compute_paddings(FIR, False, True, is_conv=True, convW=kk)
        return R.t_conv_downsample2d(x, wc, k, pad0, pad1, reduce_height)
    return R.t_conv2d_same(x, wc)


def discriminator(P: Params, cfg, images):
    """discriminator.py:202-214 (+ :68-84, :132-142)."""
    x = conv2d_layer(P["initial_fromrgb.conv.w"], images)
    x = R.t_bias_act(x, P["initial_fromrgb.apply_bias_act.b"], "lrelu")
    res = cfg.discrim_resolutions
    for i in range(len(res) - 1):
        pre = f"blocks.{i}"
        rh = res[i][0] != res[i + 1][0]
        residual = x
        x = conv2d_layer(P[pre + ".conv_0.w"], x)
        x = R.t_bias_act(x, P[pre + ".apply_bias_act_0.b"], "lrelu")
        x = conv2d_layer(P[pre + ".conv_1.w"], x, down=True, reduce_height=rh)
        x = R.t_bias_act(x, P[pre + ".apply_bias_act_1.b"], "lrelu")
        residual = conv2d_layer(P[pre + ".conv_skip.w"], residual, down=True, reduce_height=rh)
        x = (x + residual) * (1.0 / math.sqrt(2.0))
    x = R.t_minibatch_std(x, 4)
    x = conv2d_layer(P["last_block.conv_0.w"], x)
    x = R.t_bias_act(x, P["last_block.apply_bias_act_0.b"], "lrelu")
    x = R.t_dense(x, P["last_block.dense_1.w"])
    x = R.t_bias_act(x, P["last_block.apply_bias_act_1.b"], "lrelu")
    x = R.t_dense(x, P["last_dense.w"])
    return R.t_bias_act(x, P["last_bias.b"], "linear")


# ----------------------------------------------------------------------------------------
# OCR wrapper (the network itself is external to the reference: passed in as a callable)
# ----------------------------------------------------------------------------------------
def ocr_convert_inputs(fake_nchw: torch.Tensor, labels: torch.Tensor, cfg, blank_label=1):
    """aster_inferer.py:153-190: NCHW->NHWC, crop to the word's width, bilinear resize
    (TF2 half-pixel centres, no antialias) to 64x256.  Returns NHWC."""
    outs = []
    H, W = cfg.aster_image_dims
    for b in range(fake_nchw.shape[0]):
        img = fake_nchw[b:b + 1]
        blanks = (labels[b] == blank_label).nonzero()
        if blanks.numel() > 0:
            img = img[:, :, :, : int(blanks[0, 0]) * cfg.char_width]
        img = F.interpolate(img, size=(H, W), mode="bilinear", align_corners=False)
        outs.append(img)
    return torch.cat(outs, 0).permute(0, 2, 3, 1)


def ocr_postprocess_simple(logits: torch.Tensor, max_char_number=8):
    """aster_inferer.py:116-151: first 8 steps; pad with 1000*onehot(class 1) if shorter."""
    logits = logits[:, :max_char_number]
    pad = max_char_number - logits.shape[1]
    if pad > 0:
        onehot = torch.zeros(logits.shape[0], pad, logits.shape[2], dtype=logits.dtype)
        onehot[:, :, 1] = 1000.0
        logits = torch.cat([logits, onehot], dim=1)
    return logits


def softmax_cross_entropy_loss(logits, labels, batch_size):
    """ocr_losses.py:8-11."""
    ce = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), labels.reshape(-1).long(), reduction="none")
    return ce.sum() / batch_size


def mean_squared_loss(y_a, y_b, batch_size):
    """ocr_losses.py:14-20 (keras mse = mean over the last axis)."""
    return (y_a - y_b).square().mean(dim=-1).sum() / batch_size


def ocr_combine_logits(forward_logits, backward_logits):
    """aster_inferer.py:84-114."""
    forward_mask = ~(forward_logits.argmax(dim=2) == 1)
    backward_mask = ~(backward_logits.argmax(dim=2) == 1)
    masked_forward = forward_logits[forward_mask]
    masked_backward = torch.flip(backward_logits[backward_mask], (0,))
    crop_masked_forward = masked_forward[: masked_backward.shape[0]]
    crop_masked_backward = masked_backward[: masked_forward.shape[0]]
    forward_max = crop_masked_forward.max(dim=1).values
    backward_max = crop_masked_backward.max(dim=1).values
    combined = torch.where(forward_max[:, None] > backward_max[:, None], crop_masked_forward, crop_masked_backward)
    return combined[None]


def ocr_postprocess_combine(prediction: dict, max_char_number=8):
    """aster_inferer.py:39-82."""
    forward_logits = prediction["forward_logits"][:, :max_char_number]
    backward_logits = prediction["backward_logits"][:, :max_char_number]
    combined_logits = ocr_combine_logits(forward_logits, backward_logits)
    remaining_logits = forward_logits[:, combined_logits.shape[1]:, :]
    padding_len = max_char_number - forward_logits.shape[1]
    padding = torch.zeros(1, padding_len, combined_logits.shape[2], dtype=forward_logits.dtype)
    padding[:, :, 1] = 1000.0
    return torch.cat([combined_logits, remaining_logits, padding], dim=1)


def ocr_call(inputs_nhwc: torch.Tensor, serve_fn: Callable, max_char_number=8, combine_forward_and_backward=False):
    """AsterInferer.call, aster_inferer.py:28-37: the SavedModel is called ONE SAMPLE AT A TIME; each sample's
    ``forward_logits`` [1, T_i, C] (T_i = that sample's own dynamic-decode length) is post-processed and the rows are
    concatenated.  ``serve_fn``: NHWC [1,64,256,3] -> {"forward_logits": [1,T,C]} (the serving signature)."""
    rows = []
    for i in range(inputs_nhwc.shape[0]):
        prediction = serve_fn(inputs_nhwc[i:i + 1])
        if combine_forward_and_backward:
            rows.append(ocr_postprocess_combine(prediction, max_char_number))
        else:
            rows.append(ocr_postprocess_simple(prediction["forward_logits"], max_char_number))
    return torch.cat(rows, dim=0)


def get_ocr_loss(fake, labels, ocr_images, cfg, ocr_fn: Callable):
    """training_step.py:375-402.  ``ocr_fn`` is the serving signature (see ocr_call)."""
    inp = ocr_convert_inputs(fake, labels, cfg)
    logits = ocr_call(inp, ocr_fn, cfg.max_char_number)
    if cfg.ocr_loss_type == "mse":
        real_logits = ocr_call(ocr_images, ocr_fn, cfg.max_char_number)
        return mean_squared_loss(real_logits, logits, cfg.batch_size)
    return softmax_cross_entropy_loss(logits, labels, cfg.batch_size)


# ----------------------------------------------------------------------------------------
# losses and regularisers
# ----------------------------------------------------------------------------------------
def generator_loss(fake_scores, batch_size):
    """gan_losses.py:8-10."""
    return F.softplus(-fake_scores).sum() / batch_size


def discriminator_loss(fake_scores, real_scores, batch_size):
    """gan_losses.py:13-16."""
    return (F.softplus(fake_scores) + F.softplus(-real_scores)).sum() / batch_size


def r1_reg(D: Params, cfg, real_images):
    """training_step.py:349-373."""
    real = real_images.detach().clone().requires_grad_(True)
    real_scores = discriminator(D, cfg, real)
    (g,) = torch.autograd.grad(real_scores.sum(), real, create_graph=True)
    pen = g.square().sum(dim=(1, 2, 3))
    pen = pen * (0.5 * 10.0) * cfg.d_opt.reg_interval
    return real_scores, pen.sum() / cfg.batch_size


def path_length_reg(G: Params, cfg, words, rand: dict, state: dict):
    """training_step.py:300-347.  state['pl_mean'] (python float tensor) is updated BEFORE use
    and read back as a constant (variable read: no gradient through the mean)."""
    Bp = cfg.batch_size_per_gpu
    shrink = 2 if Bp // 2 >= 1 else Bp
    pl_mb = max(1, Bp // shrink)
    img, style = generator(G, cfg, words[:pl_mb], rand["pl_z"], rand, training=False, ret_style=True,
                           noises_key="pl_noises")
    pl_noise = rand["pl_noise"] * (1.0 / math.sqrt(float(cfg.image_width) * float(cfg.char_height)))
    (g,) = torch.autograd.grad((img * pl_noise).sum(), style, create_graph=True)
    lengths = g.square().sum(dim=2).mean(dim=1).sqrt()
    with torch.no_grad():
        state["pl_mean"] = state["pl_mean"] + 0.01 * (lengths.mean() - state["pl_mean"])
    pen = (lengths - state["pl_mean"]).square() * shrink * cfg.g_opt.reg_interval
    return pen.sum() / cfg.batch_size


# ----------------------------------------------------------------------------------------
# Keras-semantics Adam (optimizer_v2/adam.py -> ResourceApplyAdam) and the g_clone EMA
# ----------------------------------------------------------------------------------------
class AdamTF:
    """m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ;
    theta -= lr * sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps)   (eps NOT bias corrected)."""

    def __init__(self, opt):
        self.lr, self.b1, self.b2, self.eps = opt.learning_rate, opt.beta1, opt.beta2, opt.epsilon
        self.iterations = 0
        self.m: Dict[str, torch.Tensor] = {}
        self.v: Dict[str, torch.Tensor] = {}

    def apply(self, P: Params, names: List[str], grads):
        t = self.iterations + 1
        lr_t = self.lr * math.sqrt(1.0 - self.b2 ** t) / (1.0 - self.b1 ** t)
        with torch.no_grad():
            for n, g in zip(names, grads):
                if g is None:
                    continue
                m = self.m.setdefault(n, torch.zeros_like(P[n]))
                v = self.v.setdefault(n, torch.zeros_like(P[n]))
                m.mul_(self.b1).add_(g, alpha=1.0 - self.b1)
                v.mul_(self.b2).addcmul_(g, g, value=1.0 - self.b2)
                P[n].sub_(lr_t * m / (v.sqrt() + self.eps))
        self.iterations = t


def ema_update(clone: Params, src: Params, beta=0.99):
    """generator.py:48-59: clone <- lerp(src, clone, beta); w_avg copied (beta 0)."""
    with torch.no_grad():
        for k in clone:
            b = 0.0 if "w_avg" in k else beta
            clone[k].copy_(lerp(src[k], clone[k], b))


# ----------------------------------------------------------------------------------------
# the training step
# ----------------------------------------------------------------------------------------
def make_state(cfg, seed=0, bench_init=False) -> dict:
    G = init_generator(cfg, seed, bench_init)
    D = init_discriminator(cfg, seed + 1, bench_init)
    return dict(
        G=G, D=D, g_clone={k: v.clone() for k, v in G.items()},
        g_opt=AdamTF(cfg.g_opt.lazy_reg_rescaled()),
        ocr_opt=AdamTF(cfg.g_opt.lazy_reg_rescaled()),
        d_opt=AdamTF(cfg.d_opt.lazy_reg_rescaled()),
        pl_mean=torch.tensor(0.0),
    )


def _req(P: Params, names):
    for n in names:
        P[n].requires_grad_(True)
    return [P[n] for n in names]


def training_step(state: dict, cfg, real_images, ocr_images, input_words, ocr_labels,
                  do_r1_reg: bool, do_pl_reg: bool, ocr_loss_weight: float, rand: dict,
                  ocr_fn: Callable, update_clone=False, return_grads=False):
    """training_step.py:138-222 + _backpropagates_gradient :224-235 (one replica; the
    all-reduce of a multi-replica run is a SUM of these per-replica gradients)."""
    G, D = state["G"], state["D"]
    Bg = cfg.batch_size
    g_names = trainable(G, ("synthesis.", "latent_encoder."))
    o_names = trainable(G, ("synthesis.", "word_encoder."))
    d_names = list(D.keys())
    _req(G, sorted(set(g_names + o_names)))
    _req(D, d_names)

    fake = generator(G, cfg, input_words, rand["z"], rand, training=True)
    fake = R.t_mask_text_box(fake, input_words, cfg.char_width)

    fake_scores = discriminator(D, cfg, fake)
    g_loss = generator_loss(fake_scores, Bg)
    pl_penalty = path_length_reg(G, cfg, input_words, rand, state) if do_pl_reg else torch.tensor(0.0)
    reg_g_loss = g_loss + pl_penalty

    if do_r1_reg:
        real_scores, r1_penalty = r1_reg(D, cfg, real_images)
    else:
        real_scores = discriminator(D, cfg, real_images)
        r1_penalty = torch.tensor(0.0)
    d_loss = discriminator_loss(fake_scores, real_scores, Bg)
    reg_d_loss = d_loss + r1_penalty

    ocr_loss = get_ocr_loss(fake, ocr_labels, ocr_images, cfg, ocr_fn)
    ocr_loss_w = ocr_loss_weight * ocr_loss

    grads_g = torch.autograd.grad(reg_g_loss, [G[n] for n in g_names], retain_graph=True, allow_unused=True)
    # (test aid: the weighted OCR loss's image gradient on its own, so that a lower-precision generator can be checked given it)
    dfake_ocr = torch.autograd.grad(ocr_loss_w, fake, retain_graph=True)[0] if return_grads else None
    grads_o = torch.autograd.grad(ocr_loss_w, [G[n] for n in o_names], retain_graph=True, allow_unused=True)
    grads_d = torch.autograd.grad(reg_d_loss, [D[n] for n in d_names], allow_unused=True)

    for P in (G, D):
        for v in P.values():
            v.requires_grad_(False)
    state["g_opt"].apply(G, g_names, grads_g)
    state["ocr_opt"].apply(G, o_names, grads_o)
    state["d_opt"].apply(D, d_names, grads_d)
    if update_clone:
        ema_update(state["g_clone"], G)

    losses = ((reg_g_loss.detach(), g_loss.detach(), pl_penalty.detach()),
              (reg_d_loss.detach(), d_loss.detach(), r1_penalty.detach()),
              (ocr_loss_w / ocr_loss_weight).detach())
    if return_grads:
        return losses, dict(g=dict(zip(g_names, grads_g)), ocr=dict(zip(o_names, grads_o)),
                            d=dict(zip(d_names, grads_d)), fake=fake.detach(), dfake_ocr=dfake_ocr.detach())
    return losses


# ----------------------------------------------------------------------------------------
# synthetic inputs + injected randomness
# ----------------------------------------------------------------------------------------
# the two vocabularies of the reference (config/char_tokens.py:4-9) and the Keras Tokenizer rule (:12-17: char_level, no
# lower-casing, "<OOV>" = index 1, fitted characters 2.. in order of first appearance) restated here, independent of the product's
# host code: main ids are (token - 1) (utils/utils.py:66-85: pad / OOV -> 0, characters 1..69), ASTER labels keep the token
# (utils/utils.py:87-105: pad / OOV = 1 = the recogniser's end-of-sequence class, characters 2..95)
_MAIN_CHARS = "0123456789abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ-'.!?,\""
_ASTER_CHARS = "0123456789abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~"


def main_to_aster_labels(input_words: np.ndarray) -> np.ndarray:
    """main-vocabulary ids (0 = pad, i = i-th character of the main vector) -> ASTER labels (1 = pad / EOS)."""
    table = np.ones(len(_MAIN_CHARS) + 1, dtype=np.int32)
    for i, ch in enumerate(_MAIN_CHARS):
        j = _ASTER_CHARS.find(ch)
        table[i + 1] = j + 2 if j >= 0 else 1
    return table[np.asarray(input_words)]


def _tokens(text: str, chars: str):
    """Keras Tokenizer(char_level=True, lower=False, oov_token="<OOV>") fitted on ``chars`` (config/char_tokens.py:12-17):
    "<OOV>" = 1, the fitted characters 2.. in order of first appearance (every character occurs once)."""
    return [chars.find(ch) + 2 if chars.find(ch) >= 0 else 1 for ch in text]


def _pad_post(seqs, maxlen: int, value: int) -> np.ndarray:
    """keras pad_sequences(maxlen, value, padding="post") with its DEFAULT truncating="pre": a longer sequence keeps its LAST
    maxlen tokens (utils/utils.py:80-85,102-105 pass no truncating argument)."""
    out = np.full((len(seqs), maxlen), value, dtype=np.int32)
    for r, q in enumerate(seqs):
        q = q[len(q) - maxlen:] if len(q) > maxlen else q
        out[r, : len(q)] = q
    return out


def string_to_main_int_sequence(words, max_char_number=8) -> np.ndarray:
    """utils/utils.py:66-85: main-vocabulary tokens, padded with 1, minus 1 (pad / OOV -> 0, characters 1..69)."""
    return _pad_post([_tokens(w, _MAIN_CHARS) for w in words], max_char_number, 1) - 1


def string_to_aster_int_sequence(words, max_char_number=8) -> np.ndarray:
    """utils/utils.py:87-105: ASTER-vocabulary tokens, padded with 1 (= the recogniser's end-of-sequence class)."""
    return _pad_post([_tokens(w, _ASTER_CHARS) for w in words], max_char_number, 1)


def make_batch(cfg, seed=1234, rank=0):
    g = np.random.default_rng(seed + rank)
    B = cfg.batch_size_per_gpu
    L = g.integers(1, cfg.max_char_number + 1, size=B)
    words = np.zeros((B, cfg.max_char_number), dtype=np.int32)
    for b in range(B):
        words[b, : L[b]] = g.integers(1, cfg.main_vocab + 1, size=L[b])
    labels = main_to_aster_labels(words)
    real = g.uniform(-1, 1, size=(B, 3, cfg.char_height, cfg.image_width)).astype(np.float32)
    for b in range(B):
        real[b, :, :, cfg.char_width * L[b]:] = 0.0
    return dict(real_images=torch.from_numpy(real), ocr_images=torch.tensor(0.0),
                input_words=torch.from_numpy(words), ocr_labels=torch.from_numpy(labels))


def make_rand(cfg, seed=99, with_pl=True):
    g = np.random.default_rng(seed)
    B = cfg.batch_size_per_gpu
    ns = n_style(cfg)

    def nrm(*shape):
        return torch.from_numpy(g.standard_normal(size=shape).astype(np.float32))

    res = cfg.generator_resolutions[1:]
    rand = dict(z=nrm(B, cfg.z_dim), z2=nrm(B, cfg.z_dim))
    u = g.uniform()
    rand["mix_cutoff"] = int(g.integers(1, ns)) if u < 0.9 else ns
    rand["dropout_mask"] = torch.from_numpy(
        (g.uniform(size=(B, cfg.max_char_number, cfg.embedding_out_dim)) < 0.7).astype(np.float32) / 0.7)
    rand["noises"] = [nrm(B, 1, h, w) for (h, w) in res for _ in range(2)]
    if with_pl:
        shrink = 2 if B // 2 >= 1 else B
        pb = max(1, B // shrink)
        rand["pl_z"] = nrm(pb, cfg.z_dim)
        rand["pl_noise"] = nrm(pb, 3, cfg.char_height, cfg.image_width)
        rand["pl_noises"] = [nrm(pb, 1, h, w) for (h, w) in res for _ in range(2)]
    return rand
