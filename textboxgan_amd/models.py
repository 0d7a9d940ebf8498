"""Generator / discriminator of TextBoxGAN on the HIP kernels.

The module tree reproduces the reference's attribute paths (= its ``tf.train.Checkpoint`` keys,
SURVEY section 5; reference models/custom_stylegan2/{generator,discriminator,latent_encoder}.py,
layers/*.py, models/word_encoder.py), and parameters keep the reference's layouts, so
``state_dict()`` keys/shapes are the checkpoint layout.  Two execution modes:

* ``mode="fused"``      first-order training/inference path: fused HIP layers (ops.*_fused).
* ``mode="composable"`` every conv / FIR / scale / activation is a HIP primitive whose backward is again a
                        primitive -- gradients of ANY order, with respect to activations AND parameters (R1's
                        discriminator pass; any caller that records a gradient).
* ``mode="fused2"``     (generator, device tensors only) the synthesis layers and toRGB as the twice-differentiable
                        node pairs of ``ops2``: first order like "fused"; a RECORDED gradient (create_graph=True) is
                        supported with respect to activations / styles / demodulation only and must run under
                        ``ops.FLAGS.no_filter_grads`` -- exactly the path-length pass (training_step.py:300-347),
                        which is its one caller.  Anything else that records gradients takes "composable".

Mapping MLP, word encoder and the discriminator's dense head stay PyTorch-ROCm GEMMs
(BASELINE.json north_star); minibatch-std stays torch (true second-order term for R1).
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops, ops2
from .config import Config
from .native import ACT_LINEAR, ACT_LRELU, SQRT2

FIR = (1, 3, 3, 1)


def _coef(weight_shape, gain=1.0, lrmul=1.0) -> float:
    """layers/commons.py:4-12."""
    fan_in = 1
    for v in weight_shape[:-1]:
        fan_in *= int(v)
    return gain / math.sqrt(fan_in) * lrmul


class Dense(nn.Module):
    """layers/dense.py:6-29 (equalised-LR dense; torch GEMM)."""

    def __init__(self, fan_in, fmaps, gain=1.0, lrmul=1.0):
        super().__init__()
        self.gain, self.lrmul = gain, lrmul
        self.w = nn.Parameter(torch.randn(fan_in, fmaps) / lrmul)

    def forward(self, x):
        return x.reshape(x.shape[0], -1) @ (self.w * _coef(self.w.shape, self.gain, self.lrmul))


class BiasAct(nn.Module):
    """layers/bias_act.py:7-34 (parameter holder; the add/activation is fused into the producer)."""

    def __init__(self, n, lrmul=1.0, act="linear"):
        super().__init__()
        self.lrmul, self.act = lrmul, act
        self.b = nn.Parameter(torch.zeros(n))

    def forward(self, x):
        b = self.b * self.lrmul
        x = x + (b if x.dim() == 2 else b.reshape(1, -1, 1, 1))
        return F.leaky_relu(x, 0.2) * SQRT2 if self.act == "lrelu" else x


class Noise(nn.Module):
    """layers/noise.py:4-22."""

    def __init__(self):
        super().__init__()
        self.noise_strength = nn.Parameter(torch.zeros(()))


class Mapping(nn.Module):
    """layers/mapping_block.py:7-45."""

    def __init__(self, z_dim, style_dim, n_mapping):
        super().__init__()
        self.dense_layers = nn.ModuleList(
            [Dense(z_dim if i == 0 else style_dim, style_dim, 1.0, 0.01) for i in range(n_mapping)])
        self.bias_act_layers = nn.ModuleList([BiasAct(style_dim, 0.01, "lrelu") for _ in range(n_mapping)])

    def forward(self, z):
        x = z * torch.rsqrt(z.square().mean(dim=1, keepdim=True) + 1e-8)
        for dense, ba in zip(self.dense_layers, self.bias_act_layers):
            if x.is_cuda:  # one GEMM + one activation launch per layer
                x = ops.dense_bias_act(x, dense.w, ba.b, _coef(dense.w.shape, dense.gain, dense.lrmul), ba.lrmul,
                                       lrelu=True)
            else:
                x = ba(dense(x))
        return x


class LatentEncoder(nn.Module):
    """models/custom_stylegan2/latent_encoder.py:9-99."""

    def __init__(self, cfg: Config, n_broadcast: int):
        super().__init__()
        self.n_broadcast = n_broadcast
        self.w_ema_decay, self.style_mixing_prob = 0.995, 0.9
        self.g_mapping = Mapping(cfg.z_dim, cfg.style_dim, cfg.n_mapping)
        self.register_buffer("w_avg", torch.zeros(cfg.style_dim))

    def forward(self, z, training: bool, truncation_psi=1.0, rand: Optional[dict] = None):
        ns = self.n_broadcast
        if training:
            if rand is not None and "z2" in rand:
                z2, cutoff = rand["z2"], int(rand["mix_cutoff"])
            else:  # :47-60 -- drawn on the device, no host sync (the step is HIP-graph capturable)
                z2 = torch.randn_like(z)
                cutoff = torch.where(torch.rand((), device=z.device) < self.style_mixing_prob,
                                     torch.randint(1, ns, (), device=z.device), torch.full((), ns, device=z.device))
            # the mapping network is row-wise: z and the style-mixing z2 (:47-60) go through it as one 2B-row batch
            w, w2 = self.g_mapping(torch.cat([z, z2], dim=0)).chunk(2, dim=0)
            wb = w[:, None, :].expand(-1, ns, -1)
            with torch.no_grad():  # :39-45
                batch_avg = w.mean(dim=0)
                self.w_avg.copy_(batch_avg + (self.w_avg - batch_avg) * self.w_ema_decay)
            idx = torch.arange(ns, device=z.device)[None, :, None]
            wb = torch.where(idx < cutoff, wb, w2[:, None, :].expand(-1, ns, -1))
        else:  # :73-78
            wb = self.g_mapping(z)[:, None, :].expand(-1, ns, -1)
            wb = self.w_avg + (wb - self.w_avg) * truncation_psi
        return wb


class WordEncoder(nn.Module):
    """models/word_encoder.py:8-63."""

    class _FC(nn.Module):
        def __init__(self, fin, fout):
            super().__init__()
            lim = math.sqrt(6.0 / (fin + fout))  # Keras glorot_uniform
            self.kernel = nn.Parameter(torch.empty(fin, fout).uniform_(-lim, lim))
            self.bias = nn.Parameter(torch.zeros(fout))

    def __init__(self, cfg: Config, dropout_rate=0.3):
        super().__init__()
        self.cfg, self.dropout_rate = cfg, dropout_rate
        self.w_embedding = nn.Parameter(torch.randn(cfg.main_vocab, cfg.embedding_out_dim))
        self.register_buffer("w0_embedding", torch.zeros(1, cfg.embedding_out_dim))
        self.fc = WordEncoder._FC(cfg.embedding_out_dim, cfg.word_encoder_dense_dim)

    def forward(self, words, batch_size=None, training=False, dropout_mask=None):
        cfg = self.cfg
        B = words.shape[0] if batch_size is None else batch_size
        table = torch.cat([self.w0_embedding, self.w_embedding], dim=0)
        emb = table[words.long()]
        if training:
            if dropout_mask is None:
                keep = 1.0 - self.dropout_rate
                dropout_mask = torch.bernoulli(torch.full_like(emb, keep)) / keep
            emb = emb * dropout_mask
        x = emb.reshape(B * cfg.max_char_number, cfg.embedding_out_dim)
        x = F.relu(x @ self.fc.kernel + self.fc.bias)
        h0, w0 = cfg.generator_resolutions[0]
        return x.reshape(B, w0, cfg.generator_feat_maps[0], h0).permute(0, 2, 3, 1).contiguous()


class ModulatedConv2D(nn.Module):
    """layers/modulated_conv2d.py:16-122 (parameters + style/demod coefficients)."""

    def __init__(self, cfg: Config, in_fmaps, out_fmaps, k, up, demodulate):
        super().__init__()
        self.up, self.demodulate, self.k = up, demodulate, k
        self.w = nn.Parameter(torch.randn(k, k, in_fmaps, out_fmaps))
        self.mod_dense = Dense(cfg.style_dim, in_fmaps, 1.0, 1.0)
        self.mod_bias = BiasAct(in_fmaps, 1.0, "linear")

    def style(self, y, mode="composable"):
        """s = mod_dense(y) + b + 1   (:74-76)."""
        if mode == "fused":
            return ops.dense_bias_act(y, self.mod_dense.w, self.mod_bias.b, _coef(self.mod_dense.w.shape), 1.0,
                                      lrelu=False, offset=1.0)
        return torch.addmm(self.mod_bias.b + 1.0, y, self.mod_dense.w * _coef(self.mod_dense.w.shape))

    def demod(self, s, mode):
        if not self.demodulate:
            return None
        if mode == "fused":
            return ops.demod_coefs(s, self.w)
        wsq = self.w.square().sum(dim=(0, 1)) * (_coef(self.w.shape) ** 2)  # (coef w)^2 summed over the taps
        return torch.rsqrt(s.square() @ wsq + 1e-8)

    def conv_composable(self, x, s, d):
        """any-order path: x*s -> conv / up-conv+FIR -> *d   (the reference's CPU branch :94-96,:119-121)."""
        coef = _coef(self.w.shape)  # rides along as the primitives' alpha: no elementwise pass over the filter
        xs = ops.scale_ch(x, s) if x.is_cuda else x * s[:, :, None, None]
        if self.up:
            y = ops.conv_transpose2d_s2(xs, torch.flip(self.w, (0, 1)), alpha=coef)
            y = ops.upfirdn2d(y, ops.fir_kernel(x.device, 4.0), pad=(1, 1, 1, 1))
        else:
            y = ops.conv2d(xs, self.w, (1, 1), (self.k // 2, self.k // 2), alpha=coef)
        if d is not None:
            y = ops.scale_ch(y, d) if y.is_cuda else y * d[:, :, None, None]
        return y


class ToRGB(nn.Module):
    """layers/to_rgb.py:7-33."""

    def __init__(self, cfg, in_ch):
        super().__init__()
        self.conv = ModulatedConv2D(cfg, in_ch, 3, 1, up=False, demodulate=False)
        self.apply_bias = BiasAct(3, 1.0, "linear")

    def forward(self, x, style, skip=None, mode="fused", s=None, colmask=None, mask_cw=0):
        """s: the precomputed style affine of this layer (fused mode: ops.style_affines does all layers in one launch).
        colmask [B, W // mask_cw]: mask_text_box (utils/utils.py:11-45) applied by the same launch (last block only)."""
        s = self.conv.style(style, mode) if s is None else s
        if mode == "fused":
            return ops.torgb_fused(x, self.conv.w, s, self.apply_bias.b, skip, colmask, mask_cw)
        if mode == "fused2" and colmask is None:  # twice-differentiable fused node (path-length pass)
            return ops2.torgb2(x, self.conv.w, s, self.apply_bias.b, skip)
        y = self.apply_bias(self.conv.conv_composable(x, s, None))
        y = y if skip is None else skip + y
        if colmask is not None:
            y = y * colmask.repeat_interleave(mask_cw, dim=1)[:, None, None, :]
        return y


class SynthesisBlock(nn.Module):
    """layers/synthesis_block.py:14-74."""

    def __init__(self, cfg, in_ch, fmaps):
        super().__init__()
        self.conv_0 = ModulatedConv2D(cfg, in_ch, fmaps, 3, up=True, demodulate=True)
        self.apply_noise_0 = Noise()
        self.apply_bias_act_0 = BiasAct(fmaps, 1.0, "lrelu")
        self.conv_1 = ModulatedConv2D(cfg, fmaps, fmaps, 3, up=False, demodulate=True)
        self.apply_noise_1 = Noise()
        self.apply_bias_act_1 = BiasAct(fmaps, 1.0, "lrelu")

    def forward(self, x, w0, w1, noise0, noise1, mode="fused", s0=None, s1=None, d0=None, d1=None, next_sink=None):
        """next_sink (fused mode): ops.UnitSink of the layer that consumes this block's output (the next block's up-convolution):
        conv_1's epilogue then writes that layer's unit tensor; conv_0's FIR epilogue always writes conv_1's."""
        for conv, nz, ba, style, noise, s, d in ((self.conv_0, self.apply_noise_0, self.apply_bias_act_0, w0, noise0, s0, d0),
                                                 (self.conv_1, self.apply_noise_1, self.apply_bias_act_1, w1, noise1, s1, d1)):
            s = conv.style(style, mode) if s is None else s
            if mode == "fused":
                fn = ops.modconv_up_fused if conv.up else ops.modconv_fused
                # the product the NEXT convolution contracts is (this output) * (its style): written by this layer's epilogue
                sink = (ops.UnitSink(s1, "s1", self.conv_1.w.shape[3]) if s1 is not None else None) if conv.up else next_sink
                x = fn(x, conv.w, s, noise, nz.noise_strength, ba.b, sink=sink)
            elif mode == "fused2":  # twice-differentiable fused layer: two autograd nodes with hand-written gradients
                x = ops2.mod_layer2(x, conv.w, s, conv.demod(s, mode) if d is None else d, noise, nz.noise_strength, ba.b,
                                    up=conv.up)
            else:  # any-order path: one launch for noise + bias + lrelu (ops.bias_act_c), gradients again primitives
                x = ops.bias_act_c(conv.conv_composable(x, s, conv.demod(s, mode)), noise, nz.noise_strength, ba.b)
        return x


class Synthesis(nn.Module):
    """layers/synthesis_block.py:90-156."""

    def __init__(self, cfg: Config):
        super().__init__()
        fm = cfg.generator_feat_maps
        self.resolutions = cfg.generator_resolutions
        self.initial_torgb = ToRGB(cfg, fm[0])
        self.synth_blocks = nn.ModuleList()
        self.torgbs = nn.ModuleList()
        prev = fm[0]
        for f in fm[1:]:
            self.synth_blocks.append(SynthesisBlock(cfg, prev, f))
            self.torgbs.append(ToRGB(cfg, f))
            prev = f

    def noise_shapes(self, B):
        return [(B, 1, h, w) for (h, w) in self.resolutions[1:] for _ in range(2)]

    def forward(self, x, style, noises: Optional[List[torch.Tensor]] = None, mode="fused", colmask=None, mask_cw=0):
        """colmask / mask_cw: the text-box mask of the final image, applied inside the last toRGB launch."""
        B = x.shape[0]
        if noises is None:  # fresh noise on every call, also at inference (noise.py:16-19): ONE generator launch for all ten maps
            shapes = self.noise_shapes(B)
            sizes = [b_ * c_ * h_ * w_ for (b_, c_, h_, w_) in shapes]
            flat = torch.randn(sum(-(-n // 4) * 4 for n in sizes), device=x.device)  # (every map starts 16-byte aligned)
            noises, o = [], 0
            for shp, n in zip(shapes, sizes):
                noises.append(flat[o:o + n].view(shp))
                o += -(-n // 4) * 4
        k_up = ops.fir_kernel(x.device, 4.0)
        ws = style.unbind(dim=1)  # one node: its backward is ONE stack of the per-layer latent gradients (16 selects
        # would each zero-fill a [B, n, 512] tensor and be summed pairwise by the autograd engine)
        nb = len(self.synth_blocks)
        s_tr, s_c0, s_c1 = [None] * (nb + 1), [None] * nb, [None] * nb
        if mode == "fused":  # all style affines in one launch (and one for their backward)
            convs = [self.initial_torgb.conv] + [c for b, t in zip(self.synth_blocks, self.torgbs)
                                                  for c in (b.conv_0, b.conv_1, t.conv)]
            rows = [0] + [r for i in range(nb) for r in (3 * i, 3 * i + 1, 3 * i + 2)]  # the latent row each layer reads
            ss = ops.style_affines(style, [c.mod_dense.w for c in convs], [c.mod_bias.b for c in convs],
                                   _coef(convs[0].mod_dense.w.shape), rows)
            s_tr[0] = ss[0]
            for i in range(nb):
                s_c0[i], s_c1[i], s_tr[i + 1] = ss[1 + 3 * i], ss[2 + 3 * i], ss[3 + 3 * i]
        d_c0, d_c1 = [None] * nb, [None] * nb
        if mode == "fused2":  # twice-differentiable pass: all affines / demodulations batched
            assert x.is_cuda, 'mode="fused2" runs on the HIP kernels (CPU tensors: mode="composable")'
            convs = [self.initial_torgb.conv] + [c for b, t in zip(self.synth_blocks, self.torgbs)
                                                  for c in (b.conv_0, b.conv_1, t.conv)]
            rows = [0] + [r for i in range(nb) for r in (3 * i, 3 * i + 1, 3 * i + 2)]
            dl = [1 + 3 * i + j for i in range(nb) for j in (0, 1)]
            ss, dd = ops2.synthesis_styles(style, rows, [c.mod_dense.w for c in convs], [c.mod_bias.b for c in convs],
                                           _coef(convs[0].mod_dense.w.shape), dl, [convs[l].w for l in dl])
            s_tr[0] = ss[0]
            for i in range(nb):
                s_c0[i], s_c1[i], s_tr[i + 1] = ss[1 + 3 * i], ss[2 + 3 * i], ss[3 + 3 * i]
                d_c0[i], d_c1[i] = dd[1 + 3 * i], dd[2 + 3 * i]
        y = self.initial_torgb(x, ws[0], None, mode, s=s_tr[0])
        for i, (block, torgb) in enumerate(zip(self.synth_blocks, self.torgbs)):
            nxt = None
            if mode == "fused" and i + 1 < nb:  # block i + 1 starts with the up-convolution conv_0 (style s_c0[i + 1])
                nxt = ops.UnitSink(s_c0[i + 1], "up", self.synth_blocks[i + 1].conv_0.w.shape[3])
            x = block(x, ws[3 * i], ws[3 * i + 1], noises[2 * i], noises[2 * i + 1], mode, s0=s_c0[i], s1=s_c1[i],
                      d0=d_c0[i], d1=d_c1[i], next_sink=nxt)
            y = ops.upfirdn2d(y, k_up, up=(2, 2), pad=(2, 1, 2, 1))  # upsample_2d, :152
            last = i == nb - 1
            y = torgb(x, ws[3 * i + 2], y, mode, s=s_tr[i + 1], colmask=colmask if last else None, mask_cw=mask_cw if last else 0)
        return y


class Generator(nn.Module):
    """models/custom_stylegan2/generator.py:10-59."""

    def __init__(self, cfg: Config):
        super().__init__()
        self.cfg = cfg
        self.word_encoder = WordEncoder(cfg)
        self.synthesis = Synthesis(cfg)
        self.n_style = 2 * len(self.synthesis.synth_blocks) + len(self.synthesis.torgbs)
        self.latent_encoder = LatentEncoder(cfg, self.n_style)

    def forward(self, inputs, batch_size=None, ret_style=False, truncation_psi=1.0, training=False,
                rand: Optional[dict] = None, noises_key="noises", mode="fused", mask_words=None):
        """mask_words [B, max_char_number]: return mask_text_box(image, mask_words, char_width) -- the mask
        (utils/utils.py:11-45) is then the epilogue of the last toRGB launch instead of a pass over the image."""
        words, z = inputs
        rand = rand or {}
        colmask = None if mask_words is None else (mask_words != 0).to(torch.float32).contiguous()
        we = self.word_encoder(words, batch_size, training=training, dropout_mask=rand.get("dropout_mask"))
        style = self.latent_encoder(z, training=training, truncation_psi=truncation_psi, rand=rand)
        if ret_style:
            style = style.clone()
        img = self.synthesis(we, style, rand.get(noises_key), mode, colmask=colmask,
                             mask_cw=self.cfg.char_width if colmask is not None else 0)
        return (img, style) if ret_style else img

    @torch.no_grad()
    def set_as_moving_average_of(self, src: "Generator", beta=0.99):
        """generator.py:48-59: every weight lerps with beta, w_avg is copied."""
        from .optim import ema_update
        ema_update(self, src, beta)


# ----------------------------------------------------------------------------------------
class Conv2D(nn.Module):
    """layers/conv.py:11-73 (parameter holder)."""

    def __init__(self, in_fmaps, out_fmaps, k):
        super().__init__()
        self.w = nn.Parameter(torch.randn(k, k, in_fmaps, out_fmaps))


class FromRGB(nn.Module):
    """layers/from_rgb.py:7-29."""

    def __init__(self, fmaps):
        super().__init__()
        self.conv = Conv2D(3, fmaps, 1)
        self.apply_bias_act = BiasAct(fmaps, 1.0, "lrelu")

    def forward(self, x, mode="fused", next_sink=None):
        if mode == "fused":
            return ops.conv_bias_act_fused(x, self.conv.w, self.apply_bias_act.b, role="d_image", sink=next_sink)
        return ops.bias_act_c(ops.conv2d(x, self.conv.w, alpha=_coef(self.conv.w.shape)), None, None, self.apply_bias_act.b)


class DiscriminatorBlock(nn.Module):
    """models/custom_stylegan2/discriminator.py:11-84."""

    def __init__(self, n_f0, n_f1, reduce_height):
        super().__init__()
        self.reduce_height = reduce_height
        self.conv_0 = Conv2D(n_f0, n_f0, 3)
        self.apply_bias_act_0 = BiasAct(n_f0, 1.0, "lrelu")
        self.conv_1 = Conv2D(n_f0, n_f1, 3)
        self.apply_bias_act_1 = BiasAct(n_f1, 1.0, "lrelu")
        self.conv_skip = Conv2D(n_f0, n_f1, 1)

    def forward(self, x, mode="fused", next_sink=None):
        """next_sink (fused mode): ops.UnitSink of the next block's conv_0 -- the residual sum's launch writes its unit tensor."""
        sh = 2 if self.reduce_height else 1
        k = ops.fir_kernel(x.device, 1.0)
        rs = 1.0 / math.sqrt(2.0)
        # skip: blur + strided 1x1 conv == (blur evaluated only at the strided sites) + 1x1 conv
        role = "d" if mode == "fused" else None  # (ops.FLAGS: which backward work a pass may skip)
        if mode == "fused" and ops.TUNING.fuse_skip_grad:  # conv_0 and the skip FIR as one node: d(x) without an add pass
            t, xd = ops.conv_bias_act_skip_fused(x, self.conv_0.w, self.apply_bias_act_0.b, k, (2, sh), (1, 2, 1, 2), role="d")
        else:
            xd = ops.upfirdn2d(x, k, down=(2, sh), pad=(1, 2, 1, 2), role=role)
        if mode == "fused":
            if not ops.TUNING.fuse_skip_grad:
                t = ops.conv_bias_act_fused(x, self.conv_0.w, self.apply_bias_act_0.b, pad=(1, 1), role="d")
            fold = ops.TUNING.fold_res_scale and ops.compute_mode() != "bf16"
            if sh == 2 and ops.blur_conv_s2_units(t.shape[0], t.shape[1], self.conv_1.w.shape[3], t.shape[2], t.shape[3]):
                # blur + strided convolution with the blurred tensor as a phase unit tensor only (ops._BlurConvS2Fused)
                u = ops.blur_conv_s2_fused(t, self.conv_1.w, self.apply_bias_act_1.b, role="d", out_mul=rs if fold else 1.0)
                return ops.conv_bias_act_fused(xd, self.conv_skip.w, None, act=ACT_LINEAR, residual=u, res_scale=1.0 if fold else rs,
                                               role="d", out_mul=rs if fold else 1.0, sink=next_sink)
            tb = ops.upfirdn2d(t, k, pad=(2, 3, 2, 3), role="d")  # conv_downsample_2d, upfirdn_2d_v2.py:106-113
            if fold:
                # (u + skip) / sqrt(2) with the factor folded into u's lrelu gain and the skip conv's scale: the sum and its
                # gradient need no scaling pass (13 elementwise passes over block-sized tensors per step otherwise).
                # fp32-grade arithmetics only: the full-width step's G-gradient error against the oracle is unchanged there
                # (5.05e-5 -> 5.08e-5 in f32x3) but went from 3.8e-2 to 9.6e-2 in bf16 mode (bar 8e-2) -- the same sums in a
                # different order land on other bf16 rounding boundaries of the operands, and that metric is that sensitive
                u = ops.conv_bias_act_fused(tb, self.conv_1.w, self.apply_bias_act_1.b, stride=(sh, 2), role="d", out_mul=rs)
                return ops.conv_bias_act_fused(xd, self.conv_skip.w, None, act=ACT_LINEAR, residual=u, res_scale=1.0, role="d",
                                               out_mul=rs, sink=next_sink)
            u = ops.conv_bias_act_fused(tb, self.conv_1.w, self.apply_bias_act_1.b, stride=(sh, 2), role="d")
            return ops.conv_bias_act_fused(xd, self.conv_skip.w, None, act=ACT_LINEAR, residual=u, res_scale=rs, role="d",
                                           sink=next_sink)
        t = ops.bias_act_c(ops.conv2d(x, self.conv_0.w, (1, 1), (1, 1), alpha=_coef(self.conv_0.w.shape)), None, None,
                           self.apply_bias_act_0.b)
        tb = ops.upfirdn2d(t, k, pad=(2, 3, 2, 3))
        u = ops.bias_act_c(ops.conv2d(tb, self.conv_1.w, (sh, 2), alpha=_coef(self.conv_1.w.shape)), None, None,
                           self.apply_bias_act_1.b)
        skip = ops.conv2d(xd, self.conv_skip.w, alpha=_coef(self.conv_skip.w.shape))
        return (u + skip) * rs


def minibatch_std(x, group_size=4):
    """layers/mini_batch_std.py:10-35 (statistics groups live inside the per-GPU batch)."""
    B, C, H, W = x.shape
    G = min(group_size, B)
    y = x.reshape(G, -1, 1, C, H, W)
    y = y - y.mean(dim=0, keepdim=True)
    y = (y.square().mean(dim=0) + 1e-8).sqrt()
    y = y.mean(dim=(2, 3, 4), keepdim=True).mean(dim=2)
    return torch.cat([x, y.repeat(G, 1, H, W)], dim=1)


class DiscriminatorLastBlock(nn.Module):
    """discriminator.py:101-142."""

    def __init__(self, n_f0, n_f1, hw):
        super().__init__()
        self.conv_0 = Conv2D(n_f0 + 1, n_f0, 3)
        self.apply_bias_act_0 = BiasAct(n_f0, 1.0, "lrelu")
        self.dense_1 = Dense(n_f0 * hw[0] * hw[1], n_f1)
        self.apply_bias_act_1 = BiasAct(n_f1, 1.0, "lrelu")

    def forward(self, x, mode="fused", parts=1):
        if mode == "fused":  # one launch each: statistics layer, conv + bias + lrelu, dense + bias + lrelu
            x = ops.minibatch_std_fused(x, 4, parts, role="d")
            x = ops.conv_bias_act_fused(x, self.conv_0.w, self.apply_bias_act_0.b, pad=(1, 1), role="d")
            return ops.dense_bias_act(x.reshape(x.shape[0], -1), self.dense_1.w, self.apply_bias_act_1.b,
                                      _coef(self.dense_1.w.shape, self.dense_1.gain, self.dense_1.lrmul),
                                      self.apply_bias_act_1.lrmul, lrelu=True)
        assert parts == 1
        x = minibatch_std(x, 4).contiguous()
        x = ops.bias_act_c(ops.conv2d(x, self.conv_0.w, (1, 1), (1, 1), alpha=_coef(self.conv_0.w.shape)), None, None,
                           self.apply_bias_act_0.b)
        return self.apply_bias_act_1(self.dense_1(x))


class Discriminator(nn.Module):
    """models/custom_stylegan2/discriminator.py:157-214."""

    def __init__(self, cfg: Config):
        super().__init__()
        res, fm = cfg.discrim_resolutions, cfg.discrim_feat_maps
        self.initial_fromrgb = FromRGB(fm[0])
        self.blocks = nn.ModuleList([
            DiscriminatorBlock(f0, f1, r[0] != rn[0]) for r, rn, f0, f1 in zip(res[:-1], res[1:], fm[:-1], fm[1:])])
        self.last_block = DiscriminatorLastBlock(fm[-2], fm[-1], res[-1])
        self.last_dense = Dense(fm[-1], 1)
        self.last_bias = BiasAct(1, 1.0, "linear")

    def forward(self, images, mode="fused", cuts=None, parts=1):
        """parts: the batch is ``parts`` independent batches laid end to end (the d-step's [fake; real]): the minibatch
        statistics stay inside each part, every other layer is per-sample (fused mode only).
        cuts: block indices k -- the activation ENTERING blocks[k] is returned too (``(scores, [h_k...])``), so that a
        caller can run the backward pass in stages (deep layers first) and exchange each stage's gradients while the next
        stage still computes (training_step.py: bucketed data-parallel all-reduce)."""
        # fused mode: every launch that produces a block's input also writes units(input) for that block's conv_0 (ops.UnitSink)
        sink_of = lambda blk: ops.UnitSink(None, "s1", blk.conv_0.w.shape[3]) if mode == "fused" else None
        x = self.initial_fromrgb(images.contiguous(), mode, next_sink=sink_of(self.blocks[0]))
        taps = []
        for i, block in enumerate(self.blocks):
            if cuts is not None and i in cuts:
                taps.append(x)
            x = block(x, mode, next_sink=sink_of(self.blocks[i + 1]) if i + 1 < len(self.blocks) else None)
        x = self.last_block(x, mode, parts)
        if mode == "fused":
            scores = ops.dense_bias_act(x, self.last_dense.w, self.last_bias.b,
                                        _coef(self.last_dense.w.shape, self.last_dense.gain, self.last_dense.lrmul),
                                        self.last_bias.lrmul, lrelu=False)
        else:
            scores = self.last_bias(self.last_dense(x))
        return scores if cuts is None else (scores, taps)


def mask_text_box(fake_images, input_words, char_width: int):
    """utils/utils.py:11-45."""
    mask = (input_words != 0).to(fake_images.dtype).repeat_interleave(char_width, dim=1)
    return fake_images * mask[:, None, None, :]


def generator_output_to_uint8(fake_images):
    """utils/utils.py:48-63."""
    x = (fake_images.clamp(-1.0, 1.0) + 1.0) * 127.5
    return x.permute(0, 2, 3, 1).to(torch.uint8)
