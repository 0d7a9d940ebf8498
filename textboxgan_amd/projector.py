"""Projector (BASELINE configs[4]): find the style vector that makes the generator reproduce a given
text box -- reference ``projector/projector.py`` (loop :122-182, step :230-273, LR schedule :65-83, latent statistics
:85-103) with the LPIPS perceptual metric of ``projector/lpips_tensorflow.py`` (preprocess :9-18, VGG16 taps :129-150,
unit-normalise / squared difference / 1x1 lin / spatial mean / sum :20-78, lin model :189-213).

MI355X path: the 13 VGG convolutions (+bias +ReLU) are ONE launch each of the fp32-MFMA implicit-GEMM kernel
(``ops.frozen_conv``: constant filters packed once, data gradient only -- only the latent is optimised), the generator runs
its fused HIP layers with the filter gradients skipped; 2x2 max-pool, channel unit-normalisation and the 1x1 "lin" layers
are small torch ops.  There is no CPU fallback inside ``LPIPS`` for device tensors.

PARITY UNPINNED for the network weights: the VGG16-imagenet and LPIPS "lin" checkpoints (``projector/perceptual_weights``)
are external downloads absent from the reference tree; ``LPIPS`` uses deterministic synthetic weights and
``load_weights_tf`` imports the real ones (Keras TF-format checkpoints = TensorBundle files, read without TensorFlow).
"""
from __future__ import annotations

import math
import os
from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .char_tokens import string_to_aster_int_sequence, string_to_main_int_sequence
from .config import Config, cfg as default_cfg
from .training_step import softmax_cross_entropy_loss

# VGG16 feature stack (tf.keras.applications.VGG16, include_top=False): (in, out) = 3x3 SAME conv + bias + ReLU, "M" = 2x2 max-pool
VGG16_LAYERS = [(3, 64), (64, 64), "M", (64, 128), (128, 128), "M", (128, 256), (256, 256), (256, 256), "M",
                (256, 512), (512, 512), (512, 512), "M", (512, 512), (512, 512), (512, 512)]
LPIPS_TAPS = (1, 3, 6, 9, 12)  # conv indices of block1_conv2, block2_conv2, block3_conv3, block4_conv3, block5_conv3
LPIPS_CHANNELS = (64, 128, 256, 512, 512)


def image_preprocess(image_nhwc: torch.Tensor) -> torch.Tensor:
    """lpips_tensorflow.py:9-18: [0,255] -> [-1,1] -> (x - shift) / scale."""
    scale = image_nhwc.new_tensor([0.458, 0.448, 0.450])
    shift = image_nhwc.new_tensor([-0.030, -0.088, -0.188])
    return ((image_nhwc / (255.0 / 2.0) - 1.0) - shift) / scale


class LPIPS(nn.Module):
    """learned_perceptual_metric_model (lpips_tensorflow.py:20-78).  Parameters keep the Keras layouts: conv kernels HWIO
    ``convs.i.kernel [3,3,I,O]`` + ``convs.i.bias [O]``, lin kernels ``lins.i.kernel [1,1,C,1]``."""

    class _Conv(nn.Module):
        def __init__(self, cin, cout, k, bias, g):
            super().__init__()
            self.kernel = nn.Parameter(torch.randn(k, k, cin, cout, generator=g) * math.sqrt(2.0 / (k * k * cin)))
            self.bias = nn.Parameter(torch.zeros(cout)) if bias else None

    def __init__(self, seed: int = 2018):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.convs = nn.ModuleList([LPIPS._Conv(c[0], c[1], 3, True, g) for c in VGG16_LAYERS if c != "M"])
        self.lins = nn.ModuleList([LPIPS._Conv(c, 1, 1, False, g) for c in LPIPS_CHANNELS])
        with torch.no_grad():
            for i, conv in enumerate(self.convs):
                conv.bias.copy_(torch.randn(conv.bias.shape, generator=g) * 0.05)
            for lin in self.lins:  # the learned LPIPS weights are non-negative
                lin.kernel.copy_(lin.kernel.abs() * 0.1)
        for p in self.parameters():
            p.requires_grad_(False)
        self._packs = {}

    def _apply(self, fn, *a, **kw):
        self._packs = {}
        return super()._apply(fn, *a, **kw)

    def load_state_dict(self, *a, **kw):
        r = super().load_state_dict(*a, **kw)
        self._packs = {}
        return r

    def load_weights_tf(self, vgg_prefix: str, lin_prefix: str) -> None:
        """``net.load_weights(vgg_ckpt)`` / ``lin.load_weights(lin_ckpt)`` (lpips_tensorflow.py:26-27): Keras TF-format
        weight checkpoints.  Variables are taken in layer order (``layer_with_weights-N/{kernel,bias}``)."""
        from . import tf_checkpoint as T

        def ordered(prefix):
            data = T.read_bundle(prefix)
            items = {}
            for k, v in data.items():
                if k.startswith("layer_with_weights-") and k.endswith(T.VAR_SUFFIX):
                    idx, name = k[len("layer_with_weights-"):-len(T.VAR_SUFFIX)].split("/", 1)
                    items.setdefault(int(idx), {})[name] = v
            return [items[i] for i in sorted(items)]
        vgg, lin = ordered(vgg_prefix), ordered(lin_prefix)
        assert len(vgg) == len(self.convs) and len(lin) == len(self.lins), "unexpected number of layers in the checkpoint"
        with torch.no_grad():
            for conv, v in zip(self.convs, vgg):
                conv.kernel.copy_(torch.from_numpy(v["kernel"])); conv.bias.copy_(torch.from_numpy(v["bias"]))
            for l, v in zip(self.lins, lin):
                l.kernel.copy_(torch.from_numpy(v["kernel"]))
        self._packs = {}

    def _conv(self, x, i):
        conv = self.convs[i]
        if not x.is_cuda:  # no CPU path in the product (the CPU definition lives in oracle/ref_projector.py)
            raise RuntimeError("LPIPS runs its convolutions on the HIP kernels: device tensors only")
        from . import ops
        key = (i, ops.compute_mode())
        if key not in self._packs:
            self._packs[key] = ops.frozen_conv_packs(conv.kernel, (1, 1))
        return ops.frozen_conv(x, conv.kernel, conv.bias, (1, 1), (1, 1), True, None, self._packs[key])

    def features(self, image_nhwc: torch.Tensor) -> List[torch.Tensor]:
        x = image_preprocess(image_nhwc).permute(0, 3, 1, 2).contiguous()
        feats, ci = [], 0
        for layer in VGG16_LAYERS:
            if layer == "M":
                x = F.max_pool2d(x, 2, 2)
            else:
                x = self._conv(x, ci)
                if ci in LPIPS_TAPS:
                    feats.append(x)
                ci += 1
        return feats

    def forward(self, image1_nhwc: torch.Tensor, image2_nhwc: torch.Tensor) -> torch.Tensor:
        """both images [B,H,W,3] in [0,255]; returns the metric summed over the 5 taps (squeezed, :74-75)."""
        total = None
        for f1, f2, lin in zip(self.features(image1_nhwc), self.features(image2_nhwc), self.lins):
            n1 = f1 * torch.rsqrt(f1.square().sum(dim=1, keepdim=True))   # :41-56 (no epsilon, as the reference)
            n2 = f2 * torch.rsqrt(f2.square().sum(dim=1, keepdim=True))
            d = (n1 - n2).square()                                         # :59-62
            v = (d * lin.kernel.reshape(1, -1, 1, 1)).sum(dim=1, keepdim=True).mean(dim=(2, 3), keepdim=True)  # :65-71
            total = v if total is None else total + v
        return total.squeeze()


class Projector:
    """Drop-in for the reference ``Projector`` (projector.py:32-273): same hyper-parameters and method names.  ``generator``
    is the EMA generator (g_clone), ``aster_ocr`` an ``AsterInferer``; both frozen."""

    def __init__(self, text_of_the_image: str, generator, aster_ocr, cfg: Config = default_cfg,
                 perceptual_loss: Optional[LPIPS] = None, device=None):
        self.cfg = cfg
        self.text_of_the_image = text_of_the_image
        self.image_width = cfg.char_width * len(text_of_the_image)
        self.char_height = cfg.char_height
        self.generator, self.aster_ocr = generator, aster_ocr
        self.device = device if device is not None else next(generator.parameters()).device
        self.perceptual_loss = (perceptual_loss if perceptual_loss is not None else LPIPS()).to(self.device)
        self.n_mean_latent = 10000
        self.num_steps = 1000
        self.save_and_log_frequency = 100
        self.lr_rampup, self.lr_rampdown, self.lr = 0.05, 0.25, 0.1
        self.noise_strength_level, self.noise_ramp = 0.05, 0.75
        self.ocr_loss_factor = 0.1
        # tf.keras.optimizers.Adam() defaults (projector.py:62); the learning rate is re-assigned every step (:149)
        self.beta_1, self.beta_2, self.epsilon = 0.9, 0.999, 1e-7
        self._m = self._v = None
        self._t = 0
        self.grad_log = None  # set to [] to record the latent gradient of every step (parity tests)

    def _get_lr(self, t: float) -> float:
        """projector.py:65-83."""
        lr_ramp = min(1.0, (1.0 - t) / self.lr_rampdown)
        lr_ramp = 0.5 - 0.5 * math.cos(lr_ramp * math.pi)
        lr_ramp = lr_ramp * min(1.0, t / self.lr_rampup)
        return self.lr * lr_ramp

    @torch.no_grad()
    def _compute_w_latent(self, z_latent: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """projector.py:85-103: mean style of n_mean_latent random latents and its scalar spread."""
        if z_latent is None:
            z_latent = torch.randn(self.n_mean_latent, self.cfg.z_dim, device=self.device)
        z_latent = z_latent.to(self.device)
        w_latent = self.generator.latent_encoder(z_latent, training=False)[:, 1, :]
        w_mean = w_latent.mean(dim=0, keepdim=True)
        w_std = ((w_latent - w_mean).square().sum() / z_latent.shape[0]).sqrt()
        return w_std, w_mean.clone()

    def _get_ocr_loss(self, ocr_label, generated_image, input_word):
        """projector.py:184-206 (blank label 0: the crop uses the MAIN-vocabulary word, whose padding is 0)."""
        inp = self.aster_ocr.convert_inputs(generated_image, input_word, blank_label=0)
        return softmax_cross_entropy_loss(self.aster_ocr(inp), ocr_label, self.cfg.batch_size)

    def get_perceptual_loss(self, generated_image, target_image):
        """projector.py:208-228."""
        gen = generated_image[:, :, :, : self.image_width].permute(0, 2, 3, 1)
        gen = (gen.clamp(-1.0, 1.0) + 1.0) * 127.5
        return self.perceptual_loss(target_image, gen)

    def _projector_step(self, w_latent_noise, w_latent_var, ocr_label, word_encoded, input_word, target_image, lr,
                        noises=None):
        """projector.py:230-273: loss = LPIPS + 0.1 * OCR-CE, gradient w.r.t. the latent only, one Keras-Adam update
        (m, v, bias-corrected step size, epsilon outside the sqrt) applied in place to ``w_latent_var``."""
        w = w_latent_var.detach().requires_grad_(True)
        w_final = (w + w_latent_noise).unsqueeze(0).expand(1, self.generator.n_style, -1).reshape(
            1, self.generator.n_style, -1)
        generated = self.generator.synthesis(word_encoded, w_final, noises)
        ocr_loss = self._get_ocr_loss(ocr_label, generated, input_word)
        p_loss = self.get_perceptual_loss(generated, target_image)
        loss = p_loss + self.ocr_loss_factor * ocr_loss
        (g,) = torch.autograd.grad(loss, [w])
        if self.grad_log is not None:  # parity tests: the latent gradient of every step (gradient parity x optimiser parity)
            self.grad_log.append(g.detach().clone())
        with torch.no_grad():
            if self._m is None:
                self._m, self._v = torch.zeros_like(g), torch.zeros_like(g)
            self._t += 1
            self._m.mul_(self.beta_1).add_(g, alpha=1.0 - self.beta_1)
            self._v.mul_(self.beta_2).addcmul_(g, g, value=1.0 - self.beta_2)
            lr_t = lr * math.sqrt(1.0 - self.beta_2 ** self._t) / (1.0 - self.beta_1 ** self._t)
            w_latent_var.sub_(lr_t * self._m / (self._v.sqrt() + self.epsilon))
        return loss.detach()

    def main(self, target_image: torch.Tensor, num_steps: Optional[int] = None, rand: Optional[dict] = None):
        """projector.py:122-182 without the file I/O (cv2.imread/resize and the PNG / latents.txt writing belong to the
        CLI): ``target_image`` is the already loaded and resized uint8-valued [1, H, image_width, 3] tensor.  Returns
        (w_latent_var, saved_latents, losses).  ``rand``: injected randomness for parity tests (z_latent, step noise)."""
        cfg, dev = self.cfg, self.device
        num_steps = self.num_steps if num_steps is None else num_steps
        target_image = target_image.to(dev, torch.float32)
        input_word = torch.from_numpy(string_to_main_int_sequence([self.text_of_the_image], cfg.max_char_number)).to(dev)
        ocr_label = torch.from_numpy(string_to_aster_int_sequence([self.text_of_the_image], cfg.max_char_number)).to(dev)
        rand = rand or {}
        w_std, w_var = self._compute_w_latent(rand.get("z_latent"))
        with torch.no_grad():
            word_encoded = self.generator.word_encoder(input_word, batch_size=1)
        saved, losses = [], []
        for step in range(1, num_steps + 1):
            t = step / self.num_steps
            lr = self._get_lr(t)
            noise_strength = w_std * self.noise_strength_level * max(0.0, 1.0 - t / self.noise_ramp) ** 2
            unit = rand["w_noise"][step - 1].to(dev) if "w_noise" in rand else torch.randn_like(w_var)
            noises = [n.to(dev) for n in rand["noises"][step - 1]] if "noises" in rand else None
            loss = self._projector_step(unit * noise_strength, w_var, ocr_label, word_encoded, input_word, target_image, lr,
                                        noises)
            losses.append(loss)
            if step % self.save_and_log_frequency == 0:
                saved.append(w_var.detach().clone())
        return w_var, saved, losses
