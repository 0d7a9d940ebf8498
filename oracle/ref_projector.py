"""TEST INFRASTRUCTURE -- CPU restatement of the reference's projector path (parity unpinned: TF / VGG / LPIPS weights
absent; pinned by hand-computable properties in tests/test_oracle_cpu.py and used as the checker of
textboxgan_amd/projector.py in tests/test_projector_gpu.py).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package.

Follows  projector/lpips_tensorflow.py  (image_preprocess :9-18, perceptual_model taps :129-150, the merged model :20-78,
linear_model :189-213)  and  projector/projector.py  (_get_lr :65-83, _compute_w_latent :85-103, the loop :146-168,
get_perceptual_loss :208-228, _projector_step :230-273; Keras Adam defaults of :62)."""
from __future__ import annotations

import math
from typing import Callable, Dict, List

import torch
import torch.nn.functional as F

from . import ref_model as M

VGG16_LAYERS = [(3, 64), (64, 64), "M", (64, 128), (128, 128), "M", (128, 256), (256, 256), (256, 256), "M",
                (256, 512), (512, 512), (512, 512), "M", (512, 512), (512, 512), (512, 512)]
TAPS = (1, 3, 6, 9, 12)


def image_preprocess(image_nhwc):
    """lpips_tensorflow.py:9-18."""
    factor, center = 255.0 / 2.0, 1.0
    scale = torch.tensor([0.458, 0.448, 0.450], dtype=image_nhwc.dtype)
    shift = torch.tensor([-0.030, -0.088, -0.188], dtype=image_nhwc.dtype)
    image = image_nhwc / factor - center
    return (image - shift) / scale


def vgg_features(P: Dict[str, torch.Tensor], image_nhwc) -> List[torch.Tensor]:
    """tf.keras.applications VGG16 conv stack (SAME 3x3 + bias + ReLU, 2x2/2 max-pools), outputs of block{1,2}_conv2 and
    block{3,4,5}_conv3 (lpips_tensorflow.py:129-150).  P: ``convs.i.kernel`` HWIO, ``convs.i.bias``."""
    x = image_preprocess(image_nhwc).permute(0, 3, 1, 2)
    feats, ci = [], 0
    for layer in VGG16_LAYERS:
        if layer == "M":
            x = F.max_pool2d(x, 2, 2)
        else:
            x = F.relu(F.conv2d(x, P[f"convs.{ci}.kernel"].permute(3, 2, 0, 1), P[f"convs.{ci}.bias"], padding=1))
            if ci in TAPS:
                feats.append(x)
            ci += 1
    return feats


def lpips(P, image1_nhwc, image2_nhwc):
    """lpips_tensorflow.py:20-78: unit-normalise over channels, squared difference, 1x1 lin (no bias; Dropout is the
    identity at inference), spatial mean, sum over the five taps, squeeze."""
    total = 0.0
    for i, (a, b) in enumerate(zip(vgg_features(P, image1_nhwc), vgg_features(P, image2_nhwc))):
        a = a * torch.rsqrt(a.square().sum(dim=1, keepdim=True))
        b = b * torch.rsqrt(b.square().sum(dim=1, keepdim=True))
        d = (a - b).square()
        lin = F.conv2d(d, P[f"lins.{i}.kernel"].permute(3, 2, 0, 1))
        total = total + lin.mean(dim=(2, 3), keepdim=True)
    return total.squeeze()


def get_lr(t: float, lr=0.1, rampup=0.05, rampdown=0.25) -> float:
    """projector.py:65-83."""
    ramp = min(1, (1 - t) / rampdown)
    ramp = 0.5 - 0.5 * math.cos(ramp * math.pi)
    ramp = ramp * min(1, t / rampup)
    return lr * ramp


def compute_w_latent(G, cfg, z_latent):
    """projector.py:85-103."""
    w = M.latent_encoder(G, cfg, z_latent, training=False, rand=None)[:, 1, :]
    mean = w.mean(dim=0, keepdim=True)
    std = ((w - mean).square().sum() / z_latent.shape[0]) ** 0.5
    return std, mean.clone()


def project(G, LP, cfg, text: str, target_image_nhwc, ocr_serve: Callable, rand: dict, num_steps: int, total_steps=1000,
            return_grads=False):
    """projector.py:122-182 + :230-273 for ``num_steps`` steps with injected randomness
    (rand: z_latent [n,512], w_noise [steps][1,512] unit normals, noises [steps][10 maps])."""
    words = torch.from_numpy(M.string_to_main_int_sequence([text], cfg.max_char_number))
    ocr_label = torch.from_numpy(M.string_to_aster_int_sequence([text], cfg.max_char_number))
    image_width = cfg.char_width * len(text)
    w_std, w_var = compute_w_latent(G, cfg, rand["z_latent"])
    word_encoded = M.word_encoder(G, cfg, words, None)
    ns = M.n_style(cfg)
    b1, b2, eps = 0.9, 0.999, 1e-7  # tf.keras.optimizers.Adam() defaults
    m, v = torch.zeros_like(w_var), torch.zeros_like(w_var)
    losses, grads = [], []
    for step in range(1, num_steps + 1):
        t = step / total_steps
        lr = get_lr(t)
        strength = w_std * 0.05 * max(0, 1 - t / 0.75) ** 2
        w = w_var.clone().requires_grad_(True)
        w_final = (w + rand["w_noise"][step - 1] * strength).unsqueeze(0).repeat(1, ns, 1)
        img = M.synthesis(G, cfg, word_encoded, w_final, rand["noises"][step - 1])
        inp = M.ocr_convert_inputs(img, words, cfg, blank_label=0)
        ocr_loss = M.softmax_cross_entropy_loss(M.ocr_call(inp, ocr_serve, cfg.max_char_number), ocr_label, cfg.batch_size)
        gen = img[:, :, :, :image_width].permute(0, 2, 3, 1)
        gen = (gen.clamp(-1.0, 1.0) + 1.0) * 127.5
        loss = lpips(LP, target_image_nhwc, gen) + 0.1 * ocr_loss
        (g,) = torch.autograd.grad(loss, [w])
        grads.append(g.detach().clone())
        m = b1 * m + (1 - b1) * g
        v = b2 * v + (1 - b2) * g * g
        lr_t = lr * math.sqrt(1 - b2 ** step) / (1 - b1 ** step)
        w_var = w_var - lr_t * m / (v.sqrt() + eps)
        losses.append(float(loss))
    return (w_var, losses, grads) if return_grads else (w_var, losses)
