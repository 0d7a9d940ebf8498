"""ORACLE (test infrastructure, NOT product code) -- definition-level operators.

CPU restatement of the operators on the TextBoxGAN ``training_step`` hot path.  Two
flavours of every operator:

* ``np_*``  : float64 numpy, written from the maths with explicit index arithmetic
              (loops / einsum) -- slow, for small cases and for pinning the torch twin.
* ``t_*``   : torch (CPU, fp32 or fp64) built from stock ``torch.nn.functional`` ops so
              autograd supplies first- and second-order gradients.

PARITY UNPINNED: the reference is TensorFlow 2.8 and cannot be imported or run in this
container (no TF wheel, no network), it ships no tests / golden vectors, and its single
native file needs TF headers.  These functions are pinned instead against (i) each other,
(ii) ``scipy.signal.upfirdn``, (iii) the reference's two internal twins restated here
(``.cu`` index maths vs ``upfirdn_2d_ref``; fused vs non-fused modconv), (iv) hand-computable
known answers.  See tests/test_oracle_*.py.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import
this package.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------
# FIR set-up and padding rules
# ----------------------------------------------------------------------------------------
def setup_kernel(k: Sequence[float]) -> np.ndarray:
    """reference upfirdn_2d_v2.py:18-25 (_setup_kernel): outer product, normalise to sum 1."""
    k = np.asarray(k, dtype=np.float32)
    if k.ndim == 1:
        k = np.outer(k, k)
    k = k / np.sum(k)
    assert k.ndim == 2 and k.shape[0] == k.shape[1]
    return k


def compute_paddings(resample_kernel, up: bool, down: bool, is_conv: bool, convW: int = 3,
                     factor: int = 2, gain: float = 1.0):
    """reference upfirdn_2d_v2.py:28-55.  Returns (k, pad0, pad1).

    NOTE the ``+ 1`` on pad1 of the conv-down case (line 46-47) -- differs from upstream
    StyleGAN2; needed because the height stride may be 1 (conv_downsample_2d:108)."""
    assert not (up and down)
    k = [1] * factor if resample_kernel is None else resample_kernel
    if up:
        k = setup_kernel(k) * (gain * (factor ** 2))
        if is_conv:
            p = (k.shape[0] - factor) - (convW - 1)
            pad0 = (p + 1) // 2 + factor - 1
            pad1 = p // 2 + 1
        else:
            p = k.shape[0] - factor
            pad0 = (p + 1) // 2 + factor - 1
            pad1 = p // 2
    elif down:
        k = setup_kernel(k) * gain
        if is_conv:
            p = (k.shape[0] - factor) + (convW - 1)
            pad0 = (p + 1) // 2
            pad1 = p // 2 + 1
        else:
            p = k.shape[0] - factor
            pad0 = (p + 1) // 2
            pad1 = p // 2
    else:
        k = resample_kernel
        pad0, pad1 = 0, 0
    return k, pad0, pad1


def upfirdn_out_size(n_in: int, up: int, down: int, pad0: int, pad1: int, ktaps: int) -> int:
    """reference upfirdn_2d.cu:254-255 / upfirdn_2d_v2.py:201-202."""
    return (n_in * up + pad0 + pad1 - ktaps) // down + 1


# ----------------------------------------------------------------------------------------
# upfirdn2d
# ----------------------------------------------------------------------------------------
def _floordiv(a: int, b: int) -> int:
    c = int(a / b)  # C truncation
    if c * b > a:
        c -= 1
    return c


def np_upfirdn2d_cu(x: np.ndarray, k: np.ndarray, upx=1, upy=1, downx=1, downy=1,
                    padx0=0, padx1=0, pady0=0, pady1=0) -> np.ndarray:
    """Index arithmetic of the reference's generic CUDA kernel, restated in float64.

    Follows upfirdn_2d.cu:64-117 (UpFirDn2DKernel_large): receptive-field clamp,
    ``kernelY = midY + kH - (inY+1)*upy`` start tap and ``-up`` tap stride.
    x: [major, inH, inW, minor]; k: [kH, kW]; returns [major, outH, outW, minor]."""
    x = np.asarray(x, dtype=np.float64)
    k = np.asarray(k, dtype=np.float64)
    major, inH, inW, minor = x.shape
    kH, kW = k.shape
    outW = (inW * upx + padx0 + padx1 - kW + downx) // downx
    outH = (inH * upy + pady0 + pady1 - kH + downy) // downy
    y = np.zeros((major, outH, outW, minor), dtype=np.float64)
    for oy in range(outH):
        midY = oy * downy + upy - 1 - pady0
        inY = min(max(_floordiv(midY, upy), 0), inH)
        h = min(max(_floordiv(midY + kH, upy), 0), inH) - inY
        kernelY = midY + kH - (inY + 1) * upy
        for ox in range(outW):
            midX = ox * downx + upx - 1 - padx0
            inX = min(max(_floordiv(midX, upx), 0), inW)
            w = min(max(_floordiv(midX + kW, upx), 0), inW) - inX
            kernelX = midX + kW - (inX + 1) * upx
            acc = np.zeros((major, minor), dtype=np.float64)
            for yy in range(h):
                for xx in range(w):
                    acc += x[:, inY + yy, inX + xx, :] * k[kernelY - yy * upy, kernelX - xx * upx]
            y[:, oy, ox, :] = acc
    return y


def t_upfirdn2d(x: torch.Tensor, k, upx=1, upy=1, downx=1, downy=1,
                padx0=0, padx1=0, pady0=0, pady1=0) -> torch.Tensor:
    """The reference's TF-ops twin (upfirdn_2d_v2.py:249-305, upfirdn_2d_ref) in torch:
    zero-insert, pad/crop, VALID correlation with the FLIPPED filter, decimate.
    x: [major, inH, inW, minor] (same layout as the op)."""
    k = torch.as_tensor(np.asarray(k), dtype=x.dtype)
    major, inH, inW, minor = x.shape
    kH, kW = k.shape
    x = x.reshape(major, inH, 1, inW, 1, minor)
    x = F.pad(x, (0, 0, 0, upx - 1, 0, 0, 0, upy - 1))
    x = x.reshape(major, inH * upy, inW * upx, minor)
    x = F.pad(x, (0, 0, max(padx0, 0), max(padx1, 0), max(pady0, 0), max(pady1, 0)))
    x = x[:, max(-pady0, 0): x.shape[1] - max(-pady1, 0), max(-padx0, 0): x.shape[2] - max(-padx1, 0), :]
    x = x.permute(0, 3, 1, 2).reshape(-1, 1, x.shape[1], x.shape[2])
    w = torch.flip(k, (0, 1))[None, None]
    x = F.conv2d(x, w)
    x = x.reshape(major, minor, x.shape[2], x.shape[3]).permute(0, 2, 3, 1)
    return x[:, ::downy, ::downx, :]


def t_simple_upfirdn2d(x: torch.Tensor, k, up=1, down=1, pad0=0, pad1=0, downy=None) -> torch.Tensor:
    """NCHW wrapper, reference upfirdn_2d_v2.py:166-183 (_simple_upfirdn_2d): reshape to
    [B*C, H, W, 1], same factors / pads on both axes.  ``downy`` (extension, not in the
    reference) lets the oracle express the decimated skip-path FIR used by the HIP path."""
    B, C, H, W = x.shape
    y = x.reshape(B * C, H, W, 1)
    y = t_upfirdn2d(y, k, upx=up, upy=up, downx=down, downy=down if downy is None else downy,
                    padx0=pad0, padx1=pad1, pady0=pad0, pady1=pad1)
    return y.reshape(B, C, y.shape[1], y.shape[2])


def upfirdn2d_grad_params(inH, inW, kH, kW, upx, upy, downx, downy, padx0, padx1, pady0, pady1):
    """Parameter transform of the op's gradient (reference upfirdn_2d_v2.py:204-209):
    dx = upfirdn(dy, k[::-1, ::-1], up<->down, gpads)."""
    outW = (inW * upx + padx0 + padx1 - kW) // downx + 1
    outH = (inH * upy + pady0 + pady1 - kH) // downy + 1
    gpadx0 = kW - padx0 - 1
    gpady0 = kH - pady0 - 1
    gpadx1 = inW * upx - outW * downx + padx0 - upx + 1
    gpady1 = inH * upy - outH * downy + pady0 - upy + 1
    return dict(upx=downx, upy=downy, downx=upx, downy=upy,
                padx0=gpadx0, padx1=gpadx1, pady0=gpady0, pady1=gpady1)


# ----------------------------------------------------------------------------------------
# equalised-LR helpers, dense, bias/act, noise
# ----------------------------------------------------------------------------------------
def runtime_coef(weight_shape, gain: float, lrmul: float) -> Tuple[float, float]:
    """reference layers/commons.py:4-12: (init_std, runtime_coef)."""
    fan_in = float(np.prod(weight_shape[:-1]))
    he_std = gain / math.sqrt(fan_in)
    return 1.0 / lrmul, he_std * lrmul


def t_dense(x: torch.Tensor, w: torch.Tensor, gain=1.0, lrmul=1.0) -> torch.Tensor:
    """reference layers/dense.py:23-29: flatten, x @ (coef * w)."""
    _, coef = runtime_coef(w.shape, gain, lrmul)
    return x.reshape(x.shape[0], -1) @ (w * coef)


def t_bias_act(x: torch.Tensor, b: torch.Tensor, act: str, lrmul=1.0) -> torch.Tensor:
    """reference layers/bias_act.py:25-34."""
    bb = b * lrmul
    x = x + (bb if x.dim() == 2 else bb.reshape(1, -1, 1, 1))
    if act == "lrelu":
        return F.leaky_relu(x, 0.2) * math.sqrt(2.0)
    assert act == "linear"
    return x


def t_noise(x: torch.Tensor, noise: torch.Tensor, strength: torch.Tensor) -> torch.Tensor:
    """reference layers/noise.py:12-22 with the noise map injected ([B,1,H,W])."""
    return x + noise * strength


# ----------------------------------------------------------------------------------------
# convolutions (TF semantics -> torch)
# ----------------------------------------------------------------------------------------
def t_conv2d_same(x, w_hwio):
    """tf.nn.conv2d(padding='SAME', stride 1) with HWIO filter (layers/conv.py:66-70)."""
    k = w_hwio.shape[0]
    return F.conv2d(x, w_hwio.permute(3, 2, 0, 1), padding=k // 2)


def t_conv2d_valid(x, w_hwio, stride):
    """tf.nn.conv2d(padding='VALID', strides=(sh, sw)) with HWIO filter."""
    return F.conv2d(x, w_hwio.permute(3, 2, 0, 1), stride=stride)


def t_upsample_conv2d(x, w_hwio, k, pad0, pad1):
    """reference upfirdn_2d_v2.py:65-103 for ONE group: flip the filter spatially, stride-2
    VALID conv2d_transpose -> [.., 2H+1, 2W+1], then FIR with (pad0, pad1).
    TF conv2d_transpose(f[kh,kw,O,I]) == torch conv_transpose2d(weight[I,O,kh,kw] = f[kh,kw,O,I]);
    the reference passes f = w[::-1, ::-1] transposed to [kh,kw,O,I]."""
    wt = torch.flip(w_hwio, (0, 1)).permute(2, 3, 0, 1)  # [I, O, kh, kw]
    y = F.conv_transpose2d(x, wt, stride=2)
    return t_simple_upfirdn2d(y, k, pad0=pad0, pad1=pad1)


def t_conv_downsample2d(x, w_hwio, k, pad0, pad1, reduce_height: bool):
    """reference upfirdn_2d_v2.py:106-113: FIR then VALID conv with stride (2|1, 2)."""
    x = t_simple_upfirdn2d(x, k, pad0=pad0, pad1=pad1)
    return t_conv2d_valid(x, w_hwio, (2 if reduce_height else 1, 2))


def t_modulated_conv2d(x, style, w, mod_w, mod_b, up: bool, demodulate: bool, fused: bool,
                       resample_kernel=(1, 3, 3, 1)):
    """reference layers/modulated_conv2d.py:66-122.

    w: [k,k,I,O]; mod_w: [style_dim, I]; mod_b: [I].  ``fused=True`` follows the GPU branch
    (per-sample weights, grouped conv, lines 85-93/115-118); ``fused=False`` follows the CPU
    branch (scale activations, shared weights, lines 94-96/119-121)."""
    kk, _, I, O = w.shape
    B = x.shape[0]
    _, coef = runtime_coef(w.shape, 1.0, 1.0)
    wc = w * coef
    s = t_bias_act(t_dense(style, mod_w), mod_b, "linear") + 1.0  # [B, I]
    ww = wc[None] * s[:, None, None, :, None]  # [B,k,k,I,O]
    d = None
    if demodulate:
        d = torch.rsqrt(ww.square().sum(dim=(1, 2, 3)) + 1e-8)  # [B, O]
        ww = ww * d[:, None, None, None, :]
    fir_k, pad0, pad1 = compute_paddings(list(resample_kernel), up, False, is_conv=True, convW=kk)
    if fused:
        outs = []
        for bi in range(B):  # one conv per group == grouped conv over [1, B*I, H, W]
            xb = x[bi:bi + 1]
            if up:
                outs.append(t_upsample_conv2d(xb, ww[bi], fir_k, pad0, pad1))
            else:
                outs.append(t_conv2d_same(xb, ww[bi]))
        return torch.cat(outs, 0)
    xs = x * s[:, :, None, None]
    y = t_upsample_conv2d(xs, wc, fir_k, pad0, pad1) if up else t_conv2d_same(xs, wc)
    if demodulate:
        y = y * d[:, :, None, None]
    return y


# ----------------------------------------------------------------------------------------
# float64 numpy definitions used to pin the torch twins
# ----------------------------------------------------------------------------------------
def np_conv2d(x: np.ndarray, w_hwio: np.ndarray, stride=(1, 1), pad=(0, 0, 0, 0)) -> np.ndarray:
    """Direct correlation, NCHW, HWIO filter; pad = (top, bottom, left, right)."""
    x = np.asarray(x, np.float64)
    w = np.asarray(w_hwio, np.float64)
    x = np.pad(x, ((0, 0), (0, 0), (pad[0], pad[1]), (pad[2], pad[3])))
    B, I, H, W = x.shape
    kh, kw, _, O = w.shape
    oh = (H - kh) // stride[0] + 1
    ow = (W - kw) // stride[1] + 1
    y = np.zeros((B, O, oh, ow))
    for a in range(kh):
        for b in range(kw):
            patch = x[:, :, a: a + (oh - 1) * stride[0] + 1: stride[0], b: b + (ow - 1) * stride[1] + 1: stride[1]]
            y += np.einsum("bihw,io->bohw", patch, w[a, b])
    return y


def np_conv_transpose2d_s2(x: np.ndarray, wt_hwio: np.ndarray) -> np.ndarray:
    """y[o, 2a+kh, 2b+kw] += x[i,a,b] * wt[kh,kw,i,o]  (scatter definition, stride 2)."""
    x = np.asarray(x, np.float64)
    w = np.asarray(wt_hwio, np.float64)
    B, I, H, W = x.shape
    kh, kw, _, O = w.shape
    y = np.zeros((B, O, (H - 1) * 2 + kh, (W - 1) * 2 + kw))
    for a in range(kh):
        for b in range(kw):
            y[:, :, a: a + 2 * H: 2, b: b + 2 * W: 2] += np.einsum("bihw,io->bohw", x, w[a, b])
    return y


def np_modulated_conv2d_def(x, s, w_hwio, demodulate=True) -> np.ndarray:
    """Per-sample-weight definition of the (non-up) modulated conv, float64."""
    x = np.asarray(x, np.float64)
    s = np.asarray(s, np.float64)
    w = np.asarray(w_hwio, np.float64)
    kk, _, I, O = w.shape
    wc = w / math.sqrt(kk * kk * I)
    outs = []
    for bi in range(x.shape[0]):
        ww = wc * s[bi][None, None, :, None]
        if demodulate:
            ww = ww / np.sqrt((ww ** 2).sum(axis=(0, 1, 2)) + 1e-8)[None, None, None, :]
        outs.append(np_conv2d(x[bi:bi + 1], ww, pad=(kk // 2,) * 4))
    return np.concatenate(outs, 0)


def np_minibatch_std(x: np.ndarray, group_size=4) -> np.ndarray:
    """reference layers/mini_batch_std.py:10-35, float64."""
    x = np.asarray(x, np.float64)
    B, C, H, W = x.shape
    G = min(group_size, B)
    y = x.reshape(G, -1, 1, C, H, W)
    y = y - y.mean(axis=0, keepdims=True)
    y = np.sqrt((y ** 2).mean(axis=0) + 1e-8)
    y = y.mean(axis=(2, 3, 4), keepdims=True).mean(axis=2)  # [M,1,1,1]
    y = np.tile(y, (G, 1, H, W))
    return np.concatenate([x, y], axis=1)


def t_minibatch_std(x: torch.Tensor, group_size=4) -> torch.Tensor:
    B, C, H, W = x.shape
    G = min(group_size, B)
    y = x.reshape(G, -1, 1, C, H, W)
    y = y - y.mean(dim=0, keepdim=True)
    y = (y.square().mean(dim=0) + 1e-8).sqrt()
    y = y.mean(dim=(2, 3, 4), keepdim=True).mean(dim=2)
    y = y.repeat(G, 1, H, W)
    return torch.cat([x, y], dim=1)


def t_mask_text_box(img: torch.Tensor, words: torch.Tensor, char_width: int) -> torch.Tensor:
    """reference utils/utils.py:11-45: zero the columns of padded character slots."""
    mask = (words != 0).to(img.dtype).repeat_interleave(char_width, dim=1)  # [B, W]
    return img * mask[:, None, None, :]
