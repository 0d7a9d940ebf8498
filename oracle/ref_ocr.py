"""TEST INFRASTRUCTURE (never imported by the product): CPU restatement of the frozen OCR network that sits under
``AsterInferer`` in the training step (reference aster_ocr_utils/aster_inferer.py:24-37: ``tf.saved_model.load`` of the ASTER
SavedModel, called one sample at a time through its serving signature).

PARITY UNPINNED -- and doubly so: the SavedModel is an external artefact that is absent here, so the network itself is the
published ASTER architecture (Shi et al., TPAMI 2018: TPS rectification with a localisation CNN, the 45-layer ResNet of its
table 1, two BiLSTM layers, a Bahdanau-attention LSTM decoder with greedy feedback; weigths_tf1_to_tf2.py:3-19 names the same
parts) with SYNTHETIC frozen weights.  What this file pins is narrower and still worth pinning: the product's HIP execution of
that network (MFMA convolutions with folded BatchNorm, lstm_step / attn_ctx kernels, hand-written backward) against an
implementation that shares NO code with it -- round 3's oracle called the product's own torch module here.

The weights are DATA: a ``state_dict`` (name -> tensor) in the layout the product's module exports (PyTorch conventions: OIHW
convolutions, LSTM gates in i, f, g, o order, ``weight_ih_l{k}[_reverse]`` ...), exactly as a checkpoint would be handed to two
implementations.  Everything else -- the TPS constants, BatchNorm folding-free evaluation, the explicit LSTM recurrences, the
decoder loop, the dynamic decode length -- is written out here from the definitions."""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

EOS = 1          # class 1 = blank / end of sequence (char_tokens.py:9,16-17; utils/utils.py:102-105)
NUM_CTRL = 20    # TPS control points (10 on the top edge, 10 on the bottom edge)
RECT_HW = (32, 100)
RES_STAGES = [(32, 3, (2, 2)), (64, 4, (2, 2)), (128, 6, (2, 1)), (256, 6, (2, 1)), (512, 3, (2, 1))]  # ASTER table 1


def tps_interpolation(num_ctrl: int = NUM_CTRL, out_hw=RECT_HW, margin: float = 0.05) -> np.ndarray:
    """[H*W, K] weights W with  source_xy(pixel) = W @ source_control_points  for a thin-plate spline whose TARGET control points
    sit evenly on the top and bottom edges (margin 0.05) of the rectified image (ASTER section 3.1, eq. 1-5):
    with U(r) = r^2 log r the spline through K points is  f(p) = [U(|p - c_k|)]_k a + b0 + B p  subject to the side conditions
    sum a = 0, sum a c = 0; solving the (K+3) system for unit data at each control point gives the interpolation weights."""
    k = num_ctrl // 2
    xs = np.linspace(margin, 1.0 - margin, k)
    c = np.concatenate([np.stack([xs, np.full(k, margin)], 1), np.stack([xs, np.full(k, 1.0 - margin)], 1)], 0)  # [K, 2]
    K = num_ctrl

    def U(d2):  # r^2 log r = 0.5 d2 log d2
        return 0.5 * d2 * np.log(np.maximum(d2, 1e-12))

    L = np.zeros((K + 3, K + 3))
    L[:K, :K] = U(((c[:, None] - c[None]) ** 2).sum(-1))
    L[:K, K], L[:K, K + 1:] = 1.0, c
    L[K, :K], L[K + 1:, :K] = 1.0, c.T
    H, W = out_hw
    gy, gx = np.meshgrid((np.arange(H) + 0.5) / H, (np.arange(W) + 0.5) / W, indexing="ij")
    p = np.stack([gx.ravel(), gy.ravel()], 1)
    lifted = np.concatenate([U(((p[:, None] - c[None]) ** 2).sum(-1)), np.ones((H * W, 1)), p], 1)
    coef = np.linalg.solve(L, np.concatenate([np.eye(K), np.zeros((3, K))], 0))  # [K+3, K]: spline coefficients per unit datum
    return lifted @ coef  # float64


class OcrOracle:
    """forward pass + the SavedModel-like serving signature; differentiable by torch autograd on the CPU."""

    def __init__(self, state: Dict[str, torch.Tensor], max_steps: int = 8, dtype=torch.float32):
        self.w = {k: v.detach().to("cpu", dtype).clone() for k, v in state.items() if v.is_floating_point()}
        self.max_steps, self.dtype = max_steps, dtype
        self.hidden = self.w["cell.weight_hh"].shape[1]
        self.num_classes = self.w["out.weight"].shape[0]
        # the spline weights are a constant fp32 tensor OF THE NETWORK (computed in float64, stored in fp32 like every weight)
        self.tps = torch.from_numpy(tps_interpolation().astype(np.float32)).to(dtype)
        self.has_backward_predictor = "bwd.out.weight" in self.w

    # ---- building blocks --------------------------------------------------------------------------------------------------
    def _conv(self, name, x, stride=(1, 1), relu=True, residual=None, bn=True):
        w = self.w[name + ".conv.weight"]
        y = F.conv2d(x, w, self.w.get(name + ".conv.bias"), stride=stride, padding=w.shape[2] // 2)
        if bn:  # frozen BatchNorm: running statistics (eps = 1e-5)
            g, b = self.w[name + ".bn.weight"], self.w[name + ".bn.bias"]
            m, v = self.w[name + ".bn.running_mean"], self.w[name + ".bn.running_var"]
            y = (y - m[None, :, None, None]) / torch.sqrt(v[None, :, None, None] + 1e-5) * g[None, :, None, None] + b[None, :, None, None]
        if residual is not None:
            y = y + residual
        return torch.relu(y) if relu else y

    def _lstm_dir(self, x, wi, wh, bi, bh, reverse):
        """one direction of one LSTM layer, x [B, T, F] -> [B, T, H]; gates i, f, g, o."""
        B, T, _ = x.shape
        H = wh.shape[1]
        h = x.new_zeros(B, H)
        c = x.new_zeros(B, H)
        out = [None] * T
        for t in (range(T - 1, -1, -1) if reverse else range(T)):
            z = x[:, t] @ wi.t() + bi + h @ wh.t() + bh
            i, f, g, o = z[:, :H], z[:, H:2 * H], z[:, 2 * H:3 * H], z[:, 3 * H:]
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            h = torch.sigmoid(o) * torch.tanh(c)
            out[t] = h
        return torch.stack(out, dim=1)

    def _bilstm(self, x):
        for layer in (0, 1):
            ys = []
            for sfx, rev in (("", False), ("_reverse", True)):
                p = lambda n: self.w[f"rnn.{n}_l{layer}{sfx}"]
                ys.append(self._lstm_dir(x, p("weight_ih"), p("weight_hh"), p("bias_ih"), p("bias_hh"), rev))
            x = torch.cat(ys, dim=2)
        return x

    def _decode(self, enc, pre=""):
        """Bahdanau attention + LSTM cell, greedy feedback (the argmax is not differentiated, as in the TF decoder)."""
        w = lambda n: self.w[pre + n]
        B, H = enc.shape[0], self.hidden
        enc_proj = enc @ w("att_enc.weight").t()
        h, c = enc.new_zeros(B, H), enc.new_zeros(B, H)
        prev = torch.full((B,), self.num_classes, dtype=torch.long)  # GO symbol = one past the classes
        outs = []
        for _ in range(self.max_steps):
            q = h @ w("att_dec.weight").t() + w("att_dec.bias")
            e = torch.tanh(enc_proj + q[:, None, :]) @ w("att_v.weight").t()            # [B, T, 1]
            a = torch.softmax(e.squeeze(2), dim=1)
            ctx = (a[:, :, None] * enc).sum(dim=1)
            z = torch.cat([ctx, w("emb.weight")[prev]], dim=1) @ w("cell.weight_ih").t() + w("cell.bias_ih") \
                + h @ w("cell.weight_hh").t() + w("cell.bias_hh")
            i, f, g, o = z[:, :H], z[:, H:2 * H], z[:, 2 * H:3 * H], z[:, 3 * H:]
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
            h = torch.sigmoid(o) * torch.tanh(c)
            logit = h @ w("out.weight").t() + w("out.bias")
            outs.append(logit)
            prev = logit.detach().argmax(dim=1)
        return torch.stack(outs, dim=1)

    # ---- the network ------------------------------------------------------------------------------------------------------
    def rectify(self, img):
        B = img.shape[0]
        f = F.interpolate(img, size=(32, 64), mode="bilinear", align_corners=False)
        for i in range(6):
            f = self._conv(f"loc_convs.{i}", f, bn=False)
            if i < 5:
                f = F.max_pool2d(f, 2, 2)
        hdn = torch.relu(f.reshape(B, -1) @ self.w["loc_fc1.weight"].t() + self.w["loc_fc1.bias"])
        ctrl = (hdn @ self.w["loc_fc2.weight"].t() + self.w["loc_fc2.bias"]).reshape(B, NUM_CTRL, 2)
        src = torch.matmul(self.tps, ctrl)  # [B, HW, 2] in [0, 1]
        grid = (src * 2.0 - 1.0).reshape(B, RECT_HW[0], RECT_HW[1], 2)
        return F.grid_sample(img, grid, mode="bilinear", padding_mode="border", align_corners=False)

    def encode(self, x):
        x = self._conv("stem", x)
        u, cin = 0, 32
        for cout, n, stride in RES_STAGES:
            for k in range(n):
                s = stride if k == 0 else (1, 1)
                sc = self._conv(f"resnet.{u}.short", x, stride=s, relu=False) if (cin != cout or s != (1, 1)) else x
                x = self._conv(f"resnet.{u}.c2", self._conv(f"resnet.{u}.c1", x, stride=s), residual=sc)
                cin, u = cout, u + 1
        return x  # [B, 512, 1, 25]

    def features(self, img_nchw):
        x = self.encode(self.rectify(img_nchw.to(self.dtype)))
        return self._bilstm(x.squeeze(2).permute(0, 2, 1))

    def forward(self, img_nchw):
        """[B, 3, 64, 256] in [-1, 1] -> forward logits [B, max_steps, classes] (every decoder step)."""
        return self._decode(self.features(img_nchw))

    __call__ = forward

    @staticmethod
    def decode_lengths(logits):
        """steps the dynamic decode emits per sample: up to and including the first greedy EOS, else all of them"""
        S = logits.shape[1]
        is_eos = logits.argmax(dim=2) == EOS
        return torch.where(is_eos.any(dim=1), is_eos.to(torch.int32).argmax(dim=1) + 1,
                           torch.full((logits.shape[0],), S, dtype=torch.int64))

    def serve(self, inputs_nhwc):
        """the serving signature as aster_inferer.py:31 calls it: NHWC [1, 64, 256, 3] -> {"forward_logits": [1, T, C]}
        (+ "backward_logits": the second predictor on the time-reversed features, weigths_tf1_to_tf2.py:8-13)."""
        assert inputs_nhwc.shape[0] == 1, "the reference calls the SavedModel one sample at a time"
        enc = self.features(inputs_nhwc.permute(0, 3, 1, 2))
        fwd = self._decode(enc)
        res = {"forward_logits": fwd[:, : int(self.decode_lengths(fwd.detach())[0])]}
        if self.has_backward_predictor:
            bwd = self._decode(enc.flip(1), pre="bwd.")
            res["backward_logits"] = bwd[:, : int(self.decode_lengths(bwd.detach())[0])]
        return res
