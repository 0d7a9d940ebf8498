"""Frozen OCR branch of the training step: the ``AsterInferer`` wrapper + an ASTER-shaped network.

Wrapper semantics (exact restatement, reference ``aster_ocr_utils/aster_inferer.py``):

* ``convert_inputs`` (:153-190): crop every fake image to its word's width (index of the first
  label == blank(1) times char_width), bilinear-resize (TF2 half-pixel centres, no antialias)
  to 64x256.  Here the per-sample crop+resize is ONE batched matmul with a per-word-length
  interpolation matrix (the resize is linear in the pixels and the height scale is 1).
* ``call`` (:28-37) + ``_postprocess_simple`` (:116-151): per sample, the forward logits
  ``[1, T_i, C]`` of the SavedModel are cut to the first 8 steps and, when ``T_i < 8``, padded
  with ``1000 * onehot(class 1)``.  The reference calls the SavedModel once per sample; the
  network is frozen/eval, so one batched call computes the same rows -- the per-sample length
  ``T_i`` (a property of the NETWORK's dynamic decode, not of the wrapper) travels beside the
  batched logits as ``lengths`` and ``_postprocess_simple`` applies the literal rule row-wise.

The network: the ASTER SavedModel is an external download that is NOT in the reference
(``aster_weights/.keep`` only) -- architecture, class count and weights cannot be derived
from it => **parity with the real ASTER is UNPINNED**.  ``AsterLikeOCR`` follows
the published ASTER design (Shi et al., TPAMI 2018: TPS rectification -> 45-layer ResNet ->
2x BiLSTM -> Bahdanau-attention LSTM decoder, greedy feedback; the decoder type is also what
``weigths_tf1_to_tf2.py:3-19`` names) with deterministic synthetic frozen weights, in plain
PyTorch-ROCm ops as BASELINE.json's north_star prescribes for this branch.
``load_weights_npz`` is the import hook for someone holding the real weights.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

EOS = 1  # blank / end-of-sequence class (char_tokens.py: <OOV> index 1; utils.py:102-105 pads with 1)


# ----------------------------------------------------------------------------------------
# wrapper pieces
# ----------------------------------------------------------------------------------------
def _resize_matrix(in_w: int, out_w: int, full_w: int) -> np.ndarray:
    """[out_w, full_w] matrix of TF2/torch half-pixel bilinear resize from the first ``in_w``
    columns (zero weight on columns >= in_w)."""
    m = np.zeros((out_w, full_w), dtype=np.float64)
    scale = in_w / out_w
    for x in range(out_w):
        src = max((x + 0.5) * scale - 0.5, 0.0)
        x0 = min(int(math.floor(src)), in_w - 1)
        x1 = min(x0 + 1, in_w - 1)
        lam = src - x0
        m[x, x0] += 1.0 - lam
        m[x, x1] += lam
    return m.astype(np.float32)


class AsterInferer(nn.Module):
    """Drop-in for the reference's ``AsterInferer`` (same method names)."""

    def __init__(self, model: Optional[nn.Module] = None, char_width: int = 32, max_char_number: int = 8,
                 image_dims=(64, 256), combine_forward_and_backward: bool = False):
        super().__init__()
        self.combine_forward_and_backward = combine_forward_and_backward
        if model is None:
            model = AsterLikeOCR(max_steps=max_char_number, backward_predictor=combine_forward_and_backward)
        if combine_forward_and_backward and not getattr(model, "has_backward_predictor", False):
            raise ValueError("combine_forward_and_backward needs a network with a backward predictor (backward_logits)")
        self.model = model
        for p in self.model.parameters():
            p.requires_grad_(False)
        self.model.train(False)
        self.char_width = char_width
        self.max_char_number = max_char_number
        self.image_dims = tuple(image_dims)
        full_w = char_width * max_char_number
        assert image_dims[1] == full_w, "aster_image_dims width must equal the text-box width"
        mats = np.stack([_resize_matrix(char_width * max(L, 1), image_dims[1], full_w)
                         for L in range(max_char_number + 1)])
        mats[0] = mats[max_char_number]  # L=0 never happens for real words; keep it finite
        self.register_buffer("resize_mats", torch.from_numpy(mats), persistent=False)

    def train(self, mode: bool = True):  # frozen: always inference behaviour
        super().train(False)
        self.model.train(False)
        return self

    def convert_inputs(self, fake_images: torch.Tensor, labels: torch.Tensor, blank_label: int = 1) -> torch.Tensor:
        """[B,3,64,256] NCHW, labels [B,8] -> NHWC [B,64,256,3] resized crop."""
        B, C, H, W = fake_images.shape
        assert H == self.image_dims[0], "height resize is the identity in the reference geometry"
        is_blank = labels == blank_label
        first_blank = torch.where(is_blank.any(dim=1), is_blank.to(torch.int32).argmax(dim=1),
                                  torch.full((B,), self.max_char_number, device=labels.device, dtype=torch.int64))
        m = self.resize_mats[first_blank]  # [B, 256, 256]
        out = torch.bmm(fake_images.reshape(B, C * H, W), m.transpose(1, 2)).reshape(B, C, H, -1)
        return out.permute(0, 2, 3, 1)

    def forward(self, inputs_nhwc: torch.Tensor) -> torch.Tensor:
        """aster_inferer.py:28-37 (``call``), batched: ``forward_logits`` of every sample, post-processed.
        combine_forward_and_backward (off by default, as in the reference :12-15,19): the literal per-sample loop with
        data-dependent shapes -- eager only (not HIP-graph capturable)."""
        if self.combine_forward_and_backward:
            rows = []
            for i in range(inputs_nhwc.shape[0]):
                rows.append(self._postprocess_combine(self.model.serve(inputs_nhwc[i:i + 1])))
            return torch.cat(rows, dim=0)
        logits, lengths = self.model.forward_logits(inputs_nhwc.permute(0, 3, 1, 2))  # [B, S, C], [B] (T_i of sample i)
        return self._postprocess_simple(logits, lengths)

    def _postprocess_combine(self, logits: dict) -> torch.Tensor:
        """aster_inferer.py:39-82 for ONE sample's {"forward_logits", "backward_logits"} [1, T, C]."""
        forward_logits = logits["forward_logits"][:, : self.max_char_number]
        backward_logits = logits["backward_logits"][:, : self.max_char_number]
        combined = self._combine_logits(forward_logits, backward_logits)
        remaining = forward_logits[:, combined.shape[1]:, :]
        padding_len = self.max_char_number - forward_logits.shape[1]
        C = combined.shape[2]
        pad_row = (torch.arange(C, device=combined.device) == EOS).to(combined.dtype) * 1000.0
        return torch.cat([combined, remaining, pad_row.expand(1, padding_len, C)], dim=1)

    @staticmethod
    def _combine_logits(forward_logits: torch.Tensor, backward_logits: torch.Tensor) -> torch.Tensor:
        """aster_inferer.py:84-114: drop the steps that predict blank, reverse the backward predictor's steps, and per step
        keep whichever predictor is more confident (larger maximum logit)."""
        forward_mask = forward_logits.argmax(dim=2) != EOS
        backward_mask = backward_logits.argmax(dim=2) != EOS
        masked_forward = forward_logits[forward_mask]
        masked_backward = backward_logits[backward_mask].flip(0)
        crop_forward = masked_forward[: masked_backward.shape[0]]
        crop_backward = masked_backward[: masked_forward.shape[0]]
        forward_max = crop_forward.max(dim=1).values
        backward_max = crop_backward.max(dim=1).values
        combined = torch.where(forward_max[:, None] > backward_max[:, None], crop_forward, crop_backward)
        return combined[None]

    def _postprocess_simple(self, logits: torch.Tensor, lengths: Optional[torch.Tensor] = None) -> torch.Tensor:
        """aster_inferer.py:116-151 for a batch of samples whose own logits have ``lengths[i]`` time steps (``None``:
        all ``logits.shape[1]``): keep the first ``max_char_number`` steps, pad the missing ones with
        ``1000 * onehot(class 1)``.  Row-wise ``where`` instead of per-sample concat: no host sync (HIP-graph capturable)."""
        logits = logits[:, : self.max_char_number]
        B, T, C = logits.shape
        pad_row = (torch.arange(C, device=logits.device) == EOS).to(logits.dtype) * 1000.0
        if T < self.max_char_number:  # padding_len > 0 for every sample (:134-149)
            logits = torch.cat([logits, pad_row.expand(B, self.max_char_number - T, C)], dim=1)
        if lengths is not None:  # sample i only HAS rows t < lengths[i]; the rest is its padding
            missing = torch.arange(self.max_char_number, device=logits.device)[None, :] >= lengths[:, None]
            logits = torch.where(missing[:, :, None], pad_row, logits)
        return logits


# ----------------------------------------------------------------------------------------
# ASTER-shaped network (published architecture; synthetic weights)
# ----------------------------------------------------------------------------------------
def _tps_constants(num_ctrl: int, out_h: int, out_w: int, margin: float = 0.05):
    """Target control points on the top/bottom edges of the rectified image, the inverse of
    the TPS system matrix and the lifted target grid (ASTER section 3.1)."""
    k = num_ctrl // 2
    xs = np.linspace(margin, 1.0 - margin, k)
    ctrl = np.concatenate([np.stack([xs, np.full(k, margin)], 1), np.stack([xs, np.full(k, 1.0 - margin)], 1)], 0)

    def phi(d2):
        return 0.5 * d2 * np.log(np.maximum(d2, 1e-12))  # r^2 log r with r^2 = d2

    d2 = ((ctrl[:, None, :] - ctrl[None, :, :]) ** 2).sum(-1)
    K = num_ctrl
    A = np.zeros((K + 3, K + 3))
    A[:K, :K] = phi(d2)
    A[:K, K] = 1.0
    A[:K, K + 1:] = ctrl
    A[K, :K] = 1.0
    A[K + 1:, :K] = ctrl.T
    inv = np.linalg.inv(A)
    gy, gx = np.meshgrid((np.arange(out_h) + 0.5) / out_h, (np.arange(out_w) + 0.5) / out_w, indexing="ij")
    pts = np.stack([gx.ravel(), gy.ravel()], 1)
    d2g = ((pts[:, None, :] - ctrl[None, :, :]) ** 2).sum(-1)
    lifted = np.concatenate([phi(d2g), np.ones((pts.shape[0], 1)), pts], 1)  # [HW, K+3]
    # source = lifted @ inv @ [ctrl_src; 0]: fold the two constant factors in float64 -- the TPS system
    # matrix is ill conditioned and its fp32 inverse alone costs 1e-3 of accuracy on the sampling grid
    interp = (lifted @ inv)[:, :K]  # [HW, K] interpolation weights of the K source control points
    return ctrl.astype(np.float32), interp.astype(np.float32)


class _ConvBN(nn.Module):
    """conv (+ optional eval-mode BatchNorm) (+ residual) (+ ReLU): the only convolution form in the net.
    ``AsterLikeOCR._run`` executes it with torch ops; ``AsterLikeOCRHip`` overrides the executor."""

    def __init__(self, cin, cout, k, stride=(1, 1), bn=True, relu=True, bias=False):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride, k // 2, bias=bias)
        self.bn = nn.BatchNorm2d(cout) if bn else None
        self.relu = relu

    def torch_forward(self, x, residual=None):
        y = self.conv(x)
        if self.bn is not None:
            y = self.bn(y)
        if residual is not None:
            y = y + residual
        return F.relu(y) if self.relu else y

    def folded(self):
        """(w [k,k,I,O] HWIO, b [O]) with the frozen BatchNorm folded in."""
        w = self.conv.weight
        b = self.conv.bias if self.conv.bias is not None else torch.zeros(w.shape[0], device=w.device)
        if self.bn is not None:
            g = self.bn.weight / torch.sqrt(self.bn.running_var + self.bn.eps)
            w = w * g[:, None, None, None]
            b = (b - self.bn.running_mean) * g + self.bn.bias
        return w.permute(2, 3, 1, 0).contiguous(), b.contiguous()


class _ResUnit(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.c1 = _ConvBN(cin, cout, 1, stride)
        self.c2 = _ConvBN(cout, cout, 3, (1, 1))  # relu applied after the residual add
        self.short = _ConvBN(cin, cout, 1, stride, relu=False) if (cin != cout or stride != (1, 1)) else None


class AsterLikeOCR(nn.Module):
    def __init__(self, num_classes: int = 97, max_steps: int = 8, hidden: int = 256, num_ctrl: int = 20,
                 rect_hw=(32, 100), seed: int = 20180625, backward_predictor: bool = False):
        super().__init__()
        self.has_backward_predictor = backward_predictor
        self.num_classes, self.max_steps, self.hidden = num_classes, max_steps, hidden
        self.rect_hw = rect_hw
        # --- rectification (STN): localisation CNN on a 32x64 thumbnail -> 2K control coordinates
        chans = [3, 32, 64, 128, 256, 256, 256]
        self.loc_convs = nn.ModuleList([_ConvBN(chans[i], chans[i + 1], 3, bn=False, bias=True) for i in range(6)])
        self.loc_fc1 = nn.Linear(256 * 1 * 2, 512)
        self.loc_fc2 = nn.Linear(512, 2 * num_ctrl)
        ctrl, interp = _tps_constants(num_ctrl, rect_hw[0], rect_hw[1])
        self.register_buffer("tps_interp", torch.from_numpy(interp), persistent=False)
        self.register_buffer("ctrl_init", torch.from_numpy(ctrl.reshape(-1)), persistent=False)
        # --- encoder: ResNet (ASTER table 1) + 2 BiLSTM
        self.stem = _ConvBN(3, 32, 3)
        cfgs = [(32, 3, (2, 2)), (64, 4, (2, 2)), (128, 6, (2, 1)), (256, 6, (2, 1)), (512, 3, (2, 1))]
        blocks, cin = [], 32
        for cout, n, stride in cfgs:
            for u in range(n):
                blocks.append(_ResUnit(cin, cout, stride if u == 0 else (1, 1)))
                cin = cout
        self.resnet = nn.ModuleList(blocks)
        self.rnn = nn.LSTM(512, hidden, num_layers=2, bidirectional=True, batch_first=True)
        # --- attention decoder (forward direction only is consumed: aster_inferer.py:35)
        self.emb = nn.Embedding(num_classes + 1, hidden)  # +1: GO symbol
        self.att_enc = nn.Linear(2 * hidden, hidden, bias=False)
        self.att_dec = nn.Linear(hidden, hidden)
        self.att_v = nn.Linear(hidden, 1, bias=False)
        self.cell = nn.LSTMCell(2 * hidden + hidden, hidden)
        self.out = nn.Linear(hidden, num_classes)
        if backward_predictor:  # ASTER's second predictor (weigths_tf1_to_tf2.py:8-13 "Backward/Predictor"): the same decoder
            # on the time-reversed encoder features; registered LAST so the forward network's synthetic weights are unchanged
            self.bwd = nn.ModuleDict(dict(emb=nn.Embedding(num_classes + 1, hidden),
                                          att_enc=nn.Linear(2 * hidden, hidden, bias=False), att_dec=nn.Linear(hidden, hidden),
                                          att_v=nn.Linear(hidden, 1, bias=False), cell=nn.LSTMCell(2 * hidden + hidden, hidden),
                                          out=nn.Linear(hidden, num_classes)))
        self._synthetic_init(seed)
        self.train(False)

    @torch.no_grad()
    def _synthetic_init(self, seed):
        g = torch.Generator().manual_seed(seed)
        for name, p in self.named_parameters():
            if p.dim() == 4:  # convolutions: He normal
                p.copy_(torch.randn(p.shape, generator=g) * math.sqrt(2.0 / p[0].numel()))
            elif p.dim() >= 2:  # dense / recurrent: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) keeps the gates unsaturated
                lim = 1.0 / math.sqrt(p.shape[1])
                p.copy_((torch.rand(p.shape, generator=g) * 2.0 - 1.0) * lim)
            else:
                p.zero_()
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.weight.fill_(1.0)
                m.running_var.fill_(1.0)
        self.loc_fc2.weight.mul_(0.05)
        self.loc_fc2.bias.copy_(self.ctrl_init)
        for n, p in self.rnn.named_parameters():  # unit forget-gate bias
            if "bias_ih" in n:
                p[self.hidden: 2 * self.hidden].fill_(1.0)

    def train(self, mode: bool = True):
        """Frozen network: BatchNorm always uses its running statistics.  The (dropout-free) LSTM
        is kept in 'training' mode because MIOpen only provides the RNN data gradient there --
        numerically identical, and the OCR loss must back-propagate into the generator."""
        super().train(False)
        self.rnn.train(True)
        return self

    def load_weights_npz(self, path: str, name_map: Optional[dict] = None):
        """Import hook: ``.npz`` of arrays keyed by this module's state_dict names (or mapped
        through ``name_map``); conv kernels expected HWIO (TF) and transposed here."""
        data = np.load(path)
        sd = self.state_dict()
        for k in sd:
            src = (name_map or {}).get(k, k)
            if src in data:
                a = torch.from_numpy(data[src])
                if a.dim() == 4 and a.shape != sd[k].shape:
                    a = a.permute(3, 2, 0, 1)
                sd[k].copy_(a.reshape(sd[k].shape))
        self._invalidate()

    # ---- real-weight import.  The ASTER SavedModel (cfg.aster_weights, aster_inferer.py:24-26) and
    # the TF1 checkpoint it was converted from (weigths_tf1_to_tf2.py) store their variables as TensorBundle files
    # (``<dir>/variables/variables.index`` + ``.data-*``, or a name-based ``model.ckpt``): both are read WITHOUT
    # TensorFlow by textboxgan_amd/tf_checkpoint.py.  Which variable feeds which layer of this stand-in cannot be
    # derived here (the graph is absent: parity unpinned), so the caller supplies ``name_map``: module parameter name ->
    # checkpoint variable name (after the renames of weigths_tf1_to_tf2.py:3-19, listed by list_tf_variables).
    @staticmethod
    def list_tf_variables(path: str):
        """{variable name: shape} of a SavedModel directory or a checkpoint prefix."""
        from . import tf_checkpoint as T
        import os
        prefix = os.path.join(path, "variables", "variables") if os.path.isdir(path) else path
        _, entries = T.read_bundle_index(prefix)
        strip = lambda k: k[: -len(T.VAR_SUFFIX)] if k.endswith(T.VAR_SUFFIX) else k
        return {strip(k): e.shape for k, e in entries.items() if k != T.OBJECT_GRAPH_KEY}

    @staticmethod
    def tf_to_torch_layout(a: np.ndarray, kind: str, hidden: int = 0) -> np.ndarray:
        """TF variable -> the layout of the matching torch parameter.
        "conv"  HWIO [kh,kw,I,O] -> OIHW;   "dense" [in,out] -> [out,in];   "same": unchanged;
        "lstm_kernel" [in+hidden, 4*hidden] with TF gate order (i, j, f, o) -> (weight_ih [4h,in], weight_hh [4h,hidden]) in
        torch order (i, f, g, o);   "lstm_bias" [4h] (i, j, f, o) -> [4h] (i, f, g, o) (TF's forget_bias is NOT added here)."""
        if kind == "conv":
            return np.ascontiguousarray(a.transpose(3, 2, 0, 1))
        if kind == "dense":
            return np.ascontiguousarray(a.T)
        if kind == "same":
            return a
        if kind in ("lstm_kernel", "lstm_bias"):
            h = hidden or a.shape[-1] // 4
            i, j, f, o = np.split(a, 4, axis=-1)
            re = np.concatenate([i, f, j, o], axis=-1)
            if kind == "lstm_bias":
                return re
            n_in = a.shape[0] - h
            return np.ascontiguousarray(re[:n_in].T), np.ascontiguousarray(re[n_in:].T)
        raise ValueError(kind)

    def load_weights_tf(self, path: str, name_map: dict, strict: bool = True, forget_bias: float = 1.0):
        """Import variables of a SavedModel directory / checkpoint prefix.  name_map: parameter name -> variable name or
        (variable name, kind) with kind as in tf_to_torch_layout (default: "conv" for 4-D, "dense" for 2-D, else "same").
        An LSTM's fused kernel maps from its ``weight_ih`` name and fills the matching ``weight_hh`` too.  An LSTM's single
        TF bias maps from one torch bias name (``bias_ih*`` or ``bias_hh*``): the PAIRED torch bias is zeroed (torch adds
        both), and ``forget_bias`` is added to the f-gate slice of the imported bias: TF's LSTMCell / BasicLSTMCell add it to the f
        gate at RUN time (it is not in the stored variable) and default to 1.0, which is what the ASTER SavedModel of
        aster_inferer.py:24-26 was exported with -- hence the default here; pass 0.0 only for cells built with forget_bias=0."""
        from . import tf_checkpoint as T
        import os
        prefix = os.path.join(path, "variables", "variables") if os.path.isdir(path) else path
        _, entries = T.read_bundle_index(prefix)
        key_of = {}
        for k in entries:
            key_of[k[: -len(T.VAR_SUFFIX)] if k.endswith(T.VAR_SUFFIX) else k] = k
        sd = self.state_dict()
        missing = []
        with torch.no_grad():
            for pname, spec in name_map.items():
                vname, kind = spec if isinstance(spec, tuple) else (spec, None)
                if vname not in key_of:
                    missing.append(vname)
                    continue
                a = T.read_bundle(prefix, [key_of[vname]])[key_of[vname]]
                kind = kind or ("conv" if a.ndim == 4 else "dense" if a.ndim == 2 else "same")
                conv = self.tf_to_torch_layout(a, kind, self.hidden)
                if kind == "lstm_kernel":
                    sd[pname].copy_(torch.from_numpy(conv[0]))
                    sd[pname.replace("weight_ih", "weight_hh")].copy_(torch.from_numpy(conv[1]))
                elif kind == "lstm_bias":
                    b = torch.from_numpy(np.ascontiguousarray(conv)).reshape(sd[pname].shape).clone()
                    h = b.numel() // 4
                    b[h: 2 * h] += forget_bias  # torch gate order i, f, g, o
                    sd[pname].copy_(b)
                    pair = pname.replace("bias_ih", "bias_hh") if "bias_ih" in pname else pname.replace("bias_hh", "bias_ih")
                    if pair != pname and pair in sd:
                        sd[pair].zero_()
                else:
                    sd[pname].copy_(torch.from_numpy(np.ascontiguousarray(conv)).reshape(sd[pname].shape))
        if missing and strict:
            raise KeyError(f"variables not found in {path}: {missing[:5]}")
        self._invalidate()
        return missing

    def _invalidate(self):
        pass

    # the single convolution executor (overridden by the HIP subclass)
    def _run(self, blk: _ConvBN, x, residual=None):
        return blk.torch_forward(x, residual)

    def rectify(self, img: torch.Tensor) -> torch.Tensor:
        B = img.shape[0]
        f = F.interpolate(img, size=(32, 64), mode="bilinear", align_corners=False)
        for i, blk in enumerate(self.loc_convs):
            f = self._run(blk, f)
            if i < 5:
                f = F.max_pool2d(f, 2, 2)
        ctrl = self.loc_fc2(F.relu(self.loc_fc1(f.reshape(B, -1)))).reshape(B, -1, 2)  # source control points
        src = torch.matmul(self.tps_interp, ctrl)  # [B, HW, 2] in [0,1]
        grid = (src * 2.0 - 1.0).reshape(B, self.rect_hw[0], self.rect_hw[1], 2)
        return F.grid_sample(img, grid, mode="bilinear", padding_mode="border", align_corners=False)

    def _encode_rnn(self, seq):
        """2x BiLSTM (overridden by the HIP subclass)."""
        return self.rnn(seq)[0]

    def encode(self, x):
        x = self._run(self.stem, x)
        for u in self.resnet:
            sc = x if u.short is None else self._run(u.short, x)
            x = self._run(u.c2, self._run(u.c1, x), residual=sc)
        return x

    def forward(self, img_nchw: torch.Tensor) -> torch.Tensor:
        """[B,3,64,256] in [-1,1] -> forward logits [B, max_steps, num_classes] (all decoder steps)."""
        x = self.encode(self.rectify(img_nchw))  # [B,512,1,25]
        seq = x.squeeze(2).permute(0, 2, 1)
        enc = self._encode_rnn(seq)  # [B,25,512]
        return self._decode(enc)

    @staticmethod
    def decode_lengths(logits: torch.Tensor) -> torch.Tensor:
        """Number of time steps T_i the SavedModel's dynamic decode emits for sample i: it stops once the greedy symbol
        is EOS (the EOS step itself is emitted), else runs to the step limit.  This models the NETWORK (tf.contrib
        seq2seq dynamic_decode inside the ASTER graph); it is a guess like the rest of the stand-in (parity unpinned)."""
        S = logits.shape[1]
        is_eos = logits.argmax(dim=2) == EOS
        first = torch.where(is_eos.any(dim=1), is_eos.to(torch.int32).argmax(dim=1) + 1,
                            torch.full((logits.shape[0],), S, device=logits.device, dtype=torch.int64))
        return first

    def forward_logits(self, img_nchw: torch.Tensor):
        """batched form of the serving signature: (logits [B,S,C], lengths [B]); rows t >= lengths[i] do not exist in
        sample i's own ``forward_logits``."""
        logits = self.forward(img_nchw)
        return logits, self.decode_lengths(logits.detach())

    def serve(self, inputs_nhwc: torch.Tensor) -> dict:
        """The SavedModel's serving signature as the reference calls it (aster_inferer.py:31, batch 1):
        NHWC [1,64,256,3] -> {"forward_logits": [1, T, C]} (+ "backward_logits" when the network has the backward
        predictor) with the decode's own length T."""
        assert inputs_nhwc.shape[0] == 1, "the reference calls the SavedModel one sample at a time"
        if not self.has_backward_predictor:
            logits, lengths = self.forward_logits(inputs_nhwc.permute(0, 3, 1, 2))
            return {"forward_logits": logits[:, : int(lengths[0])]}
        x = self.encode(self.rectify(inputs_nhwc.permute(0, 3, 1, 2)))
        enc = self._encode_rnn(x.squeeze(2).permute(0, 2, 1))
        fwd = self._decode(enc)
        bwd = self._decode_with(enc.flip(1), self.bwd.emb, self.bwd.att_enc, self.bwd.att_dec, self.bwd.att_v, self.bwd.cell,
                                self.bwd.out)
        return {"forward_logits": fwd[:, : int(self.decode_lengths(fwd.detach())[0])],
                "backward_logits": bwd[:, : int(self.decode_lengths(bwd.detach())[0])]}

    def _decode(self, enc):
        """Bahdanau-attention LSTM decoder, greedy feedback (overridden by the HIP subclass)."""
        return self._decode_with(enc, self.emb, self.att_enc, self.att_dec, self.att_v, self.cell, self.out)

    def _decode_with(self, enc, emb, att_enc, att_dec, att_v, cell, out):
        B = enc.shape[0]
        enc_proj = att_enc(enc)
        h = enc.new_zeros(B, self.hidden)
        c = enc.new_zeros(B, self.hidden)
        prev = torch.full((B,), self.num_classes, dtype=torch.long, device=enc.device)  # GO
        outs = []
        for _ in range(self.max_steps):
            e = att_v(torch.tanh(enc_proj + att_dec(h)[:, None, :])).squeeze(2)
            a = torch.softmax(e, dim=1)
            ctx = torch.bmm(a[:, None, :], enc).squeeze(1)
            h, c = cell(torch.cat([ctx, emb(prev)], dim=1), (h, c))
            logit = out(h)
            outs.append(logit)
            prev = logit.argmax(dim=1)  # greedy feedback (non-differentiable, as in the TF decoder)
        return torch.stack(outs, dim=1)


class AsterLikeOCRHip(AsterLikeOCR):
    """Same network; every convolution (+folded BN, +residual, +ReLU) runs as ONE launch of the
    fp32-MFMA implicit-GEMM kernel (forward) and one for the data gradient (the net is frozen, so no
    weight gradients).  MIOpen served several of these shapes with its naive fallback kernels."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self._cache = {}

    def _invalidate(self):
        self._cache = {}

    def _apply(self, fn, *a, **kw):
        self._cache = {}
        return super()._apply(fn, *a, **kw)

    def load_state_dict(self, *a, **kw):
        r = super().load_state_dict(*a, **kw)
        self._cache = {}  # folded filters, packed filters, stacked LSTM / decoder constants derive from the weights
        return r

    def _encode_rnn(self, seq):
        """The frozen BiLSTM stack on batched GEMMs + one pointwise launch per step (ops.frozen_bilstm_layer) instead of
        MIOpen's per-direction per-step GEMM + pointwise pairs."""
        from . import ops
        if "rnn" not in self._cache:
            with torch.no_grad():
                layers = []
                for l in range(self.rnn.num_layers):
                    g = lambda n: torch.stack([getattr(self.rnn, f"{n}_l{l}"), getattr(self.rnn, f"{n}_l{l}_reverse")])
                    layers.append((g("weight_ih").contiguous(), g("weight_hh").contiguous(),
                                   (g("bias_ih") + g("bias_hh")).contiguous(), g("weight_hh").transpose(1, 2).contiguous()))
                self._cache["rnn"] = layers
        for w_ih, w_hh, b, w_hhT in self._cache["rnn"]:
            seq = ops.frozen_bilstm_layer(seq, w_ih, w_hh, b, w_hhT)
        return seq

    def _decode(self, enc):
        """The frozen decoder as one autograd node (ops.frozen_attn_decoder): attention context and LSTM-cell pointwise are
        single HIP launches per step; ~100 launches per forward+backward instead of ~350 torch kernels."""
        from . import ops
        if "dec" not in self._cache:
            with torch.no_grad():
                E = self.att_enc.weight.shape[1]
                w_ih, w_hh = self.cell.weight_ih, self.cell.weight_hh
                c = lambda t: t.detach().contiguous()
                self._cache["dec"] = ops.FrozenDecoderWeights(
                    w_enc=c(self.att_enc.weight), w_dT=c(self.att_dec.weight.t()), b_d=c(self.att_dec.bias),
                    w_d=c(self.att_dec.weight), v=c(self.att_v.weight.reshape(-1)),
                    etab=c(self.emb.weight @ w_ih[:, E:].t() + self.cell.bias_ih + self.cell.bias_hh),
                    w_ctx=c(w_ih[:, :E]), w_ctxT=c(w_ih[:, :E].t()), w_hh=c(w_hh), w_hhT=c(w_hh.t()),
                    w_o=c(self.out.weight), w_oT=c(self.out.weight.t()), b_o=c(self.out.bias),
                    w_cat=c(torch.cat([w_ih[:, :E], w_hh], dim=1)), w_catT=c(torch.cat([w_ih[:, :E], w_hh], dim=1).t()))
        return ops.frozen_attn_decoder(enc, self._cache["dec"], self.max_steps, self.num_classes)

    def encode(self, x):
        """stem + ResNet as one autograd node (ops.frozen_resnet): the ReLU gates and residual sums of the backward ride on the
        data-gradient launches' epilogues."""
        from . import ops
        if not x.is_cuda:
            raise RuntimeError("AsterLikeOCRHip runs on the GPU only (use AsterLikeOCR for the CPU definition)")
        key = ("resnet", ops.compute_mode())
        if key not in self._cache:
            with torch.no_grad():
                const = lambda blk: ops.FrozenConvConst(*blk.folded(), tuple(blk.conv.stride), tuple(blk.conv.padding), blk.relu)
                self._cache[key] = (const(self.stem), [(const(u.c1), const(u.c2), const(u.short) if u.short is not None else None)
                                                       for u in self.resnet])
        return ops.frozen_resnet(x, *self._cache[key])

    def _run(self, blk: _ConvBN, x, residual=None):
        from . import ops
        if not x.is_cuda:
            raise RuntimeError("AsterLikeOCRHip runs on the GPU only (use AsterLikeOCR for the CPU definition)")
        key = (id(blk), ops.compute_mode())  # the packed filters are dtype specific
        if key not in self._cache:
            with torch.no_grad():
                w, b = blk.folded()
                self._cache[key] = (w, b, ops.frozen_conv_packs(w, blk.conv.stride))  # packed once: the net is frozen
        w, b, packs = self._cache[key]
        return ops.frozen_conv(x, w, b, tuple(blk.conv.stride), tuple(blk.conv.padding), blk.relu, residual, packs)
