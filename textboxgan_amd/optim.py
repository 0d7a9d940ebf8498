"""Flat parameter storage, Keras-semantics Adam and the g_clone EMA on the HIP kernels.

* ``FlatParams`` re-homes module parameters into ONE contiguous fp32 buffer (16-byte aligned
  slices) so an optimiser update, an EMA update or a gradient all-reduce is one launch / one
  collective over a flat buffer (39.96 / 34.75 / 62.38 MB per step at full size).
* ``AdamTF`` = ``tf.keras.optimizers.Adam`` as configured in reference train.py:58-75,110-129
  (epsilon outside the bias-corrected sqrt; ``iterations`` is the step counter train.py:179 reads).
* ``ema_update`` = ``Generator.set_as_moving_average_of`` (generator.py:48-59).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn as nn

from . import ops
from .config import OptParams

ALIGN = 4  # floats (16 bytes): conv weight pointers must be 16-byte aligned


class FlatParams:
    def __init__(self, named_params: Sequence[Tuple[str, nn.Parameter]], device):
        self.names = [n for n, _ in named_params]
        self.params = [p for _, p in named_params]
        self.offsets: List[int] = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
        self.total = off
        self.flat = torch.zeros(self.total, device=device, dtype=torch.float32)
        for p, o in zip(self.params, self.offsets):
            view = self.flat[o: o + p.numel()].view(p.shape)
            view.copy_(p.data.to(device))
            p.data = view

    def range_of(self, prefixes: Tuple[str, ...]) -> Tuple[int, int]:
        """Contiguous [begin, end) of the parameters whose name starts with one of ``prefixes``."""
        idx = [i for i, n in enumerate(self.names) if n.startswith(prefixes)]
        assert idx == list(range(idx[0], idx[-1] + 1)), "parameter group is not contiguous"
        end = self.offsets[idx[-1] + 1] if idx[-1] + 1 < len(self.offsets) else self.total
        return self.offsets[idx[0]], end

    def select(self, prefixes: Tuple[str, ...]) -> List[nn.Parameter]:
        return [p for n, p in zip(self.names, self.params) if n.startswith(prefixes)]

    def make_grad_buffer(self, begin: int, end: int):
        """Flat gradient buffer for [begin, end) + per-parameter views into it."""
        buf = torch.zeros(end - begin, device=self.flat.device, dtype=torch.float32)
        views = GradViews(buf[o - begin: o - begin + p.numel()].view(p.shape)
                          for p, o in zip(self.params, self.offsets) if begin <= o < end)
        return buf, views


class GradViews(list):
    """Per-parameter views that tile one flat gradient buffer (FlatParams.make_grad_buffer).  Carries write_grads' copy
    plan as an attribute, so the plan lives and dies with the list it describes -- no module-level cache keyed by id()
    (an id can be recycled once its object is freed)."""
    plan = None


def write_grads(views: List[torch.Tensor], grads) -> None:
    """gradients -> the flat buffer slice the consecutive ``views`` tile (16-byte alignment gaps between them).  ONE
    ``torch.cat(..., out=slice)`` (a batched copy kernel over up to 128 tensors per launch) instead of one copy per
    parameter: ``_foreach_copy_`` lowered to ~320 separate D2D copies per step (profiles/r02_pmc_report.txt)."""
    if not views:
        return
    hit = getattr(views, "plan", None)
    if hit is None:
        base = views[0]._base if views[0]._base is not None else views[0]
        starts = [v.storage_offset() for v in views]
        ends = [o + v.numel() for o, v in zip(starts, views)]
        gaps = [(starts[i + 1] - ends[i]) if i + 1 < len(views) else 0 for i in range(len(views))]
        assert all(g >= 0 for g in gaps), "views must be consecutive slices of one flat buffer"
        dev = views[0].device
        pads = {g: torch.zeros(g, device=dev) for g in set(gaps) if g > 0}
        zeros = {}
        hit = (None, base.view(-1)[starts[0]:ends[-1]], gaps, pads, zeros)
        if isinstance(views, GradViews):
            views.plan = hit
    _, out, gaps, pads, zeros = hit
    pieces = []
    for i, (v, g) in enumerate(zip(views, grads)):
        if g is None:
            z = zeros.get(v.numel())
            if z is None:
                z = zeros[v.numel()] = torch.zeros(v.numel(), device=v.device)
            pieces.append(z)
        else:
            pieces.append(g.reshape(-1))
        if gaps[i]:
            pieces.append(pads[gaps[i]])
    torch.cat(pieces, out=out)


class AdamTF:
    def __init__(self, theta_flat: torch.Tensor, opt: OptParams):
        self.theta = theta_flat
        self.m = torch.zeros_like(theta_flat)
        self.v = torch.zeros_like(theta_flat)
        self.step = torch.zeros(1, dtype=torch.int64, device=theta_flat.device)
        self.lr, self.beta1, self.beta2, self.eps = opt.learning_rate, opt.beta1, opt.beta2, opt.epsilon
        self._iterations = 0

    @property
    def iterations(self) -> int:
        return self._iterations

    def apply_gradients(self, g_flat: torch.Tensor) -> None:
        ops.adam_tf_(self.theta, self.m, self.v, g_flat, self.step, self.lr, self.beta1, self.beta2, self.eps)
        self.step += 1  # device counter (part of the captured graph); the host mirror is bumped by the caller

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return dict(m=self.m, v=self.v, step=self.step)

    def load_state_dict(self, sd) -> None:
        self.m.copy_(sd["m"]); self.v.copy_(sd["v"]); self.step.copy_(sd["step"])
        self._iterations = int(self.step.item())


@torch.no_grad()
def ema_update(clone: nn.Module, src: nn.Module, beta: float = 0.99) -> None:
    fc, fs = getattr(clone, "_flat", None), getattr(src, "_flat", None)
    if fc is not None and fs is not None and fc.total == fs.total:
        ops.ema_lerp_(fc.flat, fs.flat, beta)
    else:
        for pc, ps in zip(clone.parameters(), src.parameters()):
            pc.copy_(ps + (pc - ps) * beta)
    for (nc, bc), (_, bs) in zip(clone.named_buffers(), src.named_buffers()):
        if "w_avg" in nc:
            bc.copy_(bs)  # beta_nontrainable = 0
        else:
            bc.copy_(bs + (bc - bs) * beta)


G_ORDER = ("latent_encoder.", "synthesis.", "word_encoder.")


def flatten_generator(G: nn.Module, device) -> FlatParams:
    """[latent_encoder | synthesis | word_encoder]: g_optimizer owns the first two groups,
    ocr_optimizer the last two (training_step.py:194-206) -- both ranges are contiguous."""
    named = dict(G.named_parameters())
    ordered = [(n, p) for pre in G_ORDER for n, p in named.items() if n.startswith(pre)]
    assert len(ordered) == len(named)
    G.to(device)
    fp = FlatParams(ordered, device)
    G._flat = fp
    return fp


def flatten_module(Mod: nn.Module, device) -> FlatParams:
    Mod.to(device)
    fp = FlatParams(list(Mod.named_parameters()), device)
    Mod._flat = fp
    return fp
