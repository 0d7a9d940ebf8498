"""Operators of the TextBoxGAN step on top of libtbg_hip.so.

Two layers:

* ``*_raw`` helpers: allocate outputs with torch's caching allocator and enqueue ONE C-ABI call
  on the current stream.  No autograd.
* ``torch.autograd.Function`` wrappers, in two families
    - composable primitives (``upfirdn2d``, ``conv2d``, ``conv2d_bwd_data``, ``conv2d_bwd_weight``)
      whose backward is written with the same primitives, so gradients of ANY order exist -- the
      property the reference gets from its recursive ``tf.custom_gradient``
      (upfirdn_2d_v2.py:204-246) and needs for R1 / path-length (training_step.py:300-373);
    - fused first-order layers (modulated conv + noise + bias + lrelu, up-conv + FIR + epilogue,
      conv + bias + lrelu, toRGB, ...) used on every non-regularised pass.

Weights stay in the reference's parameter layout ([k,k,I,O], modulated_conv2d.py:57-64).
"""
from __future__ import annotations

import ctypes as C
import math
import threading
from typing import NamedTuple, Optional, Tuple

import numpy as np
import torch

from . import native as N
from .native import ACT_LINEAR, ACT_LRELU, SQRT2


class _Flags:
    """Set by the training step around each of its three backward passes so layers skip
    gradients nobody consumes (torch gives custom Functions no per-call pruning info).  The flags act only on nodes that
    carry the matching ROLE tag (set by the discriminator's layers when they call conv_bias_act_fused): a node without a
    tag is never pruned, whatever its shape or owner."""

    def __init__(self):
        self.skip_d_wgrad = False     # G-loss pass only needs dL/d(image) through the discriminator  (role "d" / "d_image")
        self.skip_image_grad = False  # D-loss pass does not need dL/d(image)                          (role "d_image")
        # The d-step runs the discriminator ONCE over [fake; real] (2B samples).  In the G-loss pass only the first B
        # samples carry a gradient (the score gradient of the real half is structurally zero): nodes with a "d" role then
        # run their backward kernels on the leading d_first_half samples only and leave the rest of the returned tensor
        # unwritten -- nothing reads it (the concatenation's backward hands the generator the first half).  0 = off.
        self.d_first_half = 0
        # The regularisers' inner gradient (d(image . noise)/d(latents) for path length, d(scores)/d(image) for R1)
        # reaches no filter: custom Functions are not pruned by autograd.grad(inputs=...), so without this flag every
        # composable conv of that pass would also launch its filter gradient and throw it away.
        self.no_filter_grads = False
        # debug aid: fill value for the never-read tail of the d_first_half gradient tensors -- None leaves
        # it uninitialised (production), 0.0 gives anomaly mode / NaN checks defined memory, NaN makes any read of it loud
        self.unread_tail_fill = None


class _State:
    """Every piece of host-side mutable state of this module: the pruning flags, the arithmetic mode and the packed-filter
    scopes.  It is PROCESS-wide on purpose -- torch runs a backward pass's Python callbacks on its per-device autograd
    thread, not on the thread that called autograd.grad, so thread-local storage would hide the flags from exactly the
    nodes that read them -- and is guarded by STATE_LOCK instead: a scope that changes it (TrainingStep._compute_grads,
    the projector, validation passes) holds the re-entrant lock from before its forward until after its last backward,
    so two step objects driven from two threads take turns instead of corrupting each other."""

    def __init__(self):
        self.flags = _Flags()
        self.compute = "f32"
        self.pack_step = None    # live only inside filter_cache(): weights are constant within one step's passes
        self.pack_store = None   # live only inside PackedStore.scope(): persistent packs of a training step


_TLS = _State()
STATE_LOCK = threading.RLock()


class _FlagsProxy:
    """``ops.FLAGS.x`` reads / writes the process-wide flags (hold STATE_LOCK around a pass that sets them)."""

    def __getattr__(self, name):
        return getattr(_TLS.flags, name)

    def __setattr__(self, name, value):
        if not hasattr(_TLS.flags, name):
            raise AttributeError(name)
        setattr(_TLS.flags, name, value)


FLAGS = _FlagsProxy()

class _Tuning:
    """Every host-side tuning constant and measurement (A/B) switch of the package in ONE documented object (VERDICT round 4,
    item 9: they were module-level globals scattered over ops.py / ops2.py).  Production never changes them; tools/ab_step.py,
    tools/bench_*.py and a few tests set attributes of ``ops.TUNING`` (process-wide, like FLAGS: hold STATE_LOCK around a pass that
    depends on a changed value).  The C library has no such state: what a launch does is a function of its arguments."""

    def __init__(self):
        # ---- unit tensors (tbg.h "UNIT TENSORS")
        self.use_units = True        # 3x3 stride-1 layers consume unit tensors (tbg_conv2d_units / tbg_conv2d_wgrad_units) where they apply
        self.units_min_blocks = 200  # tbg_conv2d_units runs ONE 512-thread block per CU: launches of fewer blocks keep the NCHW
                                     # kernel (profiles/r04_units_isolated.txt: 128 blocks 136 vs 145 TFLOP/s, 256 blocks 196 vs 154)
        self.use_units_s2 = True     # blur + 3x3 stride-2 layers (and the up-convolution's backward) through PHASE unit tensors where
                                     # both the strided convolution and its filter gradient take them
        self.use_units_t2 = True     # 3x3 stride-2 TRANSPOSED convolutions (up-conv forward, data gradient of the strided layers)
        self.units_min_blocks_t2 = 96    # the same gate for tbg_conv2d_units_t2 in f32x3 (its NCHW alternative -- four output-parity
                                     # classes of mostly-empty tiles, 38-80 TFLOP/s -- loses earlier than the stride-1 kernel's): -0.10 ms
                                     # per step.  bf16 keeps units_min_blocks: there the gain is 0.07 ms, and the full-width step's
                                     # noisiest scalar gradient (a noise strength: one heavily cancelling sum) moved from 0.22 to 0.27
                                     # relative against its 0.25 bar when more layers changed kernels (profiles/r05_ab_one_box.txt)
        self.save_units = True       # keep units(x * s) of a layer input alive from forward to backward for the filter gradient
                                     # (beside the fp32 x: +6 B / element in f32x3, +2 B in bf16); False = drop it and pack again in
                                     # the backward pass (one tbg_units_pack_f32 launch per layer) when activation memory matters.
                                     # Peak memory of a step is in bench.py's line (`peak_hbm_gb`)
        self.unit_sinks = True       # round 5: producers (conv / FIR / split-K epilogues) write the NEXT layer's unit tensor themselves
                                     # (tbg_epilogue.units_out); False = the stand-alone tbg_units_pack_f32 pass of round 4
        # ---- small maps (tbg.h "SMALL MAPS", csrc/conv_small.hip)
        self.use_small = True        # small-map convolutions take tbg_conv2d_units_small (K split inside the block, one launch) instead
                                     # of the NCHW kernel's split-K pair (convolution into HBM slabs + tbg_slab_epilogue_f32)
        self.small_max_blocks_taps = 1280  # the same bound for the tap-list form of the stride-2 transposed k x k layers (their NCHW
                                     # alternative -- four class launches of mostly-empty tiles -- loses later: 640 -> 1280 = 17.81 -> 17.73 ms)
        self.small_max_blocks = 512  # ... when the launch is at most this many blocks (two rounds of one block per CU): beyond, every
                                     # pixel tile re-reads its filter slice too often and the 128 x 128 NCHW tiles win
        # ---- split-K of the small-map launches
        self.force_ksplit = None     # experiment knob (tools/bench_ksplit.py)
        self.ksplit_target_blocks = 448  # blocks a split-K launch aims for (A/B on one box: 288 -> 21.22, 448 -> 21.11, 512 -> 20.97
                                     # vs 448 -> 20.94, 640 -> 21.02, 900 -> 21.04, 200 -> 21.50 ms per step)
        self.one_per_cu_split = True  # two K splits for launches of exactly one tile per CU (profiles/r03_cold_conv.txt)
        self.force_variant = 0       # experiment knob (tools/bench_variants_conv.py): tbg_conv2d_*_variant's instantiation family
        # ---- graph structure
        self.fold_res_scale = True   # DiscriminatorBlock folds its 1/sqrt(2) into both branches (fp32-grade arithmetics only)
        self.fuse_skip_grad = True   # DiscriminatorBlock: conv_0 + skip FIR as ONE node (False = two nodes + the engine's add)
        self.use_fused2 = True       # path-length pass on the twice-differentiable node pairs of ops2 (False = composable primitives)
        # ---- recurrent layers of the frozen recogniser
        self.fused_decoder = True    # two launches per decoder step and direction of the pass (tbg.h: cell over [context; hidden] + the
                                     # per-sample half) instead of 8 / 6 library-GEMM, attention, cell and index launches
        self.fused_lstm = True       # one launch per BiLSTM step (tbg_lstm_fused_*: projection + cell) instead of a library GEMM + a
                                     # pointwise launch
        # ---- dense layers
        self.dense_small_k = 768     # above: a library GEMM; below: launch-bound, one hand-written launch per direction


    def __setattr__(self, name, value):
        # a misspelt knob (tools/ab_*.py set attributes from the command line) must not silently create a new one
        if getattr(self, "_sealed", False) and not hasattr(self, name):
            raise AttributeError(f"ops.TUNING has no knob {name!r}")
        object.__setattr__(self, name, value)


TUNING = _Tuning()
object.__setattr__(TUNING, "_sealed", True)


# ----------------------------------------------------------------------------------------
# arithmetic of the MFMA contractions:
#   "f32"   v_mfma_f32_32x32x2_f32, exact fp32 products
#   "bf16"  operands rounded to bf16 while staging, fp32 accumulate (BASELINE configs[2])
#   "f32x3" fp32 operands split into three bf16 terms, six partial products on the bf16 pipe, fp32 accumulate: fp32-grade
#           results (tbg.h "f32x3 forms")
# HBM tensors, epilogues, master weights and Adam stay fp32 in every mode.
# ----------------------------------------------------------------------------------------
COMPUTE_MODES = ("f32", "bf16", "f32x3")
FMT_F32, FMT_BF16, FMT_X3 = 0, 1, 2
_FMT = {"f32": FMT_F32, "bf16": FMT_BF16, "f32x3": FMT_X3}


class compute_dtype:
    """``with ops.compute_dtype("bf16"): ...`` -- scope in which conv / filter-gradient launches use that arithmetic."""

    def __init__(self, dtype: str):
        assert dtype in COMPUTE_MODES, dtype
        self.mode = dtype

    def __enter__(self):
        self._prev, _TLS.compute = _TLS.compute, self.mode
        return self

    def __exit__(self, *exc):
        _TLS.compute = self._prev
        return False


def compute_mode() -> str:
    return _TLS.compute


def is_bf16() -> bool:
    return _TLS.compute == "bf16"


class _Profile:
    """bench.py's roofline pass: bracket kernel launches with HIP events on the stream the kernel is enqueued on and
    attribute their ALGORITHMIC work to the kernel that ran: FLOPs (2 x MACs of the dense contraction) for the MFMA
    convolutions, bytes (every operand read once + every result written once) for the HBM-bound kernels.  For the
    convolutions the name is the instantiation the descriptor selects (rocprofv3 spelling).  Off by default."""

    def __init__(self):
        self.on = False
        self.by_shape = False
        self.recs = []

    def enable(self):
        self.on, self.recs = True, []

    def disable(self):
        self.on = False

    def launch(self, name, flops, fn, shape=None, nbytes=0.0):
        if not self.on:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn()
        e1.record()
        if callable(name):  # resolved lazily: the instantiation the descriptor selects
            name = name()
        if self.by_shape and shape is not None:
            name = name + " " + shape
        self.recs.append((name, flops, nbytes, e0, e1))
        return rc

    def collect(self):
        torch.cuda.synchronize()
        out = {}
        for name, flops, nbytes, e0, e1 in self.recs:
            r = out.setdefault(name, dict(n=0, ms=0.0, flops=0.0, bytes=0.0))
            r["n"] += 1
            r["ms"] += e0.elapsed_time(e1)
            r["flops"] += flops
            r["bytes"] += nbytes
        self.recs = []
        return out


PROFILE = _Profile()


# ----------------------------------------------------------------------------------------
# FIR filters (upfirdn_2d_v2.py:18-25) cached per device
# ----------------------------------------------------------------------------------------
_FIR_CACHE = {}
_FIR_SEP = {}  # data_ptr of a cached 2-D filter -> (k2d, kx, ky): its 1-D factors, k2d == outer(ky, kx)


def fir_kernel(device, gain: float = 1.0, taps=(1, 3, 3, 1)) -> torch.Tensor:
    """_setup_kernel (upfirdn_2d_v2.py:18-25): k = outer(taps, taps) / sum * gain.  The 2-D tensor is the handle the
    model passes around; its separable factors (kx = taps/sum, ky = gain*taps/sum) are kept beside it so the launches
    can take the separable kernel (tbg_upfirdn2d_sep_f32)."""
    key = (str(device), float(gain), tuple(taps))
    if key not in _FIR_CACHE:
        t = np.asarray(taps, dtype=np.float32)
        k = np.outer(t, t)
        k = k / k.sum() * gain
        k2 = torch.from_numpy(k).to(device)
        _FIR_CACHE[key] = k2
        if k2.is_cuda:
            t64 = np.asarray(taps, dtype=np.float64)
            kx = torch.from_numpy((t64 / t64.sum()).astype(np.float32)).to(device)
            ky = torch.from_numpy((t64 / t64.sum() * gain).astype(np.float32)).to(device)
            _FIR_SEP[k2.data_ptr()] = (k2, kx, ky)
    return _FIR_CACHE[key]


def _sep_factors(k: torch.Tensor):
    hit = _FIR_SEP.get(k.data_ptr())  # the registry keeps hit[0] alive, so its address cannot be recycled
    return (hit[1], hit[2]) if hit is not None and hit[0].shape == k.shape and k.is_contiguous() else None


# ----------------------------------------------------------------------------------------
# raw launches
# ----------------------------------------------------------------------------------------
def upfirdn2d_raw(x: torch.Tensor, k: torch.Tensor, up=(1, 1), down=(1, 1), pad=(0, 0, 0, 0),
                  in_scale: Optional[torch.Tensor] = None, epi: Optional[N.Epilogue] = None,
                  out: Optional[torch.Tensor] = None, sink: Optional["UnitSink"] = None):
    """x NCHW (treated as [B*C, H, W, 1], upfirdn_2d_v2.py:166-183); up/down = (x, y);
    pad = (x0, x1, y0, y1).  out: a contiguous [B, C, outH, outW] tensor to write (e.g. a leading-batch view).
    sink (the model's separable blur with an epilogue only): returns (y, UnitTensor | None) -- units(y * sink.scale) written by the
    same launch (fir_units_kernel)."""
    B, Cc, H, W = x.shape
    kH, kW = k.shape
    outW = (W * up[0] + pad[0] + pad[1] - kW + down[0]) // down[0]
    outH = (H * up[1] + pad[2] + pad[3] - kH + down[1]) // down[1]
    y = torch.empty((B, Cc, outH, outW), device=x.device, dtype=torch.float32) if out is None else out
    assert tuple(y.shape) == (B, Cc, outH, outW)
    sep = _sep_factors(k)
    if sink is not None:
        U = None
        if sep is not None and epi is not None and tuple(up) == (1, 1) and tuple(down) == (1, 1):
            epi, U = _sink_epi(epi, sink, B, Cc, outH, outW, x.device)
        if U is None:
            return upfirdn2d_raw(x, k, up, down, pad, in_scale, epi, out), None
        _nb = 4.0 * (x.numel() + y.numel() + (B * outH * outW if epi.noise else 0)) + 2.0 * U.data.numel()
        N.check(PROFILE.launch(f"fir_units_kernel<{U.planes}>", 0.0, lambda: N.lib().tbg_upfirdn2d_sep_f32(
            N.ptr(x), N.ptr(sep[0]), N.ptr(sep[1]), N.ptr(y), B * Cc, H, W, kH, kW, 1, 1, 1, 1, pad[0], pad[1], pad[2], pad[3],
            N.ptr(in_scale), Cc, C.byref(epi), N.stream()), nbytes=_nb), "tbg_upfirdn2d_sep (unit sink)")
        return y, U
    if sep is not None:  # the model's filters: separable passes
        _nb = 4.0 * (x.numel() + y.numel() + (B * outH * outW if epi is not None and epi.noise else 0))
        rc = PROFILE.launch(f"upfirdn2d_tile_kernel up{up} down{down}", 0.0, lambda: N.lib().tbg_upfirdn2d_sep_f32(
            N.ptr(x), N.ptr(sep[0]), N.ptr(sep[1]), N.ptr(y), B * Cc, H, W, kH, kW, up[0], up[1], down[0], down[1], pad[0],
            pad[1], pad[2], pad[3], N.ptr(in_scale), Cc, C.byref(epi) if epi is not None else None, N.stream()), nbytes=_nb)
    elif in_scale is None and epi is None:
        rc = N.lib().tbg_upfirdn2d_f32(N.ptr(x), N.ptr(k), N.ptr(y), B * Cc, H, W, 1, kH, kW, up[0], up[1], down[0],
                                       down[1], pad[0], pad[1], pad[2], pad[3], N.stream())
    else:
        rc = N.lib().tbg_upfirdn2d_ex_f32(N.ptr(x), N.ptr(k), N.ptr(y), B * Cc, H, W, kH, kW, up[0], up[1], down[0],
                                          down[1], pad[0], pad[1], pad[2], pad[3], N.ptr(in_scale), Cc,
                                          C.byref(epi) if epi is not None else None, N.stream())
    N.check(rc, "tbg_upfirdn2d")
    return y




def conv2d_raw(x: torch.Tensor, w: torch.Tensor, M: int, KH: int, KW: int, out_hw: Tuple[int, int], stride=(1, 1),
               pad=(0, 0), transposed=False, flip=False, in_scale=None, epi: Optional[N.Epilogue] = None,
               ldw: Optional[int] = None, allow_split=True, dot=None, out: Optional[torch.Tensor] = None,
               sink: Optional["UnitSink"] = None, allow_small=True):
    """w: a PackedFilter (pack_filter), or a tensor in GEMM layout [KH*KW, C, ldw] (any view with that memory layout,
    e.g. the HWIO parameter) which is packed here.
    dot = (aux, out): out[b,m] = sum_p (alpha*acc)[b,m,p] * aux[b,m,p]  (fused when K is not split).  dot = (aux, None): the
    launch's partial sums are returned as they are -- (y, partial [B, M, slots]) -- for a consumer that sums the slots itself
    (tbg_modconv_bwd_smalls_f32: one reduction launch less per layer and pass).
    sink: returns (y, UnitTensor | None) -- units(y * sink.scale) written by the launch that writes y (the convolution's epilogue,
    or the split-K pass's second half)."""
    if sink is not None:
        assert dot is None and out is None
        epi = N.epilogue() if epi is None else epi
        epi_s, U = _sink_epi(epi, sink, x.shape[0], M, out_hw[0], out_hw[1], x.device)
        y = conv2d_raw(x, w, M, KH, KW, out_hw, stride, pad, transposed, flip, in_scale, epi_s, ldw, allow_split, allow_small=allow_small)
        return y, U
    B, Cc, Hin, Win = x.shape
    if not isinstance(w, PackedFilter):
        ldw = M if ldw is None else ldw
        w = pack_filter(w.reshape(KH * KW, Cc, ldw), transpose=False, flip=False)
    assert w.C == Cc and w.M >= M and w.T == KH * KW, "packed filter does not match the convolution"
    ldw = w.M
    fmt = w.fmt
    bf16 = fmt != FMT_F32  # a 16-bit operand pipe (bf16 or f32x3): shares the merged-class rule of the library
    _conv = (N.lib().tbg_conv2d_f32, N.lib().tbg_conv2d_bf16, N.lib().tbg_conv2d_x3)[fmt]
    if TUNING.force_variant:
        _v, _fn = TUNING.force_variant, (N.lib().tbg_conv2d_f32_variant, N.lib().tbg_conv2d_bf16_variant,
                                  N.lib().tbg_conv2d_x3_variant)[fmt]
        _conv = lambda d_, x_, w_, y_, s_, e_, st_: _fn(d_, x_, w_, y_, s_, e_, _v, st_)
    Hout, Wout = out_hw
    if fmt != FMT_F32 and TUNING.use_small and allow_small and allow_split and TUNING.force_variant == 0 and TUNING.force_ksplit is None:
        # small map: tbg_conv2d_units_small (K split inside the block, the real epilogue in the same launch) instead of this entry's
        # slabs + second half, reading units(x * in_scale) -- attached by the launch that produced x, or packed here
        d0 = N.ConvDesc(B, Cc, M, Hin, Win, Hout, Wout, KH, KW, stride[0], stride[1], pad[0], pad[1], int(transposed), int(flip), ldw, 1)
        nb = N.lib().tbg_conv2d_units_small_blocks(C.byref(d0), unit_planes(fmt))
        taps = not (KH == 3 and KW == 3 and tuple(stride) == (1, 1) and not transposed) and KH * KW > 1
        # (tap-list forms of k x k layers: the transposed ones win 1.5 - 3x over this entry's output-parity classes; the strided
        # VALID ones -- few output tiles under a long reduction -- need the split over BLOCKS and stay here: 17 vs 53 us on
        # 5x17 -> 2x8 512 -> 128, profiles/r06_j_bench_taps_x3.txt)
        limit = (TUNING.small_max_blocks_taps if transposed else 0) if taps else TUNING.small_max_blocks
        if 0 < nb <= limit and (
                dot is None or N.lib().tbg_conv2d_units_small_dot_slots(C.byref(d0), unit_planes(fmt)) > 0):
            XU = take_units(x, in_scale, unit_planes(fmt)) if x.is_contiguous() else None
            XU = units_pack(x.contiguous(), in_scale, unit_planes(fmt)) if XU is None else XU
            return conv2d_small_raw(XU, w, M, KH, out_hw, stride, transposed, flip, epi, dot, out, KW=KW, pad=pad)
    w = w.data
    nchunks = math.ceil(Cc / (16 if fmt == FMT_BF16 else 8))
    ksplit = 1
    if allow_split:
        # the tile choice belongs to the library: ask it how many blocks this descriptor launches unsplit (all output-parity
        # classes of a transposed launch included) instead of mirroring its rules here
        d1 = N.ConvDesc(B, Cc, M, Hin, Win, Hout, Wout, KH, KW, stride[0], stride[1], pad[0], pad[1], int(transposed),
                        int(flip), ldw, 1)
        tiles = N.lib().tbg_conv2d_blocks(C.byref(d1), int(in_scale is not None), int(fmt))
        N.check(min(tiles, 0), "tbg_conv2d_blocks")
        # exactly one block per CU (256 tiles) leaves the second block slot of a stride-1 tile empty -- nothing overlaps its
        # staging: two splits there measured 165 vs 126-138 TFLOP/s (f32x3 16x64 256->256, profiles/r03_cold_conv.txt); a
        # stride-2 f32x3 tile fills the CU's LDS alone and only loses to the slab pass (109 vs 132)
        one_per_cu = (TUNING.one_per_cu_split and tiles == 256 and tuple(stride) == (1, 1) and not transposed and
                      dot is None)  # (a fused dot needs K whole)
        if (tiles < 256 or one_per_cu) and nchunks >= 8:  # tools/bench_ksplit.py: ~300 blocks is the sweet spot (slab traffic ~ ksplit)
            target = max(1, min(nchunks // 4, math.ceil(TUNING.ksplit_target_blocks / tiles)))
            # a split that divides the chunk count keeps the splits even (32 chunks: 6 splits = 6,6,6,6,6,2 ran slower than 4)
            divs = [k for k in range(1, nchunks // 2 + 1) if nchunks % k == 0]
            ksplit = min(divs, key=lambda k: abs(math.log(k / target)))
        if TUNING.force_ksplit is not None:
            ksplit = max(1, min(nchunks, TUNING.force_ksplit))
    d = N.ConvDesc(B, Cc, M, Hin, Win, Hout, Wout, KH, KW, stride[0], stride[1], pad[0], pad[1], int(transposed),
                   int(flip), ldw, ksplit)
    if epi is None:
        epi = N.epilogue()
    _what = (f"tbg_conv2d[B={B} C={Cc} M={M} in={Hin}x{Win} out={Hout}x{Wout} k={KH}x{KW} s={tuple(stride)} "
             f"p={tuple(pad)} T={int(transposed)} ldw={ldw} ksplit={ksplit}]")
    _flops = 2.0 * B * M * Cc * KH * KW * (Hin * Win if transposed else Hout * Wout)
    _bytes = 4.0 * (B * Cc * Hin * Win + B * M * Hout * Wout * ksplit + KH * KW * Cc * M)  # x + y (slabs) + filter, once each
    _kname = lambda: N.conv_kernel_name(d, in_scale is not None, fmt)
    dot_slots = 0
    if dot is not None and ksplit == 1:
        dot_slots = N.lib().tbg_conv2d_dot_slots(C.byref(d), int(in_scale is not None), int(fmt))
        N.check(min(dot_slots, 0), "tbg_conv2d_dot_slots")
    if ksplit > 1 or (dot is not None and dot_slots == 0):
        # split-K: every split stores alpha*acc into its own slab (no zero-fill, no atomics); one flat pass sums the
        # slabs and applies the real epilogue.  (Also the route of a fused dot the tiling cannot serve -- several small
        # images per tile: one "slab", the dot from the finished accumulators.)
        slabs = torch.empty((ksplit, B, M, Hout, Wout), device=x.device, dtype=torch.float32)
        e0 = N.epilogue(alpha=epi.alpha)  # (store-only: a unit sink belongs to the second half)
        N.check(PROFILE.launch(_kname, _flops, lambda: _conv(
            C.byref(d), N.ptr(x), N.ptr(w), N.ptr(slabs), N.ptr(in_scale), C.byref(e0), N.stream()), _what, _bytes), _what)
        e1 = N.Epilogue.from_buffer_copy(epi)
        e1.alpha = 1.0
        dpart = None
        if dot is not None:  # the fused dot needs the complete sum: reduce first (the smallest G layers' backward)
            e1.dot_aux, e1.dot_out = None, None
            if ksplit == 1:
                tmp = slabs[0]
            else:
                tmp = torch.empty((B, M, Hout, Wout), device=x.device, dtype=torch.float32)
                N.check(N.lib().tbg_slab_epilogue_f32(N.ptr(slabs), N.ptr(tmp), B, M, Hout * Wout, ksplit,
                                                      C.byref(N.epilogue()), N.stream()), "tbg_slab_epilogue")
            # sum_p tmp * aux per (b, m) as ONE launch: the per-plane sums of tbg_bias_act_bwd_f32 (dpre = tmp, y_rec = aux)
            _, _, _, _, dpart = bias_act_bwd_raw(tmp, dot[0], N.epilogue(), want_dpre=False, want_db=False, want_dyy=True)
            if dot[1] is not None:
                torch.sum(dpart, dim=2, out=dot[1].view(B, M))
            slabs, nslab = tmp, 1
        else:
            nslab = ksplit
        y = torch.empty((B, M, Hout, Wout), device=x.device, dtype=torch.float32) if out is None else out
        if e1.units_out:  # a unit sink rides on the split's second half
            N.check(PROFILE.launch(f"slab_epilogue_units_kernel<{e1.units_planes}>", 0.0, lambda: N.lib().tbg_slab_epilogue_units_f32(
                N.ptr(slabs), N.ptr(y), B, M, Hout, Wout, nslab, C.byref(e1), N.stream()),
                nbytes=4.0 * (nslab + 1) * y.numel()), "tbg_slab_epilogue_units")
            return y
        N.check(N.lib().tbg_slab_epilogue_f32(N.ptr(slabs), N.ptr(y), B, M, Hout * Wout, nslab, C.byref(e1), N.stream()),
                "tbg_slab_epilogue")
        return (y, dpart) if (dot is not None and dot[1] is None) else y
    partial = None
    if dot is not None:  # per-(tile, wave column) partial sums, plain stores: summed here in a fixed order (deterministic)
        partial = torch.empty((B, M, dot_slots), device=x.device, dtype=torch.float32)
        epi = N.Epilogue.from_buffer_copy(epi)
        epi.dot_aux, epi.dot_out = N.ptr(dot[0]), N.ptr(partial)
    y = torch.empty((B, M, Hout, Wout), device=x.device, dtype=torch.float32) if out is None else out
    N.check(PROFILE.launch(_kname, _flops, lambda: _conv(
        C.byref(d), N.ptr(x), N.ptr(w), N.ptr(y), N.ptr(in_scale), C.byref(epi), N.stream()), _what, _bytes), _what)
    if partial is not None:
        if dot[1] is None:
            return y, partial
        torch.sum(partial, dim=2, out=dot[1].view(B, M))
    return y


def _workspace(device, nbytes: int) -> torch.Tensor:
    """Stream-ordered temporary from the caching allocator (the role of TF's allocate_temp): consumed by the two launches
    of one filter-gradient call.  Deliberately NOT cached across calls -- a module-level grow-only buffer first allocated
    during a HIP-graph capture would live in that graph's private pool and then be shared with other captures and with
    eager launches."""
    return torch.empty((max(nbytes, 16) + 3) // 4, device=device, dtype=torch.float32)


def _bias_rider(d: N.WgradDesc, bias):
    """bias = (parts [Bp, CS, nch], db [CS]): the bias gradient db = sum of the bias_act backward's partial sums rides on the filter
    gradient's reduce launch (tbg_wgrad_desc.bias_*) instead of being a reduction launch of its own."""
    if bias is not None:
        parts, db = bias
        assert parts.dim() == 3 and parts.shape[1] == d.CS and db.numel() == d.CS and parts.is_contiguous()
        d.bias_parts, d.bias_grad, d.bias_B, d.bias_nch = N.ptr(parts), N.ptr(db), parts.shape[0], parts.shape[2]
    return d


def wgrad_raw(S: torch.Tensor, L: torch.Tensor, KH: int, KW: int, stride, pad, out: torch.Tensor, st_t: int, st_l: int,
              st_s: int, alpha: float, s_scale=None, l_scale=None, out_offset: int = 0, add=None, bias=None):
    """Overwrites ``out`` (every in-range element written exactly once).
    add = (addw, addq, gamma): out += gamma * addw (laid out like out) * addq[CL x CS plane].  bias: see _bias_rider."""
    B, CS, Hs, Ws = S.shape
    _, CL, Hl, Wl = L.shape
    d = _bias_rider(N.WgradDesc(B, CS, CL, Hs, Ws, Hl, Wl, KH, KW, stride[0], stride[1], pad[0], pad[1], st_t, st_l, st_s, alpha), bias)
    nbytes = N.lib().tbg_conv2d_wgrad_workspace_bytes(C.byref(d))
    if nbytes < 0:
        raise N.TbgError("tbg_conv2d_wgrad: unsupported geometry")
    ws = _workspace(S.device, nbytes)
    _flops = 2.0 * B * CS * CL * Hs * Ws * KH * KW
    fmt = _FMT[_TLS.compute]
    if fmt == FMT_X3 and not HAVE_WGRAD_X3:
        fmt = FMT_F32
    _kname = lambda: N.wgrad_kernel_name(d, fmt)
    _wg = (N.lib().tbg_conv2d_wgrad_ex_f32, N.lib().tbg_conv2d_wgrad_bf16,
           getattr(N.lib(), "tbg_conv2d_wgrad_x3", None))[fmt]
    addw, addq, gamma = add if add is not None else (None, None, 0.0)
    N.check(PROFILE.launch(_kname, _flops, lambda: _wg(
        C.byref(d), N.ptr(S), N.ptr(L), N.ptr(out) + 4 * out_offset, N.ptr(s_scale), N.ptr(l_scale),
        (N.ptr(addw) + 4 * out_offset) if addw is not None else None, N.ptr(addq), gamma, N.ptr(ws),
        ws.numel() * 4, N.stream()), f"wgrad[B={B} CS={CS} CL={CL} S={Hs}x{Ws} L={Hl}x{Wl} k={KH} s={tuple(stride)}]",
        4.0 * (S.numel() + L.numel() + KH * KW * CS * CL)), "tbg_conv2d_wgrad")
    return out


class UnitTensor(NamedTuple):
    """an activation in the 8-channel-unit layout (tbg.h "unit tensors"): U[planes][B][ceil(C/8)][H+2][W+2][8] bf16 -- the form
    the matrix-core kernels DMA straight into LDS.  planes = 3: f32x3 terms (hi | mid | lo), 1: bf16."""
    data: torch.Tensor  # flat bf16
    B: int
    C: int
    H: int
    W: int
    planes: int


def unit_planes(fmt=None) -> int:
    fmt = _FMT[_TLS.compute] if fmt is None else fmt
    return {FMT_BF16: 1, FMT_X3: 3}[fmt]


def units_alloc(B, Cc, H, W, planes, device) -> UnitTensor:
    """an uninitialised unit tensor of the activation geometry [B, Cc, H, W] (a producer writes every unit, ring included)"""
    nbytes = N.lib().tbg_units_bytes(B, Cc, H, W, planes)
    N.check(min(nbytes, 0), "tbg_units_bytes")
    return UnitTensor(torch.empty(nbytes // 2, device=device, dtype=torch.bfloat16), B, Cc, H, W, planes)


class UnitSink:
    """What a producer launch is asked to write BESIDE its fp32 result (tbg.h "UNIT SINK"): units(out * scale) for the convolution
    that consumes ``out`` next -- scale = that convolution's style modulation s [B, C] (modulated_conv2d.py:94-96: x * s is exactly
    what it and its filter gradient contract) or None for the discriminator's plain convolutions.  kind / O_next describe the
    consumer ("s1": 3x3 stride-1 C -> O_next on the same grid; "up": the 3x3 stride-2 transposed up-convolution C -> O_next), so that
    the producer can tell whether ANY launch of the consumer's forward or backward takes unit tensors for this geometry; if none
    does, nothing is written."""
    __slots__ = ("scale", "kind", "O_next", "produced")

    def __init__(self, scale, kind: str, O_next: int):
        assert kind in ("s1", "up")
        self.scale, self.kind, self.O_next = scale, kind, int(O_next)
        # the UnitTensor the producer wrote (set inside the producing autograd node's forward; read by the layer wrapper, which
        # attaches it to the output tensor).  Deliberately NOT a second output of the node: autograd materialises a zero
        # "gradient" for every non-differentiable output in the backward pass -- a fill of the whole unit tensor per node and pass
        # (measured: 20 launches, 0.25 ms per step, and the L2 pollution slowed the kernels behind them).
        self.produced = None

    def wanted(self, B, Cc, H, W) -> bool:
        if not (TUNING.use_units and TUNING.unit_sinks) or _FMT[_TLS.compute] == FMT_F32 or Cc % 8 != 0:
            return False
        if self.kind == "s1":
            return _units_conv(B, Cc, self.O_next, H, W) or _units_wgrad(Cc, self.O_next, H, W)
        return (_units_t2(B, Cc, self.O_next, H, W, 2 * H + 1, 2 * W + 1) or
                _units_s2(B, self.O_next, Cc, 2 * H + 1, 2 * W + 1))  # (the up-convolution's backward: stride-2 conv O_next -> C)


class _NoSink(UnitSink):
    def __init__(self):
        self.scale, self.kind, self.O_next, self.produced = None, "s1", 0, None

    def wanted(self, B, Cc, H, W) -> bool:
        return False


_NO_SINK = _NoSink()  # "no sink" for the raw helpers' (y, units) return form


def _sink_epi(epi: N.Epilogue, sink: Optional[UnitSink], B, M, H, W, device):
    """(epilogue carrying the sink, UnitTensor | None) for a launch with the fp32 output geometry [B, M, H, W]"""
    if sink is None or not sink.wanted(B, M, H, W):
        return epi, None
    U = units_alloc(B, M, H, W, unit_planes(), device)
    e = N.Epilogue.from_buffer_copy(epi)
    e.units_out, e.units_planes = N.ptr(U.data), U.planes
    e.units_scale = N.ptr(sink.scale.contiguous()) if sink.scale is not None else None
    return e, U


def attach_units(t: torch.Tensor, U: Optional[UnitTensor], scale) -> torch.Tensor:
    """remember on the activation tensor ``t`` that units(t * scale) exists (a Python attribute: the consumer layer finds it)"""
    if U is not None:
        t._tbg_units = (U, scale, None if scale is None else scale._version, t._version)
    return t


def take_units(x: torch.Tensor, scale, planes: Optional[int] = None) -> Optional[UnitTensor]:
    """the unit tensor a producer attached to ``x`` -- if it holds exactly units(x * scale) in the current arithmetic's planes"""
    hit = getattr(x, "_tbg_units", None)
    if hit is None:
        return None
    U, sc, ver, xver = hit
    if x._version != xver:  # x was modified in place after its producer wrote the unit tensor
        return None
    planes = unit_planes() if planes is None else planes
    same = (sc is None and scale is None) or (sc is not None and scale is not None and sc.data_ptr() == scale.data_ptr() and
                                              sc.shape == scale.shape and sc._version == ver)
    if not same or U.planes != planes or (U.B, U.C, U.H, U.W) != tuple(x.shape):
        return None
    return U


def units_pack(x: torch.Tensor, scale: Optional[torch.Tensor] = None, planes: Optional[int] = None) -> UnitTensor:
    """x [B,C,H,W] fp32 (x scale[b,c]) -> its unit tensor (stand-alone producer; fused producers write it from their epilogue)."""
    B, Cc, H, W = x.shape
    planes = unit_planes() if planes is None else planes
    nbytes = N.lib().tbg_units_bytes(B, Cc, H, W, planes)
    N.check(min(nbytes, 0), "tbg_units_bytes")
    U = torch.empty(nbytes // 2, device=x.device, dtype=torch.bfloat16)
    N.check(PROFILE.launch("units_pack_kernel", 0.0, lambda: N.lib().tbg_units_pack_f32(
        N.ptr(x), N.ptr(scale), N.ptr(U), B, Cc, H, W, planes, N.stream()), nbytes=4.0 * x.numel() + nbytes), "tbg_units_pack")
    return UnitTensor(U, B, Cc, H, W, planes)


def bias_act_bwd_units_raw(dout, out_act, epi: N.Epilogue, planes=None, want_dpre=False, want_db=True, want_dn=False,
                           want_dyy=False):
    """bias_act_bwd_raw that writes units(dpre * alpha * out_scale) (tbg_bias_act_bwd_units): returns (DU, dpre | None, pdb, pdn,
    pdy) with the partial sums as [B, M, row chunks]."""
    B, M, H, W = dout.shape
    planes = unit_planes() if planes is None else planes
    nbytes = N.lib().tbg_units_bytes(B, M, H, W, planes)
    N.check(min(nbytes, 0), "tbg_units_bytes")
    U = torch.empty(nbytes // 2, device=dout.device, dtype=torch.bfloat16)
    nch = N.lib().tbg_bias_act_bwd_units_chunks(H)
    mk = lambda: torch.empty((B, M, nch), device=dout.device, dtype=torch.float32)
    dpre = torch.empty_like(dout) if want_dpre else None
    pdb, pdn, pdy = (mk() if want_db else None), (mk() if want_dn else None), (mk() if want_dyy else None)
    _nb = 4.0 * dout.numel() * (2 + int(want_dpre)) + nbytes + (4.0 * B * H * W if epi.noise else 0.0)
    N.check(PROFILE.launch("bias_act_bwd_units_kernel", 0.0, lambda: N.lib().tbg_bias_act_bwd_units(
        N.ptr(dout), N.ptr(out_act), N.ptr(U), planes, N.ptr(dpre), N.ptr(pdb), N.ptr(pdn), N.ptr(pdy), B, M, H, W, C.byref(epi),
        N.stream()), nbytes=_nb), "tbg_bias_act_bwd_units")
    return UnitTensor(U, B, M, H, W, planes), dpre, pdb, pdn, pdy


def conv_units_ok(C_in, M, H, W, KH, KW, stride, pad, transposed, planes) -> bool:
    """geometry of tbg_conv2d_units (3x3 stride-1 pad-1 layers in whole 8 x 32-pixel tiles and 64-channel tiles)"""
    return (KH == 3 and KW == 3 and tuple(stride) == (1, 1) and tuple(pad) == (1, 1) and not transposed and H % 8 == 0 and
            W % 32 == 0 and M % 64 == 0 and C_in % (8 if planes == 3 else 16) == 0)


def conv2d_units_raw(XU: UnitTensor, w: "PackedFilter", M: int, flip=False, epi: Optional[N.Epilogue] = None, dot=None,
                     out: Optional[torch.Tensor] = None, sink: Optional["UnitSink"] = None):
    """3x3 stride-1 pad-1 convolution of the activation behind the unit tensor XU (its scale already inside) with a packed
    filter of the matching format; fp32 NCHW output through the fused epilogue.  dot = (aux, out) as in conv2d_raw.
    sink: returns (y, UnitTensor | None), see conv2d_raw."""
    assert w.fmt == (FMT_X3 if XU.planes == 3 else FMT_BF16) and w.C == XU.C and w.M >= M and w.T == 9
    B, H, W = XU.B, XU.H, XU.W
    if not _units_conv_big(B, XU.C, M, H, W) and _small_conv(B, XU.C, M, H, W, H, W, 3):  # small map: the K split inside the block
        return conv2d_small_raw(XU, w, M, 3, (H, W), flip=flip, epi=epi, dot=dot, out=out, sink=sink)
    if sink is not None:
        epi_s, U = _sink_epi(N.epilogue() if epi is None else epi, sink, B, M, H, W, XU.data.device)
        return conv2d_units_raw(XU, w, M, flip, epi_s, dot, out), U
    d = N.ConvDesc(B, XU.C, M, H, W, H, W, 3, 3, 1, 1, 1, 1, 0, int(flip), w.M, 1)
    epi = N.epilogue() if epi is None else epi
    partial = None
    if dot is not None:
        slots = N.lib().tbg_conv2d_units_dot_slots(C.byref(d), XU.planes)
        N.check(min(slots, 0), "tbg_conv2d_units_dot_slots")
        partial = torch.empty((B, M, slots), device=XU.data.device, dtype=torch.float32)
        epi = N.Epilogue.from_buffer_copy(epi)
        epi.dot_aux, epi.dot_out = N.ptr(dot[0]), N.ptr(partial)
    y = torch.empty((B, M, H, W), device=XU.data.device, dtype=torch.float32) if out is None else out
    _flops = 2.0 * B * M * XU.C * 9 * H * W
    _wtm = N.lib().tbg_conv2d_units_tile_channels(C.byref(d), XU.planes) // 64
    N.check(PROFILE.launch(f"conv_units_fprop_kernel<{XU.planes}, {_wtm}, {N.epilogue_opt(epi)}>", _flops, lambda: N.lib().tbg_conv2d_units(
        C.byref(d), N.ptr(XU.data), XU.planes, N.ptr(w.data), N.ptr(y), C.byref(epi), N.stream()),
        f"conv_units[B={B} C={XU.C} M={M} {H}x{W}]", 2.0 * XU.data.numel() + 4.0 * y.numel() + 2.0 * XU.planes * 9 * XU.C * M),
        "tbg_conv2d_units")
    if partial is not None:
        if dot[1] is None:
            return y, partial
        torch.sum(partial, dim=2, out=dot[1].view(B, M))
    return y


def conv_small_ok(C_in, M, Hin, Win, Hout, Wout, KH, KW, stride, pad, transposed, planes) -> bool:
    """geometry of tbg_conv2d_units_small (tbg.h "SMALL MAPS"): 3x3 stride-1 pad-1, 1x1 with stride 1 | 2, 1x1 transposed"""
    d = N.ConvDesc(1, C_in, M, Hin, Win, Hout, Wout, KH, KW, stride[0], stride[1], pad[0], pad[1], int(transposed), 0, M, 1)
    return N.lib().tbg_conv2d_units_small_blocks(C.byref(d), planes) > 0


def conv2d_small_raw(XU: UnitTensor, w: "PackedFilter", M: int, KH: int, out_hw, stride=(1, 1), transposed=False, flip=False,
                     epi: Optional[N.Epilogue] = None, dot=None, out: Optional[torch.Tensor] = None,
                     sink: Optional["UnitSink"] = None, want_y=True, KW: Optional[int] = None, pad=None):
    """tbg_conv2d_units_small: a small-map convolution of the activation behind the unit tensor XU with the K split inside the block --
    one launch, no slabs.  Geometries: tbg.h "SMALL MAPS" (3x3 stride-1 pad-1; k x k pad-0 with stride; 1x1 / k x k transposed);
    pad defaults to (1, 1) for a 3x3 stride-1 layer and (0, 0) otherwise.  Arguments and return forms as conv2d_units_raw (dot = (aux,
    out | None), sink -> (y, UnitTensor | None)); want_y = False (with a sink that is wanted): the result leaves as a unit tensor
    only (y = None)."""
    KW = KH if KW is None else KW
    assert w.fmt == (FMT_X3 if XU.planes == 3 else FMT_BF16) and w.C == XU.C and w.M >= M and w.T == KH * KW
    B, Hin, Win = XU.B, XU.H, XU.W
    Hout, Wout = out_hw
    if pad is None:
        pad = (1, 1) if (KH == 3 and KW == 3 and tuple(stride) == (1, 1) and not transposed) else (0, 0)
    if sink is not None:
        epi_s, U = _sink_epi(N.epilogue() if epi is None else epi, sink, B, M, Hout, Wout, XU.data.device)
        return conv2d_small_raw(XU, w, M, KH, out_hw, stride, transposed, flip, epi_s, dot, out, None, want_y or U is None, KW, pad), U
    d = N.ConvDesc(B, XU.C, M, Hin, Win, Hout, Wout, KH, KW, stride[0], stride[1], pad[0], pad[1], int(transposed), int(flip), w.M, 1)
    epi = N.epilogue() if epi is None else epi
    partial = None
    if dot is not None:
        slots = N.lib().tbg_conv2d_units_small_dot_slots(C.byref(d), XU.planes)
        N.check(min(slots, 0), "tbg_conv2d_units_small_dot_slots")
        assert slots > 0, "the fused dot needs pixel tiles that stay inside one sample"
        partial = torch.empty((B, M, slots), device=XU.data.device, dtype=torch.float32)
        epi = N.Epilogue.from_buffer_copy(epi)
        epi.dot_aux, epi.dot_out = N.ptr(dot[0]), N.ptr(partial)
    y = out
    if y is None and (want_y or not epi.units_out):
        y = torch.empty((B, M, Hout, Wout), device=XU.data.device, dtype=torch.float32)
    _flops = 2.0 * B * M * XU.C * KH * KW * (Hin * Win if transposed else Hout * Wout)
    _tn = N.lib().tbg_conv2d_units_small_tile_pixels(C.byref(d), XU.planes) // 32
    _kw = 3 if (KH == 3 and KW == 3 and tuple(stride) == (1, 1) and not transposed and tuple(pad) == (1, 1)) else 1
    N.check(PROFILE.launch(f"conv_small_kernel<{XU.planes}, {_kw}, {_tn}>", _flops, lambda: N.lib().tbg_conv2d_units_small(
        C.byref(d), N.ptr(XU.data), XU.planes, N.ptr(w.data), N.ptr(y), C.byref(epi), N.stream()),
        f"conv_small[B={B} C={XU.C} M={M} in={Hin}x{Win} out={Hout}x{Wout} k={KH}x{KW} s={tuple(stride)} T={int(transposed)}]",
        2.0 * XU.data.numel() + (4.0 if y is not None else 0.0) * B * M * Hout * Wout + 2.0 * XU.planes * KH * KW * XU.C * M),
        "tbg_conv2d_units_small")
    if partial is not None:
        if dot[1] is None:
            return y, partial
        torch.sum(partial, dim=2, out=dot[1].view(B, M))
    return y


class PhaseUnitTensor(NamedTuple):
    """the input t [B,C,Hin,Win] of a 3x3 stride-2 pad-0 convolution with Ho x Wo outputs, de-interleaved by parity (tbg.h "PHASE
    unit tensors"): P[planes][B][ceil(C/8)][4][Ho+1][Wo+1][8] bf16 -- inside a phase plane the stride is gone."""
    data: torch.Tensor  # flat bf16
    B: int
    C: int
    Hin: int
    Win: int
    Ho: int
    Wo: int
    planes: int


def units_pack_s2(x: torch.Tensor, scale: Optional[torch.Tensor] = None, planes: Optional[int] = None) -> PhaseUnitTensor:
    """x [B,C,Hin,Win] fp32 (x scale[b,c]) -> its phase unit tensor (stand-alone producer)."""
    B, Cc, Hin, Win = x.shape
    Ho, Wo = (Hin - 3) // 2 + 1, (Win - 3) // 2 + 1
    planes = unit_planes() if planes is None else planes
    nbytes = N.lib().tbg_units_s2_bytes(B, Cc, Ho, Wo, planes)
    N.check(min(nbytes, 0), "tbg_units_s2_bytes")
    U = torch.empty(nbytes // 2, device=x.device, dtype=torch.bfloat16)
    N.check(PROFILE.launch("units_pack_s2_kernel", 0.0, lambda: N.lib().tbg_units_pack_s2_f32(
        N.ptr(x), N.ptr(scale), N.ptr(U), B, Cc, Hin, Win, Ho, Wo, planes, N.stream()), nbytes=4.0 * x.numel() + nbytes),
        "tbg_units_pack_s2")
    return PhaseUnitTensor(U, B, Cc, Hin, Win, Ho, Wo, planes)


def upfirdn2d_units_s2(x: torch.Tensor, k: torch.Tensor, pad=(0, 0, 0, 0), in_scale: Optional[torch.Tensor] = None,
                       planes: Optional[int] = None) -> PhaseUnitTensor:
    """the phase unit tensor of upfirdn2d_raw(x, k, pad=pad, in_scale=in_scale) (k: one of fir_kernel's separable filters) in one
    launch -- the FIR pass in front of a stride-2 convolution as the PRODUCER of that convolution's operand."""
    B, Cc, H, W = x.shape
    kH, kW = k.shape
    sep = _sep_factors(k)
    assert sep is not None, "the fused producer takes the model's separable filters (fir_kernel)"
    Ht, Wt = H + pad[2] + pad[3] - kH + 1, W + pad[0] + pad[1] - kW + 1
    Ho, Wo = (Ht - 3) // 2 + 1, (Wt - 3) // 2 + 1
    planes = unit_planes() if planes is None else planes
    nbytes = N.lib().tbg_units_s2_bytes(B, Cc, Ho, Wo, planes)
    N.check(min(nbytes, 0), "tbg_units_s2_bytes")
    U = torch.empty(nbytes // 2, device=x.device, dtype=torch.bfloat16)
    N.check(PROFILE.launch("fir_units_s2_kernel", 0.0, lambda: N.lib().tbg_upfirdn2d_units_s2_f32(
        N.ptr(x), N.ptr(sep[0]), N.ptr(sep[1]), N.ptr(U), B, Cc, H, W, kH, kW, pad[0], pad[1], pad[2], pad[3], N.ptr(in_scale),
        planes, N.stream()), nbytes=4.0 * x.numel() + nbytes), "tbg_upfirdn2d_units_s2")
    return PhaseUnitTensor(U, B, Cc, Ht, Wt, Ho, Wo, planes)


def conv_units_s2_ok(C_in, M, Hin, Win, planes) -> bool:
    """geometry of tbg_conv2d_units_s2 (3x3 stride-2 pad-0 layers with whole 8 x 32-pixel output tiles and 64-channel tiles)"""
    Ho, Wo = (Hin - 3) // 2 + 1, (Win - 3) // 2 + 1
    return Hin >= 3 and Win >= 3 and Ho % 8 == 0 and Wo % 32 == 0 and M % 64 == 0 and C_in % (8 if planes == 3 else 16) == 0


def conv2d_units_s2_raw(XP: PhaseUnitTensor, w: "PackedFilter", M: int, flip=False, epi: Optional[N.Epilogue] = None, dot=None,
                        out: Optional[torch.Tensor] = None, sink: Optional["UnitSink"] = None):
    """3x3 stride-2 pad-0 convolution of the tensor behind the phase unit tensor XP with a packed filter of the matching format;
    fp32 NCHW output through the fused epilogue.  dot = (aux, out) as in conv2d_raw.  sink: returns (y, UnitTensor | None)."""
    assert w.fmt == (FMT_X3 if XP.planes == 3 else FMT_BF16) and w.C == XP.C and w.M >= M and w.T == 9
    B, Ho, Wo = XP.B, XP.Ho, XP.Wo
    if sink is not None:
        epi_s, U = _sink_epi(N.epilogue() if epi is None else epi, sink, B, M, Ho, Wo, XP.data.device)
        return conv2d_units_s2_raw(XP, w, M, flip, epi_s, dot, out), U
    d = N.ConvDesc(B, XP.C, M, XP.Hin, XP.Win, Ho, Wo, 3, 3, 2, 2, 0, 0, 0, int(flip), w.M, 1)
    epi = N.epilogue() if epi is None else epi
    partial = None
    if dot is not None:
        slots = N.lib().tbg_conv2d_units_s2_dot_slots(C.byref(d), XP.planes)
        N.check(min(slots, 0), "tbg_conv2d_units_s2_dot_slots")
        partial = torch.empty((B, M, slots), device=XP.data.device, dtype=torch.float32)
        epi = N.Epilogue.from_buffer_copy(epi)
        epi.dot_aux, epi.dot_out = N.ptr(dot[0]), N.ptr(partial)
    y = torch.empty((B, M, Ho, Wo), device=XP.data.device, dtype=torch.float32) if out is None else out
    _flops = 2.0 * B * M * XP.C * 9 * Ho * Wo
    _wtm = N.lib().tbg_conv2d_units_s2_tile_channels(C.byref(d), XP.planes) // 64
    N.check(min(_wtm, 0), "tbg_conv2d_units_s2_tile_channels")
    N.check(PROFILE.launch(f"conv_units_s2_fprop_kernel<{XP.planes}, {_wtm}>", _flops, lambda: N.lib().tbg_conv2d_units_s2(
        C.byref(d), N.ptr(XP.data), XP.planes, N.ptr(w.data), N.ptr(y), C.byref(epi), N.stream()),
        f"conv_units_s2[B={B} C={XP.C} M={M} {XP.Hin}x{XP.Win}->{Ho}x{Wo}]",
        2.0 * XP.data.numel() + 4.0 * y.numel() + 2.0 * XP.planes * 9 * XP.C * M), "tbg_conv2d_units_s2")
    if partial is not None:
        if dot[1] is None:
            return y, partial
        torch.sum(partial, dim=2, out=dot[1].view(B, M))
    return y


def wgrad_units_ok(CS, CL, Hs, Ws, Hl, Wl, KH, KW, stride, pad) -> bool:
    """geometry of tbg_conv2d_wgrad_units (3x3 stride-1 pad-1 layers with whole 2 x 32-pixel chunks and 64-channel tiles)"""
    return (KH == 3 and KW == 3 and tuple(stride) == (1, 1) and tuple(pad) == (1, 1) and (Hl, Wl) == (Hs, Ws) and Ws % 32 == 0
            and Hs % 2 == 0 and CS % 64 == 0 and CL % 64 == 0)


def wgrad_units_raw(SU: UnitTensor, LU: UnitTensor, out: torch.Tensor, st_t: int, st_l: int, st_s: int, alpha: float,
                    out_offset: int = 0, add=None, bias=None):
    """filter gradient of a 3x3 stride-1 pad-1 convolution from the unit tensors of S (output-grid tensor, scale inside) and
    L (input-grid tensor, scale inside).  Overwrites ``out`` like wgrad_raw."""
    assert SU.planes == LU.planes and SU.B == LU.B
    d = _bias_rider(N.WgradDesc(SU.B, SU.C, LU.C, SU.H, SU.W, LU.H, LU.W, 3, 3, 1, 1, 1, 1, st_t, st_l, st_s, alpha), bias)
    nbytes = N.lib().tbg_conv2d_wgrad_units_workspace_bytes(C.byref(d))
    N.check(min(nbytes, 0), "tbg_conv2d_wgrad_units_workspace_bytes")
    ws = _workspace(out.device, nbytes)
    addw, addq, gamma = add if add is not None else (None, None, 0.0)
    _flops = 2.0 * SU.B * SU.C * LU.C * SU.H * SU.W * 9
    N.check(PROFILE.launch(f"conv_wgrad_units_kernel<{SU.planes}>", _flops, lambda: N.lib().tbg_conv2d_wgrad_units(
        C.byref(d), N.ptr(SU.data), N.ptr(LU.data), SU.planes, N.ptr(out) + 4 * out_offset,
        (N.ptr(addw) + 4 * out_offset) if addw is not None else None, N.ptr(addq), gamma, N.ptr(ws), ws.numel() * 4, N.stream()),
        f"wgrad_units[B={SU.B} CS={SU.C} CL={LU.C} {SU.H}x{SU.W}]",
        2.0 * SU.planes * (SU.data.numel() + LU.data.numel()) / SU.planes + 36.0 * SU.C * LU.C), "tbg_conv2d_wgrad_units")
    return out


def conv_units_t2_ok(C_in, M, planes) -> bool:
    """geometry of tbg_conv2d_units_t2 (3x3 stride-2 transposed convolutions in 64-channel tiles)"""
    return M % 64 == 0 and C_in % (8 if planes == 3 else 16) == 0


def conv2d_units_t2_raw(XU: UnitTensor, w: "PackedFilter", M: int, out_hw, flip=False, alpha=1.0,
                        out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """alpha * the 3x3 stride-2 transposed convolution of the activation behind the unit tensor XU (its scale inside) with a
    packed filter of the matching format; out_hw in {2H + 1, 2H + 2} x {2W + 1, 2W + 2}; fp32 NCHW."""
    assert w.fmt == (FMT_X3 if XU.planes == 3 else FMT_BF16) and w.C == XU.C and w.M >= M and w.T == 9
    B, H, W = XU.B, XU.H, XU.W
    Hout, Wout = out_hw
    d = N.ConvDesc(B, XU.C, M, H, W, Hout, Wout, 3, 3, 2, 2, 0, 0, 1, int(flip), w.M, 1)
    y = torch.empty((B, M, Hout, Wout), device=XU.data.device, dtype=torch.float32) if out is None else out
    _flops = 2.0 * B * M * XU.C * 9 * H * W
    N.check(PROFILE.launch(f"conv_units_t2_kernel<{XU.planes}>", _flops, lambda: N.lib().tbg_conv2d_units_t2(
        C.byref(d), N.ptr(XU.data), XU.planes, N.ptr(w.data), N.ptr(y), alpha, N.stream()),
        f"conv_units_t2[B={B} C={XU.C} M={M} {H}x{W}->{Hout}x{Wout}]",
        2.0 * XU.data.numel() + 4.0 * y.numel() + 2.0 * XU.planes * 9 * XU.C * M), "tbg_conv2d_units_t2")
    return y


def wgrad_units_s2_ok(CS, CL, Hs, Ws, Hl, Wl) -> bool:
    """geometry of tbg_conv2d_wgrad_units_s2 (3x3 stride-2 pad-0 layers with whole 32-pixel output rows, 128 S- / 64 L-channel tiles)"""
    return Hl >= 3 and Wl >= 3 and Hs == (Hl - 3) // 2 + 1 and Ws == (Wl - 3) // 2 + 1 and Ws % 32 == 0 and CS % 128 == 0 and CL % 64 == 0


def wgrad_units_s2_raw(SU: UnitTensor, LP: PhaseUnitTensor, out: torch.Tensor, st_t: int, st_l: int, st_s: int, alpha: float,
                       out_offset: int = 0, add=None, bias=None):
    """filter gradient of a 3x3 stride-2 pad-0 convolution from the unit tensor of S (output-grid tensor, scale inside) and the
    phase unit tensor of L (input-grid tensor, scale inside).  Overwrites ``out`` like wgrad_raw."""
    assert SU.planes == LP.planes and SU.B == LP.B and (SU.H, SU.W) == (LP.Ho, LP.Wo)
    d = _bias_rider(N.WgradDesc(SU.B, SU.C, LP.C, SU.H, SU.W, LP.Hin, LP.Win, 3, 3, 2, 2, 0, 0, st_t, st_l, st_s, alpha), bias)
    nbytes = N.lib().tbg_conv2d_wgrad_units_s2_workspace_bytes(C.byref(d))
    N.check(min(nbytes, 0), "tbg_conv2d_wgrad_units_s2_workspace_bytes")
    ws = _workspace(out.device, nbytes)
    addw, addq, gamma = add if add is not None else (None, None, 0.0)
    _flops = 2.0 * SU.B * SU.C * LP.C * SU.H * SU.W * 9
    N.check(PROFILE.launch(f"conv_wgrad_units_s2_kernel<{SU.planes}>", _flops, lambda: N.lib().tbg_conv2d_wgrad_units_s2(
        C.byref(d), N.ptr(SU.data), N.ptr(LP.data), SU.planes, N.ptr(out) + 4 * out_offset,
        (N.ptr(addw) + 4 * out_offset) if addw is not None else None, N.ptr(addq), gamma, N.ptr(ws), ws.numel() * 4, N.stream()),
        f"wgrad_units_s2[B={SU.B} CS={SU.C} CL={LP.C} {SU.H}x{SU.W}]",
        2.0 * (SU.data.numel() + LP.data.numel()) + 36.0 * SU.C * LP.C), "tbg_conv2d_wgrad_units_s2")
    return out


HAVE_WGRAD_X3 = True  # tbg_conv2d_wgrad_x3 (falls back to the exact fp32 kernel inside the library for the small geometries)


class PackedFilter(NamedTuple):
    """Wp[T][ceil(C/4)][M][4] fp32 (tbg_weight_pack_f32) or Wp[T][ceil(C/8)][M][8] bf16 (tbg_weight_pack_bf16): the filter
    formats of tbg_conv2d_f32 / tbg_conv2d_bf16."""
    data: torch.Tensor
    T: int
    C: int
    M: int
    fmt: int = FMT_F32  # FMT_F32 | FMT_BF16 | FMT_X3 (three bf16 planes, tbg_weight_pack_x3)

    @property
    def bf16(self) -> bool:
        return self.fmt == FMT_BF16




class PackedStore:
    """Persistent packed filters of ONE training step object, refreshed by ONE multi-tensor launch per step
    (tbg_weight_pack_multi) instead of ~140 small pack launches.

    Contract: the owner calls ``refresh()`` at the START of every step (after the previous optimiser update, before the
    forward) and keeps the weights constant until the step's backward passes are done -- exactly the lifetime of
    ``filter_cache()``.  Inside ``scope()`` a ``pack_filter`` of a live parameter returns its persistent pack; the first time a
    (parameter, orientation, format) is seen it is registered and packed on the spot (so the step that discovers it is
    correct too) and joins the table for the following steps.  Nothing is served outside ``scope()``, so code that changes
    weights without going through the owner (load_state_dict, tests) can never meet a stale pack."""

    def __init__(self):
        self.items = {}        # key -> (PackedFilter, PackItem fields)
        self._table = None     # device uint8 tensor holding the tbg_pack_item array
        self._n = 0
        self._keep = []        # superseded tables: HIP graphs captured earlier still read them
        self._fresh = False    # set by refresh(), cleared when the scope ends

    def scope(self):
        store = self

        class _Scope:
            def __enter__(self_):
                self_._outer, _TLS.pack_store = _TLS.pack_store, store
                return store

            def __exit__(self_, *exc):
                _TLS.pack_store = self_._outer
                store._fresh = False
                # a step that discovered new filters (always an EAGER step: get() never registers during capture) rebuilds the
                # device table here, outside any capture, so that refresh() never has to copy from the host inside one
                if store.items and store._n != len(store.items) and not torch.cuda.is_current_stream_capturing():
                    store._build_table()
                return False
        return _Scope()

    def _build_table(self):
        arr = (N.PackItem * len(self.items))()
        for k, (pf, f) in enumerate(self.items.values()):
            arr[k] = N.PackItem(*f)
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        dev = next(iter(self.items.values()))[0].data.device
        if self._table is not None:
            self._keep.append(self._table)
        self._table, self._n = host.to(dev), len(self.items)

    def refresh(self):
        """pack every registered filter from the current weights (one launch); call before the step's forward."""
        if self.items:
            if self._table is None:
                self._build_table()
            N.check(N.lib().tbg_weight_pack_multi(self._table.data_ptr(), self._n, N.stream()), "tbg_weight_pack_multi")
        self._fresh = True

    def get(self, w, T, I, O, transpose, flip, fmt):
        if not self._fresh:
            return None
        key = (w.data_ptr(), T, I, O, bool(transpose), bool(flip), int(fmt))
        hit = self.items.get(key)
        if hit is not None:
            # in the table since an earlier step -> refreshed this step (entries added after this step's refresh carry
            # their own on-the-spot pack, see below)
            return hit[0]
        if torch.cuda.is_current_stream_capturing():
            return None  # never grow the table while a graph is being captured: the caller packs on demand
        pf = _pack_now(w, T, I, O, transpose, flip, fmt)
        self.items[key] = (pf, (w.data_ptr(), pf.data.data_ptr(), T, I, O, int(transpose), int(flip), int(fmt)))
        return pf


class filter_cache:
    """Scope in which packed filters of PARAMETERS are reused (forward, and the data gradients of the three backward
    passes, see the same weights).  Nothing outlives the scope, so an optimiser update can never meet a stale pack."""

    def __enter__(self):
        self._outer = _TLS.pack_step
        if _TLS.pack_step is None:
            _TLS.pack_step = {}
        return self

    def __exit__(self, *exc):
        _TLS.pack_step = self._outer
        return False


def _fmt_of(bf16) -> int:
    """None: the current ops.compute_dtype; bool (legacy): bf16 / f32; str: a COMPUTE_MODES name; int: a FMT_* value."""
    if bf16 is None:
        return _FMT[_TLS.compute]
    if isinstance(bf16, str):
        return _FMT[bf16]
    if isinstance(bf16, bool):
        return FMT_BF16 if bf16 else FMT_F32
    return int(bf16)


def pack_filter(w: torch.Tensor, transpose: bool, flip: bool, bf16=None) -> PackedFilter:
    """HWIO [KH,KW,I,O] / [T,I,O] parameter -> PackedFilter.  transpose: C = O, M = I (data gradient).
    bf16: the filter format (see _fmt_of; None = the current ops.compute_dtype)."""
    fmt = _fmt_of(bf16)
    if w.dim() == 4:
        T, I, O = w.shape[0] * w.shape[1], w.shape[2], w.shape[3]
    else:
        T, I, O = w.shape
    cache = None
    if _TLS.pack_step is not None and w.is_leaf and w.requires_grad:  # live parameters only: the key is an address
        cache = _TLS.pack_step
    key = (w.data_ptr(), T, I, O, bool(transpose), bool(flip), w._version, fmt)
    if cache is not None and key in cache:
        return cache[key]
    if cache is not None and _TLS.pack_store is not None and w.is_contiguous():
        pf = _TLS.pack_store.get(w, T, I, O, transpose, flip, fmt)
        if pf is not None:
            return pf
    pf = _pack_now(w.contiguous(), T, I, O, transpose, flip, fmt)
    if cache is not None:
        cache[key] = pf
    return pf


def _pack_now(w, T, I, O, transpose, flip, fmt) -> PackedFilter:
    fmt = int(fmt)
    if fmt == FMT_X3:
        nb = N.lib().tbg_weight_pack_x3_bytes(T, I, O, int(transpose))
        out = torch.empty(nb // 2, device=w.device, dtype=torch.bfloat16)
        N.check(N.lib().tbg_weight_pack_x3(N.ptr(w), N.ptr(out), T, I, O, int(transpose), int(flip), N.stream()),
                "tbg_weight_pack_x3")
    elif fmt == FMT_BF16:
        nb = N.lib().tbg_weight_pack_bf16_bytes(T, I, O, int(transpose))
        out = torch.empty(nb // 2, device=w.device, dtype=torch.bfloat16)
        N.check(N.lib().tbg_weight_pack_bf16(N.ptr(w), N.ptr(out), T, I, O, int(transpose), int(flip), N.stream()),
                "tbg_weight_pack_bf16")
    else:
        n = N.lib().tbg_weight_pack_floats(T, I, O, int(transpose))
        out = torch.empty(n, device=w.device, dtype=torch.float32)
        N.check(N.lib().tbg_weight_pack_f32(N.ptr(w), N.ptr(out), T, I, O, int(transpose), int(flip), N.stream()),
                "tbg_weight_pack")
    return PackedFilter(out, T, O if transpose else I, I if transpose else O, fmt)


def bias_act_bwd_raw(dout, out_act, epi: N.Epilogue, want_dx=False, want_dpre=True, want_db=True, want_dn=False,
                     want_dyy=False):
    B, M = dout.shape[0], dout.shape[1]
    HW = dout.numel() // (B * M)
    nch = N.lib().tbg_bias_act_bwd_chunks(HW)
    mk = lambda: torch.empty((B, M, nch), device=dout.device, dtype=torch.float32)
    dx = torch.empty_like(dout) if want_dx else None
    dpre = torch.empty_like(dout) if want_dpre else None
    pdb = mk() if want_db else None
    pdn = mk() if want_dn else None
    pdy = mk() if want_dyy else None
    _nb = 4.0 * dout.numel() * (2 + int(want_dx) + int(want_dpre)) + (4.0 * B * HW if epi.noise else 0.0)
    N.check(PROFILE.launch("bias_act_bwd_kernel" if HW > 1024 else "bias_act_bwd_small_kernel", 0.0,
                           lambda: N.lib().tbg_bias_act_bwd_f32(N.ptr(dout), N.ptr(out_act), N.ptr(dx), N.ptr(dpre), N.ptr(pdb),
                                                                N.ptr(pdn), N.ptr(pdy), B, M, HW, C.byref(epi), N.stream()),
                           nbytes=_nb), "tbg_bias_act_bwd")
    return dx, dpre, pdb, pdn, pdy


def bias_act_fwd_raw(x, epi: N.Epilogue):
    B, M = x.shape[0], x.shape[1]
    HW = x.numel() // (B * M)
    y = torch.empty_like(x)
    N.check(N.lib().tbg_bias_act_fwd_f32(N.ptr(x), N.ptr(y), B, M, HW, C.byref(epi), N.stream()), "tbg_bias_act_fwd")
    return y


def _check_colmask(colmask, B, W, mask_cw):
    """the kernels index colmask[b * ceil(W / mask_cw) + column // mask_cw] unchecked: a mask of another batch / width would
    be read out of bounds or from the wrong row, where the mask_text_box multiply it replaces raised a broadcast error."""
    if colmask is None:
        return
    nb = -(-W // int(mask_cw)) if mask_cw > 0 else -1
    if (tuple(colmask.shape) != (B, nb) or colmask.dtype != torch.float32 or not colmask.is_contiguous()):
        raise ValueError(f"colmask must be a contiguous float32 [{B}, {nb}] tensor (W = {W}, mask_cw = {mask_cw}), got "
                         f"{tuple(colmask.shape)} {colmask.dtype}")


def rgb_project_raw(x, w2d, O, scale, bias, skip, alpha, bias_mul=1.0, out=None, colmask=None, mask_cw=0):
    """y[b,o,p] = (alpha * sum_c x[b,c,p] w2d[c,o] scale[b,c] + bias[o] + skip[b,o,p]) * m   (O <= 4; w2d rows of ldw = O).
    colmask [B, W // mask_cw]: m = colmask[b, column // mask_cw] (mask_text_box as the epilogue)."""
    B, Cc, H, W = x.shape
    _check_colmask(colmask, B, W, mask_cw)
    y = torch.empty((B, O, H, W), device=x.device, dtype=torch.float32) if out is None else out
    _nb = 4.0 * (x.numel() + y.numel() * (2 if skip is not None else 1))
    N.check(PROFILE.launch("rgb_project_kernel", 0.0, lambda: N.lib().tbg_rgb_project_f32(
        N.ptr(x), N.ptr(w2d), N.ptr(scale), N.ptr(bias), N.ptr(skip), N.ptr(y), B, Cc, O, O, H * W, alpha, bias_mul,
        N.ptr(colmask), W, int(mask_cw), N.stream()), nbytes=_nb), "tbg_rgb_project")
    return y


def rgb_backproject_raw(x, dy, w2d, scale, alpha, want_dx=True, want_G=True, colmask=None, mask_cw=0, want_dym=False,
                        parts=False):
    """dym = dy * m;  dx[b,c,p] = alpha*scale[b,c]*sum_o w2d[c,o] dym[b,o,p];  G[b,c,o] = sum_p x[b,c,p] dym[b,o,p]
    (the kernel writes one partial G per 2048-pixel chunk -- no atomics, deterministic -- summed here).
    returns (dx, G) or (dx, G, dym).  parts=True: G stays [B, C, chunks, O] and the per-chunk sums of dym [B, chunks, O] come
    too -- (dx, Gparts, dysum[, dym]) -- for tbg_torgb_bwd_smalls_f32, which sums both (no reduction launches in between)."""
    B, Cc, H, W = x.shape
    O = dy.shape[1]
    _check_colmask(colmask, B, W, mask_cw)
    dx = torch.empty_like(x) if want_dx else None
    nchunk = N.lib().tbg_rgb_backproject_chunks(H * W)
    Gp = torch.empty((B, Cc, nchunk, O), device=x.device, dtype=torch.float32) if want_G else None
    dym = torch.empty_like(dy) if want_dym else None
    dysum = torch.empty((B, nchunk, O), device=x.device, dtype=torch.float32) if parts else None
    _nb = 4.0 * (x.numel() * (int(want_dx) + int(want_G)) + dy.numel())
    N.check(PROFILE.launch("rgb_backproject_kernel", 0.0, lambda: N.lib().tbg_rgb_backproject_f32(
        N.ptr(x), N.ptr(dy), N.ptr(w2d), N.ptr(scale), N.ptr(dx), N.ptr(Gp), B, Cc, O, O, H * W, alpha, N.ptr(colmask), W,
        int(mask_cw), N.ptr(dym), N.ptr(dysum), N.stream()), nbytes=_nb), "tbg_rgb_backproject")
    if parts:
        return (dx, Gp, dysum, dym) if want_dym else (dx, Gp, dysum)
    G = (Gp.sum(dim=2) if nchunk > 1 else Gp[:, :, 0]) if want_G else None
    return (dx, G, dym) if want_dym else (dx, G)


def modconv_bwd_smalls_raw(pdb, pdn, pdy, d, s, wsq, ds_conv):
    """one launch for the small-tensor tail of a modulated-conv backward: (db [O], dstrength [], ds [B,I], dwsq [I,O])."""
    B, O, nch = pdy.shape
    I = s.shape[1]
    dev = s.device
    slots = ds_conv.shape[2] if ds_conv.dim() == 3 else 1  # [B, I, slots]: the conv launch's dot partials, summed by the kernel
    db = torch.empty(O, device=dev, dtype=torch.float32)
    dstrength = torch.empty((), device=dev, dtype=torch.float32) if pdn is not None else None
    ds = torch.empty_like(s)
    dwsq = torch.empty((I, O), device=dev, dtype=torch.float32)
    N.check(N.lib().tbg_modconv_bwd_smalls_f32(N.ptr(pdb), N.ptr(pdn), N.ptr(pdy), N.ptr(d), N.ptr(s), N.ptr(wsq),
                                               N.ptr(ds_conv), N.ptr(db), N.ptr(dstrength), N.ptr(ds), N.ptr(dwsq), B, I, O,
                                               nch, slots, N.stream()), "tbg_modconv_bwd_smalls")
    return db, dstrength, ds, dwsq


def torgb_bwd_smalls_raw(G, w2d, s, coef, dysum=None):
    """G [B, C, O] or its per-chunk partials [B, C, chunks, O]; dysum [B, chunks, O] (optional) -> (ds, dw[, db])."""
    if G.dim() == 3:
        G = G.unsqueeze(2)
    B, Cc, nchunk, O = G.shape
    ds = torch.empty((B, Cc), device=G.device, dtype=torch.float32)
    dw = torch.empty((Cc, O), device=G.device, dtype=torch.float32)
    db = torch.empty(O, device=G.device, dtype=torch.float32) if dysum is not None else None
    N.check(N.lib().tbg_torgb_bwd_smalls_f32(N.ptr(G.contiguous()), N.ptr(w2d), N.ptr(s), N.ptr(ds), N.ptr(dw), B, Cc, O, coef, nchunk,
                                             N.ptr(dysum), N.ptr(db), N.stream()), "tbg_torgb_bwd_smalls")
    return (ds, dw, db) if dysum is not None else (ds, dw)


def demod_coefs_raw(s: torch.Tensor, w: torch.Tensor, coef: float):
    KH, KW, I, O = w.shape
    B = s.shape[0]
    wsq = torch.empty((I, O), device=w.device, dtype=torch.float32)
    d = torch.empty((B, O), device=w.device, dtype=torch.float32)
    N.check(N.lib().tbg_demod_coefs_f32(N.ptr(s), N.ptr(w), N.ptr(wsq), N.ptr(d), B, KH * KW, I, O, coef, N.stream()),
            "tbg_demod_coefs")
    return d, wsq


# ----------------------------------------------------------------------------------------
# composable primitives (gradients of any order)
# ----------------------------------------------------------------------------------------
_FLIP_CACHE = {}


def _flipped_fir(k: torch.Tensor) -> torch.Tensor:
    """flip(k) for the cached FIR constants (fir_kernel): one flip per filter instead of one per backward call.  Only
    tensors owned by _FIR_CACHE are memoised (they live for the process, so their address is a safe key)."""
    key = k.data_ptr()
    if any(v is k for v in _FIR_CACHE.values()) or key in _FLIP_CACHE and _FLIP_CACHE[key][0] is k:
        hit = _FLIP_CACHE.get(key)
        if hit is None or hit[0] is not k:
            kf = torch.flip(k, (0, 1)).contiguous()
            _FLIP_CACHE[key] = (k, kf)
            _FLIP_CACHE[kf.data_ptr()] = (kf, k)  # flip(flip(k)) = k: gradients of gradients bounce between the two
            sep = _sep_factors(k)
            if sep is not None:
                _FIR_SEP[kf.data_ptr()] = (kf, torch.flip(sep[0], (0,)).contiguous(), torch.flip(sep[1], (0,)).contiguous())
            return kf
        return hit[1]
    return torch.flip(k, (0, 1)).contiguous()


def _half(role, B):
    """leading samples a "d"-role node has to differentiate in this pass (FLAGS.d_first_half), or 0 = the whole batch."""
    h = FLAGS.d_first_half
    return h if (h and role in ("d", "d_image") and h < B) else 0


def _tail_empty(shape, device):
    """gradient tensor of a d_first_half node whose trailing samples nobody reads: uninitialised, or filled with
    FLAGS.unread_tail_fill (debug aid: a test runs a step with NaN tails and with zero tails and requires finite, equal
    results, i.e. proves the tail is never read)."""
    fill = FLAGS.unread_tail_fill
    if fill is None:
        return torch.empty(shape, device=device, dtype=torch.float32)
    return torch.full(shape, float(fill), device=device, dtype=torch.float32)


class _UpFirDn2D(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, up, down, pad, role=None):
        ctx.save_for_backward(k)
        ctx.geom = (x.shape[2], x.shape[3], up, down, pad)
        ctx.role = role
        ctx.xshape = tuple(x.shape)
        return upfirdn2d_raw(x.contiguous(), k, up, down, pad)

    @staticmethod
    def backward(ctx, dy):
        (k,) = ctx.saved_tensors
        inH, inW, up, down, pad = ctx.geom
        kH, kW = k.shape
        outW = (inW * up[0] + pad[0] + pad[1] - kW) // down[0] + 1
        outH = (inH * up[1] + pad[2] + pad[3] - kH) // down[1] + 1
        # upfirdn_2d_v2.py:204-209
        gpad = (kW - pad[0] - 1, inW * up[0] - outW * down[0] + pad[0] - up[0] + 1,
                kH - pad[2] - 1, inH * up[1] - outH * down[1] + pad[2] - up[1] + 1)
        h = _half(ctx.role, dy.shape[0])
        if h:  # first-order pass over the leading samples only; the rest of dx is never read
            dx = _tail_empty(ctx.xshape, dy.device)
            upfirdn2d_raw(dy.contiguous()[:h], _flipped_fir(k), down, up, gpad, out=dx[:h])
            return dx, None, None, None, None, None
        return _UpFirDn2D.apply(dy, _flipped_fir(k), down, up, gpad), None, None, None, None, None


def upfirdn2d(x, k, up=(1, 1), down=(1, 1), pad=(0, 0, 0, 0), role=None):
    """The reference op on NCHW input; differentiable to any order.  role: see FLAGS ("d": a discriminator layer)."""
    return _UpFirDn2D.apply(x, k, tuple(up), tuple(down), tuple(pad), role)


class _Geom:
    """Geometry of y = conv(x, w): stride, top/left pad, filter, spatial sizes of x and y."""
    __slots__ = ("stride", "pad", "KH", "KW", "xhw", "yhw")

    def __init__(self, stride, pad, KH, KW, xhw, yhw):
        self.stride, self.pad, self.KH, self.KW, self.xhw, self.yhw = stride, pad, KH, KW, xhw, yhw


def _fwd_launch(x, w, g: _Geom, alpha=1.0):
    O = w.shape[3]
    return conv2d_raw(x, pack_filter(w, False, False), O, g.KH, g.KW, g.yhw, g.stride, g.pad, epi=N.epilogue(alpha=alpha))


def _bwd_data_launch(dy, w, g: _Geom, alpha=1.0, in_scale=None, epi=None, dot=None, out=None):
    """dx[i,Y,X] = sum dy[o,oy,ox] w[kh,kw,i,o] over oy*s - p + kh = Y."""
    I = w.shape[2]
    epi = epi if epi is not None else N.epilogue(alpha=alpha)
    if g.stride == (1, 1):
        wt = pack_filter(w, transpose=True, flip=True)
        return conv2d_raw(dy, wt, I, g.KH, g.KW, g.xhw, (1, 1), (g.KH - 1 - g.pad[0], g.KW - 1 - g.pad[1]),
                          in_scale=in_scale, epi=epi, dot=dot, out=out)
    assert g.pad == (0, 0), "strided convolutions on this path are VALID"
    wt = pack_filter(w, transpose=True, flip=False)
    return conv2d_raw(dy, wt, I, g.KH, g.KW, g.xhw, g.stride, (0, 0), transposed=True, in_scale=in_scale, epi=epi,
                      dot=dot, out=out)


def _bwd_weight_launch(x, dy, g: _Geom, I, O, alpha=1.0, x_scale=None, dy_scale=None, add=None, bias=None):
    dw = torch.empty((g.KH, g.KW, I, O), device=x.device, dtype=torch.float32)
    wgrad_raw(dy, x, g.KH, g.KW, g.stride, g.pad, dw, I * O, O, 1, alpha, s_scale=dy_scale, l_scale=x_scale, add=add, bias=bias)
    return dw


class _Conv2d(torch.autograd.Function):
    """y = alpha * conv(x, w).  The three primitives (forward, data gradient, filter gradient) are each other's derivatives,
    so the family is closed under differentiation to any order; the scalar alpha (the equalised-LR coefficient) rides along
    instead of being a separate elementwise pass over the filter in every order of derivative."""

    @staticmethod
    def forward(ctx, x, w, g, alpha=1.0):
        ctx.save_for_backward(x, w)
        ctx.g, ctx.alpha = g, alpha
        return _fwd_launch(x.contiguous(), w.contiguous(), g, alpha)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx = _Conv2dBwdData.apply(dy, w, ctx.g, ctx.alpha) if ctx.needs_input_grad[0] else None
        dw = (_Conv2dBwdWeight.apply(x, dy, ctx.g, ctx.alpha)
              if (ctx.needs_input_grad[1] and not FLAGS.no_filter_grads) else None)
        return dx, dw, None, None


class _Conv2dBwdData(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dy, w, g, alpha=1.0):
        ctx.save_for_backward(dy, w)
        ctx.g, ctx.alpha = g, alpha
        return _bwd_data_launch(dy.contiguous(), w.contiguous(), g, alpha)

    @staticmethod
    def backward(ctx, gdx):
        dy, w = ctx.saved_tensors
        g_dy = _Conv2d.apply(gdx, w, ctx.g, ctx.alpha) if ctx.needs_input_grad[0] else None
        g_w = (_Conv2dBwdWeight.apply(gdx, dy, ctx.g, ctx.alpha)
               if (ctx.needs_input_grad[1] and not FLAGS.no_filter_grads) else None)
        return g_dy, g_w, None, None


class _Conv2dBwdWeight(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dy, g, alpha=1.0):
        ctx.save_for_backward(x, dy)
        ctx.g, ctx.alpha = g, alpha
        return _bwd_weight_launch(x.contiguous(), dy.contiguous(), g, x.shape[1], dy.shape[1], alpha)

    @staticmethod
    def backward(ctx, gw):
        x, dy = ctx.saved_tensors
        gw = gw.contiguous()
        g_x = _Conv2dBwdData.apply(dy, gw, ctx.g, ctx.alpha) if ctx.needs_input_grad[0] else None
        g_dy = _Conv2d.apply(x, gw, ctx.g, ctx.alpha) if ctx.needs_input_grad[1] else None
        return g_x, g_dy, None, None


def conv2d(x, w, stride=(1, 1), pad=(0, 0), alpha=1.0):
    """y[b,o] = alpha * sum x[b,i, oy*s-p+kh, ox*s-p+kw] w[kh,kw,i,o]  (HWIO filter)."""
    KH, KW = w.shape[0], w.shape[1]
    H, W = x.shape[2], x.shape[3]
    yhw = ((H + 2 * pad[0] - KH) // stride[0] + 1, (W + 2 * pad[1] - KW) // stride[1] + 1)
    if stride != (1, 1):
        assert pad == (0, 0)
    return _Conv2d.apply(x, w, _Geom(tuple(stride), tuple(pad), KH, KW, (H, W), yhw), float(alpha))


def conv_transpose2d_s2(x, wt, alpha=1.0):
    """y[b,o,2a+kh,2b'+kw] += alpha * x[b,i,a,b'] wt[kh,kw,i,o]: the data gradient of a stride-2 VALID conv whose
    'input channels' are o -- expressed with the same primitive so it stays differentiable."""
    KH, KW = wt.shape[0], wt.shape[1]
    H, W = x.shape[2], x.shape[3]
    yhw = ((H - 1) * 2 + KH, (W - 1) * 2 + KW)
    g = _Geom((2, 2), (0, 0), KH, KW, yhw, (H, W))
    return _Conv2dBwdData.apply(x, wt.transpose(2, 3).contiguous(), g, float(alpha))


# ----------------------------------------------------------------------------------------
# composable elementwise primitives (gradients of any order) for the regularised passes.  The path-length and R1 terms
# differentiate THROUGH a gradient (training_step.py:300-373), so their networks cannot use the once-differentiable fused
# layers; written with torch broadcasting ops, the per-layer chain x*s -> conv -> *d -> + noise*strength -> + b -> lrelu -> *sqrt2
# and its first and second derivative cost ~540 elementwise launches per PL step (profiles/r02_e_launch_sources_pl_step.txt).
# Every map in that chain is bilinear or piecewise linear, so three primitives closed under differentiation cover it, each
# ONE launch of an existing kernel:
#   scale_ch(x, s)[b,c,p] = x[b,c,p] s[b,c]        (tbg_bias_act_fwd_f32, out_scale)          d/dx = scale_ch(g, s), d/ds = dot_hw(g, x)
#   dot_hw(a, b)[b,c]     = sum_p a b               (tbg_bias_act_bwd_f32's per-plane sums)    d/da = scale_ch(b, g), d/db = scale_ch(a, g)
#   mask_mul(g, out)      = g * lrelu'(out) * gain  (tbg_bias_act_bwd_f32, dpre)               d/dg = mask_mul(., out); lrelu'' = 0
# and bias_act_c = lrelu(y + noise*strength + b)*sqrt2 (one tbg_bias_act_fwd_f32 launch) whose backward is mask_mul + sums.
# ----------------------------------------------------------------------------------------
class _ScaleCh(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, s):
        x, s = x.contiguous(), s.contiguous()
        ctx.save_for_backward(x, s)
        return bias_act_fwd_raw(x, N.epilogue(out_scale=s))

    @staticmethod
    def backward(ctx, g):
        x, s = ctx.saved_tensors
        dx = _ScaleCh.apply(g, s) if ctx.needs_input_grad[0] else None
        ds = _DotHW.apply(g, x) if ctx.needs_input_grad[1] else None
        return dx, ds


class _DotHW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        ctx.save_for_backward(a, b)
        _, _, _, _, pdy = bias_act_bwd_raw(a, b, N.epilogue(), want_dpre=False, want_db=False, want_dyy=True)
        return pdy.sum(dim=2) if pdy.shape[2] > 1 else pdy[:, :, 0]

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = g.contiguous()
        da = _ScaleCh.apply(b, g) if ctx.needs_input_grad[0] else None
        db = _ScaleCh.apply(a, g) if ctx.needs_input_grad[1] else None
        return da, db


class _MaskMul(torch.autograd.Function):
    """g * (out > 0 ? 1 : 0.2) * sqrt2: the LeakyReLU derivative taken from the saved OUTPUT (sign(out) = sign(pre))."""

    @staticmethod
    def forward(ctx, g, out):
        ctx.save_for_backward(out)
        _, dpre, _, _, _ = bias_act_bwd_raw(g.contiguous(), out, _lrelu_epi(), want_dpre=True, want_db=False)
        return dpre

    @staticmethod
    def backward(ctx, gg):
        (out,) = ctx.saved_tensors
        return (_MaskMul.apply(gg, out) if ctx.needs_input_grad[0] else None), None


class _BiasActC(torch.autograd.Function):
    """out = lrelu(y + noise*strength + b) * sqrt2   (noise.py:12-22 + bias_act.py:25-34), differentiable to any order."""

    @staticmethod
    def forward(ctx, y, noise, strength, b):
        y = y.contiguous()
        out = bias_act_fwd_raw(y, _lrelu_epi(bias=b, noise=noise, strength=strength))
        ctx.save_for_backward(out, noise)
        return out

    @staticmethod
    def backward(ctx, dout):
        out, noise = ctx.saved_tensors
        dpre = _MaskMul.apply(dout, out)
        dstrength = None
        params = not FLAGS.no_filter_grads  # a recorded inner gradient (R1 / path length) ends at the activations: no parameter sums
        if noise is not None and ctx.needs_input_grad[2] and params:
            dstrength = (dpre.sum(dim=1, keepdim=True) * noise).sum()
        db = dpre.sum(dim=(0, 2, 3)) if (ctx.needs_input_grad[3] and params) else None
        return (dpre if ctx.needs_input_grad[0] else None), None, dstrength, db


def scale_ch(x, s):
    """x [B,C,H,W] * s [B,C] (style modulation / demodulation of the composable path)."""
    return _ScaleCh.apply(x, s)


def bias_act_c(y, noise, strength, b):
    """lrelu(y + noise*strength + b) * sqrt2 with gradients of any order (noise / strength may be None)."""
    return _BiasActC.apply(y, noise, strength, b)


# ----------------------------------------------------------------------------------------
# fused first-order layers
# ----------------------------------------------------------------------------------------


def _units_conv_big(B, C_in, M, H, W) -> bool:
    """does a 3x3 stride-1 pad-1 convolution C_in -> M on B x H x W take tbg_conv2d_units (one tile per block, K whole)?"""
    fmt = _FMT[_TLS.compute]
    if not TUNING.use_units or fmt == FMT_F32:
        return False
    if not conv_units_ok(C_in, M, H, W, 3, 3, (1, 1), (1, 1), False, unit_planes(fmt)):
        return False
    # the tile choice (128- or 64-channel blocks) belongs to the library: ask it for the block count
    d = N.ConvDesc(B, C_in, M, H, W, H, W, 3, 3, 1, 1, 1, 1, 0, 0, M, 1)
    return N.lib().tbg_conv2d_units_blocks(C.byref(d), unit_planes(fmt)) >= TUNING.units_min_blocks


def _small_dot_ok(B, C_in, M, H, W) -> bool:
    d = N.ConvDesc(B, C_in, M, H, W, H, W, 3, 3, 1, 1, 1, 1, 0, 0, M, 1)
    return N.lib().tbg_conv2d_units_small_dot_slots(C.byref(d), unit_planes()) > 0


def _units_conv(B, C_in, M, H, W, dot=False) -> bool:
    """does a 3x3 stride-1 pad-1 convolution C_in -> M on B x H x W read a unit tensor in the current arithmetic -- tbg_conv2d_units
    on the large maps, tbg_conv2d_units_small on the small ones (conv2d_units_raw picks)?  dot: the launch carries a fused dot
    product (the small-map kernel serves it only when its pixel tiles stay inside one sample)."""
    if _units_conv_big(B, C_in, M, H, W):
        return True
    return TUNING.use_units and _small_conv(B, C_in, M, H, W, H, W, 3) and (not dot or _small_dot_ok(B, C_in, M, H, W))


def _units_wgrad(I, O, H, W) -> bool:
    fmt = _FMT[_TLS.compute]
    return TUNING.use_units and fmt != FMT_F32 and wgrad_units_ok(O, I, H, W, H, W, 3, 3, (1, 1), (1, 1))




def _units_s2(B, C_in, M, Ht, Wt) -> bool:
    """does a 3x3 stride-2 pad-0 convolution C_in -> M of a B x Ht x Wt tensor take the phase-unit kernels (forward / data-gradient
    form tbg_conv2d_units_s2 AND the filter gradient tbg_conv2d_wgrad_units_s2) in the current arithmetic?"""
    fmt = _FMT[_TLS.compute]
    if not (TUNING.use_units and TUNING.use_units_s2) or fmt == FMT_F32 or Ht < 3 or Wt < 3:
        return False
    planes = unit_planes(fmt)
    Ho, Wo = (Ht - 3) // 2 + 1, (Wt - 3) // 2 + 1
    if not conv_units_s2_ok(C_in, M, Ht, Wt, planes) or not wgrad_units_s2_ok(M, C_in, Ho, Wo, Ht, Wt):
        return False
    d = N.ConvDesc(B, C_in, M, Ht, Wt, Ho, Wo, 3, 3, 2, 2, 0, 0, 0, 0, M, 1)
    return N.lib().tbg_conv2d_units_s2_blocks(C.byref(d), planes) >= TUNING.units_min_blocks




def _units_t2(B, C_in, M, H, W, Hout, Wout) -> bool:
    """does the 3x3 stride-2 transposed convolution C_in -> M of a B x H x W map take tbg_conv2d_units_t2 in the current arithmetic?"""
    fmt = _FMT[_TLS.compute]
    if not (TUNING.use_units and TUNING.use_units_t2) or fmt == FMT_F32 or not conv_units_t2_ok(C_in, M, unit_planes(fmt)):
        return False
    d = N.ConvDesc(B, C_in, M, H, W, Hout, Wout, 3, 3, 2, 2, 0, 0, 1, 0, M, 1)
    gate = TUNING.units_min_blocks_t2 if fmt == FMT_X3 else max(TUNING.units_min_blocks, TUNING.units_min_blocks_t2)
    return N.lib().tbg_conv2d_units_t2_blocks(C.byref(d), unit_planes(fmt)) >= gate


def _small_conv(B, C_in, M, Hin, Win, Hout, Wout, k, stride=(1, 1), transposed=False) -> bool:
    """does this convolution take tbg_conv2d_units_small in the current arithmetic?  (3x3 stride-1 pad-1, 1x1 with stride or
    transposed stride; launches of at most TUNING.small_max_blocks blocks)"""
    fmt = _FMT[_TLS.compute]
    if not TUNING.use_small or fmt == FMT_F32 or k not in (1, 3):
        return False
    d = N.ConvDesc(B, C_in, M, Hin, Win, Hout, Wout, k, k, stride[0], stride[1], k // 2, k // 2, int(transposed), 0, M, 1)
    return 0 < N.lib().tbg_conv2d_units_small_blocks(C.byref(d), unit_planes(fmt)) <= TUNING.small_max_blocks


class _ForceSink(UnitSink):
    """a sink the CALLER has already decided on (it knows its consumer): plain units(out), wanted whatever the geometry"""

    def __init__(self):
        self.scale, self.kind, self.O_next, self.produced = None, "s1", 0, None

    def wanted(self, B, Cc, H, W) -> bool:
        return _FMT[_TLS.compute] != FMT_F32 and Cc % 8 == 0


_FORCE_SINK = _ForceSink()


def _unit_tensor(data, like: torch.Tensor, planes=None) -> Optional[UnitTensor]:
    """re-wrap the flat buffer of a unit tensor saved by a forward pass.  The plane count is read off the buffer's SIZE, not off
    the current arithmetic (a forward in bf16 and a backward in f32x3 would otherwise read 3 planes from a 1-plane
    buffer); a tensor whose planes do not match the arithmetic now in force is not reused (None: the caller packs again)."""
    if data is None:
        return None
    B, Cc, H, W = like.shape
    per_plane = N.lib().tbg_units_bytes(B, Cc, H, W, 1) // 2
    have = data.numel() // per_plane
    want = unit_planes() if planes is None else planes
    if have * per_plane != data.numel() or have != want:
        return None
    return UnitTensor(data, B, Cc, H, W, have)


class _Bwd3x3:
    """backward launches of a 3x3 stride-1 pad-1 convolution, through unit tensors where they apply: DU = units(dpre * dscale)
    is written once -- by the fused bias_act backward (from_bias_act) or by the stand-alone producer -- and feeds the data
    gradient (tbg_conv2d_units on the transposed, flipped pack) AND the filter gradient (tbg_conv2d_wgrad_units, with XU =
    units(x * x_scale) from the forward pass or packed here); geometries the unit kernels do not take keep the NCHW launches."""

    def __init__(self, B, I, O, H, W, want_dx=True, want_dw=True, dot=False):
        self.I, self.O = I, O
        self.g = _Geom((1, 1), (1, 1), 3, 3, (H, W), (H, W))
        self.u_dx = want_dx and _units_conv(B, O, I, H, W, dot=dot)
        self.u_dw = want_dw and _units_wgrad(I, O, H, W)
        self.units = self.u_dx or self.u_dw
        # is the NCHW fp32 gradient needed at all?
        self.need_dpre = (want_dx and not self.u_dx) or (want_dw and not self.u_dw)
        self.dpre = self.dscale = self.DU = None

    def from_bias_act(self, dout, out_act, epi, dscale, **want):
        """bias / noise / LeakyReLU backward: returns (pdb, pdn, pdy); epi.out_scale must be dscale (or None)"""
        self.dscale = dscale
        if self.units:
            self.DU, self.dpre, pdb, pdn, pdy = bias_act_bwd_units_raw(dout, out_act, epi, want_dpre=self.need_dpre, **want)
        else:
            _, self.dpre, pdb, pdn, pdy = bias_act_bwd_raw(dout, out_act, epi, **want)
        return pdb, pdn, pdy

    def from_dpre(self, dpre, dscale=None):
        self.dpre, self.dscale = dpre, dscale
        if self.units:
            self.DU = units_pack(dpre, dscale)
        return self

    def dx(self, w, epi, dot=None, out=None):
        if self.u_dx:
            return conv2d_units_raw(self.DU, pack_filter(w, transpose=True, flip=True), self.I, epi=epi, dot=dot, out=out)
        return _bwd_data_launch(self.dpre, w, self.g, in_scale=self.dscale, epi=epi, dot=dot, out=out)

    def dw(self, x, XU, coef, x_scale=None, add=None, bias=None):
        if self.u_dw:
            XU = units_pack(x, x_scale) if XU is None else XU
            dw = torch.empty((3, 3, self.I, self.O), device=x.device, dtype=torch.float32)
            return wgrad_units_raw(self.DU, XU, dw, self.I * self.O, self.O, 1, coef, add=add, bias=bias)
        return _bwd_weight_launch(x, self.dpre, self.g, self.I, self.O, alpha=coef, x_scale=x_scale, dy_scale=self.dscale, add=add,
                                  bias=bias)


def _lrelu_epi(**kw):
    return N.epilogue(act=ACT_LRELU, slope=0.2, gain=SQRT2, **kw)


def _demod_backward(pdy, d, s, wsq, ds_conv):
    """Shared tail of the two modulated-conv backward passes (modulated_conv2d.py:78-82).
    d = rsqrt(s^2 @ wsq + eps);  dd = sum_p(dy*yy)/d;  t = dd d^3 = P d^2
    returns ds_total = ds_conv - s * (t @ wsq^T)   and   dwsq' = (s^2)^T @ t   (the -1/2 is folded into gamma)."""
    t = pdy.sum(dim=2) * d.square()
    ds = torch.addcmul(ds_conv, s, t @ wsq.t(), value=-1.0)
    return ds, s.square().t() @ t


class _ModConvFused(torch.autograd.Function):
    """out = lrelu(d * coef*conv(s*x, w) + noise*strength + b) * sqrt2   (3x3, SAME), d = demodulation of (s, w).
    modulated_conv2d.py:66-122 (activation-scaling form :94-96,:119-121) + noise.py + bias_act.py.  The demodulation
    coefficients and their gradient live inside the node: the filter-gradient launch adds the demodulation term
    2 coef^2 w * dwsq while it writes dW (tbg_conv2d_wgrad_ex_f32), so no extra pass over the filter is needed.
    Unit tensors: units(x * s) comes from the layer that produced x (take_units) or is packed here; with ``sink`` this launch's
    epilogue writes units(out * sink.scale) for the next layer (handed back through sink.produced)."""

    @staticmethod
    def forward(ctx, x, w, s, noise, strength, b, sink=None):
        KH, KW, I, O = w.shape
        coef = 1.0 / math.sqrt(KH * KW * I)
        assert x.shape[0] == s.shape[0] and s.shape[1] == I, "one style row per sample (a short batch must not reach the fused layers)"
        XU = take_units(x, s) if (x.is_contiguous() and s.is_contiguous()) else None
        x = x.contiguous(); s = s.contiguous()
        d, wsq = demod_coefs_raw(s, w.contiguous(), coef)
        epi = _lrelu_epi(out_scale=d, bias=b, noise=noise, strength=strength, alpha=coef)
        B, H, W = x.shape[0], x.shape[2], x.shape[3]
        if KH == 3 and _units_conv(B, I, O, H, W):
            # x * s exists ONCE as a unit tensor: this launch DMAs its halo tiles from it, and the filter gradient of the
            # backward pass consumes the same tensor (modulated_conv2d.py:94-96: both use exactly this product)
            XU = units_pack(x, s) if XU is None else XU
            out, U = conv2d_units_raw(XU, pack_filter(w, False, False), O, epi=epi, sink=sink or _NO_SINK)
        else:
            out, U = conv2d_raw(x, pack_filter(w, False, False), O, KH, KW, (H, W), (1, 1), (KH // 2, KW // 2), in_scale=s, epi=epi,
                                sink=sink or _NO_SINK)
            if XU is not None and not (KH == 3 and _units_wgrad(I, O, H, W)):
                XU = None  # nobody in the backward pass reads it
        ctx.save_for_backward(x, w, s, d, wsq, noise, strength, b, out, XU.data if (XU is not None and TUNING.save_units) else None)
        ctx.coef = coef
        if sink is not None:
            sink.produced = U
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout):
        x, w, s, d, wsq, noise, strength, b, out, xu = ctx.saved_tensors
        KH, KW, I, O = w.shape
        coef = ctx.coef
        epi = _lrelu_epi(out_scale=d, bias=b, noise=noise, strength=strength, alpha=1.0)
        g = _Geom((1, 1), (KH // 2, KW // 2), KH, KW, (x.shape[2], x.shape[3]), (out.shape[2], out.shape[3]))
        want_dw = ctx.needs_input_grad[1]  # frozen generator (projector.py: only the latent is optimised): no filter gradient
        if KH == 3:
            bw = _Bwd3x3(x.shape[0], I, O, x.shape[2], x.shape[3], want_dw=want_dw, dot=True)
            pdb, pdn, pdy = bw.from_bias_act(dout.contiguous(), out, epi, d, want_dn=True, want_dyy=True)  # units(dpre * d)
            dx, ds_conv = bw.dx(w, N.epilogue(alpha=coef, out_scale=s), dot=(x, None))  # (the style dot's partial slots)
            db, dstrength, ds, dwsq = modconv_bwd_smalls_raw(pdb, pdn, pdy, d, s, wsq, ds_conv)  # (dwsq needs ds_conv)
            dw = bw.dw(x, _unit_tensor(xu, x), coef, x_scale=s, add=(w, dwsq, -coef * coef)) if want_dw else None
            return dx, dw, ds, None, dstrength, db, None
        _, dpre, pdb, pdn, pdy = bias_act_bwd_raw(dout.contiguous(), out, epi, want_dn=True, want_dyy=True)
        dx, ds_conv = _bwd_data_launch(dpre, w, g, in_scale=d, epi=N.epilogue(alpha=coef, out_scale=s), dot=(x, None))
        db, dstrength, ds, dwsq = modconv_bwd_smalls_raw(pdb, pdn, pdy, d, s, wsq, ds_conv)
        dw = None
        if want_dw:
            dw = _bwd_weight_launch(x, dpre, g, I, O, alpha=coef, x_scale=s, dy_scale=d, add=(w, dwsq, -coef * coef))
        return dx, dw, ds, None, dstrength, db, None


class _ModConvUpFused(torch.autograd.Function):
    """out = lrelu(d * FIR(coef*convT_s2(s*x, flip w)) + noise*strength + b) * sqrt2.
    upfirdn_2d_v2.py:65-103 (upsample_conv_2d) + the same epilogue, the FIR pass carries the epilogue;
    demodulation inside the node as in _ModConvFused.  Unit tensors as in _ModConvFused (the sink rides on the FIR launch)."""

    @staticmethod
    def forward(ctx, x, w, s, noise, strength, b, sink=None):
        KH, KW, I, O = w.shape
        coef = 1.0 / math.sqrt(KH * KW * I)
        assert x.shape[0] == s.shape[0] and s.shape[1] == I, "one style row per sample (a short batch must not reach the fused layers)"
        XU = take_units(x, s) if (x.is_contiguous() and s.is_contiguous()) else None
        x = x.contiguous(); s = s.contiguous()
        d, wsq = demod_coefs_raw(s, w.contiguous(), coef)
        B, H, W = x.shape[0], x.shape[2], x.shape[3]
        # (a map nobody wrote units for -- the word encoder's 2 x 8 output -- keeps the NCHW kernel unless it is big enough for a
        # stand-alone pack launch's fixed cost to disappear in it)
        if KH == 3 and _units_t2(B, I, O, H, W, 2 * H + 1, 2 * W + 1) and (XU is not None or H * W >= 256 or not TUNING.unit_sinks):
            # x * s exists ONCE as a unit tensor: the transposed convolution DMAs its tiles from it, and the filter gradient of the
            # backward pass contracts the same tensor with the blur^T phase tensor
            XU = units_pack(x, s) if XU is None else XU
            y_up = conv2d_units_t2_raw(XU, pack_filter(w, False, False), O, (2 * H + 1, 2 * W + 1), flip=True, alpha=coef)
        else:
            y_up = conv2d_raw(x, pack_filter(w, False, False), O, KH, KW, (2 * H + 1, 2 * W + 1), (2, 2), (0, 0), transposed=True,
                              flip=True, in_scale=s, epi=N.epilogue(alpha=coef))
            if XU is not None and not (KH == 3 and _units_s2(B, O, I, 2 * H + 1, 2 * W + 1)):
                XU = None  # nobody in the backward pass reads it
        k = fir_kernel(x.device, gain=4.0)
        epi = _lrelu_epi(out_scale=d.reshape(-1), bias=b, noise=noise, strength=strength, alpha=1.0)
        out, U = upfirdn2d_raw(y_up, k, pad=(1, 1, 1, 1), epi=epi, sink=sink or _NO_SINK)
        ctx.save_for_backward(x, w, s, d, wsq, noise, strength, b, out, XU.data if (XU is not None and TUNING.save_units) else None)
        ctx.coef = coef
        if sink is not None:
            sink.produced = U
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout):
        x, w, s, d, wsq, noise, strength, b, out, xu = ctx.saved_tensors
        KH, KW, I, O = w.shape
        coef = ctx.coef
        H, W = x.shape[2], x.shape[3]
        epi = _lrelu_epi(out_scale=d, bias=b, noise=noise, strength=strength, alpha=1.0)
        _, dpre, pdb, pdn, pdy = bias_act_bwd_raw(dout.contiguous(), out, epi, want_dn=True, want_dyy=True)
        k = fir_kernel(x.device, gain=4.0)  # symmetric: flipped == itself
        wt = pack_filter(w, transpose=True, flip=True)
        T = KH * KW
        # the data gradient is a 3x3 stride-2 convolution O -> I of dy_up = blur^T(dpre * d) [B,O,2H+1,2W+1], the filter gradient
        # contracts dy_up with x * s: where the phase-unit kernels take the layer, the blur writes dy_up ONCE as a phase unit tensor
        # (no fp32 dy_up) and both launches DMA their tiles from it
        s2 = KH == 3 and _units_s2(x.shape[0], O, I, 2 * H + 1, 2 * W + 1)
        if s2:
            DYP = upfirdn2d_units_s2(dpre, k, pad=(2, 2, 2, 2), in_scale=d.reshape(-1))
            dx, ds_conv = conv2d_units_s2_raw(DYP, wt, I, epi=N.epilogue(alpha=coef, out_scale=s), dot=(x, None))
        else:
            dy_up = upfirdn2d_raw(dpre, k, pad=(2, 2, 2, 2), in_scale=d.reshape(-1))  # [B,O,2H+1,2W+1]
            dx, ds_conv = conv2d_raw(dy_up, wt, I, KH, KW, (H, W), (2, 2), (0, 0),
                                     epi=N.epilogue(alpha=coef, out_scale=s), dot=(x, None))
        db, dstrength, ds, dwsq = modconv_bwd_smalls_raw(pdb, pdn, pdy, d, s, wsq, ds_conv)
        dw = None
        if ctx.needs_input_grad[1]:
            dw = torch.empty_like(w)
            # dW_t[t][i][o] = sum x*s . dy_up shifted;  w = flip(w_t)  -> write tap t at T-1-t
            if s2:
                XU = _unit_tensor(xu, x)
                wgrad_units_s2_raw(XU if XU is not None else units_pack(x, s), DYP, dw, -I * O, 1, O, coef,
                                   out_offset=(T - 1) * I * O,
                                   add=(w, dwsq, -coef * coef))
            else:
                wgrad_raw(x, dy_up, KH, KW, (2, 2), (0, 0), dw, -I * O, 1, O, coef, s_scale=s, out_offset=(T - 1) * I * O,
                          add=(w, dwsq, -coef * coef))
        return dx, dw, ds, None, dstrength, db, None


class _ToRGBFused(torch.autograd.Function):
    """y = (coef*conv1x1(s*x, w) + b (+ skip)) * m.  to_rgb.py:28-33 (modconv without demod), the
    ``y = upsample(y) + torgb`` add of synthesis_block.py:152-153 and -- on the last block of a training step --
    mask_text_box (utils/utils.py:11-45: m = 1 on the columns of real characters, else 0) as the launch's epilogue; the
    backward launch masks dy while it stages it."""

    @staticmethod
    def forward(ctx, x, w, s, b, skip, colmask, mask_cw):
        _, _, I, O = w.shape
        coef = 1.0 / math.sqrt(I)
        assert x.shape[0] == s.shape[0] and s.shape[1] == I, "one style row per sample"
        x = x.contiguous()
        y = rgb_project_raw(x, w, O, s, b, None if skip is None else skip.contiguous(), coef, colmask=colmask, mask_cw=mask_cw)
        ctx.save_for_backward(x, w, s, colmask)
        ctx.has_skip = skip is not None
        ctx.coef, ctx.mask_cw = coef, mask_cw
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, w, s, colmask = ctx.saved_tensors
        _, _, I, O = w.shape
        dy = dy.contiguous()
        # two launches: the streaming pass (dx, per-chunk Gram and dy sums) and the small-tensor tail that sums the chunks
        if colmask is not None:
            dx, Gp, dysum, dy = rgb_backproject_raw(x, dy, w, s, ctx.coef, colmask=colmask, mask_cw=ctx.mask_cw, want_dym=True,
                                                    parts=True)
        else:
            dx, Gp, dysum = rgb_backproject_raw(x, dy, w, s, ctx.coef, want_dx=True, want_G=True, parts=True)  # G = sum_p x*dy
        ds, dw, db = torgb_bwd_smalls_raw(Gp, w.reshape(I, O), s, ctx.coef, dysum=dysum)
        return dx, dw.reshape(w.shape), ds, db, (dy if ctx.has_skip else None), None, None


class _ConvBiasActFused(torch.autograd.Function):
    """out = act(coef*conv(x, w) + b) * gain  (optionally (.. + residual)*res_scale with act linear).
    conv.py:51-73 + bias_act.py:25-34; discriminator.py:68-84 for the residual form."""

    @staticmethod
    def forward(ctx, x, w, b, residual, stride, pad, act, res_scale, role, out_mul=1.0, sink=None):
        """out_mul: a constant folded into the launch -- the conv scale of a linear layer, the gain of an lrelu layer
        (DiscriminatorBlock folds its 1/sqrt(2) into both branches so that no pass has to scale the sum or its gradient).
        sink: the convolution that consumes the result next (UnitSink): its unit tensor leaves this launch's epilogue; the
        tensor is handed back through sink.produced."""
        KH, KW, I, O = w.shape
        coef = 1.0 / math.sqrt(KH * KW * I)
        gain = SQRT2 if act == ACT_LRELU else 1.0
        if act == ACT_LRELU:
            gain *= out_mul
        else:
            coef *= out_mul
            assert b is None or out_mul == 1.0, "a bias would need the factor too"
        XU = take_units(x, None) if x.is_contiguous() else None
        x = x.contiguous()
        H, W = x.shape[2], x.shape[3]
        yhw = ((H + 2 * pad[0] - KH) // stride[0] + 1, (W + 2 * pad[1] - KW) // stride[1] + 1)
        epi = N.epilogue(alpha=coef, bias=b, act=act, gain=gain, residual=residual, res_scale=res_scale)
        ctx.s1_3x3 = KH == 3 and KW == 3 and stride == (1, 1) and pad == (1, 1)
        if ctx.s1_3x3 and _units_conv(x.shape[0], I, O, H, W):
            # units(x) exists once (written by the layer that produced x, or packed here): this launch's halo tiles and the
            # backward pass's filter gradient read it
            XU = units_pack(x) if XU is None else XU
            out, U = conv2d_units_raw(XU, pack_filter(w, False, False), O, epi=epi, sink=sink or _NO_SINK)
        else:
            out, U = conv2d_raw(x, pack_filter(w, False, False), O, KH, KW, yhw, stride, pad, epi=epi, sink=sink or _NO_SINK)
            if XU is not None and not (ctx.s1_3x3 and _units_wgrad(I, O, H, W)):
                XU = None  # nobody in the backward pass reads it
        ctx.save_for_backward(x, w, b, out if act == ACT_LRELU else None, XU.data if (XU is not None and TUNING.save_units) else None)
        ctx.cfgv = (stride, pad, act, res_scale, coef, residual is not None, yhw)
        ctx.role = role
        ctx.gain = gain
        if sink is not None:
            sink.produced = U
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout):
        x, w, b, out, xu = ctx.saved_tensors
        stride, pad, act, res_scale, coef, has_res, yhw = ctx.cfgv
        KH, KW, I, O = w.shape
        dout = dout.contiguous()
        h = _half(ctx.role, dout.shape[0])
        if h:  # G-loss pass over [fake; real]: only the leading h samples carry a gradient (FLAGS.d_first_half)
            assert FLAGS.skip_d_wgrad, "the first-half mode belongs to the pass that skips the discriminator's filter gradients"
            dout_f, x_f, out_f = dout, x, out
            dout, x, out = dout[:h], x[:h], (out[:h] if out is not None else None)
        dres = None
        bias_rider = None
        prune_w = FLAGS.skip_d_wgrad and ctx.role in ("d", "d_image")
        want_dx = ctx.needs_input_grad[0] and not (FLAGS.skip_image_grad and ctx.role == "d_image")
        bw = _Bwd3x3(dout.shape[0], I, O, x.shape[2], x.shape[3], want_dx=want_dx, want_dw=not prune_w) if ctx.s1_3x3 else None
        if act == ACT_LRELU:
            assert not has_res
            epi_b = N.epilogue(act=ACT_LRELU, slope=0.2, gain=ctx.gain, bias=b)
            if bw is not None:  # (writes units(dpre) for the launches below; the NCHW dpre only if one of them needs it)
                pdb, _, _ = bw.from_bias_act(dout, out, epi_b, None, want_db=b is not None)
                dpre = bw.dpre
            else:
                _, dpre, pdb, _, _ = bias_act_bwd_raw(dout, out, epi_b, want_db=b is not None)
            db = None
            if b is not None and not prune_w:
                thin_ = KH == 1 and KW == 1 and I <= 4 and stride == (1, 1)
                if thin_:
                    db = pdb.sum(dim=(0, 2))
                else:  # summed by the filter gradient's reduce launch below (tbg_wgrad_desc.bias_*): no reduction launch of its own
                    db = torch.empty(O, device=dout.device, dtype=torch.float32)
                    bias_rider = (pdb, db)
        elif has_res and res_scale == 1.0:  # (the sum's scale folded into both branches: the gradient passes through as it is)
            dpre = dout
            dres = dout_f if h else dout
            db = dpre.sum(dim=(0, 2, 3)) if b is not None else None
        elif has_res and h:  # the residual branch continues into another "d" node: full-size tensor, leading part written
            dres = _tail_empty(dout_f.shape, dout_f.device)
            dpre = torch.mul(dout, res_scale, out=dres[:h])
            db = dpre.sum(dim=(0, 2, 3)) if b is not None else None
        else:
            dpre = dout * res_scale if has_res else dout
            dres = dpre if has_res else None
            db = dpre.sum(dim=(0, 2, 3)) if b is not None else None
        g = _Geom(stride, pad, KH, KW, (x.shape[2], x.shape[3]), yhw)
        dx = None
        thin = KH == 1 and KW == 1 and I <= 4 and stride == (1, 1)  # fromRGB: streaming kernels, not MFMA tiles
        if bw is not None and act != ACT_LRELU:
            bw.from_dpre(dpre.contiguous())
        if ctx.needs_input_grad[0] and not (FLAGS.skip_image_grad and ctx.role == "d_image"):
            dx_out = None
            if h:
                dx = _tail_empty(x_f.shape, x_f.device)
                dx_out = dx[:h]
            # gradient of ANOTHER consumer of x, added by this launch's epilogue instead of by an elementwise pass of the autograd
            # engine (_ConvBiasActSkipFused: the discriminator block's skip branch)
            xres = getattr(ctx, "dx_residual", None)
            if xres is not None:
                assert not thin
                xres = xres[:h] if h else xres
            epi_dx = N.epilogue(alpha=coef, residual=xres)
            if thin:  # d(image)[b,c,p] = coef * sum_o w[c,o] dpre[b,o,p]
                r = rgb_project_raw(dpre, w.reshape(I, O).t().contiguous(), I, None, None, None, coef, out=dx_out)
            elif bw is not None:
                r = bw.dx(w, epi_dx, out=dx_out)
            else:
                r = _bwd_data_launch(dpre, w, g, epi=epi_dx, out=dx_out)
            dx = dx if h else r
        dw = None
        if not prune_w:
            if thin:  # G[b,o,c] = sum_p dpre[b,o,p] x[b,c,p]
                _, G = rgb_backproject_raw(dpre, x, None, None, 1.0, want_dx=False, want_G=True)
                dw = (coef * G.sum(dim=0).t()).reshape(w.shape).contiguous()
            elif bw is not None:
                dw = bw.dw(x, _unit_tensor(xu, x) if not h else None, coef, bias=bias_rider)
            else:
                dw = _bwd_weight_launch(x, dpre, g, I, O, alpha=coef, bias=bias_rider)
        else:
            db = None
        return dx, dw, db, dres, None, None, None, None, None, None, None


class _ConvBiasActSkipFused(torch.autograd.Function):
    """(t, xd) = (lrelu(coef * conv3x3(x, w) + b) * sqrt2,  FIR-decimate(x)): the two consumers of a DiscriminatorBlock's input
    (discriminator.py:68-84: conv_0 and the skip branch's blur + strided 1x1 convolution, whose blur is evaluated at the strided
    sites only) as ONE node, so that d(x) = conv^T(dt) + FIR^T(dxd) is formed by the data-gradient launch's epilogue (residual =
    FIR^T(dxd)) instead of two activation-sized tensors and an add of the autograd engine."""

    @staticmethod
    def forward(ctx, x, w, b, k, down, fpad, role):
        out = _ConvBiasActFused.forward(ctx, x, w, b, None, (1, 1), (1, 1), ACT_LRELU, 1.0, role)  # (saves its tensors on ctx)
        ctx.fir = (k, tuple(down), tuple(fpad))
        return out, upfirdn2d_raw(x.contiguous(), k, (1, 1), down, fpad)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dt, dxd):
        x = ctx.saved_tensors[0]
        k, down, pad = ctx.fir
        inH, inW = x.shape[2], x.shape[3]
        kH, kW = k.shape
        outW = (inW + pad[0] + pad[1] - kW) // down[0] + 1
        outH = (inH + pad[2] + pad[3] - kH) // down[1] + 1
        gpad = (kW - pad[0] - 1, inW - outW * down[0] + pad[0], kH - pad[2] - 1, inH - outH * down[1] + pad[2])  # upfirdn_2d_v2.py:204-209
        h = _half(ctx.role, dxd.shape[0])
        if h:  # first-order pass over the leading samples only: the rest of the tensor is never read
            dxs = _tail_empty(tuple(x.shape), x.device)
            upfirdn2d_raw(dxd.contiguous()[:h], _flipped_fir(k), down, (1, 1), gpad, out=dxs[:h])
        else:
            dxs = upfirdn2d_raw(dxd.contiguous(), _flipped_fir(k), down, (1, 1), gpad)
        ctx.dx_residual = dxs
        try:
            dx, dw, db = _ConvBiasActFused.backward(ctx, dt)[:3]
        finally:
            ctx.dx_residual = None
        return dx, dw, db, None, None, None, None


def conv_bias_act_skip_fused(x, w, b, k, down, fpad, role=None):
    """(lrelu(coef * conv3x3(x, w) + b) * sqrt2, upfirdn2d(x, k, down=down, pad=fpad)) with one gradient launch chain for x."""
    return _ConvBiasActSkipFused.apply(x, w, b, k, tuple(down), tuple(fpad), role)




class _BlurConvS2Fused(torch.autograd.Function):
    """out = lrelu(coef * conv_s2(blur(x), w) + b) * gain: conv_downsample_2d (upfirdn_2d_v2.py:106-113: FIR with pad (2,3), then the
    3x3 stride-2 convolution) + bias_act.py:25-34, with the blurred tensor existing ONLY as a phase unit tensor: the FIR launch is
    the producer (tbg_upfirdn2d_units_s2_f32), the strided convolution and -- in the backward pass -- its filter gradient DMA their
    tiles from it (tbg_conv2d_units_s2 / tbg_conv2d_wgrad_units_s2).  The data gradient keeps the NCHW transposed kernel + the
    FIR's adjoint."""

    @staticmethod
    def forward(ctx, x, w, b, role, out_mul):
        KH, KW, I, O = w.shape
        coef = 1.0 / math.sqrt(KH * KW * I)
        gain = SQRT2 * out_mul
        x = x.contiguous()
        k = fir_kernel(x.device, 1.0)
        TP = upfirdn2d_units_s2(x, k, pad=(2, 3, 2, 3))
        out = conv2d_units_s2_raw(TP, pack_filter(w, False, False), O, epi=N.epilogue(alpha=coef, bias=b, act=ACT_LRELU, gain=gain))
        ctx.save_for_backward(w, b, out, TP.data)
        ctx.meta = (tuple(x.shape), tuple(TP[1:]), coef, gain, role)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout):
        w, b, out, tp = ctx.saved_tensors
        xshape, tpm, coef, gain, role = ctx.meta
        B, I, H, W = xshape
        O = w.shape[3]
        _, _, Ht, Wt, Ho, Wo, _ = tpm
        dout = dout.contiguous()
        h = _half(role, B)
        if h:  # G-loss pass over [fake; real]: only the leading h samples carry a gradient (FLAGS.d_first_half)
            assert FLAGS.skip_d_wgrad, "the first-half mode belongs to the pass that skips the discriminator's filter gradients"
            dout, out = dout[:h], out[:h]
        prune_w = FLAGS.skip_d_wgrad and role in ("d", "d_image")
        want_dx = ctx.needs_input_grad[0]
        epi_b = N.epilogue(act=ACT_LRELU, slope=0.2, gain=gain, bias=b)
        DU = None
        # the data gradient (a transposed convolution O -> I of dpre) reads units(dpre) too where tbg_conv2d_units_t2 takes it: the
        # NCHW dpre is then never written
        t2 = want_dx and _units_t2(dout.shape[0], O, I, Ho, Wo, Ht, Wt)
        if not prune_w or t2:  # units(dpre) for the filter gradient / the unit-tensor data gradient
            DU, dpre, pdb, _, _ = bias_act_bwd_units_raw(dout, out, epi_b, want_dpre=want_dx and not t2, want_db=b is not None)
        else:
            _, dpre, pdb, _, _ = bias_act_bwd_raw(dout, out, epi_b, want_db=b is not None)
        db = torch.empty(O, device=dout.device, dtype=torch.float32) if (b is not None and not prune_w) else None
        dx = None
        if want_dx:
            g = _Geom((2, 2), (0, 0), 3, 3, (Ht, Wt), (Ho, Wo))
            if t2:
                dtb = conv2d_units_t2_raw(DU, pack_filter(w, transpose=True, flip=False), I, (Ht, Wt), flip=False, alpha=coef)
            else:
                dtb = _bwd_data_launch(dpre, w, g, alpha=coef)  # d(blurred tensor) [h | B, I, Ht, Wt]
            k = fir_kernel(dout.device, 1.0)
            gpad = (4 - 2 - 1, W - Wt + 2, 4 - 2 - 1, H - Ht + 2)  # upfirdn_2d_v2.py:204-209 for up = down = 1, pad (2, 3)
            if h:
                dx = _tail_empty(xshape, dout.device)
                upfirdn2d_raw(dtb, _flipped_fir(k), pad=gpad, out=dx[:h])
            else:
                dx = upfirdn2d_raw(dtb, _flipped_fir(k), pad=gpad)
        dw = None
        if not prune_w:
            dw = torch.empty_like(w)
            wgrad_units_s2_raw(DU, PhaseUnitTensor(tp, *tpm), dw, I * O, O, 1, coef, bias=(pdb, db) if db is not None else None)
        else:
            db = None
        return dx, dw, db, None, None


def blur_conv_s2_units(B, I, O, H, W) -> bool:
    """does DiscriminatorBlock's blur (pad 2,3) + 3x3 stride-2 convolution I -> O of a B x H x W map take _BlurConvS2Fused?"""
    return _units_s2(B, I, O, H + 2, W + 2)


def blur_conv_s2_fused(x, w, b, role=None, out_mul=1.0):
    """lrelu(conv_downsample_2d(x, w) + b) * sqrt2 * out_mul through phase unit tensors (see blur_conv_s2_units)."""
    return _BlurConvS2Fused.apply(x, w, b, role, out_mul)


def _with_units(out, sink):
    """attach the unit tensor a fused layer wrote beside ``out`` (sink.produced) for the consumer named by ``sink``"""
    if sink is not None and sink.produced is not None:
        attach_units(out, sink.produced, sink.scale)
        sink.produced = None
    return out


def modconv_fused(x, w, s, noise, strength, b, sink: Optional[UnitSink] = None):
    """demodulated 3x3 modulated conv + noise + bias + lrelu (demodulation computed inside the node).
    sink: the layer that consumes the result next (UnitSink): its unit tensor is written by this layer's epilogue."""
    return _with_units(_ModConvFused.apply(x, w, s, noise, strength, b, sink), sink)


def modconv_up_fused(x, w, s, noise, strength, b, sink: Optional[UnitSink] = None):
    return _with_units(_ModConvUpFused.apply(x, w, s, noise, strength, b, sink), sink)


def torgb_fused(x, w, s, b, skip=None, colmask=None, mask_cw=0):
    """colmask [B, W // mask_cw] (float 0/1): multiply the output's column bands by it (mask_text_box fused in)."""
    return _ToRGBFused.apply(x, w, s, b, skip, colmask, int(mask_cw))


def conv_bias_act_fused(x, w, b, stride=(1, 1), pad=(0, 0), act=ACT_LRELU, residual=None, res_scale=1.0, role=None,
                        out_mul=1.0, sink: Optional[UnitSink] = None):
    """role: None (never pruned), "d" (a discriminator layer: its filter/bias gradients are skipped while
    FLAGS.skip_d_wgrad), "d_image" (the discriminator's fromRGB: additionally its input gradient is skipped while
    FLAGS.skip_image_grad).  sink: the convolution that consumes the result next (its unit tensor is written by this launch)."""
    return _with_units(_ConvBiasActFused.apply(x, w, b, residual, tuple(stride), tuple(pad), act, res_scale, role, float(out_mul), sink),
                       sink)


class _DemodCoefs(torch.autograd.Function):
    """d[b,o] = rsqrt(sum_i s^2 wsq + 1e-8), wsq = coef^2 sum_t w^2  (modulated_conv2d.py:78-82)."""

    @staticmethod
    def forward(ctx, s, w):
        KH, KW, I, O = w.shape
        coef = 1.0 / math.sqrt(KH * KW * I)
        d, wsq = demod_coefs_raw(s.contiguous(), w.contiguous(), coef)
        ctx.save_for_backward(s, w, wsq, d)
        ctx.coef = coef
        return d

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dd):
        s, w, wsq, d = ctx.saved_tensors
        t = dd * d * d * d  # [B,O];  dq = -1/2 t
        ds = -s * (t @ wsq.t())
        dwsq = -0.5 * (s.square().t() @ t)  # [I,O]
        dw = (2.0 * ctx.coef * ctx.coef) * w * dwsq[None, None]
        return ds, dw


def demod_coefs(s, w):
    return _DemodCoefs.apply(s, w)


class _MinibatchStd(torch.autograd.Function):
    """mini_batch_std.py:10-35 as one launch forward and one backward (first order: the R1 pass, which needs the second-
    order term, keeps the torch composition in models.minibatch_std).  parts > 1: the batch is that many independent
    batches laid end to end (the d-step's [fake; real]) -- the statistics groups never mix them, exactly as in the
    reference's separate discriminator calls."""

    @staticmethod
    def forward(ctx, x, group, parts, role):
        x = x.contiguous()
        B, Cc, H, W = x.shape
        assert B % parts == 0
        n = B // parts
        y = torch.empty((B, Cc + 1, H, W), device=x.device, dtype=torch.float32)
        for i in range(parts):
            N.check(N.lib().tbg_minibatch_std_fwd_f32(N.ptr(x[i * n:(i + 1) * n]), N.ptr(y[i * n:(i + 1) * n]), n, Cc, H * W,
                                                      group, N.stream()), "tbg_minibatch_std_fwd")
        ctx.save_for_backward(x)
        ctx.cfgv = (group, parts, role)
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        group, parts, role = ctx.cfgv
        B, Cc, H, W = x.shape
        n = B // parts
        dy = dy.contiguous()
        h = _half(role, B)
        assert h in (0, n), "first-half mode: the leading part is the differentiated one"
        dx = _tail_empty(x.shape, x.device) if h else torch.empty_like(x)
        for i in range(1 if h else parts):
            N.check(N.lib().tbg_minibatch_std_bwd_f32(N.ptr(x[i * n:(i + 1) * n]), N.ptr(dy[i * n:(i + 1) * n]),
                                                      N.ptr(dx[i * n:(i + 1) * n]), n, Cc, H * W, group, N.stream()),
                    "tbg_minibatch_std_bwd")
        return dx, None, None, None


def minibatch_std_fused(x, group=4, parts=1, role=None):
    return _MinibatchStd.apply(x, int(group), int(parts), role)


# ----------------------------------------------------------------------------------------
# equalised-LR dense layers (1-4 MFLOP each: one hand-written launch per direction instead of 3-5 library launches)
# ----------------------------------------------------------------------------------------


class _DenseBiasAct(torch.autograd.Function):
    """out = act(coef * x @ w + lrmul * b) [* sqrt2 if lrelu] + offset   (dense.py:23-29 + bias_act.py:25-34).
    lrelu is positively homogeneous, so the sqrt2 gain is folded into alpha/beta; one launch forward
    (tbg_dense_fwd_f32) and one backward (tbg_dense_bwd_f32: mask, dx, dw and db together)."""

    @staticmethod
    def forward(ctx, x, w, b, coef, lrmul, lrelu, offset):
        g = math.sqrt(2.0) if lrelu else 1.0
        x = x.contiguous()
        R, K = x.shape
        Nn = w.shape[1]
        out = torch.empty((R, Nn), device=x.device, dtype=torch.float32)
        N.check(N.lib().tbg_dense_fwd_f32(N.ptr(x), N.ptr(w.contiguous()), N.ptr(b), N.ptr(out), R, K, Nn, coef * g,
                                          lrmul * g, int(lrelu), offset, N.stream()), "tbg_dense_fwd")
        ctx.save_for_backward(x, w, out if lrelu else None)
        ctx.cfgv = (coef * g, lrmul * g, lrelu, offset)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout):
        x, w, out = ctx.saved_tensors
        alpha, beta, lrelu, offset = ctx.cfgv
        dout = dout.contiguous()
        R, K = x.shape
        Nn = w.shape[1]
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w, memory_format=torch.contiguous_format) if ctx.needs_input_grad[1] else None
        db = dout.new_empty(Nn) if ctx.needs_input_grad[2] else None
        if dx is not None or dw is not None or db is not None:
            N.check(N.lib().tbg_dense_bwd_f32(N.ptr(x), N.ptr(w.contiguous()), N.ptr(out), N.ptr(dout), N.ptr(dx), N.ptr(dw),
                                              N.ptr(db), R, K, Nn, alpha, beta, int(lrelu), offset, N.stream()),
                    "tbg_dense_bwd")
        return dx, dw, db, None, None, None, None


class _DenseBiasActGemm(torch.autograd.Function):
    """out = act(coef * x @ w + lrmul * b) [* sqrt2 if lrelu] + offset   (dense.py:23-29 + bias_act.py:25-34).
    Library-GEMM form for the one large layer (the discriminator head's dense_1, K = C*H*W = 32768): forward = addmm +
    leaky_relu, backward = mask + 2 GEMMs + 1 GEMV."""

    @staticmethod
    def forward(ctx, x, w, b, coef, lrmul, lrelu, offset):
        g = math.sqrt(2.0) if lrelu else 1.0
        out = torch.addmm(b, x, w, beta=lrmul * g, alpha=coef * g)
        if lrelu:
            out = torch.nn.functional.leaky_relu(out, 0.2)
        if offset != 0.0:
            out = out.add_(offset) if not lrelu else out + offset
        ctx.save_for_backward(x, w, out if lrelu else None)
        ctx.cfgv = (coef * g, lrmul * g, lrelu, offset)
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout):
        x, w, out = ctx.saved_tensors
        alpha, beta, lrelu, offset = ctx.cfgv
        dout = dout.contiguous()
        if lrelu:  # sign(out - offset) = sign(pre-activation)
            dout = torch.ops.aten.leaky_relu_backward(dout, out if offset == 0.0 else out - offset, 0.2, True)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.addmm(x, dout, w.t(), beta=0.0, alpha=alpha)
        if ctx.needs_input_grad[1]:
            dw = torch.addmm(w, x.t(), dout, beta=0.0, alpha=alpha)
        if ctx.needs_input_grad[2]:
            db = torch.addmv(b_like(dout), dout.t(), _ones(dout.device, dout.shape[0]), beta=0.0, alpha=beta)
        return dx, dw, db, None, None, None, None


def b_like(dout):
    return dout.new_empty(dout.shape[1])


class _StyleAffines(torch.autograd.Function):
    """s_l = coef * style[:, rows[l], :] @ W_l + b_l + 1 for every modulated layer l of the synthesis network at once
    (modulated_conv2d.py:52-56, 74-76; synthesis_block.py:120-156: layer l reads row rows[l] of the broadcast latents -- the
    first toRGB and the first conv share row 0): one launch forward, one backward (tbg_dense_multi_*), instead of 2 and
    3-4 library launches per layer.  d(style) is written row by row into ONE [B, n_rows, K] tensor (a row used by two
    layers: the second contribution goes through a [B, K] scratch and one add)."""

    @staticmethod
    def forward(ctx, style, coef, rows, *wb):
        style = style.contiguous()
        B, NR, K = style.shape
        L = len(rows)
        ws, bs = wb[:L], wb[L:]
        assert len(ws) == L and len(bs) == L and L <= N.DENSE_MAX_ITEMS and max(rows) < NR
        outs = [torch.empty((B, w.shape[1]), device=style.device, dtype=torch.float32) for w in ws]
        items = (N.DenseItem * L)()
        sp = N.ptr(style)
        for l, (w, b, o) in enumerate(zip(ws, bs, outs)):
            items[l] = N.DenseItem(x=sp + 4 * rows[l] * K, w=N.ptr(w), b=N.ptr(b), out=N.ptr(o), N=w.shape[1], ldx=NR * K)
        N.check(N.lib().tbg_dense_multi_fwd_f32(items, L, B, K, coef, 1.0, 1.0, N.stream()), "tbg_dense_multi_fwd")
        ctx.save_for_backward(style, *ws)
        ctx.cfgv = (coef, rows)
        return tuple(outs)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *douts):
        style, *ws = ctx.saved_tensors
        coef, rows = ctx.cfgv
        B, NR, K = style.shape
        L = len(rows)
        need_x = ctx.needs_input_grad[0]
        dstyle = torch.empty_like(style) if need_x else None
        dws = [torch.empty_like(w) if ctx.needs_input_grad[3 + l] else None for l, w in enumerate(ws)]
        dbs = [torch.empty(w.shape[1], device=w.device, dtype=torch.float32) if ctx.needs_input_grad[3 + L + l] else None
               for l, w in enumerate(ws)]
        items = (N.DenseItem * L)()
        sp, dsp = N.ptr(style), N.ptr(dstyle)
        keep, n, written, extra = [], 0, set(), []
        for l, w in enumerate(ws):
            d = douts[l]
            if d is None:  # an unused style (never on the training path)
                if dws[l] is not None: dws[l].zero_()
                if dbs[l] is not None: dbs[l].zero_()
                continue
            d = d.contiguous(); keep.append(d)
            if not (need_x or dws[l] is not None or dbs[l] is not None):
                continue
            xp, dxp, ldx = sp + 4 * rows[l] * K, None, NR * K
            if need_x and rows[l] not in written:
                dxp = dsp + 4 * rows[l] * K
                written.add(rows[l])
            elif need_x:  # second layer on the same latent row: x and dx share one row pitch in the kernel, so both go dense
                xrow = style[:, rows[l]].contiguous()
                scratch = torch.empty((B, K), device=style.device, dtype=torch.float32)
                keep.append(xrow); extra.append((rows[l], scratch))
                xp, dxp, ldx = N.ptr(xrow), N.ptr(scratch), K
            items[n] = N.DenseItem(x=xp, w=N.ptr(w), dout=N.ptr(d), dx=dxp, dw=N.ptr(dws[l]), db=N.ptr(dbs[l]),
                                   N=w.shape[1], ldx=ldx)
            n += 1
        if n:
            N.check(N.lib().tbg_dense_multi_bwd_f32(items, n, B, K, coef, 1.0, N.stream()), "tbg_dense_multi_bwd")
        if need_x:
            for r in range(NR):
                if r not in written:
                    dstyle[:, r].zero_()
            for r, scratch in extra:
                dstyle[:, r] += scratch
        return (dstyle, None, None, *dws, *dbs)


def style_affines(style, ws, bs, coef, rows=None):
    """style [B, NR, K]; ws[l] [K, I_l], bs[l] [I_l]; rows[l] = the latent row layer l reads (default l)
    ->  tuple of L tensors [B, I_l] = coef * style[:, rows[l]] @ ws[l] + bs[l] + 1."""
    rows = tuple(range(len(ws))) if rows is None else tuple(int(r) for r in rows)
    return _StyleAffines.apply(style, float(coef), rows, *ws, *bs)


_ONES = {}


def _ones(device, n):
    """constant ones vector (the bias gradient as a GEMV), one per (device, length) for the life of the process."""
    key = (device, n)
    if key not in _ONES:
        _ONES[key] = torch.ones(n, device=device, dtype=torch.float32)
    return _ONES[key]


def dense_bias_act(x, w, b, coef, lrmul=1.0, lrelu=False, offset=0.0):
    # measured per layer in graph replay (tools/bench_dense.py, profiles/r02_dense_forms.txt): the one-launch kernels halve
    # the lrelu layers (mapping network: 47 -> 24 us forward + backward); the linear style affines are a single library
    # launch forward already (4.7 us) and a wash backward, so they keep the library form
    fn = _DenseBiasAct if (lrelu and x.shape[1] <= TUNING.dense_small_k) else _DenseBiasActGemm
    return fn.apply(x, w, b, float(coef), float(lrmul), bool(lrelu), float(offset))


# ----------------------------------------------------------------------------------------
# frozen bidirectional LSTM layer (OCR encoder): batched GEMMs + one pointwise launch per step
# ----------------------------------------------------------------------------------------
class _FrozenBiLSTMLayer(torch.autograd.Function):
    """x [B,T,In] -> [B,T,D*H]; w_ih [D,4H,In], w_hh [D,4H,H], bias [D,4H] (= b_ih + b_hh); gradient w.r.t. x only
    (the OCR network is frozen).  Per step ONE launch for both directions (tbg_lstm_fused_*: recurrent projection + cell, round 6);
    hidden sizes the fused kernels do not take keep the round-2 pair (one bmm over the directions + tbg_lstm_step_*)."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, bias, w_hhT=None):
        B, T, In = x.shape
        D, H = w_hh.shape[0], w_hh.shape[2]
        dev = x.device
        x_tm = x.transpose(0, 1).reshape(T * B, In)  # time-major
        gx = torch.baddbmm(bias[:, None, :], x_tm.expand(D, T * B, In), w_ih.transpose(1, 2))  # [D, T*B, 4H]
        act = torch.empty((D, T, B, 4 * H), device=dev, dtype=torch.float32)
        cs = torch.empty((D, T, B, H), device=dev, dtype=torch.float32)
        seq = torch.empty((B, T, D * H), device=dev, dtype=torch.float32)
        fused = TUNING.fused_lstm and H % 32 == 0
        if fused:
            hT = torch.empty((2, D, H, B), device=dev, dtype=torch.float32)  # the state, transposed, in alternating buffers
            for s in range(T):
                N.check(PROFILE.launch("lstm_fused_fwd_kernel", 2.0 * D * B * 4 * H * H, lambda: N.lib().tbg_lstm_fused_fwd_f32(
                    N.ptr(gx), N.ptr(w_hh), N.ptr(hT[(s + 1) & 1]), N.ptr(hT[s & 1]), N.ptr(act), N.ptr(cs), N.ptr(seq), D, T, B, H,
                    s, N.stream())), "tbg_lstm_fused_fwd")
        else:
            h = torch.empty((D, B, H), device=dev, dtype=torch.float32)
            w_hhT_ = w_hh.transpose(1, 2)
            hw = None
            for s in range(T):
                if s > 0:
                    hw = torch.bmm(h, w_hhT_)
                N.check(N.lib().tbg_lstm_step_fwd_f32(N.ptr(gx), N.ptr(hw), N.ptr(act), N.ptr(cs), N.ptr(h), N.ptr(seq), D, T, B,
                                                      H, s, N.stream()), "tbg_lstm_step_fwd")
        ctx.save_for_backward(act, cs, w_ih, w_hh, w_hhT)
        ctx.dims = (B, T, In, D, H)
        ctx.fused = fused
        return seq

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout):
        act, cs, w_ih, w_hh, w_hhT = ctx.saved_tensors
        B, T, In, D, H = ctx.dims
        dev = dout.device
        dseq = dout.contiguous()
        dg = torch.empty((D, T, B, 4 * H), device=dev, dtype=torch.float32)
        dc = torch.empty((D, B, H), device=dev, dtype=torch.float32)
        if ctx.fused:
            w_hhT = w_hh.transpose(1, 2).contiguous() if w_hhT is None else w_hhT  # [D, H, 4H]
            dgT = torch.empty((2, D, 4 * H, B), device=dev, dtype=torch.float32)
            for s in range(T - 1, -1, -1):
                N.check(PROFILE.launch("lstm_fused_bwd_kernel", 2.0 * D * B * 4 * H * H, lambda: N.lib().tbg_lstm_fused_bwd_f32(
                    N.ptr(dseq), N.ptr(w_hhT), N.ptr(dgT[(s + 1) & 1]), N.ptr(dgT[s & 1]), N.ptr(dc), N.ptr(act), N.ptr(cs),
                    N.ptr(dg), D, T, B, H, s, int(s == T - 1), N.stream())), "tbg_lstm_fused_bwd")
        else:
            dgates = torch.empty((D, B, 4 * H), device=dev, dtype=torch.float32)
            dh_rec = None
            for s in range(T - 1, -1, -1):
                N.check(N.lib().tbg_lstm_step_bwd_f32(N.ptr(dseq), N.ptr(dh_rec), N.ptr(dc), N.ptr(act), N.ptr(cs), N.ptr(dg),
                                                      N.ptr(dgates), D, T, B, H, s, int(s == T - 1), N.stream()),
                        "tbg_lstm_step_bwd")
                if s > 0:
                    dh_rec = torch.bmm(dgates, w_hh)
        dx_tm = torch.bmm(dg.view(D, T * B, 4 * H), w_ih).sum(dim=0)  # [T*B, In]
        return dx_tm.view(T, B, In).transpose(0, 1), None, None, None, None


def frozen_bilstm_layer(x, w_ih, w_hh, bias, w_hhT=None):
    """w_hhT: the owner's cached [D, H, 4H] transpose of w_hh (the backward steps read it; made on the spot when absent)"""
    return _FrozenBiLSTMLayer.apply(x.contiguous(), w_ih, w_hh, bias, w_hhT)


# ----------------------------------------------------------------------------------------
# frozen attention decoder of the OCR branch (greedy feedback): 8 launches per step forward, 6 backward
# ----------------------------------------------------------------------------------------
class FrozenDecoderWeights(NamedTuple):
    """constants of the decoder, prepared once by the owner (AsterLikeOCRHip):
    w_enc [H,E] (att_enc), w_dT [H,H] = att_dec.weight^T, b_d [H], w_d [H,H] = att_dec.weight, v [H],
    etab [C+1,4H] = emb @ W_ih[:, E:]^T + b_ih + b_hh, w_ctx [4H,E], w_ctxT [E,4H], w_hh [4H,H], w_hhT [H,4H],
    w_o [C,H], w_oT [H,C], b_o [C]."""
    w_enc: torch.Tensor
    w_dT: torch.Tensor
    b_d: torch.Tensor
    w_d: torch.Tensor
    v: torch.Tensor
    etab: torch.Tensor
    w_ctx: torch.Tensor
    w_ctxT: torch.Tensor
    w_hh: torch.Tensor
    w_hhT: torch.Tensor
    w_o: torch.Tensor
    w_oT: torch.Tensor
    b_o: torch.Tensor
    w_cat: Optional[torch.Tensor] = None   # [4H, E + H] = [w_ctx | w_hh]: the cell's weights over its input [context; hidden]
    w_catT: Optional[torch.Tensor] = None  # [E + H, 4H]: its transpose (the fused backward steps)


class _FrozenAttnDecoder(torch.autograd.Function):
    """enc [B,T,E] -> logits [B,S,C] of the Bahdanau-attention LSTM decoder with greedy (argmax) feedback; gradient w.r.t.
    enc only.  Attention context and the LSTM cell's pointwise half are one HIP launch each per step
    (tbg_attn_ctx_*, tbg_lstm_step_* with D = 1); the dense parts stay library GEMMs."""

    @staticmethod
    def forward(ctx, enc, W: FrozenDecoderWeights, steps: int, go: int):
        enc = enc.contiguous()
        B, T, E = enc.shape
        H = W.w_d.shape[0]
        Cn = W.w_o.shape[0]
        dev = enc.device
        f32 = dict(device=dev, dtype=torch.float32)
        ep = torch.matmul(enc, W.w_enc.t())  # [B,T,H]
        qs = torch.empty((steps, B, H), **f32)
        a_all = torch.empty((steps, B, T), **f32)
        gbuf = torch.empty((1, steps, B, 4 * H), **f32)   # gate pre-activations, the layout tbg_lstm_step_fwd reads
        act = torch.empty((1, steps, B, 4 * H), **f32)
        cs = torch.empty((1, steps, B, H), **f32)
        lbuf = torch.empty((steps, B, Cn), **f32)
        fused = (TUNING.fused_decoder and W.w_cat is not None and H % 32 == 0 and (E + H) % 16 == 0 and T <= 64)
        ctx.fused = fused
        if fused:
            # two launches per step (tbg.h): the cell over [context; hidden] for every sample, then the per-sample half (logits,
            # greedy symbol, next embedding row, next query, next attention context); the state travels transposed
            K = E + H
            stT = torch.zeros((2, K, B), **f32)  # [context^T; hidden^T] of the step about to run, alternating
            L = N.lib()
            samp = lambda hT, lg, gxn, s1, out: N.check(PROFILE.launch("dec_sample_fwd_kernel", 0.0, lambda: L.tbg_dec_sample_fwd_f32(
                N.ptr(hT), N.ptr(W.w_oT), N.ptr(W.b_o), N.ptr(W.w_dT), N.ptr(W.b_d), N.ptr(ep), N.ptr(enc), N.ptr(W.v), N.ptr(W.etab),
                N.ptr(lg), N.ptr(gxn), N.ptr(qs[s1]) if gxn is not None else None, N.ptr(a_all[s1]) if gxn is not None else None,
                N.ptr(out[:E]) if gxn is not None else None, B, T, H, E, Cn, int(go), N.stream())), "tbg_dec_sample_fwd")
            samp(None, None, gbuf[0, 0], 0, stT[0])
            for s in range(steps):
                cur, nxt = stT[s & 1], stT[(s + 1) & 1]
                N.check(PROFILE.launch("lstm_fused_fwd_kernel", 2.0 * B * 4 * H * K, lambda: L.tbg_lstm_cell_fused_fwd_f32(
                    N.ptr(gbuf), N.ptr(W.w_cat), N.ptr(cur), N.ptr(nxt[E:]), N.ptr(act), N.ptr(cs), steps, B, H, K, s, N.stream())),
                    "tbg_lstm_cell_fused_fwd")
                last = s == steps - 1
                samp(nxt[E:], lbuf[s], None if last else gbuf[0, s + 1], min(s + 1, steps - 1), nxt)
        else:
            ctxs = torch.empty((steps, B, E), **f32)
            h = torch.zeros((1, B, H), **f32)
            prev = torch.full((B,), go, dtype=torch.long, device=dev)
            for s in range(steps):
                torch.addmm(W.b_d, h[0], W.w_dT, out=qs[s])
                N.check(N.lib().tbg_attn_ctx_fwd_f32(N.ptr(qs[s]), N.ptr(ep), N.ptr(enc), N.ptr(W.v), N.ptr(ctxs[s]),
                                                     N.ptr(a_all[s]), B, T, H, E, N.stream()), "tbg_attn_ctx_fwd")
                # gates = etab[prev] + ctx @ w_ctx^T + h @ w_hh^T, accumulated IN PLACE in the buffer the cell kernel reads (addmm with a
                # full-matrix `self` first copies it into the result: two device copies per step)
                gs = gbuf[0, s]
                torch.index_select(W.etab, 0, prev, out=gs)
                gs.addmm_(ctxs[s], W.w_ctxT)
                gs.addmm_(h[0], W.w_hhT)
                N.check(N.lib().tbg_lstm_step_fwd_f32(N.ptr(gbuf), None, N.ptr(act), N.ptr(cs), N.ptr(h), None, 1, steps, B, H, s,
                                                      N.stream()), "tbg_lstm_step_fwd")
                torch.addmm(W.b_o, h[0], W.w_oT, out=lbuf[s])
                prev = lbuf[s].argmax(dim=1)  # greedy feedback (non-differentiable, as in the TF decoder)
        ctx.save_for_backward(enc, ep, qs, a_all, act, cs)
        ctx.W, ctx.dims = W, (B, T, E, H, Cn, steps)
        return lbuf.transpose(0, 1)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dlogits):
        enc, ep, qs, a_all, act, cs = ctx.saved_tensors
        W = ctx.W
        B, T, E, H, Cn, steps = ctx.dims
        dev = enc.device
        f32 = dict(device=dev, dtype=torch.float32)
        dl = dlogits.transpose(0, 1).contiguous()  # [S,B,C]
        dep = torch.zeros((B, T, H), **f32)
        dctxs = torch.empty((steps, B, E), **f32)
        if ctx.fused:
            K = E + H
            dlo = torch.mm(dl.view(steps * B, Cn), W.w_o).view(steps, B, H)  # d(hidden) through the logits, every step at once
            dc = torch.empty((B, H), **f32)
            dgT = torch.empty((4 * H, B), **f32)
            outT = torch.empty((K, B), **f32)
            L = N.lib()

            def samp(s, att, cell_s, first):  # attention part of step s (att) + cell part of step cell_s (or None)
                c = cell_s
                N.check(PROFILE.launch("dec_sample_bwd_kernel", 0.0, lambda: L.tbg_dec_sample_bwd_f32(
                    N.ptr(outT[:E]) if att else None, N.ptr(outT[E:]) if att else None, N.ptr(a_all[s]), N.ptr(qs[s]), N.ptr(ep),
                    N.ptr(enc), N.ptr(W.v), N.ptr(W.w_d), N.ptr(dep), N.ptr(dctxs[s]),
                    N.ptr(dlo[c]) if c is not None else None, N.ptr(act[0, c]) if c is not None else None,
                    N.ptr(cs[0, c]) if c is not None else None, N.ptr(cs[0, c - 1]) if (c is not None and c > 0) else None,
                    N.ptr(dc), N.ptr(dgT), B, T, H, E, int(first), N.stream())), "tbg_dec_sample_bwd")
            samp(steps - 1, False, steps - 1, True)
            for s in range(steps - 1, -1, -1):
                N.check(PROFILE.launch("rows_gemv_kernel", 2.0 * B * 4 * H * K, lambda: L.tbg_rows_gemv_t_f32(
                    N.ptr(dgT), N.ptr(W.w_catT), N.ptr(outT), 4 * H, K, B, N.stream())), "tbg_rows_gemv_t")
                samp(s, True, s - 1 if s > 0 else None, False)
        else:
            dgates = torch.empty((1, B, 4 * H), **f32)
            dc = torch.empty((1, B, H), **f32)
            dq = torch.empty((B, H), **f32)
            dh = torch.empty((1, B, H), **f32)
            dh_next = None
            for s in range(steps - 1, -1, -1):
                if dh_next is None:
                    torch.mm(dl[s], W.w_o, out=dh[0])
                else:  # dh = dh_next + dl[s] @ w_o, in place in dh_next's buffer (no copy into a second one)
                    dh_next.addmm_(dl[s], W.w_o)
                    dh = dh_next.view(1, B, H)
                N.check(N.lib().tbg_lstm_step_bwd_f32(None, N.ptr(dh), N.ptr(dc), N.ptr(act), N.ptr(cs), None, N.ptr(dgates), 1,
                                                      steps, B, H, s, int(s == steps - 1), N.stream()), "tbg_lstm_step_bwd")
                torch.mm(dgates[0], W.w_ctx, out=dctxs[s])
                N.check(N.lib().tbg_attn_ctx_bwd_f32(N.ptr(dctxs[s]), N.ptr(a_all[s]), N.ptr(qs[s]), N.ptr(ep), N.ptr(enc), N.ptr(W.v),
                                                     N.ptr(dq), N.ptr(dep), None, B, T, H, E, N.stream()),
                        "tbg_attn_ctx_bwd")
                if s > 0:  # h_{s-1} feeds the cell (W_hh) and the attention query (att_dec)
                    dh_next = torch.mm(dgates[0], W.w_hh).addmm_(dq, W.w_d)
        # d(enc) through the context sums: sum_s a_s (x) dctx_s as ONE batched GEMM over the images, + through enc_proj
        denc = torch.bmm(a_all.permute(1, 2, 0), dctxs.permute(1, 0, 2))  # [B,T,S] @ [B,S,E]
        denc.view(B * T, E).addmm_(dep.view(B * T, H), W.w_enc)
        return denc, None, None, None


def frozen_attn_decoder(enc, W: FrozenDecoderWeights, steps: int, go: int):
    return _FrozenAttnDecoder.apply(enc, W, steps, go)


# ----------------------------------------------------------------------------------------
# optimiser / EMA over flat buffers
# ----------------------------------------------------------------------------------------
def adam_tf_(theta, m, v, g, step, lr, beta1, beta2, eps):
    N.check(PROFILE.launch("adam_tf_kernel", 0.0, lambda: N.lib().tbg_adam_tf_f32(
        N.ptr(theta), N.ptr(m), N.ptr(v), N.ptr(g), theta.numel(), lr, beta1, beta2, eps, N.ptr(step), N.stream()),
        nbytes=28.0 * theta.numel()), "tbg_adam_tf")


def ema_lerp_(dst, src, beta):
    N.check(N.lib().tbg_ema_lerp_f32(N.ptr(dst), N.ptr(src), dst.numel(), beta, N.stream()), "tbg_ema_lerp")


# ----------------------------------------------------------------------------------------
# frozen convolution of the OCR branch (no weight gradient)
# ----------------------------------------------------------------------------------------
class _FrozenConv(torch.autograd.Function):
    """y = [relu]( conv(x, w) + b [+ residual] ) with constant (w, b); only d/dx and d/dresidual exist."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, relu, residual, packs):
        KH, KW, I, O = w.shape
        x = x.contiguous()
        H, W = x.shape[2], x.shape[3]
        yhw = ((H + 2 * pad[0] - KH) // stride[0] + 1, (W + 2 * pad[1] - KW) // stride[1] + 1)
        if residual is not None:
            residual = residual.contiguous()
        epi = N.epilogue(bias=b, residual=residual, res_first=1, act=ACT_LRELU if relu else ACT_LINEAR, slope=0.0,
                         gain=1.0)
        pf_fwd, pf_bwd = packs if packs is not None else frozen_conv_packs(w, stride)
        y = conv2d_raw(x, pf_fwd, O, KH, KW, yhw, stride, pad, epi=epi)
        ctx.save_for_backward(w, y if relu else None)
        ctx.cfgv = (stride, pad, relu, residual is not None, (H, W), yhw)
        ctx.pf_bwd = pf_bwd
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        w, y = ctx.saved_tensors
        stride, pad, relu, has_res, xhw, yhw = ctx.cfgv
        KH, KW, I, O = w.shape
        dy = dy.contiguous()
        if relu:  # mask only (no bias gradient: frozen) -> one flat elementwise pass; sign(y) = sign(pre-activation)
            dy = torch.ops.aten.threshold_backward(dy, y, 0.0)
        dx = None
        if ctx.needs_input_grad[0]:
            if stride == (1, 1):
                dx = conv2d_raw(dy, ctx.pf_bwd, I, KH, KW, xhw, (1, 1), (KH - 1 - pad[0], KW - 1 - pad[1]))
            else:
                assert KH == 1 and KW == 1 and pad == (0, 0), "strided OCR convolutions are 1x1"
                dx = conv2d_raw(dy, ctx.pf_bwd, I, 1, 1, xhw, stride, (0, 0), transposed=True)
        return dx, None, None, None, None, None, (dy if has_res else None), None


class FrozenConvConst:
    """Constants of one frozen convolution (+ folded BatchNorm): bias, geometry, forward and data-gradient packs."""
    __slots__ = ("b", "KH", "KW", "I", "O", "stride", "pad", "relu", "pf_fwd", "pf_bwd")

    def __init__(self, w, b, stride, pad, relu):
        self.KH, self.KW, self.I, self.O = (int(v) for v in w.shape)
        self.b, self.stride, self.pad, self.relu = b, tuple(stride), tuple(pad), bool(relu)
        self.pf_fwd, self.pf_bwd = frozen_conv_packs(w, self.stride)

    def out_hw(self, H, W):
        return ((H + 2 * self.pad[0] - self.KH) // self.stride[0] + 1, (W + 2 * self.pad[1] - self.KW) // self.stride[1] + 1)

    def _fwd_epi(self, residual):
        return N.epilogue(bias=self.b, residual=residual, res_first=1, act=ACT_LRELU if self.relu else ACT_LINEAR, slope=0.0, gain=1.0)

    def fwd(self, x, residual=None):
        return conv2d_raw(x, self.pf_fwd, self.O, self.KH, self.KW, self.out_hw(x.shape[2], x.shape[3]), self.stride, self.pad,
                          epi=self._fwd_epi(residual))

    def bwd(self, dy, xhw, residual=None, gate=None):
        """d/dx of the convolution, + residual, then zeroed where gate <= 0 (the ReLU in front of x), in the one launch."""
        epi = N.epilogue(residual=residual, res_first=1, gate=gate) if (residual is not None or gate is not None) else None
        if self.stride == (1, 1):
            return conv2d_raw(dy, self.pf_bwd, self.I, self.KH, self.KW, xhw, (1, 1),
                              (self.KH - 1 - self.pad[0], self.KW - 1 - self.pad[1]), epi=epi)
        assert self.KH == 1 and self.KW == 1 and self.pad == (0, 0), "strided OCR convolutions are 1x1"
        return conv2d_raw(dy, self.pf_bwd, self.I, 1, 1, xhw, self.stride, (0, 0), transposed=True, epi=epi)

    # ---- the same two launches with unit tensors either side (round 6): on the small maps tbg_conv2d_units_small reads units(x)
    # and the launch that produces a tensor writes the units its consumer reads (tbg_epilogue's sink) -- (fp32 | None, units | None)
    def small_fwd(self, B, H, W) -> bool:
        return (self.KH == self.KW and self.pad == (self.KH // 2, self.KW // 2) and
                _small_conv(B, self.I, self.O, H, W, *self.out_hw(H, W), self.KH, self.stride, False))

    def small_bwd(self, B, Ho, Wo, xhw) -> bool:
        return (self.KH == self.KW and self.pad == (self.KH // 2, self.KW // 2) and
                _small_conv(B, self.O, self.I, Ho, Wo, xhw[0], xhw[1], self.KH, self.stride, self.stride != (1, 1)))

    def fwd_u(self, x, XU, residual=None, want_units=False):
        B, H, W = x.shape[0], x.shape[2], x.shape[3]
        sink = _FORCE_SINK if want_units else _NO_SINK
        if self.small_fwd(B, H, W):
            XU = units_pack(x) if XU is None else XU
            return conv2d_small_raw(XU, self.pf_fwd, self.O, self.KH, self.out_hw(H, W), self.stride, False,
                                    epi=self._fwd_epi(residual), sink=sink)
        return conv2d_raw(x, self.pf_fwd, self.O, self.KH, self.KW, self.out_hw(H, W), self.stride, self.pad,
                          epi=self._fwd_epi(residual), sink=sink)

    def bwd_u(self, dy, DU, xhw, residual=None, gate=None, want_units=False, want_y=True):
        """dy: fp32 (may be None when DU is given and this launch takes the small-map kernel)"""
        epi = N.epilogue(residual=residual, res_first=1, gate=gate)
        sink = _FORCE_SINK if want_units else _NO_SINK
        B, Ho, Wo = (DU.B, DU.H, DU.W) if DU is not None else (dy.shape[0], dy.shape[2], dy.shape[3])
        if self.small_bwd(B, Ho, Wo, xhw):
            DU = units_pack(dy) if DU is None else DU
            return conv2d_small_raw(DU, self.pf_bwd, self.I, self.KH, xhw, self.stride, self.stride != (1, 1), epi=epi, sink=sink,
                                    want_y=want_y)
        assert dy is not None
        if self.stride == (1, 1):
            return conv2d_raw(dy, self.pf_bwd, self.I, self.KH, self.KW, xhw, (1, 1),
                              (self.KH - 1 - self.pad[0], self.KW - 1 - self.pad[1]), epi=epi, sink=sink)
        y = conv2d_raw(dy, self.pf_bwd, self.I, 1, 1, xhw, self.stride, (0, 0), transposed=True, epi=epi)
        return y, (units_pack(y) if want_units and _FORCE_SINK.wanted(B, self.I, *xhw) else None)


class _FrozenResNet(torch.autograd.Function):
    """The frozen OCR encoder's ResNet (stem + units of 1x1 -> 3x3 (+ 1x1 shortcut)) as ONE autograd node: constant
    weights, so the backward is a chain of data-gradient launches whose epilogues carry the residual sum and the ReLU gate
    of the unit in front -- no elementwise launches between them (they were 73 of the branch's ~630 launches).
    Round 6: every convolution on a small map is ONE launch of tbg_conv2d_units_small (no split-K slabs, no second half); the
    activations / gradients travel between them as unit tensors written by the producing launch's epilogue, the fp32 copy is
    written only where something reads it (residual sums, ReLU gates, a consumer on the NCHW kernel)."""

    @staticmethod
    def forward(ctx, x, stem, units):
        x = x.contiguous()
        B = x.shape[0]
        small = lambda convs, t: any(c is not None and c.small_fwd(B, t[0], t[1]) for c in convs)
        hw = stem.out_hw(x.shape[2], x.shape[3])
        y, YU = stem.fwd_u(x, None, want_units=small((units[0][0], units[0][2]), hw)) if units else (stem.fwd(x), None)
        acts = [x.shape[2:], y]  # y0
        for u, (c1, c2, short) in enumerate(units):
            xin, XU = acts[-1], YU
            sc = xin if short is None else short.fwd_u(xin, XU)[0]
            h1, HU = c1.fwd_u(xin, XU, want_units=small((c2,), c1.out_hw(xin.shape[2], xin.shape[3])))
            nxt = (units[u + 1][0], units[u + 1][2]) if u + 1 < len(units) else ()
            y, YU = c2.fwd_u(h1, HU, residual=sc, want_units=small(nxt, h1.shape[2:]))
            acts += [h1, y]
        ctx.consts = (stem, units)
        ctx.in_hw = tuple(acts[0])
        ctx.save_for_backward(*acts[1:])
        return acts[-1]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        stem, units = ctx.consts
        acts = ctx.saved_tensors  # y0, (h1, y) per unit
        B = dy.shape[0]
        hw = lambda t: tuple(t.shape[2:])
        # gradient at the last unit's pre-activation -- as a unit tensor too when the first data-gradient launches read one
        c1, c2, short = units[-1]
        yhw = hw(acts[-1])
        if c2.small_bwd(B, *yhw, hw(acts[-2])) or (short is not None and short.small_bwd(B, *yhw, hw(acts[-3]))):
            GU, g, _, _, _ = bias_act_bwd_units_raw(dy.contiguous(), acts[-1], N.epilogue(act=ACT_LRELU, slope=0.0, gain=1.0),
                                                    want_dpre=True, want_db=False)
        else:
            g, GU = torch.ops.aten.threshold_backward(dy.contiguous(), acts[-1], 0.0), None
        for u in range(len(units) - 1, -1, -1):
            c1, c2, short = units[u]
            xin, h1 = acts[2 * u], acts[2 * u + 1]
            xhw, ghw = hw(xin), hw(h1)
            t_units = c1.small_bwd(B, *ghw, xhw)  # t is read by c1's data gradient only
            t, TU = c2.bwd_u(g, GU, ghw, gate=h1, want_units=t_units, want_y=not t_units)
            res = g if short is None else short.bwd_u(g, GU, xhw)[0]
            if u > 0:
                p1, p2, pshort = units[u - 1]
                nxt = p2.small_bwd(B, *xhw, hw(acts[2 * u - 1])) or (pshort is not None and pshort.small_bwd(B, *xhw, hw(acts[2 * u - 2])))
            else:
                nxt = False  # (the stem's data gradient: 3 output channels on the 32 x 100 map, the NCHW kernel)
            g, GU = c1.bwd_u(t, TU, xhw, residual=res, gate=xin, want_units=nxt)  # xin = relu output of the unit (or stem) in front
        dx = stem.bwd(g, ctx.in_hw) if ctx.needs_input_grad[0] else None
        return dx, None, None


def frozen_resnet(x, stem: FrozenConvConst, units):
    """units: [(c1, c2, short | None)] of FrozenConvConst."""
    return _FrozenResNet.apply(x, stem, units)


def frozen_conv_packs(w, stride):
    """(forward, data-gradient) packed filters of a frozen convolution.  The caller that owns the constant weight keeps
    them (AsterLikeOCRHip caches them per layer) -- there is deliberately no global cache keyed by address."""
    stride = tuple(stride)
    return pack_filter(w, False, False), pack_filter(w, transpose=True, flip=(stride == (1, 1)))


def frozen_conv(x, w, b, stride=(1, 1), pad=(0, 0), relu=True, residual=None, packs=None):
    return _FrozenConv.apply(x, w, b, tuple(stride), tuple(pad), relu, residual, packs)
