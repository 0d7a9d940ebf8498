"""Fused layers that can be differentiated TWICE: the synthesis layers of the path-length pass (training_step.py:300-347).

The path-length term differentiates THROUGH a gradient, so its generator pass cannot use the once-differentiable fused
layers of ``ops``.  Rounds 1-3 ran it on composable primitives (x*s -> conv -> *d -> noise/bias/lrelu as four launches whose
gradients are again primitives): correct to any order, but every scale is a full elementwise pass and the second backward
walks ~25 small nodes per layer (profiles/r04_i_launch_sources_pl_step.txt: 237 launches in _ScaleChBackward alone).

Here every layer is TWO autograd nodes with hand-written gradients:

  A  out = act(d * L_w(s * x) + noise * strength + b)                 (modulated_conv2d.py:66-122, noise.py:12-22, bias_act.py:25-34)
       L_w = coef * conv3x3 (SAME), or coef * FIR(convT_s2) for the up-sampling layer (upfirdn_2d_v2.py:65-103); the
       demodulation d [B,O] is an INPUT (a few tiny tensor ops outside, differentiable by the framework).
  B  (dx, ds, dd) = grad of A with respect to (x, s, d) for a cotangent dout -- the node A's backward returns when the
       gradient itself is being recorded (create_graph=True).  With m = act'(out), p = dout * m, r = L_w^T(d * p):
           dx = s * r,   ds = sum_p x * r,   dd = sum_p p * yc          (yc = L_w(s * x), recovered from out)
     and B's own backward, for cotangents (gdx, gds, gdd), with u = s * gdx + gds * x, c = L_w(u):
           g_dout = m * (d * c + gdd * yc)            g_d = sum_p p * c - gdd * dd / d
           g_x    = gds * r                            g_s = sum_p gdx * r
           g_w    = W(u, d * p)                        g_out = (gdd / d) * dout       (W = the filter gradient of L_w)
     -- the dd term is differentiated through ``out`` (yc = (act^-1(out) - noise * strength - b) / d): node A's backward, which
     runs later in the same pass with g_out in its cotangent, delivers s * L_w^T(gdd * p), sum_p x * L_w^T(gdd * p) and
     W(s * x, gdd * p) inside the launches it issues anyway (see _ModLayer2Bwd); every other map is multilinear in its
     arguments except the LeakyReLU mask, whose derivative is zero almost everywhere.

Launches: A forward 1-2, B forward 3-4, B backward 6-8 (ONE convolution, ONE filter gradient, four elementwise passes:
tbg_axpby_planes_f32 x2, tbg_bias_act_bwd2_f32, one per-plane scale), A backward (first order) 3-4 -- against ~4 / ~8 / ~25 / ~8
on the composable primitives.

toRGB (to_rgb.py:28-33: 1x1 modulated convolution to 3 channels without demodulation, + bias + skip) gets the same pair on the
streaming kernels tbg_rgb_project_f32 / tbg_rgb_backproject_f32."""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import native as N
from . import ops
from .ops import (ACT_LRELU, FLAGS, _Geom, _bwd_data_launch, _bwd_weight_launch, _lrelu_epi, bias_act_bwd_raw,
                  bias_act_fwd_raw, conv2d_raw, fir_kernel, pack_filter, rgb_backproject_raw, rgb_project_raw,
                  torgb_bwd_smalls_raw, upfirdn2d_raw, wgrad_raw)


# (the A/B switch of this path is ops.TUNING.use_fused2)


# ----------------------------------------------------------------------------------------
# raw launches
# ----------------------------------------------------------------------------------------
def axpby_planes_raw(a, sa, b=None, sb=None, c=None, want_y=True):
    """y = sa[plane] * a + sb[plane] * b over [B, C, H, W] planes (sa / sb [B, C] or None = 1); with c: also sum_p c * a -> [B, C]."""
    B, Cc = a.shape[0], a.shape[1]
    HW = a.numel() // (B * Cc)
    nch = N.lib().tbg_bias_act_bwd_chunks(HW)
    y = torch.empty_like(a) if want_y else None
    part = torch.empty((B, Cc, nch), device=a.device, dtype=torch.float32) if c is not None else None
    N.check(ops.PROFILE.launch("axpby_planes_kernel", 0.0, lambda: N.lib().tbg_axpby_planes_f32(
        N.ptr(a), N.ptr(sa), N.ptr(b), N.ptr(sb), N.ptr(c), N.ptr(y), N.ptr(part), B * Cc, HW, N.stream()),
        nbytes=4.0 * a.numel() * (1 + int(b is not None) + int(c is not None) + int(want_y))), "tbg_axpby_planes")
    dot = None if part is None else (part.sum(dim=2) if nch > 1 else part[:, :, 0])
    return y, dot


def bias_act_bwd2_raw(c, out_act, dout, gdd, epi: N.Epilogue):
    """g_dout = m * (d * c + gdd * yd / d) and sum_p dout * m * c -> [B, M]   (see tbg.h)."""
    B, M = c.shape[0], c.shape[1]
    HW = c.numel() // (B * M)
    nch = N.lib().tbg_bias_act_bwd_chunks(HW)
    g = torch.empty_like(c)
    part = torch.empty((B, M, nch), device=c.device, dtype=torch.float32)
    N.check(ops.PROFILE.launch("bias_act_bwd2_kernel", 0.0, lambda: N.lib().tbg_bias_act_bwd2_f32(
        N.ptr(c), N.ptr(out_act), N.ptr(dout), N.ptr(gdd), N.ptr(g), N.ptr(part), B, M, HW, C.byref(epi), N.stream()),
        nbytes=4.0 * c.numel() * 4), "tbg_bias_act_bwd2")
    return g, (part.sum(dim=2) if nch > 1 else part[:, :, 0])


# ----------------------------------------------------------------------------------------
# the linear map L_w of a layer, its adjoint and its filter gradient as launches
# ----------------------------------------------------------------------------------------
class _Lin:
    """L_w = coef * conv (k x k, stride 1, SAME) or, up=True, FIR_4(coef * convT_s2(., flip w)) (gain-4 [1,3,3,1] blur, pad 1)."""

    def __init__(self, w, up, xhw):
        self.KH, self.KW, self.I, self.O = w.shape
        self.up = bool(up)
        self.coef = 1.0 / math.sqrt(self.KH * self.KW * self.I)
        self.H, self.W = xhw
        if self.up:
            assert self.KH == 3 and self.KW == 3
            self.yhw = (2 * self.H, 2 * self.W)
        else:
            self.yhw = (self.H, self.W)
            self.g = _Geom((1, 1), (self.KH // 2, self.KW // 2), self.KH, self.KW, xhw, xhw)

    def fwd(self, x, w, in_scale, act=False, **kw):
        """act(out_scale * L_w(in_scale * x) + noise * strength + bias); kw: out_scale, bias, noise, strength."""
        mk = (lambda **k: _lrelu_epi(**k)) if act else (lambda **k: N.epilogue(**k))
        if not self.up:
            return conv2d_raw(x, pack_filter(w, False, False), self.O, self.KH, self.KW, self.yhw, (1, 1),
                              (self.KH // 2, self.KW // 2), in_scale=in_scale, epi=mk(alpha=self.coef, **kw))
        H, W = self.H, self.W
        y_up = conv2d_raw(x, pack_filter(w, False, False), self.O, 3, 3, (2 * H + 1, 2 * W + 1), (2, 2), (0, 0), transposed=True,
                          flip=True, in_scale=in_scale, epi=N.epilogue(alpha=self.coef))
        if kw.get("out_scale") is not None:
            kw["out_scale"] = kw["out_scale"].reshape(-1)
        return upfirdn2d_raw(y_up, fir_kernel(x.device, gain=4.0), pad=(1, 1, 1, 1), epi=mk(alpha=1.0, **kw))

    def adj(self, y, w, in_scale, out_scale=None, dot=None):
        """(out_scale * L_w^T(in_scale * y), aux); dot = (t, out [B, I]): out = sum_p t * L_w^T(in_scale * y) (before out_scale).
        aux: the blurred gradient FIR^T(in_scale * y) of the up layer (its filter gradient contracts exactly that tensor)."""
        epi = N.epilogue(alpha=self.coef, out_scale=out_scale)
        if not self.up:
            return _bwd_data_launch(y, w, self.g, in_scale=in_scale, epi=epi, dot=dot), None
        dy_up = upfirdn2d_raw(y, fir_kernel(y.device, gain=4.0), pad=(2, 2, 2, 2),
                              in_scale=None if in_scale is None else in_scale.reshape(-1))  # symmetric taps: flipped == itself
        r = conv2d_raw(dy_up, pack_filter(w, transpose=True, flip=True), self.I, 3, 3, (self.H, self.W), (2, 2), (0, 0),
                       epi=epi, dot=dot)
        return r, dy_up

    def wgrad(self, x, w, y, x_scale, y_scale, aux=None):
        """d<L_w(x_scale * x), y_scale * y>/dw; aux = FIR^T(y_scale * y) if the caller has it (up layer)."""
        if not self.up:
            return _bwd_weight_launch(x, y, self.g, self.I, self.O, alpha=self.coef, x_scale=x_scale, dy_scale=y_scale)
        if aux is None:
            aux = upfirdn2d_raw(y, fir_kernel(y.device, gain=4.0), pad=(2, 2, 2, 2),
                                in_scale=None if y_scale is None else y_scale.reshape(-1))
        dw = torch.empty_like(w)
        T, I, O = 9, self.I, self.O
        # dW_t[t][i][o] = sum (x * x_scale) . aux shifted;  w = flip(w_t)  ->  tap t is written at T-1-t
        wgrad_raw(x, aux, 3, 3, (2, 2), (0, 0), dw, -I * O, 1, O, self.coef, s_scale=x_scale, out_offset=(T - 1) * I * O)
        return dw


def _act_epi(d, noise, strength, b):
    return _lrelu_epi(out_scale=d, bias=b, noise=noise, strength=strength, alpha=1.0)


# ----------------------------------------------------------------------------------------
# modulated convolution + noise + bias + LeakyReLU
# ----------------------------------------------------------------------------------------
class _ModLayer2(torch.autograd.Function):
    """A: out = lrelu(d * L_w(s * x) + noise * strength + b) * sqrt2, d [B,O] given."""

    @staticmethod
    def forward(ctx, x, w, s, d, noise, strength, b, up):
        assert x.is_contiguous() and w.is_contiguous() and s.is_contiguous() and d.is_contiguous()
        assert x.shape[0] == s.shape[0] == d.shape[0] and s.shape[1] == w.shape[2] and d.shape[1] == w.shape[3]
        L = _Lin(w, up, (x.shape[2], x.shape[3]))
        out = L.fwd(x, w, s, act=True, out_scale=d, bias=b, noise=noise, strength=strength)
        ctx.save_for_backward(x, w, s, d, noise, strength, b, out)
        ctx.up = up
        return out

    @staticmethod
    def backward(ctx, dout):
        x, w, s, d, noise, strength, b, out = ctx.saved_tensors
        dout = dout.contiguous()
        if torch.is_grad_enabled():  # the gradient is being recorded (path-length pass): a node that can be differentiated again
            if not FLAGS.no_filter_grads:
                raise RuntimeError('mode="fused2" records gradients with respect to activations / styles only and under '
                                   'ops.FLAGS.no_filter_grads (the path-length pass); use mode="composable" for any other '
                                   'recorded gradient')
            dx, ds, dd = _ModLayer2Bwd.apply(dout, out, x, w, s, d, noise, strength, b, ctx.up)
            return dx, None, ds, dd, None, None, None, None
        L = _Lin(w, ctx.up, (x.shape[2], x.shape[3]))
        _, p, pdb, pdn, pdy = bias_act_bwd_raw(dout, out, _act_epi(d, noise, strength, b), want_dn=noise is not None, want_dyy=True)
        ds = torch.empty_like(s)
        dx, aux = L.adj(p, w, d, out_scale=s, dot=(x, ds))
        dd = pdy.sum(dim=2) / d
        db = pdb.sum(dim=(0, 2)) if ctx.needs_input_grad[6] else None
        dstrength = pdn.sum() if (noise is not None and ctx.needs_input_grad[5]) else None
        dw = L.wgrad(x, w, p, s, d, aux) if (ctx.needs_input_grad[1] and not FLAGS.no_filter_grads) else None
        return dx, dw, ds, dd, None, dstrength, db, None


class _ModLayer2Bwd(torch.autograd.Function):
    """B: (dx, ds, dd) of A for the cotangent dout (see the module docstring); once more differentiable.

    dd = sum_p p * yc is differentiated THROUGH ``out``: yc = (act^-1(out) - noise * strength - b) / d, so
    d<gdd, dd>/d(out) = (gdd / d) * dout (the mask cancels: act^-1' = 1 / act') and node A's own backward -- which runs later in
    the same pass with that tensor added to its cotangent -- delivers the x, s and w parts of the term (s * L^T(gdd * p),
    sum_p x * L^T(gdd * p), W(s * x, gdd * p)) inside the convolution and filter-gradient launches it issues anyway; what it adds
    beyond them (its dd, db, dstrength see the extra cotangent too) is exactly minus the DIRECT derivatives of yc with respect
    to d, b and strength, returned here.  One convolution and one filter gradient per layer less than forming those parts here."""

    @staticmethod
    def forward(ctx, dout, out, x, w, s, d, noise, strength, b, up):
        L = _Lin(w, up, (x.shape[2], x.shape[3]))
        dout = dout.contiguous()
        _, p, pdb, pdn, pdy = bias_act_bwd_raw(dout, out, _act_epi(d, noise, strength, b), want_db=True, want_dn=noise is not None,
                                               want_dyy=True)
        ds = torch.empty_like(s)
        r, aux = L.adj(p, w, d, dot=(x, ds))
        dx = bias_act_fwd_raw(r, N.epilogue(out_scale=s))
        dd = pdy.sum(dim=2) / d
        sp = pdb.sum(dim=2)                                  # sum_p p        [B, O]
        sn = pdn.sum(dim=2) if noise is not None else None   # sum_p p * n    [B, O]
        ctx.save_for_backward(dout, out, x, w, s, d, noise, strength, b, p, r, aux, dd, sp, sn)
        ctx.up = up
        return dx, ds, dd

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gdx, gds, gdd):
        dout, out, x, w, s, d, noise, strength, b, p, r, aux, dd, sp, sn = ctx.saved_tensors
        L = _Lin(w, ctx.up, (x.shape[2], x.shape[3]))
        gdx, gds, gdd = gdx.contiguous(), gds.contiguous(), gdd.contiguous()
        u, _ = axpby_planes_raw(gdx, s, x, gds)                      # u = s * gdx + gds * x
        c = L.fwd(u, w, None)                                        # c = L_w(u)
        g_dout, pc = bias_act_bwd2_raw(c, out, dout, gdd, _act_epi(d, noise, strength, b))   # m * (d * c + gdd * yc), sum_p p * c
        gq = (gdd / d).contiguous()
        g_out = bias_act_fwd_raw(dout, N.epilogue(out_scale=gq))     # (gdd / d) * dout: the dd term, routed through node A
        g_x, gr = axpby_planes_raw(r, gds, None, None, c=gdx)        # g_x = gds * r, gr = sum_p gdx * r
        g_d = pc - gq * dd                                           # (- gdd * dd / d: the direct d-dependence of yc)
        g_b = -(gq * sp).sum(dim=0) if (b is not None and ctx.needs_input_grad[8]) else None
        g_str = -(gq * sn).sum() if (sn is not None and ctx.needs_input_grad[7]) else None
        g_w = L.wgrad(u, w, p, None, d, aux) if ctx.needs_input_grad[3] else None
        return g_dout, g_out, g_x, g_w, gr, g_d, None, g_str, g_b, None


def mod_layer2(x, w, s, d, noise, strength, b, up=False):
    """lrelu(d * L_w(s * x) + noise * strength + b) * sqrt2, differentiable twice (d: the demodulation coefficients [B, O])."""
    # contiguous copies are taken HERE, as recorded operations: a copy made inside forward() would be saved as a constant and cut
    # s / d out of the graph of the recorded gradient
    return _ModLayer2.apply(x.contiguous(), w.contiguous(), s.contiguous(), d.contiguous(), noise, strength, b, bool(up))


# ----------------------------------------------------------------------------------------
# toRGB: y = coef * conv1x1(s * x, w) + b (+ skip)
# ----------------------------------------------------------------------------------------
class _ToRGB2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, s, b, skip):
        _, _, I, O = w.shape
        assert x.is_contiguous() and s.is_contiguous()
        ctx.coef = 1.0 / math.sqrt(I)
        y = rgb_project_raw(x, w, O, s, b, skip, ctx.coef)
        ctx.save_for_backward(x, w, s)
        ctx.has_skip = skip is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, s = ctx.saved_tensors
        dy = dy.contiguous()
        dskip = dy if ctx.has_skip else None
        if torch.is_grad_enabled():
            if not FLAGS.no_filter_grads:
                raise RuntimeError('mode="fused2" records gradients under ops.FLAGS.no_filter_grads only; use mode="composable"')
            dx, ds = _ToRGB2Bwd.apply(dy, x, w, s)
            return dx, None, ds, None, dskip
        _, _, I, O = w.shape
        dx, G = rgb_backproject_raw(x, dy, w, s, ctx.coef)
        ds, dw = torgb_bwd_smalls_raw(G.contiguous(), w.reshape(I, O), s, ctx.coef)
        db = dy.sum(dim=(0, 2, 3)) if ctx.needs_input_grad[3] else None
        return dx, dw.reshape(w.shape), ds, db, dskip


class _ToRGB2Bwd(torch.autograd.Function):
    """dx = coef * s * sum_o w[c,o] dy[o];  ds = coef * sum_o w[c,o] G[b,c,o],  G = sum_p x[c] dy[o]."""

    @staticmethod
    def forward(ctx, dy, x, w, s):
        _, _, I, O = w.shape
        coef = 1.0 / math.sqrt(I)
        dx, G = rgb_backproject_raw(x, dy, w, s, coef)
        G = G.contiguous()
        ds, _ = torgb_bwd_smalls_raw(G, w.reshape(I, O), s, coef)
        ctx.save_for_backward(dy, x, w, s, G)
        ctx.coef = coef
        return dx, ds

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gdx, gds):
        dy, x, w, s, G = ctx.saved_tensors
        _, _, I, O = w.shape
        coef, w2 = ctx.coef, w.reshape(I, O)
        gdx, gds = gdx.contiguous(), gds.contiguous()
        # g_dy = coef * sum_c w[c,o] (s gdx + gds x): two projections, the second adds the first
        t = rgb_project_raw(gdx, w, O, s, None, None, coef)
        g_dy = rgb_project_raw(x, w, O, gds, None, t, coef)
        g_x, _ = rgb_backproject_raw(x, dy, w, gds, coef, want_dx=True, want_G=False)       # gds * r
        _, G2 = rgb_backproject_raw(gdx, dy, w, None, coef, want_dx=False, want_G=True)     # sum_p gdx[c] dy[o]
        g_s, dw_a = torgb_bwd_smalls_raw(G2.contiguous(), w2, s, coef)      # coef sum_o G2 w ; coef sum_b G2 s
        _, dw_b = torgb_bwd_smalls_raw(G, w2, gds, coef)                    #                   coef sum_b G gds
        return g_dy, g_x, (dw_a + dw_b).reshape(w.shape), g_s


def torgb2(x, w, s, b, skip=None):
    """coef * conv1x1(s * x, w) + b (+ skip), differentiable twice."""
    return _ToRGB2.apply(x.contiguous(), w.contiguous(), s.contiguous(), b, None if skip is None else skip.contiguous())


# ----------------------------------------------------------------------------------------
# style affines and demodulation coefficients of ALL layers in a handful of batched tensor ops (differentiable to any order by
# the framework).  Per layer they are four to six tiny launches forward and several times that in each derivative
# (profiles/r04_j_launch_sources_pl_step.txt: ~600 launches, ~3 ms of a path-length step in Pow / Mm / Addmm / Mul / Rsqrt nodes).
# ----------------------------------------------------------------------------------------
class _PadStack(torch.autograd.Function):
    """[L, R, width] <- the parameter tensors t_l ([R, n_l] or [n_l]) side by side, zero padded to ``width`` columns.  The
    parameters are never on the path of a recorded gradient (that path runs from the styles), so first order suffices."""

    @staticmethod
    def forward(ctx, width, *ts):
        R = ts[0].shape[0] if ts[0].dim() == 2 else 1
        out = ts[0].new_zeros((len(ts), R, width))
        for l, t in enumerate(ts):
            out[l, :, :t.shape[-1]].copy_(t.reshape(R, -1))
        ctx.shapes = [tuple(t.shape) for t in ts]
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        return (None, *[g[l, :, :shp[-1]].reshape(shp) for l, shp in enumerate(ctx.shapes)])


class _Wsq(torch.autograd.Function):
    """wsq[i,o] = coef^2 sum_taps w[.,.,i,o]^2   (the demodulation's filter norms, modulated_conv2d.py:78-82)."""

    @staticmethod
    def forward(ctx, w):
        KH, KW, I, O = w.shape
        ctx.c2 = 1.0 / (KH * KW * I)
        ctx.save_for_backward(w)
        return (w * w).sum(dim=(0, 1)).mul_(ctx.c2)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (w,) = ctx.saved_tensors
        return w * (g * (2.0 * ctx.c2))


_IDX = {}


def _index(device, values):
    """constant index vector, one per (device, values) for the life of the process (a host-to-device copy is not capturable)."""
    key = (device, tuple(int(v) for v in values))
    if key not in _IDX:
        # (first use of a path-length variant is an eager warm-up step, TrainingStep._graphed_step; a first use INSIDE a capture
        # would fail on the copy and leave a process-lifetime tensor in the graph's private pool -- make that loud)
        assert not (device.type == "cuda" and torch.cuda.is_current_stream_capturing()), \
            "ops2._index: constant index tensors must be created outside HIP-graph capture (run one eager step first)"
        _IDX[key] = torch.tensor(key[1], device=device, dtype=torch.long)
    return _IDX[key]


def synthesis_styles(style, rows, mod_ws, mod_bs, coef, demod_layers, conv_ws):
    """s_l = coef * style[:, rows[l]] @ mod_ws[l] + mod_bs[l] + 1 for every layer l (modulated_conv2d.py:74-76) and, for the layers
    listed in demod_layers, d_l = rsqrt(s_l^2 @ wsq_l + 1e-8) with wsq_l from conv_ws (:78-82).  -> (list of s_l, {l: d_l})."""
    K = style.shape[2]
    width = max(max(w.shape[1] for w in mod_ws), max(w.shape[3] for w in conv_ws))
    Wp = _PadStack.apply(width, *mod_ws)                      # [L, K, width]
    bp = _PadStack.apply(width, *mod_bs)                      # [L, 1, width]
    xr = style.index_select(1, _index(style.device, rows)).transpose(0, 1)           # [L, B, K]
    S = torch.baddbmm(bp + 1.0, xr, Wp, alpha=float(coef))    # [L, B, width]; padded columns = 1
    Sl = S.unbind(0)
    ss = [Sl[l][:, :w.shape[1]] for l, w in enumerate(mod_ws)]
    WSQ = _PadStack.apply(width, *[torch.nn.functional.pad(_Wsq.apply(w), (0, 0, 0, width - w.shape[2])) if w.shape[2] < width
                                   else _Wsq.apply(w) for w in conv_ws])    # [Ld, width, width], zero rows past I_l
    Sd = S.index_select(0, _index(style.device, demod_layers))
    D = torch.rsqrt(torch.bmm(Sd.square(), WSQ) + 1e-8)       # padded columns: 1e4, never read
    Dl = D.unbind(0)
    ds = {l: Dl[j][:, :conv_ws[j].shape[3]] for j, l in enumerate(demod_layers)}
    return ss, ds
