// HBM-bound kernels of libtbg_hip.so (gfx950): bias_act fwd/bwd, split-K slab epilogue, filter packing,
// Keras-semantics Adam, EMA lerp, demodulation coefficients.  (upfirdn2d lives in upfirdn.hip.)
#include "common.h"

// ============================================================================================
// bias_act forward / backward
// ============================================================================================
#define BA_CHUNK 8192  // elements of one (b,m) plane handled by one block

extern "C" int tbg_bias_act_bwd_chunks(int HW) { return HW < 1 ? 0 : (HW + BA_CHUNK - 1) / BA_CHUNK; }

struct BiasActP {
  const float *x;
  float *y;
  int B, M, HW;
  EpiK e;
};

__global__ __launch_bounds__(256) void bias_act_fwd_kernel(const BiasActP p) {
  const int plane = blockIdx.x;  // b*M + m
  const int b = plane / p.M, m = plane - b * p.M;
  const int p0 = blockIdx.y * BA_CHUNK;
  const int p1 = min(p0 + BA_CHUNK, p.HW);
  const float sc = p.e.alpha * (p.e.out_scale ? p.e.out_scale[plane] : 1.f);
  const float bias = p.e.bias ? p.e.bias[m] * p.e.bias_mul : 0.f;
  const float str = p.e.noise ? p.e.strength[0] : 0.f;
  const float *xin = p.x + (size_t)plane * p.HW;
  const float *nz = p.e.noise ? p.e.noise + (size_t)b * p.HW : nullptr;
  const float *rs = p.e.residual ? p.e.residual + (size_t)plane * p.HW : nullptr;
  float *yo = p.y + (size_t)plane * p.HW;
  const bool rf = rs && p.e.res_first;
  const float *gt = p.e.gate ? p.e.gate + (size_t)plane * p.HW : nullptr;
  if ((p.HW & 3) == 0) {
    for (int i = p0 + threadIdx.x * 4; i < p1; i += 1024) {
      float4 v = *reinterpret_cast<const float4 *>(xin + i);
      float4 n = nz ? *reinterpret_cast<const float4 *>(nz + i) : make_float4(0, 0, 0, 0);
      float4 r = rs ? *reinterpret_cast<const float4 *>(rs + i) : make_float4(0, 0, 0, 0);
      const float4 ra = rf ? r : make_float4(0, 0, 0, 0);
      float4 o;
      o.x = epi_act(p.e, v.x * sc + n.x * str + bias + ra.x);
      o.y = epi_act(p.e, v.y * sc + n.y * str + bias + ra.y);
      o.z = epi_act(p.e, v.z * sc + n.z * str + bias + ra.z);
      o.w = epi_act(p.e, v.w * sc + n.w * str + bias + ra.w);
      if (rs && !rf) {
        o.x = (o.x + r.x) * p.e.res_scale; o.y = (o.y + r.y) * p.e.res_scale;
        o.z = (o.z + r.z) * p.e.res_scale; o.w = (o.w + r.w) * p.e.res_scale;
      }
      if (gt) {
        const float4 g = *reinterpret_cast<const float4 *>(gt + i);
        o.x = g.x > 0.f ? o.x : 0.f; o.y = g.y > 0.f ? o.y : 0.f; o.z = g.z > 0.f ? o.z : 0.f; o.w = g.w > 0.f ? o.w : 0.f;
      }
      *reinterpret_cast<float4 *>(yo + i) = o;
    }
  } else {
    for (int i = p0 + threadIdx.x; i < p1; i += 256) {
      float o = epi_act(p.e, xin[i] * sc + (nz ? nz[i] * str : 0.f) + bias + (rf ? rs[i] : 0.f));
      if (rs && !rf) o = (o + rs[i]) * p.e.res_scale;
      if (gt) o = gt[i] > 0.f ? o : 0.f;
      yo[i] = o;
    }
  }
}

extern "C" int tbg_bias_act_fwd_f32(const float *x, float *y, int B, int M, int HW, const tbg_epilogue *epi,
                                    void *stream) {
  if (!x || !y || B < 1 || M < 1 || HW < 1 || !epi_valid(epi) || epi_has_sink(epi)) return TBG_EINVAL;
  if ((double)B * M * HW > 2147483647.0) return TBG_ERANGE;
  if (HW <= 1024 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0 && !(epi && epi->dot_aux))
    return tbg_slab_epilogue_f32(x, y, B, M, HW, 1, epi, stream);  // small planes: one flat pass
  BiasActP p{x, y, B, M, HW, make_epi(epi)};
  dim3 grid(B * M, tbg_bias_act_bwd_chunks(HW));
  hipLaunchKernelGGL(bias_act_fwd_kernel, grid, dim3(256), 0, tbg_stream(stream), p);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

// Split-K epilogue / small-plane forward: y[i] = epilogue(sum_s x[s*slab + i]) as ONE flat grid-stride pass (a block per
// (plane, chunk) is wasteful when planes hold a few hundred elements -- exactly the layers that split K).
struct SlabEpiP {
  const float *x;
  float *y;
  int B, M, HW, nslab;
  long long slab, total;
  EpiK e;
};

__global__ __launch_bounds__(256) void slab_epilogue_kernel(const SlabEpiP p) {
  const float str = p.e.noise ? p.e.strength[0] : 0.f;
  const bool rf = p.e.residual && p.e.res_first;
  const bool vec = (p.HW & 3) == 0;
  const long long n = vec ? p.total >> 2 : p.total;
  for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < n; q += (long long)gridDim.x * 256) {
    const long long i = vec ? q << 2 : q;
    const int plane = (int)(i / p.HW), pix = (int)(i - (long long)plane * p.HW);
    const int b = plane / p.M, m = plane - b * p.M;
    const float sc = p.e.alpha * (p.e.out_scale ? p.e.out_scale[plane] : 1.f);
    const float bias = p.e.bias ? p.e.bias[m] * p.e.bias_mul : 0.f;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    const int w = vec ? 4 : 1;
    if (vec) {  // 4 slabs in flight per lane, summed in slab order
      int s = 0;
      for (; s + 4 <= p.nslab; s += 4) {
        float4 t[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t[u] = *reinterpret_cast<const float4 *>(p.x + (size_t)(s + u) * p.slab + i);
#pragma unroll
        for (int u = 0; u < 4; ++u) { v[0] += t[u].x; v[1] += t[u].y; v[2] += t[u].z; v[3] += t[u].w; }
      }
      for (; s < p.nslab; ++s) {
        const float4 t = *reinterpret_cast<const float4 *>(p.x + (size_t)s * p.slab + i);
        v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
      }
    } else {
      for (int s = 0; s < p.nslab; ++s) v[0] += p.x[(size_t)s * p.slab + i];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (k < w) {
        float o = v[k] * sc + bias;
        if (p.e.noise) o += p.e.noise[(size_t)b * p.HW + pix + k] * str;
        if (rf) o += p.e.residual[i + k];
        o = epi_act(p.e, o);
        if (p.e.residual && !rf) o = (o + p.e.residual[i + k]) * p.e.res_scale;
        if (p.e.gate) o = p.e.gate[i + k] > 0.f ? o : 0.f;
        v[k] = o;
      }
    }
    if (vec) *reinterpret_cast<float4 *>(p.y + i) = make_float4(v[0], v[1], v[2], v[3]);
    else p.y[i] = v[0];
  }
}

extern "C" int tbg_slab_epilogue_f32(const float *x, float *y, int B, int M, int HW, int nslab, const tbg_epilogue *epi,
                                     void *stream) {
  if (!x || !y || B < 1 || M < 1 || HW < 1 || nslab < 1 || !epi_valid(epi) || (epi && epi->dot_aux) || epi_has_sink(epi)) return TBG_EINVAL;
  if ((double)B * M * HW > 2147483647.0) return TBG_ERANGE;
  if ((((uintptr_t)x | (uintptr_t)y) & 15) != 0) return TBG_EINVAL;
  SlabEpiP p{x, y, B, M, HW, nslab, (long long)B * M * HW, (long long)B * M * HW, make_epi(epi)};
  const long long work = (HW & 3) == 0 ? p.total >> 2 : p.total;
  long long blocks = (work + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(slab_epilogue_kernel, dim3((int)blocks), dim3(256), 0, tbg_stream(stream), p);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

struct BiasActBwdP {
  const float *dout, *out_act;
  float *dx, *dpre_out, *part_db, *part_dn, *part_dyy;
  int B, M, HW, nchunks;
  EpiK e;
};

__global__ __launch_bounds__(256) void bias_act_bwd_kernel(const BiasActBwdP p) {
  __shared__ float red[3][4];
  const int plane = blockIdx.x;
  const int b = plane / p.M, m = plane - b * p.M;
  const int p0 = blockIdx.y * BA_CHUNK;
  const int p1 = min(p0 + BA_CHUNK, p.HW);
  const float sc = p.e.alpha * (p.e.out_scale ? p.e.out_scale[plane] : 1.f);
  const float bias = p.e.bias ? p.e.bias[m] * p.e.bias_mul : 0.f;
  const float str = p.e.noise ? p.e.strength[0] : 0.f;
  const float gin = p.e.residual ? p.e.res_scale : 1.f;  // residual != NULL only flags "fused residual"
  const float g_pos = p.e.gain, g_neg = p.e.gain * (p.e.act == TBG_ACT_LRELU ? p.e.slope : 1.f);
  const float ig_pos = 1.f / g_pos, ig_neg = g_neg != 0.f ? 1.f / g_neg : 0.f;  // slope 0 = ReLU
  const float *dout = p.dout + (size_t)plane * p.HW;
  const float *oa = p.out_act + (size_t)plane * p.HW;
  const float *nz = p.e.noise ? p.e.noise + (size_t)b * p.HW : nullptr;
  float s_db = 0.f, s_dn = 0.f, s_dyy = 0.f;
  if ((p.HW & 3) == 0) {  // 16-byte loads / stores: 4 independent quads per lane and chunk in flight
    float *dxo = p.dx ? p.dx + (size_t)plane * p.HW : nullptr;
    float *dpo = p.dpre_out ? p.dpre_out + (size_t)plane * p.HW : nullptr;
#pragma unroll 4
    for (int i = p0 + threadIdx.x * 4; i < p1; i += 1024) {
      const float4 o4 = *reinterpret_cast<const float4 *>(oa + i);
      const float4 d4 = *reinterpret_cast<const float4 *>(dout + i);
      const float4 n4 = nz ? *reinterpret_cast<const float4 *>(nz + i) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float ov[4] = {o4.x, o4.y, o4.z, o4.w}, dv[4] = {d4.x, d4.y, d4.z, d4.w}, nv[4] = {n4.x, n4.y, n4.z, n4.w};
      float dp[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool pos = ov[e] > 0.f;
        dp[e] = dv[e] * gin * (pos ? g_pos : g_neg);
        const float pre = ov[e] * (pos ? ig_pos : ig_neg);
        s_db += dp[e];
        s_dn += dp[e] * nv[e];
        s_dyy += dp[e] * (pre - nv[e] * str - bias);
      }
      if (dxo) *reinterpret_cast<float4 *>(dxo + i) = make_float4(dp[0] * sc, dp[1] * sc, dp[2] * sc, dp[3] * sc);
      if (dpo) *reinterpret_cast<float4 *>(dpo + i) = make_float4(dp[0], dp[1], dp[2], dp[3]);
    }
  } else {
  for (int i = p0 + threadIdx.x; i < p1; i += 256) {
    const float o = oa[i];
    const bool pos = o > 0.f;
    const float dpre = dout[i] * gin * (pos ? g_pos : g_neg);
    const float n = nz ? nz[i] : 0.f;
    const float pre = o * (pos ? ig_pos : ig_neg);
    s_db += dpre;
    s_dn += dpre * n;
    s_dyy += dpre * (pre - n * str - bias);
    if (p.dx) p.dx[(size_t)plane * p.HW + i] = dpre * sc;
    if (p.dpre_out) p.dpre_out[(size_t)plane * p.HW + i] = dpre;
  }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  {  // the three wave sums together (common.h wave_tree_sum: 7 lane exchanges in six steps; three chains of six before)
    float t[4] = {s_db, s_dn, s_dyy, 0.f};
    wave_tree_sum<4, 4, 32>(t, lane);
    const int row = wave_tree_row<4>(lane);
    if ((lane & 15) == 0 && row < 3) red[row][wave] = t[0];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const size_t o = (size_t)plane * p.nchunks + blockIdx.y;
    if (p.part_db) p.part_db[o] = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    if (p.part_dn) p.part_dn[o] = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    if (p.part_dyy) p.part_dyy[o] = red[2][0] + red[2][1] + red[2][2] + red[2][3];
  }
}

// small planes (HW <= 1024: the low-resolution layers and the whole OCR branch): one WAVE per (b,m) plane, four planes per
// block, shuffle reductions only -- a 256-thread block per plane left most lanes idle and cost 16 us for a 2 MB tensor
__global__ __launch_bounds__(256) void bias_act_bwd_small_kernel(const BiasActBwdP p) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int plane = blockIdx.x * 4 + wave;
  if (plane >= p.B * p.M) return;
  const int b = plane / p.M, m = plane - b * p.M;
  const float sc = p.e.alpha * (p.e.out_scale ? p.e.out_scale[plane] : 1.f);
  const float bias = p.e.bias ? p.e.bias[m] * p.e.bias_mul : 0.f;
  const float str = p.e.noise ? p.e.strength[0] : 0.f;
  const float gin = p.e.residual ? p.e.res_scale : 1.f;
  const float g_pos = p.e.gain, g_neg = p.e.gain * (p.e.act == TBG_ACT_LRELU ? p.e.slope : 1.f);
  const float ig_pos = 1.f / g_pos, ig_neg = g_neg != 0.f ? 1.f / g_neg : 0.f;
  const float *dout = p.dout + (size_t)plane * p.HW;
  const float *oa = p.out_act + (size_t)plane * p.HW;
  const float *nz = p.e.noise ? p.e.noise + (size_t)b * p.HW : nullptr;
  float *dxo = p.dx ? p.dx + (size_t)plane * p.HW : nullptr;
  float *dpo = p.dpre_out ? p.dpre_out + (size_t)plane * p.HW : nullptr;
  float s_db = 0.f, s_dn = 0.f, s_dyy = 0.f;
  for (int i = lane; i < p.HW; i += 64) {
    const float o = oa[i];
    const bool pos = o > 0.f;
    const float dpre = dout[i] * gin * (pos ? g_pos : g_neg);
    const float n = nz ? nz[i] : 0.f;
    const float pre = o * (pos ? ig_pos : ig_neg);
    s_db += dpre;
    s_dn += dpre * n;
    s_dyy += dpre * (pre - n * str - bias);
    if (dxo) dxo[i] = dpre * sc;
    if (dpo) dpo[i] = dpre;
  }
  {  // the three wave sums together: lane 0 / 16 / 32 ends up with the total of db / dn / dyy
    float t[4] = {s_db, s_dn, s_dyy, 0.f};
    wave_tree_sum<4, 4, 32>(t, lane);
    const int row = wave_tree_row<4>(lane);
    float *dst = row == 0 ? p.part_db : row == 1 ? p.part_dn : row == 2 ? p.part_dyy : nullptr;
    if ((lane & 15) == 0 && dst) dst[plane] = t[0];
  }
}

extern "C" int tbg_bias_act_bwd_f32(const float *dout, const float *out_act, float *dx, float *dpre_out,
                                    float *part_db, float *part_dn, float *part_dyy, int B, int M, int HW,
                                    const tbg_epilogue *epi, void *stream) {
  if (!dout || !out_act || B < 1 || M < 1 || HW < 1 || !epi || !epi_valid(epi) || epi_has_sink(epi)) return TBG_EINVAL;
  if ((double)B * M * HW > 2147483647.0) return TBG_ERANGE;
  if (part_dn && !epi->noise) return TBG_EINVAL;
  if (epi->gate) return TBG_EINVAL;  // a forward-only epilogue term
  BiasActBwdP p{dout, out_act, dx, dpre_out, part_db, part_dn, part_dyy, B, M, HW, tbg_bias_act_bwd_chunks(HW), make_epi(epi)};
  if (HW <= 1024) {  // nchunks == 1: same partial-sum layout [B*M][1]
    hipLaunchKernelGGL(bias_act_bwd_small_kernel, dim3((B * M + 3) / 4), dim3(256), 0, tbg_stream(stream), p);
  } else {
    dim3 grid(B * M, p.nchunks);
    hipLaunchKernelGGL(bias_act_bwd_kernel, grid, dim3(256), 0, tbg_stream(stream), p);
  }
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

// ============================================================================================
// second-order pieces of the regularised passes (path length: training_step.py:300-347, R1: :349-373).  The inner gradient
// of a fused layer out = act(d * L_w(s * x) + noise * strength + b) is itself a function (dx, ds, dd) of (dout, x, w, s, d);
// its gradient needs two per-plane elementwise forms besides the convolution launches:
//   tbg_axpby_planes_f32:   y = sa[plane] * a + sb[plane] * b   (+ part[plane][chunk] = sum_i c * a)
//       u   = s * gdx + gds * x            (the operand of the forward map in the second-order pass)
//       g_x = gds * r + s * r2             and   sum_p gdx * r   (d/ds of dx = s * r)
//   tbg_bias_act_bwd2_f32:  with m = act'(out) * gain, p = dout * m, d = out_scale[plane], yd = pre(out) - noise*strength - b (= d * yc):
//       g_dout = m * (d * c + gdd[plane] * yd / d),   part[plane][chunk] = sum_i p * c        (d/dd of dx, ds via c = L_w(u))
// Both use the chunking of tbg_bias_act_bwd_f32 (tbg_bias_act_bwd_chunks(HW) partial sums per plane).
// ============================================================================================
struct AxpbyP {
  const float *a, *sa, *b, *sb, *c;
  float *y, *part;
  int planes, HW, nchunks;
};

__global__ __launch_bounds__(256) void axpby_planes_kernel(const AxpbyP p) {
  __shared__ float red[4];
  const int plane = blockIdx.x;
  const int p0 = blockIdx.y * BA_CHUNK;
  const int p1 = min(p0 + BA_CHUNK, p.HW);
  const float fa = p.sa ? p.sa[plane] : 1.f, fb = p.sb ? p.sb[plane] : 1.f;
  const float *a = p.a + (size_t)plane * p.HW;
  const float *b = p.b ? p.b + (size_t)plane * p.HW : nullptr;
  const float *c = p.c ? p.c + (size_t)plane * p.HW : nullptr;
  float *y = p.y ? p.y + (size_t)plane * p.HW : nullptr;
  float acc = 0.f;
  if ((p.HW & 3) == 0) {
#pragma unroll 4
    for (int i = p0 + threadIdx.x * 4; i < p1; i += 1024) {
      const float4 a4 = *reinterpret_cast<const float4 *>(a + i);
      const float4 b4 = b ? *reinterpret_cast<const float4 *>(b + i) : make_float4(0.f, 0.f, 0.f, 0.f);
      if (c) {
        const float4 c4 = *reinterpret_cast<const float4 *>(c + i);
        acc += c4.x * a4.x + c4.y * a4.y + c4.z * a4.z + c4.w * a4.w;
      }
      if (y) *reinterpret_cast<float4 *>(y + i) = make_float4(fa * a4.x + fb * b4.x, fa * a4.y + fb * b4.y,
                                                               fa * a4.z + fb * b4.z, fa * a4.w + fb * b4.w);
    }
  } else {
    for (int i = p0 + threadIdx.x; i < p1; i += 256) {
      const float av = a[i];
      if (c) acc += c[i] * av;
      if (y) y[i] = fa * av + fb * (b ? b[i] : 0.f);
    }
  }
  if (p.part) {
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) p.part[(size_t)plane * p.nchunks + blockIdx.y] = red[0] + red[1] + red[2] + red[3];
  }
}

extern "C" int tbg_axpby_planes_f32(const float *a, const float *sa, const float *b, const float *sb, const float *c, float *y,
                                    float *part, int planes, int HW, void *stream) {
  if (!a || planes < 1 || HW < 1 || (!y && !part) || (part && !c) || (sb && !b)) return TBG_EINVAL;
  if ((double)planes * HW > 2147483647.0) return TBG_ERANGE;
  if ((HW & 3) == 0 && ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)y) & 15) != 0)) return TBG_EINVAL;
  AxpbyP p{a, sa, b, sb, c, y, part, planes, HW, tbg_bias_act_bwd_chunks(HW)};
  hipLaunchKernelGGL(axpby_planes_kernel, dim3(planes, p.nchunks), dim3(256), 0, tbg_stream(stream), p);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

struct BiasActBwd2P {
  const float *c, *out_act, *dout, *gdd;
  float *g_dout, *part;
  int B, M, HW, nchunks;
  EpiK e;
};

__global__ __launch_bounds__(256) void bias_act_bwd2_kernel(const BiasActBwd2P p) {
  __shared__ float red[4];
  const int plane = blockIdx.x;
  const int b = plane / p.M, m = plane - b * p.M;
  const int p0 = blockIdx.y * BA_CHUNK;
  const int p1 = min(p0 + BA_CHUNK, p.HW);
  const float d = p.e.alpha * (p.e.out_scale ? p.e.out_scale[plane] : 1.f);
  const float gq = p.gdd ? p.gdd[plane] / d : 0.f;  // gdd * (yd / d)
  const float bias = p.e.bias ? p.e.bias[m] * p.e.bias_mul : 0.f;
  const float str = p.e.noise ? p.e.strength[0] : 0.f;
  const float g_pos = p.e.gain, g_neg = p.e.gain * (p.e.act == TBG_ACT_LRELU ? p.e.slope : 1.f);
  const float ig_pos = 1.f / g_pos, ig_neg = g_neg != 0.f ? 1.f / g_neg : 0.f;
  const float *cc = p.c + (size_t)plane * p.HW;
  const float *oa = p.out_act + (size_t)plane * p.HW;
  const float *dout = p.dout ? p.dout + (size_t)plane * p.HW : nullptr;
  const float *nz = p.e.noise ? p.e.noise + (size_t)b * p.HW : nullptr;
  float *go = p.g_dout + (size_t)plane * p.HW;
  float acc = 0.f;
  for (int i = p0 + threadIdx.x; i < p1; i += 256) {
    const float o = oa[i];
    const bool pos = o > 0.f;
    const float mk = pos ? g_pos : g_neg;
    const float cv = cc[i];
    const float yd = o * (pos ? ig_pos : ig_neg) - (nz ? nz[i] * str : 0.f) - bias;
    go[i] = mk * (d * cv + gq * yd);
    if (dout) acc += dout[i] * mk * cv;
  }
  if (p.part) {
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) p.part[(size_t)plane * p.nchunks + blockIdx.y] = red[0] + red[1] + red[2] + red[3];
  }
}

extern "C" int tbg_bias_act_bwd2_f32(const float *c, const float *out_act, const float *dout, const float *gdd, float *g_dout,
                                     float *part, int B, int M, int HW, const tbg_epilogue *epi, void *stream) {
  if (!c || !out_act || !g_dout || B < 1 || M < 1 || HW < 1 || !epi || !epi_valid(epi) || epi_has_sink(epi) || (part && !dout)) return TBG_EINVAL;
  if ((double)B * M * HW > 2147483647.0) return TBG_ERANGE;
  if (epi->gate || epi->residual) return TBG_EINVAL;
  BiasActBwd2P p{c, out_act, dout, gdd, g_dout, part, B, M, HW, tbg_bias_act_bwd_chunks(HW), make_epi(epi)};
  hipLaunchKernelGGL(bias_act_bwd2_kernel, dim3(B * M, p.nchunks), dim3(256), 0, tbg_stream(stream), p);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

// ============================================================================================
// filter packing for the MFMA convolutions: HWIO parameter [T][I][O] -> Wp[T][C/4][M][4]
//   transpose = 0: C = I (reduction), M = O  -- forward correlation
//   transpose = 1: C = O, M = I              -- data gradient (the transposed filter); flip reverses the taps
// A 16-byte unit holds 4 consecutive reduction channels of one output channel (zero padded past C): the conv
// kernel DMAs units straight into LDS and reads an MFMA A operand for 4 k-steps with one ds_read_b128.
// ============================================================================================
__global__ __launch_bounds__(256) void weight_pack_kernel(const float *__restrict__ src, float4 *__restrict__ dst,
                                                          int T, int I, int O, int transpose, int flip) {
  const int t = blockIdx.z;
  const int td = flip ? T - 1 - t : t;
  const int C = transpose ? O : I, M = transpose ? I : O;
  const int C4 = (C + 3) >> 2;
  const int c4 = blockIdx.y;
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  float v[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = 4 * c4 + e;
    const int i = transpose ? m : c, o = transpose ? c : m;
    v[e] = (c < C) ? src[((size_t)t * I + i) * O + o] : 0.f;
  }
  dst[((size_t)td * C4 + c4) * M + m] = make_float4(v[0], v[1], v[2], v[3]);
}

extern "C" long long tbg_weight_pack_floats(int T, int I, int O, int transpose) {
  if (T < 1 || I < 1 || O < 1) return -1;
  const long long C = transpose ? O : I, M = transpose ? I : O;
  return (long long)T * ((C + 3) / 4) * M * 4;
}

extern "C" int tbg_weight_pack_f32(const float *src, float *dst, int T, int I, int O, int transpose, int flip,
                                   void *stream) {
  if (!src || !dst || T < 1 || I < 1 || O < 1) return TBG_EINVAL;
  if ((reinterpret_cast<uintptr_t>(dst) & 15) != 0) return TBG_EINVAL;
  const int C = transpose ? O : I, M = transpose ? I : O;
  dim3 grid((M + 255) / 256, (C + 3) / 4, T);
  hipLaunchKernelGGL(weight_pack_kernel, grid, dim3(256), 0, tbg_stream(stream), src, reinterpret_cast<float4 *>(dst), T, I,
                     O, transpose, flip);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

// bf16 form of the packed filter (tbg_conv2d_bf16): Wp[T][ceil(C/8)][M][8] bf16 -- a 16-byte unit holds 8 consecutive
// reduction channels of one output channel (RNE rounding of the fp32 master weight, zero padded past C).
typedef __bf16 tbg_bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void weight_pack_bf16_kernel(const float *__restrict__ src, tbg_bf16x8 *__restrict__ dst,
                                                               int T, int I, int O, int transpose, int flip) {
  const int t = blockIdx.z;
  const int td = flip ? T - 1 - t : t;
  const int C = transpose ? O : I, M = transpose ? I : O;
  const int C8 = (C + 7) >> 3;
  const int c8 = blockIdx.y;
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  tbg_bf16x8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = 8 * c8 + e;
    const int i = transpose ? m : c, o = transpose ? c : m;
    v[e] = (__bf16)((c < C) ? src[((size_t)t * I + i) * O + o] : 0.f);
  }
  dst[((size_t)td * C8 + c8) * M + m] = v;
}

extern "C" long long tbg_weight_pack_bf16_bytes(int T, int I, int O, int transpose) {
  if (T < 1 || I < 1 || O < 1) return -1;
  const long long C = transpose ? O : I, M = transpose ? I : O;
  return (long long)T * ((C + 7) / 8) * M * 16;
}

extern "C" int tbg_weight_pack_bf16(const float *src, void *dst, int T, int I, int O, int transpose, int flip,
                                    void *stream) {
  if (!src || !dst || T < 1 || I < 1 || O < 1) return TBG_EINVAL;
  if ((reinterpret_cast<uintptr_t>(dst) & 15) != 0) return TBG_EINVAL;
  const int C = transpose ? O : I, M = transpose ? I : O;
  dim3 grid((M + 255) / 256, (C + 7) / 8, T);
  hipLaunchKernelGGL(weight_pack_bf16_kernel, grid, dim3(256), 0, tbg_stream(stream), src,
                     reinterpret_cast<tbg_bf16x8 *>(dst), T, I, O, transpose, flip);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

// f32x3 form of the packed filter (tbg_conv2d_x3): THREE bf16 planes Wp[plane][T][ceil(C/8)][M][8], plane 0 = hi = RNE(w),
// plane 1 = mid = RNE(w - hi), plane 2 = lo = RNE(w - hi - mid): hi + mid + lo == w exactly (3 x 8 significand bits).
__device__ __forceinline__ void tbg_split3(const float (&v)[8], tbg_bf16x8 &h, tbg_bf16x8 &m, tbg_bf16x8 &l) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const __bf16 bh = (__bf16)v[e];
    const float r1 = v[e] - (float)bh;
    const __bf16 bm = (__bf16)r1;
    h[e] = bh; m[e] = bm; l[e] = (__bf16)(r1 - (float)bm);
  }
}

__global__ __launch_bounds__(256) void weight_pack_x3_kernel(const float *__restrict__ src, tbg_bf16x8 *__restrict__ dst,
                                                             int T, int I, int O, int transpose, int flip) {
  const int t = blockIdx.z;
  const int td = flip ? T - 1 - t : t;
  const int C = transpose ? O : I, M = transpose ? I : O;
  const int C8 = (C + 7) >> 3;
  const int c8 = blockIdx.y;
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = 8 * c8 + e;
    const int i = transpose ? m : c, o = transpose ? c : m;
    v[e] = (c < C) ? src[((size_t)t * I + i) * O + o] : 0.f;
  }
  tbg_bf16x8 h, mi, l;
  tbg_split3(v, h, mi, l);
  const size_t plane = (size_t)T * C8 * M, d = ((size_t)td * C8 + c8) * M + m;
  dst[d] = h; dst[plane + d] = mi; dst[2 * plane + d] = l;
}

extern "C" long long tbg_weight_pack_x3_bytes(int T, int I, int O, int transpose) {
  const long long b = tbg_weight_pack_bf16_bytes(T, I, O, transpose);
  return b < 0 ? b : 3 * b;
}

extern "C" int tbg_weight_pack_x3(const float *src, void *dst, int T, int I, int O, int transpose, int flip,
                                  void *stream) {
  if (!src || !dst || T < 1 || I < 1 || O < 1) return TBG_EINVAL;
  if ((reinterpret_cast<uintptr_t>(dst) & 15) != 0) return TBG_EINVAL;
  const int C = transpose ? O : I, M = transpose ? I : O;
  dim3 grid((M + 255) / 256, (C + 7) / 8, T);
  hipLaunchKernelGGL(weight_pack_x3_kernel, grid, dim3(256), 0, tbg_stream(stream), src,
                     reinterpret_cast<tbg_bf16x8 *>(dst), T, I, O, transpose, flip);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

// Multi-tensor form: ONE launch packs every filter of a model (both orientations, either format) from a device table
// -- the training step refreshes all its packed filters at the start of a step instead of ~140 small launches.
__global__ __launch_bounds__(256) void weight_pack_multi_kernel(const tbg_pack_item *__restrict__ items) {
  const tbg_pack_item it = items[blockIdx.y];
  const int C = it.transpose ? it.O : it.I, M = it.transpose ? it.I : it.O;
  const int KP = it.bf16 ? 8 : 4;
  const int CU = (C + KP - 1) / KP;
  const long long units = (long long)it.T * CU * M;
  for (long long u = (long long)blockIdx.x * 256 + threadIdx.x; u < units; u += (long long)gridDim.x * 256) {
    const int m = (int)(u % M);
    const long long r = u / M;
    const int cu = (int)(r % CU), t = (int)(r / CU);
    const int td = it.flip ? it.T - 1 - t : t;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = KP * cu + e;
      const int i = it.transpose ? m : c, o = it.transpose ? c : m;
      v[e] = (e < KP && c < C) ? it.src[((size_t)t * it.I + i) * it.O + o] : 0.f;
    }
    const size_t d = ((size_t)td * CU + cu) * M + m;
    if (it.bf16 == 2) {  // f32x3: hi | mid | lo planes
      tbg_bf16x8 h, mi, l;
      tbg_split3(v, h, mi, l);
      tbg_bf16x8 *dp = reinterpret_cast<tbg_bf16x8 *>(it.dst);
      dp[d] = h; dp[(size_t)units + d] = mi; dp[2 * (size_t)units + d] = l;
    } else if (it.bf16) {
      tbg_bf16x8 b;
#pragma unroll
      for (int e = 0; e < 8; ++e) b[e] = (__bf16)v[e];
      reinterpret_cast<tbg_bf16x8 *>(it.dst)[d] = b;
    } else {
      reinterpret_cast<float4 *>(it.dst)[d] = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

extern "C" int tbg_weight_pack_multi(const tbg_pack_item *items_dev, int n_items, void *stream) {
  if (!items_dev || n_items < 1) return TBG_EINVAL;
  if (n_items > 65535) return TBG_ERANGE;
  // 512 blocks per item: the few multi-million-element filters dominate the launch (64 blocks each left most of the chip idle:
  // 183 us for ~370 MB); blocks past a small item's units exit at once
  hipLaunchKernelGGL(weight_pack_multi_kernel, dim3(512, n_items), dim3(256), 0, tbg_stream(stream), items_dev);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

// ============================================================================================
// Adam (Keras / ResourceApplyAdam semantics) and EMA lerp over flat buffers
// ============================================================================================
__global__ __launch_bounds__(256) void adam_tf_kernel(float *__restrict__ theta, float *__restrict__ m,
                                                      float *__restrict__ v, const float *__restrict__ g, long long n,
                                                      float lr, float b1, float b2, float eps,
                                                      const long long *__restrict__ step) {
  const double t = (double)(step[0] + 1);
  const double c1 = 1.0 - pow((double)b1, t);
  const float lr_t = (float)((double)lr * sqrt(1.0 - pow((double)b2, t)) / c1);
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 gg = reinterpret_cast<const float4 *>(g)[i];
    float4 mm = reinterpret_cast<float4 *>(m)[i];
    float4 vv = reinterpret_cast<float4 *>(v)[i];
    float4 th = reinterpret_cast<float4 *>(theta)[i];
#define TBG_ADAM1(c)                                   \
  mm.c = b1 * mm.c + (1.f - b1) * gg.c;                \
  vv.c = b2 * vv.c + (1.f - b2) * gg.c * gg.c;         \
  th.c -= lr_t * mm.c / (sqrtf(vv.c) + eps);
    TBG_ADAM1(x) TBG_ADAM1(y) TBG_ADAM1(z) TBG_ADAM1(w)
#undef TBG_ADAM1
    reinterpret_cast<float4 *>(m)[i] = mm;
    reinterpret_cast<float4 *>(v)[i] = vv;
    reinterpret_cast<float4 *>(theta)[i] = th;
  }
  for (long long i = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    const float gg = g[i];
    const float mm = b1 * m[i] + (1.f - b1) * gg;
    const float vv = b2 * v[i] + (1.f - b2) * gg * gg;
    m[i] = mm; v[i] = vv;
    theta[i] -= lr_t * mm / (sqrtf(vv) + eps);
  }
}

extern "C" int tbg_adam_tf_f32(float *theta, float *m, float *v, const float *g, long long n, float lr, float beta1,
                               float beta2, float eps, const long long *step, void *stream) {
  if (!theta || !m || !v || !g || !step || n < 1) return TBG_EINVAL;
  if ((((uintptr_t)theta | (uintptr_t)m | (uintptr_t)v | (uintptr_t)g) & 15) != 0) return TBG_EINVAL;
  long long blocks = (n / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(adam_tf_kernel, dim3((int)blocks), dim3(256), 0, tbg_stream(stream), theta, m, v, g, n, lr, beta1,
                     beta2, eps, step);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

__global__ __launch_bounds__(256) void ema_lerp_kernel(float *__restrict__ dst, const float *__restrict__ src,
                                                       long long n, float beta) {
  const long long n4 = n >> 2;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    float4 s = reinterpret_cast<const float4 *>(src)[i];
    float4 d = reinterpret_cast<float4 *>(dst)[i];
    d.x = s.x + (d.x - s.x) * beta; d.y = s.y + (d.y - s.y) * beta;
    d.z = s.z + (d.z - s.z) * beta; d.w = s.w + (d.w - s.w) * beta;
    reinterpret_cast<float4 *>(dst)[i] = d;
  }
  for (long long i = (n4 << 2) + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
    dst[i] = src[i] + (dst[i] - src[i]) * beta;
}

extern "C" int tbg_ema_lerp_f32(float *dst, const float *src, long long n, float beta, void *stream) {
  if (!dst || !src || n < 1) return TBG_EINVAL;
  if ((((uintptr_t)dst | (uintptr_t)src) & 15) != 0) return TBG_EINVAL;
  long long blocks = (n / 4 + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(ema_lerp_kernel, dim3((int)blocks), dim3(256), 0, tbg_stream(stream), dst, src, n, beta);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

// ============================================================================================
// demodulation coefficients
// ============================================================================================
__global__ __launch_bounds__(256) void wsq_kernel(const float *__restrict__ w, float *__restrict__ wsq, int T, int IO,
                                                  float coef2) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= IO) return;
  float a = 0.f;
  for (int t = 0; t < T; ++t) {
    const float v = w[(size_t)t * IO + i];
    a += v * v;
  }
  wsq[i] = a * coef2;
}

// block = 4 waves; lane = (isub = lane>>4, o = lane&15): every wave reduces its 4 i-subgroups
// with two wavefront shuffles, the 4 waves meet in LDS.
__global__ __launch_bounds__(256) void demod_kernel(const float *__restrict__ s, const float *__restrict__ wsq,
                                                    float *__restrict__ d, int I, int O) {
  __shared__ float red[4][16];
  const int b = blockIdx.y;
  const int o = blockIdx.x * 16 + (threadIdx.x & 15);
  const int isub = threadIdx.x >> 4;  // 0..15
  float a = 0.f;
  if (o < O)
    for (int i = isub; i < I; i += 16) {
      const float sv = s[(size_t)b * I + i];
      a += sv * sv * wsq[(size_t)i * O + o];
    }
  a += __shfl_xor(a, 16, 64);
  a += __shfl_xor(a, 32, 64);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane < 16) red[wave][lane] = a;
  __syncthreads();
  if (threadIdx.x < 16 && o < O)
    d[(size_t)b * O + o] = rsqrtf(red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x] + 1e-8f);
}

extern "C" int tbg_demod_coefs_f32(const float *s, const float *w, float *wsq, float *d, int B, int T, int I, int O,
                                   float coef, void *stream) {
  if (!s || !w || !wsq || !d || B < 1 || T < 1 || I < 1 || O < 1) return TBG_EINVAL;
  hipStream_t st = tbg_stream(stream);
  hipLaunchKernelGGL(wsq_kernel, dim3((I * O + 255) / 256), dim3(256), 0, st, w, wsq, T, I * O, coef * coef);
  TBG_LAUNCH_CHECK();
  hipLaunchKernelGGL(demod_kernel, dim3((O + 15) / 16, B), dim3(256), 0, st, s, wsq, d, I, O);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

// ============================================================================================
extern "C" int tbg_version(void) { return 200; }

extern "C" const char *tbg_strerror(int code) {
  switch (code) {
    case TBG_OK: return "ok";
    case TBG_EINVAL: return "invalid argument (shape / pointer / factor)";
    case TBG_ERANGE: return "tensor has more than INT32_MAX elements";
    case TBG_EHIP: return "HIP kernel launch failed";
    case TBG_EUNSUPPORTED: return "configuration not supported by this build";
    default: return "unknown tbg error";
  }
}
