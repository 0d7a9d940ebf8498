// Pointwise halves of one time step of a (bi)directional LSTM layer with FROZEN weights -- the recurrent encoder of
// the OCR branch (aster_inferer.py:28-190 runs the ASTER SavedModel; its encoder is 2x BiLSTM).
//
// The step's two GEMMs stay library GEMMs, batched over the directions (north_star: dense layers are PyTorch-ROCm
// GEMMs): forward  hw[d] = h[d] @ Whh[d]^T,  backward  dh_rec[d] = dgates[d] @ Whh[d].  These kernels do everything
// else of the step in ONE launch for both directions: MIOpen's LSTM issues a GEMM and a pointwise kernel per
// direction per step (4 launches per step and layer); this path issues 2.
//
// Layouts (D directions, S = step index, direction d works on time t = d == 0 ? s : T-1-s):
//   gx, dg : [D][T][B][4H]   input projections (+ both biases) / their gradients, TIME-major; gate order i,f,g,o
//   hw     : [D][B][4H]      recurrent projection of this step (NULL at s = 0)
//   act    : [D][S][B][4H]   post-activation gates, saved for backward
//   cs     : [D][S][B][H]    cell state after step s
//   h      : [D][B][H]       hidden state (contiguous, the next step's GEMM operand)
//   seq    : [B][T][D*H]     layer output (batch-first, directions concatenated) / its gradient dseq
#include "common.h"

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(256) void lstm_step_fwd_kernel(const float *__restrict__ gx, const float *__restrict__ hw,
                                                            float *__restrict__ act, float *__restrict__ cs,
                                                            float *__restrict__ h, float *__restrict__ seq, int D, int T,
                                                            int B, int H, int s) {
  const int n = D * B * H;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
    const int u = e % H, b = (e / H) % B, d = e / (H * B);
    const int t = d == 0 ? s : T - 1 - s;
    const float *g = gx + (((size_t)d * T + t) * B + b) * 4 * H;
    float pi = g[u], pf = g[H + u], pg = g[2 * H + u], po = g[3 * H + u];
    if (hw) {
      const float *r = hw + ((size_t)d * B + b) * 4 * H;
      pi += r[u]; pf += r[H + u]; pg += r[2 * H + u]; po += r[3 * H + u];
    }
    const float i = sigmoidf_(pi), f = sigmoidf_(pf), gg = tanhf(pg), o = sigmoidf_(po);
    const float cp = s > 0 ? cs[(((size_t)d * T + s - 1) * B + b) * H + u] : 0.f;
    const float c = f * cp + i * gg;
    const float hh = o * tanhf(c);
    float *a = act + (((size_t)d * T + s) * B + b) * 4 * H;
    a[u] = i; a[H + u] = f; a[2 * H + u] = gg; a[3 * H + u] = o;
    cs[(((size_t)d * T + s) * B + b) * H + u] = c;
    h[((size_t)d * B + b) * H + u] = hh;
    if (seq) seq[((size_t)b * T + t) * (D * H) + d * H + u] = hh;
  }
}

// dh = dseq[t] + dh_rec ;  through h = o * tanh(c), c = f * c_prev + i * g ; dc carries the cell-state gradient
__global__ __launch_bounds__(256) void lstm_step_bwd_kernel(const float *__restrict__ dseq, const float *__restrict__ dh_rec,
                                                            float *__restrict__ dc, const float *__restrict__ act,
                                                            const float *__restrict__ cs, float *__restrict__ dg,
                                                            float *__restrict__ dgates, int D, int T, int B, int H, int s,
                                                            int first) {
  const int n = D * B * H;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += gridDim.x * 256) {
    const int u = e % H, b = (e / H) % B, d = e / (H * B);
    const int t = d == 0 ? s : T - 1 - s;
    float dh = dseq ? dseq[((size_t)b * T + t) * (D * H) + d * H + u] : 0.f;
    if (dh_rec) dh += dh_rec[((size_t)d * B + b) * H + u];
    const float *a = act + (((size_t)d * T + s) * B + b) * 4 * H;
    const float i = a[u], f = a[H + u], gg = a[2 * H + u], o = a[3 * H + u];
    const float c = cs[(((size_t)d * T + s) * B + b) * H + u];
    const float cp = s > 0 ? cs[(((size_t)d * T + s - 1) * B + b) * H + u] : 0.f;
    const float tc = tanhf(c);
    const size_t ic = ((size_t)d * B + b) * H + u;
    const float dcc = dh * o * (1.f - tc * tc) + (first ? 0.f : dc[ic]);
    const float d_o = dh * tc * o * (1.f - o);
    const float d_i = dcc * gg * i * (1.f - i);
    const float d_f = dcc * cp * f * (1.f - f);
    const float d_g = dcc * i * (1.f - gg * gg);
    dc[ic] = dcc * f;
    float *q = dgates + ((size_t)d * B + b) * 4 * H;
    q[u] = d_i; q[H + u] = d_f; q[2 * H + u] = d_g; q[3 * H + u] = d_o;
    if (dg) {
      float *w = dg + (((size_t)d * T + t) * B + b) * 4 * H;
      w[u] = d_i; w[H + u] = d_f; w[2 * H + u] = d_g; w[3 * H + u] = d_o;
    }
  }
}

static int lstm_args_ok(int D, int T, int B, int H, int s) {
  return D >= 1 && D <= 2 && T >= 1 && B >= 1 && H >= 1 && s >= 0 && s < T && (long long)D * T * B * 4 * H < 2147483647LL;
}

extern "C" int tbg_lstm_step_fwd_f32(const float *gx, const float *hw, float *act, float *cs, float *h, float *seq, int D,
                                     int T, int B, int H, int s, void *stream) {
  if (!gx || !act || !cs || !h || !lstm_args_ok(D, T, B, H, s)) return TBG_EINVAL;
  const int n = D * B * H;
  hipLaunchKernelGGL(lstm_step_fwd_kernel, dim3((n + 255) / 256), dim3(256), 0, tbg_stream(stream), gx, hw, act, cs, h, seq, D, T,
                     B, H, s);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

extern "C" int tbg_lstm_step_bwd_f32(const float *dseq, const float *dh_rec, float *dc, const float *act, const float *cs,
                                     float *dg, float *dgates, int D, int T, int B, int H, int s, int first, void *stream) {
  if (!dc || !act || !cs || !dgates || !lstm_args_ok(D, T, B, H, s)) return TBG_EINVAL;
  if (!dseq && !dh_rec) return TBG_EINVAL;
  const int n = D * B * H;
  hipLaunchKernelGGL(lstm_step_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, tbg_stream(stream), dseq, dh_rec, dc, act, cs, dg,
                     dgates, D, T, B, H, s, first);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

// ============================================================================================
// Bahdanau attention context of the OCR decoder (one block per image), frozen weights:
//   e[t] = sum_k v[k] * tanh(enc_proj[b,t,k] + q[b,k]);  a = softmax_t(e);  ctx[b,:] = sum_t a[t] * enc[b,t,:]
// forward saves a; backward recomputes tanh and ACCUMULATES into denc_proj / denc (the decoder runs max_steps steps over the
// same encoder output), writes dq.  T <= 64 (one wavefront does the softmax).
// ============================================================================================
#define ATT_MAXT 64

__global__ __launch_bounds__(256) void attn_ctx_fwd_kernel(const float *__restrict__ q, const float *__restrict__ ep,
                                                           const float *__restrict__ enc, const float *__restrict__ v,
                                                           float *__restrict__ ctx, float *__restrict__ a_out, int T, int H,
                                                           int E) {
  __shared__ float part[4][ATT_MAXT];
  __shared__ float a_s[ATT_MAXT];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float *epb = ep + (size_t)b * T * H;
  for (int t = 0; t < T; ++t) {
    float s = 0.f;
    for (int k = tid; k < H; k += 256) s += v[k] * tanhf(epb[(size_t)t * H + k] + q[(size_t)b * H + k]);
    s = wave_sum(s);
    if (lane == 0) part[wave][t] = s;
  }
  __syncthreads();
  if (wave == 0) {
    const float e = lane < T ? part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane] : -3.0e38f;
    float mx = e;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    const float ex = lane < T ? expf(e - mx) : 0.f;
    const float den = wave_sum(ex);
    if (lane < T) {
      const float a = ex / den;
      a_s[lane] = a;
      a_out[(size_t)b * T + lane] = a;
    }
  }
  __syncthreads();
  const float *eb = enc + (size_t)b * T * E;
  for (int j = tid; j < E; j += 256) {
    float c = 0.f;
    for (int t = 0; t < T; ++t) c += a_s[t] * eb[(size_t)t * E + j];
    ctx[(size_t)b * E + j] = c;
  }
}

__global__ __launch_bounds__(256) void attn_ctx_bwd_kernel(const float *__restrict__ dctx, const float *__restrict__ a_in,
                                                           const float *__restrict__ q, const float *__restrict__ ep,
                                                           const float *__restrict__ enc, const float *__restrict__ v,
                                                           float *__restrict__ dq, float *__restrict__ dep,
                                                           float *__restrict__ denc, int T, int H, int E) {
  __shared__ float part[4][ATT_MAXT];
  __shared__ float de_s[ATT_MAXT], a_s[ATT_MAXT];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float *eb = enc + (size_t)b * T * E;
  const float *dcb = dctx + (size_t)b * E;
  if (tid < T) a_s[tid] = a_in[(size_t)b * T + tid];
  for (int t = 0; t < T; ++t) {  // da[t] = <dctx, enc[t]>
    float s = 0.f;
    for (int j = tid; j < E; j += 256) s += dcb[j] * eb[(size_t)t * E + j];
    s = wave_sum(s);
    if (lane == 0) part[wave][t] = s;
  }
  __syncthreads();
  if (wave == 0) {
    const float da = lane < T ? part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane] : 0.f;
    const float a = lane < T ? a_s[lane] : 0.f;
    const float dot = wave_sum(a * da);
    if (lane < T) de_s[lane] = a * (da - dot);
  }
  __syncthreads();
  if (denc) {  // denc[t, j] += a[t] * dctx[j]  (callers may instead form sum_s a_s (x) dctx_s with one batched GEMM)
    for (int j = tid; j < E; j += 256) {
      const float d = dcb[j];
      for (int t = 0; t < T; ++t) denc[((size_t)b * T + t) * E + j] += a_s[t] * d;
    }
  }
  const float *epb = ep + (size_t)b * T * H;
  for (int k = tid; k < H; k += 256) {
    const float qk = q[(size_t)b * H + k], vk = v[k];
    float acc = 0.f;
    for (int t0 = 0; t0 < T; t0 += 8) {  // 8 read-modify-writes in flight (one at a time was a serial round-trip chain)
      float e8[8], o8[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int t = min(t0 + u, T - 1);
        e8[u] = epb[(size_t)t * H + k];
        o8[u] = dep[((size_t)b * T + t) * H + k];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (t0 + u < T) {
          const float th = tanhf(e8[u] + qk);
          const float dp = de_s[t0 + u] * vk * (1.f - th * th);
          dep[((size_t)b * T + t0 + u) * H + k] = o8[u] + dp;
          acc += dp;
        }
      }
    }
    dq[(size_t)b * H + k] = acc;
  }
}

extern "C" int tbg_attn_ctx_fwd_f32(const float *q, const float *enc_proj, const float *enc, const float *v, float *ctx, float *a,
                                    int B, int T, int H, int E, void *stream) {
  if (!q || !enc_proj || !enc || !v || !ctx || !a || B < 1 || T < 1 || H < 1 || E < 1) return TBG_EINVAL;
  if (T > ATT_MAXT) return TBG_EUNSUPPORTED;
  hipLaunchKernelGGL(attn_ctx_fwd_kernel, dim3(B), dim3(256), 0, tbg_stream(stream), q, enc_proj, enc, v, ctx, a, T, H, E);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

extern "C" int tbg_attn_ctx_bwd_f32(const float *dctx, const float *a, const float *q, const float *enc_proj, const float *enc,
                                    const float *v, float *dq, float *denc_proj, float *denc, int B, int T, int H, int E,
                                    void *stream) {
  if (!dctx || !a || !q || !enc_proj || !enc || !v || !dq || !denc_proj || B < 1 || T < 1 || H < 1 || E < 1)
    return TBG_EINVAL;
  if (T > ATT_MAXT) return TBG_EUNSUPPORTED;
  hipLaunchKernelGGL(attn_ctx_bwd_kernel, dim3(B), dim3(256), 0, tbg_stream(stream), dctx, a, q, enc_proj, enc, v, dq, denc_proj,
                     denc, T, H, E);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}
