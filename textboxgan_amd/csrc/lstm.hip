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
    seq[((size_t)b * T + t) * (D * H) + d * H + u] = hh;
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
    float dh = dseq[((size_t)b * T + t) * (D * H) + d * H + u];
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
    float *w = dg + (((size_t)d * T + t) * B + b) * 4 * H;
    w[u] = d_i; w[H + u] = d_f; w[2 * H + u] = d_g; w[3 * H + u] = d_o;
  }
}

static int lstm_args_ok(int D, int T, int B, int H, int s) {
  return D >= 1 && D <= 2 && T >= 1 && B >= 1 && H >= 1 && s >= 0 && s < T && (long long)D * T * B * 4 * H < 2147483647LL;
}

extern "C" int tbg_lstm_step_fwd_f32(const float *gx, const float *hw, float *act, float *cs, float *h, float *seq, int D,
                                     int T, int B, int H, int s, void *stream) {
  if (!gx || !act || !cs || !h || !seq || !lstm_args_ok(D, T, B, H, s)) return TBG_EINVAL;
  if (s > 0 && !hw) return TBG_EINVAL;
  const int n = D * B * H;
  hipLaunchKernelGGL(lstm_step_fwd_kernel, dim3((n + 255) / 256), dim3(256), 0, tbg_stream(stream), gx, s > 0 ? hw : nullptr,
                     act, cs, h, seq, D, T, B, H, s);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

extern "C" int tbg_lstm_step_bwd_f32(const float *dseq, const float *dh_rec, float *dc, const float *act, const float *cs,
                                     float *dg, float *dgates, int D, int T, int B, int H, int s, int first, void *stream) {
  if (!dseq || !dc || !act || !cs || !dg || !dgates || !lstm_args_ok(D, T, B, H, s)) return TBG_EINVAL;
  if (!first && !dh_rec) return TBG_EINVAL;
  const int n = D * B * H;
  hipLaunchKernelGGL(lstm_step_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, tbg_stream(stream), dseq,
                     first ? nullptr : dh_rec, dc, act, cs, dg, dgates, D, T, B, H, s, first);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}
