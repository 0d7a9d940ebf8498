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

// ============================================================================================
// FUSED recurrent steps (round 6): the step's recurrent projection AND its pointwise half in ONE launch for both directions.
//
// The per-step pair (library GEMM  hw = h @ Whh^T  of [B, H] x [H, 4H], then tbg_lstm_step_*) cost 7.7 + 5.1 us per step and layer
// under graph replay -- two launch floors and two memory round trips for 2 MFLOP.  Here a block owns LSF_UB hidden units of one
// direction (all four gates: a 4 LSF_UB x H slice of Whh, 32 KB) and up to 16 samples: it forms its slice of the projection from the
// previous state and finishes the cell for its units -- the kernel boundary between two steps is the only synchronisation the
// recurrence needs.  The state travels TRANSPOSED between launches (hT [D][H][B], dgT [D][4H][B]: a block's units are contiguous
// runs for the producer and the whole state is a linear LDS image for the consumer) in two buffers used alternately (every block of
// step s reads ALL of step s-1's state while it writes its own part of step s's).
//   forward   pre[b, j] = gx[d, t, b, j] + sum_k hT[k, b] Whh[d, j, k]            (j = g H + u over the block's units u)
//   backward  dh[b, u]  = dseq[b, t, d H + u] + sum_j dgT[j, b] WhhT[d, u, j]     (WhhT [D][H][4H]: the transposed copy)
// followed by exactly the arithmetic of lstm_step_fwd_kernel / lstm_step_bwd_kernel.  Threads: (row r, K slice) pairs, the row's
// weights in registers (128 contiguous bytes per thread), the state read from LDS as 16-byte broadcasts, partial sums through LDS.
// ============================================================================================
#define LSF_UB 8

struct LstmFusedP {
  const float *gx, *w, *state_in, *dseq, *act_in, *cs_in;
  float *act, *cs, *state_out, *seq, *dc, *dg;
  int D, T, B, H, s, first;
  int K, project;  // forward: rows of the state / columns of a weight row (H for a plain LSTM layer; E + H for the decoder's cell,
                   // whose input is [context; hidden]); project = 0: no recurrent term at all (s = 0 of a layer)
};

// the state of one direction (rows x B floats, row-major) -> this batch group's LDS image [rows][BB].  Whole batch groups of a
// batch that is a multiple of 4: 16-byte loads, eight in flight per thread (a loop of dependent scalar load / store pairs cost one
// memory round trip per iteration: 16 of them per forward step, 64 per backward step)
#define LSF_THREADS 512
#define LSF_CH 16  // columns of a weight chunk held in registers per thread
template <int BB>
__device__ __forceinline__ void lstm_stage_state(float *dst, const float *src, int rows, int B, int b0, int nb, int tid) {
  if (nb == BB && (B & 3) == 0) {
    constexpr int Q4 = BB / 4;
    const int total = rows * Q4;
    for (int base = 0; base < total; base += LSF_THREADS * 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = min(base + u * LSF_THREADS + tid, total - 1);
        const int k = e / Q4, q = e - k * Q4;
        v[u] = *reinterpret_cast<const float4 *>(src + (size_t)k * B + b0 + 4 * q);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = base + u * LSF_THREADS + tid;
        if (e < total) reinterpret_cast<float4 *>(dst)[e] = v[u];
      }
    }
  } else {
    for (int e = tid; e < rows * BB; e += LSF_THREADS) {
      const int k = e / BB, bb = e - k * BB;
      dst[e] = bb < nb ? src[(size_t)k * B + b0 + bb] : 0.f;
    }
  }
}

// acc[bb] += sum_i wv[i] * st[(k0 + i) * BB + bb]
template <int BB>
__device__ __forceinline__ void lstm_chunk_fma(float (&acc)[BB], const float (&wv)[LSF_CH], const float *st) {
#pragma unroll
  for (int i = 0; i < LSF_CH; ++i) {
#pragma unroll
    for (int bb = 0; bb < BB; bb += 4) {
      const float4 h4 = *reinterpret_cast<const float4 *>(st + (size_t)i * BB + bb);
      acc[bb] += wv[i] * h4.x; acc[bb + 1] += wv[i] * h4.y; acc[bb + 2] += wv[i] * h4.z; acc[bb + 3] += wv[i] * h4.w;
    }
  }
}

template <int BB>  // samples per block (power of two <= 16; the batch is split over blockIdx.z)
__global__ __launch_bounds__(LSF_THREADS) void lstm_fused_fwd_kernel(const LstmFusedP p) {
  extern __shared__ __attribute__((aligned(16))) float lsm[];
  const int H = p.H, B = p.B, T = p.T, s = p.s, K = p.K;
  constexpr int KS = LSF_THREADS / 32;           // K slices: thread (row r, slice ks) takes the chunks ks, ks + KS, ... of LSF_CH columns
  float *hs = lsm;                               // [K][BB]   previous state of this batch group
  float *part = lsm + (size_t)K * BB;            // [KS][32][BB] partial projections
  const int tid = threadIdx.x;
  const int d = blockIdx.y, u0 = blockIdx.x * LSF_UB, b0 = blockIdx.z * BB;
  const int nb = min(BB, B - b0);
  const int t = d == 0 ? s : T - 1 - s;
  const int r = tid & 31, ks = tid >> 5;         // row: gate g = r / UB, unit u0 + r % UB
  const int NCH = K / LSF_CH;                    // (K % 16 == 0 checked on the host)
  const int j = (r / LSF_UB) * H + u0 + (r % LSF_UB);
  // pointwise ownership: thread (b, ul) for tid < BB * UB
  const int pb = tid / LSF_UB, pu = tid % LSF_UB;
  const bool pw = tid < BB * LSF_UB && pb < nb;
  float gv[4] = {0.f, 0.f, 0.f, 0.f}, cp = 0.f;
  if (pw) {
    const float *g = p.gx + (((size_t)d * T + t) * B + b0 + pb) * 4 * H + u0 + pu;
#pragma unroll
    for (int q = 0; q < 4; ++q) gv[q] = g[q * H];
    if (s > 0) cp = p.cs[(((size_t)d * T + s - 1) * B + b0 + pb) * H + u0 + pu];
  }
  if (p.project) {
    // this thread's weights of its first chunk go out BEFORE the state is staged: one memory round trip per step, not two
    float wv[LSF_CH];
    const float *wrow = p.w + ((size_t)d * 4 * H + j) * K;
#pragma unroll
    for (int i = 0; i < LSF_CH; i += 4)
      *reinterpret_cast<float4 *>(&wv[i]) = *reinterpret_cast<const float4 *>(wrow + min(ks, NCH - 1) * LSF_CH + i);
    lstm_stage_state<BB>(hs, p.state_in + (size_t)d * K * B, K, B, b0, nb, tid);
    float acc[BB];
#pragma unroll
    for (int bb = 0; bb < BB; ++bb) acc[bb] = 0.f;
    __syncthreads();
    for (int ch = ks; ch < NCH; ch += KS) {
      if (ch != ks) {
#pragma unroll
        for (int i = 0; i < LSF_CH; i += 4) *reinterpret_cast<float4 *>(&wv[i]) = *reinterpret_cast<const float4 *>(wrow + ch * LSF_CH + i);
      }
      lstm_chunk_fma<BB>(acc, wv, hs + (size_t)ch * LSF_CH * BB);
    }
#pragma unroll
    for (int bb = 0; bb < BB; bb += 4)
      *reinterpret_cast<float4 *>(part + ((size_t)ks * 32 + r) * BB + bb) = make_float4(acc[bb], acc[bb + 1], acc[bb + 2], acc[bb + 3]);
    __syncthreads();
    if (pw) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float a = 0.f;
#pragma unroll
        for (int k2 = 0; k2 < KS; ++k2) a += part[((size_t)k2 * 32 + q * LSF_UB + pu) * BB + pb];  // fixed order: deterministic
        gv[q] += a;
      }
    }
  }
  if (!pw) return;
  const int b = b0 + pb, u = u0 + pu;
  const float i_ = sigmoidf_(gv[0]), f_ = sigmoidf_(gv[1]), g_ = tanhf(gv[2]), o_ = sigmoidf_(gv[3]);
  const float c = f_ * cp + i_ * g_;
  const float hh = o_ * tanhf(c);
  float *a = p.act + (((size_t)d * T + s) * B + b) * 4 * H + u;
  a[0] = i_; a[H] = f_; a[2 * H] = g_; a[3 * H] = o_;
  p.cs[(((size_t)d * T + s) * B + b) * H + u] = c;
  p.state_out[((size_t)d * H + u) * B + b] = hh;
  if (p.seq) p.seq[((size_t)b * T + t) * (p.D * H) + d * H + u] = hh;
}

template <int BB>
__global__ __launch_bounds__(LSF_THREADS) void lstm_fused_bwd_kernel(const LstmFusedP p) {
  extern __shared__ __attribute__((aligned(16))) float lsm[];
  const int H = p.H, B = p.B, T = p.T, s = p.s, G = 4 * H;
  constexpr int JS = LSF_THREADS / LSF_UB;       // J slices: thread (unit ul, slice js) takes the chunks js, js + JS, ... of LSF_CH gate rows
  float *dgs = lsm;                              // [4H][BB]  gate gradients of the step after this one
  float *part = lsm + (size_t)G * BB;            // [JS][UB][BB]
  const int tid = threadIdx.x;
  const int d = blockIdx.y, u0 = blockIdx.x * LSF_UB, b0 = blockIdx.z * BB;
  const int nb = min(BB, B - b0);
  const int t = d == 0 ? s : T - 1 - s;
  const int ul = tid & (LSF_UB - 1), js = tid / LSF_UB;
  const int NCH = G / LSF_CH;
  const int pb = tid / LSF_UB, pu = tid % LSF_UB;
  const bool pw = tid < BB * LSF_UB && pb < nb;
  float dh = 0.f, av[4] = {0.f, 0.f, 0.f, 0.f}, c = 0.f, cp = 0.f, dcin = 0.f;
  if (pw) {
    const int b = b0 + pb, u = u0 + pu;
    if (p.dseq) dh = p.dseq[((size_t)b * T + t) * (p.D * H) + d * H + u];
    const float *a = p.act_in + (((size_t)d * T + s) * B + b) * 4 * H + u;
#pragma unroll
    for (int q = 0; q < 4; ++q) av[q] = a[q * H];
    c = p.cs_in[(((size_t)d * T + s) * B + b) * H + u];
    if (s > 0) cp = p.cs_in[(((size_t)d * T + s - 1) * B + b) * H + u];
    if (!p.first) dcin = p.dc[((size_t)d * B + b) * H + u];
  }
  if (!p.first) {
    const float *wrow = p.w + ((size_t)d * H + u0 + ul) * G;  // WhhT[d][u][j]
    float wv[LSF_CH];
#pragma unroll
    for (int i = 0; i < LSF_CH; i += 4)
      *reinterpret_cast<float4 *>(&wv[i]) = *reinterpret_cast<const float4 *>(wrow + min(js, NCH - 1) * LSF_CH + i);
    lstm_stage_state<BB>(dgs, p.state_in + (size_t)d * G * B, G, B, b0, nb, tid);
    float acc[BB];
#pragma unroll
    for (int bb = 0; bb < BB; ++bb) acc[bb] = 0.f;
    __syncthreads();
    for (int ch = js; ch < NCH; ch += JS) {
      if (ch != js) {
#pragma unroll
        for (int i = 0; i < LSF_CH; i += 4) *reinterpret_cast<float4 *>(&wv[i]) = *reinterpret_cast<const float4 *>(wrow + ch * LSF_CH + i);
      }
      lstm_chunk_fma<BB>(acc, wv, dgs + (size_t)ch * LSF_CH * BB);
    }
#pragma unroll
    for (int bb = 0; bb < BB; bb += 4)
      *reinterpret_cast<float4 *>(part + ((size_t)js * LSF_UB + ul) * BB + bb) = make_float4(acc[bb], acc[bb + 1], acc[bb + 2], acc[bb + 3]);
    __syncthreads();
    if (pw) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;  // four chains, combined in a fixed order: deterministic
      for (int k2 = 0; k2 < JS; k2 += 4) {
        a0 += part[((size_t)(k2 + 0) * LSF_UB + pu) * BB + pb]; a1 += part[((size_t)(k2 + 1) * LSF_UB + pu) * BB + pb];
        a2 += part[((size_t)(k2 + 2) * LSF_UB + pu) * BB + pb]; a3 += part[((size_t)(k2 + 3) * LSF_UB + pu) * BB + pb];
      }
      dh += (a0 + a1) + (a2 + a3);
    }
  }
  if (!pw) return;
  const int b = b0 + pb, u = u0 + pu;
  const float i_ = av[0], f_ = av[1], g_ = av[2], o_ = av[3];
  const float tc = tanhf(c);
  const float dcc = dh * o_ * (1.f - tc * tc) + dcin;
  const float d_o = dh * tc * o_ * (1.f - o_);
  const float d_i = dcc * g_ * i_ * (1.f - i_);
  const float d_f = dcc * cp * f_ * (1.f - f_);
  const float d_g = dcc * i_ * (1.f - g_ * g_);
  p.dc[((size_t)d * B + b) * H + u] = dcc * f_;
  float *q = p.state_out + ((size_t)d * G + u) * B + b;  // dgT[d][g H + u][b]
  q[0] = d_i; q[(size_t)H * B] = d_f; q[(size_t)2 * H * B] = d_g; q[(size_t)3 * H * B] = d_o;
  if (p.dg) {
    float *w = p.dg + (((size_t)d * T + t) * B + b) * 4 * H + u;
    w[0] = d_i; w[H] = d_f; w[2 * H] = d_g; w[3 * H] = d_o;
  }
}

// samples per block: the smallest group that still leaves the launch within one block per CU -- every block stages the WHOLE state of
// its samples, so smaller groups mean less staging per block and more blocks in flight (B = 16, one BiLSTM layer forward + backward,
// 50 launches: 429 us with 16-sample groups, 349 with 8, 287 with 4; B = 32: 8-sample groups are best -- profiles/r06_lstm_groups.txt)
static int lstm_fused_bb(int B, int blocks_per_group) {
  for (int bb = 4; bb < 16; bb *= 2)
    if (B <= bb || (long long)blocks_per_group * ((B + bb - 1) / bb) <= 256) return bb;
  return 16;
}

static bool lstm_fused_ok(int D, int T, int B, int H, int s) {
  return lstm_args_ok(D, T, B, H, s) && H % 32 == 0 && H >= 32 && H <= 1024;
}

template <int BB, bool FWD>
static int lstm_fused_launch(const LstmFusedP &p, hipStream_t st) {
  const int groups = (p.B + BB - 1) / BB;
  const size_t lds = FWD ? ((size_t)p.K * BB + (size_t)(LSF_THREADS / 32) * 32 * BB) * sizeof(float)
                         : ((size_t)4 * p.H * BB + (size_t)LSF_THREADS * BB) * sizeof(float);
  auto kern = FWD ? lstm_fused_fwd_kernel<BB> : lstm_fused_bwd_kernel<BB>;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return TBG_EHIP;
  if (lds > 160 * 1024) return TBG_EUNSUPPORTED;
  hipLaunchKernelGGL(kern, dim3(p.H / LSF_UB, p.D, groups), dim3(LSF_THREADS), lds, st, p);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

// one forward step (s = 0 .. T-1 in order) of a frozen (bi)directional LSTM layer, projection included.
//   gx [D][T][B][4H], w_hh [D][4H][H], hT_in / hT_out [D][H][B] (two different buffers, used alternately; hT_in unused at s = 0),
//   act [D][T][B][4H], cs [D][T][B][H] (written at step index s, read at s - 1), seq [B][T][D H] or NULL
extern "C" int tbg_lstm_fused_fwd_f32(const float *gx, const float *w_hh, const float *hT_in, float *hT_out, float *act, float *cs,
                                      float *seq, int D, int T, int B, int H, int s, void *stream) {
  if (!gx || !w_hh || !hT_out || !act || !cs || (s > 0 && !hT_in) || hT_in == hT_out) return TBG_EINVAL;
  if (!lstm_fused_ok(D, T, B, H, s)) return lstm_args_ok(D, T, B, H, s) ? TBG_EUNSUPPORTED : TBG_EINVAL;
  LstmFusedP p{};
  p.gx = gx; p.w = w_hh; p.state_in = hT_in; p.state_out = hT_out; p.act = act; p.cs = cs; p.seq = seq;
  p.D = D; p.T = T; p.B = B; p.H = H; p.s = s; p.K = H; p.project = s > 0;
  hipStream_t st = tbg_stream(stream);
  switch (lstm_fused_bb(B, D * (H / LSF_UB))) {
    case 4: return lstm_fused_launch<4, true>(p, st);
    case 8: return lstm_fused_launch<8, true>(p, st);
    default: return lstm_fused_launch<16, true>(p, st);
  }
}

// one backward step (s = T-1 .. 0; first = 1 at s = T-1: no incoming recurrent / cell gradient).
//   dseq [B][T][D H] or NULL, w_hhT [D][H][4H] (the transposed recurrent weights), dgT_in / dgT_out [D][4H][B] (alternating buffers),
//   dc [D][B][H] (read unless first, then overwritten), act / cs as written by the forward steps, dg [D][T][B][4H] or NULL
extern "C" int tbg_lstm_fused_bwd_f32(const float *dseq, const float *w_hhT, const float *dgT_in, float *dgT_out, float *dc,
                                      const float *act, const float *cs, float *dg, int D, int T, int B, int H, int s, int first,
                                      void *stream) {
  if (!w_hhT || !dgT_out || !dc || !act || !cs || (!first && !dgT_in) || dgT_in == dgT_out) return TBG_EINVAL;
  if (!dseq && first) return TBG_EINVAL;
  if (!lstm_fused_ok(D, T, B, H, s)) return lstm_args_ok(D, T, B, H, s) ? TBG_EUNSUPPORTED : TBG_EINVAL;
  LstmFusedP p{};
  p.dseq = dseq; p.w = w_hhT; p.state_in = dgT_in; p.state_out = dgT_out; p.dc = dc; p.act_in = act; p.cs_in = cs; p.dg = dg;
  p.D = D; p.T = T; p.B = B; p.H = H; p.s = s; p.first = first;
  hipStream_t st = tbg_stream(stream);
  switch (lstm_fused_bb(B, D * (H / LSF_UB))) {
    case 4: return lstm_fused_launch<4, false>(p, st);
    case 8: return lstm_fused_launch<8, false>(p, st);
    default: return lstm_fused_launch<16, false>(p, st);
  }
}

// the decoder's cell (one direction, `steps` steps): the same launch with the input [context; hidden] -- stateT_in [K][B], K = E + H,
// w [4H][K] = [W_ih(context part) | W_hh] -- and gx[s] = the embedding row of the previous symbol (+ biases).  hT_out [H][B].
extern "C" int tbg_lstm_cell_fused_fwd_f32(const float *gx, const float *w, const float *stateT_in, float *hT_out, float *act,
                                           float *cs, int steps, int B, int H, int K, int s, void *stream) {
  if (!gx || !w || !stateT_in || !hT_out || !act || !cs || stateT_in == hT_out) return TBG_EINVAL;
  if (!lstm_args_ok(1, steps, B, H, s) || K < 1) return TBG_EINVAL;
  if (H % 32 != 0 || H > 1024 || K % LSF_CH != 0 || K > 2048) return TBG_EUNSUPPORTED;
  LstmFusedP p{};
  p.gx = gx; p.w = w; p.state_in = stateT_in; p.state_out = hT_out; p.act = act; p.cs = cs; p.seq = nullptr;
  p.D = 1; p.T = steps; p.B = B; p.H = H; p.s = s; p.K = K; p.project = 1;
  hipStream_t st = tbg_stream(stream);
  switch (lstm_fused_bb(B, H / LSF_UB)) {
    case 4: return lstm_fused_launch<4, true>(p, st);
    case 8: return lstm_fused_launch<8, true>(p, st);
    default: return lstm_fused_launch<16, true>(p, st);
  }
}

// outT [R][B] = Wt [R][J] applied to a transposed state: outT[r][b] = sum_j stateT[j][b] Wt[r][j] -- the GEMV of
// lstm_fused_bwd_kernel without a cell behind it (the decoder's backward: rows = [d(context); recurrent part of d(hidden)]).
struct RowsGemvP { const float *stateT, *w; float *outT; int J, R, B; };

template <int BB>
__global__ __launch_bounds__(LSF_THREADS) void rows_gemv_kernel(const RowsGemvP p) {
  extern __shared__ __attribute__((aligned(16))) float lsm[];
  constexpr int JS = LSF_THREADS / LSF_UB;
  const int J = p.J, B = p.B;
  float *st = lsm, *part = lsm + (size_t)J * BB;
  const int tid = threadIdx.x, r0 = blockIdx.x * LSF_UB, b0 = blockIdx.z * BB;
  const int nb = min(BB, B - b0);
  const int ul = tid & (LSF_UB - 1), js = tid / LSF_UB, NCH = J / LSF_CH;
  const float *wrow = p.w + (size_t)min(r0 + ul, p.R - 1) * J;
  float wv[LSF_CH];
#pragma unroll
  for (int i = 0; i < LSF_CH; i += 4) *reinterpret_cast<float4 *>(&wv[i]) = *reinterpret_cast<const float4 *>(wrow + min(js, NCH - 1) * LSF_CH + i);
  lstm_stage_state<BB>(st, p.stateT, J, B, b0, nb, tid);
  float acc[BB];
#pragma unroll
  for (int bb = 0; bb < BB; ++bb) acc[bb] = 0.f;
  __syncthreads();
  for (int ch = js; ch < NCH; ch += JS) {
    if (ch != js) {
#pragma unroll
      for (int i = 0; i < LSF_CH; i += 4) *reinterpret_cast<float4 *>(&wv[i]) = *reinterpret_cast<const float4 *>(wrow + ch * LSF_CH + i);
    }
    lstm_chunk_fma<BB>(acc, wv, st + (size_t)ch * LSF_CH * BB);
  }
#pragma unroll
  for (int bb = 0; bb < BB; bb += 4)
    *reinterpret_cast<float4 *>(part + ((size_t)js * LSF_UB + ul) * BB + bb) = make_float4(acc[bb], acc[bb + 1], acc[bb + 2], acc[bb + 3]);
  __syncthreads();
  const int pb = tid / LSF_UB, pu = tid % LSF_UB;
  if (tid < BB * LSF_UB && pb < nb && r0 + pu < p.R) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int k2 = 0; k2 < JS; k2 += 4) {
      a0 += part[((size_t)(k2 + 0) * LSF_UB + pu) * BB + pb]; a1 += part[((size_t)(k2 + 1) * LSF_UB + pu) * BB + pb];
      a2 += part[((size_t)(k2 + 2) * LSF_UB + pu) * BB + pb]; a3 += part[((size_t)(k2 + 3) * LSF_UB + pu) * BB + pb];
    }
    p.outT[(size_t)(r0 + pu) * B + b0 + pb] = (a0 + a1) + (a2 + a3);
  }
}

template <int BB>
static int rows_gemv_launch(const RowsGemvP &p, hipStream_t st) {
  const size_t lds = ((size_t)p.J * BB + (size_t)LSF_THREADS * BB) * sizeof(float);
  if (lds > 160 * 1024) return TBG_EUNSUPPORTED;
  auto kern = rows_gemv_kernel<BB>;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return TBG_EHIP;
  hipLaunchKernelGGL(kern, dim3((p.R + LSF_UB - 1) / LSF_UB, 1, (p.B + BB - 1) / BB), dim3(LSF_THREADS), lds, st, p);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

extern "C" int tbg_rows_gemv_t_f32(const float *stateT, const float *w, float *outT, int J, int R, int B, void *stream) {
  if (!stateT || !w || !outT || J < 1 || R < 1 || B < 1) return TBG_EINVAL;
  if (J % LSF_CH != 0) return TBG_EUNSUPPORTED;
  const RowsGemvP p{stateT, w, outT, J, R, B};
  hipStream_t st = tbg_stream(stream);
  switch (lstm_fused_bb(B, (R + LSF_UB - 1) / LSF_UB)) {
    case 4: return rows_gemv_launch<4>(p, st);
    case 8: return rows_gemv_launch<8>(p, st);
    default: return rows_gemv_launch<16>(p, st);
  }
}

// ============================================================================================
// The decoder's per-SAMPLE halves of a step (one block per image), fused around the two launches above:
//   forward  (after the cell of step s): logits[s] = b_o + h W_o^T, greedy symbol = argmax (first maximum, as torch), the next
//            step's embedding row, its attention query q = b_d + h W_d^T, the attention context of attn_ctx_fwd_kernel -- and the
//            context stored TRANSPOSED as the first E rows of the next cell launch's state.
//   backward (after the rows launch of step s): attention backward of attn_ctx_bwd_kernel (d(enc_proj) accumulated, dq), dh of the
//            step before = rows part + dq W_d + dl[s-1] W_o (precomputed), and that step's cell backward (lstm_step_bwd_kernel's
//            arithmetic) writing its gate gradients transposed for the next rows launch.
// Replaces, per step, 8 forward / 6 backward launches of library GEMMs, tbg_attn_ctx_*, tbg_lstm_step_* and torch index / argmax ops.
// ============================================================================================
struct DecFwdP {
  const float *hT, *w_oT, *b_o, *w_dT, *b_d, *ep, *enc, *v, *etab;
  float *logits, *gx_next, *q_out, *a_out, *ctxT_out;
  int B, T, H, E, C, go;
};

// 1024 threads per sample: every matvec is split over K as well, its weights fetched as ONE batch of independent loads per thread
// (the first version -- 256 threads walking K in a loop of load / multiply-add pairs -- paid one memory round trip per iteration:
// 27 us per launch).
#define DEC_THREADS 1024
__global__ __launch_bounds__(DEC_THREADS) void dec_sample_fwd_kernel(const DecFwdP p) {
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  const int B = p.B, T = p.T, H = p.H, E = p.E, Cn = p.C;
  float *hs = dsm, *qv = dsm + H, *lg = dsm + 2 * H;      // [H], [H], [C]
  float *scr = dsm + 2 * H + Cn;                           // partial sums: max(8 C, 4 H, 2 E) floats
  __shared__ float part[16][ATT_MAXT];
  __shared__ float a_s[ATT_MAXT];
  __shared__ int prev_s;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int k = tid; k < H; k += DEC_THREADS) hs[k] = p.hT ? p.hT[(size_t)k * B + b] : 0.f;
  __syncthreads();
  const bool nxt = p.gx_next != nullptr;
  // ---- logits partials (8 K slices) and query partials (4 K slices): all weight loads of a thread in flight together
  const int KL = H / 8, KQ = H / 4;  // (H % 32 == 0 checked on the host)
  if (p.logits) {
    const int c0 = tid & 127, ks = tid >> 7;
    for (int c = c0; c < Cn; c += 128) {
      float acc = 0.f;
      for (int k0 = ks * KL; k0 < (ks + 1) * KL; k0 += 32) {
        float w[32];
#pragma unroll
        for (int i = 0; i < 32; ++i) w[i] = p.w_oT[(size_t)(k0 + i) * Cn + c];
#pragma unroll
        for (int i = 0; i < 32; ++i) acc += hs[k0 + i] * w[i];
      }
      scr[ks * Cn + c] = acc;
    }
    __syncthreads();
    for (int c = tid; c < Cn; c += DEC_THREADS) {
      float l = p.b_o[c];
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2) l += scr[k2 * Cn + c];
      lg[c] = l;
      p.logits[(size_t)b * Cn + c] = l;
    }
    __syncthreads();
    if (wave == 0) {  // first maximum
      float best = -3.4e38f;
      int bi = 0x7fffffff;
      for (int c = lane; c < Cn; c += 64)
        if (lg[c] > best) { best = lg[c]; bi = c; }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const float ob = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
      }
      if (lane == 0) prev_s = bi == 0x7fffffff ? 0 : bi;
    }
  }
  if (!nxt) return;
  {
    const int j0 = tid & 255, ks = tid >> 8;
    for (int j = j0; j < H; j += 256) {
      float acc = 0.f;
      if (p.hT)
        for (int k0 = ks * KQ; k0 < (ks + 1) * KQ; k0 += 32) {
          float w[32];
#pragma unroll
          for (int i = 0; i < 32; ++i) w[i] = p.w_dT[(size_t)(k0 + i) * H + j];
#pragma unroll
          for (int i = 0; i < 32; ++i) acc += hs[k0 + i] * w[i];
        }
      scr[8 * Cn + ks * H + j] = acc;
    }
  }
  __syncthreads();  // (also orders prev_s)
  const int prev = p.logits ? prev_s : p.go;
  for (int j = tid; j < 4 * H; j += DEC_THREADS) p.gx_next[(size_t)b * 4 * H + j] = p.etab[(size_t)prev * 4 * H + j];
  for (int j = tid; j < H; j += DEC_THREADS) {
    const float q = p.b_d[j] + ((scr[8 * Cn + j] + scr[8 * Cn + H + j]) + (scr[8 * Cn + 2 * H + j] + scr[8 * Cn + 3 * H + j]));
    qv[j] = q;
    p.q_out[(size_t)b * H + j] = q;
  }
  __syncthreads();
  // ---- attention energies: thread (k, t residue): the T / 4 time steps of its residue class, loads first
  {
    const float *epb = p.ep + (size_t)b * T * H;
    const int tq = tid >> 8, k = tid & 255;  // 4 residue classes of t (1024 threads / 256 columns; H <= 256 checked on the host)
    const bool kok = k < H;
    float e[(ATT_MAXT + 3) / 4];
#pragma unroll
    for (int i = 0; i < (ATT_MAXT + 3) / 4; ++i) {
      const int t = tq + 4 * i;
      e[i] = (kok && t < T) ? epb[(size_t)t * H + k] : 0.f;
    }
    const float vk = kok ? p.v[k] : 0.f, qk = kok ? qv[k] : 0.f;
    // the wave sums of this residue class together (common.h wave_tree_sum: 15 + 2 lane exchanges; one chain of six per time step before)
    constexpr int NE = (ATT_MAXT + 3) / 4;
    static_assert(NE == 16, "wave_tree_sum takes a power of two");
#pragma unroll
    for (int i = 0; i < NE; ++i) e[i] = (tq + 4 * i < T) ? vk * tanhf(e[i] + qk) : 0.f;
    wave_tree_sum<NE, NE, 32>(e, lane);
    const int ti = tq + 4 * wave_tree_row<NE>(lane);
    if ((lane & 3) == 0 && ti < T) part[wave & 3][ti] = e[0];
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
      p.a_out[(size_t)b * T + lane] = a;
    }
  }
  __syncthreads();
  // ---- context: thread (column j, half of the time steps)
  {
    const float *eb = p.enc + (size_t)b * T * E;
    const int th = tid >> 9, TH = (T + 1) / 2;
    float *cp = scr;  // [2][E]
    for (int j = tid & 511; j < E; j += 512) {
      float c = 0.f;
      for (int i0 = 0; i0 < TH; i0 += 8) {  // eight loads in flight
        float x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int t = th * TH + i0 + i;
          x[i] = (i0 + i < TH && t < T) ? eb[(size_t)t * E + j] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int t = th * TH + i0 + i;
          if (i0 + i < TH && t < T) c += a_s[t] * x[i];
        }
      }
      cp[th * E + j] = c;
    }
    __syncthreads();
    for (int j = tid; j < E; j += DEC_THREADS) p.ctxT_out[(size_t)j * B + b] = cp[j] + cp[E + j];
  }
}

struct DecBwdP {
  const float *dctxT, *dhpT, *a, *q, *ep, *enc, *v, *w_d, *dlo, *act, *cs_cur, *cs_prev;
  float *dep, *dctx_out, *dc, *dgT_out;
  int B, T, H, E, first;
};

__global__ __launch_bounds__(DEC_THREADS) void dec_sample_bwd_kernel(const DecBwdP p) {
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  const int B = p.B, T = p.T, H = p.H, E = p.E;
  float *dcx = dsm, *dq_s = dsm + E, *scr = dsm + E + H;   // [E], [H], partial sums: 4 H floats
  __shared__ float part[16][ATT_MAXT];
  __shared__ float de_s[ATT_MAXT], a_s[ATT_MAXT];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool att = p.dctxT != nullptr;
  if (att) {
    for (int e = tid; e < E; e += DEC_THREADS) {
      const float d = p.dctxT[(size_t)e * B + b];
      dcx[e] = d;
      p.dctx_out[(size_t)b * E + e] = d;
    }
    if (tid < T) a_s[tid] = p.a[(size_t)b * T + tid];
    __syncthreads();
    {  // da[t] = <dctx, enc[t]>: thread (column j, half of the time steps), loads first, one wave reduction per time step
      const float *eb = p.enc + (size_t)b * T * E;
      const int th = tid >> 9, TH = (T + 1) / 2;
      for (int i0 = 0; i0 < TH; i0 += 8) {  // eight time steps at a time: loads first, one wave reduction per time step
        float da[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) da[i] = 0.f;
        for (int j = tid & 511; j < E; j += 512) {
          float x[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int t = th * TH + i0 + i;
            x[i] = (i0 + i < TH && t < T) ? eb[(size_t)t * E + j] : 0.f;
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) da[i] += dcx[j] * x[i];
        }
        {  // the eight wave sums together (7 + 3 lane exchanges): lane 8 i ends up with the total of da[i]
          wave_tree_sum<8, 8, 32>(da, lane);
          const int i = wave_tree_row<8>(lane), t = th * TH + i0 + i;
          if ((lane & 7) == 0 && i0 + i < TH && t < T) part[wave & 7][t] = da[0];  // (8 waves per half)
        }
      }
    }
    __syncthreads();
    if (wave == 0) {
      float da = 0.f;
      if (lane < T)
#pragma unroll
        for (int w2 = 0; w2 < 8; ++w2) da += part[w2][lane];
      const float a = lane < T ? a_s[lane] : 0.f;
      const float dot = wave_sum(a * da);
      if (lane < T) de_s[lane] = a * (da - dot);
    }
    __syncthreads();
    {  // d(enc_proj) accumulated, dq: thread (column k, residue class of t)
      const float *epb = p.ep + (size_t)b * T * H;
      const int tq = tid >> 8;
      for (int k = tid & 255; k < H; k += 256) {
        const float qk = p.q[(size_t)b * H + k], vk = p.v[k];
        float acc = 0.f;
        for (int i0 = 0; tq + 4 * i0 < T; i0 += 8) {  // eight read-modify-writes in flight
          float e8[8], o8[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int t = min(tq + 4 * (i0 + i), T - 1);
            e8[i] = epb[(size_t)t * H + k];
            o8[i] = p.dep[((size_t)b * T + t) * H + k];
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int t = tq + 4 * (i0 + i);
            if (t < T) {
              const float th = tanhf(e8[i] + qk);
              const float dp = de_s[t] * vk * (1.f - th * th);
              p.dep[((size_t)b * T + t) * H + k] = o8[i] + dp;
              acc += dp;
            }
          }
        }
        scr[tq * H + k] = acc;
      }
    }
    __syncthreads();
    for (int k = tid; k < H; k += DEC_THREADS) dq_s[k] = (scr[k] + scr[H + k]) + (scr[2 * H + k] + scr[3 * H + k]);
    __syncthreads();
  }
  if (!p.act) return;
  if (att) {  // dq W_d: thread (column u, quarter of the rows), the rows' weights as batches of independent loads
    const int KQ = H / 4, js = tid >> 8;
    for (int u = tid & 255; u < H; u += 256) {
      float acc = 0.f;
      for (int j0 = js * KQ; j0 < (js + 1) * KQ; j0 += 16) {
        float w[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) w[i] = p.w_d[(size_t)(j0 + i) * H + u];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += dq_s[j0 + i] * w[i];
      }
      scr[js * H + u] = acc;
    }
    __syncthreads();
  }
  for (int u = tid; u < H; u += DEC_THREADS) {
    float dh = p.dlo[(size_t)b * H + u];
    if (att) dh += p.dhpT[(size_t)u * B + b] + ((scr[u] + scr[H + u]) + (scr[2 * H + u] + scr[3 * H + u]));
    const float *a = p.act + (size_t)b * 4 * H;
    const float i_ = a[u], f_ = a[H + u], g_ = a[2 * H + u], o_ = a[3 * H + u];
    const float c = p.cs_cur[(size_t)b * H + u];
    const float cp = p.cs_prev ? p.cs_prev[(size_t)b * H + u] : 0.f;
    const float tc = tanhf(c);
    const float dcc = dh * o_ * (1.f - tc * tc) + (p.first ? 0.f : p.dc[(size_t)b * H + u]);
    p.dc[(size_t)b * H + u] = dcc * f_;
    p.dgT_out[(size_t)u * B + b] = dcc * g_ * i_ * (1.f - i_);
    p.dgT_out[(size_t)(H + u) * B + b] = dcc * cp * f_ * (1.f - f_);
    p.dgT_out[(size_t)(2 * H + u) * B + b] = dcc * i_ * (1.f - g_ * g_);
    p.dgT_out[(size_t)(3 * H + u) * B + b] = dh * tc * o_ * (1.f - o_);
  }
}

// forward half of a decoder step for every sample.  hT NULL: the initial launch (h = 0, symbol = go); logits NULL: no logits / argmax
// (the initial launch); gx_next NULL: last step (only the logits).  Shapes in the DecFwdP comment above; H % 4 == 0, T <= 64.
extern "C" int tbg_dec_sample_fwd_f32(const float *hT, const float *w_oT, const float *b_o, const float *w_dT, const float *b_d,
                                      const float *enc_proj, const float *enc, const float *v, const float *etab, float *logits,
                                      float *gx_next, float *q_out, float *a_out, float *ctxT_out, int B, int T, int H, int E, int C,
                                      int go, void *stream) {
  if (B < 1 || T < 1 || H < 1 || E < 1 || C < 1 || go < 0 || (!logits && !gx_next)) return TBG_EINVAL;
  if (logits && (!hT || !w_oT || !b_o)) return TBG_EINVAL;
  if (gx_next && (!w_dT || !b_d || !enc_proj || !enc || !v || !etab || !q_out || !a_out || !ctxT_out)) return TBG_EINVAL;
  if (T > ATT_MAXT || (H & 31) != 0 || H > 256 || E > 512 * 8) return TBG_EUNSUPPORTED;
  const DecFwdP p{hT, w_oT, b_o, w_dT, b_d, enc_proj, enc, v, etab, logits, gx_next, q_out, a_out, ctxT_out, B, T, H, E, C, go};
  const size_t scr = (size_t)8 * C + 4 * H > (size_t)2 * E ? (size_t)8 * C + 4 * H : (size_t)2 * E;
  const size_t lds = ((size_t)2 * H + C + scr) * sizeof(float);
  if (lds > 48 * 1024) return TBG_EUNSUPPORTED;
  hipLaunchKernelGGL(dec_sample_fwd_kernel, dim3(B), dim3(DEC_THREADS), lds, tbg_stream(stream), p);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

// backward half of a decoder step for every sample: dctxT NULL: no attention part (the first launch of the backward pass);
// act NULL: no cell part (the last).  first = 1: the cell part is the last step's (dc not read).
extern "C" int tbg_dec_sample_bwd_f32(const float *dctxT, const float *dhpT, const float *a, const float *q, const float *enc_proj,
                                      const float *enc, const float *v, const float *w_d, float *denc_proj, float *dctx_out,
                                      const float *dlo, const float *act, const float *cs_cur, const float *cs_prev, float *dc,
                                      float *dgT_out, int B, int T, int H, int E, int first, void *stream) {
  if (B < 1 || T < 1 || H < 1 || E < 1 || (!dctxT && !act)) return TBG_EINVAL;
  if (dctxT && (!a || !q || !enc_proj || !enc || !v || !denc_proj || !dctx_out)) return TBG_EINVAL;
  if (act && (!dlo || !cs_cur || !dc || !dgT_out || (dctxT && (!dhpT || !w_d)))) return TBG_EINVAL;
  if (T > ATT_MAXT || (H & 31) != 0 || H > 256) return TBG_EUNSUPPORTED;
  const DecBwdP p{dctxT, dhpT, a, q, enc_proj, enc, v, w_d, dlo, act, cs_cur, cs_prev, denc_proj, dctx_out, dc, dgT_out, B, T, H, E, first};
  const size_t lds = ((size_t)E + 5 * H) * sizeof(float);
  if (lds > 48 * 1024) return TBG_EUNSUPPORTED;
  hipLaunchKernelGGL(dec_sample_bwd_kernel, dim3(B), dim3(DEC_THREADS), lds, tbg_stream(stream), p);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}
