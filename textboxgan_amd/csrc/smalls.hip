// Small-tensor tails of the layer gradients (gfx950).  Each of these replaces a chain of 6-25 tiny framework launches
// (reductions, squares, small GEMMs, fills) that cost ~5 us apiece whatever their size: on this path ~1600 of the ~2100
// launches of a G+D step were such tails (profiles/r02_census_gd.txt), 9 ms of a 33 ms step.
#include "common.h"

// ---------------------------------------------------------------------------------------------------------------
// tail of the modulated-conv backward (modulated_conv2d.py:78-82 in the activation-scaling form):
//   t[b,o]    = (sum_ch pdy[b,o,ch]) * d[b,o]^2                 pdy: partial sums of dpre * y_rec from bias_act_bwd
//   ds[b,i]   = ds_conv[b,i] - s[b,i] * sum_o t[b,o] wsq[i,o]   (gradient of the style scale, incl. the demod term)
//   dwsq[i,o] = sum_b s[b,i]^2 t[b,o]                           (enters dW through tbg_conv2d_wgrad_ex_f32's addq)
//   db[o]     = sum_{b,ch} pdb[b,o,ch];   dstrength = sum pdn
// grid: ceil(I / MS_IT) blocks own MS_IT input channels each (all b, all o) + ceil(O / 256) blocks for db / dstrength.
// ---------------------------------------------------------------------------------------------------------------
struct ModSmallP {
  const float *pdb, *pdn, *pdy, *d, *s, *wsq, *ds_conv;
  float *db, *dstrength, *ds, *dwsq;
  int B, I, O, nch, ds_slots;
};

#define MS_IT 2  // input channels per block: 256 blocks at I = 512 (16: 33 blocks, 30 us; 4: 12.6 us; 2: 11.3 us at B = 16, 16.2 vs 20.8 at B = 32 -- the per-block LDS dot chains scale with it, tools/bench_smalls.py)

__global__ __launch_bounds__(256) void modconv_bwd_smalls_kernel(const ModSmallP p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x;
  const int nI = (p.I + MS_IT - 1) / MS_IT;
  if ((int)blockIdx.x >= nI) {  // ---- bias / noise-strength gradients: blocks nI .. nI + ceil(O/256) - 1
    __shared__ float red[4];
    const int o = ((int)blockIdx.x - nI) * 256 + tid;
    if (o < p.O) {  // sixteen samples per trip: independent load chains (a single chain was one round trip per element)
      float a16[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) a16[u] = 0.f;
      for (int b = 0; b < p.B; b += 16)
        for (int c = 0; c < p.nch; ++c) {
#pragma unroll
          for (int u = 0; u < 16; ++u)
            if (b + u < p.B) a16[u] += p.pdb[((size_t)(b + u) * p.O + o) * p.nch + c];
        }
      float a = 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u) a += a16[u];
      p.db[o] = a;
    }
    if ((int)blockIdx.x == nI && p.pdn) {
      float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const int n = p.B * p.O * p.nch;
      for (int i = tid; i < n; i += 256 * 8) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (i + 256 * u < n) a8[u] += p.pdn[i + 256 * u];
      }
      float a = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
      a = wave_sum(a);
      if ((tid & 63) == 0) red[tid >> 6] = a;
      __syncthreads();
      if (tid == 0) p.dstrength[0] = red[0] + red[1] + red[2] + red[3];
    }
    return;
  }
  float *t = sm;                          // [B][O]
  float *wq = t + p.B * p.O;              // [MS_IT][O]
  float *s2 = wq + MS_IT * p.O;           // [B][MS_IT]
  float *part = s2 + p.B * MS_IT;         // [B][MS_IT][4] partial dots
  const int i0 = blockIdx.x * MS_IT;
  {  // t[b,o]: TU (b,o) entries per trip with ALL their partial-sum loads in flight together: every block re-derives the whole
     // [B, O] table, and one entry per trip was a chain of B*O/256 dependent round trips (most of the first version's 29 us;
     // four per trip: 15 us at B = 16, 26 us at B = 32; sixteen per trip: the table is 2 - 4 trips)
    constexpr int TU = 16;
    const int BO = p.B * p.O;
    for (int e0 = tid; e0 < BO; e0 += 256 * TU) {
      float a[TU], dv[TU];
      int ee[TU];
#pragma unroll
      for (int u = 0; u < TU; ++u) { ee[u] = min(e0 + 256 * u, BO - 1); a[u] = 0.f; }
#pragma unroll
      for (int u = 0; u < TU; ++u) dv[u] = p.d[ee[u]];
      if (p.nch <= 4) {  // (the step's chunk counts: 1 - 3) every load of the trip issued before the first use
        float v[TU][4];
#pragma unroll
        for (int u = 0; u < TU; ++u)
#pragma unroll
          for (int c = 0; c < 4; ++c) v[u][c] = p.pdy[(size_t)ee[u] * p.nch + min(c, p.nch - 1)];
#pragma unroll
        for (int u = 0; u < TU; ++u)
#pragma unroll
          for (int c = 0; c < 4; ++c) a[u] += c < p.nch ? v[u][c] : 0.f;
      } else {
        for (int c = 0; c < p.nch; ++c) {
#pragma unroll
          for (int u = 0; u < TU; ++u) a[u] += p.pdy[(size_t)ee[u] * p.nch + c];
        }
      }
#pragma unroll
      for (int u = 0; u < TU; ++u)
        if (e0 + 256 * u < BO) t[e0 + 256 * u] = a[u] * dv[u] * dv[u];
    }
  }
  for (int e0 = tid; e0 < MS_IT * p.O; e0 += 256 * 8) {  // 8 loads in flight per lane
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = min(e0 + 256 * u, MS_IT * p.O - 1);
      const int ii = e / p.O, o = e - ii * p.O;
      v[u] = p.wsq[(size_t)min(i0 + ii, p.I - 1) * p.O + o];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + 256 * u;
      if (e < MS_IT * p.O) wq[e] = (i0 + e / p.O < p.I) ? v[u] : 0.f;
    }
  }
  for (int e = tid; e < p.B * MS_IT; e += 256) {
    const int b = e / MS_IT, ii = e - b * MS_IT;
    s2[e] = (i0 + ii < p.I) ? p.s[(size_t)b * p.I + i0 + ii] : 0.f;
  }
  __syncthreads();
  // ds: (b, ii) pairs x 4 lanes that split the o range (consecutive lanes -> consecutive o: conflict-free LDS reads)
  for (int e = tid; e < p.B * MS_IT * 4; e += 256) {
    const int q = e & 3, pair = e >> 2;
    const int b = pair / MS_IT, ii = pair - b * MS_IT;
    const float *tb = t + b * p.O, *wr = wq + ii * p.O;
    // eight LDS read pairs in flight per trip, four accumulators (a single chain was one LDS round trip per term: 128 of them)
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int o = q;
    for (; o + 28 < p.O; o += 32) {
      float tv[8], wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { tv[u] = tb[o + 4 * u]; wv[u] = wr[o + 4 * u]; }
      a0 += tv[0] * wv[0] + tv[4] * wv[4]; a1 += tv[1] * wv[1] + tv[5] * wv[5];
      a2 += tv[2] * wv[2] + tv[6] * wv[6]; a3 += tv[3] * wv[3] + tv[7] * wv[7];
    }
    for (; o < p.O; o += 4) a0 += tb[o] * wr[o];
    part[e] = (a0 + a1) + (a2 + a3);
  }
  __syncthreads();
  for (int pair = tid; pair < p.B * MS_IT; pair += 256) {
    const int b = pair / MS_IT, ii = pair - b * MS_IT;
    if (i0 + ii < p.I) {
      const float a = part[4 * pair] + part[4 * pair + 1] + part[4 * pair + 2] + part[4 * pair + 3];
      const size_t idx = (size_t)b * p.I + i0 + ii;
      float dsc = 0.f;  // the convolution's style dot: its per-(pixel tile, wave column) partial sums, in slot order
      for (int k = 0; k < p.ds_slots; ++k) dsc += p.ds_conv[idx * p.ds_slots + k];
      p.ds[idx] = dsc - s2[pair] * a;
    }
  }
  for (int e = tid; e < MS_IT * p.O; e += 256) {  // dwsq
    const int ii = e / p.O, o = e - ii * p.O;
    if (i0 + ii < p.I) {
      float a0 = 0.f, a1 = 0.f;
      int b = 0;
      for (; b + 8 <= p.B; b += 8) {  // eight LDS read pairs in flight per trip
        float sv[8], tv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { sv[u] = s2[(b + u) * MS_IT + ii]; tv[u] = t[(b + u) * p.O + o]; }
#pragma unroll
        for (int u = 0; u < 8; u += 2) { a0 += sv[u] * sv[u] * tv[u]; a1 += sv[u + 1] * sv[u + 1] * tv[u + 1]; }
      }
      for (; b < p.B; ++b) {
        const float sv = s2[b * MS_IT + ii];
        a0 += sv * sv * t[b * p.O + o];
      }
      p.dwsq[(size_t)(i0 + ii) * p.O + o] = a0 + a1;
    }
  }
}

extern "C" int tbg_modconv_bwd_smalls_f32(const float *pdb, const float *pdn, const float *pdy, const float *d,
                                          const float *s, const float *wsq, const float *ds_conv, float *db,
                                          float *dstrength, float *ds, float *dwsq, int B, int I, int O, int nch,
                                          int ds_slots, void *stream) {
  if (!pdb || !pdy || !d || !s || !wsq || !ds_conv || !db || !ds || !dwsq) return TBG_EINVAL;
  if (B < 1 || I < 1 || O < 1 || nch < 1 || ds_slots < 1 || ((pdn == nullptr) != (dstrength == nullptr))) return TBG_EINVAL;
  const size_t lds = ((size_t)B * O + (size_t)MS_IT * O + (size_t)B * MS_IT * 5) * sizeof(float);
  if (lds > 160 * 1024) return TBG_EUNSUPPORTED;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void *>(modconv_bwd_smalls_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds) != hipSuccess)
    return TBG_EHIP;
  ModSmallP p{pdb, pdn, pdy, d, s, wsq, ds_conv, db, dstrength, ds, dwsq, B, I, O, nch, ds_slots};
  const int blocks = (I + MS_IT - 1) / MS_IT + (O + 255) / 256;
  hipLaunchKernelGGL(modconv_bwd_smalls_kernel, dim3(blocks), dim3(256), lds, tbg_stream(stream), p);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// tail of the toRGB backward (to_rgb.py:28-33): from the channel Gram G[b,c,o] = sum_p x[b,c,p] dy[b,o,p]
//   ds[b,c] = coef * sum_o G[b,c,o] w[c,o];     dw[c,o] = coef * sum_b G[b,c,o] s[b,c]
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void torgb_bwd_smalls_kernel(const float *__restrict__ G, const float *__restrict__ w,
                                                              const float *__restrict__ s, float *__restrict__ ds,
                                                              float *__restrict__ dw, int B, int C, int O, float coef, int nchunk,
                                                              const float *__restrict__ dysum, float *__restrict__ db) {
  // G [B][C][nchunk][O]: per-pixel-chunk partial sums of tbg_rgb_backproject_f32, summed here in chunk order.  Every loop below loads
  // eight values per trip before it adds the first (one value per trip was one memory round trip per value: B x nchunk = 128 of them
  // in a row for the filter gradient, 15.7 us per launch); the additions keep their order.
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < B * C) {
    const int c = e % C;
    float a = 0.f;
    for (int o = 0; o < O; ++o) {
      float g = 0.f;
      for (int k0 = 0; k0 < nchunk; k0 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = k0 + u < nchunk ? G[((size_t)e * nchunk + k0 + u) * O + o] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (k0 + u < nchunk) g += v[u];
      }
      a += g * w[c * O + o];
    }
    ds[e] = a * coef;
  }
  if (e < C * O) {
    const int c = e / O, o = e - c * O;
    float a = 0.f, g = 0.f;
    const int n = B * nchunk;  // (b, k) pairs in order: b = i / nchunk, k = i % nchunk
    for (int i0 = 0; i0 < n; i0 += 8) {
      float v[8], sv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = min(i0 + u, n - 1), b = i / nchunk, k = i - b * nchunk;
        v[u] = G[(((size_t)b * C + c) * nchunk + k) * O + o];
        sv[u] = s[(size_t)b * C + c];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + u;
        if (i < n) {
          g += v[u];
          if ((i + 1) % nchunk == 0) { a += g * sv[u]; g = 0.f; }  // the last chunk of sample b
        }
      }
    }
    dw[e] = a * coef;
  }
  if (db && e < O) {  // db[o] = sum_{b, chunk} dysum[b][chunk][o]  (the bias gradient: sum of the masked dy over samples and pixels)
    float a = 0.f;
    const int n = B * nchunk;
    for (int i0 = 0; i0 < n; i0 += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = i0 + u < n ? dysum[(size_t)(i0 + u) * O + e] : 0.f;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (i0 + u < n) a += v[u];
    }
    db[e] = a;
  }
}

extern "C" int tbg_torgb_bwd_smalls_f32(const float *G, const float *w, const float *s, float *ds, float *dw, int B, int C,
                                        int O, float coef, int nchunk, const float *dysum, float *db, void *stream) {
  if (!G || !w || !s || !ds || !dw || B < 1 || C < 1 || O < 1 || nchunk < 1 || ((dysum == nullptr) != (db == nullptr))) return TBG_EINVAL;
  const int n = B * C > C * O ? B * C : C * O;
  hipLaunchKernelGGL(torgb_bwd_smalls_kernel, dim3((n + 255) / 256), dim3(256), 0, tbg_stream(stream), G, w, s, ds, dw, B, C, O,
                     coef, nchunk, dysum, db);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// minibatch standard deviation (mini_batch_std.py:10-35), first order.  x [B,C,HW] with B = G*M (G = min(4, B) samples per
// statistics group, group member g of group m is sample g*M + m):
//   std[m,e] = sqrt(mean_g (x[g,m,e] - mean_g x)^2 + 1e-8);  stat[m] = mean_e std[m,e]
//   y [B, C+1, HW] = cat(x, stat[n mod M] broadcast)
// backward: dstat[m] = sum_{g,p} dy[g*M+m, C, p];  dx[g,m,e] = dy[g,m,e] + dstat[m]/(E*G) * (x[g,m,e] - mean)/std[m,e]
// One block per statistics group m.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float *red) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  const float r = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  return r;
}

#define MB_NCH 16  // element chunks per statistics group: M x 16 blocks (one block per group took 100 us at M = 4)

__global__ __launch_bounds__(256) void mbstd_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, int G, int M, int C,
                                                       int HW) {
  __shared__ float red[4];
  const int m = blockIdx.x, E = C * HW;
  // chunk 0 of a group computes the group's statistic (a small, L2-resident reduction over all E) and writes the extra
  // channel; the other chunks only copy their share of x
  float stat = 0.f;
  if (blockIdx.y == 0) {
    float acc = 0.f;
    // four elements per trip, all their G x 4 loads in flight together (one element per trip was E / 256 = 32 dependent round trips:
    // most of this kernel's 37 us); same per-element arithmetic, same order of the additions into acc
    for (int e0 = threadIdx.x; e0 < E; e0 += 256 * 4) {
      float v[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int e = e0 + 256 * u;
          v[u][g] = (g < G && e < E) ? x[(size_t)(g * M + m) * E + e] : 0.f;
        }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (e0 + 256 * u >= E) break;
        float mean = 0.f;
        for (int g = 0; g < G; ++g) mean += v[u][g];
        mean /= G;
        float var = 0.f;
        for (int g = 0; g < G; ++g) { const float dlt = v[u][g] - mean; var += dlt * dlt; }
        acc += sqrtf(var / G + 1e-8f);
      }
    }
    stat = block_sum(acc, red) / E;
  }
  const int per = (E + MB_NCH - 1) / MB_NCH, e0 = blockIdx.y * per, e1 = min(e0 + per, E);
  for (int g = 0; g < G; ++g)
    for (int e = e0 + threadIdx.x; e < e1; e += 256)
      y[(size_t)(g * M + m) * (E + HW) + e] = x[(size_t)(g * M + m) * E + e];
  if (blockIdx.y == 0)
    for (int i = threadIdx.x; i < G * HW; i += 256) {
      const int g = i / HW, px = i - g * HW;
      y[(size_t)(g * M + m) * (E + HW) + E + px] = stat;
    }
}

__global__ __launch_bounds__(256) void mbstd_bwd_kernel(const float *__restrict__ x, const float *__restrict__ dy,
                                                       float *__restrict__ dx, int G, int M, int C, int HW) {
  __shared__ float red[4];
  const int m = blockIdx.x, E = C * HW;
  float a = 0.f;
  for (int i = threadIdx.x; i < G * HW; i += 256) {
    const int g = i / HW, px = i - g * HW;
    a += dy[(size_t)(g * M + m) * (E + HW) + E + px];
  }
  const float k = block_sum(a, red) / ((float)E * G);
  const int per = (E + MB_NCH - 1) / MB_NCH, e0 = blockIdx.y * per, e1 = min(e0 + per, E);
  for (int e = e0 + threadIdx.x; e < e1; e += 256) {
    float v[4], mean = 0.f;
    for (int g = 0; g < G; ++g) { v[g] = x[(size_t)(g * M + m) * E + e]; mean += v[g]; }
    mean /= G;
    float var = 0.f;
    for (int g = 0; g < G; ++g) { const float dlt = v[g] - mean; var += dlt * dlt; }
    const float inv = k / sqrtf(var / G + 1e-8f);
    for (int g = 0; g < G; ++g) dx[(size_t)(g * M + m) * E + e] = dy[(size_t)(g * M + m) * (E + HW) + e] + (v[g] - mean) * inv;
  }
}

extern "C" int tbg_minibatch_std_fwd_f32(const float *x, float *y, int B, int C, int HW, int group, void *stream) {
  if (!x || !y || B < 1 || C < 1 || HW < 1 || group < 1) return TBG_EINVAL;
  const int G = group < B ? group : B;
  if (G > 4 || B % G != 0) return TBG_EINVAL;  // mini_batch_std.py: the batch must be a multiple of the group size
  hipLaunchKernelGGL(mbstd_fwd_kernel, dim3(B / G, MB_NCH), dim3(256), 0, tbg_stream(stream), x, y, G, B / G, C, HW);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

extern "C" int tbg_minibatch_std_bwd_f32(const float *x, const float *dy, float *dx, int B, int C, int HW, int group,
                                         void *stream) {
  if (!x || !dy || !dx || B < 1 || C < 1 || HW < 1 || group < 1) return TBG_EINVAL;
  const int G = group < B ? group : B;
  if (G > 4 || B % G != 0) return TBG_EINVAL;
  hipLaunchKernelGGL(mbstd_bwd_kernel, dim3(B / G, MB_NCH), dim3(256), 0, tbg_stream(stream), x, dy, dx, G, B / G, C, HW);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// equalised-LR dense layers of the mapping network and the per-layer style affines (dense.py:23-29 + bias_act.py:25-34):
//   out[r,n] = act(alpha * sum_k x[r,k] w[k,n] + beta * b[n]) + offset          R <= a few dozen rows, K, N <= 512
// These are 1-4 MFLOP problems that went through 3 (forward) and 5 (backward) library launches each -- ~240 of the step's
// launches.  One launch per direction here; the K (forward) / row (backward) reductions are split over the block's waves.
// ---------------------------------------------------------------------------------------------------------------
struct DenseP {
  const float *x, *w, *b, *out_in, *dout;
  float *out, *dx, *dw, *db;
  int R, K, N, lrelu, nb_dw, ldx;  // ldx: row pitch of x and dx (floats)
  float alpha, beta, offset;
};

#define DN_RT 16   // rows per block
#define DN_KT 64   // k per thread and pass (all their filter loads are issued before the first use: one latency round)

// block = 64 output columns x 4 k-quarters, DN_RT rows; dynamic LDS: xs[DN_RT][K + 4] + red[3][DN_RT][64]
__device__ __forceinline__ void dense_fwd_body(const DenseP &p, const int bx, const int by) {
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  const int KP = ((p.K + 4 * DN_KT - 1) / (4 * DN_KT)) * (4 * DN_KT) + 4;  // whole (zero-filled) passes: no k guards on the reads
  float *xs = dsm;                    // [DN_RT][KP]
  float *red = dsm + DN_RT * KP;      // [3][DN_RT][64]
  const int tid = threadIdx.x, nl = tid & 63, kq = tid >> 6;
  const int n = bx * 64 + nl, r0 = by * DN_RT;
  const int nc = min(n, p.N - 1);
  float acc[DN_RT];
#pragma unroll
  for (int r = 0; r < DN_RT; ++r) acc[r] = 0.f;
  // first filter pass in flight while x is staged
  float wv[DN_KT];
  const int kspan = 4 * DN_KT;
#pragma unroll
  for (int j = 0; j < DN_KT; ++j) {
    const int k = kq * DN_KT + j;
    wv[j] = (k < p.K) ? p.w[(size_t)k * p.N + nc] : 0.f;
  }
  // x rows.  Rows of whole 16-byte units: float4 loads, 8 per lane in flight (ONE round for K <= 512); the tail of the padded
  // row (k >= K) is zero-filled by the scalar loop below
  const bool xvec = (p.K & 3) == 0 && (p.ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(p.x) & 15) == 0;
  if (xvec) {
    const int KQ = KP >> 2;  // quads per padded row (KP % 4 == 0)
    for (int e0 = tid; e0 < DN_RT * KQ; e0 += 256 * 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + 256 * u;
        const int r = e / KQ, kk = (e - r * KQ) * 4;
        const bool ok = e < DN_RT * KQ && r0 + r < p.R && kk < p.K;
        v[u] = *reinterpret_cast<const float4 *>(p.x + (ok ? (size_t)(r0 + r) * p.ldx + kk : 0));
        if (!ok) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (e0 + 256 * u < DN_RT * KQ) *reinterpret_cast<float4 *>(xs + (size_t)(e0 + 256 * u) * 4) = v[u];
    }
  }
  // scalar form: batches of 8 loads per lane in flight (a load-store-load-store loop is one HBM round trip per element)
  for (int e0 = tid; !xvec && e0 < DN_RT * KP; e0 += 256 * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int e = e0 + 256 * u;
      const int r = e / KP, kk = e - r * KP;
      v[u] = (e < DN_RT * KP && r0 + r < p.R && kk < p.K) ? p.x[(size_t)(r0 + r) * p.ldx + kk] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (e0 + 256 * u < DN_RT * KP) xs[e0 + 256 * u] = v[u];
  }
  __syncthreads();
  for (int kb = 0; kb < p.K; kb += kspan) {
    const int k0 = kb + kq * DN_KT;
    if (k0 < p.K) {  // wave-uniform
#pragma unroll
      for (int j = 0; j < DN_KT; j += 4) {
#pragma unroll
        for (int r = 0; r < DN_RT; ++r) {
          const float4 xv = *reinterpret_cast<const float4 *>(&xs[r * KP + k0 + j]);
          acc[r] += xv.x * wv[j] + xv.y * wv[j + 1] + xv.z * wv[j + 2] + xv.w * wv[j + 3];
        }
      }
    }
    if (kb + kspan < p.K) {
#pragma unroll
      for (int j = 0; j < DN_KT; ++j) {
        const int k = kb + kspan + kq * DN_KT + j;
        wv[j] = (k < p.K) ? p.w[(size_t)k * p.N + nc] : 0.f;
      }
    }
  }
  if (kq > 0) {
#pragma unroll
    for (int r = 0; r < DN_RT; ++r) red[((kq - 1) * DN_RT + r) * 64 + nl] = acc[r];
  }
  __syncthreads();
  if (kq == 0 && n < p.N) {
    const float bb = p.b ? p.beta * p.b[n] : 0.f;
#pragma unroll
    for (int r = 0; r < DN_RT; ++r) {
      if (r0 + r < p.R) {
        const float pre = p.alpha * (acc[r] + red[r * 64 + nl] + red[(DN_RT + r) * 64 + nl] + red[(2 * DN_RT + r) * 64 + nl]) + bb;
        p.out[(size_t)(r0 + r) * p.N + n] = ((p.lrelu && pre < 0.f) ? 0.2f * pre : pre) + p.offset;
      }
    }
  }
}

// gm = dout * act'(pre);   dw[k,n] = alpha sum_r x[r,k] gm[r,n];   db[n] = beta sum_r gm[r,n];   dx[r,k] = alpha sum_n gm[r,n] w[k,n]
// blocks [0, nb_dw): one 16(k) x 64(n) tile of dw over all rows (and db from the k-tile-0 blocks);
// blocks [nb_dw, ..): 16 filter rows k (4 per wave) x DN_RT rows of dx: a wave reads its filter rows coalesced along n (all loads
// issued before the gm tile is staged), accumulates per-lane partial sums for the DN_RT rows and folds them across the wave.
#define DN_NMAX 512  // columns a dx block keeps in registers per filter row (8 per lane); wider layers loop
__device__ __forceinline__ void dense_bwd_body(const DenseP &p, const int bx) {
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  const int tid = threadIdx.x, l = tid & 63, q = tid >> 6;
  const int ntn = (p.N + 63) / 64;
  // gm tile [nrows][ncols] -> LDS, 8 elements (16 loads) per lane in flight
  auto stage_gm = [&](float *dst, int r0, int nrows, int n0, int ncols) {
    const int total = nrows * ncols;
    for (int e0 = tid; e0 < total; e0 += 256 * 8) {
      float g[8], o[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + 256 * u;
        const int r = e / ncols, nn = e - r * ncols;
        const bool ok = e < total && r0 + r < p.R && n0 + nn < p.N;
        const size_t i = ok ? (size_t)(r0 + r) * p.N + n0 + nn : 0;
        g[u] = ok ? p.dout[i] : 0.f;
        o[u] = (ok && p.lrelu) ? p.out_in[i] : 1.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (e0 + 256 * u < total) dst[e0 + 256 * u] = (p.lrelu && o[u] - p.offset <= 0.f) ? 0.2f * g[u] : g[u];
    }
  };
  if (bx < p.nb_dw) {
    float *gs = dsm;             // [32][64]
    float *xt = dsm + 32 * 64;   // [32][16]
    const int tn = bx % ntn, tk = bx / ntn;
    const int n = tn * 64 + l, k0 = tk * 16 + q * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, accb = 0.f;
    for (int r0 = 0; r0 < p.R; r0 += 32) {
      __syncthreads();
      float xv[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = tid + 256 * u, r = e >> 4, kk = e & 15;
        xv[u] = (r0 + r < p.R && tk * 16 + kk < p.K) ? p.x[(size_t)(r0 + r) * p.ldx + tk * 16 + kk] : 0.f;
      }
      stage_gm(gs, r0, 32, tn * 64, 64);
      xt[tid] = xv[0]; xt[tid + 256] = xv[1];
      __syncthreads();
#pragma unroll 8
      for (int r = 0; r < 32; ++r) {
        const float g = gs[r * 64 + l];
        accb += g;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += xt[r * 16 + q * 4 + j] * g;
      }
    }
    if (n < p.N) {
      if (p.dw) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (k0 + j < p.K) p.dw[(size_t)(k0 + j) * p.N + n] = p.alpha * acc[j];
      }
      if (p.db && tk == 0 && q == 0) p.db[n] = p.beta * accb;
    }
    return;
  }
  const int bi = bx - p.nb_dw;
  const int ntk = (p.K + 15) / 16;
  const int tk = bi % ntk, tr = bi / ntk;
  const int r0 = tr * DN_RT;
  float *gs = dsm;  // [DN_RT][NP]   NP = min(N, DN_NMAX) rounded up to 64
  float acc[4][DN_RT];
#pragma unroll
  for (int kr = 0; kr < 4; ++kr)
#pragma unroll
    for (int r = 0; r < DN_RT; ++r) acc[kr][r] = 0.f;
  for (int n0 = 0; n0 < p.N; n0 += DN_NMAX) {
    const int nspan = min(DN_NMAX, ((p.N - n0) + 63) & ~63);
    float wv[4][DN_NMAX / 64];
#pragma unroll
    for (int kr = 0; kr < 4; ++kr) {
      const int k = tk * 16 + q * 4 + kr;
#pragma unroll
      for (int c = 0; c < DN_NMAX / 64; ++c) {
        const int n = n0 + c * 64 + l;
        wv[kr][c] = (k < p.K && c * 64 < nspan && n < p.N) ? p.w[(size_t)k * p.N + n] : 0.f;
      }
    }
    __syncthreads();
    stage_gm(gs, r0, DN_RT, n0, nspan);
    __syncthreads();
#pragma unroll
    for (int c = 0; c < DN_NMAX / 64; ++c) {
      if (c * 64 < nspan) {
#pragma unroll
        for (int r = 0; r < DN_RT; ++r) {
          const float g = gs[r * nspan + c * 64 + l];
#pragma unroll
          for (int kr = 0; kr < 4; ++kr) acc[kr][r] += g * wv[kr][c];
        }
      }
    }
  }
  {  // the 4 x DN_RT wave sums together (63 lane exchanges in six steps; 64 chains of six before): lane l ends up with value l = kr * 16 + r
    static_assert(DN_RT == 16, "the 64 partial sums of a wave map one to one onto its 64 lanes");
    float t[64];
#pragma unroll
    for (int kr = 0; kr < 4; ++kr)
#pragma unroll
      for (int r = 0; r < DN_RT; ++r) t[kr * DN_RT + r] = acc[kr][r];
    wave_tree_sum<64, 64, 32>(t, l);
    const int kr = l >> 4, r = l & 15;  // (wave_tree_row<64>(l) == l)
    const int k = tk * 16 + q * 4 + kr;
    if (k < p.K && r0 + r < p.R) p.dx[(size_t)(r0 + r) * p.ldx + k] = p.alpha * t[0];
  }
}

__global__ __launch_bounds__(256) void dense_fwd_kernel(const DenseP p) { dense_fwd_body(p, blockIdx.x, blockIdx.y); }
__global__ __launch_bounds__(256) void dense_bwd_kernel(const DenseP p) { dense_bwd_body(p, blockIdx.x); }

// several layers that share R, K and the scalars in ONE launch (grid.z = layer): the generator's 16 style affines
// s_l = style[:, l, :] @ W_l * coef + b_l + 1 (modulated_conv2d.py:52-56) read rows of one [R, L, K] tensor (ldx = L K) and
// their backward writes d(style) in place, slot by slot.
struct DenseMultiP { DenseP g[TBG_DENSE_MAX_ITEMS]; };
__global__ __launch_bounds__(256) void dense_multi_fwd_kernel(const DenseMultiP mp) {
  const DenseP &p = mp.g[blockIdx.z];
  if ((int)blockIdx.x * 64 >= p.N) return;
  dense_fwd_body(p, blockIdx.x, blockIdx.y);
}
__global__ __launch_bounds__(256) void dense_multi_bwd_kernel(const DenseMultiP mp, const int nb_dx_of) {
  const DenseP &p = mp.g[blockIdx.z];
  const int nb_dx = p.dx ? ((p.K + 15) / 16) * ((p.R + DN_RT - 1) / DN_RT) : 0;
  if ((int)blockIdx.x >= p.nb_dw + nb_dx) return;
  dense_bwd_body(p, blockIdx.x);
}

extern "C" int tbg_dense_fwd_f32(const float *x, const float *w, const float *b, float *out, int R, int K, int N,
                                 float alpha, float beta, int lrelu, float offset, void *stream) {
  if (!x || !w || !out || R < 1 || K < 1 || N < 1) return TBG_EINVAL;
  DenseP p = {};
  p.x = x; p.w = w; p.b = b; p.out = out; p.R = R; p.K = K; p.N = N; p.lrelu = lrelu; p.alpha = alpha; p.beta = beta;
  p.offset = offset; p.ldx = K;
  if (K > 768) return TBG_ERANGE;  // x rows are staged whole in LDS (<= 64 KB); large-K layers are GEMMs, not this kernel's job
  const size_t KP = (size_t)((K + 4 * DN_KT - 1) / (4 * DN_KT)) * (4 * DN_KT) + 4;
  const size_t lds = ((size_t)DN_RT * KP + 3 * DN_RT * 64) * sizeof(float);
  hipLaunchKernelGGL(dense_fwd_kernel, dim3((N + 63) / 64, (R + DN_RT - 1) / DN_RT), dim3(256), lds, tbg_stream(stream), p);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

extern "C" int tbg_dense_bwd_f32(const float *x, const float *w, const float *out, const float *dout, float *dx, float *dw,
                                 float *db, int R, int K, int N, float alpha, float beta, int lrelu, float offset,
                                 void *stream) {
  if (!x || !w || !dout || R < 1 || K < 1 || N < 1 || (lrelu && !out) || (!dx && !dw && !db)) return TBG_EINVAL;
  DenseP p = {};
  p.x = x; p.w = w; p.out_in = out; p.dout = dout; p.dx = dx; p.dw = dw; p.db = db; p.R = R; p.K = K; p.N = N;
  p.lrelu = lrelu; p.alpha = alpha; p.beta = beta; p.offset = offset; p.ldx = K;
  const int ntn = (N + 63) / 64;
  p.nb_dw = dw ? ntn * ((K + 15) / 16) : (db ? ntn : 0);  // db alone: the k-tile-0 row of blocks
  const int nb_dx = dx ? ((K + 15) / 16) * ((R + DN_RT - 1) / DN_RT) : 0;
  const int nsp = ((N < DN_NMAX ? N : DN_NMAX) + 63) & ~63;
  const size_t lds_dx = (size_t)DN_RT * nsp * sizeof(float), lds_dw = (size_t)(32 * 64 + 32 * 16) * sizeof(float);
  hipLaunchKernelGGL(dense_bwd_kernel, dim3(p.nb_dw + nb_dx), dim3(256), lds_dx > lds_dw ? lds_dx : lds_dw,
                     tbg_stream(stream), p);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

extern "C" int tbg_dense_multi_fwd_f32(const tbg_dense_item *items, int n, int R, int K, float alpha, float beta,
                                       float offset, void *stream) {
  if (!items || n < 1 || n > TBG_DENSE_MAX_ITEMS || R < 1 || K < 1) return TBG_EINVAL;
  if (K > 768) return TBG_ERANGE;
  DenseMultiP mp = {};
  int maxN = 0;
  for (int i = 0; i < n; ++i) {
    const tbg_dense_item &it = items[i];
    if (!it.x || !it.w || !it.out || it.N < 1 || it.ldx < K) return TBG_EINVAL;
    DenseP &p = mp.g[i];
    p.x = it.x; p.w = it.w; p.b = it.b; p.out = it.out; p.R = R; p.K = K; p.N = it.N; p.lrelu = 0; p.alpha = alpha;
    p.beta = beta; p.offset = offset; p.ldx = it.ldx;
    if (it.N > maxN) maxN = it.N;
  }
  const size_t KP = (size_t)((K + 4 * DN_KT - 1) / (4 * DN_KT)) * (4 * DN_KT) + 4;
  const size_t lds = ((size_t)DN_RT * KP + 3 * DN_RT * 64) * sizeof(float);
  hipLaunchKernelGGL(dense_multi_fwd_kernel, dim3((maxN + 63) / 64, (R + DN_RT - 1) / DN_RT, n), dim3(256), lds,
                     tbg_stream(stream), mp);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

extern "C" int tbg_dense_multi_bwd_f32(const tbg_dense_item *items, int n, int R, int K, float alpha, float beta,
                                       void *stream) {
  if (!items || n < 1 || n > TBG_DENSE_MAX_ITEMS || R < 1 || K < 1) return TBG_EINVAL;
  DenseMultiP mp = {};
  int maxB = 0, maxN = 0;
  for (int i = 0; i < n; ++i) {
    const tbg_dense_item &it = items[i];
    if (!it.x || !it.w || !it.dout || it.N < 1 || it.ldx < K || (!it.dx && !it.dw && !it.db)) return TBG_EINVAL;
    DenseP &p = mp.g[i];
    p.x = it.x; p.w = it.w; p.dout = it.dout; p.dx = it.dx; p.dw = it.dw; p.db = it.db; p.R = R; p.K = K; p.N = it.N;
    p.lrelu = 0; p.alpha = alpha; p.beta = beta; p.offset = 0.f; p.ldx = it.ldx;
    const int ntn = (it.N + 63) / 64;
    p.nb_dw = it.dw ? ntn * ((K + 15) / 16) : (it.db ? ntn : 0);
    const int nb = p.nb_dw + (it.dx ? ((K + 15) / 16) * ((R + DN_RT - 1) / DN_RT) : 0);
    if (nb > maxB) maxB = nb;
    if (it.N > maxN) maxN = it.N;
  }
  const int nsp = ((maxN < DN_NMAX ? maxN : DN_NMAX) + 63) & ~63;
  const size_t lds_dx = (size_t)DN_RT * nsp * sizeof(float), lds_dw = (size_t)(32 * 64 + 32 * 16) * sizeof(float);
  hipLaunchKernelGGL(dense_multi_bwd_kernel, dim3(maxB, 1, n), dim3(256), lds_dx > lds_dw ? lds_dx : lds_dw,
                     tbg_stream(stream), mp, 0);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}
