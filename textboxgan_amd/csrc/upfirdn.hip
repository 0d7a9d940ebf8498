// upfirdn2d for gfx950 -- written from the operator's DEFINITION (what the reference's TF op "UpFirDn2D" computes,
// upfirdn_2d.cu:310-324 / upfirdn_2d_v2.py:249-305), not from its kernels:
//
//   u = zero-insert(x, up)                                   u[Y*up] = x[Y], zeros elsewhere, length in*up
//   v[Y][X] = sum_{i<kH, j<kW} u[Y + i - pady0][X + j - padx0] * k[kH-1-i][kW-1-j]      (k applied flipped)
//   y[oy][ox] = v[oy*downy][ox*downx],   out = (in*up + pad0 + pad1 - k + down) / down
//
// Polyphase view used by both kernels: along one axis output o reads input samples i = first .. with
//   first = ceil((o*DN - pad0) / UP),  tap index j = first*UP - (o*DN - pad0) + t*UP  (t = 0, 1, ...; j < K).
//
// Two kernels:
//   * upfirdn2d_tile_kernel<UPX,UPY,DNX,DNY,RX,RY,SEP>: the model's forms (filters <= 4x4, factors <= 2, minor == 1).
//     NO LDS and no barriers: one lane owns a 4 (x) x 4 (y) patch of outputs; its input window rows arrive as 16-byte
//     global loads (a row is only 4-byte aligned: e.g. 257-float rows -- global_load_dwordx4 needs no more) that are all in
//     flight together -- plane edges included: the loads are branch-free, padding is zeroed by selects afterwards -- and
//     outputs leave as 16-byte stores.  Because a lane's first output is a multiple of 4 and the
//     up-factors are 1 or 2, which taps meet which window element depends only on (pad0 mod UP) -- template parameters
//     RX/RY -- so every register index is a compile-time constant and the taps sit in SGPRs.
//     SEP: k = ky (x) kx (the model's [1,3,3,1] (x) [1,3,3,1]): a horizontal pass over the window rows, then a vertical
//     pass -- 4+4 instead of 16 MACs per output for the blur.
//   * upfirdn2d_direct_kernel: everything else (any filter size, factors, minor): one lane per output, walks only the
//     taps that land on an input sample.
// Both fuse an optional per-plane input scale and the shared epilogue (demodulation scale, noise, bias, LeakyReLU).
#include <stdio.h>

#include <hip/hip_fp16.h>

#include "conv_common.h"  // (split3 / bf16 unit helpers of the unit-sink form; includes common.h)

struct UpfirdnP {
  const float *x, *k, *kx, *ky, *in_scale;  // k: 2-D [kH][kW] or NULL; kx/ky: 1-D factors or NULL
  float *y;
  int major, inH, inW, minor, kH, kW;
  int upx, upy, downx, downy, padx0, pady0;
  int outH, outW;
  int M, has_epi;
  int txw, tyn;  // tile kernel: lanes per output row / per output column
  long long total;
  EpiK e;
};

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

__host__ __device__ constexpr int uf_cdiv(int a, int b) { return a >= 0 ? (a + b - 1) / b : -((-a) / b); }
__host__ __device__ constexpr int uf_fdiv(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

// compile-time polyphase tables of one axis: NO consecutive outputs starting at a multiple of NO, pad0 = UP*q + R
template <int UP, int DN, int R, int NO>
struct UfAxis {
  static constexpr int first(int n) { return uf_cdiv(n * DN - R, UP); }
  static constexpr int base = uf_cdiv(-R, UP);
  static constexpr int rel(int n) { return first(n) - base; }               // window element of output n's first tap
  static constexpr int j0(int n) { return first(n) * UP - (n * DN - R); }   // its tap index (flipped order)
  static constexpr int TPP = (4 + UP - 1) / UP;                             // taps per phase of a <= 4-tap filter
  static constexpr int NI = rel(NO - 1) + TPP;                              // window elements
};

template <int UPX, int UPY, int DNX, int DNY, int RX, int RY, bool SEP>
__global__ __launch_bounds__(256) void upfirdn2d_tile_kernel(const UpfirdnP p) {
  constexpr int NOX = 4, NOY = 4;
  typedef UfAxis<UPX, DNX, RX, NOX> AX;
  typedef UfAxis<UPY, DNY, RY, NOY> AY;
  constexpr int NIX = AX::NI, NIY = AY::NI, NV = (NIX + 3) / 4, NIXP = 4 * NV;
  static_assert((NOX * DNX) % UPX == 0 && (NOY * DNY) % UPY == 0, "a lane's first output sits on phase 0");

  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  if (gid >= p.total) return;
  const int tx = (int)(gid % p.txw);
  const long long r0 = gid / p.txw;
  const int ty = (int)(r0 % p.tyn);
  const int plane = (int)(r0 / p.tyn);
  const int ox0 = tx * NOX, oy0 = ty * NOY;

  // flipped taps, zero padded to 4 (uniform addresses -> scalar loads)
  float kfx[4], kfy[4], k2[4][4];
  if constexpr (SEP) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      kfx[j] = j < p.kW ? p.kx[p.kW - 1 - j] : 0.f;
      kfy[j] = j < p.kH ? p.ky[p.kH - 1 - j] : 0.f;
    }
  } else {
#pragma unroll
    for (int jy = 0; jy < 4; ++jy)
#pragma unroll
      for (int jx = 0; jx < 4; ++jx)
        k2[jy][jx] = (jy < p.kH && jx < p.kW) ? p.k[(p.kH - 1 - jy) * p.kW + (p.kW - 1 - jx)] : 0.f;
  }

  const int qx = uf_fdiv(p.padx0, UPX), qy = uf_fdiv(p.pady0, UPY);  // pad0 = UP*q + R
  const int ix0 = (ox0 * DNX) / UPX + AX::base - qx;
  const int iy0 = (oy0 * DNY) / UPY + AY::base - qy;
  const float *xin = p.x + (size_t)plane * p.inH * p.inW;
  // Window loads are branch-free for every lane whose window lies inside the TENSOR (not merely inside its plane row): the
  // 16-byte loads are issued at the true addresses -- a window that hangs over the left / right end of a row reads the
  // neighbouring row's elements, one that hangs over the top / bottom of the plane reads a clamped row -- and the elements
  // that are padding are zeroed afterwards with selects (plane edges are NOT a slow path: the first version sent every
  // wave of a 256-wide plane through a second, divergent load sequence for its two edge lanes).  Only the few lanes
  // whose window would leave the allocation itself (first / last elements of the whole tensor) load element by element.
  const long long plane_off = (long long)plane * p.inH * p.inW;
  const long long n_total = (long long)p.major * p.inH * p.inW;
  const int iy_lo = min(max(iy0, 0), p.inH - 1), iy_hi = min(max(iy0 + NIY - 1, 0), p.inH - 1);
  const bool safe = plane_off + (long long)iy_lo * p.inW + ix0 >= 0 &&
                    plane_off + (long long)iy_hi * p.inW + ix0 + NIXP <= n_total;
  float w[NIY][NIXP];
  if (safe) {
#pragma unroll
    for (int m = 0; m < NIY; ++m) {
      const int iyc = min(max(iy0 + m, 0), p.inH - 1);
      const float *src = xin + (long long)iyc * p.inW + ix0;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const f32x4u t = *reinterpret_cast<const f32x4u *>(src + 4 * v);
        w[m][4 * v] = t.x; w[m][4 * v + 1] = t.y; w[m][4 * v + 2] = t.z; w[m][4 * v + 3] = t.w;
      }
    }
  } else {
#pragma unroll
    for (int m = 0; m < NIY; ++m) {
      const int iyc = min(max(iy0 + m, 0), p.inH - 1);
      const float *row = xin + (size_t)iyc * p.inW;
#pragma unroll
      for (int e = 0; e < NIXP; ++e) {
        const int ix = ix0 + e;
        w[m][e] = row[min(max(ix, 0), p.inW - 1)];
      }
    }
  }
  bool col_ok[NIXP];
#pragma unroll
  for (int e = 0; e < NIXP; ++e) col_ok[e] = e < NIX && ix0 + e >= 0 && ix0 + e < p.inW;
#pragma unroll
  for (int m = 0; m < NIY; ++m) {
    const bool row_ok = iy0 + m >= 0 && iy0 + m < p.inH;
#pragma unroll
    for (int e = 0; e < NIXP; ++e) w[m][e] = (row_ok && col_ok[e]) ? w[m][e] : 0.f;
  }

  float out[NOY][NOX];
  if constexpr (SEP) {
    float h[NIY][NOX];
#pragma unroll
    for (int m = 0; m < NIY; ++m)
#pragma unroll
      for (int n = 0; n < NOX; ++n) {
        float a = 0.f;
#pragma unroll
        for (int t = 0; t < AX::TPP; ++t)
          if (AX::j0(n) + t * UPX < 4) a += w[m][AX::rel(n) + t] * kfx[AX::j0(n) + t * UPX];
        h[m][n] = a;
      }
#pragma unroll
    for (int ny = 0; ny < NOY; ++ny)
#pragma unroll
      for (int n = 0; n < NOX; ++n) {
        float a = 0.f;
#pragma unroll
        for (int t = 0; t < AY::TPP; ++t)
          if (AY::j0(ny) + t * UPY < 4) a += h[AY::rel(ny) + t][n] * kfy[AY::j0(ny) + t * UPY];
        out[ny][n] = a;
      }
  } else {
#pragma unroll
    for (int ny = 0; ny < NOY; ++ny)
#pragma unroll
      for (int n = 0; n < NOX; ++n) {
        float a = 0.f;
#pragma unroll
        for (int t = 0; t < AY::TPP; ++t)
#pragma unroll
          for (int s = 0; s < AX::TPP; ++s)
            if (AY::j0(ny) + t * UPY < 4 && AX::j0(n) + s * UPX < 4)
              a += w[AY::rel(ny) + t][AX::rel(n) + s] * k2[AY::j0(ny) + t * UPY][AX::j0(n) + s * UPX];
        out[ny][n] = a;
      }
  }

  // ---- per-plane terms, then 16-byte stores
  const float isc = p.in_scale ? p.in_scale[plane] : 1.f;
  float sc = isc, bias = 0.f, str = 0.f, slope = 1.f, gain = 1.f;
  bool lrelu = false;
  const float *nzp = nullptr;
  if (p.has_epi) {
    const int b = plane / p.M, m = plane - b * p.M;
    sc = isc * p.e.alpha * (p.e.out_scale ? p.e.out_scale[plane] : 1.f);
    bias = p.e.bias ? p.e.bias[m] * p.e.bias_mul : 0.f;
    if (p.e.noise) { str = p.e.strength[0]; nzp = p.e.noise + (size_t)b * p.outH * p.outW; }
    lrelu = p.e.act == TBG_ACT_LRELU;
    slope = p.e.slope; gain = p.e.gain;
  }
  const bool full = ox0 + NOX <= p.outW;
  float *yo = p.y + (size_t)plane * p.outH * p.outW;
#pragma unroll
  for (int ny = 0; ny < NOY; ++ny) {
    const int oy = oy0 + ny;
    if (oy >= p.outH) break;
    const size_t off = (size_t)oy * p.outW + ox0;
    float nz[NOX] = {0.f, 0.f, 0.f, 0.f};
    if (nzp) {
      if (full) {
        const f32x4u t = *reinterpret_cast<const f32x4u *>(nzp + off);
        nz[0] = t.x; nz[1] = t.y; nz[2] = t.z; nz[3] = t.w;
      } else {
#pragma unroll
        for (int n = 0; n < NOX; ++n)
          if (ox0 + n < p.outW) nz[n] = nzp[off + n];
      }
    }
    float res[NOX];
#pragma unroll
    for (int n = 0; n < NOX; ++n) {
      float v = out[ny][n] * sc + nz[n] * str + bias;
      if (p.has_epi) v = (lrelu ? (v > 0.f ? v : v * slope) : v) * gain;
      res[n] = v;
    }
    if (full) {
      f32x4u t; t.x = res[0]; t.y = res[1]; t.z = res[2]; t.w = res[3];
      *reinterpret_cast<f32x4u *>(yo + off) = t;
    } else {
#pragma unroll
      for (int n = 0; n < NOX; ++n)
        if (ox0 + n < p.outW) yo[off + n] = res[n];
    }
  }
}

// ---- unit-sink form (tbg.h "UNIT SINK"): the model's blur after the up-convolution (upfirdn_2d_v2.py:97-103: FIR, up = down = 1)
// with the full epilogue, whose result ALSO (or only) leaves as the unit tensor units(out * units_scale) that the next
// convolution DMAs its tiles from -- the fp32 -> unit pass (tbg_units_pack_f32: read 4 B + write 2 B x planes per element, one
// launch) disappears.  A unit needs 8 channels of one pixel in one lane while the FIR wants a lane to own a patch of ONE plane
// (its window rows are 16-byte loads, coalesced along x), so the block (256 lanes = 8 channels x 32 lanes; tile = 8 rows x 64
// columns of the output) computes per plane exactly as upfirdn2d_tile_kernel<1,1,1,1,0,0,true> does -- same window, same
// horizontal-then-vertical order, same epilogue expression -- writes the fp32 result (if wanted) straight from registers as
// 16-byte stores, parks result * units_scale in a 16 KB LDS tile, and after one barrier every lane gathers the 8 channels of two
// pixels and stores their units; border tiles also write the ring of zero units next to them.
struct FirUnitsP {
  const float *x, *kx, *ky, *in_scale;
  float *y;
  int B, C, inH, inW, Ht, Wt, kH, kW, padx0, pady0, tilesX, tilesY;
  EpiK e;
};

template <int NP>
__global__ __launch_bounds__(256) void fir_units_kernel(const FirUnitsP p) {
  constexpr int TR = 8, TC = 64, PITCH = TC + 1;
  __shared__ float tile[8][TR][PITCH];
  const int tid = threadIdx.x;
  const int C8 = (p.C + 7) >> 3;
  int bid = blockIdx.x;
  const int tx = bid % p.tilesX; bid /= p.tilesX;
  const int ty = bid % p.tilesY; bid /= p.tilesY;
  const int cu = bid % C8, b = bid / C8;
  {
    const int cc = tid >> 5, l = tid & 31;
    const int c = cu * 8 + cc;
    const int ly = 4 * (l >> 4), lx = 4 * (l & 15);
    const int Y0 = ty * TR + ly, X0 = tx * TC + lx;
    float res[4][4];
#pragma unroll
    for (int ny = 0; ny < 4; ++ny)
#pragma unroll
      for (int n = 0; n < 4; ++n) res[ny][n] = 0.f;
    float us = 0.f;
    if (c < p.C && Y0 < p.Ht && X0 < p.Wt) {
      float kfx[4], kfy[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        kfx[j] = j < p.kW ? p.kx[p.kW - 1 - j] : 0.f;
        kfy[j] = j < p.kH ? p.ky[p.kH - 1 - j] : 0.f;
      }
      const int iy0 = Y0 - p.pady0, ix0 = X0 - p.padx0;
      const int plane = b * p.C + c;
      const long long plane_off = (long long)plane * p.inH * p.inW;
      const long long n_total = (long long)p.B * p.C * p.inH * p.inW;
      const float *xin = p.x + plane_off;
      const int iy_lo = min(max(iy0, 0), p.inH - 1), iy_hi = min(max(iy0 + 6, 0), p.inH - 1);
      const bool safe = plane_off + (long long)iy_lo * p.inW + ix0 >= 0 && plane_off + (long long)iy_hi * p.inW + ix0 + 8 <= n_total;
      float w[7][8];
      if (safe) {
#pragma unroll
        for (int m = 0; m < 7; ++m) {
          const float *src = xin + (long long)min(max(iy0 + m, 0), p.inH - 1) * p.inW + ix0;
#pragma unroll
          for (int v = 0; v < 2; ++v) {
            const f32x4u t = *reinterpret_cast<const f32x4u *>(src + 4 * v);
            w[m][4 * v] = t.x; w[m][4 * v + 1] = t.y; w[m][4 * v + 2] = t.z; w[m][4 * v + 3] = t.w;
          }
        }
      } else {
#pragma unroll
        for (int m = 0; m < 7; ++m) {
          const float *row = xin + (size_t)min(max(iy0 + m, 0), p.inH - 1) * p.inW;
#pragma unroll
          for (int e = 0; e < 8; ++e) w[m][e] = row[min(max(ix0 + e, 0), p.inW - 1)];
        }
      }
#pragma unroll
      for (int m = 0; m < 7; ++m) {
        const bool row_ok = iy0 + m >= 0 && iy0 + m < p.inH;
#pragma unroll
        for (int e = 0; e < 8; ++e) w[m][e] = (row_ok && e < 7 && ix0 + e >= 0 && ix0 + e < p.inW) ? w[m][e] : 0.f;
      }
      float h[7][4];
#pragma unroll
      for (int m = 0; m < 7; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          float a = 0.f;
#pragma unroll
          for (int t = 0; t < 4; ++t) a += w[m][n + t] * kfx[t];
          h[m][n] = a;
        }
      // per-plane terms of the epilogue (upfirdn2d_tile_kernel's expression)
      const float isc = p.in_scale ? p.in_scale[plane] : 1.f;
      const float sc = isc * p.e.alpha * (p.e.out_scale ? p.e.out_scale[plane] : 1.f);
      const float bias = p.e.bias ? p.e.bias[c] * p.e.bias_mul : 0.f;
      const float str = p.e.noise ? p.e.strength[0] : 0.f;
      const float *nzp = p.e.noise ? p.e.noise + (size_t)b * p.Ht * p.Wt : nullptr;
      const bool lrelu = p.e.act == TBG_ACT_LRELU;
      const float slope = p.e.slope, gain = p.e.gain;
      us = p.e.units_scale ? p.e.units_scale[plane] : 1.f;
      const bool full = X0 + 4 <= p.Wt;
      float *yo = p.y ? p.y + (size_t)plane * p.Ht * p.Wt : nullptr;
#pragma unroll
      for (int ny = 0; ny < 4; ++ny) {
        const int oy = Y0 + ny;
        if (oy >= p.Ht) break;
        const size_t off = (size_t)oy * p.Wt + X0;
        float nz[4] = {0.f, 0.f, 0.f, 0.f};
        if (nzp) {
          if (full) {
            const f32x4u t = *reinterpret_cast<const f32x4u *>(nzp + off);
            nz[0] = t.x; nz[1] = t.y; nz[2] = t.z; nz[3] = t.w;
          } else {
#pragma unroll
            for (int n = 0; n < 4; ++n)
              if (X0 + n < p.Wt) nz[n] = nzp[off + n];
          }
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          float a = 0.f;
#pragma unroll
          for (int t = 0; t < 4; ++t) a += h[ny + t][n] * kfy[t];
          float v = a * sc + nz[n] * str + bias;
          v = (lrelu ? (v > 0.f ? v : v * slope) : v) * gain;
          res[ny][n] = X0 + n < p.Wt ? v : 0.f;
        }
        if (yo) {
          if (full) {
            f32x4u t; t.x = res[ny][0]; t.y = res[ny][1]; t.z = res[ny][2]; t.w = res[ny][3];
            *reinterpret_cast<f32x4u *>(yo + off) = t;
          } else {
#pragma unroll
            for (int n = 0; n < 4; ++n)
              if (X0 + n < p.Wt) yo[off + n] = res[ny][n];
          }
        }
      }
    }
#pragma unroll
    for (int ny = 0; ny < 4; ++ny)
#pragma unroll
      for (int n = 0; n < 4; ++n) tile[cc][ly + ny][lx + n] = res[ny][n] * us;
  }
  __syncthreads();
  const int Hp = p.Ht + 2, Wp = p.Wt + 2;
  bf16x8 *Ub = static_cast<bf16x8 *>(p.e.units_out) + ((size_t)b * C8 + cu) * Hp * Wp;
  const size_t plane_units = (size_t)(p.e.units_plane >> 4);
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int idx = tid + 256 * q;
    const int r = idx >> 6, col = idx & 63;
    const int Y = ty * TR + r, X = tx * TC + col;
    if (Y >= p.Ht || X >= p.Wt) continue;
    float v[8];
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) v[cc] = tile[cc][r][col];
    const size_t u = (size_t)(Y + 1) * Wp + X + 1;
    if constexpr (NP == 3) {
      bf16x8 hh, mm, ll;
      split3_bf16x8(v, hh, mm, ll);
      Ub[u] = hh; Ub[plane_units + u] = mm; Ub[2 * plane_units + u] = ll;
    } else {
      Ub[u] = pack_bf16x8(v);
    }
  }
  // ring positions next to this tile (padded tile = 10 x 66 positions; corners may be written by several blocks: all zeros)
  const bool edge = ty == 0 || tx == 0 || (ty + 1) * TR >= p.Ht || (tx + 1) * TC >= p.Wt;
  if (edge) {
    const u32x4v zero = {0u, 0u, 0u, 0u};
    for (int pos = tid; pos < (TR + 2) * (TC + 2); pos += 256) {
      const int pr = pos / (TC + 2), pc = pos - pr * (TC + 2);
      const int Y = ty * TR - 1 + pr, X = tx * TC - 1 + pc;
      if (Y > p.Ht || X > p.Wt) continue;
      if (!(Y == -1 || Y == p.Ht || X == -1 || X == p.Wt)) continue;
      const size_t u = (size_t)(Y + 1) * Wp + X + 1;
      for (int pl = 0; pl < NP; ++pl) *reinterpret_cast<u32x4v *>(Ub + pl * plane_units + u) = zero;
    }
  }
}

static int launch_fir_units(const UpfirdnP &q, int padx1, int pady1, float *y, hipStream_t st) {
  if (q.upx != 1 || q.upy != 1 || q.downx != 1 || q.downy != 1 || q.minor != 1 || q.kH > 4 || q.kW > 4 || !q.kx || !q.ky)
    return TBG_EUNSUPPORTED;  // the sink rides on the model's separable blur only
  if (q.M < 1 || q.major % q.M != 0) return TBG_EINVAL;
  FirUnitsP p{};
  p.x = q.x; p.kx = q.kx; p.ky = q.ky; p.in_scale = q.in_scale; p.y = y;
  p.B = q.major / q.M; p.C = q.M; p.inH = q.inH; p.inW = q.inW; p.Ht = q.outH; p.Wt = q.outW; p.kH = q.kH; p.kW = q.kW;
  p.padx0 = q.padx0; p.pady0 = q.pady0;
  p.tilesX = (p.Wt + 63) / 64; p.tilesY = (p.Ht + 7) / 8;
  p.e = q.e;
  if (const int rc = epi_sink_geometry(p.e, p.B, p.C, p.Ht, p.Wt)) return rc;
  const long long nblk = (long long)p.B * (p.C / 8) * p.tilesX * p.tilesY;
  if (nblk > 2147483647LL) return TBG_ERANGE;
  if (p.e.units_planes == 3) hipLaunchKernelGGL(fir_units_kernel<3>, dim3((unsigned)nblk), dim3(256), 0, st, p);
  else hipLaunchKernelGGL(fir_units_kernel<1>, dim3((unsigned)nblk), dim3(256), 0, st, p);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

// General form: one lane per output element; only taps that land on an input sample are visited.
__global__ __launch_bounds__(256) void upfirdn2d_direct_kernel(const UpfirdnP p) {
  const long long total = (long long)p.major * p.outH * p.outW * p.minor;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    long long r = idx;
    const int mi = (int)(r % p.minor); r /= p.minor;
    const int ox = (int)(r % p.outW); r /= p.outW;
    const int oy = (int)(r % p.outH);
    const int major = (int)(r / p.outH);
    const int By = oy * p.downy - p.pady0, Bx = ox * p.downx - p.padx0;  // up-sampled coordinate under tap 0
    int jy0 = (-By) % p.upy; if (jy0 < 0) jy0 += p.upy;                   // first tap on an input row
    int jx0 = (-Bx) % p.upx; if (jx0 < 0) jx0 += p.upx;
    float acc = 0.f;
    for (int jy = jy0; jy < p.kH; jy += p.upy) {
      const int iy = (By + jy) / p.upy;  // exact
      if (iy < 0 || iy >= p.inH) continue;
      const float *row = p.x + ((size_t)major * p.inH + iy) * p.inW * p.minor + mi;
      const float kyv = p.k ? 1.f : p.ky[p.kH - 1 - jy];
      const float *krow = p.k ? p.k + (size_t)(p.kH - 1 - jy) * p.kW : nullptr;
      for (int jx = jx0; jx < p.kW; jx += p.upx) {
        const int ix = (Bx + jx) / p.upx;
        if (ix < 0 || ix >= p.inW) continue;
        const float kv = krow ? krow[p.kW - 1 - jx] : kyv * p.kx[p.kW - 1 - jx];
        acc += row[(size_t)ix * p.minor] * kv;
      }
    }
    if (p.in_scale) acc *= p.in_scale[major];
    if (p.has_epi) {
      const int b = major / p.M, m = major - b * p.M;
      float pre = acc * p.e.alpha;
      if (p.e.out_scale) pre *= p.e.out_scale[major];
      if (p.e.noise) pre += p.e.noise[(size_t)b * p.outH * p.outW + (size_t)oy * p.outW + ox] * p.e.strength[0];
      if (p.e.bias) pre += p.e.bias[m] * p.e.bias_mul;
      acc = epi_act(p.e, pre);
    }
    p.y[idx] = acc;
  }
}

template <int UPX, int UPY, int DNX, int DNY, int RX, int RY>
static int launch_tile(UpfirdnP &p, hipStream_t st) {
  p.txw = (p.outW + 3) / 4;
  p.tyn = (p.outH + 3) / 4;
  p.total = (long long)p.major * p.txw * p.tyn;
  const long long blocks = (p.total + 255) / 256;
  if (blocks > 2147483647LL) return TBG_ERANGE;
  if (p.k)
    hipLaunchKernelGGL((upfirdn2d_tile_kernel<UPX, UPY, DNX, DNY, RX, RY, false>), dim3((unsigned)blocks), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL((upfirdn2d_tile_kernel<UPX, UPY, DNX, DNY, RX, RY, true>), dim3((unsigned)blocks), dim3(256), 0, st, p);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

static inline int pmod(int a, int b) { int r = a % b; return r < 0 ? r + b : r; }

static int upfirdn_dispatch(UpfirdnP &p, hipStream_t st) {
  if (p.minor == 1 && p.kW <= 4 && p.kH <= 4) {
    const int rx = pmod(p.padx0, p.upx), ry = pmod(p.pady0, p.upy);
#define TBG_UF(ux, uy, dx, dy, RX, RY)                                                               \
  if (p.upx == ux && p.upy == uy && p.downx == dx && p.downy == dy && rx == RX && ry == RY)          \
    return launch_tile<ux, uy, dx, dy, RX, RY>(p, st);
    TBG_UF(1, 1, 1, 1, 0, 0)
    TBG_UF(2, 2, 1, 1, 0, 0) TBG_UF(2, 2, 1, 1, 1, 0) TBG_UF(2, 2, 1, 1, 0, 1) TBG_UF(2, 2, 1, 1, 1, 1)
    TBG_UF(1, 1, 2, 2, 0, 0)
    TBG_UF(1, 1, 2, 1, 0, 0)
    TBG_UF(2, 1, 1, 1, 0, 0) TBG_UF(2, 1, 1, 1, 1, 0)
#undef TBG_UF
  }
  const long long total = (long long)p.major * p.outH * p.outW * p.minor;
  long long blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(upfirdn2d_direct_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

// argument checks mirror the op's OP_REQUIRES list (upfirdn_2d.cu:228-229,241-256,266)
static int upfirdn_fill(UpfirdnP &p, const float *x, float *y, int major, int inH, int inW, int minor, int kH, int kW,
                        int upx, int upy, int downx, int downy, int padx0, int padx1, int pady0, int pady1) {
  if (!x || !y) return TBG_EINVAL;
  if (major < 1 || inH < 1 || inW < 1 || minor < 1) return TBG_EINVAL;
  if (upx < 1 || upy < 1 || downx < 1 || downy < 1) return TBG_EINVAL;
  if (kW < 1 || kH < 1) return TBG_EINVAL;
  const long long outW = ((long long)inW * upx + padx0 + padx1 - kW + downx) / downx;
  const long long outH = ((long long)inH * upy + pady0 + pady1 - kH + downy) / downy;
  if (outW < 1 || outH < 1) return TBG_EINVAL;
  if ((double)major * inH * inW * minor > 2147483647.0 || (double)major * outH * outW * minor > 2147483647.0)
    return TBG_ERANGE;
  p.x = x; p.y = y; p.k = nullptr; p.kx = nullptr; p.ky = nullptr; p.in_scale = nullptr;
  p.major = major; p.inH = inH; p.inW = inW; p.minor = minor; p.kH = kH; p.kW = kW;
  p.upx = upx; p.upy = upy; p.downx = downx; p.downy = downy; p.padx0 = padx0; p.pady0 = pady0;
  p.outH = (int)outH; p.outW = (int)outW; p.M = 1; p.has_epi = 0; p.e = make_epi(nullptr);
  p.txw = p.tyn = 0; p.total = 0;
  return TBG_OK;
}

static int upfirdn_epi(UpfirdnP &p, const float *in_scale, int M, const tbg_epilogue *epi, bool sink_ok = false) {
  if (!epi_valid(epi) || (epi && (epi->residual || epi->dot_aux || epi->gate))) return TBG_EINVAL;
  if (epi_has_sink(epi) && !sink_ok) return TBG_EINVAL;
  if (epi && (M < 1 || p.major % M != 0)) return TBG_EINVAL;
  p.in_scale = in_scale;
  if (epi) { p.has_epi = 1; p.M = M; p.e = make_epi(epi); }
  return TBG_OK;
}

extern "C" int tbg_upfirdn2d_f32(const float *x, const float *k, float *y, int major, int inH, int inW,
                                 int minor, int kH, int kW, int upx, int upy, int downx, int downy,
                                 int padx0, int padx1, int pady0, int pady1, void *stream) {
  UpfirdnP p;
  if (!k) return TBG_EINVAL;
  int rc = upfirdn_fill(p, x, y, major, inH, inW, minor, kH, kW, upx, upy, downx, downy, padx0, padx1, pady0, pady1);
  if (rc != TBG_OK) return rc;
  p.k = k;
  return upfirdn_dispatch(p, tbg_stream(stream));
}

extern "C" int tbg_upfirdn2d_ex_f32(const float *x, const float *k, float *y, int major, int inH, int inW,
                                    int kH, int kW, int upx, int upy, int downx, int downy, int padx0,
                                    int padx1, int pady0, int pady1, const float *in_scale, int M,
                                    const tbg_epilogue *epi, void *stream) {
  UpfirdnP p;
  if (!k) return TBG_EINVAL;
  int rc = upfirdn_fill(p, x, y, major, inH, inW, 1, kH, kW, upx, upy, downx, downy, padx0, padx1, pady0, pady1);
  if (rc != TBG_OK) return rc;
  if ((rc = upfirdn_epi(p, in_scale, M, epi)) != TBG_OK) return rc;
  p.k = k;
  return upfirdn_dispatch(p, tbg_stream(stream));
}

extern "C" int tbg_upfirdn2d_sep_f32(const float *x, const float *kx, const float *ky, float *y, int major, int inH,
                                     int inW, int kH, int kW, int upx, int upy, int downx, int downy, int padx0,
                                     int padx1, int pady0, int pady1, const float *in_scale, int M,
                                     const tbg_epilogue *epi, void *stream) {
  UpfirdnP p;
  if (!kx || !ky) return TBG_EINVAL;
  const bool sink = epi_has_sink(epi);
  static float dummy_y;  // (upfirdn_fill checks the pointers for NULL only: a sink may be the only output)
  int rc = upfirdn_fill(p, x, (sink && !y) ? &dummy_y : y, major, inH, inW, 1, kH, kW, upx, upy, downx, downy, padx0, padx1, pady0,
                        pady1);
  if (rc != TBG_OK) return rc;
  if ((rc = upfirdn_epi(p, in_scale, M, epi, true)) != TBG_OK) return rc;
  p.kx = kx; p.ky = ky;
  if (sink) return launch_fir_units(p, padx1, pady1, y, tbg_stream(stream));
  return upfirdn_dispatch(p, tbg_stream(stream));
}

// ---- 16-bit form: the reference op is registered for `float` AND `half` (REGISTER_KERNEL_BUILDER ... TypeConstraint<Eigen::half>,
// upfirdn_2d.cu:323-324), both accumulating in fp32 (.cu:101,195: `float v = 0`).  x, k and y are IEEE half tensors, the
// accumulator and the filter product are fp32, the result is rounded to half once (round-to-nearest-even) -- the op's own
// numerics.  One lane per output element (the reference's "ref-like" form): nothing in the training step uses a 16-bit FIR
// (activations stay fp32 in HBM), so this entry completes the boundary rather than a hot path.
__global__ __launch_bounds__(256) void upfirdn2d_direct_f16_kernel(const __half *__restrict__ x, const __half *__restrict__ k,
                                                                   __half *__restrict__ y, const UpfirdnP p) {
  const long long total = (long long)p.major * p.outH * p.outW * p.minor;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    long long r = idx;
    const int mi = (int)(r % p.minor); r /= p.minor;
    const int ox = (int)(r % p.outW); r /= p.outW;
    const int oy = (int)(r % p.outH);
    const int major = (int)(r / p.outH);
    const int By = oy * p.downy - p.pady0, Bx = ox * p.downx - p.padx0;
    int jy0 = (-By) % p.upy; if (jy0 < 0) jy0 += p.upy;
    int jx0 = (-Bx) % p.upx; if (jx0 < 0) jx0 += p.upx;
    float acc = 0.f;
    for (int jy = jy0; jy < p.kH; jy += p.upy) {
      const int iy = (By + jy) / p.upy;
      if (iy < 0 || iy >= p.inH) continue;
      const __half *row = x + ((size_t)major * p.inH + iy) * p.inW * p.minor + mi;
      const __half *krow = k + (size_t)(p.kH - 1 - jy) * p.kW;
      for (int jx = jx0; jx < p.kW; jx += p.upx) {
        const int ix = (Bx + jx) / p.upx;
        if (ix < 0 || ix >= p.inW) continue;
        acc += __half2float(row[(size_t)ix * p.minor]) * __half2float(krow[p.kW - 1 - jx]);
      }
    }
    y[idx] = __float2half_rn(acc);
  }
}

extern "C" int tbg_upfirdn2d_f16(const void *x, const void *k, void *y, int major, int inH, int inW, int minor, int kH,
                                 int kW, int upx, int upy, int downx, int downy, int padx0, int padx1, int pady0, int pady1,
                                 void *stream) {
  UpfirdnP p;
  if (!k) return TBG_EINVAL;
  static const float dummy = 0.f;  // upfirdn_fill checks the pointers only for NULL
  int rc = upfirdn_fill(p, x ? &dummy : nullptr, y ? const_cast<float *>(&dummy) : nullptr, major, inH, inW, minor, kH, kW, upx,
                        upy, downx, downy, padx0, padx1, pady0, pady1);
  if (rc != TBG_OK) return rc;
  const long long total = (long long)p.major * p.outH * p.outW * p.minor;
  long long blocks = (total + 255) / 256;
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(upfirdn2d_direct_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, tbg_stream(stream),
                     static_cast<const __half *>(x), static_cast<const __half *>(k), static_cast<__half *>(y), p);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

// Kernel instantiation (rocprofv3 spelling) the three entries above select: a pure function of the geometry (the same
// conditions as upfirdn_dispatch).  sep: the separable entry (tbg_upfirdn2d_sep_f32).  Test / profiling aid, like
// tbg_conv2d_kernel_name.
extern "C" int tbg_upfirdn2d_kernel_name(int minor, int kH, int kW, int upx, int upy, int downx, int downy, int padx0,
                                         int pady0, int sep, char *buf, int n) {
  if (!buf || n < 1) return TBG_EINVAL;
  buf[0] = 0;
  if (minor < 1 || kH < 1 || kW < 1 || upx < 1 || upy < 1 || downx < 1 || downy < 1) return TBG_EINVAL;
  if (minor == 1 && kW <= 4 && kH <= 4) {
    const int rx = pmod(padx0, upx), ry = pmod(pady0, upy);
    static const int tab[][6] = {{1, 1, 1, 1, 0, 0}, {2, 2, 1, 1, 0, 0}, {2, 2, 1, 1, 1, 0}, {2, 2, 1, 1, 0, 1}, {2, 2, 1, 1, 1, 1},
                                 {1, 1, 2, 2, 0, 0}, {1, 1, 2, 1, 0, 0}, {2, 1, 1, 1, 0, 0}, {2, 1, 1, 1, 1, 0}};
    for (const auto &t : tab)
      if (upx == t[0] && upy == t[1] && downx == t[2] && downy == t[3] && rx == t[4] && ry == t[5]) {
        snprintf(buf, n, "upfirdn2d_tile_kernel<%d, %d, %d, %d, %d, %d, %s>", t[0], t[1], t[2], t[3], t[4], t[5],
                 sep ? "true" : "false");
        return TBG_OK;
      }
  }
  snprintf(buf, n, "upfirdn2d_direct_kernel");
  return TBG_OK;
}
