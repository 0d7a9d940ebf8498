// Stride-2 3x3 convolutions (conv_downsample_2d's strided convolution, upfirdn_2d_v2.py:106-113; the data gradient of the
// up-convolution, :65-103) from PHASE unit tensors: the input t[B][C][Hin][Win] of a 3x3 / stride-2 / pad-0 convolution with
// Ho x Wo outputs, stored de-interleaved by row / column parity
//     P[plane][b][c / 8][ph = 2 py + px][i][j][c % 8] = t[b][c][2 i + py][2 j + px]      bf16 (planes as tbg.h "unit tensors"),
//     i in [0, Ho], j in [0, Wo]  (Hq = Ho + 1 rows of Wq = Wo + 1 units per phase), zero where t has no element.
// Tap (kh, kw) of output (y, x) reads t[2y + kh][2x + kw] = P[kh & 1][kw & 1][y + (kh >> 1)][x + (kw >> 1)]: inside a phase plane
// the stride is gone -- a tap shift is a unit-stride address offset, exactly as in conv_units.hip -- so the halo tiles go
// HBM -> LDS by DMA with no staging pass (the NCHW stride-2 kernels stage a 5 x 65 halo per 64 outputs on the VALU, split it
// three ways in f32x3 and de-interleave it on the way: 0.20-0.27 of the roof).
#include <type_traits>

#include "conv_common.h"

static inline long long s2_units_per_plane(int B, int C, int Ho, int Wo) {
  return (long long)B * ((C + 7) / 8) * 4 * (Ho + 1) * (Wo + 1);
}

extern "C" long long tbg_units_s2_bytes(int B, int C, int Ho, int Wo, int planes) {
  if (B < 1 || C < 1 || Ho < 1 || Wo < 1 || (planes != 1 && planes != 3)) return TBG_EINVAL;
  return s2_units_per_plane(B, C, Ho, Wo) * planes * 16;
}

// ---- stand-alone producer: NCHW fp32 (x optional per-(b,c) scale) -> phase unit tensor.  One lane per INPUT position of the
// padded domain (2 Hq x 2 Wq, x fastest: the 8 channel loads are coalesced along x; the stores of a wave alternate between two
// phase planes, each a contiguous run of units).
template <int NP>
__global__ __launch_bounds__(256) void units_pack_s2_kernel(const float *__restrict__ x, const float *__restrict__ scale,
                                                            bf16x8 *__restrict__ U, int B, int C, int Hin, int Win, int Hq,
                                                            int Wq, long long plane) {
  const int C8 = (C + 7) >> 3, W2 = 2 * Wq, H2 = 2 * Hq;
  const long long HW = (long long)Hin * Win;
  const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
  if (n >= plane) return;
  const int X = (int)(n % W2);
  long long t = n / W2;
  const int Y = (int)(t % H2);
  t /= H2;
  const int cu = (int)(t % C8), b = (int)(t / C8);
  const bool inside = Y < Hin && X < Win;
  const long long g0 = ((long long)b * C + cu * 8) * HW + (long long)Y * Win + X;
  float v[8];
#pragma unroll
  for (int cc = 0; cc < 8; ++cc) {
    const bool ok = inside && cu * 8 + cc < C;
    float a = x[ok ? g0 + cc * HW : 0];
    if (scale) a *= scale[ok ? b * C + cu * 8 + cc : 0];
    v[cc] = ok ? a : 0.f;
  }
  const long long u = ((((long long)b * C8 + cu) * 4 + (Y & 1) * 2 + (X & 1)) * Hq + (Y >> 1)) * Wq + (X >> 1);
  if constexpr (NP == 3) {
    bf16x8 h, m, l;
    split3_bf16x8(v, h, m, l);
    U[u] = h; U[plane + u] = m; U[2 * plane + u] = l;
  } else {
    U[u] = pack_bf16x8(v);
  }
}

extern "C" int tbg_units_pack_s2_f32(const float *x, const float *scale, void *U, int B, int C, int Hin, int Win, int Ho, int Wo,
                                     int planes, void *stream) {
  if (!x || !U || B < 1 || C < 1 || Hin < 3 || Win < 3 || Ho < 1 || Wo < 1 || (planes != 1 && planes != 3)) return TBG_EINVAL;
  if ((reinterpret_cast<uintptr_t>(U) & 15) != 0) return TBG_EINVAL;
  if (Ho != (Hin - 3) / 2 + 1 || Wo != (Win - 3) / 2 + 1) return TBG_EINVAL;
  const long long plane = s2_units_per_plane(B, C, Ho, Wo);
  if (plane * 8 > 2147483647LL || (long long)B * C * Hin * Win > 2147483647LL) return TBG_ERANGE;
  const dim3 grid((unsigned)((plane + 255) / 256));
  if (planes == 3)
    hipLaunchKernelGGL(units_pack_s2_kernel<3>, grid, dim3(256), 0, tbg_stream(stream), x, scale, reinterpret_cast<bf16x8 *>(U), B, C,
                       Hin, Win, Ho + 1, Wo + 1, plane);
  else
    hipLaunchKernelGGL(units_pack_s2_kernel<1>, grid, dim3(256), 0, tbg_stream(stream), x, scale, reinterpret_cast<bf16x8 *>(U), B, C,
                       Hin, Win, Ho + 1, Wo + 1, plane);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

// ---- fused producer: the FIR pass in front of a stride-2 convolution (conv_downsample_2d's blur, upfirdn_2d_v2.py:106-113; the
// blur's adjoint in the backward pass of upsample_conv_2d, :204-209) writing its result t = upfirdn2d(x, k, pad) * in_scale as a
// PHASE unit tensor -- the fp32 NCHW tensor between the blur and the strided convolution / its filter gradient never exists.
// up = down = 1, separable filter of <= 4 x 4 taps (the model's [1,3,3,1] x [1,3,3,1]): same arithmetic as
// upfirdn2d_tile_kernel<1,1,1,1,..,SEP> (horizontal pass over the window rows, then the vertical pass, then the scale), so the
// result is tbg_units_pack_s2_f32 of that kernel's output up to single fp32 roundings (the compiler contracts a few multiply-adds
// differently in the two kernels; both sit at the same distance from float64).
// A unit needs 8 channels of one pixel in one lane, the FIR wants a lane to own a patch of ONE plane (its window rows are 16-byte
// loads, coalesced along x): the block (256 lanes = 8 channels x 32 lanes, tile = 8 rows x 64 columns of t) computes per plane --
// lane = 4 rows x 4 columns, a 7 x 8 window as fourteen 16-byte loads (the patch of upfirdn2d_tile_kernel), lanes whose patch lies
// outside t load nothing -- into a 16 KB LDS tile, and after one barrier every lane gathers the 8 channels of TWO pixels and stores
// their units (per plane): the stores of a half-wave are one 512-byte run of a phase plane row.
struct FirS2P {
  const float *x, *kx, *ky, *in_scale;
  bf16x8 *U;
  int B, C, inH, inW, Ht, Wt, kH, kW, padx0, pady0, Hq, Wq, tilesX, tilesY;
  long long plane;
};

typedef float f32x4u2 __attribute__((ext_vector_type(4), aligned(4)));

template <int NP>
__global__ __launch_bounds__(256) void fir_units_s2_kernel(const FirS2P p) {
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
    float out[4][4];
#pragma unroll
    for (int ny = 0; ny < 4; ++ny)
#pragma unroll
      for (int n = 0; n < 4; ++n) out[ny][n] = 0.f;
    if (c < p.C && Y0 < p.Ht && X0 < p.Wt) {
      float kfx[4], kfy[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        kfx[j] = j < p.kW ? p.kx[p.kW - 1 - j] : 0.f;
        kfy[j] = j < p.kH ? p.ky[p.kH - 1 - j] : 0.f;
      }
      const int iy0 = Y0 - p.pady0, ix0 = X0 - p.padx0;
      const long long plane_off = ((long long)b * p.C + c) * p.inH * p.inW;
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
            const f32x4u2 t = *reinterpret_cast<const f32x4u2 *>(src + 4 * v);
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
      const float isc = p.in_scale ? p.in_scale[b * p.C + c] : 1.f;
#pragma unroll
      for (int ny = 0; ny < 4; ++ny)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          float a = 0.f;
#pragma unroll
          for (int t = 0; t < 4; ++t) a += h[ny + t][n] * kfy[t];
          out[ny][n] = (Y0 + ny < p.Ht && X0 + n < p.Wt) ? a * isc : 0.f;
        }
    }
#pragma unroll
    for (int ny = 0; ny < 4; ++ny)
#pragma unroll
      for (int n = 0; n < 4; ++n) tile[cc][ly + ny][lx + n] = out[ny][n];
  }
  __syncthreads();
  // the two pixels of this lane: idx = tid, tid + 256 of [phase 4][row 4][column 32] of the tile's phase planes
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int idx = tid + 256 * q;
    const int ph = idx >> 7, il = (idx >> 5) & 3, jl = idx & 31;
    const int i = ty * (TR / 2) + il, j = tx * (TC / 2) + jl;
    if (i >= p.Hq || j >= p.Wq) continue;
    float v[8];
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) v[cc] = tile[cc][2 * il + (ph >> 1)][2 * jl + (ph & 1)];
    const long long u = ((((long long)b * C8 + cu) * 4 + ph) * p.Hq + i) * p.Wq + j;
    if constexpr (NP == 3) {
      bf16x8 hh, mm, ll;
      split3_bf16x8(v, hh, mm, ll);
      p.U[u] = hh; p.U[p.plane + u] = mm; p.U[2 * p.plane + u] = ll;
    } else {
      p.U[u] = pack_bf16x8(v);
    }
  }
}

extern "C" int tbg_upfirdn2d_units_s2_f32(const float *x, const float *kx, const float *ky, void *U, int B, int C, int inH, int inW,
                                          int kH, int kW, int padx0, int padx1, int pady0, int pady1, const float *in_scale,
                                          int planes, void *stream) {
  if (!x || !kx || !ky || !U || B < 1 || C < 1 || inH < 1 || inW < 1 || kH < 1 || kW < 1 || kH > 4 || kW > 4 ||
      (planes != 1 && planes != 3))
    return TBG_EINVAL;
  if ((reinterpret_cast<uintptr_t>(U) & 15) != 0) return TBG_EINVAL;
  const int Ht = inH + pady0 + pady1 - kH + 1, Wt = inW + padx0 + padx1 - kW + 1;
  if (Ht < 3 || Wt < 3) return TBG_EINVAL;
  const int Ho = (Ht - 3) / 2 + 1, Wo = (Wt - 3) / 2 + 1;
  const long long plane = s2_units_per_plane(B, C, Ho, Wo);
  if (plane * 8 > 2147483647LL || (long long)B * C * inH * inW > 2147483647LL) return TBG_ERANGE;
  FirS2P p{};
  p.x = x; p.kx = kx; p.ky = ky; p.in_scale = in_scale; p.U = reinterpret_cast<bf16x8 *>(U);
  p.B = B; p.C = C; p.inH = inH; p.inW = inW; p.Ht = Ht; p.Wt = Wt; p.kH = kH; p.kW = kW; p.padx0 = padx0; p.pady0 = pady0;
  p.Hq = Ho + 1; p.Wq = Wo + 1; p.tilesX = (2 * p.Wq + 63) / 64; p.tilesY = (2 * p.Hq + 7) / 8;
  p.plane = plane;
  const long long nblk = (long long)B * ((C + 7) / 8) * p.tilesX * p.tilesY;
  if (nblk > 2147483647LL) return TBG_ERANGE;
  const dim3 grid((unsigned)nblk);
  if (planes == 3) hipLaunchKernelGGL(fir_units_s2_kernel<3>, grid, dim3(256), 0, tbg_stream(stream), p);
  else hipLaunchKernelGGL(fir_units_s2_kernel<1>, grid, dim3(256), 0, tbg_stream(stream), p);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

// ============================================================================================
// forward convolution (3x3, stride 2, pad 0) from a phase unit tensor
// ============================================================================================
// The structure of conv_units_fprop_kernel (conv_units.hip): implicit GEMM (M = output channels, N = 8 x 32 output pixels, K =
// taps x channels), packed filter, f32x3 term pairing, fused epilogue, ONE 512-thread block per CU, every operand byte by
// LDS-DMA, two LDS buffers.  What differs: a channel chunk is contracted in TWO STAGES, each with its own buffer --
//     stage 0: phases (0,0) + (1,1), taps (0,0) (0,2) (2,0) (2,2) | (1,1)      5 taps, 2 phase tiles of 9 x 34 positions
//     stage 1: phases (0,1) + (1,0), taps (0,1) (2,1) | (1,0) (1,2)            4 taps, 2 phase tiles
// (all four phase tiles and nine filter taps of a chunk at once are 114 KB in f32x3: no double buffer; split this way the two
// buffers are 60 + 54 KB and the DMA of one stage runs under the MFMAs of the other: 16 B per matrix cycle and CU against 10 of
// the stride-1 kernel -- the phase tiles are 3.6x the bytes of a stride-1 halo for the same MFMA work).
struct ConvS2P {
  const char *XU, *Wf;
  long long x_plane, w_plane;  // 16-byte units per plane
  float *y;
  int B, C8, M, Ho, Wo, Hq, Wq, ldw, flip;
  int tilesU, tilesV, dot_slots;
  EpiK e;
};

// stage tables: tap t = 3 kh + kw of a stage's tap slot, its phase slot in the stage's buffer, and its (row, column) offset
// inside the phase tile  (stage 0: t = 0 2 6 8 | 4;  stage 1: t = 1 7 | 3 5)
constexpr int s2_tap(int st, int ts) { return st == 0 ? (ts == 0 ? 0 : ts == 1 ? 2 : ts == 2 ? 6 : ts == 3 ? 8 : 4) : (ts == 0 ? 1 : ts == 1 ? 7 : ts == 2 ? 3 : 5); }
constexpr int s2_slot(int st, int ts) { return st == 0 ? (ts == 4 ? 1 : 0) : (ts >= 2 ? 1 : 0); }
constexpr int s2_roff(int st, int ts) { return st == 0 ? ((ts == 2 || ts == 3) ? 1 : 0) : (ts == 1 ? 1 : 0); }
constexpr int s2_coff(int st, int ts) { return st == 0 ? ((ts == 1 || ts == 3) ? 1 : 0) : (ts == 3 ? 1 : 0); }

template <int NP, int WTM>
struct S2Cfg {
  static constexpr int BM = 2 * WTM * 32, CKU = NP == 3 ? 1 : 2;
  static constexpr int XS = 9 * 34;                       // positions of one phase tile (9 rows, pitch 34)
  static constexpr int X_UNITS = NP * CKU * 2 * XS;       // Xs[plane][unit][slot][XS]
  static constexpr int NT0 = 5, NT1 = 4;
  static constexpr int A0 = NP * NT0 * CKU * BM, A1 = NP * NT1 * CKU * BM;  // As[plane][tap slot][unit][BM]
  static constexpr int NPIECE0 = (A0 + X_UNITS + 63) / 64, NPIECE1 = (A1 + X_UNITS + 63) / 64;
  static constexpr int BUF0 = NPIECE0 * 64, BUF1 = NPIECE1 * 64;
  static constexpr int PPW0 = (NPIECE0 + 7) / 8, PPW1 = (NPIECE1 + 7) / 8;
  static_assert(A0 % 64 == 0 && A1 % 64 == 0, "a DMA piece never straddles the filter / phase-tile regions");
};

template <int NP, int WTM>
__global__ __launch_bounds__(512, 2) void conv_units_s2_fprop_kernel(const ConvS2P p) {
  using Cf = S2Cfg<NP, WTM>;
  constexpr int WTN = 2, BM = Cf::BM, CKU = Cf::CKU, XS = Cf::XS, X_UNITS = Cf::X_UNITS;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [BUF0 | BUF1] units
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3, half = lane >> 5;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;

  // block -> (pixel tile, channel tile): one contiguous eighth of the list per XCD (conv_units_fprop_kernel: tiles that share halo
  // units and the channel tiles of one pixel tile meet in one L2 instead of eight)
  const int mtiles = p.M / BM, total = gridDim.x;
  int lin = blockIdx.x;
  if ((total & 7) == 0) lin = (blockIdx.x & 7) * (total >> 3) + (blockIdx.x >> 3);
  const int tn = lin / mtiles, mt = lin - tn * mtiles;
  const int tv = tn % p.tilesV, t2 = tn / p.tilesV;
  const int tu = t2 % p.tilesU, b = t2 / p.tilesU;
  const int m0 = mt * BM, y0 = tu * 8, x0 = tv * 32;

  // ---- DMA descriptors (registers, as conv_units_fprop_kernel): per-lane source address at chunk 0 of every piece this wave
  // issues in stage 0 / stage 1 (piece q = wave + 8 k)
  const char *dsrc0[Cf::PPW0], *dsrc1[Cf::PPW1];
  const char *const xu = p.XU, *const wf = p.Wf;
  auto describe = [&](int st, int q) -> const char * {
    const int NT = st == 0 ? Cf::NT0 : Cf::NT1, A_UNITS = st == 0 ? Cf::A0 : Cf::A1;
    const int n = q * 64 + lane;
    if (n < A_UNITS) {
      const int row = n / BM, m = n - row * BM;
      const int u = row % CKU, ts = (row / CKU) % NT, pl = row / (NT * CKU);
      const int t = s2_tap(st, ts);
      const int tw = p.flip ? 8 - t : t;
      return wf + ((pl * p.w_plane + (long long)(tw * p.C8 + u) * p.ldw + m0 + m) << 4);
    }
    const int m2 = min(n - A_UNITS, X_UNITS - 1);
    const int row = m2 / XS, pos = m2 - row * XS;
    const int slot = row & 1, u = (row >> 1) % CKU, pl = (row >> 1) / CKU;
    const int ph = st == 0 ? (slot == 0 ? 0 : 3) : (slot == 0 ? 1 : 2);
    const int r = pos / 34, c = min(pos - r * 34, 32);  // (pitch slot 33 is never read: it re-fetches column 32)
    return xu + ((pl * p.x_plane + ((((long long)(b * p.C8 + u)) * 4 + ph) * p.Hq + y0 + r) * p.Wq + x0 + c) << 4);
  };
#pragma unroll
  for (int k = 0; k < Cf::PPW0; ++k) dsrc0[k] = describe(0, min(wave + 8 * k, Cf::NPIECE0 - 1));
#pragma unroll
  for (int k = 0; k < Cf::PPW1; ++k) dsrc1[k] = describe(1, min(wave + 8 * k, Cf::NPIECE1 - 1));
  const long long a_step = (long long)CKU * p.ldw * 16, x_step = (long long)CKU * 4 * p.Hq * p.Wq * 16;  // bytes per chunk
  auto issue0 = [&](int k, int kc) {
    const int q = min(__builtin_amdgcn_readfirstlane(wave) + 8 * k, Cf::NPIECE0 - 1);
    dma16(dsrc0[k] + kc * (q * 64 < Cf::A0 ? a_step : x_step), lds0 + (unsigned)(q * 64 * 16));
  };
  auto issue1 = [&](int k, int kc) {
    const int q = min(__builtin_amdgcn_readfirstlane(wave) + 8 * k, Cf::NPIECE1 - 1);
    dma16(dsrc1[k] + kc * (q * 64 < Cf::A1 ? a_step : x_step), lds0 + (unsigned)((Cf::BUF0 + q * 64) * 16));
  };

  // ---- operand addressing (bytes inside a stage's buffer)
  const int a_lane = (wm * (WTM * 32) + (lane & 31)) * 16;
  int b_lane[WTN];
#pragma unroll
  for (int j = 0; j < WTN; ++j) b_lane[j] = ((wn * WTN + j) * 34 + (lane & 31)) * 16;
  constexpr int X_PL = CKU * 2 * XS * 16;
  const int bZ = 2 * (1 - half) * X_PL, bH = NP == 3 ? 0 : half * 2 * XS * 16;

  f32x16 acc[WTM][WTN];
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // one stage: NT taps of MFMAs from this stage's buffer; behind them the operand reads of the next tap and the DMA pieces of
  // the NEXT stage (chunk kc_next) into the other buffer
  auto run_stage = [&](auto stc, int kc_next) {
    constexpr int ST = decltype(stc)::value;
    constexpr int NT = ST == 0 ? Cf::NT0 : Cf::NT1, A_UNITS = ST == 0 ? Cf::A0 : Cf::A1;
    constexpr int PPW_NEXT = ST == 0 ? Cf::PPW1 : Cf::PPW0;
    constexpr int A_PL = NT * CKU * BM * 16;
    const int aX = NP == 3 ? half * A_PL : half * BM * 16, aY = 2 * half * A_PL;
    const char *Ab = smem + (size_t)(ST == 0 ? 0 : Cf::BUF0) * 16 + a_lane;
    const char *Xb = smem + (size_t)((ST == 0 ? 0 : Cf::BUF0) + A_UNITS) * 16;
    constexpr int NA = NP == 3 ? 2 * WTM : WTM, NB = NP == 3 ? 3 * WTN : WTN, NL = NA + NB;
    constexpr int NM = (NP == 3 ? 3 : 1) * WTM * WTN;  // MFMAs per tap
    bf16x8 av[2][NA], bw[2][NB];
    auto ld1 = [&](int ts, int bs, int idx) {
      if (idx < NA) {
        const int i = NP == 3 ? idx >> 1 : idx;
        const int plane_off = NP == 3 ? ((idx & 1) ? aY : aX) : aX;
        av[bs][idx] = *reinterpret_cast<const bf16x8 *>(Ab + ts * CKU * BM * 16 + plane_off + i * 32 * 16);
      } else if (idx < NL) {
        const int e = idx - NA;
        const int j = NP == 3 ? e / 3 : e, w = NP == 3 ? e - 3 * j : 0;
        const int plane_off = NP == 3 ? (w == 0 ? 0 : w == 1 ? X_PL : bZ) : bH;
        bw[bs][e] = *reinterpret_cast<const bf16x8 *>(Xb + b_lane[j] + (s2_slot(ST, ts) * XS + s2_roff(ST, ts) * 34 + s2_coff(ST, ts)) * 16 +
                                                      plane_off);
      }
    };
#pragma unroll
    for (int idx = 0; idx < NL; ++idx) ld1(0, 0, idx);
#pragma unroll
    for (int ts = 0; ts < NT; ++ts) {
      const int bs = ts & 1;
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mm = 0; mm < NM; ++mm) {
        const int grp = mm / (WTM * WTN), ij = mm - grp * (WTM * WTN);
        const int i = ij / WTN, j = ij - i * WTN;
        if constexpr (NP == 3) {  // smallest terms first: (hi|lo)x(lo|hi), then (hi|mid) x mid, then (hi|mid) x hi
          const int ai = 2 * i + (grp == 0 ? 1 : 0), bi = 3 * j + (grp == 0 ? 2 : grp == 1 ? 1 : 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[bs][ai], bw[bs][bi], acc[i][j], 0, 0, 0);
        } else {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[bs][i], bw[bs][j], acc[i][j], 0, 0, 0);
        }
        if (ts + 1 < NT) {
          constexpr int LPM = (NL + NM - 1) / NM;
#pragma unroll
          for (int e = 0; e < LPM; ++e) ld1(ts + 1, bs ^ 1, mm * LPM + e);
        }
        {
          constexpr int STRIDE = (NT * NM / 2) / PPW_NEXT > 0 ? (NT * NM / 2) / PPW_NEXT : 1;
          static_assert(NT * NM >= PPW_NEXT, "every piece of the next stage has an MFMA to hide behind");
          const int slot = ts * NM + mm;
          if (slot % STRIDE == 0 && slot / STRIDE < PPW_NEXT) {
            if constexpr (ST == 0) issue1(slot / STRIDE, kc_next); else issue0(slot / STRIDE, kc_next);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next stage's tiles have landed
    __syncthreads();                                   // ... for every wave, and every wave is done with this buffer
  };

  const int nchunks = p.C8 / CKU;
#pragma unroll
  for (int k = 0; k < Cf::PPW0; ++k) issue0(k, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kc = 0; kc < nchunks; ++kc) {
    run_stage(std::integral_constant<int, 0>{}, kc);
    run_stage(std::integral_constant<int, 1>{}, kc + 1 < nchunks ? kc + 1 : kc);  // (the last chunk re-fetches itself: no branch)
  }

  int e_pix[WTN], e_b[WTN];
#pragma unroll
  for (int j = 0; j < WTN; ++j) {
    e_pix[j] = (y0 + wn * WTN + j) * p.Wo + x0 + (lane & 31);
    e_b[j] = b;
  }
  conv_epilogue<WTM, WTN, 4, true, 16, 3, true>(acc, p.e, p.y, nullptr, p.M, p.Ho * p.Wo, m0 + wm * WTM * 32, lane, e_pix, e_b, true, b, p.dot_slots,
                             (tu * p.tilesV + tv) * 4 + wn, p.Ho, p.Wo);
}

static bool conv_units_s2_ok(const tbg_conv_desc *d, int planes) {
  const int cku = planes == 3 ? 8 : 16;
  return !d->transposed && d->KH == 3 && d->KW == 3 && d->sy == 2 && d->sx == 2 && d->py == 0 && d->px == 0 && d->Hin >= 3 &&
         d->Win >= 3 && d->Hout == (d->Hin - 3) / 2 + 1 && d->Wout == (d->Win - 3) / 2 + 1 && (d->Hout % 8) == 0 &&
         (d->Wout % 32) == 0 && (d->C % cku) == 0 && (d->M % 64) == 0 && d->ksplit == 1 && d->ldw >= d->M;
}

#ifndef UNITS_FULL
#define UNITS_FULL 200
#endif
static int units_s2_wtm(const tbg_conv_desc *d) {  // (the rule of conv_units.hip units_wtm)
  if (d->M % 128 != 0) return 1;
  const long long b2 = (long long)d->B * (d->Hout / 8) * (d->Wout / 32) * (d->M / 128);
  return (b2 < UNITS_FULL && 2 * b2 >= UNITS_FULL) ? 1 : 2;
}

extern "C" int tbg_conv2d_units_s2_blocks(const tbg_conv_desc *d, int planes) {
  if (!d || (planes != 1 && planes != 3)) return TBG_EINVAL;
  if (!conv_units_s2_ok(d, planes)) return TBG_EUNSUPPORTED;
  const long long n = (long long)d->B * (d->Hout / 8) * (d->Wout / 32) * (d->M / (64 * units_s2_wtm(d)));
  return n > 2147483647LL ? TBG_ERANGE : (int)n;
}

extern "C" int tbg_conv2d_units_s2_tile_channels(const tbg_conv_desc *d, int planes) {
  if (!d || (planes != 1 && planes != 3)) return TBG_EINVAL;
  if (!conv_units_s2_ok(d, planes)) return TBG_EUNSUPPORTED;
  return 64 * units_s2_wtm(d);
}

extern "C" int tbg_conv2d_units_s2_dot_slots(const tbg_conv_desc *d, int planes) {
  if (!d || (planes != 1 && planes != 3)) return TBG_EINVAL;
  if (!conv_units_s2_ok(d, planes)) return TBG_EUNSUPPORTED;
  return (d->Hout / 8) * (d->Wout / 32) * 4;
}

template <int NP, int WTM>
static int launch_conv_units_s2(ConvS2P &p, hipStream_t st) {
  using Cf = S2Cfg<NP, WTM>;
  const size_t lds = (size_t)(Cf::BUF0 + Cf::BUF1) * 16;
  auto kern = conv_units_s2_fprop_kernel<NP, WTM>;
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return TBG_EHIP;
  hipLaunchKernelGGL(kern, dim3(p.B * p.tilesU * p.tilesV * (p.M / Cf::BM)), dim3(512), lds, st, p);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

extern "C" int tbg_conv2d_units_s2(const tbg_conv_desc *d, const void *XU, int planes, const void *w, float *y,
                                   const tbg_epilogue *epi, void *stream) {
  if (!d || !XU || !w || (!y && !epi_has_sink(epi)) || (planes != 1 && planes != 3) || !epi_valid(epi)) return TBG_EINVAL;
  if (d->B < 1 || d->C < 1 || d->M < 1 || d->Hin < 1 || d->Win < 1) return TBG_EINVAL;
  if (((reinterpret_cast<uintptr_t>(XU) | reinterpret_cast<uintptr_t>(w)) & 15) != 0) return TBG_EINVAL;
  if (!conv_units_s2_ok(d, planes)) return TBG_EUNSUPPORTED;
  if ((double)d->B * d->M * d->Hout * d->Wout > 2147483647.0) return TBG_ERANGE;
  ConvS2P p{};
  p.XU = reinterpret_cast<const char *>(XU); p.Wf = reinterpret_cast<const char *>(w);
  p.x_plane = s2_units_per_plane(d->B, d->C, d->Hout, d->Wout);
  p.C8 = d->C / 8;
  p.w_plane = (long long)9 * p.C8 * d->ldw;
  if (p.x_plane * planes > 2147483647LL / 2 || p.w_plane * planes > 2147483647LL / 2) return TBG_ERANGE;
  p.y = y; p.B = d->B; p.M = d->M; p.Ho = d->Hout; p.Wo = d->Wout; p.Hq = d->Hout + 1; p.Wq = d->Wout + 1; p.ldw = d->ldw;
  p.flip = d->flip;
  p.tilesU = d->Hout / 8; p.tilesV = d->Wout / 32;
  p.dot_slots = p.tilesU * p.tilesV * 4;
  p.e = make_epi(epi);
  if (const int rcs = epi_sink_geometry(p.e, d->B, d->M, d->Hout, d->Wout)) return rcs;
  hipStream_t st = tbg_stream(stream);
  if (units_s2_wtm(d) == 2) return planes == 3 ? launch_conv_units_s2<3, 2>(p, st) : launch_conv_units_s2<1, 2>(p, st);
  return planes == 3 ? launch_conv_units_s2<3, 1>(p, st) : launch_conv_units_s2<1, 1>(p, st);
}

// ============================================================================================
// filter gradient of the 3x3 stride-2 pad-0 convolution from unit tensors
// ============================================================================================
// dW[t][cl][cs] = sum_{b,y,x} S[b,cs,y,x] * t_in[b,cl,2y+kh,2x+kw]   with S (the gradient on the Ho x Wo output grid) as a
// stride-1 unit tensor (conv_units.hip: ring of one zero unit) and t_in as a PHASE unit tensor, so that every tap is a
// unit-stride shifted read of one phase plane and the contraction over PIXELS runs on transposing LDS reads exactly as in
// conv_wgrad_units_kernel.  What the stride costs is operand traffic: per 32 output pixels the four phase tiles are 3 x the
// bytes of a stride-1 halo -- a 64 x 64-channel block of 4 waves would need 28 B of LDS-DMA per matrix cycle and CU (the
// stride-1 kernel: 11.6; the pipe sustains < 20).  Hence:
//   * block = 128 S-channels x 64 L-channels, EIGHT waves (2 per SIMD; wave = 32 x 32 x 9 taps = 144 accumulator registers):
//     the phase tiles feed twice the MFMAs -- 16 B per matrix cycle;
//   * K chunk = ONE output row x 32 pixels, contracted in the two stages of conv_units_s2_fprop_kernel (phases (0,0)+(1,1): 5
//     taps; (0,1)+(1,0): 4 taps), each stage's L tiles in its own buffer (x3: 41 KB each), the S tile (x3: 27 KB) double-buffered:
//     the L tiles of stage 1 and the S tile of the next chunk arrive under the MFMAs of stage 0, the L tiles of the next chunk's
//     stage 0 under stage 1;
//   * the partial tiles are written in the 64 x 64 layout of the 4-wave kernels (waves 0-3 / 4-7 = S-channels 0-63 / 64-127), so
//     the shared reduce kernels sum them.
// LDS rows: S [plane][16 units][S_ROW = 36] (32 used), L [plane][8 units][L_ROW = 108]: slot 0 = 2 rows x 34 positions of the
// stage's first phase, slot 1 (at 68) = 1 row x 34 of its second (pitches = 4 / 12 mod 16: the four channel units a transposing
// read touches fall on disjoint banks).
struct WgS2P {
  const char *SU, *LU;
  long long s_plane, l_plane;  // units per plane
  int CS8, CL8, Hps, Wps, Hq, Wq;
  int Ho, tilesV, nchunks, ksplit;
  float *ws;
  int tx2, ty;  // 128-channel S tiles, 64-channel L tiles; the launch is 1-D: tx2 * ty * ksplit blocks
};

template <int NP>
__global__ __launch_bounds__(512, 2) void conv_wgrad_units_s2_kernel(const WgS2P p) {
  constexpr int S_ROW = 36, L_ROW = 108, NT = 9;
  constexpr int S_UNITS = NP * 16 * S_ROW, L_RAW = NP * 8 * L_ROW;
  static_assert(S_UNITS % 64 == 0, "the S tile is a whole number of DMA pieces");
  constexpr int S_PIECES = S_UNITS / 64, L_PIECES = (L_RAW + 63) / 64, LBUF = L_PIECES * 64;
  constexpr int PPW_A = (L_PIECES + S_PIECES + 7) / 8;  // issued during stage 0: L of stage 1 (this chunk) + S of the next chunk
  constexpr int PPW_B = (L_PIECES + 7) / 8;             // issued during stage 1: L of stage 0 (next chunk)
  constexpr int NQ = NP == 3 ? 6 : 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // units: [S0][S1][L stage 0][L stage 1]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sgrp = wave >> 1, wl = wave & 1;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;
  constexpr int OFF_L0 = 2 * S_UNITS, OFF_L1 = 2 * S_UNITS + LBUF;

  // ---- DMA descriptors: per-lane source OFFSET (units, relative to the chunk base of its tensor) of every piece this wave issues
  // (32-bit: the tensors hold < 2^31 bytes per plane set, checked on the host)
  auto l_off = [&](int st, int q) -> int {  // piece q of a stage's L buffer
    const int n = min(q * 64 + lane, L_RAW - 1);
    const int row = n / L_ROW, pos = n - row * L_ROW;
    const int pl = row >> 3, lu = row & 7;
    const int slot = pos >= 68 ? 1 : 0, ps = min(pos - 68 * slot, slot ? 33 : 67);
    const int r = ps / 34, c = min(ps - r * 34, 32);
    const int ph = st == 0 ? (slot == 0 ? 0 : 3) : (slot == 0 ? 1 : 2);
    return (int)(pl * p.l_plane) + ((lu * 4 + ph) * p.Hq + r) * p.Wq + c;
  };
  auto s_off = [&](int q) -> int {
    const int n = q * 64 + lane;
    const int row = n / S_ROW, pix = min(n - row * S_ROW, 31);
    const int pl = row >> 4, su = row & 15;
    return (int)(pl * p.s_plane) + su * p.Hps * p.Wps + pix;
  };
  int offA[PPW_A], offB[PPW_B];
#pragma unroll
  for (int k = 0; k < PPW_A; ++k) {
    const int q = min(wave + 8 * k, L_PIECES + S_PIECES - 1);
    offA[k] = q < L_PIECES ? l_off(1, q) : s_off(q - L_PIECES);
  }
#pragma unroll
  for (int k = 0; k < PPW_B; ++k) offB[k] = l_off(0, min(wave + 8 * k, L_PIECES - 1));
  const char *const su_ = p.SU, *const lu_ = p.LU;
  // block -> (S tile, L tile, K slice): the blocks of one K slice read the same pixel chunks, so whole groups of 8 slices are dealt
  // over the XCDs and a slice's blocks share one L2 (conv_wgrad_units_kernel)
  const int tiles = p.tx2 * p.ty;
  int bt, bz;
  if ((p.ksplit & 7) == 0) {
    const int slot = blockIdx.x >> 3;
    bt = slot % tiles;
    bz = (slot / tiles) * 8 + (blockIdx.x & 7);
  } else {
    bt = blockIdx.x % tiles;
    bz = blockIdx.x / tiles;
  }
  const int bx = bt % p.tx2, by = bt / p.tx2;
  const int cs8 = bx * 16, cl8 = by * 8;

  auto chunk_bases = [&](int chunk, int &sb, int &lb) {
    const int tv = chunk % p.tilesV;
    const int t2 = chunk / p.tilesV;
    const int y = t2 % p.Ho, b = t2 / p.Ho;
    sb = ((b * p.CS8 + cs8) * p.Hps + y + 1) * p.Wps + tv * 32 + 1;  // interior starts at (1, 1)
    lb = ((b * p.CL8 + cl8) * 4 * p.Hq + y) * p.Wq + tv * 32;
  };
  // pieces of stage 0's issue list: q < L_PIECES -> L buffer of stage 1 (chunk lb), else S buffer `sbuf` (chunk sb_next)
  auto issueA = [&](int k, int lb, int sb_next, int sbuf) {
    const int q = min(__builtin_amdgcn_readfirstlane(wave) + 8 * k, L_PIECES + S_PIECES - 1);
    if (q < L_PIECES) dma16(lu_ + ((long long)(lb + offA[k]) << 4), lds0 + (unsigned)((OFF_L1 + q * 64) * 16));
    else dma16(su_ + ((long long)(sb_next + offA[k]) << 4), lds0 + (unsigned)((sbuf * S_UNITS + (q - L_PIECES) * 64) * 16));
  };
  auto issueB = [&](int k, int lb_next) {
    const int q = min(__builtin_amdgcn_readfirstlane(wave) + 8 * k, L_PIECES - 1);
    dma16(lu_ + ((long long)(lb_next + offB[k]) << 4), lds0 + (unsigned)((OFF_L0 + q * 64) * 16));
  };

  // ---- operand addressing (conv_wgrad_units_kernel: lane i of a 16-lane group supplies the address of (pixel i / 4, channels
  // 4 (i % 4) .. + 3) and receives channel i at the 4 pixels; two reads = 8 consecutive pixels of one channel)
  const int i16 = lane & 15, grp = lane >> 4;
  const int jj = i16 >> 2, cq = i16 & 3, chblk = grp & 1, khalf = grp >> 1;
  const int a_lane = ((sgrp * 4 + chblk * 2 + (cq >> 1)) * S_ROW + 8 * khalf + jj) * 16 + (cq & 1) * 8;
  const int b_lane = ((wl * 4 + chblk * 2 + (cq >> 1)) * L_ROW + 8 * khalf + jj) * 16 + (cq & 1) * 8;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  auto rd = [&](const char *ptr) -> bf16x8 {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(ptr));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(ptr + 64));
    return __builtin_bit_cast(bf16x8, s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
  };

  // one stage of one chunk: steps (16-pixel group g, tap slot ts), NQ x 1 MFMAs each; behind them the operand reads of the next
  // step and this stage's DMA issue list
  auto run_stage = [&](auto stc, int sbuf, auto &&issue) {
    constexpr int ST = decltype(stc)::value;
    constexpr int NTS = ST == 0 ? 5 : 4, NS = 2 * NTS, PPW = ST == 0 ? PPW_A : PPW_B;
    const char *Ab = smem + (size_t)sbuf * S_UNITS * 16 + a_lane;
    const char *Bb = smem + (size_t)(ST == 0 ? OFF_L0 : OFF_L1) * 16 + b_lane;
    bf16x8 a[2][NP], bv[2][NP];
    auto ld = [&](int s, int idx) {  // operand idx of step s: the NP B planes, then (first tap of a group) the NP A planes
      const int g = s / NTS, ts = s - g * NTS;
      if (idx < NP)
        bv[s & 1][idx] = rd(Bb + (idx * 8 * L_ROW + s2_slot(ST, ts) * 68 + s2_roff(ST, ts) * 34 + 16 * g + s2_coff(ST, ts)) * 16);
      else if (ts == 0 && idx < 2 * NP)
        a[g][idx - NP] = rd(Ab + ((idx - NP) * 16 * S_ROW + 16 * g) * 16);
    };
#pragma unroll
    for (int idx = 0; idx < 2 * NP; ++idx) ld(0, idx);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int g = s / NTS, ts = s - g * NTS, t = s2_tap(ST, ts);
      __builtin_amdgcn_sched_barrier(0);
      // six partial products per tap, smallest first: (hi,lo) (lo,hi) (mid,mid) (hi,mid) (mid,hi) (hi,hi)
      constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
      for (int m = 0; m < NQ; ++m) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[g][NP == 3 ? PA[m] : 0], bv[s & 1][NP == 3 ? PB[m] : 0], acc[t], 0, 0, 0);
        if (s + 1 < NS) {
          constexpr int LPM = (2 * NP + NQ - 1) / NQ;
#pragma unroll
          for (int e = 0; e < LPM; ++e) ld(s + 1, m * LPM + e);
        }
        {
          constexpr int STRIDE = (NS * NQ / 2) / PPW > 0 ? (NS * NQ / 2) / PPW : 1;
          static_assert(NS * NQ >= PPW, "every DMA piece has an MFMA to hide behind");
          const int slot = s * NQ + m;
          if (slot % STRIDE == 0 && slot / STRIDE < PPW) issue(slot / STRIDE);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };

  int chunk = bz;
  int sb = 0, lb = 0, sbn = 0, lbn = 0;
  if (chunk < p.nchunks) {  // prologue: S tile and stage-0 L tiles of the first chunk
    chunk_bases(chunk, sb, lb);
#pragma unroll
    for (int k = 0; k < PPW_B; ++k) issueB(k, lb);
#pragma unroll
    for (int k = 0; k < PPW_A; ++k) {
      const int q = min(__builtin_amdgcn_readfirstlane(wave) + 8 * k, L_PIECES + S_PIECES - 1);
      if (q >= L_PIECES) dma16(su_ + ((long long)(sb + offA[k]) << 4), lds0 + (unsigned)(((q - L_PIECES) * 64) * 16));
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int sbuf = 0;
  for (; chunk < p.nchunks; chunk += p.ksplit) {
    chunk_bases(chunk + p.ksplit < p.nchunks ? chunk + p.ksplit : chunk, sbn, lbn);  // (the last chunk re-fetches itself)
    run_stage(std::integral_constant<int, 0>{}, sbuf, [&](int k) { issueA(k, lb, sbn, sbuf ^ 1); });
    run_stage(std::integral_constant<int, 1>{}, sbuf, [&](int k) { issueB(k, lbn); });
    sb = sbn; lb = lbn;
    sbuf ^= 1;
  }

  // partial tile, in the 64 x 64 layout of the 4-wave kernels
  const int tile = sgrp >> 1, tid4 = ((sgrp & 1) * 2 + wl) * 64 + lane;
  const size_t blk = ((size_t)bz * p.ty + by) * (2 * p.tx2) + 2 * bx + tile;
  float *wsp = p.ws + blk * (size_t)(NT * 16 * 256);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r16 = 0; r16 < 16; ++r16) wsp[(t * 16 + r16) * 256 + tid4] = acc[t][r16];
}

static bool wgrad_units_s2_ok(const tbg_wgrad_desc *d) {
  return d->KH == 3 && d->KW == 3 && d->sy == 2 && d->sx == 2 && d->py == 0 && d->px == 0 && d->Hl >= 3 && d->Wl >= 3 &&
         d->Hs == (d->Hl - 3) / 2 + 1 && d->Ws == (d->Wl - 3) / 2 + 1 && (d->Ws % 32) == 0 && (d->CS % 128) == 0 && (d->CL % 64) == 0;
}

static int wgrad_units_s2_ksplit(const tbg_wgrad_desc *d) {
  const int tiles = (d->CS / 128) * (d->CL / 64);
  const int nchunks = d->B * d->Hs * (d->Ws / 32);
  return wgrad_ksplit(tiles, nchunks, 256);  // one block per CU
}

extern "C" long long tbg_conv2d_wgrad_units_s2_workspace_bytes(const tbg_wgrad_desc *d) {
  if (!d || d->B < 1 || d->CS < 1 || d->CL < 1 || d->Hs < 1 || d->Ws < 1) return TBG_EINVAL;
  if (!wgrad_units_s2_ok(d)) return TBG_EUNSUPPORTED;
  return (long long)wgrad_units_s2_ksplit(d) * (d->CS / 64) * (d->CL / 64) * 9 * 16 * 256 * (long long)sizeof(float);
}

template <int NP>
static int launch_wgrad_units_s2(WgS2P &u, WgradP &p, hipStream_t st) {
  constexpr int S_UNITS = NP * 16 * 36, LBUF = ((NP * 8 * 108 + 63) / 64) * 64;
  const size_t lds = (size_t)(2 * S_UNITS + 2 * LBUF) * 16;
  auto kern = conv_wgrad_units_s2_kernel<NP>;
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return TBG_EHIP;
  const int tx = p.CS / 64, ty = p.CL / 64;
  u.tx2 = p.CS / 128; u.ty = ty;
  hipLaunchKernelGGL(kern, dim3(u.tx2 * ty * u.ksplit), dim3(512), lds, st, u);
  TBG_LAUNCH_CHECK();
  if (tx * ty * 9 >= 256)
    hipLaunchKernelGGL((conv_wgrad_reduce_kernel<2, 2, 9>), dim3(tx, ty, 9), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL((conv_wgrad_reduce_wide_kernel<2, 2, 9>), dim3(tx, ty, 9 * 16), dim3(256), 0, st, p);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

extern "C" int tbg_conv2d_wgrad_units_s2(const tbg_wgrad_desc *d, const void *SU, const void *LU, int planes, float *dW,
                                         const float *addw, const float *addq, float gamma, float *workspace,
                                         long long workspace_bytes, void *stream) {
  if (!d || !SU || !LU || !dW || !workspace || ((addw == nullptr) != (addq == nullptr)) || (planes != 1 && planes != 3))
    return TBG_EINVAL;
  if (d->B < 1 || d->CS < 1 || d->CL < 1 || d->Hs < 1 || d->Ws < 1 || d->Hl < 1 || d->Wl < 1) return TBG_EINVAL;
  if (((reinterpret_cast<uintptr_t>(SU) | reinterpret_cast<uintptr_t>(LU)) & 15) != 0) return TBG_EINVAL;
  if (!wgrad_units_s2_ok(d)) return TBG_EUNSUPPORTED;
  const long long s_plane = (long long)d->B * (d->CS / 8) * (d->Hs + 2) * (d->Ws + 2);
  const long long l_plane = s2_units_per_plane(d->B, d->CL, d->Hs, d->Ws);
  if (s_plane * planes > 2147483647LL / 16 || l_plane * planes > 2147483647LL / 16) return TBG_ERANGE;  // 32-bit unit offsets
  WgS2P u{};
  u.SU = reinterpret_cast<const char *>(SU); u.LU = reinterpret_cast<const char *>(LU);
  u.s_plane = s_plane; u.l_plane = l_plane;
  u.CS8 = d->CS / 8; u.CL8 = d->CL / 8;
  u.Hps = d->Hs + 2; u.Wps = d->Ws + 2; u.Hq = d->Hs + 1; u.Wq = d->Ws + 1;
  u.Ho = d->Hs; u.tilesV = d->Ws / 32;
  u.nchunks = d->B * d->Hs * u.tilesV;
  u.ksplit = wgrad_units_s2_ksplit(d);
  u.ws = workspace;
  if ((long long)u.ksplit * (d->CS / 64) * (d->CL / 64) * 9 * 16 * 256 * (long long)sizeof(float) > workspace_bytes) return TBG_EINVAL;
  WgradP p{};  // what the reduce kernels read
  p.CS = d->CS; p.CL = d->CL; p.st_t = d->st_t; p.st_l = d->st_l; p.st_s = d->st_s; p.alpha = d->alpha;
  p.dW = dW; p.ws = workspace; p.addw = addw; p.addq = addq; p.gamma = gamma; p.ksplit = u.ksplit;
  if (const int rcb = wgrad_bias_rider(p, d)) return rcb;
  return planes == 3 ? launch_wgrad_units_s2<3>(u, p, tbg_stream(stream)) : launch_wgrad_units_s2<1>(u, p, tbg_stream(stream));
}

// ============================================================================================
// transposed convolution (3x3, stride 2) from a stride-1 unit tensor: the up-convolution's forward, the strided convolution's
// data gradient
// ============================================================================================
// y[b,m,2u+kh,2v+kw] += x[b,c,u,v] w[kh,kw,c,m] in the MERGED-class form of conv_fprop_kernel<..,TM>: grid position (u, v) owns the
// four outputs (2u + cy, 2v + cx), tap (kh, kw) accumulates into class (kh & 1, kw & 1) reading x[u - (kh == 2), v - (kw == 2)] --
// nine taps, four accumulator sets, one staged tile.  The class grids are (H + 1) x (W + 1) (33 x 129 for a 32 x 128 map): 8 x 32
// tiles pad that to 40 x 160 (+46 %; the NCHW kernel's 4 x 32 tiles: +35 %).  Here the grid is tiled FLAT: a tile is 256
// consecutive positions q = (u + 1) Wp + (v + 1) of the PADDED unit tensor (row pitch Wp = W + 2, u in [0, H], v in [-1, W]; the
// v = -1 column is a dummy position whose outputs are masked), so that
//   * the four operands of a position are q, q - 1, q - Wp, q - Wp - 1: two contiguous runs of 257 units per channel unit and
//     plane -- the simplest DMA there is; the padding (row u = H, column v = W, the row above u = 0) IS the zero ring;
//   * a tap shift is again a unit-stride LDS offset (run select + one unit);
//   * (H + 1) Wp positions per image in tiles of 256: 33 x 130 = 4290 -> 17 tiles (+5 %).
// Block = 64 output channels x 256 positions, 8 waves (wave = 32 channels x 64 positions x 4 classes = 128 accumulator
// registers), both tiles double-buffered, everything else as conv_units_fprop_kernel.  Store-only epilogue (alpha * acc), scattered
// by class.
struct ConvT2P {
  const char *XU, *Wf;
  long long x_plane, w_plane;  // 16-byte units per plane
  float *y;
  int B, C8, M, H, W, Hp, Wp, Hout, Wout, ldw, flip, tilesQ;
  float alpha;
};

template <int NP>
struct T2Cfg {
  static constexpr int BM = 64, CKU = NP == 3 ? 1 : 2, RUNP = 260;
  static constexpr int A_UNITS = NP * 9 * CKU * BM, X_UNITS = NP * CKU * 2 * RUNP;
  static constexpr int NPIECE = (A_UNITS + X_UNITS + 63) / 64, BUF = NPIECE * 64, PPW = (NPIECE + 7) / 8;
  static_assert(A_UNITS % 64 == 0, "a DMA piece never straddles the filter / run regions");
};

template <int NP>
__global__ __launch_bounds__(512, 2) void conv_units_t2_kernel(const ConvT2P p) {
  using Cf = T2Cfg<NP>;
  constexpr int WTN = 2, BM = Cf::BM, CKU = Cf::CKU, RUNP = Cf::RUNP, A_UNITS = Cf::A_UNITS, X_UNITS = Cf::X_UNITS;
  constexpr int NPIECE = Cf::NPIECE, BUF = Cf::BUF, PPW = Cf::PPW;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][BUF] units
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3, half = lane >> 5;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;

  const int mtiles = p.M / BM, total = gridDim.x;  // block -> (position tile, channel tile): one contiguous eighth per XCD, as above
  int lin = blockIdx.x;
  if ((total & 7) == 0) lin = (blockIdx.x & 7) * (total >> 3) + (blockIdx.x >> 3);
  const int tn = lin / mtiles, mt = lin - tn * mtiles;
  const int tq = tn % p.tilesQ, b = tn / p.tilesQ;
  const int m0 = mt * BM;
  const int q0 = p.Wp + tq * 256, qmax = p.Hp * p.Wp - 1;

  const char *dsrc[PPW];
  const char *const xu = p.XU, *const wf = p.Wf;
#pragma unroll
  for (int k = 0; k < PPW; ++k) {
    const int n = min(wave + 8 * k, NPIECE - 1) * 64 + lane;
    if (n < A_UNITS) {  // As[plane][tap][unit][BM]
      const int row = n / BM, m = n - row * BM;
      const int u = row % CKU, t = (row / CKU) % 9, pl = row / (9 * CKU);
      dsrc[k] = wf + ((pl * p.w_plane + (long long)((p.flip ? 8 - t : t) * p.C8 + u) * p.ldw + m0 + m) << 4);
    } else {            // Xs[plane][unit][run][RUNP]: run 0 = units q0 - 1 .., run 1 = the same one padded row up
      const int m2 = min(n - A_UNITS, X_UNITS - 1);
      const int row = m2 / RUNP, pos = min(m2 - row * RUNP, 256);
      const int run = row & 1, u = (row >> 1) % CKU, pl = (row >> 1) / CKU;
      const int q = min(max(q0 - 1 + pos - run * p.Wp, 0), qmax);
      dsrc[k] = xu + ((pl * p.x_plane + (long long)(b * p.C8 + u) * p.Hp * p.Wp + q) << 4);
    }
  }
  const long long a_step = (long long)CKU * p.ldw * 16, x_step = (long long)CKU * p.Hp * p.Wp * 16;  // bytes per chunk
  auto issue_piece = [&](int k, int kc, int buf) {
    const int q = min(__builtin_amdgcn_readfirstlane(wave) + 8 * k, NPIECE - 1);
    const long long step = q * 64 < A_UNITS ? a_step : x_step;  // wave-uniform
    dma16(dsrc[k] + kc * step, lds0 + (unsigned)((buf * BUF + q * 64) * 16));
  };

  const int a_lane = (wm * 32 + (lane & 31)) * 16;
  int b_lane[WTN];
#pragma unroll
  for (int j = 0; j < WTN; ++j) b_lane[j] = A_UNITS * 16 + (1 + wn * 64 + j * 32 + (lane & 31)) * 16;
  constexpr int A_PL = 9 * CKU * BM * 16, X_PL = CKU * 2 * RUNP * 16;
  const int aX = NP == 3 ? half * A_PL : half * BM * 16, aY = 2 * half * A_PL;
  const int bZ = 2 * (1 - half) * X_PL, bH = NP == 3 ? 0 : half * 2 * RUNP * 16;

  f32x16 acc[4][WTN];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][j][r] = 0.f;

  const int nchunks = p.C8 / CKU;
#pragma unroll
  for (int k = 0; k < PPW; ++k) issue_piece(k, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int buf = 0;
  for (int kc = 0; kc < nchunks; ++kc) {
    const int kn = kc + 1 < nchunks ? kc + 1 : kc;  // (the last chunk re-fetches itself: no branch in the loop body)
    const char *Ab = smem + (size_t)buf * BUF * 16 + a_lane;
    const char *Xb = smem + (size_t)buf * BUF * 16;
    constexpr int NA = NP == 3 ? 2 : 1, NB = NP == 3 ? 3 * WTN : WTN, NL = NA + NB;
    constexpr int NM = (NP == 3 ? 3 : 1) * WTN;  // MFMAs per tap
    bf16x8 av[2][NA], bw[2][NB];
    auto ld1 = [&](int t, int bs, int idx) {
      const int kh = t / 3, kw = t - 3 * kh;
      if (idx < NA) {
        const int plane_off = NP == 3 ? ((idx & 1) ? aY : aX) : aX;
        av[bs][idx] = *reinterpret_cast<const bf16x8 *>(Ab + t * CKU * BM * 16 + plane_off);
      } else if (idx < NL) {
        const int e = idx - NA;
        const int j = NP == 3 ? e / 3 : e, w = NP == 3 ? e - 3 * j : 0;
        const int plane_off = NP == 3 ? (w == 0 ? 0 : w == 1 ? X_PL : bZ) : bH;
        bw[bs][e] = *reinterpret_cast<const bf16x8 *>(Xb + b_lane[j] + ((kh == 2 ? RUNP : 0) - (kw == 2 ? 1 : 0)) * 16 + plane_off);
      }
    };
#pragma unroll
    for (int idx = 0; idx < NL; ++idx) ld1(0, 0, idx);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int bs = t & 1, kh = t / 3, kw = t - 3 * kh, c = (kh & 1) * 2 + (kw & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int mm = 0; mm < NM; ++mm) {
        const int grp = mm / WTN, j = mm - grp * WTN;
        if constexpr (NP == 3) {  // smallest terms first: (hi|lo)x(lo|hi), then (hi|mid) x mid, then (hi|mid) x hi
          const int ai = grp == 0 ? 1 : 0, bi = 3 * j + (grp == 0 ? 2 : grp == 1 ? 1 : 0);
          acc[c][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[bs][ai], bw[bs][bi], acc[c][j], 0, 0, 0);
        } else {
          acc[c][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[bs][0], bw[bs][j], acc[c][j], 0, 0, 0);
        }
        if (t + 1 < 9) {
          constexpr int LPM = (NL + NM - 1) / NM;
#pragma unroll
          for (int e = 0; e < LPM; ++e) ld1(t + 1, bs ^ 1, mm * LPM + e);
        }
        {
          constexpr int STRIDE = (9 * NM / 2) / PPW > 0 ? (9 * NM / 2) / PPW : 1;
          static_assert(9 * NM >= PPW, "every DMA piece has an MFMA to hide behind");
          const int slot = t * NM + mm;
          if (slot % STRIDE == 0 && slot / STRIDE < PPW) issue_piece(slot / STRIDE, kn, buf ^ 1);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    buf ^= 1;
  }

  // ---- store-only epilogue: position q -> (u, v); class (cy, cx) -> output (2u + cy, 2v + cx).  The two column classes of a
  // position are neighbours in memory: one 8-byte store (rows are only 4-byte aligned: 257-float rows), and the 32 positions of a
  // half-wave make one contiguous 256-byte run per output row and channel
  typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
  const float alpha = p.alpha;
  const int HWout = p.Hout * p.Wout;
#pragma unroll
  for (int j = 0; j < WTN; ++j) {
    const int q = q0 + wn * 64 + j * 32 + (lane & 31);
    const int row = q / p.Wp, v = q - row * p.Wp - 1, u = row - 1;
    const bool okp = v >= 0 && u <= p.H;
#pragma unroll
    for (int cy = 0; cy < 2; ++cy) {
      const int Y = 2 * u + cy, X = 2 * v;
      const bool ok0 = okp && Y < p.Hout && X < p.Wout, ok1 = ok0 && X + 1 < p.Wout;
      float *yb = p.y + (size_t)b * p.M * HWout + (size_t)(ok0 ? Y * p.Wout + X : 0);
#pragma unroll
      for (int r16 = 0; r16 < 16; ++r16) {
        const int m = m0 + wm * 32 + (r16 & 3) + 8 * (r16 >> 2) + 4 * (lane >> 5);
        const float v0 = acc[2 * cy][j][r16] * alpha, v1 = acc[2 * cy + 1][j][r16] * alpha;
        if (ok1) *reinterpret_cast<f32x2u *>(yb + (size_t)m * HWout) = f32x2u{v0, v1};
        else if (ok0) yb[(size_t)m * HWout] = v0;
      }
    }
  }
}

static bool conv_units_t2_ok(const tbg_conv_desc *d, int planes) {
  const int cku = planes == 3 ? 8 : 16;
  return d->transposed && d->KH == 3 && d->KW == 3 && d->sy == 2 && d->sx == 2 && d->py == 0 && d->px == 0 &&
         d->Hout >= 2 * d->Hin + 1 && d->Hout <= 2 * d->Hin + 2 && d->Wout >= 2 * d->Win + 1 && d->Wout <= 2 * d->Win + 2 &&
         (d->C % cku) == 0 && (d->M % 64) == 0 && d->ksplit == 1 && d->ldw >= d->M;
}

extern "C" int tbg_conv2d_units_t2_blocks(const tbg_conv_desc *d, int planes) {
  if (!d || (planes != 1 && planes != 3)) return TBG_EINVAL;
  if (!conv_units_t2_ok(d, planes)) return TBG_EUNSUPPORTED;
  const long long n = (long long)d->B * (((long long)(d->Hin + 1) * (d->Win + 2) + 255) / 256) * (d->M / 64);
  return n > 2147483647LL ? TBG_ERANGE : (int)n;
}

template <int NP>
static int launch_conv_units_t2(ConvT2P &p, hipStream_t st) {
  using Cf = T2Cfg<NP>;
  const size_t lds = (size_t)2 * Cf::BUF * 16;
  auto kern = conv_units_t2_kernel<NP>;
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return TBG_EHIP;
  hipLaunchKernelGGL(kern, dim3(p.B * p.tilesQ * (p.M / Cf::BM)), dim3(512), lds, st, p);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

extern "C" int tbg_conv2d_units_t2(const tbg_conv_desc *d, const void *XU, int planes, const void *w, float *y, float alpha,
                                   void *stream) {
  if (!d || !XU || !w || !y || (planes != 1 && planes != 3)) return TBG_EINVAL;
  if (d->B < 1 || d->C < 1 || d->M < 1 || d->Hin < 1 || d->Win < 1) return TBG_EINVAL;
  if (((reinterpret_cast<uintptr_t>(XU) | reinterpret_cast<uintptr_t>(w)) & 15) != 0) return TBG_EINVAL;
  if (!conv_units_t2_ok(d, planes)) return TBG_EUNSUPPORTED;
  if ((double)d->B * d->M * d->Hout * d->Wout > 2147483647.0) return TBG_ERANGE;
  ConvT2P p{};
  p.XU = reinterpret_cast<const char *>(XU); p.Wf = reinterpret_cast<const char *>(w);
  p.C8 = d->C / 8;
  p.Hp = d->Hin + 2; p.Wp = d->Win + 2;
  p.x_plane = (long long)d->B * p.C8 * p.Hp * p.Wp;
  p.w_plane = (long long)9 * p.C8 * d->ldw;
  if (p.x_plane * planes > 2147483647LL / 2 || p.w_plane * planes > 2147483647LL / 2) return TBG_ERANGE;
  p.y = y; p.B = d->B; p.M = d->M; p.H = d->Hin; p.W = d->Win; p.Hout = d->Hout; p.Wout = d->Wout; p.ldw = d->ldw;
  p.flip = d->flip; p.alpha = alpha;
  p.tilesQ = ((d->Hin + 1) * p.Wp + 255) / 256;
  hipStream_t st = tbg_stream(stream);
  return planes == 3 ? launch_conv_units_t2<3>(p, st) : launch_conv_units_t2<1>(p, st);
}
