// Convolution operands in the 8-CHANNEL-UNIT layout, and the kernels that consume them without a staging pass.
//
// Unit tensor of an activation x[B][C][H][W] (tbg.h "unit tensors"):
//     U[plane][b][c/8][1 + y][1 + x][c % 8]   bf16,   planes = 1 (bf16 mode: RNE(x)) or 3 (f32x3: hi | mid | lo, split3)
// i.e. one 16-byte UNIT = 8 consecutive channels of one pixel, rows of W + 2 units, H + 2 rows: a ring of ZERO units
// around every (b, c/8) plane, channels past C zero.  Why this layout:
//   * a 16-byte unit is the LDS-DMA granule (global_load_lds_dwordx4: 16 B per lane, lane-linear LDS image, arbitrary per-lane
//     source address), so ANY tile -- halo included, the padding served by the zero ring -- goes HBM -> LDS with no staging
//     registers and no VALU work: the x * s modulation, the fp32 -> 3 x bf16 split and the bf16 rounding were done ONCE by
//     the producer instead of once per consumer tile (conv.hip re-does them in every forward, data-gradient and
//     filter-gradient launch: 25-40 % of those kernels);
//   * forward / data gradient contract over CHANNELS: the unit is one ds_read_b128 MFMA operand (K = 8 channels per half-wave);
//   * the filter gradient contracts over PIXELS: gfx950's transposing LDS read (ds_read_b64_tr_b16) turns a [4 pixels][16
//     channels] block of the same image into "4 consecutive pixels of one channel per lane", so the SAME tensor feeds it --
//     and a tap shift is an address offset (the NCHW bf16 form needed a 10-pixel window cut with v_alignbit per tap row).
#include "conv_common.h"

#define UNIT_RING 1
// TBG_EXP: ablation builds of conv_units_fprop_kernel for tools/archive/exp_units_fprop.sh (where its time goes).  0 = product.
#ifndef TBG_EXP
#define TBG_EXP 0
#endif

static inline long long units_per_plane(int B, int C, int H, int W) {
  return (long long)B * ((C + 7) / 8) * (H + 2 * UNIT_RING) * (W + 2 * UNIT_RING);
}

extern "C" long long tbg_units_bytes(int B, int C, int H, int W, int planes) {
  if (B < 1 || C < 1 || H < 1 || W < 1 || (planes != 1 && planes != 3)) return TBG_EINVAL;
  return units_per_plane(B, C, H, W) * planes * 16;
}

// ---- producer of last resort: NCHW fp32 (x optional per-(b,c) scale) -> unit tensor, ring and channel tail included.
// One lane per unit (two units per thread, both load batches in flight together): 8 channel loads (each coalesced along x
// across the wave), one 16-byte store per plane.
template <int NP>
__global__ __launch_bounds__(256) void units_pack_kernel(const float *__restrict__ x, const float *__restrict__ scale,
                                                         bf16x8 *__restrict__ U, int B, int C, int H, int W, long long plane) {
  const int Wp = W + 2, Hp = H + 2, C8 = (C + 7) >> 3;
  const long long HW = (long long)H * W;
  const long long half = (plane + 1) >> 1;
  const long long n0 = (long long)blockIdx.x * 256 + threadIdx.x;
  if (n0 >= half) return;
  float v[2][8];
  long long nn[2];
  bool live[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const long long n = n0 + k * half;
    nn[k] = n; live[k] = n < plane;
    const long long nc = live[k] ? n : 0;
    const int xp = (int)(nc % Wp);
    long long t = nc / Wp;
    const int yp = (int)(t % Hp);
    t /= Hp;
    const int cu = (int)(t % C8), b = (int)(t / C8);
    const bool inside = live[k] && xp >= 1 && xp <= W && yp >= 1 && yp <= H;
    const long long g0 = ((long long)b * C + cu * 8) * HW + (long long)(yp - 1) * W + (xp - 1);
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
      const bool ok = inside && cu * 8 + cc < C;
      float a = x[ok ? g0 + cc * HW : 0];
      if (scale) a *= scale[ok ? b * C + cu * 8 + cc : 0];
      v[k][cc] = ok ? a : 0.f;
    }
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (!live[k]) continue;
    if constexpr (NP == 3) {
      bf16x8 h, m, l;
      split3_bf16x8(v[k], h, m, l);
      U[nn[k]] = h; U[plane + nn[k]] = m; U[2 * plane + nn[k]] = l;
    } else {
      U[nn[k]] = pack_bf16x8(v[k]);
    }
  }
}

extern "C" int tbg_units_pack_f32(const float *x, const float *scale, void *U, int B, int C, int H, int W, int planes,
                                  void *stream) {
  if (!x || !U || B < 1 || C < 1 || H < 1 || W < 1 || (planes != 1 && planes != 3)) return TBG_EINVAL;
  if ((reinterpret_cast<uintptr_t>(U) & 15) != 0) return TBG_EINVAL;
  const long long plane = units_per_plane(B, C, H, W);
  if (plane * 8 > 2147483647LL || (long long)B * C * H * W > 2147483647LL) return TBG_ERANGE;
  const dim3 grid((unsigned)(((plane + 1) / 2 + 255) / 256));
  if (planes == 3)
    hipLaunchKernelGGL(units_pack_kernel<3>, grid, dim3(256), 0, tbg_stream(stream), x, scale, reinterpret_cast<bf16x8 *>(U), B, C,
                       H, W, plane);
  else
    hipLaunchKernelGGL(units_pack_kernel<1>, grid, dim3(256), 0, tbg_stream(stream), x, scale, reinterpret_cast<bf16x8 *>(U), B, C,
                       H, W, plane);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

// ---- split-K second half with a unit sink: y = epilogue(sum_s x[s]) (tbg_slab_epilogue_f32) whose result ALSO (or only) leaves
// as units(y * units_scale).  Split-K layers are the small maps, so one lane per padded position of a (b, channel unit) plane
// (units_pack_kernel's mapping: ring and channel tail written as zeros by the lane that owns them): 8 channels x nslab loads per
// lane, slabs summed in slab order (the order of slab_epilogue_kernel: same bits), one 16-byte store per plane.
struct SlabUnitsP {
  const float *x;
  float *y;
  int B, M, H, W, nslab;
  long long slab, plane;
  EpiK e;
};

template <int NP>
__global__ __launch_bounds__(256) void slab_epilogue_units_kernel(const SlabUnitsP p) {
  const int Wp = p.W + 2, Hp = p.H + 2, C8 = p.M >> 3, HW = p.H * p.W;
  const long long n = (long long)blockIdx.x * 256 + threadIdx.x;
  if (n >= p.plane) return;
  const int xp = (int)(n % Wp);
  long long t = n / Wp;
  const int yp = (int)(t % Hp);
  t /= Hp;
  const int cu = (int)(t % C8), b = (int)(t / C8);
  const bool inside = xp >= 1 && xp <= p.W && yp >= 1 && yp <= p.H;
  const int pix = inside ? (yp - 1) * p.W + (xp - 1) : 0;
  float v[8];
#pragma unroll
  for (int cc = 0; cc < 8; ++cc) v[cc] = 0.f;
  if (inside) {
    const size_t g0 = ((size_t)b * p.M + cu * 8) * HW + pix;
    for (int s = 0; s < p.nslab; ++s) {
      float tv[8];
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) tv[cc] = p.x[(size_t)s * p.slab + g0 + (size_t)cc * HW];
#pragma unroll
      for (int cc = 0; cc < 8; ++cc) v[cc] += tv[cc];
    }
    const float str = p.e.noise ? p.e.strength[0] : 0.f;
    const float nz = p.e.noise ? p.e.noise[(size_t)b * HW + pix] * str : 0.f;
    const bool rf = p.e.residual && p.e.res_first;
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
      const int m = cu * 8 + cc, plane = b * p.M + m;
      const size_t gi = g0 + (size_t)cc * HW;
      const float sc = p.e.alpha * (p.e.out_scale ? p.e.out_scale[plane] : 1.f);
      float o = v[cc] * sc + (p.e.bias ? p.e.bias[m] * p.e.bias_mul : 0.f);
      if (p.e.noise) o += nz;
      if (rf) o += p.e.residual[gi];
      o = epi_act(p.e, o);
      if (p.e.residual && !rf) o = (o + p.e.residual[gi]) * p.e.res_scale;
      if (p.e.gate) o = p.e.gate[gi] > 0.f ? o : 0.f;
      if (p.y) p.y[gi] = o;
      v[cc] = o * (p.e.units_scale ? p.e.units_scale[plane] : 1.f);
    }
  }
  bf16x8 *U = static_cast<bf16x8 *>(p.e.units_out);
  if constexpr (NP == 3) {
    bf16x8 h, m, l;
    split3_bf16x8(v, h, m, l);
    U[n] = h; U[p.plane + n] = m; U[2 * p.plane + n] = l;
  } else {
    U[n] = pack_bf16x8(v);
  }
}

extern "C" int tbg_slab_epilogue_units_f32(const float *x, float *y, int B, int M, int H, int W, int nslab, const tbg_epilogue *epi,
                                           void *stream) {
  if (!x || B < 1 || M < 1 || H < 1 || W < 1 || nslab < 1 || !epi || !epi_valid(epi) || epi->dot_aux || !epi->units_out) return TBG_EINVAL;
  if ((double)B * M * H * W > 2147483647.0) return TBG_ERANGE;
  SlabUnitsP p{x, y, B, M, H, W, nslab, (long long)B * M * H * W, 0, make_epi(epi)};
  if (const int rc = epi_sink_geometry(p.e, B, M, H, W)) return rc;
  p.plane = p.e.units_plane >> 4;
  const dim3 grid((unsigned)((p.plane + 255) / 256));
  if (p.e.units_planes == 3) hipLaunchKernelGGL(slab_epilogue_units_kernel<3>, grid, dim3(256), 0, tbg_stream(stream), p);
  else hipLaunchKernelGGL(slab_epilogue_units_kernel<1>, grid, dim3(256), 0, tbg_stream(stream), p);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

// ---- fused producer: the backward of bias + noise + LeakyReLU (tbg_bias_act_bwd_f32) writing its result as a UNIT TENSOR.
//   dpre = dout * (residual_fused ? res_scale : 1) * gain * (out_act > 0 ? 1 : slope)
//   U    = units(dpre * alpha * out_scale[b,m])        (what the data-gradient and filter-gradient launches consume)
//   dpre_out (optional, NCHW fp32) and the per-(b, m, row chunk) partial sums part_db / part_dn / part_dyy as that entry.
// One block = one (b, 8-channel unit) and UB_ROWS rows of the padded plane; one lane = one padded position: 8 + 8 channel loads
// (each coalesced along x across the wave) + the noise value, one 16-byte store per plane; the 3 x 8 running sums are reduced
// once per block.  Replaces bias_act_bwd_kernel + units_pack_kernel: 8 + 2 planes bytes per element instead of 12 + 4 + 2 planes.
#define UB_ROWS 24
struct BabUnitsP {
  const float *__restrict__ dout, *__restrict__ out_act;
  bf16x8 *__restrict__ U;
  float *dpre_out, *part_db, *part_dn, *part_dyy;
  int B, M, H, W, nchunks;
  long long plane;
  EpiK e;
};

extern "C" int tbg_bias_act_bwd_units_chunks(int H) { return H < 1 ? 0 : (H + 2 + UB_ROWS - 1) / UB_ROWS; }

template <int NP>
__global__ __launch_bounds__(256) void bias_act_bwd_units_kernel(const BabUnitsP p) {
  __shared__ float red[4][24];
  const int Wp = p.W + 2, Hp = p.H + 2, C8 = (p.M + 7) >> 3, HW = p.H * p.W;
  const int cu = blockIdx.x % C8, b = blockIdx.x / C8;
  const int r0 = blockIdx.y * UB_ROWS, r1 = min(r0 + UB_ROWS, Hp);
  const float str = p.e.noise ? p.e.strength[0] : 0.f;
  const float gin = p.e.residual ? p.e.res_scale : 1.f;  // residual != NULL only flags "fused residual"
  const float g_pos = p.e.gain, g_neg = p.e.gain * (p.e.act == TBG_ACT_LRELU ? p.e.slope : 1.f);
  const float ig_pos = 1.f / g_pos, ig_neg = g_neg != 0.f ? 1.f / g_neg : 0.f;  // slope 0 = ReLU
  float sc[8], bias[8];
  bool chok[8];
#pragma unroll
  for (int cc = 0; cc < 8; ++cc) {
    const int m = cu * 8 + cc;
    chok[cc] = m < p.M;
    sc[cc] = chok[cc] ? p.e.alpha * (p.e.out_scale ? p.e.out_scale[b * p.M + m] : 1.f) : 0.f;
    bias[cc] = (chok[cc] && p.e.bias) ? p.e.bias[m] * p.e.bias_mul : 0.f;
  }
  float s_db[8], s_dn[8], s_dy[8];
#pragma unroll
  for (int cc = 0; cc < 8; ++cc) { s_db[cc] = 0.f; s_dn[cc] = 0.f; s_dy[cc] = 0.f; }
  const size_t g0 = ((size_t)b * p.M + cu * 8) * HW;
  bf16x8 *Ub = p.U + ((size_t)b * C8 + cu) * Hp * Wp;
  const int npos = (r1 - r0) * Wp;
  for (int e = threadIdx.x; e < npos; e += 256) {
    const int rr = e / Wp, xp = e - rr * Wp, yp = r0 + rr;
    const bool inside = xp >= 1 && xp <= p.W && yp >= 1 && yp <= p.H;
    const int pix = inside ? (yp - 1) * p.W + (xp - 1) : 0;
    float ov[8], dv[8];
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
      const size_t gi = (inside && chok[cc]) ? g0 + (size_t)cc * HW + pix : 0;
      ov[cc] = p.out_act[gi];
      dv[cc] = p.dout[gi];
    }
    const float n = (inside && p.e.noise) ? p.e.noise[(size_t)b * HW + pix] : 0.f;
    float v[8];
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) {
      const bool ok = inside && chok[cc];
      const bool pos = ov[cc] > 0.f;
      const float dp = ok ? dv[cc] * gin * (pos ? g_pos : g_neg) : 0.f;
      const float pre = ov[cc] * (pos ? ig_pos : ig_neg);
      s_db[cc] += dp;
      s_dn[cc] += dp * n;
      s_dy[cc] += ok ? dp * (pre - n * str - bias[cc]) : 0.f;
      v[cc] = dp * sc[cc];
      if (p.dpre_out && ok) p.dpre_out[g0 + (size_t)cc * HW + pix] = dp;
    }
    const size_t u = (size_t)yp * Wp + xp;
    if constexpr (NP == 3) {
      bf16x8 h, m, l;
      split3_bf16x8(v, h, m, l);
      Ub[u] = h; Ub[p.plane + u] = m; Ub[2 * p.plane + u] = l;
    } else {
      Ub[u] = pack_bf16x8(v);
    }
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  {  // the 24 wave sums together (31 + 1 lane exchanges; 24 chains of 6 before)
    float t[32];
#pragma unroll
    for (int cc = 0; cc < 8; ++cc) { t[cc] = s_db[cc]; t[8 + cc] = s_dn[cc]; t[16 + cc] = s_dy[cc]; t[24 + cc] = 0.f; }
    wave_tree_sum<32, 32, 32>(t, lane);
    const int row = wave_tree_row<32>(lane);
    if ((lane & 1) == 0 && row < 24) red[wave][row] = t[0];
  }
  __syncthreads();
  if (threadIdx.x < 24) {
    const int kind = threadIdx.x >> 3, cc = threadIdx.x & 7, m = cu * 8 + cc;
    float *dst = kind == 0 ? p.part_db : kind == 1 ? p.part_dn : p.part_dyy;
    if (dst && m < p.M)
      dst[((size_t)b * p.M + m) * p.nchunks + blockIdx.y] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
  }
}

extern "C" int tbg_bias_act_bwd_units(const float *dout, const float *out_act, void *U, int planes, float *dpre_out,
                                      float *part_db, float *part_dn, float *part_dyy, int B, int M, int H, int W,
                                      const tbg_epilogue *epi, void *stream) {
  if (!dout || !out_act || !U || B < 1 || M < 1 || H < 1 || W < 1 || !epi || !epi_valid(epi) || (planes != 1 && planes != 3))
    return TBG_EINVAL;
  if ((reinterpret_cast<uintptr_t>(U) & 15) != 0) return TBG_EINVAL;
  if (part_dn && !epi->noise) return TBG_EINVAL;
  if (epi->gate || epi->units_out) return TBG_EINVAL;  // forward-only epilogue terms
  const long long plane = units_per_plane(B, M, H, W);
  if (plane * 8 > 2147483647LL || (long long)B * M * H * W > 2147483647LL) return TBG_ERANGE;
  BabUnitsP p{dout, out_act, reinterpret_cast<bf16x8 *>(U), dpre_out, part_db, part_dn, part_dyy, B, M, H, W,
              tbg_bias_act_bwd_units_chunks(H), plane, make_epi(epi)};
  const dim3 grid(B * ((M + 7) / 8), p.nchunks);
  if (planes == 3) hipLaunchKernelGGL(bias_act_bwd_units_kernel<3>, grid, dim3(256), 0, tbg_stream(stream), p);
  else hipLaunchKernelGGL(bias_act_bwd_units_kernel<1>, grid, dim3(256), 0, tbg_stream(stream), p);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

// ============================================================================================
// filter gradient from unit tensors (3x3, stride 1, pad 1)
// ============================================================================================
// dW[t][cl][cs] = sum_{b,u,v} S[b,cs,u,v] * L[b,cl,u-1+kh,v-1+kw] with S, L given as unit tensors (their scales / splits already
// inside).  Block = 64 S-channels x 64 L-channels, wave = 32 x 32 with the 9 taps as 9 accumulators (144 registers), K = pixels in
// chunks of 2 rows x 32 columns -- the block / wave / partial-tile structure of conv_wgrad_x3_kernel, so the reduce kernels are
// shared.  What differs: the chunk's tiles (S: 8 units x 64 pixels, L: 8 units x 4 x 34 halo positions, per plane) arrive by
// LDS-DMA into one of TWO buffers (the NCHW kernel's split + store pass, ~3400 cycles per chunk with one wave per SIMD, is gone:
// the DMA of chunk k+1 is issued between the MFMAs of chunk k), and the MFMA operands are read with ds_read_b64_tr_b16.
//
// LDS image of one buffer, in 16-byte units:  S rows [plane][8][S_ROW = 68]  then  L rows [plane][8][L_ROW = 140]
// (64 / 136 used: the row pitches put the four channel units a transposing read touches on disjoint banks).  The whole
// buffer is ONE lane-linear DMA target of BUF/64 pieces; which global unit a lane fetches for a piece does not depend on the
// chunk (only a wave-uniform base does), so the per-lane source offsets are computed once.
struct WgUnitsP {
  const char *SU, *LU;
  long long s_plane, l_plane;  // units per plane
  int CS8, CL8, Hps, Wps, Hpl, Wpl;
  int tilesU, tilesV, nchunks, ksplit;
  float *ws;
  int tx, ty;  // channel tiles (S, L); the launch is 1-D: tx * ty * ksplit blocks
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int NP>
__global__ __launch_bounds__(256, 1) void conv_wgrad_units_kernel(const WgUnitsP p) {
  constexpr int S_ROW = 68, L_ROW = 140, NT = 9;
  constexpr int S_UNITS = NP * 8 * S_ROW, L_UNITS = NP * 8 * L_ROW, BUF = S_UNITS + L_UNITS;
  static_assert(BUF % 64 == 0, "the buffer is a whole number of 64-unit DMA pieces");
  constexpr int NPIECE = BUF / 64, PPW = (NPIECE + 3) / 4;
  constexpr int NQ = NP == 3 ? 6 : 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][BUF] units
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ws = wave >> 1, wl = wave & 1;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;

  // ---- DMA descriptors of this wave's pieces (piece q = wave + 4 k): per-lane source ADDRESS relative to the chunk base of its
  // tensor (held in registers: nothing about a piece is re-derived from kernel arguments inside the K loop), and which tensor.
  // Pad slots of a row are never read: they fetch the row's last unit.
  const char *dsrc[PPW];
  unsigned isl = 0;
  const char *const su_ = p.SU, *const lu_ = p.LU;
#pragma unroll
  for (int k = 0; k < PPW; ++k) {
    const int n = min(wave + 4 * k, NPIECE - 1) * 64 + lane;
    if (n < S_UNITS) {
      const int row = n / S_ROW, pix = min(n - row * S_ROW, 63);
      const int pl = row >> 3, su = row & 7;
      dsrc[k] = su_ + ((pl * p.s_plane + (long long)(su * p.Hps + (pix >> 5)) * p.Wps + (pix & 31)) << 4);
    } else {
      const int m = n - S_UNITS;
      const int row = m / L_ROW, pos = min(m - row * L_ROW, 135);
      const int pl = row >> 3, lu = row & 7;
      const int r = pos / 34, c = pos - r * 34;
      dsrc[k] = lu_ + ((pl * p.l_plane + (long long)(lu * p.Hpl + r) * p.Wpl + c) << 4);
      isl |= 1u << k;
    }
  }
  // block -> (S tile, L tile, K slice).  The hardware deals consecutive blocks round-robin over the 8 XCDs, and the tx * ty blocks of
  // one K slice read the SAME pixel chunks (each S tile ty times, each L tile tx times): numbered consecutively they would sit on
  // tx * ty different XCDs and every re-read would go to memory (2 x 2 tiles: 1.87 x the algorithmic bytes, profiles/r06_z_pmc_report).
  // With whole groups of 8 slices the slices are dealt over the XCDs instead and a slice's blocks share one L2.
  const int tiles = p.tx * p.ty;
  int bt, bz;
#ifndef WGU_NO_XCD_MAP
  if ((p.ksplit & 7) == 0) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    bt = slot % tiles;
    bz = (slot / tiles) * 8 + xcd;
  } else
#endif
  {
    bt = blockIdx.x % tiles;
    bz = blockIdx.x / tiles;
  }
  const int bx = bt % p.tx, by = bt / p.tx;
  const int cs8 = bx * 8, cl8 = by * 8;

  auto chunk_bases = [&](int chunk, int &sb, int &lb) {
    const int tv = chunk % p.tilesV;
    const int t2 = chunk / p.tilesV;
    const int tu = t2 % p.tilesU, b = t2 / p.tilesU;
    const int u0 = tu * 2, v0 = tv * 32;
    sb = ((b * p.CS8 + cs8) * p.Hps + u0 + 1) * p.Wps + v0 + 1;  // interior starts at (1, 1)
    lb = ((b * p.CL8 + cl8) * p.Hpl + u0) * p.Wpl + v0;          // halo: y = u0 - 1 -> padded row u0
  };
  // (branch-free: a slot past the buffer re-issues the last piece -- same bytes to the same place)
  auto issue_piece = [&](int k, int sb, int lb, int buf) {
    const int q = min(__builtin_amdgcn_readfirstlane(wave) + 4 * k, NPIECE - 1);
    const int cb = ((isl >> k) & 1u) ? lb : sb;  // (one piece straddles the S / L regions: per-lane select)
    dma16(dsrc[k] + ((long long)cb << 4), lds0 + (unsigned)((buf * BUF + q * 64) * 16));
  };

  // ---- operand addressing.  A transposing read serves 16 lanes with a [4 pixels][16 channels] block: lane i of the group
  // supplies the address of (pixel j = i / 4, channels 4 (i % 4) .. + 3) = half a unit, and receives channel i at the 4 pixels.
  // Groups 0 / 1 = channels 0-15 / 16-31 of the wave's 32 at K half 0, groups 2 / 3 the same at K half 1 (pixels + 8): two
  // reads (pixels +0..3, +4..7) make one v_mfma_f32_32x32x16_bf16 operand.
  const int i16 = lane & 15, grp = lane >> 4;
  const int jj = i16 >> 2, cq = i16 & 3, chblk = grp & 1, khalf = grp >> 1;
  const int a_lane = ((ws * 4 + chblk * 2 + (cq >> 1)) * S_ROW + 8 * khalf + jj) * 16 + (cq & 1) * 8;
  const int b_lane = S_UNITS * 16 + ((wl * 4 + chblk * 2 + (cq >> 1)) * L_ROW + 8 * khalf + jj) * 16 + (cq & 1) * 8;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  auto rd = [&](const char *ptr) -> bf16x8 {  // 8 consecutive pixels of this lane's channel
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(ptr));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(ptr + 64));
    return __builtin_bit_cast(bf16x8, s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
  };

  int chunk = bz;
  int sb = 0, lb = 0;
  if (chunk < p.nchunks) {
    chunk_bases(chunk, sb, lb);
#pragma unroll
    for (int k = 0; k < PPW; ++k) issue_piece(k, sb, lb, 0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int buf = 0;
  for (; chunk < p.nchunks; chunk += p.ksplit) {
    // tiles of the next chunk -> the other buffer (the last chunk re-fetches itself: no branch in the loop body)
    chunk_bases(chunk + p.ksplit < p.nchunks ? chunk + p.ksplit : chunk, sb, lb);
    const char *Ab = smem + (size_t)buf * BUF * 16 + a_lane;
    const char *Bb = smem + (size_t)buf * BUF * 16 + b_lane;
    // 12 steps of (16-pixel group g, filter row kh), 6 x 3 MFMAs each.  The instruction stream is laid out by hand and pinned
    // (one sched_barrier per MFMA): with ONE wave per SIMD nothing else fills the matrix pipe while this wave issues anything
    // else, so every LDS read of step i+1 (two register sets) and every DMA piece of the next chunk sits in the shadow of an
    // MFMA of step i -- never in a block of its own between two steps.
    constexpr int NST = 12, NM = NQ * 3;
    bf16x8 a[2][NP], bv[2][NP][3];
    auto ld1 = [&](int st, int bs, int idx) {  // load idx of step st's operand set: the 3 NP B operands, then (kh == 0) the NP A operands
      const int g = st / 3, kh = st - 3 * g;
      if (idx < 3 * NP) {
        const int pl = idx / 3, kw = idx - 3 * pl;
        bv[bs][pl][kw] = rd(Bb + (pl * 8 * L_ROW + ((g >> 1) + kh) * 34 + (g & 1) * 16 + kw) * 16);
      } else if (kh == 0 && idx < 4 * NP) {
        const int pl = idx - 3 * NP;
        a[g & 1][pl] = rd(Ab + (pl * 8 * S_ROW + 16 * g) * 16);
      }
    };
#pragma unroll
    for (int idx = 0; idx < 4 * NP; ++idx) ld1(0, 0, idx);
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      const int bs = st & 1, g = st / 3, kh = st - 3 * g;
      __builtin_amdgcn_sched_barrier(0);
      // six partial products per tap, smallest first: (hi,lo) (lo,hi) (mid,mid) (hi,mid) (mid,hi) (hi,hi)
      constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        const int q = m / 3, kw = m - 3 * q;
        acc[3 * kh + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[g & 1][NP == 3 ? PA[q] : 0], bv[bs][NP == 3 ? PB[q] : 0][kw],
                                                                   acc[3 * kh + kw], 0, 0, 0);
        // behind this MFMA: one operand load of the next step (NM >= 4 NP slots only in f32x3; bf16 doubles up) ...
        if (st + 1 < NST) {
          constexpr int LPM = (4 * NP + NM - 1) / NM;
#pragma unroll
          for (int e = 0; e < LPM; ++e) ld1(st + 1, bs ^ 1, m * LPM + e);
        }
        // ... and the next chunk's DMA pieces in the FIRST HALF of the phase, evenly spaced, so that they have landed when it ends
        {
          constexpr int STRIDE = (NST * NM / 2) / PPW > 0 ? (NST * NM / 2) / PPW : 1;
          const int slot = st * NM + m;
          if (slot % STRIDE == 0 && slot / STRIDE < PPW) issue_piece(slot / STRIDE, sb, lb, buf ^ 1);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the next chunk's tiles have landed
    __syncthreads();                                   // ... for every wave, and every wave is done with this buffer
    buf ^= 1;
  }

  const size_t blk = ((size_t)bz * p.ty + by) * p.tx + bx;
  float *wsp = p.ws + blk * (size_t)(NT * 16 * 256);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r16 = 0; r16 < 16; ++r16) wsp[(t * 16 + r16) * 256 + tid] = acc[t][r16];
}

// geometry the unit kernel takes: 3x3, stride 1, pad 1, whole 2 x 32-pixel chunks, whole 64-channel tiles
static bool wgrad_units_ok(const tbg_wgrad_desc *d) {
  return d->KH == 3 && d->KW == 3 && d->sy == 1 && d->sx == 1 && d->py == 1 && d->px == 1 && d->Hl == d->Hs && d->Wl == d->Ws &&
         (d->Ws % 32) == 0 && (d->Hs % 2) == 0 && (d->CS % 64) == 0 && (d->CL % 64) == 0;
}

static int wgrad_units_ksplit(const tbg_wgrad_desc *d) {
  const int tiles = (d->CS / 64) * (d->CL / 64);
  const int nchunks = d->B * (d->Hs / 2) * (d->Ws / 32);
  return wgrad_ksplit(tiles, nchunks, 256);  // one block per CU (two 78 KB LDS buffers)
}

extern "C" long long tbg_conv2d_wgrad_units_workspace_bytes(const tbg_wgrad_desc *d) {
  if (!d || d->B < 1 || d->CS < 1 || d->CL < 1 || d->Hs < 1 || d->Ws < 1) return TBG_EINVAL;
  if (!wgrad_units_ok(d)) return TBG_EUNSUPPORTED;
  return (long long)wgrad_units_ksplit(d) * (d->CS / 64) * (d->CL / 64) * 9 * 16 * 256 * (long long)sizeof(float);
}

template <int NP>
static int launch_wgrad_units(WgUnitsP &u, WgradP &p, hipStream_t st, const char **name) {
  if (name) { *name = NP == 3 ? "conv_wgrad_units_kernel<3>" : "conv_wgrad_units_kernel<1>"; return TBG_OK; }
  constexpr int BUF = NP * 8 * (68 + 140);
  const size_t lds = (size_t)2 * BUF * 16;
  auto kern = conv_wgrad_units_kernel<NP>;
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return TBG_EHIP;
  const int tx = p.CS / 64, ty = p.CL / 64;
  u.tx = tx; u.ty = ty;
  hipLaunchKernelGGL(kern, dim3(tx * ty * u.ksplit), dim3(256), lds, st, u);
  TBG_LAUNCH_CHECK();
  if (tx * ty * 9 >= 256)
    hipLaunchKernelGGL((conv_wgrad_reduce_kernel<2, 2, 9>), dim3(tx, ty, 9), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL((conv_wgrad_reduce_wide_kernel<2, 2, 9>), dim3(tx, ty, 9 * 16), dim3(256), 0, st, p);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

extern "C" int tbg_conv2d_wgrad_units(const tbg_wgrad_desc *d, const void *SU, const void *LU, int planes, float *dW,
                                      const float *addw, const float *addq, float gamma, float *workspace,
                                      long long workspace_bytes, void *stream) {
  if (!d || !SU || !LU || !dW || !workspace || ((addw == nullptr) != (addq == nullptr)) || (planes != 1 && planes != 3))
    return TBG_EINVAL;
  if (d->B < 1 || d->CS < 1 || d->CL < 1 || d->Hs < 1 || d->Ws < 1 || d->Hl < 1 || d->Wl < 1) return TBG_EINVAL;
  if (((reinterpret_cast<uintptr_t>(SU) | reinterpret_cast<uintptr_t>(LU)) & 15) != 0) return TBG_EINVAL;
  if (!wgrad_units_ok(d)) return TBG_EUNSUPPORTED;
  const long long s_plane = units_per_plane(d->B, d->CS, d->Hs, d->Ws), l_plane = units_per_plane(d->B, d->CL, d->Hl, d->Wl);
  if (s_plane * planes > 2147483647LL / 2 || l_plane * planes > 2147483647LL / 2) return TBG_ERANGE;
  WgUnitsP u{};
  u.SU = reinterpret_cast<const char *>(SU); u.LU = reinterpret_cast<const char *>(LU);
  u.s_plane = s_plane; u.l_plane = l_plane;
  u.CS8 = d->CS / 8; u.CL8 = d->CL / 8;
  u.Hps = d->Hs + 2; u.Wps = d->Ws + 2; u.Hpl = d->Hl + 2; u.Wpl = d->Wl + 2;
  u.tilesU = d->Hs / 2; u.tilesV = d->Ws / 32;
  u.nchunks = d->B * u.tilesU * u.tilesV;
  u.ksplit = wgrad_units_ksplit(d);
  u.ws = workspace;
  if ((long long)u.ksplit * (d->CS / 64) * (d->CL / 64) * 9 * 16 * 256 * (long long)sizeof(float) > workspace_bytes) return TBG_EINVAL;
  WgradP p{};  // what the reduce kernels read
  p.CS = d->CS; p.CL = d->CL; p.st_t = d->st_t; p.st_l = d->st_l; p.st_s = d->st_s; p.alpha = d->alpha;
  p.dW = dW; p.ws = workspace; p.addw = addw; p.addq = addq; p.gamma = gamma; p.ksplit = u.ksplit;
  if (const int rcb = wgrad_bias_rider(p, d)) return rcb;
  return planes == 3 ? launch_wgrad_units<3>(u, p, tbg_stream(stream), nullptr) : launch_wgrad_units<1>(u, p, tbg_stream(stream), nullptr);
}

// ============================================================================================
// forward / data-gradient convolution from a unit tensor (3x3, stride 1, pad 1)
// ============================================================================================
// y[b,m,Y,X] = epilogue( sum_{t=(kh,kw), c} XU[b,c,Y-1+kh,X-1+kw] * Wp[t'][c][m] ): the implicit GEMM of conv_fprop_kernel (M =
// output channels, N = pixels, K = taps x channels, packed filter tbg_weight_pack_x3 / _bf16, same MFMA term pairing and the same
// fused epilogue), with EVERY operand byte arriving by LDS-DMA: the filter slice as before, and the halo tile straight from the
// unit tensor (the x * s modulation, the split / rounding and the zero padding were paid once by the producer) -- no staging
// registers, no VALU pass, no clamps.  That frees the kernel to be ONE 512-thread block per CU (8 waves, 2 per SIMD; tile =
// 64 WTM output channels x 8 rows x 32 pixels) with BOTH tiles double-buffered in LDS (x3: 2 x 70 KB): the DMA of chunk k+1 is
// issued piece by piece behind the MFMAs of chunk k, one barrier per chunk, and the two waves of a SIMD cover each other's
// operand reads.  (conv_fprop_kernel runs two 256-thread blocks per CU with single buffers: its filter DMA wait and its split +
// store pass are exposed, 25 % of the large layers -- DESIGN 4.1b.)
struct ConvUnitsP {
  const char *XU, *Wf;
  long long x_plane, w_plane;  // 16-byte units per plane
  float *y;
  int B, C8, M, H, W, ldw;
  int tilesU, tilesV, dot_slots;
  int wtap[9];
  EpiK e;
};

#if TBG_EXP == 3
#define FPROP_SCHED_BARRIER()
#else
#define FPROP_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#endif
template <int NP, int WTM, int OPT>  // OPT: conv_epilogue's optional operand paths (conv_common.h)
__global__ __launch_bounds__(512, 2) void conv_units_fprop_kernel(const ConvUnitsP p) {
  constexpr int WGN = 4, WTN = 2, BM = 2 * WTM * 32;
  constexpr int CKU = NP == 3 ? 1 : 2;       // channel units per chunk (x3: 8 channels x 3 planes; bf16: 16 channels)
  constexpr int HALO = 10 * 34;
  constexpr int A_UNITS = NP * 9 * CKU * BM, X_UNITS = NP * CKU * HALO;
  constexpr int NPIECE = (A_UNITS + X_UNITS + 63) / 64, BUF = NPIECE * 64, PPW = (NPIECE + 7) / 8;
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][BUF] units
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3, half = lane >> 5;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)smem;

  // block -> (pixel tile, channel tile).  The hardware deals consecutive blocks round-robin over the 8 XCDs; dealt in tile order,
  // the tiles that share halo rows / columns and the channel tiles that read the SAME pixel tile would sit in 8 different L2s and
  // every shared unit would come from memory again.  Each XCD takes one contiguous eighth of the (pixel tile, channel tile) list
  // instead (channel tiles of a pixel tile adjacent, then the tiles of a row, then the rows of an image).
  const int mtiles = p.M / BM, total = gridDim.x;
  int lin = blockIdx.x;
#ifndef CU_NO_XCD_MAP
  if ((total & 7) == 0) lin = (blockIdx.x & 7) * (total >> 3) + (blockIdx.x >> 3);
#endif
  const int tn = lin / mtiles, mt = lin - tn * mtiles;
  const int tv = tn % p.tilesV, t2 = tn / p.tilesV;
  const int tu = t2 % p.tilesU, b = t2 / p.tilesU;
  const int m0 = mt * BM, y0 = tu * 8, x0 = tv * 32;
  const int Hp = p.H + 2, Wp = p.W + 2;

  // ---- DMA descriptors of this wave's pieces (piece q = wave + 8 k): the per-lane source ADDRESS for chunk 0 (registers:
  // the descriptor must not be re-derived from kernel arguments inside the K loop -- a vector load there puts an
  // s_waitcnt vmcnt(0) in front of the next piece, i.e. behind every piece still in flight).  The filter region is a whole
  // number of pieces, so a piece belongs to one tensor and its per-chunk address step is wave-uniform.  Slots past the halo
  // region (the last piece's tail) re-fetch the halo's last unit.
  static_assert(A_UNITS % 64 == 0, "a DMA piece never straddles the filter / halo regions");
  const char *dsrc[PPW];
  const char *const xu = p.XU, *const wf = p.Wf;
#pragma unroll
  for (int k = 0; k < PPW; ++k) {
    const int q = min(wave + 8 * k, NPIECE - 1);  // (a slot past the buffer re-issues the wave's previous piece ... or the last)
    const int n = q * 64 + lane;
    if (n < A_UNITS) {  // As[plane][tap][unit][BM]
      const int row = n / BM, m = n - row * BM;
      const int u = row % CKU, t = (row / CKU) % 9, pl = row / (9 * CKU);
      dsrc[k] = wf + ((pl * p.w_plane + (long long)(p.wtap[t] * p.C8 + u) * p.ldw + m0 + m) << 4);
    } else {            // Xs[plane][unit][10][34]
      const int m2 = min(n - A_UNITS, X_UNITS - 1);
      const int row = m2 / HALO, pos = m2 - row * HALO;
      const int u = row % CKU, pl = row / CKU;
      const int r = pos / 34, c = pos - r * 34;
      dsrc[k] = xu + ((pl * p.x_plane + (long long)((b * p.C8 + u) * Hp + y0 + r) * Wp + x0 + c) << 4);
    }
  }
  const long long a_step = (long long)CKU * p.ldw * 16, x_step = (long long)CKU * Hp * Wp * 16;  // bytes per chunk
  auto issue_piece = [&](int k, int kc, int buf) {
    const int q = min(__builtin_amdgcn_readfirstlane(wave) + 8 * k, NPIECE - 1);
    const long long step = q * 64 < A_UNITS ? a_step : x_step;  // wave-uniform
    dma16(dsrc[k] + kc * step, lds0 + (unsigned)((buf * BUF + q * 64) * 16));
  };

  // ---- operand addressing (bytes inside a buffer)
  const int a_lane = (wm * (WTM * 32) + (lane & 31)) * 16;
  int b_lane[WTN];
#pragma unroll
  for (int j = 0; j < WTN; ++j) b_lane[j] = A_UNITS * 16 + ((wn * WTN + j) * 34 + (lane & 31)) * 16;
  // x3: half-wave h supplies K half h of each MFMA:  A (hi | mid) x B hi,  A (hi | mid) x B mid,  A (hi | lo) x B (lo | hi)
  // bf16: half-wave h holds channel unit h of the chunk
  constexpr int A_PL = 9 * CKU * BM * 16, X_PL = CKU * HALO * 16;
  const int aX = NP == 3 ? half * A_PL : half * BM * 16, aY = 2 * half * A_PL;
  const int bZ = 2 * (1 - half) * X_PL, bH = NP == 3 ? 0 : half * HALO * 16;

  f32x16 acc[WTM][WTN];
#pragma unroll
  for (int i = 0; i < WTM; ++i)
#pragma unroll
    for (int j = 0; j < WTN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

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
    constexpr int NA = NP == 3 ? 2 * WTM : WTM, NB = NP == 3 ? 3 * WTN : WTN, NL = NA + NB;
    constexpr int NM = (NP == 3 ? 3 : 1) * WTM * WTN;  // MFMAs per tap
    bf16x8 av[2][NA], bw[2][NB];
    auto ld1 = [&](int t, int bs, int idx) {  // operand idx of tap t: the A tiles (x3: aX then aY per tile), then the B tiles
      const int kh = t / 3, kw = t - 3 * kh;
      if (idx < NA) {
        const int i = NP == 3 ? idx >> 1 : idx;
        const int plane_off = NP == 3 ? ((idx & 1) ? aY : aX) : aX;
        av[bs][idx] = *reinterpret_cast<const bf16x8 *>(Ab + t * CKU * BM * 16 + plane_off + i * 32 * 16);
      } else if (idx < NL) {
        const int e = idx - NA;
        const int j = NP == 3 ? e / 3 : e, w = NP == 3 ? e - 3 * j : 0;
        const int plane_off = NP == 3 ? (w == 0 ? 0 : w == 1 ? X_PL : bZ) : bH;
        bw[bs][e] = *reinterpret_cast<const bf16x8 *>(Xb + b_lane[j] + (kh * 34 + kw) * 16 + plane_off);
      }
    };
#pragma unroll
    for (int idx = 0; idx < NL; ++idx) ld1(0, 0, idx);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int bs = TBG_EXP == 4 ? 0 : (t & 1);
      FPROP_SCHED_BARRIER();
#pragma unroll
      for (int mm = 0; mm < NM; ++mm) {
        // x3, smallest terms first: (hi|lo)x(lo|hi), then (hi|mid) x mid, then (hi|mid) x hi
        const int grp = mm / (WTM * WTN), ij = mm - grp * (WTM * WTN);
        const int i = ij / WTN, j = ij - i * WTN;
        if constexpr (NP == 3) {
          const int ai = 2 * i + (grp == 0 ? 1 : 0), bi = 3 * j + (grp == 0 ? 2 : grp == 1 ? 1 : 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[bs][ai], bw[bs][bi], acc[i][j], 0, 0, 0);
        } else {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[bs][i], bw[bs][j], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < 9 && TBG_EXP != 4) {  // behind this MFMA: operand loads of the next tap
          constexpr int LPM = (NL + NM - 1) / NM;
#pragma unroll
          for (int e = 0; e < LPM; ++e) ld1(t + 1, bs ^ 1, mm * LPM + e);
        }
        // the next chunk's DMA pieces go out in the FIRST HALF of the phase, evenly spaced behind MFMAs, so that they have landed
        // when the phase ends: a piece issued under the last tap would be waited for in full at the barrier
        {
          constexpr int STRIDE = (9 * NM / 2) / PPW > 0 ? (9 * NM / 2) / PPW : 1;
          const int slot = t * NM + mm;
          if (TBG_EXP != 2 && slot % STRIDE == 0 && slot / STRIDE < PPW) issue_piece(slot / STRIDE, kn, buf ^ 1);
        }
        FPROP_SCHED_BARRIER();
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    buf ^= 1;
  }

  int e_pix[WTN], e_b[WTN];
#pragma unroll
  for (int j = 0; j < WTN; ++j) {
    e_pix[j] = (y0 + wn * WTN + j) * p.W + x0 + (lane & 31);
    e_b[j] = b;
  }
  if (TBG_EXP == 1) {  // ablation: no epilogue
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
      for (int j = 0; j < WTN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    if (sum == 123.f) p.y[0] = sum;
    return;
  }
  conv_epilogue<WTM, WTN, 4, true, 16, OPT, true, !(NP == 1 && WTM == 1)>(acc, p.e, p.y, nullptr, p.M, p.H * p.W, m0 + wm * WTM * 32, lane, e_pix, e_b, true, b, p.dot_slots,
                             (tu * p.tilesV + tv) * WGN + wn, p.H, p.W);
}

static bool conv_units_ok(const tbg_conv_desc *d, int planes) {
  const int cku = planes == 3 ? 8 : 16;
  return !d->transposed && d->KH == 3 && d->KW == 3 && d->sy == 1 && d->sx == 1 && d->py == 1 && d->px == 1 &&
         d->Hout == d->Hin && d->Wout == d->Win && (d->Hin % 8) == 0 && (d->Win % 32) == 0 && (d->C % cku) == 0 &&
         (d->M % 64) == 0 && d->ksplit == 1 && d->ldw >= d->M;
}

// tile height in output channels: 128 (WTM = 2) where the layer has them, unless that leaves half the chip idle (ONE block per CU:
// fewer than UNITS_FULL blocks) while 64-channel tiles fill it -- the 256-channel 16x64 layers at B = 16: 128 -> 256 blocks
#ifndef UNITS_FULL  // (-DUNITS_FULL=0: always the 128-channel tile -- tools/ab_lib.py)
#define UNITS_FULL 200
#endif
static int units_wtm(const tbg_conv_desc *d) {
  if (d->M % 128 != 0) return 1;
  const long long b2 = (long long)d->B * (d->Hin / 8) * (d->Win / 32) * (d->M / 128);
  return (b2 < UNITS_FULL && 2 * b2 >= UNITS_FULL) ? 1 : 2;
}

extern "C" int tbg_conv2d_units_blocks(const tbg_conv_desc *d, int planes) {
  if (!d || (planes != 1 && planes != 3)) return TBG_EINVAL;
  if (!conv_units_ok(d, planes)) return TBG_EUNSUPPORTED;
  const long long n = (long long)d->B * (d->Hin / 8) * (d->Win / 32) * (d->M / (64 * units_wtm(d)));
  return n > 2147483647LL ? TBG_ERANGE : (int)n;
}

extern "C" int tbg_conv2d_units_tile_channels(const tbg_conv_desc *d, int planes) {
  if (!d || (planes != 1 && planes != 3)) return TBG_EINVAL;
  if (!conv_units_ok(d, planes)) return TBG_EUNSUPPORTED;
  return 64 * units_wtm(d);
}

extern "C" int tbg_conv2d_units_dot_slots(const tbg_conv_desc *d, int planes) {
  if (!d || (planes != 1 && planes != 3)) return TBG_EINVAL;
  if (!conv_units_ok(d, planes)) return TBG_EUNSUPPORTED;
  return (d->Hin / 8) * (d->Win / 32) * 4;
}

template <int NP, int WTM, int OPT>
static int launch_conv_units_opt(ConvUnitsP &p, hipStream_t st) {
  constexpr int BM = 2 * WTM * 32, CKU = NP == 3 ? 1 : 2;
  constexpr int NPIECE = (NP * 9 * CKU * BM + NP * CKU * 340 + 63) / 64;
  const size_t lds = (size_t)2 * NPIECE * 64 * 16;
  auto kern = conv_units_fprop_kernel<NP, WTM, OPT>;
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return TBG_EHIP;
  hipLaunchKernelGGL(kern, dim3(p.B * p.tilesU * p.tilesV * (p.M / BM)), dim3(512), lds, st, p);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

template <int NP, int WTM>
static int launch_conv_units(ConvUnitsP &p, hipStream_t st) {
  switch (epi_opt(p.e)) {
    case 0: return launch_conv_units_opt<NP, WTM, 0>(p, st);
    case 1: return launch_conv_units_opt<NP, WTM, 1>(p, st);
    case 2: return launch_conv_units_opt<NP, WTM, 2>(p, st);
    default: return launch_conv_units_opt<NP, WTM, 3>(p, st);
  }
}

extern "C" int tbg_conv2d_units(const tbg_conv_desc *d, const void *XU, int planes, const void *w, float *y,
                                const tbg_epilogue *epi, void *stream) {
  if (!d || !XU || !w || (!y && !epi_has_sink(epi)) || (planes != 1 && planes != 3) || !epi_valid(epi)) return TBG_EINVAL;
  if (d->B < 1 || d->C < 1 || d->M < 1 || d->Hin < 1 || d->Win < 1) return TBG_EINVAL;
  if (((reinterpret_cast<uintptr_t>(XU) | reinterpret_cast<uintptr_t>(w)) & 15) != 0) return TBG_EINVAL;
  if (!conv_units_ok(d, planes)) return TBG_EUNSUPPORTED;
  if ((double)d->B * d->M * d->Hout * d->Wout > 2147483647.0) return TBG_ERANGE;
  ConvUnitsP p{};
  p.XU = reinterpret_cast<const char *>(XU); p.Wf = reinterpret_cast<const char *>(w);
  p.x_plane = units_per_plane(d->B, d->C, d->Hin, d->Win);
  p.C8 = d->C / 8;
  p.w_plane = (long long)9 * p.C8 * d->ldw;
  if (p.x_plane * planes > 2147483647LL / 2 || p.w_plane * planes > 2147483647LL / 2) return TBG_ERANGE;
  p.y = y; p.B = d->B; p.M = d->M; p.H = d->Hin; p.W = d->Win; p.ldw = d->ldw;
  p.tilesU = d->Hin / 8; p.tilesV = d->Win / 32;
  p.dot_slots = p.tilesU * p.tilesV * 4;
  for (int t = 0; t < 9; ++t) p.wtap[t] = d->flip ? 8 - t : t;
  p.e = make_epi(epi);
  if (const int rcs = epi_sink_geometry(p.e, d->B, d->M, d->Hin, d->Win)) return rcs;
  hipStream_t st = tbg_stream(stream);
  if (units_wtm(d) == 2) return planes == 3 ? launch_conv_units<3, 2>(p, st) : launch_conv_units<1, 2>(p, st);
  return planes == 3 ? launch_conv_units<3, 1>(p, st) : launch_conv_units<1, 1>(p, st);
}
