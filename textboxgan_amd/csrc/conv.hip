// fp32 implicit-GEMM convolutions on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32: exact
// f32, 64 cycles / instruction / SIMD, 157 TFLOP/s chip peak).
//
// Design (MI355X-first, not a port of cuDNN calls):
//   * GEMM view  M = output channels, N = output pixels (batch folded in), K = taps x channels.
//   * One 256-thread block (4 wave64) owns a BM x BN tile; every wave owns WTM x WTN MFMA tiles
//     of 32x32 whose accumulators stay in registers for the whole K loop.
//   * K is walked in chunks of CK input channels.  Per chunk the block stages into LDS
//       As4[tap][quad][BM]  the PACKED filter slice (tbg_weight_pack_f32: 16-byte units of 4 consecutive reduction
//                           channels of one output channel) by direct global->LDS DMA (global_load_lds), and
//       Xs4[quad][position] ONE halo tile of the input (4 channels of a position = one ds_write_b128) -- the 9 taps
//                           re-read it at shifted LDS offsets (implicit im2col, no 9x expansion); the style
//                           modulation s[b,c] is multiplied in while staging; halo loads are branch-free.
//     MFMA operands: half-wave h reads channel quad 2o+h, so ONE ds_read_b128 per operand sub-tile feeds 4 k-steps.
//     Stride-2 inputs are de-interleaved (even | odd columns) on the way in so the strided taps stay unit-stride.
//   * Stride-2 TRANSPOSED convolution is decomposed into its sy*sx output-parity classes; each
//     class is a dense small-tap correlation on the class grid, so the same kernel runs it with
//     a per-class tap table and a strided output mapping (no zero-insertion, no atomics).
//   * Epilogue fused: runtime coef, demodulation d[b,m], noise, bias, LeakyReLU*sqrt2, residual, per-(b,m) dot.
//   * Small-spatial / wide-channel layers split K over channel chunks into SLABS (plain stores, no atomics, no
//     zero-fill); tbg_slab_epilogue_f32 sums the slabs and applies the real epilogue.
// Weight gradient: M = 32 S-channels, N = 32 L-channels per wave, 9 taps = 9 accumulators that share ONE staged halo
// tile of L; K = pixels; blocks split K and write partial tiles to a caller workspace that a second kernel sums
// (deterministic, no atomics).
// The dispatch is a pure function of the descriptor (no environment knobs, no mutable state): tbg_conv2d_kernel_name /
// tbg_conv2d_wgrad_kernel_name report the instantiation a descriptor selects.
#include <stdio.h>
#include <stdlib.h>

#include "conv_common.h"
#include <type_traits>

#define MAXCLS 4
#define MAXTAPS 9
#define MAXNJ 6

struct ClassInfo {
  int ntaps, KWc;
  int py, px;
  int ooy, oox;
  int Ug, Vg;
  int tilesU, tilesV;
  int wtap[MAXTAPS];
};

struct ConvP {
  const float *x, *w, *in_scale;
  float *y;
  int B, C, M, Hin, Win, Hout, Wout;
  int sy, sx, osy, osx;
  int ldw;
  int logTW, logTHs, NSEG, IHs, IWs, IWp, HALFW, planeStride, ppc, NJ, nBG;
  int nclass, ksplit, nchunks, cps;
  long long slab;  // elements per split-K slab (= B*M*Hout*Wout)
  int a_floats, ck_rt, dot_slots;
  int wplane;  // X3: floats between two planes (hi | mid | lo) of the packed filter
  ClassInfo cls[MAXCLS];
  EpiK e;
};


// BF = true: the bf16-in / fp32-accumulate form (BASELINE configs[2]).  Activations stay fp32 in HBM; they are rounded to
// bf16 (RNE) on their way into LDS, the filter arrives pre-packed in bf16 (tbg_weight_pack_bf16) and the contraction runs on
// v_mfma_f32_32x32x16_bf16 (16x the fp32 matrix rate).  A 16-byte LDS unit then holds 8 channels instead of 4, so the
// same tile geometry / DMA / operand-read code serves both: only the unit width KP and the MFMA differ.
// TM = true: MERGED stride-2 transposed 3x3 convolution (up-conv forward, data gradient of the stride-2 convs).  The four
// output-parity classes of y[2a+kh, 2b+kw] += x[a,b] w[kh,kw] are computed by ONE block from ONE staged halo tile: the tile
// indexes INPUT positions (u,v) (halo: one row above, one column left), the 9 taps are unrolled statically and tap (kh,kw)
// accumulates into class (kh&1, kw&1) -- 4 x (WTM x WTN) accumulator tiles -- reading x[u - (kh==2), v - (kw==2)].  Same MFMA
// count as a 3x3 stride-1 convolution on the input grid and no per-class re-staging (the class-per-block form staged the
// halo four times and ran 1/2/2/4-tap K loops).  Store-only epilogue (alpha * acc; split-K slabs allowed).
// X3 = true (implies BF): "f32x3" -- fp32 operands split into three bf16 terms each (split3_bf16x8) and contracted on the bf16
// pipe: per 8-channel unit THREE v_mfma_f32_32x32x16_bf16 whose two K halves carry different term pairs
//     half-wave 0 / 1:   A (hi | mid) x B hi,   A (hi | mid) x B mid,   A (hi | lo) x B (lo | hi)
// = the six largest of the nine partial products (hh, mh, hm, mm, hl, lh; the dropped ones are <= 2^-24 relative), fp32
// accumulate: fp32-grade results at 3 bf16 MFMAs per 8 channels = 96 matrix cycles instead of 256 for v_mfma_f32_32x32x2_f32.
// LDS holds three planes of the filter slice (packed hi | mid | lo by tbg_weight_pack_x3, DMA'd) and three of the halo tile
// (split while staging); 2 A reads + 3 B reads per sub-tile row / column feed the three MFMAs.
template <int WGM, int WGN, int WTM, int WTN, int CK, int MT, int PF, int OCC = 3, bool BF = false, bool TM = false, bool X3 = false>
__global__ __launch_bounds__(256, OCC) void conv_fprop_kernel(const ConvP p) {
  constexpr int BM = WGM * WTM * 32;
  constexpr int NC = TM ? 4 : 1;            // accumulator sets (output-parity classes of the merged transposed form)
  static_assert(!TM || (PF == 0 && CK >= 8 && MT == 9), "merged transposed form: 9 taps, plain K loop");
  static_assert(!X3 || BF, "the split-operand form runs on the bf16 pipe");
  constexpr int KP = BF ? 8 : 4;            // channels per 16-byte unit
  constexpr int NP = X3 ? 3 : 1;            // operand planes in LDS
  constexpr int G4 = (CK + KP - 1) / KP;    // units (fp32: channel quads, bf16: octets) per chunk
  constexpr int CB = CK < 8 ? CK : 8;       // channels per halo load batch
  constexpr int NB = PF > 0 ? 2 : 1;        // LDS buffers (PF > 0: software-pipelined K loop; PF < 0: register-prefetch loop)
  constexpr int NJC = PF > 0 ? PF : MAXNJ;  // halo positions per thread this instance can hold
  static_assert(X3 ? (CK % 8 == 0) : BF ? (CK % 16 == 0 && PF <= 0) : (CK == 4 || CK % 8 == 0),
                "chunk = one quad (half-waves split it) or whole unit pairs (half-wave h reads unit 2o+h); X3: whole units");
  static_assert(PF <= 0 || CK <= 8, "the pipelined variant prefetches one load batch");
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *As = smem;                                    // [NB][plane][tap][quad][BM][4]
  float *Xs = smem + NB * NP * p.a_floats;             // [NB][plane][quad][halo position][4]
  float *Ss = Xs + NB * NP * G4 * 4 * p.planeStride;   // [NSEG][C] style modulation of this block's images
  const int xbuf_floats = G4 * 4 * p.planeStride;      // one plane of one X buffer

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave - wm * WGN;
  const int cls = blockIdx.z / p.ksplit;
  const int ks = blockIdx.z - cls * p.ksplit;
  const ClassInfo &ci = p.cls[cls];

  const int tn = blockIdx.x;
  const int tv = tn % ci.tilesV;
  const int t2 = tn / ci.tilesV;
  const int tu = t2 % ci.tilesU;
  const int bg = t2 / ci.tilesU;
  if (bg >= p.nBG) return;
  const int m0 = blockIdx.y * BM;
  const int u0 = tu << p.logTHs, v0 = tv << p.logTW;
  const int ntaps = ci.ntaps;
  const int TWm = (1 << p.logTW) - 1, THm = (1 << p.logTHs) - 1;

  // ---- per-thread staging descriptors for the input halo tile (same for every chunk)
  int goff[NJC], loff[NJC], sseg[NJC];
#pragma unroll
  for (int j = 0; j < NJC; ++j) {
    goff[j] = -1; loff[j] = -1; sseg[j] = 0;
    const int e = tid + 256 * j;
    if (j < p.NJ && e < p.ppc) {
      const int per = p.IHs * p.IWs;
      const int seg = e / per;
      const int rem = e - seg * per;
      const int iyl = rem / p.IWs;
      const int ixl = rem - iyl * p.IWs;
      const int b = bg * p.NSEG + seg;
      const int iy = u0 * p.sy - ci.py + iyl;
      const int ix = v0 * p.sx - ci.px + ixl;
      const bool ok = b < p.B && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
      goff[j] = ok ? ((b * p.C) * p.Hin + iy) * p.Win + ix : -1;
      sseg[j] = seg * p.C;
      const int col = (p.sx == 2) ? (ixl & 1) * p.HALFW + (ixl >> 1) : ixl;
      loff[j] = (seg * p.IHs + iyl) * p.IWp + col;
    }
  }
  if (p.in_scale) {  // s[b, :] of the block's images -> LDS once (read back while storing the halo tile)
    for (int e = tid; e < p.NSEG * p.C; e += 256) {
      const int seg = e / p.C, b = bg * p.NSEG + seg;
      Ss[e] = b < p.B ? p.in_scale[(size_t)b * p.C + (e - seg * p.C)] : 0.f;
    }
  }

  // ---- per-lane pixel decode for the B operand / epilogue
  // MFMA 32x32x2: lanes 0-31 supply k = 0, lanes 32-63 k = 1.  CK >= 8: half-wave h reads channel quad 2*o + h, so one
  // 16-byte read per operand sub-tile feeds 4 k-steps; CK == 4: the half-waves split the single quad (8-byte reads, 2
  // k-steps).  The k order inside a chunk is free as long as A and B agree.
  const int half = lane >> 5;
  int bbytes[WTN];  // byte offset of this lane's B operand (tap 0) inside one X buffer
#pragma unroll
  for (int j = 0; j < WTN; ++j) {
    const int n = (wn * WTN + j) * 32 + (lane & 31);
    const int q = n & TWm, rr = n >> p.logTW;
    const int seg = rr >> p.logTHs, r = rr & THm;
    const int pos = (seg * p.IHs + r * p.sy) * p.IWp + q;
    bbytes[j] = X3 ? pos * 16 : CK == 4 ? pos * 16 + half * 8 : (half * p.planeStride + pos) * 16;
  }
  int tofft;  // lane t holds the LDS shift of tap t: fetched with v_readlane in the tap loop (no memory, no SALU chain)
  {
    const int khp = lane / ci.KWc, kwp = lane - khp * ci.KWc;
    tofft = (khp * p.IWp + (p.sx == 2 ? (kwp & 1) * p.HALFW + (kwp >> 1) : kwp)) * 16;
  }
  const int abytes = X3 ? (wm * (WTM * 32) + (lane & 31)) * 16
                      : CK == 4 ? (wm * (WTM * 32) + (lane & 31)) * 16 + half * 8
                                : (half * BM + wm * (WTM * 32) + (lane & 31)) * 16;

  f32x16 acc[NC][WTM][WTN];
#pragma unroll
  for (int c = 0; c < NC; ++c)
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
      for (int j = 0; j < WTN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][i][j][r] = 0.f;

  const int HWin = p.Hin * p.Win;
  const int kbeg = ks * p.cps;
  const int kend = min(kbeg + p.cps, p.nchunks);

  // filter-DMA addressing, hoisted out of the K loop.  The packed filter (tbg_weight_pack_f32) is
  // Wp[tap][C/4][ldw][4]: one 16-byte DMA unit = 4 consecutive input channels of one output channel, so the LDS image
  // As4[tap][quad][BM] is a linear copy and an MFMA A operand for 4 k-steps is ONE ds_read_b128.
  // The tap table lives in the kernarg segment: a scalar load per DMA piece cost ~500 cycles each (s_memtime), so the
  // per-pass tap base is hoisted (wave-uniform -> SGPRs).
  constexpr int UPT = BM * G4;  // DMA units per tap
  constexpr int PPT = UPT >= 256 ? UPT / 256 : 1, TPP = UPT >= 256 ? 1 : 256 / UPT;
  constexpr int A_IT_ = UPT >= 256 ? MT * PPT : (MT + TPP - 1) / TPP;
  static_assert(UPT >= 64 && (UPT >= 256 ? UPT % 256 == 0 : 256 % UPT == 0), "tile/chunk combination");
  const int C4 = (p.C + KP - 1) / KP;  // 16-byte units along the reduction
  const int t_local = UPT >= 256 ? 0 : __builtin_amdgcn_readfirstlane(tid / UPT);
  int tapbase[A_IT_];
#pragma unroll
  for (int it = 0; it < A_IT_; ++it) {
    const int t = UPT >= 256 ? it / PPT : it * TPP + t_local;
    tapbase[it] = (t < ntaps) ? ci.wtap[t] * C4 * p.ldw * 4 : 0;
  }

  // stage the filter slice of chunk kc with the direct global->LDS DMA (global_load_lds, 16 B per lane, 1 KiB per wave
  // instruction): no staging registers and every piece is in flight at once.  LDS destination = wave-uniform base +
  // lane*16 (linear image); the per-lane SOURCE address carries the tap table / channel-quad / M-tail clamps.
  auto issue_filter_dma = [&](int kc, float *Abuf) {
#pragma unroll
    for (int it = 0; it < A_IT_; ++it) {
      const int t = UPT >= 256 ? it / PPT : it * TPP + t_local;
      const int uu = UPT >= 256 ? (it % PPT) * 256 + tid : tid % UPT;
      const int gq = uu / BM, mm = uu - gq * BM;
      if (t < ntaps) {
        const float *src = p.w + tapbase[it] + (min(kc * G4 + gq, C4 - 1) * p.ldw + min(m0 + mm, p.ldw - 1)) * 4;
#pragma unroll
        for (int pl = 0; pl < NP; ++pl)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + (size_t)pl * p.wplane),
                                           (__attribute__((address_space(3))) void *)(Abuf + pl * p.a_floats +
                                                                                      (size_t)(256 * it + (wave << 6)) * 4),
                                           16, 0, 0);
      }
    }
  };
  // Halo loads are branch-free: every lane always loads (address clamped to element 0 when the position is padding /
  // out of range) and the value is zeroed at store time, so the CB channel loads of a position are in flight together
  // instead of one memory round trip per `if (valid)` basic block.
  auto load_halo = [&](int c0, int j, float (&xv)[CB]) {
    const int gi = goff[j] >= 0 ? goff[j] : 0;
#pragma unroll
    for (int cc = 0; cc < CB; ++cc) xv[cc] = p.x[gi + min(c0 + cc, p.C - 1) * HWin];
  };
  auto store_halo = [&](int c0, int qb, int j, float (&xv)[CB], float *Xbuf) {  // Xs4[quad][position]: ds_write_b128
    if (loff[j] >= 0) {
      const bool lane_ok = goff[j] >= 0;
      if (p.in_scale) {
#pragma unroll
        for (int cc = 0; cc < CB; ++cc) xv[cc] *= Ss[sseg[j] + min(c0 + cc, p.C - 1)];
      }
#pragma unroll
      for (int cc = 0; cc < CB; ++cc) xv[cc] = (lane_ok && (c0 + cc) < p.C) ? xv[cc] : 0.f;
      if constexpr (X3) {  // hi | mid | lo planes of the 8-channel unit
        bf16x8 h8, m8, l8;
        split3_bf16x8(xv, h8, m8, l8);
        bf16x8 *X8 = reinterpret_cast<bf16x8 *>(Xbuf) + qb * p.planeStride + loff[j];
        const int pstride = xbuf_floats / 4;  // 16-byte units per plane
        X8[0] = h8; X8[pstride] = m8; X8[2 * pstride] = l8;
      } else if constexpr (BF) {  // 8 channels of a position = one 16-byte unit of bf16
        reinterpret_cast<bf16x8 *>(Xbuf)[qb * p.planeStride + loff[j]] = pack_bf16x8(xv);
      } else {
        f32x4 *X4 = reinterpret_cast<f32x4 *>(Xbuf);
        X4[qb * p.planeStride + loff[j]] = f32x4{xv[0], xv[1], xv[2], xv[3]};
        if constexpr (CB == 8) X4[(qb + 1) * p.planeStride + loff[j]] = f32x4{xv[4], xv[5], xv[6], xv[7]};
      }
    }
  };
  // X3: byte offsets of the operand planes this lane reads (half-wave h supplies K half h of each MFMA)
  const int a_pl = p.a_floats * 4, x_pl = xbuf_floats * 4;
  const int aX = half * a_pl, aY = 2 * half * a_pl;  // A: (hi | mid) and (hi | lo)
  const int bZ = 2 * (1 - half) * x_pl;              // B of the third product pair: (lo | hi)
  auto x3_unit = [&](f32x16 (&ac)[WTM][WTN], const char *Au, const char *Xu) {
    bf16x8 ax[WTM], ay[WTM], b0[WTN], b1[WTN], b2[WTN];
#pragma unroll
    for (int i = 0; i < WTM; ++i) {
      ax[i] = *reinterpret_cast<const bf16x8 *>(Au + aX + i * 32 * 16);
      ay[i] = *reinterpret_cast<const bf16x8 *>(Au + aY + i * 32 * 16);
    }
#pragma unroll
    for (int j = 0; j < WTN; ++j) {
      b0[j] = *reinterpret_cast<const bf16x8 *>(Xu + bbytes[j]);
      b1[j] = *reinterpret_cast<const bf16x8 *>(Xu + x_pl + bbytes[j]);
      b2[j] = *reinterpret_cast<const bf16x8 *>(Xu + bZ + bbytes[j]);
    }
    // smallest terms first: hi*lo + lo*hi, then hi*mid + mid*mid, then hi*hi + mid*hi
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
      for (int j = 0; j < WTN; ++j) ac[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ay[i], b2[j], ac[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
      for (int j = 0; j < WTN; ++j) ac[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ax[i], b1[j], ac[i][j], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
      for (int j = 0; j < WTN; ++j) ac[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ax[i], b0[j], ac[i][j], 0, 0, 0);
  };
  auto mfma_taps = [&](const float *Abuf, const float *Xbuf) {
    const char *Ab = reinterpret_cast<const char *>(Abuf) + abytes;
    const char *Xb = reinterpret_cast<const char *>(Xbuf);
    if constexpr (TM) {
      const int rowb = p.IWp * 16;  // one halo row, in bytes
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int t = kh * 3 + kw;
          const int c = (kh & 1) * 2 + (kw & 1);
          const int toffb = (kh == 2 ? 0 : rowb) + (kw == 2 ? 0 : 16);  // x[u - (kh==2), v - (kw==2)] inside the halo tile
          if constexpr (X3) {
#pragma unroll
            for (int u = 0; u < CK / 8; ++u)
              x3_unit(acc[c], Ab + ((t * G4 + u) * BM) * 16, Xb + toffb + u * p.planeStride * 16);
          } else
#pragma unroll
          for (int o = 0; o < CK / (2 * KP); ++o) {
            if constexpr (BF) {
              bf16x8 a[WTM], b[WTN];
#pragma unroll
              for (int i = 0; i < WTM; ++i)
                a[i] = *reinterpret_cast<const bf16x8 *>(Ab + ((t * G4 + o * 2) * BM + i * 32) * 16);
#pragma unroll
              for (int j = 0; j < WTN; ++j)
                b[j] = *reinterpret_cast<const bf16x8 *>(Xb + (bbytes[j] + toffb + o * 2 * p.planeStride * 16));
#pragma unroll
              for (int i = 0; i < WTM; ++i)
#pragma unroll
                for (int j = 0; j < WTN; ++j)
                  acc[c][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[c][i][j], 0, 0, 0);
            } else {
              f32x4 a[WTM], b[WTN];
#pragma unroll
              for (int i = 0; i < WTM; ++i)
                a[i] = *reinterpret_cast<const f32x4 *>(Ab + ((t * G4 + o * 2) * BM + i * 32) * 16);
#pragma unroll
              for (int j = 0; j < WTN; ++j)
                b[j] = *reinterpret_cast<const f32x4 *>(Xb + (bbytes[j] + toffb + o * 2 * p.planeStride * 16));
#pragma unroll
              for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < WTM; ++i)
#pragma unroll
                  for (int j = 0; j < WTN; ++j)
                    acc[c][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[c][i][j], 0, 0, 0);
            }
          }
        }
      return;
    }
    if constexpr (X3 && WTM * WTN <= 4) {
      // all MT taps present (every 3x3 launch): straight-line code over (tap, unit) steps with the operand reads of step
      // i+1 issued before the MFMAs of step i (two register sets).  The rolled loop below exposed one LDS round trip per tap:
      // 12 bf16 MFMAs (384 cycles) cover far less of it than the 16 fp32 ones (1024 cycles) of the exact kernel.
      if (ntaps == MT) {
        constexpr int NS = MT * (CK / 8);
        bf16x8 ax[2][WTM], ay[2][WTM], b0[2][WTN], b1[2][WTN], b2[2][WTN];
        auto ld = [&](int st, int buf) {
          const int t = st / (CK / 8), u = st % (CK / 8);
          const char *Au = Ab + ((t * G4 + u) * BM) * 16;
          const char *Xu = Xb + __builtin_amdgcn_readlane(tofft, t) + u * p.planeStride * 16;
#pragma unroll
          for (int i = 0; i < WTM; ++i) {
            ax[buf][i] = *reinterpret_cast<const bf16x8 *>(Au + aX + i * 32 * 16);
            ay[buf][i] = *reinterpret_cast<const bf16x8 *>(Au + aY + i * 32 * 16);
          }
#pragma unroll
          for (int j = 0; j < WTN; ++j) {
            b0[buf][j] = *reinterpret_cast<const bf16x8 *>(Xu + bbytes[j]);
            b1[buf][j] = *reinterpret_cast<const bf16x8 *>(Xu + x_pl + bbytes[j]);
            b2[buf][j] = *reinterpret_cast<const bf16x8 *>(Xu + bZ + bbytes[j]);
          }
        };
        ld(0, 0);
#pragma unroll
        for (int st = 0; st < NS; ++st) {
          const int cb = st & 1;
          if (st + 1 < NS) ld(st + 1, cb ^ 1);
          __builtin_amdgcn_sched_barrier(0);  // keep the reads ahead of this step's MFMAs (the scheduler sinks them otherwise)
#pragma unroll
          for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int j = 0; j < WTN; ++j)
              acc[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ay[cb][i], b2[cb][j], acc[0][i][j], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int j = 0; j < WTN; ++j)
              acc[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ax[cb][i], b1[cb][j], acc[0][i][j], 0, 0, 0);
#pragma unroll
          for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int j = 0; j < WTN; ++j)
              acc[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ax[cb][i], b0[cb][j], acc[0][i][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
        return;
      }
    }
    for (int t = 0; t < ntaps; ++t) {
      const int toffb = __builtin_amdgcn_readlane(tofft, t);  // tap shift in bytes (lane t of the table)
      if constexpr (X3) {
#pragma unroll
        for (int u = 0; u < CK / 8; ++u)
          x3_unit(acc[0], Ab + ((t * G4 + u) * BM) * 16, Xb + toffb + u * p.planeStride * 16);
      } else if constexpr (BF) {
#pragma unroll
        for (int o = 0; o < CK / 16; ++o) {  // half-wave h holds k = 8h .. 8h+7 = the 8 channels of unit 2o+h
          bf16x8 a[WTM], b[WTN];
#pragma unroll
          for (int i = 0; i < WTM; ++i)
            a[i] = *reinterpret_cast<const bf16x8 *>(Ab + ((t * G4 + o * 2) * BM + i * 32) * 16);
#pragma unroll
          for (int j = 0; j < WTN; ++j)
            b[j] = *reinterpret_cast<const bf16x8 *>(Xb + (bbytes[j] + toffb + o * 2 * p.planeStride * 16));
#pragma unroll
          for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int j = 0; j < WTN; ++j)
              acc[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[0][i][j], 0, 0, 0);
        }
      } else if constexpr (CK == 4) {
        f32x2 a[WTM], b[WTN];
#pragma unroll
        for (int i = 0; i < WTM; ++i) a[i] = *reinterpret_cast<const f32x2 *>(Ab + (t * BM + i * 32) * 16);
#pragma unroll
        for (int j = 0; j < WTN; ++j) b[j] = *reinterpret_cast<const f32x2 *>(Xb + (bbytes[j] + toffb));
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int j = 0; j < WTN; ++j)
              acc[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[0][i][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int o = 0; o < CK / 8; ++o) {
          f32x4 a[WTM], b[WTN];
#pragma unroll
          for (int i = 0; i < WTM; ++i)
            a[i] = *reinterpret_cast<const f32x4 *>(Ab + ((t * G4 + o * 2) * BM + i * 32) * 16);
#pragma unroll
          for (int j = 0; j < WTN; ++j)
            b[j] = *reinterpret_cast<const f32x4 *>(Xb + (bbytes[j] + toffb + o * 2 * p.planeStride * 16));
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < WTM; ++i)
#pragma unroll
              for (int j = 0; j < WTN; ++j)
                acc[0][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[0][i][j], 0, 0, 0);
        }
      }
    }
  };

  bool done = false;
  if constexpr ((X3 && PF == 0) || PF < 0) {
    // f32x3 K loop with the halo tile of chunk k+1 PREFETCHED INTO REGISTERS while chunk k's MFMAs run (single LDS buffer: three
    // operand planes leave no room for a second one at 2 blocks/CU).  Per chunk: barrier | split + store the prefetched tile
    // | issue the filter DMA | issue the loads of the next tile | wait for the DMA only (counted vmcnt: the loads stay in
    // flight) | barrier | MFMAs.  The HBM round trip of the halo no longer sits between two MFMA phases; what stays exposed
    // is the filter DMA (L2-resident) and the store pass.
    constexpr int NJX = CK == 8 ? 3 : 2, NBT = CK / CB;  // positions per thread this path holds (3: the 9 x 65 halo of a
                                                         // stride-2 128-pixel tile); 8-channel load batches per position
    if (p.NJ <= NJX) {
      done = true;
      float xr[NJX][NBT][CB];
      auto loadx = [&](int c0) {
#pragma unroll
        for (int j = 0; j < NJX; ++j)
          if (j < p.NJ) {  // uniform
#pragma unroll
            for (int bt = 0; bt < NBT; ++bt) load_halo(c0 + bt * CB, j, xr[j][bt]);
          }
      };
      if (kbeg < kend) loadx(kbeg * CK);
      if (p.in_scale) __syncthreads();  // Ss visible
      for (int kc = kbeg; kc < kend; ++kc) {
        if (kc > kbeg) {  // every wave is done reading the previous chunk's tiles
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
        }
#pragma unroll
        for (int j = 0; j < NJX; ++j)
          if (j < p.NJ) {
#pragma unroll
            for (int bt = 0; bt < NBT; ++bt) store_halo(kc * CK + bt * CB, bt * (CB / KP), j, xr[j][bt], Xs);
          }
        issue_filter_dma(kc, As);
        // the counted s_waitcnt below assumes every DMA piece was issued BEFORE the halo loads of the next chunk: pin that order
        // (nothing else keeps the compiler from hoisting a global load above the LDS-DMA builtins)
        __builtin_amdgcn_sched_barrier(0);
        if (kc + 1 < kend) {
          loadx((kc + 1) * CK);
          // the DMA pieces are older than the loads just issued: leave exactly those loads outstanding
          // (vmcnt is a 6-bit field: 63 still leaves every halo load but one in flight)
          constexpr int V1 = NBT * CB < 63 ? NBT * CB : 63, V2 = 2 * NBT * CB < 63 ? 2 * NBT * CB : 63;
          constexpr int V3 = NJX * NBT * CB < 63 ? NJX * NBT * CB : 63;
          if (p.NJ == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(V1) : "memory");
          else if (p.NJ == 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(V2) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(V3) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        mfma_taps(As, Xs);
      }
    }
  }
  if (done) {
  } else if constexpr (PF > 0) {
    // Software-pipelined K loop, ONE barrier per chunk: the filter DMA and the halo loads of chunk k+1 are issued
    // before the MFMA work of chunk k (they land in the other LDS buffer / in registers while the matrix cores run),
    // so no memory round trip is exposed.  Measured need: without it the staging of co-resident blocks did NOT overlap
    // their MFMA phases (ablation: 0.70 ms full, 0.60 ms with the filter DMA removed, 0.50 ms pure MFMA time).
    float xv[NJC][CB];
    if (kbeg < kend) {
      issue_filter_dma(kbeg, As);
#pragma unroll
      for (int j = 0; j < NJC; ++j)
        if (j < p.NJ) load_halo(kbeg * CK, j, xv[j]);
      if (p.in_scale) __syncthreads();  // Ss visible
#pragma unroll
      for (int j = 0; j < NJC; ++j)
        if (j < p.NJ) store_halo(kbeg * CK, 0, j, xv[j], Xs);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    for (int kc = kbeg; kc < kend; ++kc) {
      const int cur = (kc - kbeg) & 1;
      const bool more = kc + 1 < kend;
      if (more) {
        issue_filter_dma(kc + 1, As + (cur ^ 1) * NP * p.a_floats);
#pragma unroll
        for (int j = 0; j < NJC; ++j)
          if (j < p.NJ) load_halo((kc + 1) * CK, j, xv[j]);
      }
      mfma_taps(As + cur * NP * p.a_floats, Xs + cur * NP * xbuf_floats);
      if (more) {
#pragma unroll
        for (int j = 0; j < NJC; ++j)
          if (j < p.NJ) store_halo((kc + 1) * CK, 0, j, xv[j], Xs + (cur ^ 1) * NP * xbuf_floats);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the filter DMA has landed (explicit, not left to the fence)
      __syncthreads();
    }
  } else {
    if (p.in_scale) __syncthreads();  // Ss visible
    for (int kc = kbeg; kc < kend; ++kc) {
      const int c0 = kc * CK;
      if (kc > kbeg) __syncthreads();
      if constexpr (BF && CK == 2 * CB) {
        // 16-channel bf16 chunks: BOTH 8-channel load batches of a position are issued before the first store (one HBM round
        // trip per chunk instead of two -- the bf16 MFMA phase is too short to hide the second)
        float xa[CB], xb[CB];
        load_halo(c0, 0, xa);
        load_halo(c0 + CB, 0, xb);
        issue_filter_dma(kc, As);
#pragma unroll
        for (int j = 0; j < NJC; ++j) {
          if (j < p.NJ) {  // uniform
            if (j > 0) { load_halo(c0, j, xa); load_halo(c0 + CB, j, xb); }
            store_halo(c0, 0, j, xa, Xs);
            store_halo(c0 + CB, CB / KP, j, xb, Xs);
          }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        mfma_taps(As, Xs);
        continue;
      }
      // halo batch 0 first: its round trip hides under the issue phase of the filter DMA below
      float xv0[CB];
      load_halo(c0, 0, xv0);
      issue_filter_dma(kc, As);
#pragma unroll 1
      for (int cb = 0; cb < CK; cb += CB) {
#pragma unroll
        for (int j = 0; j < NJC; ++j) {
          if (j < p.NJ) {  // uniform
            float xv[CB];
            if (j == 0 && cb == 0) {
#pragma unroll
              for (int cc = 0; cc < CB; ++cc) xv[cc] = xv0[cc];
            } else {
              load_halo(c0 + cb, j, xv);
            }
            store_halo(c0 + cb, cb / KP, j, xv, Xs);
          }
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the filter DMA has landed (explicit, not left to the fence)
      __syncthreads();
      mfma_taps(As, Xs);
    }
  }

  // ---- epilogue (conv_common.h conv_epilogue)
  const int HWout = p.Hout * p.Wout;
  if constexpr (TM) {  // the four classes of input position (u, v) land on outputs (2u + cy, 2v + cx)
    float *const yb = p.ksplit > 1 ? p.y + (size_t)ks * p.slab : p.y;
    const float alpha = p.e.alpha;
#pragma unroll
    for (int j = 0; j < WTN; ++j) {
      const int n = (wn * WTN + j) * 32 + (lane & 31);
      const int q = n & TWm, rr = n >> p.logTW;
      const int seg = rr >> p.logTHs, r = rr & THm;
      const int b = bg * p.NSEG + seg, u = u0 + r, v = v0 + q;
      const bool okp = b < p.B && u < ci.Ug && v < ci.Vg;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int Y = 2 * u + (c >> 1), X = 2 * v + (c & 1);
        const bool ok = okp && Y < p.Hout && X < p.Wout;
        const int pix = Y * p.Wout + X;
#pragma unroll
        for (int i = 0; i < WTM; ++i)
#pragma unroll
          for (int r16 = 0; r16 < 16; ++r16) {
            const int m = m0 + (wm * WTM + i) * 32 + (r16 & 3) + 8 * (r16 >> 2) + 4 * (lane >> 5);
            if (ok && m < p.M) yb[(b * p.M + m) * HWout + pix] = acc[c][i][j][r16] * alpha;
          }
      }
    }
    return;
  }
  int e_pix[WTN], e_b[WTN];
#pragma unroll
  for (int j = 0; j < WTN; ++j) {
    const int n = (wn * WTN + j) * 32 + (lane & 31);
    const int q = n & TWm, rr = n >> p.logTW;
    const int seg = rr >> p.logTHs, r = rr & THm;
    const int b = bg * p.NSEG + seg, u = u0 + r, v = v0 + q;
    const bool okpix = b < p.B && u < ci.Ug && v < ci.Vg;
    const int Y = u * p.osy + ci.ooy, X = v * p.osx + ci.oox;
    e_pix[j] = okpix ? Y * p.Wout + X : -1;
    e_b[j] = okpix ? b : 0;
  }
  constexpr int RG = OCC == 4 ? 2 : 4;  // rows per load batch (register budget: the epilogue must not raise the kernel's allocation)
  conv_epilogue<WTM, WTN, RG, BF>(acc[0], p.e, p.y, p.ksplit > 1 ? p.y + (size_t)ks * p.slab : nullptr, p.M, HWout,
                              m0 + wm * WTM * 32, lane, e_pix, e_b, bg < p.B, bg, p.dot_slots, (tu * ci.tilesV + tv) * WGN + wn,
                              p.Hout, p.Wout);
}

template <int WGM, int WGN, int WTM, int WTN, int CK, int MT, int PF = 0, int OCC = 3, bool BF = false, bool TM = false, bool X3 = false>
static int launch_fprop(ConvP &p, hipStream_t st, int maxtaps, int maxTilesN, const NameOut *name) {
  constexpr int BM = WGM * WTM * 32;
  constexpr int G4 = (CK + (BF ? 7 : 3)) / (BF ? 8 : 4);
  if (PF > 0 && p.NJ > PF) return TBG_EUNSUPPORTED;
  p.a_floats = maxtaps * G4 * 4 * BM;
  p.ck_rt = CK;
  p.nchunks = ceil_div(p.C, CK);
  p.cps = ceil_div(p.nchunks, p.ksplit);  // splits past the last chunk run no K loop and store zeros: every slab is fully written
  const size_t lds = ((size_t)(PF > 0 ? 2 : 1) * (X3 ? 3 : 1) * ((size_t)p.a_floats + (size_t)G4 * 4 * p.planeStride) +
                      (p.in_scale ? (size_t)p.NSEG * p.C : 0)) * sizeof(float);
  if (lds > 160 * 1024) return TBG_EUNSUPPORTED;
  if (maxtaps > MT) return TBG_EUNSUPPORTED;
  // fused dot product: one slot per (pixel tile of the image, wave column); needs one image per tile and one class
  p.dot_slots = (p.NSEG == 1 && p.nclass == 1 && p.ksplit == 1 && !TM) ? p.cls[0].tilesU * p.cls[0].tilesV * WGN : 0;
  if (p.e.dot_aux && p.dot_slots == 0) return TBG_EUNSUPPORTED;
  if (name) {
    if (name->dot_slots) *name->dot_slots = p.dot_slots;
    if (name->blocks) *name->blocks = maxTilesN * ceil_div(p.M, BM) * p.nclass * p.ksplit;
    if (!name->buf) return TBG_OK;
    snprintf(name->buf, name->n, "conv_fprop_kernel<%d, %d, %d, %d, %d, %d, %d, %d, %s, %s, %s>", WGM, WGN, WTM, WTN, CK, MT, PF,
             OCC, BF ? "true" : "false", TM ? "true" : "false", X3 ? "true" : "false");
    return TBG_OK;
  }
  auto kern = conv_fprop_kernel<WGM, WGN, WTM, WTN, CK, MT, PF, OCC, BF, TM, X3>;
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return TBG_EHIP;
  }
  dim3 grid(maxTilesN, ceil_div(p.M, BM), p.nclass * p.ksplit);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}


// mode: 0 = exact fp32 (v_mfma_f32_32x32x2_f32), 1 = bf16 operands, 2 = f32x3 (three bf16 terms per fp32 operand)
static int conv2d_tiled(const tbg_conv_desc *d, const float *x, const float *w, float *y, const float *in_scale,
                        const tbg_epilogue *epi, void *stream, const NameOut *name, int mode, int variant, bool force64);

static int conv2d_impl(const tbg_conv_desc *d, const float *x, const float *w, float *y, const float *in_scale,
                       const tbg_epilogue *epi, void *stream, const NameOut *name, int mode = 0, int variant = 0) {
  int rc = conv2d_tiled(d, x, w, y, in_scale, epi, stream, name, mode, variant, false);
  // a tile whose halo does not fit LDS (three operand planes in f32x3; many small images per 256-pixel tile): the 64 x 64
  // tile packs a quarter of the images.  launch_fprop refuses BEFORE launching anything, so the retry is clean.
  if (rc == TBG_EUNSUPPORTED && variant == 0) rc = conv2d_tiled(d, x, w, y, in_scale, epi, stream, name, mode, variant, true);
  return rc;
}

static int conv2d_tiled(const tbg_conv_desc *d, const float *x, const float *w, float *y, const float *in_scale,
                        const tbg_epilogue *epi, void *stream, const NameOut *name, int mode, int variant, bool force64) {
  const bool bf = mode == 1, x3 = mode == 2;
  if (!d || !epi_valid(epi)) return TBG_EINVAL;
  if (!name && (!x || !w || (!y && !epi_has_sink(epi)))) return TBG_EINVAL;  // (a unit sink may be the only output)
  if (d->B < 1 || d->C < 1 || d->M < 1 || d->Hin < 1 || d->Win < 1 || d->Hout < 1 || d->Wout < 1) return TBG_EINVAL;
  if (d->KH < 1 || d->KW < 1 || d->KH * d->KW > MAXTAPS) return TBG_EUNSUPPORTED;
  if (d->sy < 1 || d->sy > 2 || d->sx < 1 || d->sx > 2) return TBG_EUNSUPPORTED;
  if (d->ldw < d->M || (reinterpret_cast<uintptr_t>(w) & 15) != 0) return TBG_EINVAL;
  if (d->ksplit < 1) return TBG_EINVAL;
  if (d->ksplit > 1 && epi && (epi->out_scale || epi->bias || epi->noise || epi->residual || epi->dot_aux || epi->gate || epi->act != TBG_ACT_LINEAR ||
                               epi->units_out))
    return TBG_EINVAL;
  if ((double)d->B * d->C * d->Hin * d->Win > 2147483647.0 || (double)d->B * d->M * d->Hout * d->Wout > 2147483647.0)
    return TBG_ERANGE;
  if (d->transposed) {
    if (d->py != 0 || d->px != 0) return TBG_EINVAL;
    if (d->Hout < (d->Hin - 1) * d->sy + d->KH || d->Wout < (d->Win - 1) * d->sx + d->KW) return TBG_EINVAL;
  } else {
    // every output pixel must exist: (Hout-1)*sy - py + KH-1 may run past Hin (zero fill) -- allowed
    if (d->py < 0 || d->px < 0) return TBG_EINVAL;
  }

  ConvP p;
  p.x = x; p.w = w; p.in_scale = in_scale; p.y = y;
  p.B = d->B; p.C = d->C; p.M = d->M; p.Hin = d->Hin; p.Win = d->Win; p.Hout = d->Hout; p.Wout = d->Wout;
  p.ldw = d->ldw;
  p.e = make_epi(epi);
  if (p.e.units_out && mode == 0) return TBG_EUNSUPPORTED;  // unit tensors belong to the bf16-pipe arithmetics (bf16, f32x3)
  if (const int rcs = epi_sink_geometry(p.e, d->B, d->M, d->Hout, d->Wout)) return rcs;
  const int T = d->KH * d->KW;
  p.wplane = T * ((d->C + 7) / 8) * d->ldw * 4;
  // merged form of the stride-2 transposed 3x3 convolution (see the TM template parameter): store-only epilogues
  const bool plain_epi = !epi || (!epi->out_scale && !epi->bias && !epi->noise && !epi->residual && !epi->dot_aux && !epi->gate &&
                                  epi->act == TBG_ACT_LINEAR && epi->gain == 1.f && !epi->units_out);
  // Measured (tools/bench_transposed.py, profiles/r02_transposed_forms.txt): in bf16 the merged form wins on the large maps
  // (216 vs 158 TFLOP/s on 32x128 128->128, 232 vs 124 on the 128->64 data gradient: the per-class form re-stages the halo
  // four times and bf16 launches are staging bound) and loses below ~16k input pixels; in fp32 (MFMA bound) the per-class
  // form's 128x128 tiles at 3 waves/SIMD beat the merged form's 64x128 tiles at 2 waves/SIMD everywhere (76 vs 68).
  // variant 5 forces the merged form, variant 4 the per-class form.
  const bool merged = d->transposed && d->sy == 2 && d->sx == 2 && d->KH == 3 && d->KW == 3 && plain_epi && variant != 4 &&
                      d->M > 32 && (variant == 5 || variant == 6 || ((bf || x3) && (long long)d->B * d->Hin * d->Win >= 16384));
  int maxUg = 0, maxVg = 0, maxKH = 0, maxKW = 0, maxtaps = 0;
  if (!d->transposed) {
    p.sy = d->sy; p.sx = d->sx; p.osy = 1; p.osx = 1; p.nclass = 1;
    ClassInfo &c = p.cls[0];
    c.ntaps = T; c.KWc = d->KW; c.py = d->py; c.px = d->px; c.ooy = 0; c.oox = 0; c.Ug = d->Hout; c.Vg = d->Wout;
    for (int t = 0; t < T; ++t) c.wtap[t] = d->flip ? T - 1 - t : t;
    maxUg = c.Ug; maxVg = c.Vg; maxKH = d->KH; maxKW = d->KW; maxtaps = T;
  } else if (merged) {
    // one "class" that carries all 9 taps: the tile indexes input positions (u, v), halo = one row above / one column left
    p.sy = 1; p.sx = 1; p.osy = 2; p.osx = 2; p.nclass = 1;
    ClassInfo &c = p.cls[0];
    c.ntaps = 9; c.KWc = 3; c.py = 1; c.px = 1; c.ooy = 0; c.oox = 0;
    c.Ug = ceil_div(d->Hout, 2); c.Vg = ceil_div(d->Wout, 2);
    for (int t = 0; t < 9; ++t) c.wtap[t] = d->flip ? 8 - t : t;
    maxUg = c.Ug; maxVg = c.Vg; maxKH = 2; maxKW = 2; maxtaps = 9;
  } else {
    p.sy = 1; p.sx = 1; p.osy = d->sy; p.osx = d->sx; p.nclass = d->sy * d->sx;
    int k = 0;
    for (int cy = 0; cy < d->sy; ++cy)
      for (int cx = 0; cx < d->sx; ++cx, ++k) {
        ClassInfo &c = p.cls[k];
        const int KHc = cy < d->KH ? ceil_div(d->KH - cy, d->sy) : 0;
        const int KWc = cx < d->KW ? ceil_div(d->KW - cx, d->sx) : 0;
        c.ntaps = KHc * KWc; c.KWc = KWc > 0 ? KWc : 1;
        c.py = KHc > 0 ? KHc - 1 : 0; c.px = KWc > 0 ? KWc - 1 : 0;
        c.ooy = cy; c.oox = cx;
        c.Ug = d->Hout > cy ? ceil_div(d->Hout - cy, d->sy) : 0;
        c.Vg = d->Wout > cx ? ceil_div(d->Wout - cx, d->sx) : 0;
        for (int khp = 0; khp < KHc; ++khp)
          for (int kwp = 0; kwp < KWc; ++kwp) {
            const int kh = cy + d->sy * (KHc - 1 - khp), kw = cx + d->sx * (KWc - 1 - kwp);
            const int t = kh * d->KW + kw;
            c.wtap[khp * KWc + kwp] = d->flip ? T - 1 - t : t;
          }
        if (c.Ug > maxUg) maxUg = c.Ug;
        if (c.Vg > maxVg) maxVg = c.Vg;
        if (KHc > maxKH) maxKH = KHc;
        if (KWc > maxKW) maxKW = KWc;
        if (c.ntaps > maxtaps) maxtaps = c.ntaps;
      }
  }
  if (maxtaps < 1) maxtaps = 1;
  if (maxKH < 1) maxKH = 1;
  if (maxKW < 1) maxKW = 1;

  // tile configuration
  int BM, BN;
  if (merged) { BM = 64; BN = 128; }  // 4 classes x (1 x 2) MFMA tiles per wave = 128 accumulators
  else if (force64) { BM = 64; BN = 64; }
  else {
    const long long npix = (long long)d->B * maxUg * maxVg;  // N of the (largest class) GEMM
    if (d->M <= 32) { BM = 32; BN = 256; }
    else if (d->M <= 64) { BM = 64; BN = (npix + 255) / 256 < 96 ? 64 : 256; }
    else {
      const long long tiles128 = (long long)ceil_div(d->M, 128) * ((npix + 127) / 128);
      if (tiles128 <= (x3 ? 32 : 16)) { BM = 64; BN = 64; }  // tiny-spatial / wide-channel: more, smaller blocks
      else { BM = 128; BN = 128; }
      // f32x3, launches of >= 1024 128x128 tiles (the 64x256 layers; the joint discriminator pass's 32x128 layers): a
      // 128 x 256 tile halves the filter DMA per MFMA and the barrier count and still fills whole rounds of the 512 slots.
      // variant 1 forces it, variant 2 forbids it (tbg_conv2d_x3_variant).
      if (x3 && !d->transposed && d->sy == 1 && d->sx == 1 && T == 9 && BM == 128 && d->ksplit == 1 &&
          (variant == 1 || (variant == 0 && tiles128 >= 1024)))
        BN = 256;
      // f32x3 stride-2 forward: the 9 x 66 halo of a 128-pixel tile (three planes, 28.5 KB) beside the 55.3 KB filter tile
      // leaves ONE block per CU; a 64-pixel tile (5 x 34 x 2 halo: 14.4 KB) fits two.  +5..9 % per layer in isolation
      // (profiles/r03_x3_half_tiles.txt).  variant 2 keeps the 128-pixel tile.
      if (x3 && !d->transposed && d->sy == 2 && d->sx == 2 && T == 9 && BM == 128 && BN == 128 && variant == 0) BN = 64;
    }
  }
  constexpr int twmax = 32;  // tile rows of at most 32 pixels (wider rows were measured slower: fewer rows per halo)
  int TW = 1, THs = 1;
  for (int attempt = 0;; ++attempt) {
    TW = pow2ceil(maxVg) < twmax ? pow2ceil(maxVg) : (twmax < BN ? twmax : BN);
    const int TR = BN / TW;
    THs = pow2ceil(maxUg) < TR ? pow2ceil(maxUg) : TR;
    if (d->transposed && !merged && attempt == 0 && variant != 4) {
      // Output-parity class grids are (H + 1) x (W + 1): 33 rows in tiles of 8 are 40 (+21 %), 17 in tiles of 4 are 20.  Lower
      // tiles (several images per tile instead) pad less but re-load the halo row(s) more often: pick the tile height with the
      // smallest padded-rows x halo-overhead product.  variant 8 / 9 force one / two halvings (measurement aid).
      int best = THs;
      if (variant == 8 || variant == 9) {
        for (int k = 0; k < variant - 7 && best > 1; ++k) best >>= 1;
      } else {
        double bestc = 1e30;
        for (int t = THs; t >= 1; t >>= 1) {
          const double c = (double)ceil_div(maxUg, t) * t * (1.0 + 0.25 * (maxKH - 1) / t);
          if (c < bestc * 0.999) { bestc = c; best = t; }
        }
      }
      THs = best;
    }
    p.logTW = ilog2(TW); p.logTHs = ilog2(THs); p.NSEG = TR / THs;
    p.IHs = (THs - 1) * p.sy + maxKH;
    p.IWs = (TW - 1) * p.sx + maxKW;
    p.HALFW = (p.IWs + 1) / 2;
    int iwp = (p.sx == 2) ? 2 * p.HALFW : p.IWs;
    if (TW < 32 && TW >= 8) {  // rows of a 32-lane B read land on disjoint banks (cheap for TW >= 8 only)
      int cand = iwp;
      for (int it = 0; it < 32 && ((p.sy * cand) & 31) != (TW & 31); ++it) ++cand;
      if (((p.sy * cand) & 31) == (TW & 31) && cand <= iwp + iwp / 2 + 8) iwp = cand;
    }
    p.IWp = iwp;
    p.planeStride = p.NSEG * p.IHs * p.IWp;
    p.ppc = p.NSEG * p.IHs * p.IWs;
    p.NJ = ceil_div(p.ppc, 256);
    if (p.NJ <= MAXNJ) break;
    // very narrow maps pack many images into a 256-pixel tile and the halo of all of them exceeds what a thread can
    // describe: fall back to the 64x64 tile once
    if (attempt == 0 && BN > 64 && !merged) { BM = 64; BN = 64; continue; }
    return TBG_EUNSUPPORTED;
  }
  p.nBG = ceil_div(p.B, p.NSEG);
  int maxTilesN = 0;
  for (int k = 0; k < p.nclass; ++k) {
    ClassInfo &c = p.cls[k];
    c.tilesU = c.Ug > 0 ? ceil_div(c.Ug, THs) : 1;
    c.tilesV = c.Vg > 0 ? ceil_div(c.Vg, TW) : 1;
    const int tiles = c.tilesU * c.tilesV * p.nBG;
    if (tiles > maxTilesN) maxTilesN = tiles;
  }
  p.ksplit = d->ksplit;
  p.slab = (long long)d->B * d->M * d->Hout * d->Wout;
  hipStream_t st = tbg_stream(stream);
  if (x3) {  // f32x3: the tile shapes of the fp32 path, 8-channel chunks (16 for the few-tap classes), 2 blocks/CU
    if (merged) return launch_fprop<2, 2, 1, 2, 8, MAXTAPS, 0, 2, true, true, true>(p, st, maxtaps, maxTilesN, name);
    if (variant != 0 && variant != 1 && variant != 2 && variant != 4 && variant != 5) return TBG_EUNSUPPORTED;
    // (The 64 x 64 tile -- the small maps of both networks and the whole frozen-OCR stack, 1.6 ms of GPU time per step -- is
    // bound by the latency of its filter stream, not by chunk count or staging: under graph replay a C = 256, M = 256 layer on
    // 2 x 25 maps takes 56 us unsplit = 1.7 us per 8-channel chunk against 0.4 us of MFMA; 16- and 32-channel chunks and the
    // double-buffered loop (DMA of chunk k+1 under the MFMAs of chunk k) all measured the same, profiles/r03_small_tile_forms.txt.
    // More than one filter slice in flight per CU would need the halo tile off the register path too -- vmcnt retires in order.)
    // (Round 3, large tiles: a "split-filter pipeline" -- the filter tile DMA'd in two tap halves, each re-filled for chunk k+1
    // as soon as every wave is done with it in chunk k, three barriers per chunk -- was built and measured: bit-identical
    // results, +0..4 % on the 64x256 layers, -1..-5 % elsewhere.  The exposed DMA wait is worth +10..20 % in the ablation
    // (profiles/r03_x3_fprop_time_split.txt), but the halo tile's register prefetch shares vmcnt with the DMA: hipcc puts
    // s_waitcnt vmcnt(0) in front of the halo stores of a loop that also issues DMA pieces -- even with every VMEM instruction
    // issued unconditionally and in the same order on every path -- which waits for the half just issued.  Not adopted.)
    if (BM == 128 && BN == 256) return launch_fprop<2, 2, 2, 4, 8, MAXTAPS, 0, 2, true, false, true>(p, st, maxtaps, maxTilesN, name);
    if (BM == 128 && BN == 64) return launch_fprop<2, 2, 2, 1, 8, MAXTAPS, 0, 2, true, false, true>(p, st, maxtaps, maxTilesN, name);
    if (maxtaps <= 4) {
      if (BM == 32) return launch_fprop<1, 4, 1, 2, 16, 4, 0, 2, true, false, true>(p, st, maxtaps, maxTilesN, name);
      if (BM == 64 && BN == 64) return launch_fprop<2, 2, 1, 1, 16, 4, 0, 2, true, false, true>(p, st, maxtaps, maxTilesN, name);
      if (BM == 64) return launch_fprop<1, 4, 2, 2, 16, 4, 0, 2, true, false, true>(p, st, maxtaps, maxTilesN, name);
      return launch_fprop<2, 2, 2, 2, 16, 4, 0, 2, true, false, true>(p, st, maxtaps, maxTilesN, name);
    }
    if (BM == 32) return launch_fprop<1, 4, 1, 2, 16, MAXTAPS, 0, 2, true, false, true>(p, st, maxtaps, maxTilesN, name);
    if (BM == 64 && BN == 64) return launch_fprop<2, 2, 1, 1, 8, MAXTAPS, 0, 2, true, false, true>(p, st, maxtaps, maxTilesN, name);
    if (BM == 64) return launch_fprop<1, 4, 2, 2, 8, MAXTAPS, 0, 2, true, false, true>(p, st, maxtaps, maxTilesN, name);
    return launch_fprop<2, 2, 2, 2, 8, MAXTAPS, 0, 2, true, false, true>(p, st, maxtaps, maxTilesN, name);
  }
  if (merged) {
    if (bf) return launch_fprop<2, 2, 1, 2, 16, MAXTAPS, 0, 2, true, true>(p, st, maxtaps, maxTilesN, name);
    if (variant == 6) return launch_fprop<2, 2, 1, 2, 16, MAXTAPS, 0, 2, false, true>(p, st, maxtaps, maxTilesN, name);
    return launch_fprop<2, 2, 1, 2, 8, MAXTAPS, 0, 2, false, true>(p, st, maxtaps, maxTilesN, name);
  }
  if (bf) {  // bf16-in MFMA: chunks of 16 channels (32 for the few-tap classes), the same four tile shapes
    if (variant == 1 && BM == 128 && BN == 128 && maxtaps > 4) {  // 128x256 tile: every filter byte feeds 256 pixels
      // re-derive the pixel tiling for BN = 256 (same rules as above)
      const int TR = 256 / TW;
      THs = pow2ceil(maxUg) < TR ? pow2ceil(maxUg) : TR;
      p.logTHs = ilog2(THs); p.NSEG = TR / THs;
      p.IHs = (THs - 1) * p.sy + maxKH;
      p.planeStride = p.NSEG * p.IHs * p.IWp;
      p.ppc = p.NSEG * p.IHs * p.IWs;
      p.NJ = ceil_div(p.ppc, 256);
      if (p.NJ > MAXNJ) return TBG_EUNSUPPORTED;
      p.nBG = ceil_div(p.B, p.NSEG);
      maxTilesN = 0;
      for (int k = 0; k < p.nclass; ++k) {
        ClassInfo &c = p.cls[k];
        c.tilesU = c.Ug > 0 ? ceil_div(c.Ug, THs) : 1;
        const int tiles = c.tilesU * c.tilesV * p.nBG;
        if (tiles > maxTilesN) maxTilesN = tiles;
      }
      return launch_fprop<2, 2, 2, 4, 16, MAXTAPS, 0, 2, true>(p, st, maxtaps, maxTilesN, name);
    }
    if (variant == 2 && BM == 128 && BN == 128 && maxtaps > 4)  // 32-channel chunks: half the barriers
      return launch_fprop<2, 2, 2, 2, 32, MAXTAPS, 0, 2, true>(p, st, maxtaps, maxTilesN, name);
    // 128 x 128 tiles of the stride-1 layers: the register-prefetch K loop of the f32x3 path (PF = -1: the halo of chunk k+1
    // is loaded while chunk k's MFMAs run) -- 628 vs 518 TFLOP/s on 64x256 128->128, 565 vs 510 on 32x128, 702 vs 659 on
    // 16x64 256->256 at B = 32 (profiles/r03_bf16_regprefetch.txt); the small tiles and the strided layers measured -2 %.
    // variant 3 forces it, variant 4 / 5 keep the plain loop.
    if (maxtaps > 4 && BM == 128 && BN == 128 &&
        (variant == 3 || (variant == 0 && !d->transposed && d->sy == 1 && d->sx == 1)))
      return launch_fprop<2, 2, 2, 2, 16, MAXTAPS, -1, 3, true>(p, st, maxtaps, maxTilesN, name);
    if (variant == 3) return TBG_EUNSUPPORTED;
    if (variant != 0 && variant != 4 && variant != 5) return TBG_EUNSUPPORTED;
    if (maxtaps > 1 && maxtaps <= 4) {
      if (BM == 32) return launch_fprop<1, 4, 1, 2, 32, 4, 0, 3, true>(p, st, maxtaps, maxTilesN, name);
      if (BM == 64 && BN == 64) return launch_fprop<2, 2, 1, 1, 32, 4, 0, 3, true>(p, st, maxtaps, maxTilesN, name);
      if (BM == 64) return launch_fprop<1, 4, 2, 2, 32, 4, 0, 3, true>(p, st, maxtaps, maxTilesN, name);
      return launch_fprop<2, 2, 2, 2, 32, 4, 0, 3, true>(p, st, maxtaps, maxTilesN, name);
    }
    if (BM == 32) return launch_fprop<1, 4, 1, 2, 16, MAXTAPS, 0, 3, true>(p, st, maxtaps, maxTilesN, name);
    if (BM == 64 && BN == 64) return launch_fprop<2, 2, 1, 1, 16, MAXTAPS, 0, 3, true>(p, st, maxtaps, maxTilesN, name);
    if (BM == 64) return launch_fprop<1, 4, 2, 2, 16, MAXTAPS, 0, 3, true>(p, st, maxtaps, maxTilesN, name);
    return launch_fprop<2, 2, 2, 2, 16, MAXTAPS, 0, 3, true>(p, st, maxtaps, maxTilesN, name);
  }
  // few-tap launches (the parity classes of a stride-2 transposed 3x3: 1/2/2/4 taps): a deeper channel chunk keeps
  // the MFMA count per barrier pair up (4 taps x 16 channels instead of 4 x 8)
  if (maxtaps > 1 && maxtaps <= 4) {
    if (BM == 32) return launch_fprop<1, 4, 1, 2, 16, 4>(p, st, maxtaps, maxTilesN, name);
    if (BM == 64 && BN == 64) return launch_fprop<2, 2, 1, 1, 16, 4>(p, st, maxtaps, maxTilesN, name);
    if (BM == 64) return launch_fprop<1, 4, 2, 2, 16, 4>(p, st, maxtaps, maxTilesN, name);
    return launch_fprop<2, 2, 2, 2, 16, 4>(p, st, maxtaps, maxTilesN, name);
  }
  if (variant != 0 && variant != 4 && variant != 5 && variant < 8) {  // explicit instantiation choice (tbg_conv2d_f32_variant: tuning / test aid, stateless)
    if (variant == 1 && p.NJ <= 3) {  // software-pipelined
      if (BM == 32) return launch_fprop<1, 4, 1, 2, 8, MAXTAPS, 3>(p, st, maxtaps, maxTilesN, name);
      if (BM == 64 && BN == 64) return launch_fprop<2, 2, 1, 1, 4, MAXTAPS, 3>(p, st, maxtaps, maxTilesN, name);
      if (BM == 64) return launch_fprop<1, 4, 2, 2, 4, MAXTAPS, 3>(p, st, maxtaps, maxTilesN, name);
      return launch_fprop<2, 2, 2, 2, 4, MAXTAPS, 3>(p, st, maxtaps, maxTilesN, name);
    }
    if (variant == 2) {  // plain CK = 8
      if (BM == 32) return launch_fprop<1, 4, 1, 2, 8, MAXTAPS>(p, st, maxtaps, maxTilesN, name);
      if (BM == 64 && BN == 64) return launch_fprop<2, 2, 1, 1, 8, MAXTAPS>(p, st, maxtaps, maxTilesN, name);
      if (BM == 64) return launch_fprop<1, 4, 2, 2, 8, MAXTAPS>(p, st, maxtaps, maxTilesN, name);
      return launch_fprop<2, 2, 2, 2, 8, MAXTAPS>(p, st, maxtaps, maxTilesN, name);
    }
    if (variant == 3 && BM == 128 && BN == 128 && p.NJ <= 3 && p.ksplit == 1)  // CK = 4 at 4 waves/SIMD
      return launch_fprop<2, 2, 2, 2, 4, MAXTAPS, 0, 4>(p, st, maxtaps, maxTilesN, name);
    if (variant == 7 && BM == 128 && BN == 128)  // register-prefetch K loop
      return launch_fprop<2, 2, 2, 2, 8, MAXTAPS, -1, 3>(p, st, maxtaps, maxTilesN, name);
    return TBG_EUNSUPPORTED;
  }
  // Software-pipelined variant (double-buffered LDS, one barrier per chunk).  Measured (tools/bench_conv.py): +12% on
  // the 64x256 tile (76 -> 86 TFLOP/s), neutral on 128x128, -5..10% on the small-spatial 64x64 tile -> 64x256 only.
  if (p.NJ <= 3 && BM == 64 && BN == 256) {
    // more tiles than the 768 slots of 3 blocks/CU (64x256 64->64 at B = 16: 1024 tiles): the same pipelined instance
    // compiled for 4 blocks/CU fills whole rounds
    if (variant != 1 && (long long)maxTilesN * ceil_div(p.M, BM) * p.nclass > 768 && p.ksplit == 1)
      return launch_fprop<1, 4, 2, 2, 4, MAXTAPS, 3, 4>(p, st, maxtaps, maxTilesN, name);
    return launch_fprop<1, 4, 2, 2, 4, MAXTAPS, 3>(p, st, maxtaps, maxTilesN, name);
  }
  if (BM == 32) return launch_fprop<1, 4, 1, 2, 8, MAXTAPS>(p, st, maxtaps, maxTilesN, name);
  if (BM == 64 && BN == 64) return launch_fprop<2, 2, 1, 1, 8, MAXTAPS>(p, st, maxtaps, maxTilesN, name);
  if (BM == 64) return launch_fprop<1, 4, 2, 2, 8, MAXTAPS>(p, st, maxtaps, maxTilesN, name);
  // Launches of more blocks than the 768 slots that 3 blocks/CU give: CK=4 chunks need 22 KB of LDS and 109 registers
  // -> 4 waves/SIMD, 1024 slots.  98 -> 104 TFLOP/s on 64x256 128->128 (2048 tiles); and a 1024-tile launch (the joint
  // discriminator pass's 32x128 layers at 2B = 32) is ONE full round instead of 768 + 256: 93 -> 126 TFLOP/s
  // (tools/bench_variants_conv.py 32, profiles/r02_conv_variants_f32_b32.txt).  Launches that fit one round of the
  // 3-blocks/CU instance lose with CK=4 (more barriers per FLOP: 116 vs 124 at 512 tiles) and keep CK=8.
  if (maxtaps == 9 && p.NJ <= 3 && (long long)maxTilesN * ceil_div(p.M, BM) * p.nclass > 768 && p.ksplit == 1)
    return launch_fprop<2, 2, 2, 2, 4, MAXTAPS, 0, 4>(p, st, maxtaps, maxTilesN, name);
  // stride-1 layers: the register-prefetch K loop (PF = -1), +3..4 % over the plain loop (profiles/r03_f32_regprefetch.txt)
  if (!d->transposed && d->sy == 1 && d->sx == 1)
    return launch_fprop<2, 2, 2, 2, 8, MAXTAPS, -1, 3>(p, st, maxtaps, maxTilesN, name);
  return launch_fprop<2, 2, 2, 2, 8, MAXTAPS>(p, st, maxtaps, maxTilesN, name);
}

extern "C" int tbg_conv2d_f32(const tbg_conv_desc *d, const float *x, const float *w, float *y,
                              const float *in_scale, const tbg_epilogue *epi, void *stream) {
  return conv2d_impl(d, x, w, y, in_scale, epi, stream, nullptr);
}

extern "C" int tbg_conv2d_f32_variant(const tbg_conv_desc *d, const float *x, const float *w, float *y,
                                      const float *in_scale, const tbg_epilogue *epi, int variant, void *stream) {
  if (variant < 0 || variant > 9) return TBG_EINVAL;
  return conv2d_impl(d, x, w, y, in_scale, epi, stream, nullptr, false, variant);
}

extern "C" int tbg_conv2d_bf16_variant(const tbg_conv_desc *d, const float *x, const void *w, float *y,
                                       const float *in_scale, const tbg_epilogue *epi, int variant, void *stream) {
  if (variant < 0 || variant > 5) return TBG_EINVAL;
  return conv2d_impl(d, x, reinterpret_cast<const float *>(w), y, in_scale, epi, stream, nullptr, 1, variant);
}

extern "C" int tbg_conv2d_bf16(const tbg_conv_desc *d, const float *x, const void *w, float *y, const float *in_scale,
                               const tbg_epilogue *epi, void *stream) {
  return conv2d_impl(d, x, reinterpret_cast<const float *>(w), y, in_scale, epi, stream, nullptr, 1);
}

extern "C" int tbg_conv2d_x3(const tbg_conv_desc *d, const float *x, const void *w, float *y, const float *in_scale,
                             const tbg_epilogue *epi, void *stream) {
  return conv2d_impl(d, x, reinterpret_cast<const float *>(w), y, in_scale, epi, stream, nullptr, 2);
}

extern "C" int tbg_conv2d_x3_variant(const tbg_conv_desc *d, const float *x, const void *w, float *y,
                                     const float *in_scale, const tbg_epilogue *epi, int variant, void *stream) {
  if (variant != 0 && variant != 1 && variant != 2 && variant != 4 && variant != 5) return TBG_EINVAL;
  return conv2d_impl(d, x, reinterpret_cast<const float *>(w), y, in_scale, epi, stream, nullptr, 2, variant);
}

static int conv_name(const tbg_conv_desc *d, int has_in_scale, char *buf, int n, int mode) {
  if (!buf || n < 1) return TBG_EINVAL;
  buf[0] = 0;
  NameOut no{buf, n, nullptr, nullptr};
  static const float dummy = 0.f;  // name-only mode never dereferences; in_scale only sizes the LDS request
  return conv2d_impl(d, nullptr, nullptr, nullptr, has_in_scale ? &dummy : nullptr, nullptr, nullptr, &no, mode);
}

extern "C" int tbg_conv2d_dot_slots(const tbg_conv_desc *d, int has_in_scale, int mode) {
  if (mode < 0 || mode > 2) return TBG_EINVAL;
  int slots = 0;
  NameOut no{nullptr, 0, &slots, nullptr};
  static const float dummy = 0.f;
  const int rc = conv2d_impl(d, nullptr, nullptr, nullptr, has_in_scale ? &dummy : nullptr, nullptr, nullptr, &no, mode);
  return rc != TBG_OK ? rc : slots;
}

extern "C" int tbg_conv2d_blocks(const tbg_conv_desc *d, int has_in_scale, int mode) {
  if (mode < 0 || mode > 2) return TBG_EINVAL;
  int blocks = 0;
  NameOut no{nullptr, 0, nullptr, &blocks};
  static const float dummy = 0.f;
  const int rc = conv2d_impl(d, nullptr, nullptr, nullptr, has_in_scale ? &dummy : nullptr, nullptr, nullptr, &no, mode);
  return rc != TBG_OK ? rc : blocks;
}

extern "C" int tbg_conv2d_kernel_name(const tbg_conv_desc *d, int has_in_scale, char *buf, int n) {
  return conv_name(d, has_in_scale, buf, n, 0);
}

extern "C" int tbg_conv2d_bf16_kernel_name(const tbg_conv_desc *d, int has_in_scale, char *buf, int n) {
  return conv_name(d, has_in_scale, buf, n, 1);
}

extern "C" int tbg_conv2d_x3_kernel_name(const tbg_conv_desc *d, int has_in_scale, char *buf, int n) {
  return conv_name(d, has_in_scale, buf, n, 2);
}

// ============================================================================================
// weight gradient
// ============================================================================================
// ---- float4 staging of one 32 x 2 pixel chunk of a stride-1 3x3 filter gradient (NSEG == 1, Ws % 4 == Wl % 4 == 0, px == 1):
// the S tile and the interior of the 4 x 34 L halo tile are read as float4 (4 + 8 loads per lane instead of 16 + 48 scalars),
// the two halo columns as scalars, all issued together (the compiler merges the two source-level rounds: 241 VGPRs, no
// spill) -- the scalar form needed six dependent round trips per chunk and cost a quarter of the fp32 kernel (tools/archive/exp_wgrad_split.py: 801 us -> 594 us without staging).
// BF: round to bf16 (RNE) on the way into LDS.  Tile pitches are compile-time constants (immediate LDS offsets).
// v_mul_legacy_f32: x * 0 = 0 for EVERY x (Inf and NaN included), IEEE otherwise.  The branch-free staging loads of padding /
// out-of-range positions read a clamped (valid) address whose content is unrelated data; their scale factor is 0, and this
// multiply keeps an Inf sitting there from becoming a NaN in the tile -- at no instruction cost (bf16 kernels; fp32: a select).
__device__ __forceinline__ float zmul_legacy(float a, float b) {
  float r;
  asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// measured: the bf16 kernels (staging bound) gain from the one-instruction form (wgrad 3x3 stride-1 287 -> 338 TFLOP/s vs the
// select form); the fp32 kernel prefers a select the scheduler can move (247 vs 254 us on the 64x256 layer)
template <bool BF> __device__ __forceinline__ float zmul_t(float a, float b) {
  if constexpr (BF) return zmul_legacy(a, b);
  else return b != 0.f ? a * b : 0.f;
}
#define zmul zmul_t<BF>

template <bool BF> struct WgVec {
  static constexpr int SP = BF ? 72 : 68;       // S channel pitch (elements)
  static constexpr int IWP = BF ? 40 : 34;      // L halo row pitch
  static constexpr int LPLANE = BF ? 168 : 137;  // L channel pitch
};

template <bool BF>
__device__ __forceinline__ void wgrad_stage_vec(const WgradP &p, void *Ssv, void *Lsv, int b, int u0, int v0, int cs0, int cl0,
                                                int tid) {
  typedef typename std::conditional<BF, __bf16, float>::type T;
  constexpr int SP = WgVec<BF>::SP, g_IWp = WgVec<BF>::IWP, g_lplane = WgVec<BF>::LPLANE;
  T *Ss = reinterpret_cast<T *>(Ssv), *Ls = reinterpret_cast<T *>(Lsv);
  const int HWs = p.Hs * p.Ws, HWl = p.Hl * p.Wl;
  {
      // scale factors: always loaded (from a valid address), selected afterwards -- no branch around any load
      const bool hs = p.s_scale != nullptr, hl = p.l_scale != nullptr;
      const float *ssp = hs ? p.s_scale : p.S, *lsp = hl ? p.l_scale : p.L;
      // Every address is ONE per-lane base plus a wave-uniform multiple of the channel pitch (lane e = tid + 256 i keeps its
      // position inside the tile and moves 8 / 16 / 32 channels per step): few live registers next to the 144 accumulators.
      const int l_ch = tid >> 5, l_row = (tid >> 3) & 3, l_qx = tid & 7;
      const int l_iy = u0 - p.py + l_row, l_ix = v0 + 4 * l_qx;
      const bool l_in = l_iy >= 0 && l_iy < p.Hl && l_ix < p.Wl;
      const unsigned l_g0 = (unsigned)((b * p.CL + cl0 + l_ch) * HWl + l_iy * p.Wl + l_ix);
      T *l_d0 = Ls + l_ch * g_lplane + l_row * g_IWp + 1 + 4 * l_qx;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        float4 lv[4];
        float lsc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // L interior: 64 channels x 4 halo rows x 8 quads, half of the channels per round
          const int ci = 8 * (4 * hf + i);
          const bool ok = l_in && cl0 + l_ch + ci < p.CL;
          // branch-free: clamp the address, always load, select 0 (an `ok ? load : 0` is compiled to one branch per load)
          lv[i] = *reinterpret_cast<const float4 *>(p.L + (ok ? l_g0 + (unsigned)(ci * HWl) : 0u));
          const float sc = lsp[(hl && ok) ? (unsigned)(b * p.CL + cl0 + l_ch + ci) : 0u];
          lsc[i] = !ok ? 0.f : (hl ? sc : 1.f);  // padding: scale 0, applied with v_mul_legacy (0 * Inf = 0, see zmul)
        }
        if (hf == 0) {
          float4 sv[4];
          float ssc[4];
          const int s_ch = tid >> 4, s_pq = (tid & 15) * 4;
          const int s_u = u0 + (s_pq >> 5), s_v = v0 + (s_pq & 31);
          const bool s_in = s_u < p.Hs && s_v < p.Ws;
          const unsigned s_g0 = (unsigned)((b * p.CS + cs0 + s_ch) * HWs + s_u * p.Ws + s_v);
#pragma unroll
          for (int i = 0; i < 4; ++i) {  // S: 64 channels x 16 quads
            const bool ok = s_in && cs0 + s_ch + 16 * i < p.CS;
            sv[i] = *reinterpret_cast<const float4 *>(p.S + (ok ? s_g0 + (unsigned)(16 * i * HWs) : 0u));
            const float sc = ssp[(hs && ok) ? (unsigned)(b * p.CS + cs0 + s_ch + 16 * i) : 0u];
            ssc[i] = !ok ? 0.f : (hs ? sc : 1.f);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float4 v = sv[i];
            v.x = zmul(v.x, ssc[i]); v.y = zmul(v.y, ssc[i]); v.z = zmul(v.z, ssc[i]); v.w = zmul(v.w, ssc[i]);
            if constexpr (BF) {
              typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
              *reinterpret_cast<bf16x4_t *>(Ss + (s_ch + 16 * i) * SP + s_pq) = bf16x4_t{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
            } else {
              *reinterpret_cast<float4 *>(Ss + (s_ch + 16 * i) * SP + s_pq) = v;
            }
          }
        } else {
          float ev[2], esc[2];
          const int e_ch = tid >> 3, e_row = (tid >> 1) & 3, e_side = tid & 1;
          const int e_iy = u0 - p.py + e_row, e_ix = e_side ? v0 + 32 : v0 - 1;
          const bool e_in = e_iy >= 0 && e_iy < p.Hl && e_ix >= 0 && e_ix < p.Wl;
          const unsigned e_g0 = (unsigned)((b * p.CL + cl0 + e_ch) * HWl + e_iy * p.Wl + e_ix);
#pragma unroll
          for (int i = 0; i < 2; ++i) {  // L halo columns: 64 channels x 4 rows x {left, right}
            const bool ok = e_in && cl0 + e_ch + 32 * i < p.CL;
            ev[i] = p.L[ok ? e_g0 + (unsigned)(32 * i * HWl) : 0u];
            const float sc = lsp[(hl && ok) ? (unsigned)(b * p.CL + cl0 + e_ch + 32 * i) : 0u];
            esc[i] = !ok ? 0.f : (hl ? sc : 1.f);
          }
#pragma unroll
          for (int i = 0; i < 2; ++i)
            Ls[(e_ch + 32 * i) * g_lplane + e_row * g_IWp + (e_side ? 33 : 0)] = (T)zmul(ev[i], esc[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          T *dst = l_d0 + 8 * (4 * hf + i) * g_lplane;
          dst[0] = (T)zmul(lv[i].x, lsc[i]); dst[1] = (T)zmul(lv[i].y, lsc[i]); dst[2] = (T)zmul(lv[i].z, lsc[i]); dst[3] = (T)zmul(lv[i].w, lsc[i]);
        }
        // bf16: keep the two rounds apart (measured: 347 vs 234 TFLOP/s on the 64x256 layer); fp32: the compiler merges them
        if constexpr (BF) __builtin_amdgcn_sched_barrier(0);
      }
  }
}

// ---- the same for one 32 x 1 pixel chunk of a stride-2 VALID 3x3 filter gradient (the discriminator's downsampling
// layers: L = the FIR-padded map, Wl = 2 Ws + 1 or + 2, so L rows are NOT 16-byte aligned -- the float4 loads are
// dword-aligned ones, which global memory accepts).  Per channel 3 halo rows of 65 columns = 16 quads + 1 scalar, stored
// de-interleaved (even | odd columns) as the compute loop expects.  Requires Ws % 32 == 0 (every tile column in range).
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
template <bool BF> struct WgVec2 {
  static constexpr int SP = BF ? 40 : 36;
  static constexpr int HALFW = BF ? 40 : 33;
  static constexpr int IWP = 2 * HALFW;
  static constexpr int LPLANE = BF ? 248 : 199;
};

template <bool BF>
__device__ __forceinline__ void wgrad_stage_vec_s2(const WgradP &p, void *Ssv, void *Lsv, int b, int u0, int v0, int cs0,
                                                   int cl0, int tid) {
  typedef typename std::conditional<BF, __bf16, float>::type T;
  constexpr int SP = WgVec2<BF>::SP, HALFW = WgVec2<BF>::HALFW, IWP = WgVec2<BF>::IWP, LPLANE = WgVec2<BF>::LPLANE;
  T *Ss = reinterpret_cast<T *>(Ssv), *Ls = reinterpret_cast<T *>(Lsv);
  const int HWs = p.Hs * p.Ws, HWl = p.Hl * p.Wl;
  const bool hs = p.s_scale != nullptr, hl = p.l_scale != nullptr;
  const float *ssp = hs ? p.s_scale : p.S, *lsp = hl ? p.l_scale : p.L;
  const int l_ch = tid >> 4, l_qx = tid & 15;
  const int iy0 = 2 * u0 - p.py;
  const unsigned l_g0 = (unsigned)((b * p.CL + cl0 + l_ch) * HWl + max(iy0, 0) * p.Wl + 2 * v0 + 4 * l_qx);
  T *l_d0 = Ls + l_ch * LPLANE + 2 * l_qx;
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    f32x4u lv[6];
    float lsc[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {  // L interior: 3 rows x 64 channels x 16 quads; step i -> row i / 4, channels + 16 (i % 4)
      const int ii = 6 * hf + i, row = ii >> 2, ci = 16 * (ii & 3);
      const bool ok = iy0 + row >= 0 && iy0 + row < p.Hl && cl0 + l_ch + ci < p.CL;
      lv[i] = *reinterpret_cast<const f32x4u *>(p.L + (ok ? l_g0 + (unsigned)(ci * HWl + (row + min(iy0, 0)) * p.Wl) : 0u));
      const float sc = lsp[(hl && ok) ? (unsigned)(b * p.CL + cl0 + l_ch + ci) : 0u];
      lsc[i] = !ok ? 0.f : (hl ? sc : 1.f);
    }
    if (hf == 0) {
      float4 sv[2];
      float ssc[2];
      const int s_ch = tid >> 3, s_pq = (tid & 7) * 4;
      const bool s_in = u0 < p.Hs && v0 + s_pq < p.Ws;
      const unsigned s_g0 = (unsigned)((b * p.CS + cs0 + s_ch) * HWs + u0 * p.Ws + v0 + s_pq);
#pragma unroll
      for (int i = 0; i < 2; ++i) {  // S: 64 channels x 8 quads
        const bool ok = s_in && cs0 + s_ch + 32 * i < p.CS;
        sv[i] = *reinterpret_cast<const float4 *>(p.S + (ok ? s_g0 + (unsigned)(32 * i * HWs) : 0u));
        const float sc = ssp[(hs && ok) ? (unsigned)(b * p.CS + cs0 + s_ch + 32 * i) : 0u];
        ssc[i] = !ok ? 0.f : (hs ? sc : 1.f);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        float4 v = sv[i];
        v.x = zmul(v.x, ssc[i]); v.y = zmul(v.y, ssc[i]); v.z = zmul(v.z, ssc[i]); v.w = zmul(v.w, ssc[i]);
        if constexpr (BF) {
          typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
          *reinterpret_cast<bf16x4_t *>(Ss + (s_ch + 32 * i) * SP + s_pq) = bf16x4_t{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
        } else {
          *reinterpret_cast<float4 *>(Ss + (s_ch + 32 * i) * SP + s_pq) = v;
        }
      }
    } else if (tid < 192) {  // the 65th column of each halo row (even part, place 32): 64 channels x 3 rows
      const int e_ch = tid & 63, e_row = tid >> 6;
      const bool ok = iy0 + e_row >= 0 && iy0 + e_row < p.Hl && cl0 + e_ch < p.CL && 2 * v0 + 64 < p.Wl;
      const float ev = p.L[ok ? (unsigned)((b * p.CL + cl0 + e_ch) * HWl + (iy0 + e_row) * p.Wl + 2 * v0 + 64) : 0u];
      const float sc = lsp[(hl && ok) ? (unsigned)(b * p.CL + cl0 + e_ch) : 0u];
      Ls[e_ch * LPLANE + e_row * IWP + 32] = (T)(!ok ? 0.f : ev * (hl ? sc : 1.f));
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int ii = 6 * hf + i, row = ii >> 2, ci = 16 * (ii & 3);
      T *dst = l_d0 + ci * LPLANE + row * IWP;
      dst[0] = (T)zmul(lv[i][0], lsc[i]); dst[1] = (T)zmul(lv[i][2], lsc[i]);
      dst[HALFW] = (T)zmul(lv[i][1], lsc[i]); dst[HALFW + 1] = (T)zmul(lv[i][3], lsc[i]);
    }
    if constexpr (BF) __builtin_amdgcn_sched_barrier(0);
  }
}

#undef zmul

template <int WGS, int WGL, int NT, int PIX, bool GRP, int VEC = 0>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradP p) {
  constexpr int BS = WGS * 32, BL = WGL * 32, SP = PIX + 4;  // 16-byte aligned S rows; 68 words = conflict-free b128
  constexpr int KWt = (NT == 9) ? 3 : 1;
  constexpr int NJC = (NT == 1 && PIX == 64) ? 1 : WG_MAXNJ;  // halo positions per lane (compile-time trip count: no guards)
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *Ss = smem;            // [BS][SP]
  float *Ls = smem + BS * SP;  // [BL][lplane]

  // tile geometry: compile-time constants on the float4-staged paths (immediate LDS offsets, no address registers)
  const int g_IWp = VEC == 1 ? WgVec<false>::IWP : VEC == 2 ? WgVec2<false>::IWP : p.IWp;
  const int g_lplane = VEC == 1 ? WgVec<false>::LPLANE : VEC == 2 ? WgVec2<false>::LPLANE : p.lplane;
  const int g_logTW = VEC ? 5 : p.logTW, g_logTHs = VEC == 1 ? 1 : VEC == 2 ? 0 : p.logTHs;
  const int g_IHs = VEC == 1 ? 4 : VEC == 2 ? 3 : p.IHs, g_sy = VEC ? VEC : p.sy, g_sx = VEC ? VEC : p.sx;
  const int g_HALFW = VEC == 2 ? WgVec2<false>::HALFW : p.HALFW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ws = wave / WGL, wl = wave - ws * WGL;
  const int cs0 = blockIdx.x * BS, cl0 = blockIdx.y * BL;
  const int TWm = (1 << g_logTW) - 1, THm = (1 << g_logTHs) - 1;

  // position descriptors inside one L channel plane: a wave stages one channel at a time
  int d_pos[NJC], d_loff[NJC];  // d_pos = seg << 16 | iyl << 8 | ixl, or -1
#pragma unroll
  for (int j = 0; j < NJC; ++j) {
    d_pos[j] = -1; d_loff[j] = 0;
    const int e = lane + 64 * j;
    if (!VEC && j < p.NJ && e < p.ppc) {
      const int per = g_IHs * p.IWs;
      const int seg = e / per;
      const int rem = e - seg * per;
      const int iyl = rem / p.IWs;
      const int ixl = rem - iyl * p.IWs;
      d_pos[j] = (seg << 16) | (iyl << 8) | ixl;
      const int col = (g_sx == 2) ? (ixl & 1) * g_HALFW + (ixl >> 1) : ixl;
      d_loff[j] = (seg * g_IHs + iyl) * g_IWp + col;
    }
  }
  // S staging: pixel fixed per thread (256 % PIX == 0)
  const int spix = tid & (PIX - 1);
  const int sch0 = tid / PIX;
  const int sq = spix & TWm, srr = spix >> g_logTW;
  const int sseg = srr >> g_logTHs, sr = srr & THm;

  int toff[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int kh = t / KWt, kw = t - kh * KWt;
    toff[t] = kh * g_IWp + (g_sx == 2 ? (kw & 1) * g_HALFW + (kw >> 1) : kw);
  }

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int HWs = p.Hs * p.Ws, HWl = p.Hl * p.Wl;
  const int half = lane >> 5;

  for (int chunk = blockIdx.z; chunk < p.nchunks; chunk += p.ksplit) {
    const int tv = chunk % p.tilesV;
    const int t2 = chunk / p.tilesV;
    const int tu = t2 % p.tilesU;
    const int bg = t2 / p.tilesU;
    const int u0 = tu << g_logTHs, v0 = tv << g_logTW;
    __syncthreads();
    if constexpr (VEC == 1) {
      static_assert(VEC != 1 || (NT == 9 && PIX == 64 && BS == 64 && BL == 64 && SP == WgVec<false>::SP), "VEC staging: 3x3, 64-pixel chunks, 64x64 tiles");
      wgrad_stage_vec<false>(p, Ss, Ls, bg, u0, v0, cs0, cl0, tid);
    } else if constexpr (VEC == 2) {
      static_assert(VEC != 2 || (NT == 9 && PIX == 32 && BS == 64 && BL == 64 && SP == WgVec2<false>::SP), "stride-2 VEC staging");
      wgrad_stage_vec_s2<false>(p, Ss, Ls, bg, u0, v0, cs0, cl0, tid);
    } else {
    // Branch-free staging (same lesson as conv_fprop_kernel): clamp the address, always load, select 0 -- the loads
    // of a batch (and their scale factors) are then in flight together instead of one round trip per `if` block.
    {  // S tile
      const int b = bg * p.NSEG + sseg, u = u0 + sr, v = v0 + sq;
      const bool ok = b < p.B && u < p.Hs && v < p.Ws;
      const int base = ok ? (b * p.CS) * HWs + u * p.Ws + v : 0;
      const int sb = ok ? b * p.CS : 0;
      constexpr int S_IT = BS / (256 / PIX), S_B = 8;
#pragma unroll
      for (int it0 = 0; it0 < S_IT; it0 += S_B) {
        float sv[S_B], sc[S_B];
#pragma unroll
        for (int u = 0; u < S_B; ++u) {
          const int chc = min(cs0 + sch0 + (it0 + u) * (256 / PIX), p.CS - 1);
          sv[u] = p.S[base + chc * HWs];
          sc[u] = p.s_scale ? p.s_scale[sb + chc] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < S_B; ++u) {
          const int ch = sch0 + (it0 + u) * (256 / PIX);
          Ss[ch * SP + spix] = (ok && cs0 + ch < p.CS) ? sv[u] * sc[u] : 0.f;
        }
      }
    }
    {  // L halo tile: wave w stages channels w, w+4, ...
      int g[NJC], bb[NJC];
#pragma unroll
      for (int j = 0; j < NJC; ++j) {
        const int seg = d_pos[j] >> 16, iyl = (d_pos[j] >> 8) & 255, ixl = d_pos[j] & 255;
        const int b = bg * p.NSEG + seg;
        const int iy = u0 * g_sy - p.py + iyl;
        const int ix = v0 * g_sx - p.px + ixl;
        const bool ok = d_pos[j] >= 0 && b < p.B && iy >= 0 && iy < p.Hl && ix >= 0 && ix < p.Wl;
        g[j] = ok ? (b * p.CL) * HWl + iy * p.Wl + ix : -1;
        bb[j] = ok ? b * p.CL : 0;
      }
      constexpr int LB = 4;  // channels per batch: LB * NJC loads (+ scales) in flight per lane
      for (int ch0 = wave; ch0 < BL; ch0 += 4 * LB) {
        float lv[LB][NJC];
#pragma unroll
        for (int u = 0; u < LB; ++u) {
          const int chc = min(cl0 + ch0 + 4 * u, p.CL - 1);
#pragma unroll
          for (int j = 0; j < NJC; ++j) lv[u][j] = p.L[(g[j] >= 0 ? g[j] : 0) + chc * HWl];
        }
        if (p.l_scale) {  // uniform
          float lsc[LB][NJC];
#pragma unroll
          for (int u = 0; u < LB; ++u) {
            const int chc = min(cl0 + ch0 + 4 * u, p.CL - 1);
#pragma unroll
            for (int j = 0; j < NJC; ++j) lsc[u][j] = p.l_scale[bb[j] + chc];
          }
#pragma unroll
          for (int u = 0; u < LB; ++u)
#pragma unroll
            for (int j = 0; j < NJC; ++j) lv[u][j] *= lsc[u][j];
        }
#pragma unroll
        for (int u = 0; u < LB; ++u) {
          const int ch = ch0 + 4 * u;
#pragma unroll
          for (int j = 0; j < NJC; ++j)
            if (d_pos[j] >= 0 && ch < BL)
              Ls[ch * g_lplane + d_loff[j]] = (g[j] >= 0 && cl0 + ch < p.CL) ? lv[u][j] : 0.f;
        }
      }
    }
    }
    __syncthreads();
    if constexpr (GRP) {
      // (tile rows of >= 4 pixels)  K (= pixels) is walked in groups of 8: half-wave h takes pixels 8g+4h .. 8g+4h+3, so
      // the A operand of 4 k-steps is ONE aligned ds_read_b128 and the B operands are immediate-offset reads from one
      // base address per tap; the reads of k-step i+1 are issued before the MFMAs of k-step i.
      const float *Sp = Ss + (ws * 32 + (lane & 31)) * SP + 4 * half;
      const float *Lp = Ls + (wl * 32 + (lane & 31)) * g_lplane;
#pragma unroll 2
      for (int gp = 0; gp < PIX / 8; ++gp) {
        const int pp = 8 * gp + 4 * half;
        const int q = pp & TWm, rr = pp >> g_logTW;
        const int seg = rr >> g_logTHs, r = rr & THm;
        const float *Lg = Lp + (seg * g_IHs + r * g_sy) * g_IWp + q;
        const f32x4 a4 = *reinterpret_cast<const f32x4 *>(Sp + 8 * gp);
        float bq[2][NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) bq[0][t] = Lg[toff[t]];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (i < 3) {
#pragma unroll
            for (int t = 0; t < NT; ++t) bq[(i + 1) & 1][t] = Lg[toff[t] + i + 1];
          }
#pragma unroll
          for (int t = 0; t < NT; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i], bq[i & 1][t], acc[t], 0, 0, 0);
        }
      }
    } else {  // narrow tiles (Ws <= 2): generic pixel walk
      const float *Sp = Ss + (ws * 32 + (lane & 31)) * SP + half;
      const float *Lp = Ls + (wl * 32 + (lane & 31)) * g_lplane;
#pragma unroll 1
      for (int kp = 0; kp < PIX / 2; ++kp) {
        const int pp = 2 * kp + half;
        const int q = pp & TWm, rr = pp >> g_logTW;
        const int seg = rr >> g_logTHs, r = rr & THm;
        const int poff = (seg * g_IHs + r * g_sy) * g_IWp + q;
        const float a = Sp[2 * kp];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const float bv = Lp[poff + toff[t]];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[t], 0, 0, 0);
        }
      }
    }
  }

  // partial tile -> workspace, fully coalesced: [block][t][r16][wave][lane]
  const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  float *wsp = p.ws + blk * (size_t)(NT * 16 * 256);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r16 = 0; r16 < 16; ++r16) wsp[(t * 16 + r16) * 256 + tid] = acc[t][r16];
}

// ---- bf16-in / fp32-accumulate filter gradient (v_mfma_f32_32x32x16_bf16).  Same block / wave / accumulator structure as
// conv_wgrad_kernel (GRP form); S and L are rounded to bf16 (RNE) while they are staged, K (= pixels) is walked in groups of
// 16: half-wave h takes pixels 16g+8h .. +7 of one tile row, so the A operand is ONE aligned ds_read_b128.  The B
// operand of tap (kh, kw) starts kw pixels further along the row -- a 2-byte shift that a 16-byte LDS read cannot make --
// so per kh ONE 10-pixel window (ds_read_b128 + ds_read_b32) is read and the three kw operands are cut from it in
// registers (kw = 1: four v_alignbit_b32; kw = 2: the dwords one place on).  With x-stride 2 the halo tile is stored
// de-interleaved (even | odd columns): kw = 0 / 2 come from the even window, kw = 1 is an aligned read of the odd one.
// Row pitches are multiples of 8 pixels; channel pitches are ODD multiples of 16 bytes (conflict-free b128 across lanes).
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int WGS, int WGL, int NT, int PIX, int SX, int VEC = 0>
__global__ __launch_bounds__(256, 2) void conv_wgrad_bf16_kernel(const WgradP p) {
  constexpr int BS = WGS * 32, BL = WGL * 32, SPB = PIX + 8;  // S row pitch (bf16 elements): 2*SPB bytes = odd * 16
  constexpr int KHn = (NT == 9) ? 3 : 1;
  constexpr int NJC = (NT == 1 && PIX == 64) ? 1 : WG_MAXNJ;
  static_assert(((SPB / 8) & 1) == 1, "S channel pitch must be an odd multiple of 16 bytes");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __bf16 *Ss = reinterpret_cast<__bf16 *>(smem);  // [BS][SPB]
  __bf16 *Ls = Ss + BS * SPB;                      // [BL][lplane]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ws = wave / WGL, wl = wave - ws * WGL;
  const int cs0 = blockIdx.x * BS, cl0 = blockIdx.y * BL;
  const int TWm = (1 << p.logTW) - 1, THm = (1 << p.logTHs) - 1;

  int d_pos[NJC], d_loff[NJC];
#pragma unroll
  for (int j = 0; j < NJC; ++j) {
    d_pos[j] = -1; d_loff[j] = 0;
    const int e = lane + 64 * j;
    if (!VEC && j < p.NJ && e < p.ppc) {
      const int per = p.IHs * p.IWs;
      const int seg = e / per;
      const int rem = e - seg * per;
      const int iyl = rem / p.IWs;
      const int ixl = rem - iyl * p.IWs;
      d_pos[j] = (seg << 16) | (iyl << 8) | ixl;
      const int col = (SX == 2) ? (ixl & 1) * p.HALFW + (ixl >> 1) : ixl;
      d_loff[j] = (seg * p.IHs + iyl) * p.IWp + col;
    }
  }
  const int spix = tid & (PIX - 1);
  const int sch0 = tid / PIX;
  const int sq = spix & TWm, srr = spix >> p.logTW;
  const int sseg = srr >> p.logTHs, sr = srr & THm;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  const int HWs = p.Hs * p.Ws, HWl = p.Hl * p.Wl;
  const int half = lane >> 5;

  for (int chunk = blockIdx.z; chunk < p.nchunks; chunk += p.ksplit) {
    const int tv = chunk % p.tilesV;
    const int t2 = chunk / p.tilesV;
    const int tu = t2 % p.tilesU;
    const int bg = t2 / p.tilesU;
    const int u0 = tu << p.logTHs, v0 = tv << p.logTW;
    __syncthreads();
    if constexpr (VEC == 1) {
      static_assert(VEC != 1 || (NT == 9 && PIX == 64 && SX == 1 && BS == 64 && BL == 64 && SPB == WgVec<true>::SP), "VEC staging");
      wgrad_stage_vec<true>(p, Ss, Ls, bg, u0, v0, cs0, cl0, tid);
    } else if constexpr (VEC == 2) {
      static_assert(VEC != 2 || (NT == 9 && PIX == 32 && SX == 2 && BS == 64 && BL == 64 && SPB == WgVec2<true>::SP), "stride-2 VEC staging");
      wgrad_stage_vec_s2<true>(p, Ss, Ls, bg, u0, v0, cs0, cl0, tid);
    } else {
    {  // S tile (branch-free, as in the fp32 kernel)
      const int b = bg * p.NSEG + sseg, u = u0 + sr, v = v0 + sq;
      const bool ok = b < p.B && u < p.Hs && v < p.Ws;
      const int base = ok ? (b * p.CS) * HWs + u * p.Ws + v : 0;
      const int sb = ok ? b * p.CS : 0;
      constexpr int S_IT = BS / (256 / PIX), S_B = 8;
#pragma unroll
      for (int it0 = 0; it0 < S_IT; it0 += S_B) {
        float sv[S_B], sc[S_B];
#pragma unroll
        for (int u = 0; u < S_B; ++u) {
          const int chc = min(cs0 + sch0 + (it0 + u) * (256 / PIX), p.CS - 1);
          sv[u] = p.S[base + chc * HWs];
          sc[u] = p.s_scale ? p.s_scale[sb + chc] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < S_B; ++u) {
          const int ch = sch0 + (it0 + u) * (256 / PIX);
          Ss[ch * SPB + spix] = (__bf16)((ok && cs0 + ch < p.CS) ? sv[u] * sc[u] : 0.f);
        }
      }
    }
    {  // L halo tile: wave w stages channels w, w+4, ...
      int g[NJC], bb[NJC];
#pragma unroll
      for (int j = 0; j < NJC; ++j) {
        const int seg = d_pos[j] >> 16, iyl = (d_pos[j] >> 8) & 255, ixl = d_pos[j] & 255;
        const int b = bg * p.NSEG + seg;
        const int iy = u0 * p.sy - p.py + iyl;
        const int ix = v0 * p.sx - p.px + ixl;
        const bool ok = d_pos[j] >= 0 && b < p.B && iy >= 0 && iy < p.Hl && ix >= 0 && ix < p.Wl;
        g[j] = ok ? (b * p.CL) * HWl + iy * p.Wl + ix : -1;
        bb[j] = ok ? b * p.CL : 0;
      }
      constexpr int LB = 4;
      for (int ch0 = wave; ch0 < BL; ch0 += 4 * LB) {
        float lv[LB][NJC];
#pragma unroll
        for (int u = 0; u < LB; ++u) {
          const int chc = min(cl0 + ch0 + 4 * u, p.CL - 1);
#pragma unroll
          for (int j = 0; j < NJC; ++j) lv[u][j] = p.L[(g[j] >= 0 ? g[j] : 0) + chc * HWl];
        }
        if (p.l_scale) {  // uniform
          float lsc[LB][NJC];
#pragma unroll
          for (int u = 0; u < LB; ++u) {
            const int chc = min(cl0 + ch0 + 4 * u, p.CL - 1);
#pragma unroll
            for (int j = 0; j < NJC; ++j) lsc[u][j] = p.l_scale[bb[j] + chc];
          }
#pragma unroll
          for (int u = 0; u < LB; ++u)
#pragma unroll
            for (int j = 0; j < NJC; ++j) lv[u][j] *= lsc[u][j];
        }
#pragma unroll
        for (int u = 0; u < LB; ++u) {
          const int ch = ch0 + 4 * u;
#pragma unroll
          for (int j = 0; j < NJC; ++j)
            if (d_pos[j] >= 0 && ch < BL)
              Ls[ch * p.lplane + d_loff[j]] = (__bf16)((g[j] >= 0 && cl0 + ch < p.CL) ? lv[u][j] : 0.f);
        }
      }
    }
    }
    __syncthreads();
    const __bf16 *Sp = Ss + (ws * 32 + (lane & 31)) * SPB + 8 * half;
    const __bf16 *Lp = Ls + (wl * 32 + (lane & 31)) * p.lplane;
#pragma unroll 2
    for (int gp = 0; gp < PIX / 16; ++gp) {
      const int pp = 16 * gp + 8 * half;  // first of this half-wave's 8 pixels (one tile row: TW >= 8)
      const int q = pp & TWm, rr = pp >> p.logTW;
      const int seg = rr >> p.logTHs, r = rr & THm;
      const __bf16 *Lg = Lp + (seg * p.IHs + r * p.sy) * p.IWp + q;  // multiple of 8 elements: 16-byte aligned
      const bf16x8 a8 = *reinterpret_cast<const bf16x8 *>(Sp + 16 * gp);
#pragma unroll
      for (int kh = 0; kh < KHn; ++kh) {
        const __bf16 *row = Lg + kh * p.IWp;
        const i32x4 e = *reinterpret_cast<const i32x4 *>(row);  // pixels 0..7 (even columns when SX == 2)
        if constexpr (NT == 1) {
          acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, __builtin_bit_cast(bf16x8, e), acc[0], 0, 0, 0);
        } else {
          const int e4 = *reinterpret_cast<const int *>(row + 8);  // pixels 8, 9
          i32x4 s1, s2;
          s1[0] = __builtin_amdgcn_alignbit(e[1], e[0], 16); s1[1] = __builtin_amdgcn_alignbit(e[2], e[1], 16);
          s1[2] = __builtin_amdgcn_alignbit(e[3], e[2], 16); s1[3] = __builtin_amdgcn_alignbit(e4, e[3], 16);
          s2[0] = e[1]; s2[1] = e[2]; s2[2] = e[3]; s2[3] = e4;
          i32x4 b0 = e, b1, b2;
          if constexpr (SX == 2) {  // kw = 0: even[0..7], kw = 1: odd[0..7], kw = 2: even[1..8]
            b1 = *reinterpret_cast<const i32x4 *>(row + p.HALFW);
            b2 = s1;
          } else {                  // kw = 0, 1, 2: window[0..7], [1..8], [2..9]
            b1 = s1;
            b2 = s2;
          }
          acc[3 * kh + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, __builtin_bit_cast(bf16x8, b0), acc[3 * kh + 0], 0, 0, 0);
          acc[3 * kh + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, __builtin_bit_cast(bf16x8, b1), acc[3 * kh + 1], 0, 0, 0);
          acc[3 * kh + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, __builtin_bit_cast(bf16x8, b2), acc[3 * kh + 2], 0, 0, 0);
        }
      }
    }
  }

  const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  float *wsp = p.ws + blk * (size_t)(NT * 16 * 256);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r16 = 0; r16 < 16; ++r16) wsp[(t * 16 + r16) * 256 + tid] = acc[t][r16];
}

// ---- f32x3 filter gradient (tbg.h "f32x3 forms"): S and L split into three bf16 terms each while they are staged (three LDS
// planes per operand), six partial products per tap on v_mfma_f32_32x32x16_bf16 (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi;
// smallest first), fp32 accumulate.  Same block / wave / accumulator structure and the same float4-staged tile geometries as
// conv_wgrad_bf16_kernel's VEC forms (VEC = 1: stride-1 3x3, 32 x 2-pixel chunks; VEC = 2: stride-2 VALID 3x3, 32 x 1).  Three
// planes of both tiles take 92 / 111 KB of LDS -> ONE block per CU (one wave per SIMD, up to 512 registers), so the overlap
// that co-resident blocks give the other kernels is built into the wave instead: the global loads of chunk k+1 are issued into
// registers before the MFMA phase of chunk k (54 MFMAs per 16 pixels and wave cover the round trip) and split + stored after it.
template <int VEC> struct WgX3 {
  static constexpr int PIX = VEC == 1 ? 64 : 32;
  static constexpr int SPB = VEC == 1 ? WgVec<true>::SP : WgVec2<true>::SP;
  static constexpr int IWP = VEC == 1 ? WgVec<true>::IWP : WgVec2<true>::IWP;
  static constexpr int LPLANE = VEC == 1 ? WgVec<true>::LPLANE : WgVec2<true>::LPLANE;
  static constexpr int HALFW = VEC == 1 ? 0 : WgVec2<true>::HALFW;
  static constexpr int NL = VEC == 1 ? 8 : 12;  // float4 loads of the L interior per lane
  static constexpr int NS = VEC == 1 ? 4 : 2;   // float4 loads of S per lane
  static constexpr int NE = VEC == 1 ? 2 : 1;   // scalar halo-column loads per lane
};

template <int VEC> struct WgX3Regs {
  f32x4u lv[WgX3<VEC>::NL];
  float lsc[WgX3<VEC>::NL];
  float4 sv[WgX3<VEC>::NS];
  float ssc[WgX3<VEC>::NS];
  float ev[WgX3<VEC>::NE], esc[WgX3<VEC>::NE];
  // validity bits of the loads above and the row-edge flags of the L quads.  wgrad_x3_load only ISSUES loads and sets these
  // bits; every operation on a loaded value (scale select, the edge shifts) waits for wgrad_x3_store -- a select on a loaded
  // register inside the load function made the compiler wait for the loads right there (vmcnt(18) .. vmcnt(0) in front of the
  // MFMA phase), i.e. nothing was in flight under the MFMAs (tools/archive/exp_wgx3_split.py: the loads cost 90-140 us of 530).
  unsigned lok, sok, eok, edge;  // edge: bit 0 = first quad starts before the row, bit 1 = last column alone
};

__device__ __forceinline__ void split3_store(__bf16 *dst, int plane_stride, float v) {
  const __bf16 bh = (__bf16)v;
  const float r1 = v - (float)bh;
  const __bf16 bm = (__bf16)r1;
  dst[0] = bh; dst[plane_stride] = bm; dst[2 * plane_stride] = (__bf16)(r1 - (float)bm);
}
typedef __bf16 bf16x4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
// four / two consecutive places: one 8-byte / 4-byte store per plane
__device__ __forceinline__ void split3_store4(__bf16 *dst, int plane_stride, float v0, float v1, float v2, float v3) {
  unsigned h0, m0, l0, h1, m1, l1;
  split3_pair(v0, v1, h0, m0, l0);
  split3_pair(v2, v3, h1, m1, l1);
  *reinterpret_cast<u32x2v *>(dst) = u32x2v{h0, h1};
  *reinterpret_cast<u32x2v *>(dst + plane_stride) = u32x2v{m0, m1};
  *reinterpret_cast<u32x2v *>(dst + 2 * plane_stride) = u32x2v{l0, l1};
}
__device__ __forceinline__ void split3_store2(__bf16 *dst, int plane_stride, float v0, float v1) {
  unsigned h, m, l;
  split3_pair(v0, v1, h, m, l);
  *reinterpret_cast<unsigned *>(dst) = h;
  *reinterpret_cast<unsigned *>(dst + plane_stride) = m;
  *reinterpret_cast<unsigned *>(dst + 2 * plane_stride) = l;
}

// issue every global load of one chunk (branch-free: clamped addresses, scale factor 0 for padding / out-of-range)
template <int VEC>
__device__ __forceinline__ void wgrad_x3_load(const WgradP &p, WgX3Regs<VEC> &r, int b, int u0, int v0, int cs0, int cl0, int tid) {
  const int HWs = p.Hs * p.Ws, HWl = p.Hl * p.Wl;
  const bool hs = p.s_scale != nullptr, hl = p.l_scale != nullptr;
  const float *ssp = hs ? p.s_scale : p.S, *lsp = hl ? p.l_scale : p.L;
  if constexpr (VEC == 1) {
    // L halo tile, rows of 34 (x = v0 - 1 .. v0 + 32 at places 0 .. 33): each lane loads the FOUR PLACES 4q .. 4q + 3, i.e.
    // x = v0 + 4q - 1 .. + 2 -- a dword-aligned float4 -- so that its three bf16 planes are 8-byte LDS stores (the x-aligned
    // quads of the bf16 kernel land on odd places and cost four 2-byte stores each).  The very first quad of a row would
    // start one float before the row (x = -1): it loads x = 0 .. 3 instead and is shifted right by one place in registers.
    const int l_ch = tid >> 5, l_row = (tid >> 3) & 3, l_qx = tid & 7;
    // The quad of the last column alone (x = Wl - 1, a ragged last tile) would run past the row: it loads x - 3 .. x and keeps
    // its last element.  No lane reads outside the row it belongs to.
    const int l_iy = u0 - p.py + l_row, l_ix = v0 + 4 * l_qx - 1;
    const bool l_sh = l_ix < 0, l_last = l_ix == p.Wl - 1;
    const bool l_in = l_iy >= 0 && l_iy < p.Hl && l_ix < p.Wl;
    const unsigned l_g0 = (unsigned)((b * p.CL + cl0 + l_ch) * HWl + l_iy * p.Wl + l_ix + (l_sh ? 1 : l_last ? -3 : 0));
    r.edge = (l_sh ? 1u : 0u) | (l_last ? 2u : 0u);
    r.lok = 0; r.sok = 0; r.eok = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bool ok = l_in && cl0 + l_ch + 8 * i < p.CL;
      r.lok |= ok ? (1u << i) : 0u;
      r.lv[i] = *reinterpret_cast<const f32x4u *>(p.L + (ok ? l_g0 + (unsigned)(8 * i * HWl) : 0u));
      if (hl) r.lsc[i] = lsp[ok ? (unsigned)(b * p.CL + cl0 + l_ch + 8 * i) : 0u];
    }
    const int s_ch = tid >> 4, s_pq = (tid & 15) * 4;
    const int s_u = u0 + (s_pq >> 5), s_v = v0 + (s_pq & 31);
    const bool s_in = s_u < p.Hs && s_v < p.Ws;
    const unsigned s_g0 = (unsigned)((b * p.CS + cs0 + s_ch) * HWs + s_u * p.Ws + s_v);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = s_in && cs0 + s_ch + 16 * i < p.CS;
      r.sok |= ok ? (1u << i) : 0u;
      r.sv[i] = *reinterpret_cast<const float4 *>(p.S + (ok ? s_g0 + (unsigned)(16 * i * HWs) : 0u));
      if (hs) r.ssc[i] = ssp[ok ? (unsigned)(b * p.CS + cs0 + s_ch + 16 * i) : 0u];
    }
    // the two places the quads leave: 32 and 33 (x = v0 + 31, v0 + 32)
    const int e_ch = tid >> 3, e_row = (tid >> 1) & 3, e_side = tid & 1;
    const int e_iy = u0 - p.py + e_row, e_ix = v0 + 31 + e_side;
    const bool e_in = e_iy >= 0 && e_iy < p.Hl && e_ix < p.Wl;
    const unsigned e_g0 = (unsigned)((b * p.CL + cl0 + e_ch) * HWl + e_iy * p.Wl + e_ix);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool ok = e_in && cl0 + e_ch + 32 * i < p.CL;
      r.eok |= ok ? (1u << i) : 0u;
      r.ev[i] = p.L[ok ? e_g0 + (unsigned)(32 * i * HWl) : 0u];
      if (hl) r.esc[i] = lsp[ok ? (unsigned)(b * p.CL + cl0 + e_ch + 32 * i) : 0u];
    }
  } else {
    r.edge = 0; r.lok = 0; r.sok = 0; r.eok = 0;
    const int l_ch = tid >> 4, l_qx = tid & 15;
    const int iy0 = 2 * u0 - p.py;
    const unsigned l_g0 = (unsigned)((b * p.CL + cl0 + l_ch) * HWl + max(iy0, 0) * p.Wl + 2 * v0 + 4 * l_qx);
#pragma unroll
    for (int ii = 0; ii < 12; ++ii) {
      const int row = ii >> 2, ci = 16 * (ii & 3);
      const bool ok = iy0 + row >= 0 && iy0 + row < p.Hl && cl0 + l_ch + ci < p.CL;
      r.lok |= ok ? (1u << ii) : 0u;
      r.lv[ii] = *reinterpret_cast<const f32x4u *>(p.L + (ok ? l_g0 + (unsigned)(ci * HWl + (row + min(iy0, 0)) * p.Wl) : 0u));
      if (hl) r.lsc[ii] = lsp[ok ? (unsigned)(b * p.CL + cl0 + l_ch + ci) : 0u];
    }
    const int s_ch = tid >> 3, s_pq = (tid & 7) * 4;
    const bool s_in = u0 < p.Hs && v0 + s_pq < p.Ws;
    const unsigned s_g0 = (unsigned)((b * p.CS + cs0 + s_ch) * HWs + u0 * p.Ws + v0 + s_pq);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool ok = s_in && cs0 + s_ch + 32 * i < p.CS;
      r.sok |= ok ? (1u << i) : 0u;
      r.sv[i] = *reinterpret_cast<const float4 *>(p.S + (ok ? s_g0 + (unsigned)(32 * i * HWs) : 0u));
      if (hs) r.ssc[i] = ssp[ok ? (unsigned)(b * p.CS + cs0 + s_ch + 32 * i) : 0u];
    }
    {  // the 65th column of each halo row (even part, place 32): 64 channels x 3 rows, lanes 0..191
      const int e_ch = tid & 63, e_row = min(tid >> 6, 2);
      const bool ok = tid < 192 && iy0 + e_row >= 0 && iy0 + e_row < p.Hl && cl0 + e_ch < p.CL && 2 * v0 + 64 < p.Wl;
      r.eok = ok ? 1u : 0u;
      r.ev[0] = p.L[ok ? (unsigned)((b * p.CL + cl0 + e_ch) * HWl + (iy0 + e_row) * p.Wl + 2 * v0 + 64) : 0u];
      if (hl) r.esc[0] = lsp[ok ? (unsigned)(b * p.CL + cl0 + e_ch) : 0u];
    }
  }
}

// scale, split into hi | mid | lo and store the chunk's tiles (three planes each)
template <int VEC>
__device__ __forceinline__ void wgrad_x3_store(const WgradP &p, WgX3Regs<VEC> &r, __bf16 *Ss, __bf16 *Ls, int tid) {
  constexpr int SPB = WgX3<VEC>::SPB, IWP = WgX3<VEC>::IWP, LPLANE = WgX3<VEC>::LPLANE, HALFW = WgX3<VEC>::HALFW;
  constexpr int SPL = 64 * SPB, LPL = 64 * LPLANE;  // plane strides (elements)
  {  // first touch of the loaded values: scale factors (0 for padding / out of range) and the row-edge shifts
    const bool hs = p.s_scale != nullptr, hl = p.l_scale != nullptr;
#pragma unroll
    for (int i = 0; i < WgX3<VEC>::NL; ++i) r.lsc[i] = !((r.lok >> i) & 1u) ? 0.f : (hl ? r.lsc[i] : 1.f);
    if constexpr (VEC == 1) {
      // the row-edge shifts touch the first / last column tiles only: a wave-uniform test skips their 8 selects per quad on
      // every other chunk (7 of 8 on 256-wide maps)
      if (__builtin_amdgcn_ballot_w64(r.edge != 0u) != 0ull) {
#pragma unroll
        for (int i = 0; i < WgX3<VEC>::NL; ++i) {
          f32x4u v = r.lv[i];
          if (r.edge & 1u) { v[3] = v[2]; v[2] = v[1]; v[1] = v[0]; v[0] = 0.f; }
          if (r.edge & 2u) { v[0] = v[3]; v[1] = 0.f; v[2] = 0.f; v[3] = 0.f; }
          r.lv[i] = v;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < WgX3<VEC>::NS; ++i) r.ssc[i] = !((r.sok >> i) & 1u) ? 0.f : (hs ? r.ssc[i] : 1.f);
#pragma unroll
    for (int i = 0; i < WgX3<VEC>::NE; ++i) r.esc[i] = !((r.eok >> i) & 1u) ? 0.f : (hl ? r.esc[i] : 1.f);
  }
  if constexpr (VEC == 1) {
    const int l_ch = tid >> 5, l_row = (tid >> 3) & 3, l_qx = tid & 7;
    __bf16 *l_d0 = Ls + l_ch * LPLANE + l_row * IWP + 4 * l_qx;  // places 4q .. 4q + 3: 8-byte aligned (IWP, LPLANE % 4 == 0)
#pragma unroll
    for (int i = 0; i < 8; ++i)
      split3_store4(l_d0 + 8 * i * LPLANE, LPL, zmul_legacy(r.lv[i][0], r.lsc[i]), zmul_legacy(r.lv[i][1], r.lsc[i]),
                    zmul_legacy(r.lv[i][2], r.lsc[i]), zmul_legacy(r.lv[i][3], r.lsc[i]));
    const int s_ch = tid >> 4, s_pq = (tid & 15) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      split3_store4(Ss + (s_ch + 16 * i) * SPB + s_pq, SPL, zmul_legacy(r.sv[i].x, r.ssc[i]), zmul_legacy(r.sv[i].y, r.ssc[i]),
                    zmul_legacy(r.sv[i].z, r.ssc[i]), zmul_legacy(r.sv[i].w, r.ssc[i]));
    const int e_ch = tid >> 3, e_row = (tid >> 1) & 3, e_side = tid & 1;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      split3_store(Ls + (e_ch + 32 * i) * LPLANE + e_row * IWP + 32 + e_side, LPL, zmul_legacy(r.ev[i], r.esc[i]));
  } else {
    const int l_ch = tid >> 4, l_qx = tid & 15;
    __bf16 *l_d0 = Ls + l_ch * LPLANE + 2 * l_qx;
#pragma unroll
    for (int ii = 0; ii < 12; ++ii) {
      const int row = ii >> 2, ci = 16 * (ii & 3);
      __bf16 *dst = l_d0 + ci * LPLANE + row * IWP;  // even | odd columns, two places each: 4-byte stores
      split3_store2(dst, LPL, zmul_legacy(r.lv[ii][0], r.lsc[ii]), zmul_legacy(r.lv[ii][2], r.lsc[ii]));
      split3_store2(dst + HALFW, LPL, zmul_legacy(r.lv[ii][1], r.lsc[ii]), zmul_legacy(r.lv[ii][3], r.lsc[ii]));
    }
    const int s_ch = tid >> 3, s_pq = (tid & 7) * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      split3_store4(Ss + (s_ch + 32 * i) * SPB + s_pq, SPL, zmul_legacy(r.sv[i].x, r.ssc[i]), zmul_legacy(r.sv[i].y, r.ssc[i]),
                    zmul_legacy(r.sv[i].z, r.ssc[i]), zmul_legacy(r.sv[i].w, r.ssc[i]));
    if (tid < 192) split3_store(Ls + (tid & 63) * LPLANE + (tid >> 6) * IWP + 32, LPL, zmul_legacy(r.ev[0], r.esc[0]));
  }
}

template <int VEC>
__global__ __launch_bounds__(256, 1) void conv_wgrad_x3_kernel(const WgradP p) {
  constexpr int PIX = WgX3<VEC>::PIX, SPB = WgX3<VEC>::SPB, IWP = WgX3<VEC>::IWP, LPLANE = WgX3<VEC>::LPLANE;
  constexpr int HALFW = WgX3<VEC>::HALFW, SX = VEC, NT = 9;
  constexpr int SPL = 64 * SPB, LPL = 64 * LPLANE;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __bf16 *Ss = reinterpret_cast<__bf16 *>(smem);  // [3][64][SPB]
  __bf16 *Ls = Ss + 3 * SPL;                       // [3][64][LPLANE]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ws = wave >> 1, wl = wave & 1;
  const int cs0 = blockIdx.x * 64, cl0 = blockIdx.y * 64;
  const int half = lane >> 5;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  auto chunk_pos = [&](int chunk, int &bg, int &u0, int &v0) {
    const int tv = chunk % p.tilesV;
    const int t2 = chunk / p.tilesV;
    const int tu = t2 % p.tilesU;
    bg = t2 / p.tilesU;
    u0 = tu << (VEC == 1 ? 1 : 0); v0 = tv << 5;
  };
  WgX3Regs<VEC> rg = {};  // (scale slots stay unread when the operand has no scale)
  int chunk = blockIdx.z;
  if (chunk < p.nchunks) {
    int bg, u0, v0;
    chunk_pos(chunk, bg, u0, v0);
    wgrad_x3_load<VEC>(p, rg, bg, u0, v0, cs0, cl0, tid);
  }
  const __bf16 *Sp = Ss + (ws * 32 + (lane & 31)) * SPB + 8 * half;
  const __bf16 *Lp = Ls + (wl * 32 + (lane & 31)) * LPLANE;
  for (; chunk < p.nchunks; chunk += p.ksplit) {
    __syncthreads();  // the previous chunk's MFMA phase is done with the tiles
    wgrad_x3_store<VEC>(p, rg, Ss, Ls, tid);
    __syncthreads();
    if (chunk + p.ksplit < p.nchunks) {  // next chunk's loads: in flight under the MFMA phase below
      int bg, u0, v0;
      chunk_pos(chunk + p.ksplit, bg, u0, v0);
      wgrad_x3_load<VEC>(p, rg, bg, u0, v0, cs0, cl0, tid);
    }
    // MFMA phase as NG x 3 steps of (16-pixel group, filter row): the LDS reads of step i+1 are issued before the 18 MFMAs of
    // step i (two small register sets, order pinned with sched_barrier).  With ONE wave per SIMD nothing else covers an LDS round
    // trip: tools/archive/exp_wgx3_split.py measured the phase at 400 us with its operand reads and 255 us without (64x256 layer) when
    // the compiler batched a group's 21 reads in front of its 54 MFMAs.  (Double-buffering whole GROUPS instead cost 256 VGPRs
    // + accumulator-file copies and ran slower: 103 vs 140 TFLOP/s.)
    // The stride-2 form (two groups per chunk, a fourth read per plane) measured 6 % SLOWER this way (105 vs 114 TFLOP/s) and
    // keeps the rolled loop below.
    if constexpr (VEC == 1) {
    constexpr int NG = PIX / 16, NST = NG * 3;
    bf16x8 a[2][3];                   // [group parity][plane]
    i32x4 we[2][3], wo[2][3];         // [step parity][plane]: the 8-pixel window (even columns when SX == 2) / the odd columns
    int w4[2][3];                     // pixels 8, 9 of the window
    auto ld = [&](int st, int bs) {
      const int gp = st / 3, kh = st - 3 * gp;
      const int pp = 16 * gp + 8 * half;  // first of this half-wave's 8 pixels (one 32-pixel tile row)
      const __bf16 *Lg = Lp + (VEC == 1 ? (pp >> 5) * IWP : 0) + (pp & 31) + kh * IWP;
      if (kh == 0) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          a[gp & 1][pl] = *reinterpret_cast<const bf16x8 *>(__builtin_assume_aligned(Sp + pl * SPL + 16 * gp, 16));
      }
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {  // window starts are multiples of 8 pixels and every pitch a multiple of 8: 16-byte
        const __bf16 *row = Lg + pl * LPL;  // aligned (said explicitly: unproven, the compiler split the reads into ds_read2_b32)
        we[bs][pl] = *reinterpret_cast<const i32x4 *>(__builtin_assume_aligned(row, 16));
        w4[bs][pl] = *reinterpret_cast<const int *>(row + 8);
        if constexpr (SX == 2) wo[bs][pl] = *reinterpret_cast<const i32x4 *>(__builtin_assume_aligned(row + HALFW, 16));
      }
    };
    ld(0, 0);
#pragma unroll
    for (int st = 0; st < NST; ++st) {
      const int bs = st & 1, gp = st / 3, kh = st - 3 * gp;
      if (st + 1 < NST) ld(st + 1, bs ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      i32x4 b[3][3];  // [plane][kw]
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        const i32x4 e = we[bs][pl];
        const int e4 = w4[bs][pl];
        i32x4 s1;
        s1[0] = __builtin_amdgcn_alignbit(e[1], e[0], 16); s1[1] = __builtin_amdgcn_alignbit(e[2], e[1], 16);
        s1[2] = __builtin_amdgcn_alignbit(e[3], e[2], 16); s1[3] = __builtin_amdgcn_alignbit(e4, e[3], 16);
        b[pl][0] = e;
        if constexpr (SX == 2) {  // kw = 0: even[0..7], kw = 1: odd[0..7], kw = 2: even[1..8]
          b[pl][1] = wo[bs][pl];
          b[pl][2] = s1;
        } else {                  // kw = 0, 1, 2: window[0..7], [1..8], [2..9]
          b[pl][1] = s1;
          b[pl][2][0] = e[1]; b[pl][2][1] = e[2]; b[pl][2][2] = e[3]; b[pl][2][3] = e4;
        }
      }
      // six partial products per tap, smallest first: (hi,lo) (lo,hi) (mid,mid) (hi,mid) (mid,hi) (hi,hi)
      constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
      for (int q = 0; q < 6; ++q)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
          acc[3 * kh + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[gp & 1][PA[q]], __builtin_bit_cast(bf16x8, b[PB[q]][kw]),
                                                                     acc[3 * kh + kw], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    } else {
#pragma unroll 1
    for (int gp = 0; gp < PIX / 16; ++gp) {
      const int pp = 16 * gp + 8 * half;
      const __bf16 *Lg = Lp + (pp & 31);
      bf16x8 a[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) a[pl] = *reinterpret_cast<const bf16x8 *>(Sp + pl * SPL + 16 * gp);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        i32x4 b[3][3];  // [plane][kw]: even[0..7], odd[0..7], even[1..8]
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          const __bf16 *row = Lg + pl * LPL + kh * IWP;
          const i32x4 e = *reinterpret_cast<const i32x4 *>(row);
          const int e4 = *reinterpret_cast<const int *>(row + 8);
          b[pl][0] = e;
          b[pl][1] = *reinterpret_cast<const i32x4 *>(row + HALFW);
          b[pl][2][0] = __builtin_amdgcn_alignbit(e[1], e[0], 16); b[pl][2][1] = __builtin_amdgcn_alignbit(e[2], e[1], 16);
          b[pl][2][2] = __builtin_amdgcn_alignbit(e[3], e[2], 16); b[pl][2][3] = __builtin_amdgcn_alignbit(e4, e[3], 16);
        }
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
          for (int kw = 0; kw < 3; ++kw)
            acc[3 * kh + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[q]], __builtin_bit_cast(bf16x8, b[PB[q]][kw]),
                                                                       acc[3 * kh + kw], 0, 0, 0);
      }
    }
    }
  }

  const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  float *wsp = p.ws + blk * (size_t)(NT * 16 * 256);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r16 = 0; r16 < 16; ++r16) wsp[(t * 16 + r16) * 256 + tid] = acc[t][r16];
}

template <int WGS, int WGL, int NT, int PIX, bool GRP, int VEC = 0>
static int launch_wgrad_impl(WgradP &p, hipStream_t st, size_t ws_bytes, const NameOut *name) {
  constexpr int BS = WGS * 32, BL = WGL * 32;
  if (name) {
    snprintf(name->buf, name->n, "conv_wgrad_kernel<%d, %d, %d, %d, %s, %d>", WGS, WGL, NT, PIX, GRP ? "true" : "false", VEC);
    return TBG_OK;
  }
  const size_t lds = ((size_t)BS * (PIX + 4) + (size_t)BL * p.lplane) * sizeof(float);
  if (p.NJ > ((NT == 1 && PIX == 64) ? 1 : WG_MAXNJ)) return TBG_EUNSUPPORTED;
  if (lds > 160 * 1024) return TBG_EUNSUPPORTED;
  auto kern = conv_wgrad_kernel<WGS, WGL, NT, PIX, GRP, VEC>;
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return TBG_EHIP;
  }
  const int tx = ceil_div(p.CS, BS), ty = ceil_div(p.CL, BL);
  p.ksplit = wgrad_ksplit(tx * ty, p.nchunks);
  if ((size_t)p.ksplit * tx * ty * NT * 16 * 256 * sizeof(float) > ws_bytes) return TBG_EINVAL;
  hipLaunchKernelGGL(kern, dim3(tx, ty, p.ksplit), dim3(256), lds, st, p);
  TBG_LAUNCH_CHECK();
  if (tx * ty * NT >= 256)
    hipLaunchKernelGGL((conv_wgrad_reduce_kernel<WGS, WGL, NT>), dim3(tx, ty, NT), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL((conv_wgrad_reduce_wide_kernel<WGS, WGL, NT>), dim3(tx, ty, NT * 16), dim3(256), 0, st, p);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

static bool wgrad_vec_ok(const WgradP &p, bool bf) {  // geometry of wgrad_stage_vec
  return p.logTW == 5 && p.logTHs == 1 && p.NSEG == 1 && p.sx == 1 && p.sy == 1 && (p.Ws & 3) == 0 && (p.Wl & 3) == 0 &&
         p.px == 1 && p.IWp == (bf ? WgVec<true>::IWP : WgVec<false>::IWP) &&
         p.lplane == (bf ? WgVec<true>::LPLANE : WgVec<false>::LPLANE) && (((uintptr_t)p.S | (uintptr_t)p.L) & 15) == 0;
}
static bool wgrad_vec2_ok(const WgradP &p, bool bf) {  // geometry of wgrad_stage_vec_s2
  return p.logTW == 5 && p.logTHs == 0 && p.NSEG == 1 && p.sx == 2 && p.sy == 2 && (p.Ws & 31) == 0 && p.px == 0 &&
         2 * p.Ws < p.Wl && p.IWp == (bf ? WgVec2<true>::IWP : WgVec2<false>::IWP) &&
         p.HALFW == (bf ? WgVec2<true>::HALFW : WgVec2<false>::HALFW) &&
         p.lplane == (bf ? WgVec2<true>::LPLANE : WgVec2<false>::LPLANE) && ((uintptr_t)p.S & 15) == 0;
}

template <int WGS, int WGL, int NT, int PIX, int SX, int VEC = 0>
static int launch_wgrad_bf16(WgradP &p, hipStream_t st, size_t ws_bytes, const NameOut *name) {
  constexpr int BS = WGS * 32, BL = WGL * 32;
  if constexpr (!VEC && NT == 9 && PIX == 64 && SX == 1 && WGS == 2 && WGL == 2) {
    if (wgrad_vec_ok(p, true)) return launch_wgrad_bf16<WGS, WGL, NT, PIX, SX, 1>(p, st, ws_bytes, name);
  }
  if constexpr (!VEC && NT == 9 && PIX == 32 && SX == 2 && WGS == 2 && WGL == 2) {
    if (wgrad_vec2_ok(p, true)) return launch_wgrad_bf16<WGS, WGL, NT, PIX, SX, 2>(p, st, ws_bytes, name);
  }
  if (name) {
    snprintf(name->buf, name->n, "conv_wgrad_bf16_kernel<%d, %d, %d, %d, %d, %d>", WGS, WGL, NT, PIX, SX, VEC);
    return TBG_OK;
  }
  const size_t lds = ((size_t)BS * (PIX + 8) + (size_t)BL * p.lplane) * 2;
  if (p.NJ > ((NT == 1 && PIX == 64) ? 1 : WG_MAXNJ)) return TBG_EUNSUPPORTED;
  if (lds > 160 * 1024) return TBG_EUNSUPPORTED;
  auto kern = conv_wgrad_bf16_kernel<WGS, WGL, NT, PIX, SX, VEC>;
  if (lds > 64 * 1024) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return TBG_EHIP;
  }
  const int tx = ceil_div(p.CS, BS), ty = ceil_div(p.CL, BL);
  p.ksplit = wgrad_ksplit(tx * ty, p.nchunks);
  if ((size_t)p.ksplit * tx * ty * NT * 16 * 256 * sizeof(float) > ws_bytes) return TBG_EINVAL;
  hipLaunchKernelGGL(kern, dim3(tx, ty, p.ksplit), dim3(256), lds, st, p);
  TBG_LAUNCH_CHECK();
  if (tx * ty * NT >= 256)
    hipLaunchKernelGGL((conv_wgrad_reduce_kernel<WGS, WGL, NT>), dim3(tx, ty, NT), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL((conv_wgrad_reduce_wide_kernel<WGS, WGL, NT>), dim3(tx, ty, NT * 16), dim3(256), 0, st, p);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

template <int WGS, int WGL, int NT, int PIX>
static int launch_wgrad(WgradP &p, hipStream_t st, size_t ws_bytes, const NameOut *name) {
  if constexpr (NT == 9 && PIX == 64 && WGS == 2 && WGL == 2) {
    // 32 x 2 pixel tiles of a stride-1 layer with 16-byte-aligned rows: float4 staging
    if (wgrad_vec_ok(p, false)) return launch_wgrad_impl<WGS, WGL, NT, PIX, true, 1>(p, st, ws_bytes, name);
  }
  if constexpr (NT == 9 && PIX == 32 && WGS == 2 && WGL == 2) {
    if (wgrad_vec2_ok(p, false)) return launch_wgrad_impl<WGS, WGL, NT, PIX, true, 2>(p, st, ws_bytes, name);
  }
  if (p.logTW >= 2) return launch_wgrad_impl<WGS, WGL, NT, PIX, true>(p, st, ws_bytes, name);
  return launch_wgrad_impl<WGS, WGL, NT, PIX, false>(p, st, ws_bytes, name);
}

// geometry shared by the launcher and the workspace query
static int wgrad_geometry(const tbg_wgrad_desc *d, WgradP &p, int &PIX, bool bf = false) {
  if (!d) return TBG_EINVAL;
  if (d->B < 1 || d->CS < 1 || d->CL < 1 || d->Hs < 1 || d->Ws < 1 || d->Hl < 1 || d->Wl < 1) return TBG_EINVAL;
  if (!((d->KH == 3 && d->KW == 3) || (d->KH == 1 && d->KW == 1))) return TBG_EUNSUPPORTED;
  if (d->sy < 1 || d->sy > 2 || d->sx < 1 || d->sx > 2) return TBG_EUNSUPPORTED;
  if ((double)d->B * d->CS * d->Hs * d->Ws > 2147483647.0 || (double)d->B * d->CL * d->Hl * d->Wl > 2147483647.0)
    return TBG_ERANGE;
  p.B = d->B; p.CS = d->CS; p.CL = d->CL; p.Hs = d->Hs; p.Ws = d->Ws; p.Hl = d->Hl; p.Wl = d->Wl;
  p.KW = d->KW; p.sy = d->sy; p.sx = d->sx; p.py = d->py; p.px = d->px;
  p.st_t = d->st_t; p.st_l = d->st_l; p.st_s = d->st_s; p.alpha = d->alpha;
  if (const int rcb = wgrad_bias_rider(p, d)) return rcb;
  const bool strided = d->sy == 2 || d->sx == 2;
  PIX = strided ? 32 : 64;
  int TW = 1, THs = 1;
  for (;;) {
    TW = pow2ceil(d->Ws) < 32 ? pow2ceil(d->Ws) : 32;
    if (TW > PIX) TW = PIX;
    const int TR = PIX / TW;
    THs = pow2ceil(d->Hs) < TR ? pow2ceil(d->Hs) : TR;
    p.logTW = ilog2(TW); p.logTHs = ilog2(THs); p.NSEG = TR / THs;
    p.IHs = (THs - 1) * d->sy + d->KH;
    p.IWs = (TW - 1) * d->sx + d->KW;
    p.HALFW = (p.IWs + 1) / 2;
    p.IWp = (d->sx == 2) ? 2 * p.HALFW : p.IWs;
    p.lplane = (p.NSEG * p.IHs * p.IWp) | 1;  // odd plane pitch: lanes walk channels conflict-free
    if (bf && TW >= 8) {  // bf16 tiles: row pitches in whole 16-byte units, channel pitch an ODD number of units
      p.HALFW = (p.HALFW + 7) & ~7;
      p.IWp = (d->sx == 2) ? 2 * p.HALFW : ((p.IWs + 7) & ~7);
      p.lplane = p.NSEG * p.IHs * p.IWp;
      if (((p.lplane / 8) & 1) == 0) p.lplane += 8;
    }
    p.ppc = p.NSEG * p.IHs * p.IWs;
    p.NJ = ceil_div(p.ppc, 64);
    const int cap = (d->KH * d->KW == 1 && PIX == 64) ? 1 : WG_MAXNJ;
    if (p.NJ <= cap) break;
    if (PIX == 64) { PIX = 32; continue; }  // very narrow maps: the 32-pixel chunk packs fewer images per tile
    return TBG_EUNSUPPORTED;
  }
  p.nBG = ceil_div(d->B, p.NSEG);
  p.tilesU = ceil_div(d->Hs, THs);
  p.tilesV = ceil_div(d->Ws, TW);
  p.nchunks = p.tilesU * p.tilesV * p.nBG;
  return TBG_OK;
}

extern "C" long long tbg_conv2d_wgrad_workspace_bytes(const tbg_wgrad_desc *d) {
  WgradP p{};
  int PIX;
  if (wgrad_geometry(d, p, PIX) != TBG_OK) return -1;
  const int tiles = ceil_div(p.CS, 64) * ceil_div(p.CL, 64);
  const int NT = d->KH * d->KW;
  return (long long)wgrad_ksplit(tiles, p.nchunks) * tiles * NT * 16 * 256 * (long long)sizeof(float);
}

extern "C" int tbg_conv2d_wgrad_f32(const tbg_wgrad_desc *d, const float *S, const float *L, float *dW,
                                    const float *s_scale, const float *l_scale, float *workspace,
                                    long long workspace_bytes, void *stream) {
  return tbg_conv2d_wgrad_ex_f32(d, S, L, dW, s_scale, l_scale, nullptr, nullptr, 0.f, workspace, workspace_bytes, stream);
}

static int wgrad_select(WgradP &p, int NT, int PIX, hipStream_t st, size_t wsb, const NameOut *name) {
  if (NT == 9) {
    if (PIX == 32) return launch_wgrad<2, 2, 9, 32>(p, st, wsb, name);
    return launch_wgrad<2, 2, 9, 64>(p, st, wsb, name);
  }
  if (PIX == 32) return launch_wgrad<2, 2, 1, 32>(p, st, wsb, name);
  return launch_wgrad<2, 2, 1, 64>(p, st, wsb, name);
}

extern "C" int tbg_conv2d_wgrad_ex_f32(const tbg_wgrad_desc *d, const float *S, const float *L, float *dW,
                                       const float *s_scale, const float *l_scale, const float *addw, const float *addq,
                                       float gamma, float *workspace, long long workspace_bytes, void *stream) {
  if (!d || !S || !L || !dW || !workspace || ((addw == nullptr) != (addq == nullptr))) return TBG_EINVAL;
  WgradP p{};
  int PIX;
  const int rc = wgrad_geometry(d, p, PIX);
  if (rc != TBG_OK) return rc;
  p.S = S; p.L = L; p.s_scale = s_scale; p.l_scale = l_scale; p.dW = dW; p.ws = workspace;
  p.addw = addw; p.addq = addq; p.gamma = gamma;
  const size_t wsb = workspace_bytes < 0 ? 0 : (size_t)workspace_bytes;
  return wgrad_select(p, d->KH * d->KW, PIX, tbg_stream(stream), wsb, nullptr);
}

// bf16 form: tile rows of >= 8 pixels take conv_wgrad_bf16_kernel; narrower maps (Ws <= 4: the 4x4 head) keep the exact fp32
// kernel -- they hold a negligible share of the work.  Same workspace size as the fp32 form (same chunking).
static int wgrad_bf16_select(WgradP &p, int NT, int PIX, int sx, hipStream_t st, size_t wsb, const NameOut *name) {
  if (NT == 9) {
    if (PIX == 32) return sx == 2 ? launch_wgrad_bf16<2, 2, 9, 32, 2>(p, st, wsb, name) : launch_wgrad_bf16<2, 2, 9, 32, 1>(p, st, wsb, name);
    return sx == 2 ? launch_wgrad_bf16<2, 2, 9, 64, 2>(p, st, wsb, name) : launch_wgrad_bf16<2, 2, 9, 64, 1>(p, st, wsb, name);
  }
  if (PIX == 32) return sx == 2 ? launch_wgrad_bf16<2, 2, 1, 32, 2>(p, st, wsb, name) : launch_wgrad_bf16<2, 2, 1, 32, 1>(p, st, wsb, name);
  return sx == 2 ? launch_wgrad_bf16<2, 2, 1, 64, 2>(p, st, wsb, name) : launch_wgrad_bf16<2, 2, 1, 64, 1>(p, st, wsb, name);
}

extern "C" int tbg_conv2d_wgrad_bf16(const tbg_wgrad_desc *d, const float *S, const float *L, float *dW,
                                     const float *s_scale, const float *l_scale, const float *addw, const float *addq,
                                     float gamma, float *workspace, long long workspace_bytes, void *stream) {
  if (!d || !S || !L || !dW || !workspace || ((addw == nullptr) != (addq == nullptr))) return TBG_EINVAL;
  WgradP p{};
  int PIX;
  int rc = wgrad_geometry(d, p, PIX, true);
  if (rc != TBG_OK) return rc;
  p.S = S; p.L = L; p.s_scale = s_scale; p.l_scale = l_scale; p.dW = dW; p.ws = workspace;
  p.addw = addw; p.addq = addq; p.gamma = gamma;
  const size_t wsb = workspace_bytes < 0 ? 0 : (size_t)workspace_bytes;
  if (p.logTW < 3) return wgrad_select(p, d->KH * d->KW, PIX, tbg_stream(stream), wsb, nullptr);
  return wgrad_bf16_select(p, d->KH * d->KW, PIX, d->sx, tbg_stream(stream), wsb, nullptr);
}

// f32x3 filter gradient: the float4-staged geometries (the large stride-1 and stride-2 3x3 layers, where the FLOPs are) take
// conv_wgrad_x3_kernel; everything else keeps the exact fp32 kernel.  Workspace = tbg_conv2d_wgrad_workspace_bytes(d).
template <int VEC>
static int launch_wgrad_x3(WgradP &p, hipStream_t st, size_t ws_bytes, const NameOut *name) {
  if (name) {
    snprintf(name->buf, name->n, "conv_wgrad_x3_kernel<%d>", VEC);
    return TBG_OK;
  }
  const size_t lds = (size_t)3 * 64 * (WgX3<VEC>::SPB + WgX3<VEC>::LPLANE) * 2;
  auto kern = conv_wgrad_x3_kernel<VEC>;
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return TBG_EHIP;
  const int tx = ceil_div(p.CS, 64), ty = ceil_div(p.CL, 64);
  p.ksplit = wgrad_ksplit(tx * ty, p.nchunks, 256);  // one block per CU (84 KB of LDS, 348 registers): one round, half the partials
  if ((size_t)p.ksplit * tx * ty * 9 * 16 * 256 * sizeof(float) > ws_bytes) return TBG_EINVAL;
  hipLaunchKernelGGL(kern, dim3(tx, ty, p.ksplit), dim3(256), lds, st, p);
  TBG_LAUNCH_CHECK();
  if (tx * ty * 9 >= 256)
    hipLaunchKernelGGL((conv_wgrad_reduce_kernel<2, 2, 9>), dim3(tx, ty, 9), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL((conv_wgrad_reduce_wide_kernel<2, 2, 9>), dim3(tx, ty, 9 * 16), dim3(256), 0, st, p);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

// 0: not an x3 geometry (caller falls back to the exact fp32 kernel); 1 / 2: the VEC form
static int wgrad_x3_form(const tbg_wgrad_desc *d, WgradP &p, int &rc) {
  int PIX;
  rc = wgrad_geometry(d, p, PIX, true);
  if (rc != TBG_OK) return 0;
  if (d->KH != 3 || d->KW != 3) return 0;
  if (PIX == 64 && wgrad_vec_ok(p, true)) return 1;
  if (PIX == 32 && wgrad_vec2_ok(p, true)) return 2;
  return 0;
}

extern "C" int tbg_conv2d_wgrad_x3(const tbg_wgrad_desc *d, const float *S, const float *L, float *dW,
                                   const float *s_scale, const float *l_scale, const float *addw, const float *addq,
                                   float gamma, float *workspace, long long workspace_bytes, void *stream) {
  if (!d || !S || !L || !dW || !workspace || ((addw == nullptr) != (addq == nullptr))) return TBG_EINVAL;
  WgradP p{};
  int rc;
  p.S = S; p.L = L;  // (the alignment test of the float4 forms reads the pointers)
  const int form = wgrad_x3_form(d, p, rc);
  if (rc != TBG_OK) return rc;
  if (!form) return tbg_conv2d_wgrad_ex_f32(d, S, L, dW, s_scale, l_scale, addw, addq, gamma, workspace, workspace_bytes, stream);
  p.s_scale = s_scale; p.l_scale = l_scale; p.dW = dW; p.ws = workspace;
  p.addw = addw; p.addq = addq; p.gamma = gamma;
  const size_t wsb = workspace_bytes < 0 ? 0 : (size_t)workspace_bytes;
  return form == 1 ? launch_wgrad_x3<1>(p, tbg_stream(stream), wsb, nullptr) : launch_wgrad_x3<2>(p, tbg_stream(stream), wsb, nullptr);
}

extern "C" int tbg_conv2d_wgrad_x3_kernel_name(const tbg_wgrad_desc *d, char *buf, int n) {
  if (!buf || n < 1) return TBG_EINVAL;
  buf[0] = 0;
  WgradP p{};
  int rc;
  const int form = wgrad_x3_form(d, p, rc);  // (name-only: the 16-byte pointer alignment the launch also requires is assumed)
  if (rc != TBG_OK) return rc;
  if (!form) return tbg_conv2d_wgrad_kernel_name(d, buf, n);
  NameOut no{buf, n};
  return form == 1 ? launch_wgrad_x3<1>(p, nullptr, 0, &no) : launch_wgrad_x3<2>(p, nullptr, 0, &no);
}

extern "C" int tbg_conv2d_wgrad_bf16_kernel_name(const tbg_wgrad_desc *d, char *buf, int n) {
  if (!buf || n < 1) return TBG_EINVAL;
  buf[0] = 0;
  WgradP p{};
  int PIX;
  const int rc = wgrad_geometry(d, p, PIX, true);
  if (rc != TBG_OK) return rc;
  NameOut no{buf, n};
  if (p.logTW < 3) return wgrad_select(p, d->KH * d->KW, PIX, nullptr, 0, &no);
  return wgrad_bf16_select(p, d->KH * d->KW, PIX, d->sx, nullptr, 0, &no);
}

extern "C" int tbg_conv2d_wgrad_kernel_name(const tbg_wgrad_desc *d, char *buf, int n) {
  if (!buf || n < 1) return TBG_EINVAL;
  buf[0] = 0;
  WgradP p{};
  int PIX;
  const int rc = wgrad_geometry(d, p, PIX);
  if (rc != TBG_OK) return rc;
  NameOut no{buf, n};
  return wgrad_select(p, d->KH * d->KW, PIX, nullptr, 0, &no);
}
