// Types and kernels shared by the convolution translation units (conv.hip: NCHW-fp32 operands; conv_units.hip: operands in
// the 8-channel-unit layout): operand vector types, the fp32 -> 3 x bf16 split, the filter-gradient parameter block and the
// kernels that sum its partial tiles.
#pragma once
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// 8 floats -> 8 bf16 (round to nearest even: v_cvt_pk_bf16_f32), one 16-byte LDS unit
__device__ __forceinline__ bf16x8 pack_bf16x8(const float (&v)[8]) {
  bf16x8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) r[i] = (__bf16)v[i];
  return r;
}

// fp32 -> three bf16 terms hi + mid + lo (8 + 8 + 8 significand bits: exact for every fp32 value whose low terms stay
// in the normal range).  Products of two bf16 values are exact in fp32, so the six largest partial products of
// (a_hi + a_mid + a_lo)(b_hi + b_mid + b_lo) reproduce a * b to ~2^-24 relative -- the "f32x3" arithmetic of the X3 kernels.
// Two values at a time, on packed instructions: v_cvt_pk_bf16_f32 (both conversions), shift / mask (the bf16 pair back to
// fp32), v_pk_add_f32 (both residuals) -- 9 VALU instructions per pair.  The element-wise form ((__bf16)v, (float)b per element)
// compiled to one conversion and one v_perm per ELEMENT and term: ~10 instructions per element in the staging passes
// (profiles/r03_ab_one_box.txt).  Same roundings, same bits.
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2p __attribute__((ext_vector_type(2)));
typedef unsigned u32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split3_pair(float a, float b, unsigned &h, unsigned &m, unsigned &l) {
  const f32x2v v = {a, b};
  const unsigned hu = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2p));
  const f32x2v hf = {__builtin_bit_cast(float, hu << 16), __builtin_bit_cast(float, hu & 0xffff0000u)};
  const f32x2v r1 = v - hf;
  const unsigned mu = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2p));
  const f32x2v mf = {__builtin_bit_cast(float, mu << 16), __builtin_bit_cast(float, mu & 0xffff0000u)};
  const f32x2v r2 = r1 - mf;
  h = hu; m = mu; l = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2p));
}
__device__ __forceinline__ void split3_bf16x8(const float (&v)[8], bf16x8 &h, bf16x8 &m, bf16x8 &l) {
  unsigned hu[4], mu[4], lu[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) split3_pair(v[2 * i], v[2 * i + 1], hu[i], mu[i], lu[i]);
  h = __builtin_bit_cast(bf16x8, u32x4v{hu[0], hu[1], hu[2], hu[3]});
  m = __builtin_bit_cast(bf16x8, u32x4v{mu[0], mu[1], mu[2], mu[3]});
  l = __builtin_bit_cast(bf16x8, u32x4v{lu[0], lu[1], lu[2], lu[3]});
}

static inline int pow2ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }
static inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// name-only mode (name != NULL): write the instantiation the descriptor selects (rocprofv3 spelling) and launch nothing
struct NameOut { char *buf; int n; int *dot_slots; int *blocks; };

// One LDS-DMA piece: 64 lanes x 16 bytes from per-lane global addresses to the lane-linear LDS range starting at the
// wave-uniform byte address `lds_base` (M0).  Inline assembly ON PURPOSE: hipcc's waitcnt pass treats a
// __builtin_amdgcn_global_load_lds in flight as a pending write to ALL of LDS and puts s_waitcnt vmcnt(0) in front of every
// later LDS read -- including the operand reads of the OTHER buffer, which serialises the DMA with the MFMA phase it is
// meant to hide under (seen in the ISA of this kernel's builtin form, and in conv.hip's split-filter pipeline).
// The kernel orders the pieces itself: s_waitcnt vmcnt(0) + s_barrier before a buffer is read.
__device__ __forceinline__ void dma16(const void *gsrc, unsigned lds_base) {
  asm volatile("s_mov_b32 m0, %1\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_base)) : "memory");
}


// ---- fused epilogue of the implicit-GEMM convolution kernels (tbg.h tbg_epilogue), for a wave that holds WTM x WTN 32x32 MFMA
// accumulator tiles: rows = output channels mrow0 + 32 i + (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column (lane & 31) of tile j =
// the output pixel e_pix[j] (offset inside the [Hout*Wout] plane, -1: not an output) of sample e_b[j].  slab != NULL: store-only
// alpha * acc into this split's slab.  Every pointer / scalar of the descriptor is hoisted into locals first, and the loads a row
// group needs (demodulation, residual, dot operand) are issued together before any arithmetic or store: the first version
// interleaved kernarg reloads, loads and stores element by element (521 s_waitcnt in 6.8k instructions) and cost 45-80 us per
// large launch -- more than the output write itself (a 134 MB fill takes 21 us).
// Fused dot (e.dot_aux): one image per tile; the partial of this (tile, wave column) goes to its own slot
// dot_out[(dot_b * M + m) * dot_slots + dot_slot] -- no atomics, the caller sums the slots in a fixed order.
// Unit sink (e.units_out, tbg.h): the same values, times units_scale[b, m], ALSO leave as the unit tensor the next convolution
// DMAs its tiles from -- U[plane][b][m/8][1 + Y][1 + X][m % 8] bf16.  The 32x32 MFMA accumulator layout gives a lane 4 CONSECUTIVE
// channels of one pixel per row group (rows r0 .. r0+3 -> channels m0 + 8 g + 4 (lane >> 5) + 0..3): that is one 8-byte half of a
// 16-byte unit per plane, the other half-wave writes the other half, 32 lanes cover 32 consecutive pixels -> one 512-byte run per
// plane and half.  The ring of zero units belongs to the tensor's contract (the consumers read padding, they never clamp): the
// lane that owns a border pixel also zeroes the ring positions next to it (its own channel half), so every unit of the tensor is
// written by the launch whatever its tiling.  y may be NULL with a sink (no fp32 output at all).
template <int NPL>
__device__ __forceinline__ void sink_store(char *ub, long long plane_bytes, long long off, const float (&v)[4]) {
  unsigned h0, m0, l0, h1, m1, l1;
  if constexpr (NPL == 3) {
    split3_pair(v[0], v[1], h0, m0, l0);
    split3_pair(v[2], v[3], h1, m1, l1);
    *reinterpret_cast<u32x2v *>(ub + off) = u32x2v{h0, h1};
    *reinterpret_cast<u32x2v *>(ub + plane_bytes + off) = u32x2v{m0, m1};
    *reinterpret_cast<u32x2v *>(ub + 2 * plane_bytes + off) = u32x2v{l0, l1};
  } else {
    const f32x2v a = {v[0], v[1]}, b = {v[2], v[3]};
    h0 = __builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2p));
    h1 = __builtin_bit_cast(unsigned, __builtin_convertvector(b, bf16x2p));
    *reinterpret_cast<u32x2v *>(ub + off) = u32x2v{h0, h1};
  }
}
__device__ __forceinline__ void sink_zero(char *ub, long long plane_bytes, int planes, long long off) {
  for (int pl = 0; pl < planes; ++pl) *reinterpret_cast<u32x2v *>(ub + pl * plane_bytes + off) = u32x2v{0u, 0u};
}

// Sum every row's 32 pixel lanes (one half-wave) for all N rows of a lane together: at distance OFF a lane hands the half of its rows
// its partner keeps and adds the partner's copy of the half it keeps itself (N / 2 exchanges), then the same on the kept half at
// OFF / 2; a single remaining row is summed across the remaining distances.  All indices are compile-time (registers, no scratch).
template <int NDOT, int N, int OFF>
__device__ __forceinline__ void dot_tree(float (&a)[NDOT], const int lane) {
  if constexpr (OFF > 0) {
    if constexpr (N > 1) {
      constexpr int H = N / 2;
      const bool up = (lane & OFF) != 0;
#pragma unroll
      for (int k = 0; k < H; ++k) {
        const float send = up ? a[k] : a[k + H], keep = up ? a[k + H] : a[k];
        a[k] = keep + __shfl_xor(send, OFF, 64);
      }
      dot_tree<NDOT, H, OFF / 2>(a, lane);
    } else {
      a[0] += __shfl_xor(a[0], OFF, 64);
      dot_tree<NDOT, 1, OFF / 2>(a, lane);
    }
  }
}

// SINK = false compiles the sink out (the exact-fp32 instantiations: no unit consumer exists in that arithmetic, and the sink's
// registers pushed the 4-waves/SIMD builds into scratch -- 320 bytes per lane, exact-fp32 step 27.1 -> 31.7 ms).
// NROW < 16 (conv_small.hip): the wave holds only accumulator rows 0 .. NROW-1 of its tile (one 4-channel row group after the
// in-block K reduction) -- the loops stop there, everything else is unchanged.
// OPT: which optional operand paths are compiled in -- bit 0 the residual, bit 1 the dot / gate operand.  A launch that has neither
// runs the OPT = 0 instantiation: a third of the epilogue's code (loads, shuffle trees, branches on flags that are never set) is not
// there (conv_units_fprop_kernel: -2 .. -4 % per launch in f32x3, -7 % in bf16, profiles/r06_epilogue_opt.txt).
// BATCH (the 512-thread unit-tensor kernels: 256 registers per lane): the unit sink's scales are loaded with the row group's other
// operands (one dependent load per element inside the arithmetic before), and -- BDOT -- the dot operand's per-row sums are kept for
// all rows and reduced together after the stores (dot_tree: NDOT - 1 exchanges in five steps of independent instructions; one dependent
// 5-exchange chain per row before, 160 exchanges for a 128-channel tile).  Measured per launch, forward + sink / data gradient + dot:
// -4 .. -8 % / -5 .. -16 % in f32x3, -18 % / -20 % in bf16 (profiles/r06_epilogue_opt.txt).  The 256-thread NCHW kernels (128 registers
// per lane) keep the old forms: the extra live registers sent them to scratch; BDOT is off for the bf16 64-channel tile, which its
// registers would take from two resident blocks per CU to one.
template <int WTM, int WTN, int RG, bool SINK = true, int NROW = 16, int OPT = 3, bool BATCH = false, bool BDOT = BATCH>
__device__ __forceinline__ void conv_epilogue(const f32x16 (&acc)[WTM][WTN], const EpiK &e, float *y, float *slab, int M_, int HWout,
                                              int mrow0, int lane, const int (&e_pix)[WTN], const int (&e_b)[WTN], bool dot_ok,
                                              int dot_b, int dot_slots, int dot_slot, int Hout = 0, int Wout = 0) {
  const float *const e_os = e.out_scale, *const e_bias = e.bias;
  const float *const e_res = (OPT & 1) ? e.residual : nullptr;
  const float *const e_aux = (OPT & 2) ? e.dot_aux : nullptr, *const e_gate = (OPT & 2) ? e.gate : nullptr;
  float *const e_dot = e.dot_out;
  const float e_alpha = e.alpha, e_bmul = e.bias_mul, e_slope = e.slope, e_gain = e.gain, e_rscale = e.res_scale;
  const bool e_lrelu = e.act == TBG_ACT_LRELU, e_rfirst = e.res_first != 0;
  const float str = e.noise ? e.strength[0] : 0.f;
  const bool split = slab != nullptr;
  const bool do_dot = e_aux != nullptr && !split;
  char *const e_ub = SINK ? static_cast<char *>(e.units_out) : nullptr;
  const bool sink = SINK && e_ub != nullptr && !split;
  const bool plain = !e_os && !e_bias && !e.noise && !e_res && !e_aux && !e_gate && !e_lrelu && e_gain == 1.f && !sink;
  const int M = M_;
  float *const ybase = split ? slab : y;
  float e_nz[WTN];
#pragma unroll
  for (int j = 0; j < WTN; ++j) e_nz[j] = (e_pix[j] >= 0 && e.noise) ? e.noise[(size_t)e_b[j] * HWout + e_pix[j]] * str : 0.f;
  if (split || plain) {  // store-only: alpha * acc (split-K slabs, plain data gradients)
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
      for (int r16 = 0; r16 < NROW; ++r16) {
        const int m = mrow0 + i * 32 + (r16 & 3) + 8 * (r16 >> 2) + 4 * (lane >> 5);
#pragma unroll
        for (int j = 0; j < WTN; ++j)
          if (m < M && e_pix[j] >= 0) ybase[(e_b[j] * M + m) * HWout + e_pix[j]] = acc[i][j][r16] * e_alpha;
      }
    return;
  }
  // unit-sink geometry of this lane's pixels: byte offset of pixel j's unit inside plane 0 (channel unit 0), ring flags
  static_assert(RG == 4 || RG == 2, "a row group is one or half of a 4-channel run");
  const float *const e_us = e.units_scale;
  const int s_Hp = Hout + 2, s_Wp = Wout + 2;
  const long long s_cu = 16LL * s_Hp * s_Wp;     // bytes between two channel units of one sample
  const long long s_plane = e.units_plane;       // bytes between two planes (hi | mid | lo)
  const int s_np = e.units_planes;
  long long s_off[WTN];
  int s_ring[WTN];  // bit 0: left edge, 1: right edge, 2: top edge, 3: bottom edge (of a live pixel)
  if (sink) {
#pragma unroll
    for (int j = 0; j < WTN; ++j) {
      const int pix = max(e_pix[j], 0);
      const int Y = pix / Wout, X = pix - Y * Wout;
      s_off[j] = (long long)e_b[j] * (M_ >> 3) * s_cu + 16LL * ((Y + 1) * s_Wp + X + 1);
      s_ring[j] = e_pix[j] < 0 ? 0 : ((X == 0 ? 1 : 0) | (X == Wout - 1 ? 2 : 0) | (Y == 0 ? 4 : 0) | (Y == Hout - 1 ? 8 : 0));
    }
  }
  constexpr int NDOT = BDOT ? WTM * NROW : 1;  // accumulator rows of this lane (row k = i * NROW + r16)
  float dacc[NDOT];
#pragma unroll
  for (int k = 0; k < NDOT; ++k) dacc[k] = 0.f;
#pragma unroll
  for (int i = 0; i < WTM; ++i) {
#pragma unroll
    for (int r0 = 0; r0 < NROW; r0 += RG) {  // accumulator rows r0 .. r0+RG-1
      int idx[RG][WTN];  // output offsets fit 31 bits (checked on the host)
      int mrow[RG];
      float bias4[RG], osv[RG][WTN], rsv[RG][WTN], axv[RG][WTN], usv[RG][WTN];
#pragma unroll
      for (int q = 0; q < RG; ++q) {
        const int r16 = r0 + q;
        const int m = mrow0 + i * 32 + (r16 & 3) + 8 * (r16 >> 2) + 4 * (lane >> 5);
        mrow[q] = m;
        const bool okm = m < M;
        bias4[q] = (okm && e_bias) ? e_bias[m] * e_bmul : 0.f;
#pragma unroll
        for (int j = 0; j < WTN; ++j) {
          idx[q][j] = (okm && e_pix[j] >= 0) ? (e_b[j] * M + m) * HWout + e_pix[j] : -1;
          osv[q][j] = 1.f; rsv[q][j] = 0.f; axv[q][j] = 0.f; usv[q][j] = 1.f;
        }
      }
      if (BATCH && sink && e_us) {
#pragma unroll
        for (int q = 0; q < RG; ++q)
#pragma unroll
          for (int j = 0; j < WTN; ++j) usv[q][j] = e_us[idx[q][j] >= 0 ? e_b[j] * M + mrow[q] : 0];
      }
      if (e_os) {
#pragma unroll
        for (int q = 0; q < RG; ++q)
#pragma unroll
          for (int j = 0; j < WTN; ++j) osv[q][j] = e_os[idx[q][j] >= 0 ? e_b[j] * M + mrow[q] : 0];
      }
      if (e_res) {
#pragma unroll
        for (int q = 0; q < RG; ++q)
#pragma unroll
          for (int j = 0; j < WTN; ++j) rsv[q][j] = e_res[max(idx[q][j], 0)];  // clamped: branch-free
      }
      if (do_dot) {
#pragma unroll
        for (int q = 0; q < RG; ++q)
#pragma unroll
          for (int j = 0; j < WTN; ++j) axv[q][j] = e_aux[max(idx[q][j], 0)];
      } else if (e_gate) {  // never together with the dot operand (epi_valid): the gate values share its registers
#pragma unroll
        for (int q = 0; q < RG; ++q)
#pragma unroll
          for (int j = 0; j < WTN; ++j) axv[q][j] = e_gate[max(idx[q][j], 0)];
      }
      float uval[RG][WTN];
#pragma unroll
      for (int q = 0; q < RG; ++q) {
        const int m = mrow[q];
        float dsum = 0.f;
#pragma unroll
        for (int j = 0; j < WTN; ++j) {
          const bool okq = idx[q][j] >= 0;
          float val = acc[i][j][r0 + q] * e_alpha;
          if (do_dot) dsum += okq ? val * axv[q][j] : 0.f;
          val = val * osv[q][j] + e_nz[j] + bias4[q];
          if (e_rfirst) val += rsv[q][j];
          val = (e_lrelu ? (val > 0.f ? val : val * e_slope) : val) * e_gain;
          if (e_res && !e_rfirst) val = (val + rsv[q][j]) * e_rscale;
          if (e_gate) val = axv[q][j] > 0.f ? val : 0.f;
          if (okq && y) y[idx[q][j]] = val;
          if (sink) uval[q][j] = val * (BATCH ? usv[q][j] : (e_us ? e_us[okq ? e_b[j] * M + m : 0] : 1.f));
        }
        if (do_dot) {  // one image per tile (checked on the host): the 32 pixel lanes of each half-wave are summed and the partial of
          // this (tile, wave column) goes to its own slot -- no atomics, the caller sums the slots in a fixed order
          if constexpr (BDOT) {
            dacc[i * NROW + r0 + q] = dsum;  // reduced with all the other rows after the stores
          } else {
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) dsum += __shfl_xor(dsum, off, 64);
            if ((lane & 31) == 0 && m < M && dot_ok) e_dot[((size_t)dot_b * M + m) * dot_slots + dot_slot] = dsum;
          }
        }
      }
      if (sink && mrow[0] < M) {  // (M % 8 == 0: a 4-channel run is inside M or outside it as a whole)
        // this row group's channels mrow[0] .. mrow[0] + RG - 1 sit at byte (mrow[0] & 7) * 2 of unit mrow[0] >> 3
        const long long cbase = (long long)(mrow[0] >> 3) * s_cu + (mrow[0] & 7) * 2;
#pragma unroll
        for (int j = 0; j < WTN; ++j) {
          if (e_pix[j] < 0) continue;
          const long long off = s_off[j] + cbase;
          if constexpr (RG == 4) {
            const float v4[4] = {uval[0][j], uval[1][j], uval[2][j], uval[3][j]};
            if (s_np == 3) sink_store<3>(e_ub, s_plane, off, v4);
            else sink_store<1>(e_ub, s_plane, off, v4);
          } else {
            for (int pl = 0; pl < s_np; ++pl) {  // RG == 2: a dword (two channels) per plane
              float r0v = uval[0][j], r1v = uval[1][j];
              unsigned h, mm, ll;
              if (s_np == 3) split3_pair(r0v, r1v, h, mm, ll);
              else { const f32x2v a2 = {r0v, r1v}; h = __builtin_bit_cast(unsigned, __builtin_convertvector(a2, bf16x2p)); mm = ll = 0u; }
              *reinterpret_cast<unsigned *>(e_ub + pl * s_plane + off) = pl == 0 ? h : pl == 1 ? mm : ll;
            }
          }
          if (s_ring[j]) {  // border pixel: zero this channel run at the ring positions next to it (corners with the column lanes)
            const int rg = s_ring[j];
            const long long rowb = 16LL * s_Wp;
            auto z = [&](long long o) {
              if constexpr (RG == 4) sink_zero(e_ub, s_plane, s_np, o);
              else for (int pl = 0; pl < s_np; ++pl) *reinterpret_cast<unsigned *>(e_ub + pl * s_plane + o) = 0u;
            };
            if (rg & 1) z(off - 16);
            if (rg & 2) z(off + 16);
            if (rg & 4) { z(off - rowb); if (rg & 1) z(off - rowb - 16); if (rg & 2) z(off - rowb + 16); }
            if (rg & 8) { z(off + rowb); if (rg & 1) z(off + rowb - 16); if (rg & 2) z(off + rowb + 16); }
          }
        }
      }
    }
  }
  if (BDOT && do_dot) {
    dot_tree<NDOT, NDOT, 16>(dacc, lane);
    // which rows this lane holds now: one bit per halving step, the lane bit of that step's distance (most significant first)
    int k = 0, nn = NDOT;
    for (int off = 16; off > 0; off >>= 1) {
      if (nn > 1) { nn >>= 1; k += (lane & off) ? nn : 0; }
    }
    // lanes that differ only in the bits of the plain-sum steps hold the same totals: the lowest one writes
    int plain_mask = 0, n2 = NDOT;
    for (int off = 16; off > 0; off >>= 1) {
      if (n2 > 1) n2 >>= 1; else plain_mask |= off;
    }
    constexpr int NLEFT = NDOT > 32 ? NDOT / 32 : 1;  // rows a lane still holds after the five steps (consecutive rows k, k + 1, ..)
    for (int t = 0; t < NLEFT; ++t) {
      const int kk = k + t, ki = kk / NROW, kr = kk - ki * NROW;
      const int m = mrow0 + ki * 32 + (kr & 3) + 8 * (kr >> 2) + 4 * (lane >> 5);
      if ((lane & plain_mask) == 0 && m < M && dot_ok) e_dot[((size_t)dot_b * M + m) * dot_slots + dot_slot] = dacc[t];
    }
  }
}

// the OPT instantiation of conv_epilogue a launch needs
static inline int epi_opt(const EpiK &e) { return (e.residual ? 1 : 0) | ((e.dot_aux || e.gate) ? 2 : 0); }

#define WG_MAXNJ 4

struct WgradP {
  const float *S, *L, *s_scale, *l_scale, *addw, *addq;
  float *dW, *ws;
  float gamma;
  int B, CS, CL, Hs, Ws, Hl, Wl, KW, sy, sx, py, px;
  int st_t, st_l, st_s;
  float alpha;
  int logTW, logTHs, NSEG, IHs, IWs, IWp, HALFW, lplane, ppc, NJ, nBG, tilesU, tilesV, nchunks, ksplit;
  const float *bias_parts;  // rider of the reduce launch (tbg_wgrad_desc): db[cs] = sum of the bias_act backward's partial sums
  float *bias_grad;
  int bias_B, bias_nch;
};

static inline int wgrad_bias_rider(WgradP &p, const tbg_wgrad_desc *d) {
  p.bias_parts = d->bias_parts; p.bias_grad = d->bias_grad; p.bias_B = d->bias_B; p.bias_nch = d->bias_nch;
  if (d->bias_grad && (!d->bias_parts || d->bias_B < 1 || d->bias_nch < 1)) return TBG_EINVAL;
  return TBG_OK;
}

// one block of a reduce launch: db[cs] = sum_{b, k} parts[(b * CS + cs) * nch + k], four independent chains per lane
__device__ __forceinline__ void wgrad_bias_reduce(const WgradP &p) {
  for (int m = threadIdx.x; m < p.CS; m += 256) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int b = 0;
    for (; b + 4 <= p.bias_B; b += 4)
      for (int k = 0; k < p.bias_nch; ++k) {
        a0 += p.bias_parts[((size_t)(b + 0) * p.CS + m) * p.bias_nch + k];
        a1 += p.bias_parts[((size_t)(b + 1) * p.CS + m) * p.bias_nch + k];
        a2 += p.bias_parts[((size_t)(b + 2) * p.CS + m) * p.bias_nch + k];
        a3 += p.bias_parts[((size_t)(b + 3) * p.CS + m) * p.bias_nch + k];
      }
    for (; b < p.bias_B; ++b)
      for (int k = 0; k < p.bias_nch; ++k) a0 += p.bias_parts[((size_t)b * p.CS + m) * p.bias_nch + k];
    p.bias_grad[m] = (a0 + a1) + (a2 + a3);
  }
}

// few output tiles, many partials: one block per (tile, tap, accumulator register) -- 16x more blocks than the
// transposing kernel below, scattered 4-byte stores
template <int WGS, int WGL, int NT>
__global__ __launch_bounds__(256) void conv_wgrad_reduce_wide_kernel(const WgradP p) {
  if (p.bias_grad && blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1 && blockIdx.z == gridDim.z - 1) wgrad_bias_reduce(p);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ws_ = wave / WGL, wl = wave - ws_ * WGL;
  const int cs0 = blockIdx.x * (WGS * 32), cl0 = blockIdx.y * (WGL * 32);
  const int tr = blockIdx.z;  // (t, r16)
  const int t = tr >> 4, r16 = tr & 15;
  const size_t tile = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  const size_t per_split = (size_t)gridDim.x * gridDim.y;
  float a = 0.f;
  const float *src = p.ws + tile * (size_t)(NT * 16 * 256) + (size_t)tr * 256 + tid;
  const size_t kstride = per_split * (size_t)(NT * 16 * 256);
  int kz = 0;
  for (; kz + 16 <= p.ksplit; kz += 16) {  // 16 partial tiles in flight per lane; summed in split order (deterministic)
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = src[(size_t)(kz + u) * kstride];
#pragma unroll
    for (int u = 0; u < 16; ++u) a += v[u];
  }
  for (; kz + 4 <= p.ksplit; kz += 4) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = src[(size_t)(kz + u) * kstride];
#pragma unroll
    for (int u = 0; u < 4; ++u) a += v[u];
  }
  for (; kz < p.ksplit; ++kz) a += src[(size_t)kz * kstride];
  const int cl = cl0 + wl * 32 + (lane & 31);
  const int cs = cs0 + ws_ * 32 + (r16 & 3) + 8 * (r16 >> 2) + 4 * (lane >> 5);
  if (cl < p.CL && cs < p.CS)
  {
    const long long iq = (long long)cl * p.st_l + (long long)cs * p.st_s, idx = (long long)t * p.st_t + iq;
    float v = a * p.alpha;
    if (p.addw) v += p.gamma * p.addw[idx] * p.addq[iq];
    p.dW[idx] = v;
  }
}

// sum the ksplit partial tiles of one (output tile, tap) and write dW (alpha applied once).  The 64x64 tile is
// transposed through LDS so that the dW rows are written as contiguous runs along whichever of (cs, cl) has unit stride.
template <int WGS, int WGL, int NT>
__global__ __launch_bounds__(256) void conv_wgrad_reduce_kernel(const WgradP p) {
  constexpr int BS = WGS * 32, BL = WGL * 32;
  __shared__ float tile[BL][BS + 1];
  if (p.bias_grad && blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1 && blockIdx.z == gridDim.z - 1) wgrad_bias_reduce(p);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ws_ = wave / WGL, wl = wave - ws_ * WGL;
  const int cs0 = blockIdx.x * BS, cl0 = blockIdx.y * BL;
  const int t = blockIdx.z;
  const size_t tl = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  const size_t per_split = (size_t)gridDim.x * gridDim.y;
  float a[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) a[r] = 0.f;
  int kz = 0;
  for (; kz + 2 <= p.ksplit; kz += 2) {  // 32 independent coalesced loads in flight
    const float *s0 = p.ws + ((size_t)kz * per_split + tl) * (size_t)(NT * 16 * 256) + (size_t)t * 16 * 256 + tid;
    const float *s1 = s0 + per_split * (size_t)(NT * 16 * 256);
    float v0[16], v1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { v0[r] = s0[r * 256]; v1[r] = s1[r * 256]; }
#pragma unroll
    for (int r = 0; r < 16; ++r) { a[r] += v0[r]; a[r] += v1[r]; }
  }
  for (; kz < p.ksplit; ++kz) {
    const float *src = p.ws + ((size_t)kz * per_split + tl) * (size_t)(NT * 16 * 256) + (size_t)t * 16 * 256 + tid;
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] += src[r * 256];
  }
  const int cl_l = wl * 32 + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) tile[cl_l][ws_ * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)] = a[r];
  __syncthreads();
  const bool s_fast = p.st_s == 1 || p.st_l != 1;
  for (int idx = tid; idx < BS * BL; idx += 256) {
    const int cl = s_fast ? idx / BS : idx % BL;
    const int cs = s_fast ? idx % BS : idx / BL;
    if (cl0 + cl < p.CL && cs0 + cs < p.CS)
    {
      const long long iq = (long long)(cl0 + cl) * p.st_l + (long long)(cs0 + cs) * p.st_s, idx = (long long)t * p.st_t + iq;
      float v = tile[cl][cs] * p.alpha;
      if (p.addw) v += p.gamma * p.addw[idx] * p.addq[iq];
      p.dW[idx] = v;
    }
  }
}

static int wgrad_ksplit(int tiles, int nchunks, int blocks = 512) {
  int ksplit = ceil_div(blocks, tiles);  // 512: ~2 resident blocks per CU
  if (ksplit > nchunks) ksplit = nchunks;
  if (ksplit < 1) ksplit = 1;
  return ksplit;
}

