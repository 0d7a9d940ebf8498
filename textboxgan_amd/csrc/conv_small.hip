// Small-map convolutions from unit tensors: the K split lives INSIDE the block, no HBM slabs, the real epilogue in the same launch.
//
// Where it applies: the 1x25 ... 8x32 maps of both networks and of the frozen recogniser's trunk (conv.py:51-73,
// discriminator.py:68-84, modulated_conv2d.py:98-112; the ResNet behind aster_inferer.py:28-37) -- GEMMs of M = 32 ... 512 output
// channels x N = 400 ... 3200 pixels x K = 256 ... 4608: 0.1 - 5 GFLOP, i.e. 0.3 - 12 us of the bf16 matrix pipe.  conv_fprop_kernel
// ran them as 64 x 64 tiles with K split over BLOCKS into HBM slabs + a second launch (tbg_slab_epilogue_f32): a launch was 6.5 us of
// fixed cost + the K loop + a 5 us second half, half of it slab traffic (profiles/r05_ab_one_box.txt (d), (e)).
//
// This kernel: one 512-thread block owns a 32-channel x (32 TN)-pixel output tile, its 8 waves split K -- wave w takes the row units
// u = w, w + 8, ... of the (8- or 16-channel chunk, filter row kh) list -- and every wave holds the WHOLE tile in accumulators.  The
// waves never synchronise inside the K loop: each stages only what it contracts itself.
//   * A (filter): straight from the packed filter (tbg_weight_pack_x3 / _bf16: 16-byte units of 8 channels of one output channel)
//     into VGPRs -- lane (m, K half) loads its own MFMA operand, 512 contiguous bytes per half-wave; no LDS at all (with K split
//     over waves no two waves of a block share a filter byte, so an LDS copy would only add a write and a read per byte);
//   * B (activations): from the unit tensor.  Pixels are tiled in FLATTENED order n = (b Ho + y) Wo + x (a 2 x 25 map wastes nothing:
//     800 pixels = 25 tiles), so a tile is a list of row segments; per filter row the wave needs, for every segment, the padded
//     columns xa .. xb + 2 -- "slot" s of the row.  One 16-byte load per lane fetches slot s (per-lane unit addresses computed
//     once: the zero ring of the unit tensor serves every padding case), one ds_write_b128 puts it into the wave's PRIVATE LDS row,
//     and the tap kw operand of pixel p is the ds_read_b128 of slot q(p) + kw.  1 x 1 filters (stride 1 / 2, and the
//     transposed-strided form whose in-between pixels read a ring unit = zero) use the same path with one slot per pixel.
//   * software pipeline per wave (plain loads: hipcc counts vmcnt itself): loads of row unit s + 2 are issued before the MFMAs of
//     unit s, the LDS row of unit s + 1 is written and its first operands read under the last tap of unit s.
//   * the 8 partial tiles meet in LDS (16 B per lane and row group, summed in wave order: deterministic), and 4 TN waves run
//     conv_epilogue (bias / noise / LeakyReLU / demodulation / residual / gate / dot slots / unit sink) on one 4-channel row group each.
// Arithmetic: the f32x3 / bf16 term pairing of conv_units_fprop_kernel (conv_units.hip).
#include <initializer_list>
#include <type_traits>
#include <utility>

#include "conv_common.h"

// TAP-LIST form (the KW = 1 instantiations): a row unit is (chunk, tap) and a tap is an input offset (dy, dx) + a filter tap index,
// so the same kernel runs 1x1 convolutions, k x k STRIDED VALID convolutions (conv_downsample_2d's strided convolution,
// upfirdn_2d_v2.py:106-113: tap (kh, kw) at offset (kh, kw) from the input pixel (sy y, sx x)) and k x k stride-2 TRANSPOSED convolutions
// (upsample_conv_2d, :65-103, and the data gradients of the strided layers) as their sy sx output-parity classes: class (cy, cx)
// owns the outputs (sy u + cy, sx v + cx) = sum over taps (cy + sy i, cx + sx j) of x[u - i, v - j] -- a dense few-tap correlation on
// the class grid, one block range per class in ONE launch.  A tap whose source falls outside the padded plane (ring included) reads
// the sample's ring corner instead (a zero unit).
struct SmallCls {
  int Ug, Vg, ooy, oox;  // class grid (tiling domain) and where it sits in the output: (osy u + ooy, osx v + oox)
  int ntaps, blk0;       // taps of the class; first block of the class in the launch
  signed char dy[9], dx[9], wt[9];  // per tap: input offset from the pixel's source (padded coordinates), filter tap
};

struct ConvSmallP {
  const char *XU, *Wf;
  long long x_plane, w_plane;  // 16-byte units per plane
  float *y;
  int B, C8, M, Hin, Win, Hout, Wout, ldw;
  int flip, mode;        // mode 0: 3x3 stride 1 pad 1 (row units) | 1: tap list | 2: tap list, 1x1 transposed stride (isy, isx)
  int isy, isx;          // tap list: the pixel (u, v) of the class grid reads from the padded input position (isy u + 1, isx v + 1)
  int osy, osx, ncls;
  int tilesM, nchunks, dot_slots;
  SmallCls cls[4];
  EpiK e;
};

template <int V> using IC = std::integral_constant<int, V>;

// prefetch distance of the per-wave software pipeline in row units (ring of SMALL_PD + 1 register sets), and whether the blocks of
// one channel tile walk their row units from different starting points (tools/ab_small.sh measures both)
#ifndef SMALL_PD
#define SMALL_PD 2
#endif
#ifndef SMALL_ROT
#define SMALL_ROT 0
#endif
// ablation builds (tools/ab_small.sh; 0 = product): 1 no MFMAs, 2 filter loads of the first row unit only, 3 activation loads of the
// first row unit only, 4 both
#ifndef SMALL_EXP
#define SMALL_EXP 0
#endif
// scheduling experiments (0 = product): 1 pin the loads of a row unit in front of its MFMAs, 2 also raise the wave's priority over the MFMAs
#ifndef SMALL_SCHED
#define SMALL_SCHED 0
#endif

template <class F, int... I>
__device__ __forceinline__ void small_ring(F &body, int s, int count, std::integer_sequence<int, I...>) {
  (void)std::initializer_list<int>{(I < count ? (body(IC<I>{}, s + I), 0) : 0)...};
}

template <int NP, int KW, int TN>
__global__ __launch_bounds__(512, 2) void conv_small_kernel(const ConvSmallP p) {
  constexpr int KH = KW, KK = KH * KW;
  constexpr int CKU = NP == 3 ? 1 : 2;             // channel units per chunk
  constexpr int ROWS = NP == 3 ? 3 : 2;            // LDS rows of one row unit: planes (x3) | the chunk's two channel units (bf16)
  constexpr int LPU = KW == 3 ? TN : 1;            // 64-slot loads per row
  constexpr int SLOTS = LPU * 64;
  constexpr int STG = ROWS * SLOTS * 16;           // bytes of one wave's staging row set
  constexpr int NA = NP == 3 ? 2 : 1, NB = (NP == 3 ? 3 : 1) * TN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // consecutive block ids sit on consecutive XCDs: the channel tile varies fastest, so an XCD's L2 sees few filter slices
  int c = 0;
  if constexpr (KW == 1)
    while (c + 1 < p.ncls && (int)blockIdx.x >= p.cls[c + 1].blk0) ++c;
  const SmallCls &cl = p.cls[c];
  const int bid = blockIdx.x - cl.blk0;
  const int mt = bid % p.tilesM, nt = bid / p.tilesM;
  const int m0 = mt * 32, n0 = nt * 32 * TN;
  const int Hp = p.Hin + 2, Wp = p.Win + 2, HWo = p.Hout * p.Wout;
  const int Ug = cl.Ug, Vg = cl.Vg, HWg = Ug * Vg, Ntot = p.B * HWg;  // (KW = 3: the class grid is the output grid)

  // ---- per-lane sources
  const char *bsrc[LPU];  // this lane's slot of the row unit (chunk 0, kh 0 / tap offset 0, plane 0)
  const char *bzero = nullptr;  // tap list: the ring corner of this lane's sample (a zero unit)
  int pu = -4, pv = -4;         // tap list: padded input position of the lane's pixel (-4: no source at all)
  const int r0 = n0 / Vg, x0 = n0 - r0 * Vg;
#pragma unroll
  for (int jj = 0; jj < LPU; ++jj) {
    long long u;
    if constexpr (KW == 3) {
      const int t = jj * 64 + lane + x0;
      const int g = t / Wp, pc = t - g * Wp;
      const int r = min(r0 + g, p.B * p.Hout - 1);
      const int b = r / p.Hout, y = r - b * p.Hout;
      u = ((long long)b * p.C8 * Hp + y) * Wp + pc;
    } else {
      const int n = min(n0 + min(lane, 32 * TN - 1), Ntot - 1);
      const int b = n / HWg, rem = n - b * HWg;
      const int y = rem / Vg, x = rem - y * Vg;
      const long long z = (long long)b * p.C8 * Hp * Wp;  // (0, 0) of the ring
      bzero = p.XU + (z << 4);
      if (p.mode == 1) {
        pu = y * p.isy + 1; pv = x * p.isx + 1;
      } else {  // 1x1 transposed: output pixel (y, x) has a source only where both coordinates are multiples of the stride
        const int ys = y / p.isy, xs = x / p.isx;
        const bool hit = ys * p.isy == y && xs * p.isx == x && ys < p.Hin && xs < p.Win;
        pu = hit ? ys + 1 : -4; pv = hit ? xs + 1 : -4;
      }
      u = z + (long long)pu * Wp + pv;
    }
    bsrc[jj] = p.XU + (u << 4);
  }
  const int mcl = min(m0 + l31, p.M - 1);  // rows past M re-read the last one (never stored)
  // A loads of one row unit.  bf16: one per tap (half-wave h: channel unit h of the chunk).  x3, 1x1: (hi | mid), (hi | lo).
  // x3, 3x3: the K halves of the three MFMAs of a tap carry (A | A') x (B | B') = (hi|mid) x (hi|hi), (hi|mid) x (mid|mid),
  // (hi|lo) x (lo|hi) -- half 0 only ever needs hi, half 1 needs mid and lo.  Swapping the halves' roles on the middle tap --
  // (mid|hi) x (hi|hi), (mid|hi) x (mid|mid), (lo|hi) x (hi|lo): the same six products -- lets FIVE loads carry the nine plane
  // slices of a row unit instead of six (hi is no longer fetched twice per tap):
  //     load      0        1        2        3        4
  //     half 0   hi t0    mid t1   lo t1    hi t2    (hi t2)
  //     half 1   mid t0   lo t0    hi t1    mid t2   lo t2
  constexpr int NAL = NP == 3 ? (KW == 3 ? 5 : 2) : KW;
  const char *const abase = p.Wf + ((long long)mcl << 4);
  unsigned aoff[NAL];  // this lane's byte offset of load i from the row unit's (chunk, first tap) slice of plane 0
  {
    const long long tstep = (long long)(p.flip ? -1 : 1) * p.C8 * p.ldw;  // units between two taps of a filter row
    if constexpr (NP == 3 && KW == 3) {
      const int pl[2][5] = {{0, 1, 2, 0, 0}, {1, 2, 0, 1, 2}}, tp[2][5] = {{0, 1, 1, 2, 2}, {0, 0, 1, 2, 2}};
#pragma unroll
      for (int i = 0; i < 5; ++i) aoff[i] = (unsigned)((pl[half][i] * p.w_plane + tp[half][i] * tstep + 2 * p.C8 * p.ldw) << 4);
    } else if constexpr (NP == 3) {
      aoff[0] = (unsigned)((half * p.w_plane) << 4);
      aoff[1] = (unsigned)((2 * half * p.w_plane) << 4);
    } else {
#pragma unroll
      for (int i = 0; i < KW; ++i) aoff[i] = (unsigned)(((long long)half * p.ldw + i * tstep + (KW - 1) * p.C8 * p.ldw) << 4);
    }
  }
  // operand slots of this lane's pixels
  int q[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int pp = 32 * j + l31;
    q[j] = KW == 3 ? pp + 2 * ((n0 + pp) / Vg - r0) : pp;
  }
  // slots of a row that some pixel of the tile reads: up to q of the last pixel + KW - 1
  const int nslot = (KW == 3 ? 32 * TN - 1 + 2 * ((n0 + 32 * TN - 1) / Vg - r0) : 32 * TN - 1) + KW;
  char *const stg = smem + wave * STG;
  const long long cu_step = (long long)Hp * Wp * 16;  // bytes between two channel units of a sample

  f32x16 acc[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int nunits = p.nchunks * (KW == 3 ? 3 : cl.ntaps);
  const int n_w = wave < nunits ? (nunits - wave + 7) >> 3 : 0;  // row units of this wave
  constexpr int PD = SMALL_PD, R = PD + 1;
  bf16x8 Areg[R][NAL], Breg[R][ROWS][LPU], Op[R > 3 ? R : 3][NB];
  const int rot = SMALL_ROT ? nt : 0;  // (deterministic: the summation order is a function of the tile)

  auto issue = [&](auto S_, int s) {  // every global load of row unit s (clamped: past the end the last unit is fetched again)
    constexpr int S = decltype(S_)::value;
    int si = min(s, n_w - 1) + rot;
    si -= (si / n_w) * n_w;
    const int u = wave + 8 * si;
    if constexpr (KW == 3) {
      const int kc = u / 3, kh = u - kc * 3;
#pragma unroll
      for (int row = 0; row < ROWS; ++row) {
        const long long off = NP == 3 ? ((row * p.x_plane) << 4) + kc * cu_step + (long long)kh * Wp * 16
                                      : (2 * kc + row) * cu_step + (long long)kh * Wp * 16;
#pragma unroll
        for (int jj = 0; jj < LPU; ++jj)
          if (jj * 64 + lane < nslot && !((SMALL_EXP == 3 || SMALL_EXP == 4) && s > 1)) Breg[S][row][jj] = *reinterpret_cast<const bf16x8 *>(bsrc[jj] + off);  // (slots past the tile's last: never read)
      }
      // (aoff is biased by (KW - 1) taps so that it stays non-negative with flip: the row unit's base is its first tap minus that)
      const int t0 = kh * KW, tt0 = p.flip ? KK - 1 - t0 : t0;
      const char *const ab = abase + (((long long)(tt0 - (KW - 1)) * p.C8 + CKU * kc) * p.ldw << 4);
#pragma unroll
      for (int i = 0; i < NAL; ++i)
        if (!((SMALL_EXP == 2 || SMALL_EXP == 4) && s > 1)) Areg[S][i] = *reinterpret_cast<const bf16x8 *>(ab + aoff[i]);
    } else {
      const int kc = u / cl.ntaps, ti = u - kc * cl.ntaps;
      const int dy = cl.dy[ti], dx = cl.dx[ti], wt = cl.wt[ti];
      const bool ok = (unsigned)(pu + dy) < (unsigned)Hp && (unsigned)(pv + dx) < (unsigned)Wp;
      const char *const src = ok ? bsrc[0] + ((long long)(dy * Wp + dx) << 4) : bzero;
#pragma unroll
      for (int row = 0; row < ROWS; ++row) {
        const long long off = NP == 3 ? ((row * p.x_plane) << 4) + kc * cu_step : (2 * kc + row) * cu_step;
        if (lane < nslot) Breg[S][row][0] = *reinterpret_cast<const bf16x8 *>(src + off);
      }
      const char *const ab = abase + (((long long)wt * p.C8 + CKU * kc) * p.ldw << 4);
#pragma unroll
      for (int i = 0; i < NAL; ++i) Areg[S][i] = *reinterpret_cast<const bf16x8 *>(ab + aoff[i]);
    }
  };
  auto stage = [&](auto S_) {  // the row unit's slots -> this wave's LDS rows
    constexpr int S = decltype(S_)::value;
#pragma unroll
    for (int row = 0; row < ROWS; ++row)
#pragma unroll
      for (int jj = 0; jj < LPU; ++jj)
        *reinterpret_cast<bf16x8 *>(stg + ((row * SLOTS + jj * 64 + lane) << 4)) = Breg[S][row][jj];
  };
  // x3: half-wave h supplies K half h of each MFMA:  A (hi | mid) x B hi,  A (hi | mid) x B mid,  A (hi | lo) x B (lo | hi); on the
  // middle tap of a 3x3 row the halves swap roles (see the A loads): B (hi | lo) there.  bf16: half-wave h holds channel unit h
  auto read_ops = [&](auto O_, int kw) {
    constexpr int O = decltype(O_)::value;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const char *base = stg + ((q[j] + kw) << 4);
      if constexpr (NP == 3) {
        Op[O][3 * j + 0] = *reinterpret_cast<const bf16x8 *>(base);
        Op[O][3 * j + 1] = *reinterpret_cast<const bf16x8 *>(base + SLOTS * 16);
        Op[O][3 * j + 2] = *reinterpret_cast<const bf16x8 *>(base + 2 * ((KW == 3 && kw == 1) ? half : 1 - half) * SLOTS * 16);
      } else {
        Op[O][j] = *reinterpret_cast<const bf16x8 *>(base + half * SLOTS * 16);
      }
    }
  };
  auto sel = [&](const bf16x8 &h1, const bf16x8 &h0) {  // half-wave 1 takes h1, half-wave 0 takes h0
    const u32x4v a = __builtin_bit_cast(u32x4v, h1), b = __builtin_bit_cast(u32x4v, h0);
    return __builtin_bit_cast(bf16x8, u32x4v{half ? a[0] : b[0], half ? a[1] : b[1], half ? a[2] : b[2], half ? a[3] : b[3]});
  };
  auto mfmas = [&](auto S_, auto O_, auto KW_) {
    constexpr int S = decltype(S_)::value, O = decltype(O_)::value, kw = decltype(KW_)::value;
    if constexpr (SMALL_EXP == 1) {
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[j][0] += (float)Areg[S][0][0] * (float)Op[O][0][j];
    } else if constexpr (NP == 3) {  // smallest terms first
      bf16x8 X, Y;
      if constexpr (KW == 3) {
        if constexpr (kw == 0) { X = Areg[S][0]; Y = sel(Areg[S][1], Areg[S][0]); }
        else if constexpr (kw == 1) { X = sel(Areg[S][2], Areg[S][1]); Y = Areg[S][2]; }
        else { X = Areg[S][3]; Y = sel(Areg[S][4], Areg[S][3]); }
      } else {
        X = Areg[S][0]; Y = Areg[S][1];
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Y, Op[O][3 * j + 2], acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X, Op[O][3 * j + 1], acc[j], 0, 0, 0);
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(X, Op[O][3 * j + 0], acc[j], 0, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Areg[S][kw], Op[O][j], acc[j], 0, 0, 0);
    }
  };
  // one row unit: ring slot S holds its operands; the loads of unit s + PD go out first, the LDS rows of unit s + 1 are written (and
  // its first operands read) under the last tap
  auto body = [&](auto S_, int s) {
    constexpr int S = decltype(S_)::value, S1 = (S + 1) % R, SP = (S + PD) % R;
    issue(IC<SP>{}, s + PD);
    if constexpr (SMALL_SCHED >= 1) __builtin_amdgcn_sched_barrier(0);
    if constexpr (SMALL_SCHED >= 2) __builtin_amdgcn_s_setprio(2);
    if constexpr (KW == 3) {
      read_ops(IC<1>{}, 1);
      mfmas(IC<S>{}, IC<0>{}, IC<0>{});
      read_ops(IC<2>{}, 2);
      mfmas(IC<S>{}, IC<1>{}, IC<1>{});
      stage(IC<S1>{});
      read_ops(IC<0>{}, 0);
      mfmas(IC<S>{}, IC<2>{}, IC<2>{});
    } else {
      stage(IC<S1>{});
      read_ops(IC<S1>{}, 0);
      mfmas(IC<S>{}, IC<S>{}, IC<0>{});
    }
    if constexpr (SMALL_SCHED >= 2) __builtin_amdgcn_s_setprio(0);
  };
  if (n_w > 0) {
    auto pro = [&](auto S_, int s) { issue(S_, s); };
    small_ring(pro, 0, PD, std::make_integer_sequence<int, R>{});
    stage(IC<0>{});
    read_ops(IC<0>{}, 0);
    // whole rounds of the ring in the loop, the ragged rest behind it: the loop header then merges only the entry and the full
    // back edge (with the rest inside the loop hipcc's vmcnt state at the header took the early-exit paths into account and
    // waited for all but 4 loads at the top of every unit)
    int s = 0;
    for (; s + R <= n_w; s += R) small_ring(body, s, R, std::make_integer_sequence<int, R>{});
    small_ring(body, s, n_w - s, std::make_integer_sequence<int, R>{});
  }

  // ---- the 8 partial tiles -> LDS, summed in wave order by the wave that finishes the row group
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  f32x4v *const red = reinterpret_cast<f32x4v *>(smem + 8 * STG);
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg)
      red[((wave * TN + j) * 4 + rg) * 64 + lane] = f32x4v{acc[j][4 * rg], acc[j][4 * rg + 1], acc[j][4 * rg + 2], acc[j][4 * rg + 3]};
  __syncthreads();
  if (wave >= 4 * TN) return;
  const int j = wave >> 2, rg = wave & 3;
  f32x4v v = red[(j * 4 + rg) * 64 + lane];
#pragma unroll
  for (int w2 = 1; w2 < 8; ++w2) v += red[((w2 * TN + j) * 4 + rg) * 64 + lane];
  f32x16 a1[1][1];
#pragma unroll
  for (int r = 0; r < 16; ++r) a1[0][0][r] = r < 4 ? v[r & 3] : 0.f;
  const int nb = n0 + 32 * j, n = nb + l31;
  int e_pix[1], e_b[1];
  e_b[0] = min(n, Ntot - 1) / HWg;
  const int rem = n - e_b[0] * HWg;
  if (KW == 3 || (p.osy == 1 && p.osx == 1)) {
    e_pix[0] = n < Ntot ? rem : -1;
  } else {  // a class of a transposed convolution: its grid sits strided in the output
    const int u = rem / Vg, v = rem - u * Vg;
    const int Y = u * p.osy + cl.ooy, X = v * p.osx + cl.oox;
    e_pix[0] = (n < Ntot && Y < p.Hout && X < p.Wout) ? Y * p.Wout + X : -1;
  }
  const int dot_b = nb / HWg;
  conv_epilogue<1, 1, 4, true, 4, 3, true>(a1, p.e, p.y, nullptr, p.M, HWo, m0 + 8 * rg, lane, e_pix, e_b, true, dot_b, p.dot_slots,
                                  (nb - dot_b * HWg) >> 5, p.Hout, p.Wout);
}

// ---- geometry: which form of the kernel a descriptor takes (-1: none)
//   0  3x3, stride 1, pad 1 (row units)          1  tap list: k x k (k <= 3) pad 0, stride (1|2, 1|2) -- 1x1 and strided VALID layers
//   2  tap list: 1x1 transposed stride           3  tap list: k x k (k = 2, 3) transposed with stride (1|2, 1|2), as output-parity classes
static int small_mode(const tbg_conv_desc *d) {
  if (d->ksplit != 1 || d->ldw < d->M || d->B < 1 || d->C < 1 || d->M < 1) return -1;
  if (d->KH == 3 && d->KW == 3 && !d->transposed && d->sy == 1 && d->sx == 1 && d->py == 1 && d->px == 1 && d->Hout == d->Hin &&
      d->Wout == d->Win && d->Wout >= 3)
    return 0;
  if (d->KH < 1 || d->KH > 3 || d->KW < 1 || d->KW > 3 || d->py != 0 || d->px != 0) return -1;
  if (d->sy < 1 || d->sy > 2 || d->sx < 1 || d->sx > 2) return -1;
  if (!d->transposed) {
    if (d->Hin < d->KH || d->Win < d->KW) return -1;
    return (d->Hout == (d->Hin - d->KH) / d->sy + 1 && d->Wout == (d->Win - d->KW) / d->sx + 1) ? 1 : -1;
  }
  if (d->KH == 1 && d->KW == 1) return (d->Hout >= (d->Hin - 1) * d->sy + 1 && d->Wout >= (d->Win - 1) * d->sx + 1) ? 2 : -1;
  return (d->Hout >= (d->Hin - 1) * d->sy + d->KH && d->Wout >= (d->Win - 1) * d->sx + d->KW) ? 3 : -1;  // (as tbg_conv2d_*)
}

static bool small_ok(const tbg_conv_desc *d, int planes) {
  if (planes != 1 && planes != 3) return false;
  const int c8 = (d->C + 7) / 8;
  return small_mode(d) >= 0 && (planes == 3 || (c8 & 1) == 0);
}

// the classes of a descriptor: grids, taps; returns the number of classes
static int small_classes(const tbg_conv_desc *d, int mode, SmallCls *cls) {
  const int T = d->KH * d->KW;
  if (mode != 3) {
    SmallCls &c = cls[0];
    c = SmallCls{};
    c.Ug = d->Hout; c.Vg = d->Wout; c.ooy = 0; c.oox = 0; c.blk0 = 0;
    c.ntaps = mode == 0 ? 9 : T;
    for (int t = 0; t < T && mode == 1; ++t) {
      c.dy[t] = (signed char)(t / d->KW); c.dx[t] = (signed char)(t % d->KW); c.wt[t] = (signed char)(d->flip ? T - 1 - t : t);
    }
    return 1;
  }
  int k = 0;
  for (int cy = 0; cy < d->sy; ++cy)
    for (int cx = 0; cx < d->sx; ++cx) {
      const int KHc = cy < d->KH ? ceil_div(d->KH - cy, d->sy) : 0, KWc = cx < d->KW ? ceil_div(d->KW - cx, d->sx) : 0;
      SmallCls &c = cls[k];
      c = SmallCls{};
      c.Ug = d->Hout > cy ? ceil_div(d->Hout - cy, d->sy) : 0;
      c.Vg = d->Wout > cx ? ceil_div(d->Wout - cx, d->sx) : 0;
      c.ooy = cy; c.oox = cx;
      if (KHc * KWc < 1 || c.Ug < 1 || c.Vg < 1) continue;  // (a class without taps cannot exist for k >= stride)
      for (int i = 0; i < KHc; ++i)
        for (int j = 0; j < KWc; ++j) {
          const int t = (cy + d->sy * i) * d->KW + cx + d->sx * j;
          c.dy[c.ntaps] = (signed char)-i; c.dx[c.ntaps] = (signed char)-j; c.wt[c.ntaps] = (signed char)(d->flip ? T - 1 - t : t);
          ++c.ntaps;
        }
      ++k;
    }
  return k;
}

// pixels per tile: 64 where 32-pixel tiles would be more than one round of one block per CU (every pixel tile re-reads the filter
// slice of its channel tile: two rounds of 32 pixels stream it twice, one round of 64 once)
static long long small_blocks(const tbg_conv_desc *d, int mode, const SmallCls *cls, int ncls, int tn) {
  long long b = 0;
  for (int k = 0; k < ncls; ++k) b += (long long)ceil_div(d->M, 32) * (((long long)d->B * cls[k].Ug * cls[k].Vg + 32 * tn - 1) / (32 * tn));
  return b;
}

static int small_tn(const tbg_conv_desc *d, int mode, const SmallCls *cls, int ncls) {
  return small_blocks(d, mode, cls, ncls, 1) > 256 ? 2 : 1;
}

extern "C" int tbg_conv2d_units_small_blocks(const tbg_conv_desc *d, int planes) {
  if (!d) return TBG_EINVAL;
  if (!small_ok(d, planes)) return TBG_EUNSUPPORTED;
  SmallCls cls[4];
  const int mode = small_mode(d), ncls = small_classes(d, mode, cls);
  const long long b = small_blocks(d, mode, cls, ncls, small_tn(d, mode, cls, ncls));
  return b > 2147483647LL ? TBG_ERANGE : (int)b;
}

// pixels of a block's tile (32 | 64): profile labels, tests
extern "C" int tbg_conv2d_units_small_tile_pixels(const tbg_conv_desc *d, int planes) {
  if (!d) return TBG_EINVAL;
  if (!small_ok(d, planes)) return TBG_EUNSUPPORTED;
  SmallCls cls[4];
  const int mode = small_mode(d), ncls = small_classes(d, mode, cls);
  return 32 * small_tn(d, mode, cls, ncls);
}

// slots of the fused dot product per (b, m): one per 32 pixels of the map; 0 = a tile would straddle two images, or the outputs
// of a tile are not one contiguous pixel range (transposed classes): not served
extern "C" int tbg_conv2d_units_small_dot_slots(const tbg_conv_desc *d, int planes) {
  if (!d) return TBG_EINVAL;
  if (!small_ok(d, planes)) return TBG_EUNSUPPORTED;
  SmallCls cls[4];
  const int mode = small_mode(d), ncls = small_classes(d, mode, cls);
  const int hw = d->Hout * d->Wout;
  return (mode != 3 && hw % (32 * small_tn(d, mode, cls, ncls)) == 0) ? hw / 32 : 0;
}

template <int NP, int KW, int TN>
static int launch_small(const ConvSmallP &p, int blocks, hipStream_t st) {
  constexpr int ROWS = NP == 3 ? 3 : 2, LPU = KW == 3 ? TN : 1;
  const size_t lds = (size_t)8 * ROWS * LPU * 64 * 16 + (size_t)8 * TN * 4096;
  auto kern = conv_small_kernel<NP, KW, TN>;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return TBG_EHIP;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, st, p);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

extern "C" int tbg_conv2d_units_small(const tbg_conv_desc *d, const void *XU, int planes, const void *w, float *y,
                                      const tbg_epilogue *epi, void *stream) {
  if (!d || !XU || !w || (!y && !epi_has_sink(epi)) || (planes != 1 && planes != 3) || !epi_valid(epi)) return TBG_EINVAL;
  if (d->B < 1 || d->C < 1 || d->M < 1 || d->Hin < 1 || d->Win < 1 || d->Hout < 1 || d->Wout < 1) return TBG_EINVAL;
  if (((reinterpret_cast<uintptr_t>(XU) | reinterpret_cast<uintptr_t>(w)) & 15) != 0) return TBG_EINVAL;
  if (!small_ok(d, planes)) return TBG_EUNSUPPORTED;
  if ((double)d->B * d->M * d->Hout * d->Wout > 2147483647.0) return TBG_ERANGE;
  ConvSmallP p{};
  p.XU = reinterpret_cast<const char *>(XU); p.Wf = reinterpret_cast<const char *>(w);
  p.C8 = (d->C + 7) / 8;
  p.x_plane = (long long)d->B * p.C8 * (d->Hin + 2) * (d->Win + 2);
  p.w_plane = (long long)d->KH * d->KW * p.C8 * d->ldw;
  if (p.x_plane * planes > 2147483647LL / 2 || p.w_plane * planes > 2147483647LL / 2) return TBG_ERANGE;
  p.y = y; p.B = d->B; p.M = d->M; p.Hin = d->Hin; p.Win = d->Win; p.Hout = d->Hout; p.Wout = d->Wout; p.ldw = d->ldw;
  p.flip = d->flip;
  const int mode = small_mode(d);
  p.mode = mode == 3 ? 1 : mode;
  p.isy = mode == 3 ? 1 : d->sy; p.isx = mode == 3 ? 1 : d->sx;
  p.osy = mode == 3 ? d->sy : 1; p.osx = mode == 3 ? d->sx : 1;
  p.ncls = small_classes(d, mode, p.cls);
  const int tn = small_tn(d, mode, p.cls, p.ncls);
  p.tilesM = ceil_div(d->M, 32);
  int blocks = 0;
  for (int k = 0; k < p.ncls; ++k) {
    p.cls[k].blk0 = blocks;
    blocks += p.tilesM * ceil_div(d->B * p.cls[k].Ug * p.cls[k].Vg, 32 * tn);
  }
  p.nchunks = p.C8 / (planes == 3 ? 1 : 2);
  p.e = make_epi(epi);
  const int hw = d->Hout * d->Wout;
  p.dot_slots = (mode != 3 && hw % (32 * tn) == 0) ? hw / 32 : 0;
  if (p.e.dot_aux && p.dot_slots == 0) return TBG_EUNSUPPORTED;
  if (p.e.units_out && mode == 3) return TBG_EUNSUPPORTED;  // (a class writes a strided part of the output: no sink)
  if (const int rcs = epi_sink_geometry(p.e, d->B, d->M, d->Hout, d->Wout)) return rcs;
  hipStream_t st = tbg_stream(stream);
  if (mode == 0) {
    if (planes == 3) return tn == 2 ? launch_small<3, 3, 2>(p, blocks, st) : launch_small<3, 3, 1>(p, blocks, st);
    return tn == 2 ? launch_small<1, 3, 2>(p, blocks, st) : launch_small<1, 3, 1>(p, blocks, st);
  }
  if (planes == 3) return tn == 2 ? launch_small<3, 1, 2>(p, blocks, st) : launch_small<3, 1, 1>(p, blocks, st);
  return tn == 2 ? launch_small<1, 1, 2>(p, blocks, st) : launch_small<1, 1, 1>(p, blocks, st);
}
