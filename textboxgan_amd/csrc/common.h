// Shared helpers for the gfx950 kernels of libtbg_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tbg.h"

#define TBG_LAUNCH_CHECK()                          \
  do {                                              \
    if (hipGetLastError() != hipSuccess) return TBG_EHIP; \
  } while (0)

static inline hipStream_t tbg_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// device-side copy of the epilogue (plain struct, passed by value in kernargs)
struct EpiK {
  const float *out_scale, *bias, *noise, *strength, *residual, *dot_aux, *gate;
  float *dot_out;
  float alpha, bias_mul, slope, gain, res_scale;
  int act, res_first;
  void *units_out;            // unit sink (tbg.h): units(out * units_scale) beside / instead of the fp32 output
  const float *units_scale;
  int units_planes;
  long long units_plane;      // bytes between two planes of units_out (set by the entry: epi_sink_geometry)
};

static inline EpiK make_epi(const tbg_epilogue *e) {
  EpiK k;
  if (e) {
    k.out_scale = e->out_scale; k.bias = e->bias; k.noise = e->noise; k.strength = e->strength;
    k.residual = e->residual; k.dot_aux = e->dot_aux; k.gate = e->gate; k.dot_out = e->dot_out; k.alpha = e->alpha; k.bias_mul = e->bias_mul; k.slope = e->slope;
    k.gain = e->gain; k.res_scale = e->res_scale; k.act = e->act; k.res_first = e->res_first;
    k.units_out = e->units_out; k.units_scale = e->units_scale; k.units_planes = e->units_planes; k.units_plane = 0;
  } else {
    k.out_scale = k.bias = k.noise = k.strength = k.residual = k.dot_aux = k.gate = nullptr; k.dot_out = nullptr;
    k.alpha = 1.f; k.bias_mul = 1.f; k.slope = 1.f; k.gain = 1.f; k.res_scale = 1.f; k.act = TBG_ACT_LINEAR; k.res_first = 0;
    k.units_out = nullptr; k.units_scale = nullptr; k.units_planes = 0; k.units_plane = 0;
  }
  return k;
}

static inline bool epi_valid(const tbg_epilogue *e) {
  if (!e) return true;
  if (e->noise && !e->strength) return false;
  if ((e->dot_aux != nullptr) != (e->dot_out != nullptr)) return false;
  if (e->gate && e->dot_aux) return false;
  if (e->act != TBG_ACT_LINEAR && e->act != TBG_ACT_LRELU) return false;
  if (e->units_out) {
    if (e->units_planes != 1 && e->units_planes != 3) return false;
    if ((reinterpret_cast<uintptr_t>(e->units_out) & 15) != 0) return false;
  } else if (e->units_scale) {
    return false;
  }
  return true;
}

// entries that do not serve a unit sink reject it (tbg.h)
static inline bool epi_has_sink(const tbg_epilogue *e) { return e && e->units_out; }

// an entry that serves the sink: fix the plane pitch of the [B, M, Hout, Wout] unit tensor (TBG_EINVAL: M % 8 != 0, TBG_ERANGE:
// more than 2^31 elements)
static inline int epi_sink_geometry(EpiK &k, int B, int M, int Hout, int Wout) {
  if (!k.units_out) return TBG_OK;
  if ((M & 7) != 0) return TBG_EINVAL;
  const long long units = (long long)B * (M >> 3) * (Hout + 2) * (Wout + 2);
  if (units * 8 > 2147483647LL) return TBG_ERANGE;
  k.units_plane = units * 16;
  return TBG_OK;
}

// v: accumulator already multiplied by alpha*out_scale by the caller when convenient
__device__ __forceinline__ float epi_act(const EpiK &e, float pre) {
  float v = (e.act == TBG_ACT_LRELU) ? (pre > 0.f ? pre : pre * e.slope) : pre;
  return v * e.gain;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// N wave-wide sums at once (N a power of two <= 64): at distance OFF a lane hands its partner the half of the values the partner keeps
// and adds the partner's copy of the half it keeps itself, then the same on the kept half at OFF / 2 -- N - 1 exchanges plus one per
// remaining distance, in six steps of independent instructions, instead of N chains of six dependent exchanges.  Afterwards a[0] of lane l
// is the total of value wave_tree_row<N>(l); lanes that differ only in the low (plain-sum) bits hold the same total.
template <int NT, int N, int OFF>
__device__ __forceinline__ void wave_tree_sum(float (&a)[NT], const int lane) {
  if constexpr (OFF > 0) {
    if constexpr (N > 1) {
      constexpr int H = N / 2;
      const bool up = (lane & OFF) != 0;
#pragma unroll
      for (int k = 0; k < H; ++k) {
        const float send = up ? a[k] : a[k + H], keep = up ? a[k + H] : a[k];
        a[k] = keep + __shfl_xor(send, OFF, 64);
      }
      wave_tree_sum<NT, H, OFF / 2>(a, lane);
    } else {
      a[0] += __shfl_xor(a[0], OFF, 64);
      wave_tree_sum<NT, 1, OFF / 2>(a, lane);
    }
  }
}
template <int N>
__device__ __forceinline__ int wave_tree_row(const int lane) {  // which of the N values lane `lane` holds after wave_tree_sum<N, N, 32>
  int k = 0, nn = N;
  for (int off = 32; off > 0; off >>= 1)
    if (nn > 1) { nn >>= 1; k += (lane & off) ? nn : 0; }
  return k;
}
