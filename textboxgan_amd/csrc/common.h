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
};

static inline EpiK make_epi(const tbg_epilogue *e) {
  EpiK k;
  if (e) {
    k.out_scale = e->out_scale; k.bias = e->bias; k.noise = e->noise; k.strength = e->strength;
    k.residual = e->residual; k.dot_aux = e->dot_aux; k.gate = e->gate; k.dot_out = e->dot_out; k.alpha = e->alpha; k.bias_mul = e->bias_mul; k.slope = e->slope;
    k.gain = e->gain; k.res_scale = e->res_scale; k.act = e->act; k.res_first = e->res_first;
  } else {
    k.out_scale = k.bias = k.noise = k.strength = k.residual = k.dot_aux = k.gate = nullptr; k.dot_out = nullptr;
    k.alpha = 1.f; k.bias_mul = 1.f; k.slope = 1.f; k.gain = 1.f; k.res_scale = 1.f; k.act = TBG_ACT_LINEAR; k.res_first = 0;
  }
  return k;
}

static inline bool epi_valid(const tbg_epilogue *e) {
  if (!e) return true;
  if (e->noise && !e->strength) return false;
  if ((e->dot_aux != nullptr) != (e->dot_out != nullptr)) return false;
  if (e->gate && e->dot_aux) return false;
  if (e->act != TBG_ACT_LINEAR && e->act != TBG_ACT_LRELU) return false;
  return true;
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
