// Practical fp32 MFMA ceiling of the chip: back-to-back v_mfma_f32_32x32x2_f32 on 4 independent accumulators,
// no memory traffic in the timed loop.  mode 0: constant operands; mode 1: random operands (16 per lane, cycled)
// -- data toggling raises power, which shows whether the ceiling is clock/power-limited under realistic data.
// usage: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float *out, const float *rnd, int iters, int mode) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = mode ? rnd[(threadIdx.x * 16 + i) & 4095] : 1.f;
    b[i] = mode ? rnd[(threadIdx.x * 16 + 8 + i) & 4095] : 2.f;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u + i) & 7], b[(u + 2 * i) & 7], acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float *out, *rnd; hipMalloc(&out, 256 * 8192 * 4); hipMalloc(&rnd, 4096 * 4);
  float h[4096]; srand(1); for (int i = 0; i < 4096; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * 1e-3f;
  hipMemcpy(rnd, h, sizeof(h), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode : {0, 1})
    for (int wpb : {1, 2, 3}) {           // blocks per CU (= waves per SIMD)
      const int blocks = 256 * wpb, iters = 10000;
      k<<<blocks, 256>>>(out, rnd, 1000, mode); hipDeviceSynchronize();
      hipEventRecord(e0); k<<<blocks, 256>>>(out, rnd, iters, mode); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double fl = (double)blocks * 4 * iters * 32 * 4096.0;
      printf("%s operands, waves/SIMD=%d  %.3f ms  %.1f TFLOP/s  (implied clock at 64 flop/clk/SIMD: %.2f GHz)\n",
             mode ? "random  " : "constant", wpb, ms, fl / ms / 1e9, fl / ms / 1e9 * 1e12 / (64.0 * 1024) / 1e9);
    }
  return 0;
}
