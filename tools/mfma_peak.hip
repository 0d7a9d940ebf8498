// Practical fp32 MFMA ceiling of the chip: back-to-back v_mfma_f32_32x32x2_f32 on 4 independent accumulators,
// no memory traffic.  usage: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, float b0) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float *out; hipMalloc(&out, 256 * 8192 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wpb : {1, 2, 3}) {           // blocks per CU (waves per SIMD)
    const int blocks = 256 * wpb, iters = 20000;
    k<<<blocks, 256>>>(out, 1000, 1.f, 2.f); hipDeviceSynchronize();
    hipEventRecord(e0); k<<<blocks, 256>>>(out, iters, 1.f, 2.f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)blocks * 4 * iters * 16 * 4096.0;
    printf("waves/SIMD=%d  %.3f ms  %.1f TFLOP/s  (implied clock at 64 flop/clk/SIMD: %.2f GHz)\n", wpb, ms, fl / ms / 1e9,
           fl / ms / 1e9 * 1e12 / (64.0 * 1024) / 1e9);
  }
  return 0;
}
