// What does ds_read_b64_tr_b16 deliver?  lds[i] = i (16-bit elements); lane l reads at element address A(l) and prints the
// four element indices it received.  Run: hipcc --offload-arch=gfx950 tools/tr16_probe.hip -o /tmp/tr16 && /tmp/tr16
// (decides the LDS image of the unit-layout filter gradient)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short *out, int mode) {
  __shared__ short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x;
  // mode 0: lane l supplies element address 4*l (a dense 64 x 4 image)
  // mode 1: 16-lane group g reads a [4 rows][16 cols] block with row pitch 64 elements: lane i of the group -> row i/4, cols 4*(i%4)
  int addr = 4 * l;
  if (mode == 1) { const int g = l >> 4, i = l & 15; addr = g * 1024 + (i >> 2) * 64 + 4 * (i & 3); }
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(lds + addr));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short *d; hipMalloc(&d, 64 * 4 * 2);
  short h[256];
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  }
  return 0;
}
