// HBM-bound "thin" 1x1 convolutions of the RGB ends of both networks (gfx950).
//   toRGB   (layers/to_rgb.py:28-33): C feature maps -> 3 channels, style-modulated, + bias + skip image
//   fromRGB (layers/from_rgb.py:26-29) gradients: 3 <-> 64 channels
// One operand has <= 4 channels, so the contraction is a streaming pass over the wide tensor: these
// kernels read it exactly once with 16-byte loads (1 KiB per wave instruction) instead of pushing a
// 3-wide GEMM through 32x32 MFMA tiles.
#include "common.h"

#define RGB_MAXO 4

struct RgbP {
  const float *x, *w, *scale, *bias, *skip, *dy, *colmask;
  float *y, *dx, *G, *dym, *dysum;
  int B, C, O, ldw, HW;
  int maskW, maskCW, maskN;  // colmask [B][maskN]: pixel p of a row-major map of width maskW takes colmask[b][(p % maskW) / maskCW]
  float alpha, bias_mul;
};

__device__ __forceinline__ float rgb_mask_at(const RgbP &p, int b, int pix) {
  return p.colmask[b * p.maskN + (pix % p.maskW) / p.maskCW];
}

// y[b,o,p] = (alpha * sum_c x[b,c,p] * w[c,o] * scale[b,c] + bias[o]*bias_mul + skip[b,o,p]) * colmask[b, column(p)]
// Streaming kernel: the block owns 64 pixel quads of one image; its four waves split the channels (wave v takes c = v, v+4,
// ...: 8 independent 16-byte loads in flight per lane) and the four partial sums meet in LDS.  1024 blocks on the 64x256 map
// at B = 16 (the one-wave-per-quad form launched 256 blocks of serial 128-channel walks and reached 1.6 TB/s).
__global__ __launch_bounds__(256) void rgb_project_kernel(const RgbP p) {
  extern __shared__ __attribute__((aligned(16))) float wsm[];  // [C][RGB_MAXO] effective weights, then [3][64][RGB_MAXO][4] partials
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < p.C * RGB_MAXO; i += 256) {
    const int c = i / RGB_MAXO, o = i - c * RGB_MAXO;
    float v = 0.f;
    if (o < p.O) v = p.w[c * p.ldw + o] * p.alpha * (p.scale ? p.scale[b * p.C + c] : 1.f);
    wsm[i] = v;
  }
  __syncthreads();
  float *part = wsm + ((p.C * RGB_MAXO + 3) & ~3);
  const int q = blockIdx.x * 64 + lane;  // pixel quad
  const int p0 = q * 4;
  const bool live = p0 < p.HW;
  const bool vec = (p.HW & 3) == 0;
  float acc[RGB_MAXO][4];
#pragma unroll
  for (int o = 0; o < RGB_MAXO; ++o)
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[o][e] = 0.f;
  const float *xb = p.x + (size_t)b * p.C * p.HW + p0;
  if (live) {
    if (vec) {
#pragma unroll 8
      for (int c = wave; c < p.C; c += 4) {
        const float4 xv = *reinterpret_cast<const float4 *>(xb + (size_t)c * p.HW);
        const float4 wv = *reinterpret_cast<const float4 *>(wsm + c * RGB_MAXO);
        const float wo[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
        for (int o = 0; o < RGB_MAXO; ++o) {
          acc[o][0] += xv.x * wo[o]; acc[o][1] += xv.y * wo[o]; acc[o][2] += xv.z * wo[o]; acc[o][3] += xv.w * wo[o];
        }
      }
    } else {
      for (int c = wave; c < p.C; c += 4)
        for (int e = 0; e < 4; ++e)
          if (p0 + e < p.HW) {
            const float xv = xb[(size_t)c * p.HW + e];
            for (int o = 0; o < RGB_MAXO; ++o) acc[o][e] += xv * wsm[c * RGB_MAXO + o];
          }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int o = 0; o < RGB_MAXO; ++o)
      *reinterpret_cast<float4 *>(part + (((wave - 1) * 64 + lane) * RGB_MAXO + o) * 4) =
          make_float4(acc[o][0], acc[o][1], acc[o][2], acc[o][3]);
  }
  __syncthreads();
  if (wave > 0 || !live) return;
#pragma unroll
  for (int v = 0; v < 3; ++v)
#pragma unroll
    for (int o = 0; o < RGB_MAXO; ++o) {
      const float4 t = *reinterpret_cast<const float4 *>(part + ((v * 64 + lane) * RGB_MAXO + o) * 4);
      acc[o][0] += t.x; acc[o][1] += t.y; acc[o][2] += t.z; acc[o][3] += t.w;
    }
  float mk[4] = {1.f, 1.f, 1.f, 1.f};
  if (p.colmask) {
#pragma unroll
    for (int e = 0; e < 4; ++e) mk[e] = (p0 + e < p.HW) ? rgb_mask_at(p, b, p0 + e) : 0.f;
  }
  for (int o = 0; o < p.O; ++o) {
    const float bo = p.bias ? p.bias[o] * p.bias_mul : 0.f;
    const size_t off = ((size_t)b * p.O + o) * p.HW + p0;
    if (vec) {
      float4 r = make_float4(acc[o][0] + bo, acc[o][1] + bo, acc[o][2] + bo, acc[o][3] + bo);
      if (p.skip) {
        const float4 s = *reinterpret_cast<const float4 *>(p.skip + off);
        r.x += s.x; r.y += s.y; r.z += s.z; r.w += s.w;
      }
      r.x *= mk[0]; r.y *= mk[1]; r.z *= mk[2]; r.w *= mk[3];
      *reinterpret_cast<float4 *>(p.y + off) = r;
    } else {
      for (int e = 0; e < 4; ++e)
        if (p0 + e < p.HW) p.y[off + e] = (acc[o][e] + bo + (p.skip ? p.skip[off + e] : 0.f)) * mk[e];
    }
  }
}

// dx[b,c,p] = alpha * scale[b,c] * sum_o w[c,o] * dym[b,o,p]     (optional)            dym = dy * colmask (or dy)
// G[b,c,chunk,o] = sum_{p in chunk} x[b,c,p] * dym[b,o,p]          (optional; per-pixel-chunk partial sums, plain stores)
// block (pixel chunk, channel group, b): the dy chunk is staged once in LDS (masked on the way in; the channel-group-0 blocks
// also write the masked dy out when dym is given -- the gradient of the skip image and of the bias); every WAVE walks one
// wide channel at a time, so the three partial sums per channel need one shuffle tree per chunk.  G is written per pixel
// chunk (deterministic: no atomics, no zero-fill); the caller sums the nchunk partials.
#define RGB_CHUNK 2048
#define RGB_CPB 8  // channels per block (2 per wave): 2048 blocks on the 64x256 x 128-channel map at B = 16
__global__ __launch_bounds__(256) void rgb_backproject_kernel(const RgbP p) {
  __shared__ __attribute__((aligned(16))) float dys[RGB_MAXO][RGB_CHUNK];
  const int b = blockIdx.z;
  const int pc0 = blockIdx.x * RGB_CHUNK;
  const int npx = min(RGB_CHUNK, p.HW - pc0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < RGB_MAXO * RGB_CHUNK; i += 256) {
    const int o = i / RGB_CHUNK, px = i - o * RGB_CHUNK;
    float v = 0.f;
    if (o < p.O && px < npx) {
      const size_t off = ((size_t)b * p.O + o) * p.HW + pc0 + px;
      v = p.dy[off];
      if (p.colmask) v *= rgb_mask_at(p, b, pc0 + px);
      if (p.dym && blockIdx.y == 0) p.dym[off] = v;
    }
    dys[o][px] = v;
  }
  __syncthreads();
  if (p.dysum && blockIdx.y == 0) {  // sum_p dym[b,o,p] over this pixel chunk (the bias gradient's partial sums): wave o sums row o
    if (wave < p.O) {
      float a = 0.f;
      for (int px = lane; px < RGB_CHUNK; px += 64) a += dys[wave][px];  // (positions past npx hold zeros)
      a = wave_sum(a);
      if (lane == 0) p.dysum[((size_t)b * gridDim.x + blockIdx.x) * p.O + wave] = a;
    }
  }
  const bool vec = (p.HW & 3) == 0;
  for (int cc = wave; cc < RGB_CPB; cc += 4) {
    const int c = blockIdx.y * RGB_CPB + cc;
    if (c >= p.C) break;
    float wv[RGB_MAXO];
    const float sc = p.alpha * (p.scale ? p.scale[b * p.C + c] : 1.f);
#pragma unroll
    for (int o = 0; o < RGB_MAXO; ++o) wv[o] = (p.dx && o < p.O) ? p.w[c * p.ldw + o] * sc : 0.f;
    const size_t base = ((size_t)b * p.C + c) * p.HW + pc0;
    float g[RGB_MAXO] = {0.f, 0.f, 0.f, 0.f};
    if (vec) {
      // all x loads of this channel's pixel chunk are issued first (8 float4 per lane in flight): one load per loop
      // iteration was a serial latency chain -- 45 us even for a 2 MB tensor
      constexpr int NIT = RGB_CHUNK / 256;
      float4 xv[NIT];
      if (p.G) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int px = lane * 4 + it * 256;
          xv[it] = px < npx ? *reinterpret_cast<const float4 *>(p.x + base + px) : make_float4(0, 0, 0, 0);
        }
      }
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int px = lane * 4 + it * 256;
        if (px < npx) {
          float4 d[RGB_MAXO];
#pragma unroll
          for (int o = 0; o < RGB_MAXO; ++o) d[o] = *reinterpret_cast<const float4 *>(&dys[o][px]);
          if (p.dx) {
            float4 r = make_float4(0, 0, 0, 0);
#pragma unroll
            for (int o = 0; o < RGB_MAXO; ++o) { r.x += wv[o] * d[o].x; r.y += wv[o] * d[o].y; r.z += wv[o] * d[o].z; r.w += wv[o] * d[o].w; }
            *reinterpret_cast<float4 *>(p.dx + base + px) = r;
          }
          if (p.G) {
#pragma unroll
            for (int o = 0; o < RGB_MAXO; ++o)
              g[o] += xv[it].x * d[o].x + xv[it].y * d[o].y + xv[it].z * d[o].z + xv[it].w * d[o].w;
          }
        }
      }
    } else {
      for (int px = lane; px < npx; px += 64) {
        if (p.dx) {
          float r = 0.f;
          for (int o = 0; o < RGB_MAXO; ++o) r += wv[o] * dys[o][px];
          p.dx[base + px] = r;
        }
        if (p.G) {
          const float xv = p.x[base + px];
          for (int o = 0; o < RGB_MAXO; ++o) g[o] += xv * dys[o][px];
        }
      }
    }
    if (p.G) {  // the RGB_MAXO wave sums together (common.h wave_tree_sum): lane 16 o ends up with the total of g[o]
      static_assert(RGB_MAXO == 4, "four partial sums per channel");
      wave_tree_sum<4, 4, 32>(g, lane);
      const int o = wave_tree_row<4>(lane);
      if ((lane & 15) == 0 && o < p.O) p.G[(((size_t)b * p.C + c) * gridDim.x + blockIdx.x) * p.O + o] = g[0];
    }
  }
}

static int rgb_mask_ok(const float *colmask, int maskW, int maskCW, int HW) {
  return !colmask || (maskW >= 1 && maskCW >= 1 && HW % maskW == 0);
}

extern "C" int tbg_rgb_project_f32(const float *x, const float *w, const float *scale, const float *bias,
                                   const float *skip, float *y, int B, int C, int O, int ldw, int HW, float alpha,
                                   float bias_mul, const float *colmask, int maskW, int maskCW, void *stream) {
  if (!x || !w || !y || B < 1 || C < 1 || O < 1 || O > RGB_MAXO || ldw < O || HW < 1) return TBG_EINVAL;
  if (!rgb_mask_ok(colmask, maskW, maskCW, HW)) return TBG_EINVAL;
  if ((double)B * C * HW > 2147483647.0) return TBG_ERANGE;
  const size_t lds = ((((size_t)C * RGB_MAXO + 3) & ~(size_t)3) + 3 * 64 * RGB_MAXO * 4) * sizeof(float);
  if (lds > 64 * 1024) return TBG_EUNSUPPORTED;
  RgbP p{};
  p.x = x; p.w = w; p.scale = scale; p.bias = bias; p.skip = skip; p.y = y;
  p.B = B; p.C = C; p.O = O; p.ldw = ldw; p.HW = HW; p.alpha = alpha; p.bias_mul = bias_mul;
  p.colmask = colmask; p.maskW = colmask ? maskW : 1; p.maskCW = colmask ? maskCW : 1;
  p.maskN = colmask ? (maskW + maskCW - 1) / maskCW : 1;
  dim3 grid((((HW + 3) / 4) + 63) / 64, B);
  hipLaunchKernelGGL(rgb_project_kernel, grid, dim3(256), lds, tbg_stream(stream), p);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}

extern "C" int tbg_rgb_backproject_chunks(int HW) { return HW < 1 ? -1 : (HW + RGB_CHUNK - 1) / RGB_CHUNK; }

extern "C" int tbg_rgb_backproject_f32(const float *x, const float *dy, const float *w, const float *scale, float *dx,
                                       float *G, int B, int C, int O, int ldw, int HW, float alpha, const float *colmask,
                                       int maskW, int maskCW, float *dym, float *dysum, void *stream) {
  if (!dy || B < 1 || C < 1 || O < 1 || O > RGB_MAXO || HW < 1 || (!dx && !G)) return TBG_EINVAL;
  if ((dx && (!w || ldw < O)) || (G && !x)) return TBG_EINVAL;
  if (!rgb_mask_ok(colmask, maskW, maskCW, HW)) return TBG_EINVAL;
  if ((double)B * C * HW > 2147483647.0) return TBG_ERANGE;
  RgbP p{};
  p.x = x; p.dy = dy; p.w = w; p.scale = scale; p.dx = dx; p.G = G; p.dym = dym; p.dysum = dysum;
  p.B = B; p.C = C; p.O = O; p.ldw = ldw; p.HW = HW; p.alpha = alpha;
  p.colmask = colmask; p.maskW = colmask ? maskW : 1; p.maskCW = colmask ? maskCW : 1;
  p.maskN = colmask ? (maskW + maskCW - 1) / maskCW : 1;
  dim3 grid((HW + RGB_CHUNK - 1) / RGB_CHUNK, (C + RGB_CPB - 1) / RGB_CPB, B);
  hipLaunchKernelGGL(rgb_backproject_kernel, grid, dim3(256), 0, tbg_stream(stream), p);
  TBG_LAUNCH_CHECK();
  return TBG_OK;
}
