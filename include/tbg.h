/*
 * tbg.h -- C ABI of libtbg_hip.so: the MI355X (gfx950) kernels behind the TextBoxGAN
 * training step.
 *
 * Boundary contract (mirrors what the reference's only native plugin does; reference
 * models/custom_stylegan2/layers/upfirdn/upfirdn_2d.cu:232-324 and the loader
 * custom_ops.py:109-213):
 *   - every entry point returns 0 (TBG_OK) or a negative error code; nothing throws, aborts,
 *     synchronises the device, allocates or frees device memory.  The caller's framework
 *     allocator provides outputs/workspaces (the role of ctx->allocate_output, .cu:258-265).
 *   - all pointers are DEVICE pointers to contiguous fp32 unless noted; kernels are enqueued
 *     asynchronously on `stream` (a hipStream_t passed as void*; NULL = default stream), the
 *     analogue of launching on TF's stream (.cu:234-235,305-306).
 *   - re-entrant; no mutable global state, no environment variables: what runs is a pure function of the arguments.
 *   - shape/argument errors -> TBG_EINVAL (OP_REQUIRES InvalidArgument, .cu:228-256),
 *     more than INT32_MAX elements -> TBG_ERANGE (.cu:243-244,266), launch failure -> TBG_EHIP
 *     (.cu:20,306).
 *
 * Layouts: activations NCHW; convolution filters in "GEMM layout" [KH*KW][C][ldw]
 * (tap-major, reduction channel, output channel contiguous) -- the reference's own HWIO
 * parameter layout [k,k,I,O] IS this layout with ldw = O, so forward passes read the
 * checkpoint-layout weights directly.
 */
#ifndef TBG_H_
#define TBG_H_

#ifdef __cplusplus
extern "C" {
#endif

#define TBG_OK 0
#define TBG_EINVAL (-1)
#define TBG_ERANGE (-2)
#define TBG_EHIP (-3)
#define TBG_EUNSUPPORTED (-4)

#define TBG_ACT_LINEAR 0
#define TBG_ACT_LRELU 1

/* library version (major*10000 + minor*100 + patch) and error strings */
int tbg_version(void);
const char *tbg_strerror(int code);

/* HOST helper (no device work, `data` is a host pointer): CRC-32C (Castagnoli) of data[0..n) continuing from `crc`
 * (0 to start) -- the checksum of TensorFlow's TensorBundle checkpoint files (tensorflow/core/lib/hash/crc32c.h), used
 * by textboxgan_amd/tf_checkpoint.py to read/write the reference's checkpoint layout (models/model_loader.py:57-81). */
unsigned int tbg_crc32c(const void *data, long long n, unsigned int crc);

/* Fused epilogue shared by the conv / FIR / bias_act kernels.  For an accumulator value
 * `acc` at (sample b, channel m, pixel p):
 *   v   = acc * alpha * (out_scale ? out_scale[b*M + m] : 1)
 *       + (noise ? noise[b*HW + p] * strength[0] : 0) + (bias ? bias[m] * bias_mul : 0)
 *   v   = (act == LRELU ? (v > 0 ? v : v * slope) : v) * gain
 *   out = residual ? (v + residual[...]) * res_scale : v
 * (res_first != 0: the residual is added BEFORE the activation instead -- ResNet units of the OCR.)
 *   out = gate ? (gate[...] > 0 ? out : 0) : out      (applied last; conv / slab / bias_act launches, not with dot_aux:
 *   the ReLU backward of a frozen ResNet unit rides on the data-gradient launch that produces its incoming gradient)
 * tbg_conv2d_f32 additionally supports a fused per-(b,m) dot product of the UNSCALED accumulator
 * with a second tensor (dot_aux/dot_out): the style gradient ds[b,i] = sum_p x[b,i,p]*dxhat[b,i,p]
 * comes out of the same launch that writes dx = s*dxhat.
 * Replaces reference layers/noise.py:12-22 + layers/bias_act.py:25-34 (+ the demodulation
 * scale of modulated_conv2d.py:119-121 and the resnet add of discriminator.py:82). */
typedef struct tbg_epilogue {
  const float *out_scale; /* [B*M] or NULL */
  const float *bias;      /* [M] or NULL */
  const float *noise;     /* [B, H*W] or NULL */
  const float *strength;  /* device scalar, required with noise */
  const float *residual;  /* output-shaped or NULL */
  const float *dot_aux;   /* conv only: output-shaped tensor or NULL */
  float *dot_out;         /* conv only: [B*M*slots] partial sums of (acc*alpha) * dot_aux[b,m,p] over pixel ranges, every
                           * element written by plain stores (no atomics): the caller sums the slots of each (b,m) in a
                           * fixed order; slots = tbg_conv2d_dot_slots(desc, ...) */
  const float *gate;      /* output-shaped or NULL */
  float alpha;
  float bias_mul;
  float slope;
  float gain;
  float res_scale;
  int act;
  int res_first;
  /* UNIT SINK (round 5; see "UNIT TENSORS" below): the launch ALSO writes its result as the unit tensor the next
   * convolution consumes -- units_out = units(out * units_scale[b*M + m]) over the launch's [B, M, Hout, Wout] output, ring of
   * zero units included (every unit of the tensor is written by the launch; the bits are those of tbg_units_pack_f32(out,
   * units_scale)).  Served by tbg_conv2d_{bf16,x3} (ksplit == 1; the exact-fp32 entry: TBG_EUNSUPPORTED), tbg_conv2d_units, tbg_conv2d_units_s2 and
   * tbg_upfirdn2d_sep_f32 (up = down = 1); M % 8 == 0.  With a sink the fp32 output pointer of those entries may be NULL:
   * the activation then exists as a unit tensor only.  Other entries return TBG_EINVAL when units_out is set. */
  void *units_out;            /* bf16 U[units_planes][B][M/8][Hout+2][Wout+2][8], 16-byte aligned, or NULL */
  const float *units_scale;   /* [B*M] (the NEXT layer's style modulation, modulated_conv2d.py:94-96) or NULL = 1 */
  int units_planes;           /* 1 (bf16: RNE) | 3 (f32x3: hi | mid | lo) */
} tbg_epilogue;

/* ------------------------------------------------------------------------------------------
 * upfirdn2d -- replaces the TF op "UpFirDn2D" (upfirdn_2d.cu:310-324; kernels :64-207).
 * x: [major, inH, inW, minor], k: [kH, kW] (applied FLIPPED, as the op does), y: [major, outH,
 * outW, minor] with out = (in*up + pad0 + pad1 - k + down) / down.  The gradient is the same
 * op with transformed parameters (upfirdn_2d_v2.py:204-244), so no separate backward entry.
 * _ex adds an optional per-`major` input scale and the fused epilogue (minor must be 1;
 * channel m = major % M, sample b = major / M).
 * _sep is _ex for a SEPARABLE filter k = ky (x) kx given by its 1-D factors kx[kW], ky[kH] -- the
 * form the model uses (_setup_kernel, upfirdn_2d_v2.py:18-25, builds k as an outer product): a
 * horizontal then a vertical pass, 4+4 instead of 16 MACs per output for the [1,3,3,1] blur.
 * Any filter size / factors / minor are accepted (filters <= 4x4 with factors <= 2 and minor == 1 take
 * the register-tiled kernel, everything else a one-lane-per-output kernel).
 * ---------------------------------------------------------------------------------------- */
int tbg_upfirdn2d_f32(const float *x, const float *k, float *y, int major, int inH, int inW,
                      int minor, int kH, int kW, int upx, int upy, int downx, int downy,
                      int padx0, int padx1, int pady0, int pady1, void *stream);

int tbg_upfirdn2d_ex_f32(const float *x, const float *k, float *y, int major, int inH, int inW,
                         int kH, int kW, int upx, int upy, int downx, int downy, int padx0,
                         int padx1, int pady0, int pady1, const float *in_scale, int M,
                         const tbg_epilogue *epi, void *stream);

int tbg_upfirdn2d_sep_f32(const float *x, const float *kx, const float *ky, float *y, int major,
                          int inH, int inW, int kH, int kW, int upx, int upy, int downx, int downy,
                          int padx0, int padx1, int pady0, int pady1, const float *in_scale, int M,
                          const tbg_epilogue *epi, void *stream);

/* The op's second registered type (upfirdn_2d.cu:323-324: `half`): x [major,inH,inW,minor], k [kH,kW] and y are IEEE fp16,
 * the accumulation is fp32 (.cu:101,195), the result is rounded to fp16 once.  Same attributes, output size rule and error
 * codes as tbg_upfirdn2d_f32. */
int tbg_upfirdn2d_f16(const void *x, const void *k, void *y, int major, int inH, int inW, int minor, int kH, int kW,
                      int upx, int upy, int downx, int downy, int padx0, int padx1, int pady0, int pady1, void *stream);

/* Kernel instantiation (rocprofv3 spelling) that the upfirdn2d entries select for a geometry (sep = 1: the separable
 * entry): a pure function of its arguments.  Tests use it to prove that every instantiation a training step launches is
 * compared with the oracle. */
int tbg_upfirdn2d_kernel_name(int minor, int kH, int kW, int upx, int upy, int downx, int downy, int padx0,
                              int pady0, int sep, char *buf, int n);

/* ------------------------------------------------------------------------------------------
 * Convolution as fp32-MFMA implicit GEMM (v_mfma_f32_32x32x2_f32).  One descriptor serves:
 *   transposed = 0 : y[b,m,oy,ox] = sum_{t=(kh,kw),c} x[b,c,oy*sy-py+kh,ox*sx-px+kw] * W[t'][c][m]
 *   transposed = 1 : y[b,m,sy*a+kh,sx*b'+kw] += x[b,c,a,b'] * W[t'][c][m]      (VALID, py=px=0)
 * with t' = flip ? KH*KW-1-t : t.  Replaces tf.nn.conv2d / conv2d_transpose as used by
 * layers/conv.py:51-73, layers/modulated_conv2d.py:85-121, upfirdn_2d_v2.py:65-113 and
 * their gradients.  in_scale [B*C] (style modulation, modulated_conv2d.py:94-96) is applied
 * while staging x; the epilogue applies demodulation / noise / bias / activation.
 * ksplit > 1 splits the reduction: y must then hold ksplit SLABS [ksplit][B,M,Hout,Wout]; split k
 * stores alpha*acc of its channel range into slab k with plain stores (every slab is fully written; no
 * zero-fill, no atomics; the epilogue must be alpha-only) and tbg_slab_epilogue_f32 sums the slabs
 * and applies the real epilogue.
 * `w` is the PACKED filter Wp[tap][ceil(C/4)][ldw][4] written by tbg_weight_pack_f32 (ldw = its M;
 * 16-byte aligned): 4 consecutive reduction channels of one output channel per 16-byte unit, which is
 * both the LDS-DMA granule and one ds_read_b128 MFMA operand.
 * ---------------------------------------------------------------------------------------- */
typedef struct tbg_conv_desc {
  int B, C, M;
  int Hin, Win, Hout, Wout;
  int KH, KW;
  int sy, sx;
  int py, px;
  int transposed;
  int flip;
  int ldw;
  int ksplit;
} tbg_conv_desc;

int tbg_conv2d_f32(const tbg_conv_desc *d, const float *x, const float *w, float *y,
                   const float *in_scale, const tbg_epilogue *epi, void *stream);

/* Kernel instantiation (rocprofv3 spelling) that tbg_conv2d_f32 selects for a descriptor (has_in_scale: whether an
 * in_scale pointer will be passed -- it enters the LDS budget).  A pure function of its arguments: the dispatch has no
 * environment knobs and the library keeps no mutable state.  Profiling aid (bench.py attributes HIP-event timings to
 * rocprofv3 kernel names with it; tests use it to prove every instantiation is compared with the oracle). */
int tbg_conv2d_kernel_name(const tbg_conv_desc *d, int has_in_scale, char *buf, int n);
/* Slots per (sample, channel) of the fused dot product's partial sums for this descriptor (mode: 0 = fp32, 1 = bf16, 2 = f32x3):
 * one per (pixel tile of the image, wave column of the tile).  0: the fused dot is not available for this launch (several
 * images per tile, several output-parity classes, or K split) -- a launch that asks for it anyway returns TBG_EUNSUPPORTED and
 * the caller reduces the finished output instead.  Negative: an error code. */
int tbg_conv2d_dot_slots(const tbg_conv_desc *d, int has_in_scale, int mode);
/* Thread blocks the launch of this descriptor consists of (pixel tiles x channel tiles x output-parity classes x d->ksplit;
 * mode as above) -- the tile choice belongs to the library, so a caller that sizes a K split (d->ksplit) asks for the
 * count at ksplit = 1 instead of mirroring the choice.  Negative: an error code. */
int tbg_conv2d_blocks(const tbg_conv_desc *d, int has_in_scale, int mode);
/* tbg_conv2d_f32 with an EXPLICIT instantiation family for 3x3 / 1x1 non-transposed-class launches (tuning and test aid):
 * variant 0 = the library's choice, 1 = software-pipelined (double-buffered LDS), 2 = plain 8-channel chunks,
 * 3 = 4-channel chunks at 4 waves/SIMD (128x128 tile only), 4 / 5 = stride-2 transposed 3x3 as one block per output-parity
 * class / as the merged-class kernel (all four classes from one staged halo tile), 6 = the merged-class kernel with
 * 16-channel chunks (fp32 only), 7 = the register-prefetch K loop (128x128 tile: the halo tile of chunk k+1 is loaded
 * while chunk k's MFMAs run; the library's choice for stride-1 layers), 8 / 9 = the per-class form with the tile height halved once / twice (variant 4 keeps the
 * full-height tiles; 0 picks the height by padded rows x halo overhead).  TBG_EUNSUPPORTED if the descriptor cannot take it. */
int tbg_conv2d_f32_variant(const tbg_conv_desc *d, const float *x, const float *w, float *y,
                           const float *in_scale, const tbg_epilogue *epi, int variant, void *stream);

/* Weight gradient:  dW[t*st_t + cl*st_l + cs*st_s] = alpha * sum_{b,u,v}
 *     S[b,cs,u,v]*s_scale[b,cs] * L[b,cl,u*sy-py+kh,v*sx-px+kw]*l_scale[b,cl]
 * S is the tensor on the (small) output grid, L the tensor on the input grid.  Every element of dW
 * in range is OVERWRITTEN.  The pixel reduction is split over blocks that write partial tiles to
 * `workspace` (caller-provided, tbg_conv2d_wgrad_workspace_bytes(d) bytes), which a second kernel
 * sums -- no atomics, deterministic.  Gradient of both forms of tbg_conv2d_f32. */
typedef struct tbg_wgrad_desc {
  int B, CS, CL;
  int Hs, Ws, Hl, Wl;
  int KH, KW;
  int sy, sx;
  int py, px;
  int st_t, st_l, st_s;
  float alpha;
  /* optional rider (round 5): the BIAS gradient of the same layer, db[cs] = sum_{b < bias_B, k < bias_nch} bias_parts[(b*CS + cs)
   * * bias_nch + k] -- the per-(sample, channel, chunk) partial sums tbg_bias_act_bwd_* left -- formed by one block of the launch
   * that sums the filter gradient's partial tiles, instead of by a reduction launch of its own.  bias_grad NULL = off. */
  const float *bias_parts;
  float *bias_grad; /* [CS] */
  int bias_B, bias_nch;
} tbg_wgrad_desc;

long long tbg_conv2d_wgrad_workspace_bytes(const tbg_wgrad_desc *d);
int tbg_conv2d_wgrad_kernel_name(const tbg_wgrad_desc *d, char *buf, int n);
int tbg_conv2d_wgrad_f32(const tbg_wgrad_desc *d, const float *S, const float *L, float *dW,
                         const float *s_scale, const float *l_scale, float *workspace,
                         long long workspace_bytes, void *stream);
/* _ex: dW[idx] = alpha*sum + gamma * addw[idx] * addq[cl*st_l + cs*st_s]  (addw laid out like dW, addq one
 * [CL x CS] plane).  Folds the demodulation term of the filter gradient (modulated_conv2d.py:78-82:
 * d(dem)/dw = 2 coef^2 w * dwsq[i,o]) into the write of the convolution's filter gradient. */
int tbg_conv2d_wgrad_ex_f32(const tbg_wgrad_desc *d, const float *S, const float *L, float *dW,
                            const float *s_scale, const float *l_scale, const float *addw,
                            const float *addq, float gamma, float *workspace,
                            long long workspace_bytes, void *stream);

/* Pack the HWIO parameter src[T][I][O] (the layout of the reference's conv kernels, layers/conv.py:30-41,
 * modulated_conv2d.py:33-45) into the filter format of tbg_conv2d_f32:
 *   transpose = 0 : C = I, M = O   (forward correlation)
 *   transpose = 1 : C = O, M = I   (data gradient: the transposed filter)
 * dst[t'][c/4][m][c%4] with t' = flip ? T-1-t : t, zero padded past C; dst holds
 * tbg_weight_pack_floats(T, I, O, transpose) floats and must be 16-byte aligned. */
long long tbg_weight_pack_floats(int T, int I, int O, int transpose);
int tbg_weight_pack_f32(const float *src, float *dst, int T, int I, int O, int transpose, int flip,
                        void *stream);

/* ------------------------------------------------------------------------------------------
 * bf16-in / fp32-accumulate forms (BASELINE configs[2]; the reference op itself registers a 16-bit type beside
 * float, upfirdn_2d.cu:323-324).  Activations, gradients, epilogue operands and outputs stay fp32 in HBM; the kernels
 * round the staged operands to bf16 (round-to-nearest-even) on their way into LDS and contract them on
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Master weights stay fp32: tbg_weight_pack_bf16 writes the bf16
 * filter Wp[T][ceil(C/8)][M][8] (tbg_weight_pack_bf16_bytes bytes, 16-byte aligned) that tbg_conv2d_bf16 consumes.
 * Descriptors, epilogues, split-K slabs, workspaces and error codes are those of the fp32 entries.  A result equals the
 * fp32 entry's result on operands pre-rounded to bf16, up to fp32 summation order.
 * ---------------------------------------------------------------------------------------- */
long long tbg_weight_pack_bf16_bytes(int T, int I, int O, int transpose);
int tbg_weight_pack_bf16(const float *src, void *dst, int T, int I, int O, int transpose, int flip,
                         void *stream);
int tbg_conv2d_bf16(const tbg_conv_desc *d, const float *x, const void *w, float *y,
                    const float *in_scale, const tbg_epilogue *epi, void *stream);
int tbg_conv2d_bf16_kernel_name(const tbg_conv_desc *d, int has_in_scale, char *buf, int n);
/* explicit instantiation family (tuning / test aid): 0 = library's choice, 1 = 128x256 tile, 2 = 32-channel chunks,
 * 3 = the register-prefetch K loop (128x128 tile; the library's choice for stride-1 layers), 4 / 5 = class-per-block /
 * merged-class transposed form (both with the plain K loop). */
int tbg_conv2d_bf16_variant(const tbg_conv_desc *d, const float *x, const void *w, float *y,
                            const float *in_scale, const tbg_epilogue *epi, int variant, void *stream);
/* filter gradient; tile rows narrower than 8 pixels (Ws <= 4) fall back to the exact fp32 kernel.  Workspace size =
 * tbg_conv2d_wgrad_workspace_bytes(d). */
int tbg_conv2d_wgrad_bf16(const tbg_wgrad_desc *d, const float *S, const float *L, float *dW,
                          const float *s_scale, const float *l_scale, const float *addw,
                          const float *addq, float gamma, float *workspace,
                          long long workspace_bytes, void *stream);
int tbg_conv2d_wgrad_bf16_kernel_name(const tbg_wgrad_desc *d, char *buf, int n);

/* ------------------------------------------------------------------------------------------
 * "f32x3" forms: fp32 results from the bf16 matrix pipe by exact operand splitting.  Every fp32 operand is written as
 * the sum of three bf16 terms hi + mid + lo (3 x 8 significand bits = fp32's 24); products of bf16 values are exact in
 * fp32, and the six largest of the nine partial products (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi) are accumulated
 * in fp32 by v_mfma_f32_32x32x16_bf16 -- the dropped terms are <= 2^-24 relative, i.e. the result carries fp32-grade
 * error (measured against float64 it is slightly BETTER than the k-ordered fmaf chain of v_mfma_f32_32x32x2_f32) at 96
 * instead of 256 matrix cycles per 8 channels.  Same tensors (fp32 in HBM), descriptors, epilogues, split-K slabs and error
 * codes as tbg_conv2d_f32; the filter is packed by tbg_weight_pack_x3 as three bf16 planes Wp[3][T][ceil(C/8)][M][8]
 * (tbg_weight_pack_x3_bytes bytes), the activations are split while they are staged into LDS.  Non-finite operands give
 * NaN where fp32 arithmetic would give Inf (Inf - Inf in the split).
 * ---------------------------------------------------------------------------------------- */
long long tbg_weight_pack_x3_bytes(int T, int I, int O, int transpose);
int tbg_weight_pack_x3(const float *src, void *dst, int T, int I, int O, int transpose, int flip, void *stream);
int tbg_conv2d_x3(const tbg_conv_desc *d, const float *x, const void *w, float *y, const float *in_scale,
                  const tbg_epilogue *epi, void *stream);
int tbg_conv2d_x3_kernel_name(const tbg_conv_desc *d, int has_in_scale, char *buf, int n);
/* filter gradient in f32x3 arithmetic: the float4-staged geometries (stride-1 3x3 with 16-byte-aligned rows, stride-2 VALID
 * 3x3 with Ws % 32 == 0 -- the layers that hold the FLOPs) run conv_wgrad_x3_kernel (both operands split into three bf16
 * terms while staged, six products per tap, fp32 accumulate); every other geometry runs the exact fp32 kernel of
 * tbg_conv2d_wgrad_ex_f32.  Arguments, workspace (tbg_conv2d_wgrad_workspace_bytes) and error codes as that entry. */
int tbg_conv2d_wgrad_x3(const tbg_wgrad_desc *d, const float *S, const float *L, float *dW,
                        const float *s_scale, const float *l_scale, const float *addw,
                        const float *addq, float gamma, float *workspace,
                        long long workspace_bytes, void *stream);
int tbg_conv2d_wgrad_x3_kernel_name(const tbg_wgrad_desc *d, char *buf, int n);
/* explicit form of the stride-2 transposed 3x3 launches (tuning / test aid): 0 = library's choice, 4 / 5 = class-per-block /
 * merged-class; 1 / 2 = force / forbid the 128x256 tile. */
int tbg_conv2d_x3_variant(const tbg_conv_desc *d, const float *x, const void *w, float *y, const float *in_scale,
                          const tbg_epilogue *epi, int variant, void *stream);

/* ------------------------------------------------------------------------------------------
 * UNIT TENSORS: convolution operands written ONCE in the form the matrix-core kernels consume, so that forward, data-gradient
 * and filter-gradient launches stage their tiles by LDS-DMA alone (no per-launch re-scaling / re-splitting / re-rounding on
 * the VALU).  The unit tensor of x[B][C][H][W] is
 *     U[plane][b][c / 8][1 + y][1 + x][c % 8]     bf16, 16-byte aligned, tbg_units_bytes(B, C, H, W, planes) bytes
 * planes = 3: the f32x3 terms hi | mid | lo of x (exact split, see "f32x3 forms"); planes = 1: RNE(x) (bf16 mode).
 * One 16-byte unit = 8 consecutive channels of one pixel; every (b, c/8) plane is (H + 2) x (W + 2) units with a ring of ZERO
 * units (the padding of a 3x3 convolution is read from it), channels past C are zero.
 * tbg_units_pack_f32: U <- x * (scale ? scale[b*C + c] : 1)  (the style modulation of modulated_conv2d.py:94-96 -- both
 * the forward convolution and its filter gradient consume exactly this product); writes every unit incl. ring and tail.
 * ---------------------------------------------------------------------------------------- */
long long tbg_units_bytes(int B, int C, int H, int W, int planes);
int tbg_units_pack_f32(const float *x, const float *scale, void *U, int B, int C, int H, int W, int planes, void *stream);

/* tbg_bias_act_bwd_f32 as a PRODUCER of a unit tensor: U = units(dpre * alpha * out_scale[b,m]) of `planes` planes for the
 * [B, M, H, W] map (what the data-gradient and filter-gradient launches of the layer consume), dpre itself (NCHW fp32) only if
 * dpre_out != NULL, and the partial sums part_db / part_dn / part_dyy as [B, M, tbg_bias_act_bwd_units_chunks(H)] (the caller
 * reduces the last axis).  Same dpre / sum definitions as tbg_bias_act_bwd_f32. */
int tbg_bias_act_bwd_units_chunks(int H);
int tbg_bias_act_bwd_units(const float *dout, const float *out_act, void *U, int planes, float *dpre_out, float *part_db,
                           float *part_dn, float *part_dyy, int B, int M, int H, int W, const tbg_epilogue *epi,
                           void *stream);

/* Forward / data-gradient convolution from a unit tensor: tbg_conv2d_x3 (planes = 3) / tbg_conv2d_bf16 (planes = 1) with the
 * input given as the unit tensor XU of x * in_scale (so there is no in_scale argument) -- same descriptor, packed filter
 * (tbg_weight_pack_x3 / tbg_weight_pack_bf16), epilogue (fused dot included: tbg_conv2d_units_dot_slots slots per (b, m)) and
 * fp32 NCHW output.  Geometry: 3x3, stride 1, pad 1, not transposed, Hin % 8 == 0, Win % 32 == 0, M % 64 == 0, C % 8 == 0
 * (planes = 1: C % 16 == 0), ksplit == 1 -- TBG_EUNSUPPORTED otherwise (the caller keeps the NCHW entry for those). */
int tbg_conv2d_units_dot_slots(const tbg_conv_desc *d, int planes);
/* blocks of the launch and output channels per block (128, or 64 where 128-channel tiles would leave half the CUs without a
 * block): what a caller needs to decide between this entry and tbg_conv2d_x3 / _bf16 for a small layer */
int tbg_conv2d_units_blocks(const tbg_conv_desc *d, int planes);
int tbg_conv2d_units_tile_channels(const tbg_conv_desc *d, int planes);
int tbg_conv2d_units(const tbg_conv_desc *d, const void *XU, int planes, const void *w, float *y,
                     const tbg_epilogue *epi, void *stream);

/* SMALL MAPS (round 6; csrc/conv_small.hip): the same contract as tbg_conv2d_units -- input = the unit tensor XU of x * in_scale
 * with the geometry [B, C, Hin, Win], packed filter tbg_weight_pack_x3 / _bf16, full epilogue incl. the unit sink, fp32 NCHW output
 * (y may be NULL with a sink) -- for the layers whose GEMM is too small for one tile per block and K whole: the 1x25 ... 8x32 maps
 * of both networks (conv.py:51-73, discriminator.py:68-84, modulated_conv2d.py:98-112) and of the recogniser's trunk
 * (aster_inferer.py:28-37).  A block owns 32 output channels x 32 (or 64) pixels of the FLATTENED pixel list n = (b Hout + y) Wout + x
 * (tiles may span rows and samples: no padding waste on 25-wide maps), its 8 waves split K and sum their partial tiles through LDS
 * before the epilogue: ONE launch, no slabs, no tbg_slab_epilogue_f32 pass, deterministic (fixed summation order).
 * Geometries (TBG_EUNSUPPORTED otherwise; ksplit must be 1):
 *   3x3, stride 1, pad 1, not transposed, Wout >= 3 ("row units": one load of the input serves the three taps of a filter row);
 *   k x k (k <= 3; KH and KW independent), pad 0, stride (1|2, 1|2), not transposed: Hout = (Hin - KH) / sy + 1 -- the 1x1 layers
 *   and the strided VALID convolutions behind a blur (upfirdn_2d_v2.py:106-113) ("tap list": a row unit is one tap);
 *   1x1 transposed with stride (sy, sx) (the data gradient of the strided 1x1 form: y[b, m, sy a, sx b'] = sum_c x[b, c, a, b'] W[c][m],
 *   every other output pixel = epilogue(0)), Hout >= (Hin - 1) sy + 1;
 *   k x k (k = 2, 3) transposed with stride (1|2, 1|2), pad 0, Hout >= (Hin - 1) sy + KH (upsample_conv_2d, :65-103, and the data
 *   gradients of the strided layers): the sy sx output-parity classes as block ranges of ONE launch; no unit sink, no fused dot.
 * Any M (channel tiles are clamped; a sink needs M % 8 == 0), any C with planes = 3, ceil(C/8) even with planes = 1.
 * Fused dot: tbg_conv2d_units_small_dot_slots slots per (b, m) = Hout Wout / 32 when a pixel tile never straddles two samples,
 * 0 (dot not served: TBG_EUNSUPPORTED) otherwise.  tbg_conv2d_units_small_blocks: blocks of the launch (for the caller's dispatch);
 * tbg_conv2d_units_small_tile_pixels: 32 | 64 pixels per block. */
int tbg_conv2d_units_small_tile_pixels(const tbg_conv_desc *d, int planes);
int tbg_conv2d_units_small_blocks(const tbg_conv_desc *d, int planes);
int tbg_conv2d_units_small_dot_slots(const tbg_conv_desc *d, int planes);
int tbg_conv2d_units_small(const tbg_conv_desc *d, const void *XU, int planes, const void *w, float *y,
                           const tbg_epilogue *epi, void *stream);

/* PHASE unit tensors and the stride-2 convolution that reads them (csrc/conv_units_s2.hip).  The input t [B, C, Hin, Win] of a 3x3 /
 * stride-2 / pad-0 convolution with Ho x Wo outputs (conv_downsample_2d's strided convolution after its blur,
 * upfirdn_2d_v2.py:106-113; the data gradient of upsample_conv_2d's transposed convolution, :65-103), de-interleaved by parity:
 *     P[plane][b][c / 8][2 (Y & 1) + (X & 1)][Y >> 1][X >> 1][c % 8] = t[b][c][Y][X]      (Ho + 1) x (Wo + 1) units per phase,
 * zero where t has no element, planes as above.  tbg_units_pack_s2_f32 is the stand-alone producer (scale as tbg_units_pack_f32).
 * tbg_conv2d_units_s2: tbg_conv2d_x3 / _bf16 on such a tensor -- same descriptor (Hin, Win = the dimensions of t), packed filter
 * and epilogue; geometry 3x3, stride 2, pad 0, Hout % 8 == 0, Wout % 32 == 0, M % 64 == 0, C % 8 == 0 (planes = 1: C % 16 == 0),
 * ksplit == 1 -- TBG_EUNSUPPORTED otherwise. */
long long tbg_units_s2_bytes(int B, int C, int Ho, int Wo, int planes);
/* fused producer: t = upfirdn2d(x, kx (x) ky, up = down = 1, pad) * in_scale[b*C + c] written as the PHASE unit tensor of t (the
 * blur in front of conv_downsample_2d's strided convolution; the blur's adjoint in upsample_conv_2d's backward pass):
 * tbg_units_pack_s2_f32 of tbg_upfirdn2d_sep_f32's output (to single fp32 roundings), without the fp32 tensor in between.  Filters of <= 4 x 4 taps. */
int tbg_upfirdn2d_units_s2_f32(const float *x, const float *kx, const float *ky, void *U, int B, int C, int inH, int inW, int kH,
                               int kW, int padx0, int padx1, int pady0, int pady1, const float *in_scale, int planes, void *stream);
int tbg_units_pack_s2_f32(const float *x, const float *scale, void *U, int B, int C, int Hin, int Win, int Ho, int Wo, int planes,
                          void *stream);
int tbg_conv2d_units_s2_blocks(const tbg_conv_desc *d, int planes);
int tbg_conv2d_units_s2_tile_channels(const tbg_conv_desc *d, int planes);
int tbg_conv2d_units_s2_dot_slots(const tbg_conv_desc *d, int planes);
int tbg_conv2d_units_s2(const tbg_conv_desc *d, const void *XU, int planes, const void *w, float *y,
                        const tbg_epilogue *epi, void *stream);

/* The stride-2 TRANSPOSED 3x3 convolution (upsample_conv_2d's transposed convolution, upfirdn_2d_v2.py:65-103; the data gradient
 * of the strided convolution above) from the stride-1 unit tensor XU of its input [B, C, Hin, Win] (scale inside): the transposed
 * form of tbg_conv2d_x3 / _bf16 -- same descriptor (transposed = 1, stride 2, pad 0, Hout in {2 Hin + 1, 2 Hin + 2}, likewise
 * Wout), packed filter and flip; y = alpha * result, fp32 NCHW (store-only).  M % 64 == 0, C % 8 == 0 (planes = 1: C % 16 == 0),
 * ksplit == 1 -- TBG_EUNSUPPORTED otherwise. */
int tbg_conv2d_units_t2_blocks(const tbg_conv_desc *d, int planes);
int tbg_conv2d_units_t2(const tbg_conv_desc *d, const void *XU, int planes, const void *w, float *y, float alpha, void *stream);

/* Filter gradient of that stride-2 convolution from unit tensors: S (the tensor on the Ho x Wo output grid) as a stride-1 unit
 * tensor, L (the input-grid tensor t) as a PHASE unit tensor; result, strides, alpha and the additive term as
 * tbg_conv2d_wgrad_units.  Geometry: 3x3, stride 2, pad 0, Ws % 32 == 0, CS % 128 == 0, CL % 64 == 0 -- TBG_EUNSUPPORTED otherwise. */
long long tbg_conv2d_wgrad_units_s2_workspace_bytes(const tbg_wgrad_desc *d);
int tbg_conv2d_wgrad_units_s2(const tbg_wgrad_desc *d, const void *SU, const void *LU, int planes, float *dW, const float *addw,
                              const float *addq, float gamma, float *workspace, long long workspace_bytes, void *stream);

/* Filter gradient from unit tensors: tbg_conv2d_wgrad_ex_f32's result (same descriptor, dW strides, alpha, addw / addq /
 * gamma) with S and L given as unit tensors SU (of [B,CS,Hs,Ws]) and LU (of [B,CL,Hl,Wl]) of `planes` planes each (their
 * scales already inside).  planes = 3: f32x3 arithmetic (six products per tap, fp32 accumulate); planes = 1: bf16 operands.
 * Geometry: 3x3, stride 1, pad 1, Ws % 32 == 0, Hs % 2 == 0, CS % 64 == 0, CL % 64 == 0 -- TBG_EUNSUPPORTED otherwise (the
 * caller keeps the NCHW entry for those).  workspace: tbg_conv2d_wgrad_units_workspace_bytes(d) bytes. */
long long tbg_conv2d_wgrad_units_workspace_bytes(const tbg_wgrad_desc *d);
int tbg_conv2d_wgrad_units(const tbg_wgrad_desc *d, const void *SU, const void *LU, int planes, float *dW,
                           const float *addw, const float *addq, float gamma, float *workspace,
                           long long workspace_bytes, void *stream);

/* Multi-tensor filter packing: items_dev is a DEVICE array of n_items descriptors; item k is packed exactly as
 * tbg_weight_pack_f32 (bf16 = 0) / tbg_weight_pack_bf16 (bf16 = 1) / tbg_weight_pack_x3 (bf16 = 2) would pack
 * (src, dst, T, I, O, transpose, flip).
 * One launch for all filters of a model (the training step refreshes its packed filters once per step). */
typedef struct tbg_pack_item {
  const float *src; /* HWIO parameter [T][I][O] */
  void *dst;        /* packed filter, 16-byte aligned */
  int T, I, O;
  int transpose, flip, bf16;
} tbg_pack_item;
int tbg_weight_pack_multi(const tbg_pack_item *items_dev, int n_items, void *stream);

/* ------------------------------------------------------------------------------------------
 * One time step of a frozen (bi)directional LSTM layer, pointwise part for all directions in one launch -- the
 * recurrent encoder of the OCR branch (aster_inferer.py:28-190 runs the ASTER SavedModel: 2x BiLSTM).  The step's
 * GEMMs (h @ Whh^T forward, dgates @ Whh backward) stay library GEMMs batched over the directions.
 * Direction d works on time t = d == 0 ? s : T-1-s.  Gate order i,f,g,o (PyTorch / cuDNN).
 *   gx, dg [D][T][B][4H] time-major input projections (+biases) / their gradients;  hw [D][B][4H] (NULL at s = 0)
 *   act [D][S][B][4H], cs [D][S][B][H] saved activations / cell states;  h [D][B][H];  seq, dseq [B][T][D*H]
 * bwd: `first` = 1 for the first step processed: dc is initialised.  Optional (NULL) operands: hw (gates already complete),
 * seq, dseq (then dh = dh_rec), dh_rec, dg -- the attention decoder's LSTM cell uses the same kernels with D = 1.
 * ---------------------------------------------------------------------------------------- */
int tbg_lstm_step_fwd_f32(const float *gx, const float *hw, float *act, float *cs, float *h, float *seq,
                          int D, int T, int B, int H, int s, void *stream);
int tbg_lstm_step_bwd_f32(const float *dseq, const float *dh_rec, float *dc, const float *act,
                          const float *cs, float *dg, float *dgates, int D, int T, int B, int H, int s,
                          int first, void *stream);

/* FUSED recurrent steps (round 6): one launch per time step does the step's recurrent projection AND its pointwise half for both
 * directions (the library GEMM h @ Whh^T + tbg_lstm_step_* pair: two launches and two memory round trips per step).  A block owns 8
 * hidden units (all four gates) of one direction and up to 16 samples; the state travels transposed between launches in two
 * alternating buffers: hT [D][H][B] forward, dgT [D][4H][B] backward.  Same arithmetic, layouts of gx / act / cs / seq / dg / dc and
 * step order as tbg_lstm_step_*; H % 32 == 0 (TBG_EUNSUPPORTED otherwise: the caller keeps the two-launch form).
 * forward:  s = 0 .. T-1; hT_in is not read at s = 0.  backward: s = T-1 .. 0, first = 1 at s = T-1 (dgT_in, dc not read);
 * w_hhT [D][H][4H] is the transposed copy of w_hh [D][4H][H].  Replaces the per-sample SavedModel call of aster_inferer.py:28-37
 * (encoder BiLSTM, weigths_tf1_to_tf2.py:3-19) together with tbg_lstm_step_*. */
int tbg_lstm_fused_fwd_f32(const float *gx, const float *w_hh, const float *hT_in, float *hT_out, float *act, float *cs, float *seq,
                           int D, int T, int B, int H, int s, void *stream);
int tbg_lstm_fused_bwd_f32(const float *dseq, const float *w_hhT, const float *dgT_in, float *dgT_out, float *dc, const float *act,
                           const float *cs, float *dg, int D, int T, int B, int H, int s, int first, void *stream);

/* The attention decoder's step in TWO launches per direction of the pass (round 6; 8 forward / 6 backward launches of library GEMMs,
 * tbg_attn_ctx_*, tbg_lstm_step_* and index / argmax ops before).  All state between launches is TRANSPOSED ([rows][B]).
 *   tbg_lstm_cell_fused_fwd_f32: tbg_lstm_fused_fwd_f32 for ONE direction whose input is [context; hidden]: stateT_in [K][B], K = E + H,
 *     w [4H][K] = [W_ih(context columns) | W_hh], gx [steps][B][4H] = embedding row of the previous symbol + biases; the projection runs
 *     at every step (s = 0 included: the first context is not zero); hT_out [H][B] = rows E.. of the NEXT step's state buffer.
 *   tbg_dec_sample_fwd_f32 (one block per sample; after the cell of step s): logits[s] = b_o + h W_o^T (w_oT [H][C]), the greedy symbol
 *     = first maximum, gx_next = etab[symbol] ([C+1][4H]), q = b_d + h W_d^T (w_dT [H][H]), the attention of tbg_attn_ctx_fwd_f32 with
 *     that q: a_out [B][T], context -> ctxT_out [E][B] (rows 0..E of the next state buffer).  hT NULL: initial launch (h = 0, symbol =
 *     go, no logits); gx_next NULL: last step (logits only).
 *   tbg_rows_gemv_t_f32: outT [R][B] = sum_j stateT[j][b] w[r][j] (w [R][J]); backward: stateT = the step's gate gradients dgT [4H][B],
 *     w = [W_ih(context)^T; W_hh^T] ([E + H][4H]) -> rows 0..E = d(context)^T, rows E.. = the recurrent part of d(hidden)^T.
 *   tbg_dec_sample_bwd_f32 (one block per sample; after the rows launch of step s): tbg_attn_ctx_bwd_f32 with dctx = column b of
 *     dctxT (also stored as dctx_out [B][E]; denc_proj accumulated), then for step s - 1: dh = dlo [B][H] (= dlogits[s-1] W_o,
 *     precomputed) + dhpT column + dq W_d (w_d [H][H]), and tbg_lstm_step_bwd_f32's cell arithmetic with act / cs_cur / cs_prev of
 *     that step -> dgT_out [4H][B], dc [B][H].  dctxT NULL: first launch of the pass (cell part of the last step only, first = 1);
 *     act NULL: last launch (attention part of step 0 only).
 * Replaces the decoder behind aster_inferer.py:28-37 (Bahdanau attention + LSTM predictor, weigths_tf1_to_tf2.py:3-19). */
int tbg_lstm_cell_fused_fwd_f32(const float *gx, const float *w, const float *stateT_in, float *hT_out, float *act, float *cs,
                                int steps, int B, int H, int K, int s, void *stream);
int tbg_rows_gemv_t_f32(const float *stateT, const float *w, float *outT, int J, int R, int B, void *stream);
int tbg_dec_sample_fwd_f32(const float *hT, const float *w_oT, const float *b_o, const float *w_dT, const float *b_d,
                           const float *enc_proj, const float *enc, const float *v, const float *etab, float *logits, float *gx_next,
                           float *q_out, float *a_out, float *ctxT_out, int B, int T, int H, int E, int C, int go, void *stream);
int tbg_dec_sample_bwd_f32(const float *dctxT, const float *dhpT, const float *a, const float *q, const float *enc_proj,
                           const float *enc, const float *v, const float *w_d, float *denc_proj, float *dctx_out, const float *dlo,
                           const float *act, const float *cs_cur, const float *cs_prev, float *dc, float *dgT_out, int B, int T,
                           int H, int E, int first, void *stream);

/* Bahdanau attention context of the OCR decoder (frozen weights), one launch per decoder step:
 *   e[t] = sum_k v[k] tanh(enc_proj[b,t,k] + q[b,k]);  a = softmax_t(e) -> a [B][T];  ctx[b,:] = sum_t a[t] enc[b,t,:]
 * bwd: dq [B][H] written; denc_proj [B][T][H] and (if not NULL) denc [B][T][E] ACCUMULATED (+=).  T <= 64. */
int tbg_attn_ctx_fwd_f32(const float *q, const float *enc_proj, const float *enc, const float *v, float *ctx,
                         float *a, int B, int T, int H, int E, void *stream);
int tbg_attn_ctx_bwd_f32(const float *dctx, const float *a, const float *q, const float *enc_proj,
                         const float *enc, const float *v, float *dq, float *denc_proj, float *denc, int B,
                         int T, int H, int E, void *stream);

/* ------------------------------------------------------------------------------------------
 * Thin 1x1 convolutions at the RGB ends (one side has O <= 4 channels): HBM-bound streaming kernels.
 * project:      y[b,o,p] = (alpha * sum_c x[b,c,p] * w[c*ldw+o] * (scale ? scale[b*C+c] : 1)
 *                           + (bias ? bias[o]*bias_mul : 0) + (skip ? skip[b,o,p] : 0)) * m[b,p]
 *               = ToRGB.call (layers/to_rgb.py:28-33) incl. the `y = upsample(y) + torgb` add
 *               (synthesis_block.py:152-153) and, on the last block, mask_text_box (utils/utils.py:11-45) as the
 *               epilogue; also the image gradient of FromRGB.
 * backproject:  dym[b,o,p] = dy[b,o,p] * m[b,p]   (written out if dym != NULL: the gradient of the skip image / bias)
 *               dx[b,c,p] = alpha * (scale ? scale[b*C+c] : 1) * sum_o w[c*ldw+o] * dym[b,o,p]          (if dx)
 *               G[b,c,k,o] = sum_{p in pixel chunk k} x[b,c,p] * dym[b,o,p]                               (if G)
 *               = data gradient of ToRGB plus the channel Gram from which d(weight) and d(style) follow;
 *               with dx = NULL the filter gradient of FromRGB (layers/from_rgb.py:26-29).  G holds
 *               tbg_rgb_backproject_chunks(HW) partial sums per (b, c): plain stores, no atomics, no zero-fill --
 *               deterministic; the caller adds them up (tbg_torgb_bwd_smalls_f32 does).  dysum (optional) [B, chunks, O] =
 *               sum of dym over each pixel chunk: the partial sums of ToRGB's bias gradient.
 * Column mask m (NULL = 1): m[b,p] = colmask[b*ceil(maskW/maskCW) + (p % maskW) / maskCW] -- one value per maskCW-wide
 * column band of a row-major map of width maskW (HW % maskW == 0).
 * ---------------------------------------------------------------------------------------- */
int tbg_rgb_project_f32(const float *x, const float *w, const float *scale, const float *bias,
                        const float *skip, float *y, int B, int C, int O, int ldw, int HW, float alpha,
                        float bias_mul, const float *colmask, int maskW, int maskCW, void *stream);
int tbg_rgb_backproject_chunks(int HW);
int tbg_rgb_backproject_f32(const float *x, const float *dy, const float *w, const float *scale,
                            float *dx, float *G, int B, int C, int O, int ldw, int HW, float alpha,
                            const float *colmask, int maskW, int maskCW, float *dym, float *dysum, void *stream);

/* ------------------------------------------------------------------------------------------
 * bias_act: stand-alone epilogue (x: [B,M,HW]) and its backward.
 * backward, from dout and the SAVED OUTPUT `out` (sign(out) == sign(pre-activation)):
 *   dpre = (residual_fused ? dout*res_scale : dout) * gain * (out_act > 0 ? 1 : slope)
 *   dx   = dpre * alpha * out_scale[b,m]                                    (written if dx)
 *   part_db[b,m,chunk]  = sum_p dpre                (partial sums; caller reduces)
 *   part_dn[b,m,chunk]  = sum_p dpre*noise[b,p]
 *   part_dyy[b,m,chunk] = sum_p dpre*y_rec, y_rec = pre - noise*strength - bias (the scaled conv
 *                         output; gives d(out_scale) = dyy / out_scale)
 * `out_act` is the activation output before any fused residual, so callers that fused a
 * residual pass that pre-residual value.  nchunks = tbg_bias_act_bwd_chunks(HW).
 * ---------------------------------------------------------------------------------------- */
int tbg_bias_act_fwd_f32(const float *x, float *y, int B, int M, int HW, const tbg_epilogue *epi,
                         void *stream);
int tbg_bias_act_bwd_chunks(int HW);
/* y[i] = epilogue(sum_{s<nslab} x[s*B*M*HW + i]) -- the second half of a split-K convolution (nslab =
 * ksplit), and the small-plane form of tbg_bias_act_fwd_f32 (nslab = 1): one flat pass. */
int tbg_slab_epilogue_f32(const float *x, float *y, int B, int M, int HW, int nslab,
                          const tbg_epilogue *epi, void *stream);
/* The same with a UNIT SINK (epi->units_out required; planes [H, W], M % 8 == 0): y (may be NULL) = epilogue(sum of slabs) as
 * above AND units_out = units(y * units_scale), ring included -- the second half of a split-K convolution whose result the next
 * convolution consumes as a unit tensor.  Same sums in the same (slab) order as tbg_slab_epilogue_f32. */
int tbg_slab_epilogue_units_f32(const float *x, float *y, int B, int M, int H, int W, int nslab,
                                const tbg_epilogue *epi, void *stream);

int tbg_bias_act_bwd_f32(const float *dout, const float *out_act, float *dx, float *dpre_out,
                         float *part_db, float *part_dn, float *part_dyy, int B, int M, int HW,
                         const tbg_epilogue *epi, void *stream);

/* Second-order pieces of the regularised passes (path-length term training_step.py:300-347, R1 :349-373): the inner gradient of a
 * fused layer out = act(d * L_w(s * x) + noise * strength + b) is a function (dx, ds, dd) of (dout, x, w, s, d) whose own gradient
 * is made of the convolution launches above plus two per-plane elementwise forms (nchunks = tbg_bias_act_bwd_chunks(HW)):
 *   tbg_axpby_planes_f32:  y[pl][i] = sa[pl] * a[pl][i] + sb[pl] * b[pl][i]   (sa / sb NULL = 1; b NULL = no second term; y NULL =
 *                          sums only);  part[pl][chunk] = sum_i c[pl][i] * a[pl][i]  when part != NULL.
 *   tbg_bias_act_bwd2_f32: with m = act'(out_act) * gain, d = alpha * out_scale[pl], yd = pre(out_act) - noise * strength - bias:
 *                          g_dout = m * (d * c + gdd[pl] * yd / d)  (gdd NULL = 0);  part[pl][chunk] = sum_i dout * m * c.
 * Replaces, in the reference's terms, the second tf.gradients pass through x*s -> conv -> *d -> noise/bias/lrelu
 * (modulated_conv2d.py:94-96,119-121; noise.py:12-22; bias_act.py:25-34). */
int tbg_axpby_planes_f32(const float *a, const float *sa, const float *b, const float *sb, const float *c, float *y,
                         float *part, int planes, int HW, void *stream);
int tbg_bias_act_bwd2_f32(const float *c, const float *out_act, const float *dout, const float *gdd, float *g_dout,
                          float *part, int B, int M, int HW, const tbg_epilogue *epi, void *stream);

/* ------------------------------------------------------------------------------------------
 * Small-tensor tails of the layer gradients: one launch each instead of a chain of tiny reductions / GEMMs.
 * tbg_modconv_bwd_smalls_f32 (modulated_conv2d.py:78-82, activation-scaling form), from the partial sums of
 * tbg_bias_act_bwd_f32 (pdb, pdn, pdy: [B,O,nch]), d [B,O], s [B,I], wsq [I,O] and the conv's style-dot ds_conv [B,I,ds_slots]
 * (the data-gradient launch's dot_out partial sums as they are: summed here in slot order, round 5):
 *   t[b,o] = (sum_ch pdy) d^2;  ds[b,i] = sum_slots ds_conv - s * sum_o t wsq[i,o];  dwsq[i,o] = sum_b s^2 t;
 *   db[o] = sum pdb;  dstrength = sum pdn (pdn / dstrength may both be NULL).
 * tbg_torgb_bwd_smalls_f32 (to_rgb.py:28-33), from the per-pixel-chunk Gram partials G [B,C,nchunk,O] and the masked-dy sums
 * dysum [B,nchunk,O] of tbg_rgb_backproject_f32 (both summed here in chunk order; dysum / db may both be NULL):
 *   ds[b,c] = coef sum_o G w[c,o];  dw[c,o] = coef sum_b G s[b,c];  db[o] = sum_{b,chunk} dysum.
 * tbg_minibatch_std_{fwd,bwd}_f32 (mini_batch_std.py:10-35, first order): x [B,C,HW] -> y [B,C+1,HW]; groups of
 * min(group, B) samples {g*M + m}; B must be a multiple of the group size (EINVAL otherwise, as the reference's reshape).
 * ---------------------------------------------------------------------------------------- */
int tbg_modconv_bwd_smalls_f32(const float *pdb, const float *pdn, const float *pdy, const float *d,
                               const float *s, const float *wsq, const float *ds_conv, float *db,
                               float *dstrength, float *ds, float *dwsq, int B, int I, int O, int nch,
                               int ds_slots, void *stream);
int tbg_torgb_bwd_smalls_f32(const float *G, const float *w, const float *s, float *ds, float *dw, int B,
                             int C, int O, float coef, int nchunk, const float *dysum, float *db, void *stream);
int tbg_minibatch_std_fwd_f32(const float *x, float *y, int B, int C, int HW, int group, void *stream);
int tbg_minibatch_std_bwd_f32(const float *x, const float *dy, float *dx, int B, int C, int HW, int group,
                              void *stream);

/* Equalised-LR dense layer, one launch per direction (reference layers/dense.py:23-29 + layers/bias_act.py:25-34;
 * the mapping network mapping_block.py:20-45 and every style affine modulated_conv2d.py:52-56):
 *   out[r,n] = act(alpha * sum_k x[r,k] w[k,n] + beta * b[n]) + offset      act = leaky_relu(0.2) if lrelu else identity
 * (the caller folds the sqrt(2) activation gain into alpha / beta: lrelu is positively homogeneous).
 * backward: gm = dout * act'(out - offset);  dx = alpha gm w^T;  dw = alpha x^T gm;  db = beta sum_r gm.
 * Any of dx / dw / db may be NULL (not computed); `out` is only read when lrelu != 0.  b may be NULL (no bias).
 * tbg_dense_fwd_f32 stages whole rows of x in LDS: K <= 768 (TBG_ERANGE above -- such layers are library GEMMs). */
int tbg_dense_fwd_f32(const float *x, const float *w, const float *b, float *out, int R, int K, int N,
                      float alpha, float beta, int lrelu, float offset, void *stream);
int tbg_dense_bwd_f32(const float *x, const float *w, const float *out, const float *dout, float *dx, float *dw,
                      float *db, int R, int K, int N, float alpha, float beta, int lrelu, float offset,
                      void *stream);

/* Several LINEAR dense layers that share the row count R, the reduction length K and alpha / beta / offset, in one launch
 * per direction -- the generator's per-layer style affines (modulated_conv2d.py:52-56: s = dense(w_latent) + b + 1, one
 * per modulated conv; synthesis_block.py:120-156 hands layer l row l of the [B, L, K] latent tensor).
 *   forward:  out_i[r,n] = alpha * sum_k x_i[r*ldx + k] w_i[k,n] + beta * b_i[n] + offset        (b_i may be NULL)
 *   backward: dx_i[r*ldx + k] = alpha sum_n dout_i[r,n] w_i[k,n];  dw_i = alpha x_i^T dout_i;  db_i = beta sum_r dout_i
 * x_i / dx_i rows have pitch ldx >= K floats (slices of one [R, L, K] tensor: x_i = base + i*K, ldx = L*K), w_i is [K, N_i]
 * row-major, out_i / dout_i are [R, N_i] contiguous; any of dx / dw / db may be NULL.  `items` is a HOST array of
 * n <= TBG_DENSE_MAX_ITEMS entries (copied into the kernel arguments: no device table, graph-capturable). K <= 768. */
#define TBG_DENSE_MAX_ITEMS 24
typedef struct tbg_dense_item {
  const float *x, *w, *b, *dout;
  float *out, *dx, *dw, *db;
  int N, ldx;
} tbg_dense_item;
int tbg_dense_multi_fwd_f32(const tbg_dense_item *items, int n, int R, int K, float alpha, float beta, float offset,
                            void *stream);
int tbg_dense_multi_bwd_f32(const tbg_dense_item *items, int n, int R, int K, float alpha, float beta, void *stream);

/* ------------------------------------------------------------------------------------------
 * Optimiser / EMA (multi-tensor over one flat buffer).
 * Keras Adam (reference train.py:58-75 -> ResourceApplyAdam): m=b1 m+(1-b1)g; v=b2 v+(1-b2)g^2;
 * theta -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps).  `step` is a DEVICE int64 holding t-1
 * (iterations before this call); the kernel does not modify it.
 * tbg_ema_lerp: dst = src + (dst - src)*beta   (generator.py:48-59).
 * ---------------------------------------------------------------------------------------- */
int tbg_adam_tf_f32(float *theta, float *m, float *v, const float *g, long long n, float lr,
                    float beta1, float beta2, float eps, const long long *step, void *stream);
int tbg_ema_lerp_f32(float *dst, const float *src, long long n, float beta, void *stream);

/* demodulation coefficients (modulated_conv2d.py:78-82 in the activation-scaling form):
 * d[b,o] = rsqrt( sum_i s[b,i]^2 * wsq[i,o] + 1e-8 ),  wsq[i,o] = coef^2 * sum_t w[t,i,o]^2.
 * wsq is a caller-provided [I*O] workspace (also an output, reused by the backward). */
int tbg_demod_coefs_f32(const float *s, const float *w, float *wsq, float *d, int B, int T, int I,
                        int O, float coef, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TBG_H_ */
