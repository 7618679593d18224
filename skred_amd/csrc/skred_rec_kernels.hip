// skred_rec_kernels.hip -- stem recorder: min/max scan and float -> int16 conversion (gfx950 / CDNA4).
//
// The two passes of the reference's save_wav (wire.c:150-181) over the recorded stems, which live in HBM
// as float[frames][n_voices][2] (the layout synth() writes through `user`, synth.c:607-611):
//   pass 1  fbig = max(0, all samples), fsmall = min(0, all samples)   (wire.c:150-156; NaN never wins
//           a `>` / `<` test, and max/min do not depend on the order, so a tree reduction is exact);
//   pass 2  every sample of a selected voice: g *= scale; clamp to [-1, 1]; (int16)(g * 32767.0f)
//           (wire.c:170-180), written densely in frame-major order.
// Both are streaming passes bound by HBM: 4 B read per sample in pass 1; pass 2 reads only the selected
// voices' 8-byte (L,R) pairs and writes 4 B per pair.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "skred_launch.h"

#define SK_REC_BLOCK 256
#define SK_REC_MAX_BLOCKS 2048

// partial[2*b] = max, partial[2*b+1] = min of block b's grid-stride slice; n4 = number of float4
__global__ __launch_bounds__(SK_REC_BLOCK) void sk_rec_minmax_kernel(const float4 *__restrict__ x, size_t n4,
                                                                   const float *__restrict__ tail, int n_tail,
                                                                   float *__restrict__ partial) {
  float big = 0.0f, small = 0.0f;
  for (size_t i = (size_t)blockIdx.x * SK_REC_BLOCK + threadIdx.x; i < n4; i += (size_t)gridDim.x * SK_REC_BLOCK) {
    const float4 v = x[i];
    if (v.x > big) big = v.x;  if (v.x < small) small = v.x;
    if (v.y > big) big = v.y;  if (v.y < small) small = v.y;
    if (v.z > big) big = v.z;  if (v.z < small) small = v.z;
    if (v.w > big) big = v.w;  if (v.w < small) small = v.w;
  }
  if (blockIdx.x == 0 && (int)threadIdx.x < n_tail) {
    const float g = tail[threadIdx.x];
    if (g > big) big = g;
    if (g < small) small = g;
  }
  __shared__ float sb[SK_REC_BLOCK], ss[SK_REC_BLOCK];
  sb[threadIdx.x] = big;
  ss[threadIdx.x] = small;
  __syncthreads();
  for (int s = SK_REC_BLOCK / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const float b2 = sb[threadIdx.x + s], s2 = ss[threadIdx.x + s];
      if (b2 > sb[threadIdx.x]) sb[threadIdx.x] = b2;
      if (s2 < ss[threadIdx.x]) ss[threadIdx.x] = s2;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial[2 * blockIdx.x] = sb[0];
    partial[2 * blockIdx.x + 1] = ss[0];
  }
}

// out[(frame * n_sel + k) * 2 + {0,1}] from voice sel[k]; one thread per (frame, k) pair
__global__ __launch_bounds__(SK_REC_BLOCK) void sk_rec_convert_kernel(const float2 *__restrict__ rec, long frames,
                                                                    int n_voices, const int *__restrict__ sel,
                                                                    int n_sel, float scale,
                                                                    short2 *__restrict__ out) {
  const size_t n = (size_t)frames * n_sel;
  for (size_t i = (size_t)blockIdx.x * SK_REC_BLOCK + threadIdx.x; i < n; i += (size_t)gridDim.x * SK_REC_BLOCK) {
    const size_t frame = i / n_sel;
    const int k = (int)(i - frame * n_sel);
    float2 g = rec[frame * n_voices + sel[k]];
    g.x *= scale;
    g.y *= scale;
    if (g.x > 1.0f) g.x = 1.0f;
    if (g.x < -1.0f) g.x = -1.0f;
    if (g.y > 1.0f) g.y = 1.0f;
    if (g.y < -1.0f) g.y = -1.0f;
    // C float -> int16 conversion truncates toward zero; NaN (0 * inf when the take was silent) -> 0
    const float a = g.x * 32767.0f, b = g.y * 32767.0f;
    out[i] = make_short2((short)(a == a ? (int)a : 0), (short)(b == b ? (int)b : 0));
  }
}

extern "C" int sk_rec_partial_floats(void) { return 2 * SK_REC_MAX_BLOCKS; }

// *n_blocks_out partial (max,min) pairs land in `partial`; the caller folds them (order-free)
extern "C" int sk_launch_rec_minmax(const float *rec, size_t n_floats, float *partial, int *n_blocks_out,
                                    hipStream_t stream) {
  const size_t n4 = n_floats / 4;
  size_t blocks = (n4 + SK_REC_BLOCK - 1) / SK_REC_BLOCK;
  if (blocks < 1) blocks = 1;
  if (blocks > SK_REC_MAX_BLOCKS) blocks = SK_REC_MAX_BLOCKS;
  hipLaunchKernelGGL(sk_rec_minmax_kernel, dim3((unsigned)blocks), dim3(SK_REC_BLOCK), 0, stream,
                     reinterpret_cast<const float4 *>(rec), n4, rec + 4 * n4, (int)(n_floats - 4 * n4), partial);
  *n_blocks_out = (int)blocks;
  return (int)hipGetLastError();
}

extern "C" int sk_launch_rec_convert(const float *rec, long frames, int n_voices, const int *sel, int n_sel,
                                     float scale, int16_t *out, hipStream_t stream) {
  const size_t n = (size_t)frames * n_sel;
  size_t blocks = (n + SK_REC_BLOCK - 1) / SK_REC_BLOCK;
  if (blocks < 1) blocks = 1;
  if (blocks > 65535u * 16u) blocks = 65535u * 16u;
  hipLaunchKernelGGL(sk_rec_convert_kernel, dim3((unsigned)blocks), dim3(SK_REC_BLOCK), 0, stream,
                     reinterpret_cast<const float2 *>(rec), frames, n_voices, sel, n_sel, scale,
                     reinterpret_cast<short2 *>(out));
  return (int)hipGetLastError();
}
