// skred_mix_kernels.hip -- partial-mix reduction and the master volume stage.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "skred_launch.h"

// ---------------------------------------------------------------- partial-mix reduction

// Stage 1: partial[W][ncols] -> tmp[SK_RED_SLABS][ncols]; stage 2: tmp -> out[ncols].
// Rows are added in a fixed order in both stages (bit-reproducible, no atomics).
#define SK_RED_SLABS 16
__global__ __launch_bounds__(256) void sk_reduce_kernel(const float *__restrict__ partial,
                                                        float *__restrict__ out, int W, int ncols) {
  __shared__ float part[4][64];
  const int c = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + c;
  const int slabs = gridDim.y;
  const int w0 = (int)((long long)W * blockIdx.y / slabs), w1 = (int)((long long)W * (blockIdx.y + 1) / slabs);
  float s = 0.0f;
  if (col < ncols)
    for (int w = w0 + slice; w < w1; w += 4) s += partial[(size_t)w * ncols + col];
  part[slice][c] = s;
  __syncthreads();
  if (slice == 0 && col < ncols)
    out[(size_t)blockIdx.y * ncols + col] = ((part[0][c] + part[1][c]) + part[2][c]) + part[3][c];
}

// ---------------------------------------------------------------- master volume

// synth.c:616-624.  The gain is a serial one-pole recurrence over frames, so one lane walks it
// (bit parity forbids a parallel scan); the other lanes then scale and interleave.
// gain_state[0] is the smoother state carried between launches.
#define SK_MASTER_TILE 1024
__global__ __launch_bounds__(256) void sk_master_kernel(const float *__restrict__ sum,
                                                        float *__restrict__ out, int num_frames,
                                                        int num_channels, float target, float k,
                                                        float *gain_state) {
  __shared__ float gains[SK_MASTER_TILE];
  __shared__ float carry;
  if (threadIdx.x == 0) carry = gain_state[0];
  __syncthreads();
  for (int f0 = 0; f0 < num_frames; f0 += SK_MASTER_TILE) {
    const int n = min(SK_MASTER_TILE, num_frames - f0);
    if (threadIdx.x == 0) {
      float vg = carry;
      for (int i = 0; i < n; ++i) {
        vg += k * (target - vg);
        gains[i] = vg;
      }
      carry = vg;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) {
      const float vg = gains[i];
      const float2 s = reinterpret_cast<const float2 *>(sum)[f0 + i];
      out[(size_t)(f0 + i) * num_channels + 0] = s.x * vg;
      out[(size_t)(f0 + i) * num_channels + 1] = s.y * vg;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) gain_state[0] = carry;
}

// Single-GPU path: the last reduction stage and the master stage in one launch.  `rows` holds W <= 4*SK_RED_SLABS
// partial rows (stage-1 output, or the workgroup rows themselves for small banks); column sums use the association
// of sk_reduce_kernel (four strided slices, then ((p0+p1)+p2)+p3), so the samples equal the unfused path's.
// One workgroup per SK_RM_FRAMES frames: wave 0 walks the serial gain recurrence from the launch's first frame up
// to its own last one (redundant across workgroups, but they run side by side) while waves 1..3 add the rows.
// The carried gain is read from gain_in and written to gain_out (two slots: no workgroup may see the new value).
#define SK_RM_FRAMES 64
__global__ __launch_bounds__(256) void sk_reduce_master_kernel(const float *__restrict__ rows, int W,
                                                               float *__restrict__ out, int num_frames,
                                                               int num_channels, float target, float k,
                                                               const float *__restrict__ gain_in,
                                                               float *__restrict__ gain_out) {
  __shared__ float gains[SK_RM_FRAMES];
  __shared__ float sums[2 * SK_RM_FRAMES];
  const int ncols = 2 * num_frames;
  const int f0 = blockIdx.x * SK_RM_FRAMES;
  const int n = min(SK_RM_FRAMES, num_frames - f0);
  if (threadIdx.x < 64) {
    if (threadIdx.x == 0) {
      float vg = gain_in[0];
      for (int i = 0; i < f0; ++i) vg += k * (target - vg);
      for (int i = 0; i < n; ++i) {
        vg += k * (target - vg);
        gains[i] = vg;
      }
      if (f0 + n == num_frames) gain_out[0] = vg;
    }
  } else {
    for (int c = threadIdx.x - 64; c < 2 * n; c += 192) {
      const float *col = rows + (size_t)2 * f0 + c;
      float p[4];
#pragma unroll
      for (int slice = 0; slice < 4; ++slice) {
        float s = 0.0f;
        for (int w = slice; w < W; w += 4) s += col[(size_t)w * ncols];
        p[slice] = s;
      }
      sums[c] = ((p[0] + p[1]) + p[2]) + p[3];
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < n) {
    const float vg = gains[threadIdx.x];
    out[(size_t)(f0 + threadIdx.x) * num_channels + 0] = sums[2 * threadIdx.x] * vg;
    out[(size_t)(f0 + threadIdx.x) * num_channels + 1] = sums[2 * threadIdx.x + 1] * vg;
  }
}

// ---------------------------------------------------------------- launchers (C linkage)

extern "C" int sk_launch_reduce(const float *partial, float *tmp, float *out, int W, int ncols,
                                hipStream_t stream) {
  const unsigned cols = (unsigned)((ncols + 63) / 64);
  if (W <= 4 * SK_RED_SLABS) {
    hipLaunchKernelGGL(sk_reduce_kernel, dim3(cols, 1), dim3(256), 0, stream, partial, out, W, ncols);
  } else {
    hipLaunchKernelGGL(sk_reduce_kernel, dim3(cols, SK_RED_SLABS), dim3(256), 0, stream, partial, tmp, W, ncols);
    hipLaunchKernelGGL(sk_reduce_kernel, dim3(cols, 1), dim3(256), 0, stream, tmp, out, SK_RED_SLABS, ncols);
  }
  return (int)hipGetLastError();
}

// partial[W][2F] -> out[F][channels]: stage 1 (large W only), then the fused last stage + master volume
extern "C" int sk_launch_reduce_master(const float *partial, float *tmp, int W, float *out, int num_frames,
                                       int num_channels, float target, float k, const float *gain_in,
                                       float *gain_out, hipStream_t stream) {
  const int ncols = 2 * num_frames;
  const float *rows = partial;
  int n_rows = W;
  if (W > 4 * SK_RED_SLABS) {
    const unsigned cols = (unsigned)((ncols + 63) / 64);
    hipLaunchKernelGGL(sk_reduce_kernel, dim3(cols, SK_RED_SLABS), dim3(256), 0, stream, partial, tmp, W, ncols);
    rows = tmp;
    n_rows = SK_RED_SLABS;
  }
  hipLaunchKernelGGL(sk_reduce_master_kernel, dim3((unsigned)((num_frames + SK_RM_FRAMES - 1) / SK_RM_FRAMES)), dim3(256),
                     0, stream, rows, n_rows, out, num_frames, num_channels, target, k, gain_in, gain_out);
  return (int)hipGetLastError();
}

extern "C" int sk_reduce_tmp_floats(int ncols) { return SK_RED_SLABS * ncols; }

extern "C" int sk_launch_master(const float *sum, float *out, int num_frames, int num_channels,
                                float target, float k, float *gain_state, hipStream_t stream) {
  hipLaunchKernelGGL(sk_master_kernel, dim3(1), dim3(256), 0, stream, sum, out, num_frames,
                     num_channels, target, k, gain_state);
  return (int)hipGetLastError();
}
