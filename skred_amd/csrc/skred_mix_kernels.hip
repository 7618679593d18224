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

extern "C" int sk_reduce_tmp_floats(int ncols) { return SK_RED_SLABS * ncols; }

extern "C" int sk_launch_master(const float *sum, float *out, int num_frames, int num_channels,
                                float target, float k, float *gain_state, hipStream_t stream) {
  hipLaunchKernelGGL(sk_master_kernel, dim3(1), dim3(256), 0, stream, sum, out, num_frames,
                     num_channels, target, k, gain_state);
  return (int)hipGetLastError();
}
