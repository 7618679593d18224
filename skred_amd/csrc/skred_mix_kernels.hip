// skred_mix_kernels.hip -- the master volume stage as a kernel of its own (multi-GPU form: it runs on the root after
// the RCCL sum of the per-GPU partial mixes).  The single-GPU block applies it inside the render kernel, which also
// adds up the workgroup rows (skred_kernel_common.hpp: sk_finish_block): there are no reduction kernels any more.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "skred_launch.h"

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

// The same stage when the gains of the block were already walked by the render kernel's gain workgroup (multi-GPU form:
// the render ran before the RCCL sum): scale, interleave, and commit the carried gain -- nothing serial is left.
__global__ __launch_bounds__(256) void sk_master_apply_kernel(const float *__restrict__ sum, const float *__restrict__ gains,
                                                              float *__restrict__ out, int num_frames, int num_channels,
                                                              const float *__restrict__ gain_pending, float *gain_state) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < num_frames) {
    const float vg = gains[i];
    const float2 s = reinterpret_cast<const float2 *>(sum)[i];
    out[(size_t)i * num_channels + 0] = s.x * vg;
    out[(size_t)i * num_channels + 1] = s.y * vg;
  }
  if (i == 0) gain_state[0] = gain_pending[0];
}

// ---------------------------------------------------------------- launchers (C linkage)

extern "C" int sk_launch_master(const float *sum, float *out, int num_frames, int num_channels,
                                float target, float k, float *gain_state, hipStream_t stream) {
  hipLaunchKernelGGL(sk_master_kernel, dim3(1), dim3(256), 0, stream, sum, out, num_frames,
                     num_channels, target, k, gain_state);
  return (int)hipGetLastError();
}

extern "C" int sk_launch_master_apply(const float *sum, const float *gains, float *out, int num_frames, int num_channels,
                                      const float *gain_pending, float *gain_state, hipStream_t stream) {
  hipLaunchKernelGGL(sk_master_apply_kernel, dim3((unsigned)((num_frames + 255) / 256)), dim3(256), 0, stream, sum, gains, out,
                     num_frames, num_channels, gain_pending, gain_state);
  return (int)hipGetLastError();
}
