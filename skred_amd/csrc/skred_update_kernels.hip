// skred_update_kernels.hip -- scatter a batch of voice updates into the device planes (gfx950 / CDNA4).
//
// One thread per record (a batch is the handful of voices a control action touched in one audio block; the
// launch is latency, not bandwidth).  Parameter planes are overwritten whole; state planes are patched word by
// word so that whatever the update does not name keeps the value the render kernels last stored.  Records of one
// launch name distinct voices (the host splits batches), so there are no write conflicts.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "skred_launch.h"

struct sk_plane_ptrs_t {
  sk_plane_t *ro[SKP_COUNT];
  sk_plane_t *rw[SKS_COUNT];
};

// Every voice a control action touches goes on the motion list of the two-per-lane render family (a bit per voice, carried on
// the device: skred_device_layout.h, mask_cur): whatever the action did to its envelope, the envelope kernel renders it in the
// next block and keeps it until it is at rest again.  The render kernels therefore never have to be TOLD that a note started.
__device__ __forceinline__ void sk_list_voice(uint64_t *mask, int v) {
  atomicOr(reinterpret_cast<unsigned long long *>(mask) + (v >> 6), 1ull << (v & 63));
}

// The batch was read (straight from the host's pinned staging buffer, for small batches): the workgroup that finishes last tells
// the host, which then reuses the buffer -- a word in pinned memory the host polls, instead of an event behind every batch
// (an event record costs the stream ~5 us of gap in front of the next kernel: four per block under note traffic).
__device__ __forceinline__ void sk_batch_done(uint32_t *cnt, uint32_t *done, uint32_t seq) {
  if (!done) return;
  __syncthreads();                                   // every thread of this workgroup holds its record in registers
  if (threadIdx.x == 0) {
    __threadfence();
    if (atomicAdd(cnt, 1u) == gridDim.x - 1) {
      __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed for the slot's next batch ...
      __threadfence_system();                                                   // ... before the host can know the slot is free
      __hip_atomic_store(done, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ __launch_bounds__(64) void sk_update_kernel(const sk_update_t *__restrict__ u, int n, sk_plane_ptrs_t p,
                                                      uint64_t now, uint64_t *mask, uint32_t *cnt, uint32_t *done, uint32_t seq) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  if (i < n) {
  const sk_update_t r = u[i];
  const int v = r.voice;
  sk_list_voice(mask, v);
  const uint32_t d = r.dirty;
  if (d & SKU_PARAMS) {
#pragma unroll
    for (int k = 0; k < SKP_COUNT; ++k)     // the envelope clock plane is its own kind: a stamped note-off must survive
      if (k != SKP_ENV_S) *reinterpret_cast<uint4 *>(&p.ro[k][v]) = *reinterpret_cast<const uint4 *>(&r.ro[k]);
  }
  if (d & SKU_ENV_CLOCK) *reinterpret_cast<uint4 *>(&p.ro[SKP_ENV_S][v]) = *reinterpret_cast<const uint4 *>(&r.ro[SKP_ENV_S]);
  uint4 s0 = *reinterpret_cast<const uint4 *>(&p.rw[SKS_OSC][v]);
  uint4 s1 = *reinterpret_cast<const uint4 *>(&p.rw[SKS_FILT][v]);
  uint4 s2 = *reinterpret_cast<const uint4 *>(&p.rw[SKS_MISC][v]);
  if (d & SKU_PHASE) { s0.x = r.rw[SKS_OSC].w[0]; s1.w = (s1.w & ~SKR_FINISHED) | (r.rw[SKS_FILT].w[3] & SKR_FINISHED); }
  if (d & SKU_ENV_STATE) s1.w = (s1.w & ~SKR_ENV_ACTIVE) | (r.rw[SKS_FILT].w[3] & SKR_ENV_ACTIVE);
  if (d & SKU_PAN) { s2.z = r.rw[SKS_MISC].w[2]; s2.w = r.rw[SKS_MISC].w[3]; }
  if (d & SKU_FILTER_STATE) { s0.z = r.rw[SKS_OSC].w[2]; s0.w = r.rw[SKS_OSC].w[3]; s1.x = r.rw[SKS_FILT].w[0]; s1.y = r.rw[SKS_FILT].w[1]; }
  if (d & SKU_SMOOTHER) s0.y = r.rw[SKS_OSC].w[1];
  if (d & SKU_HOLD) { s2.x = r.rw[SKS_MISC].w[0]; s2.y = r.rw[SKS_MISC].w[1]; }
  if (d & SKU_SAMPLE) s1.z = r.rw[SKS_FILT].w[2];
  if (d & (SKU_STAMP_TRIGGER | SKU_STAMP_RELEASE)) {
    uint4 es = *reinterpret_cast<const uint4 *>(&p.ro[SKP_ENV_S][v]);   // after the ENV_CLOCK write above, if any
    if (d & SKU_STAMP_TRIGGER) {        // amp_envelope_trigger (velocity travels with the parameters)
      es.x = (uint32_t)now; es.y = (uint32_t)(now >> 32); es.z = 0; es.w = 0;
      s1.w |= SKR_ENV_ACTIVE;
    }
    if ((d & SKU_STAMP_RELEASE) && (s1.w & SKR_ENV_ACTIVE)) {           // amp_envelope_release
      es.z = (uint32_t)now; es.w = (uint32_t)(now >> 32);
    }
    *reinterpret_cast<uint4 *>(&p.ro[SKP_ENV_S][v]) = es;
  }
  *reinterpret_cast<uint4 *>(&p.rw[SKS_OSC][v]) = s0;
  *reinterpret_cast<uint4 *>(&p.rw[SKS_FILT][v]) = s1;
  *reinterpret_cast<uint4 *>(&p.rw[SKS_MISC][v]) = s2;
  }
  sk_batch_done(cnt, done, seq);
}

// note-ons / note-offs only: a list of voice ids, the clock, which stamp
__global__ __launch_bounds__(256) void sk_stamp_kernel(const int32_t *__restrict__ ids, int n, uint32_t dirty, sk_plane_ptrs_t p,
                                                       uint64_t now, uint64_t *mask, uint32_t *cnt, uint32_t *done, uint32_t seq) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
  const int v = ids[i];
  sk_list_voice(mask, v);
  uint32_t *rwflags = reinterpret_cast<uint32_t *>(&p.rw[SKS_FILT][v]) + 3;
  uint4 es = *reinterpret_cast<const uint4 *>(&p.ro[SKP_ENV_S][v]);
  uint32_t f = *rwflags;
  if (dirty & SKU_STAMP_TRIGGER) {
    es.x = (uint32_t)now; es.y = (uint32_t)(now >> 32); es.z = 0; es.w = 0;
    f |= SKR_ENV_ACTIVE;
  }
  if ((dirty & SKU_STAMP_RELEASE) && (f & SKR_ENV_ACTIVE)) {
    es.z = (uint32_t)now; es.w = (uint32_t)(now >> 32);
  }
  *reinterpret_cast<uint4 *>(&p.ro[SKP_ENV_S][v]) = es;
  *rwflags = f;
  }
  sk_batch_done(cnt, done, seq);
}

extern "C" int sk_launch_stamp(const int32_t *d_ids, int n, uint32_t dirty, sk_plane_t *const ro[SKP_COUNT],
                               sk_plane_t *const rw[SKS_COUNT], uint64_t now, uint64_t *mask, uint32_t *cnt, uint32_t *done,
                               uint32_t seq, hipStream_t stream) {
  if (n <= 0) return 0;
  sk_plane_ptrs_t p;
  for (int k = 0; k < SKP_COUNT; ++k) p.ro[k] = ro[k];
  for (int k = 0; k < SKS_COUNT; ++k) p.rw[k] = rw[k];
  hipLaunchKernelGGL(sk_stamp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_ids, n, dirty, p, now, mask, cnt, done, seq);
  return (int)hipGetLastError();
}

extern "C" int sk_launch_update(const sk_update_t *d_updates, int n, sk_plane_t *const ro[SKP_COUNT],
                                sk_plane_t *const rw[SKS_COUNT], uint64_t now, uint64_t *mask, uint32_t *cnt, uint32_t *done,
                                uint32_t seq, hipStream_t stream) {
  if (n <= 0) return 0;
  sk_plane_ptrs_t p;
  for (int k = 0; k < SKP_COUNT; ++k) p.ro[k] = ro[k];
  for (int k = 0; k < SKS_COUNT; ++k) p.rw[k] = rw[k];
  hipLaunchKernelGGL(sk_update_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, stream, d_updates, n, p, now, mask, cnt, done, seq);
  return (int)hipGetLastError();
}

// Packed lanes (skred_device_layout.h: pack_mask): the reference sets voice_sample = 0 on every frame for a voice it skips
// (synth.c:531-542).  A skipped voice with a lane gets that from the render kernel; one WITHOUT a lane -- it cannot sound and no
// voice that can sound reads it -- gets it here, once, whenever the words changed or state was written from outside.
__global__ __launch_bounds__(256) void sk_pack_zero_kernel(const uint64_t *__restrict__ mask, sk_plane_t *filt, int n_voices_padded) {
  const int v = blockIdx.x * 256 + threadIdx.x;
  if (v >= n_voices_padded) return;
  if ((mask[v >> 6] >> (v & 63)) & 1) return;
  uint32_t *smp = reinterpret_cast<uint32_t *>(&filt[v]) + 2;
  if (*smp != 0u) *smp = 0u;
}

extern "C" int sk_launch_pack_zero(const uint64_t *d_mask, sk_plane_t *filt, int n_voices_padded, hipStream_t stream) {
  hipLaunchKernelGGL(sk_pack_zero_kernel, dim3((unsigned)((n_voices_padded + 255) / 256)), dim3(256), 0, stream, d_mask, filt, n_voices_padded);
  return (int)hipGetLastError();
}
