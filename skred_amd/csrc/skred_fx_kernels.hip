// skred_fx_kernels.hip -- the FIXED-POINT render path on gfx950 (definition: oracle/cpu_ref_fxpt.c,
// include/skred_amd_fxpt.h).  Same mapping as the float path: one lane per voice, the frame loop
// inside the kernel with phase / smoother state in registers, the int16 LUT pool staged in LDS,
// integer DPP wave sum, wave sums combined as int64 in LDS every 64 frames, per-workgroup int64
// partials, fixed-order reduction.  Every operation is integer, so voices AND the mix are
// bit-exact against the CPU definition whatever the order of the additions.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "skred_fx_layout.h"

// integer wave sum of two values into lane 63 (see skred_kernel_common.hpp: wave_sum2_to_lane63)
__device__ __forceinline__ void wave_isum2_to_lane63(int &l, int &r) {
  asm volatile(
      "s_nop 1\n\t"
      "v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_u32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_add_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_u32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_add_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_u32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_add_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_u32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_add_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_u32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_add_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_u32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(l), "+v"(r));
}

__device__ __forceinline__ uint32_t sat32(uint64_t x) { return x > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)x; }

template <bool STEMS>
__global__ __launch_bounds__(SKX_GROUP) void sk_fx_render_kernel(const skx_args_t a) {
  extern __shared__ int16_t lut_lds[];                               // [lds_entries] then int2 wsum[4][SKX_CHUNK]
  int2 *wsum = reinterpret_cast<int2 *>(reinterpret_cast<char *>(lut_lds) + a.lds_bytes_tables);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool lut_in_lds = a.lds_bytes_tables > 0;
  if (lut_in_lds) {
    const int n4 = a.lds_bytes_tables >> 4;
    const uint4 *src = reinterpret_cast<const uint4 *>(a.tables);
    uint4 *dst = reinterpret_cast<uint4 *>(lut_lds);
    for (int i = tid; i < n4; i += SKX_GROUP) dst[i] = src[i];
    __syncthreads();
  }
  const size_t part_base = (size_t)blockIdx.x * (size_t)a.num_frames * 2;
  bool first_pass = true;

  for (int g = blockIdx.x; g < a.n_groups; g += gridDim.x) {
    const int v = g * SKX_GROUP + tid;
    const uint4 p0 = *reinterpret_cast<const uint4 *>(&a.ro[SKX_OSC][v]);
    const uint4 p1 = *reinterpret_cast<const uint4 *>(&a.ro[SKX_GAIN][v]);
    const uint4 p2 = *reinterpret_cast<const uint4 *>(&a.ro[SKX_ENV][v]);
    const uint4 p3 = *reinterpret_cast<const uint4 *>(&a.ro[SKX_RECIP][v]);
    const uint4 p4 = *reinterpret_cast<const uint4 *>(&a.ro[SKX_TIME][v]);
    const uint4 st = *reinterpret_cast<const uint4 *>(&a.rw[v]);
    const uint32_t inc = p0.x;
    const int toff = (int)p0.y, L = (int)(p0.z & 31u);
    const uint32_t flags = p0.z >> 8;
    const int amp = (int)p0.w;
    const int pan_l = (int)p1.x, pan_r = (int)p1.y, k = (int)p1.z, vel = (int)p1.w;
    const uint32_t A = p2.x, D = p2.y, R = p2.z;
    const int S = (int)p2.w;
    const uint32_t rA = p3.x, rD = p3.y, rR = p3.z;
    const uint64_t AD = (uint64_t)A + D;
    const uint64_t t_start = ((uint64_t)p4.y << 32) | p4.x, t_release = ((uint64_t)p4.w << 32) | p4.z;
    uint32_t phase = st.x;
    int sg = (int)st.y, sample = (int)st.z;
    uint32_t active = st.w & 1u;
    const bool dead = amp == 0 || (flags & SKXF_INERT);
    const int16_t *lut = (lut_in_lds ? lut_lds : a.tables) + toff;
    const uint32_t mask = (1u << L) - 1u;

    for (int c0 = 0; c0 < a.num_frames; c0 += SKX_CHUNK) {
      const int cn = min(SKX_CHUNK, a.num_frames - c0);
      for (int j = 0; j < cn; ++j) {
        const uint64_t now = a.count0 + (uint64_t)(c0 + j) + 1;
        int l = 0, r = 0;
        if (!dead) {
          phase += inc;
          const uint32_t idx = phase >> (32 - L);
          int s = lut[idx];
          if (a.interp) {
            const int nxt = lut[(idx + 1) & mask];
            const int frac = (int)((uint32_t)(phase << L) >> 17);
            s = s + (((nxt - s) * frac) >> 15);
          }
          int e = 32768;
          if (flags & SKXF_USE_ENV) {
            int lvl = 0;
            if (active) {
              const uint32_t t = sat32(now - t_start);
              if (t < A) {
                lvl = (int)((uint32_t)(t * rA) >> 17);
              } else if ((uint64_t)t < AD) {
                const int prog = (int)((uint32_t)((t - A) * rD) >> 17);
                lvl = 32768 - ((prog * (32768 - S)) >> 15);
              } else if (t_release == 0) {
                lvl = S;
              } else {
                const uint32_t tr = sat32(now - t_release);
                if (tr < R) {
                  const int prog = (int)((uint32_t)(tr * rR) >> 17);
                  lvl = S - ((prog * S) >> 15);
                } else {
                  active = 0;
                }
              }
            }
            e = (lvl * vel) >> 15;
          }
          int gain = (amp * e) >> 15;
          if (flags & SKXF_SMOOTH) {
            sg += ((gain - sg) * k) >> 15;
            gain = sg;
          }
          sample = (s * gain) >> 15;
          if (!(flags & SKXF_MUTED)) {
            l = (sample * pan_l) >> 15;
            r = (sample * pan_r) >> 15;
          }
        }
        if (STEMS) {
          if (v < a.n_voices)
            reinterpret_cast<int2 *>(a.stems)[(size_t)(c0 + j) * (size_t)a.n_voices + (size_t)v] = make_int2(l, r);
        }
        wave_isum2_to_lane63(l, r);                      // |per-voice| < 2^17, 64 of them: fits int32
        if (lane == 63) wsum[wave * SKX_CHUNK + j] = make_int2(l, r);
      }
      __syncthreads();
      if (tid < 2 * cn) {
        const int *w = reinterpret_cast<const int *>(wsum);
        long long s = (long long)w[0 * 2 * SKX_CHUNK + tid] + w[1 * 2 * SKX_CHUNK + tid] +
                      w[2 * 2 * SKX_CHUNK + tid] + w[3 * 2 * SKX_CHUNK + tid];
        long long *p = a.partial + part_base + (size_t)c0 * 2 + tid;
        if (first_pass) *p = s; else *p += s;
      }
      __syncthreads();
    }
    if (dead) sample = 0;
    uint4 o;
    o.x = phase; o.y = (uint32_t)sg; o.z = (uint32_t)sample; o.w = active;
    *reinterpret_cast<uint4 *>(&a.rw[v]) = o;
    first_pass = false;
  }
}

__global__ __launch_bounds__(256) void sk_fx_reduce_kernel(const long long *__restrict__ partial,
                                                           long long *__restrict__ out, int W, int ncols) {
  const int col = blockIdx.x * 256 + threadIdx.x;
  if (col >= ncols) return;
  long long s = 0;
  for (int w = 0; w < W; ++w) s += partial[(size_t)w * ncols + col];
  out[col] = s;
}

extern "C" int skx_launch_render(const skx_args_t *args, int n_workgroups, hipStream_t stream) {
  const size_t lds = (size_t)args->lds_bytes_tables + (size_t)4 * SKX_CHUNK * sizeof(int2);
  dim3 grid((unsigned)n_workgroups), block(SKX_GROUP);
  if (args->stems) hipLaunchKernelGGL((sk_fx_render_kernel<true>), grid, block, lds, stream, *args);
  else             hipLaunchKernelGGL((sk_fx_render_kernel<false>), grid, block, lds, stream, *args);
  return (int)hipGetLastError();
}

extern "C" int skx_launch_reduce(const long long *partial, long long *out, int W, int ncols, hipStream_t stream) {
  hipLaunchKernelGGL(sk_fx_reduce_kernel, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, stream, partial, out, W, ncols);
  return (int)hipGetLastError();
}
