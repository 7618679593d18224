// tools/ta_rate.hip -- microbenchmark: cost of a scattered per-lane gather by access width (gfx950).
// Every lane walks its own region of a 4.7 MB float pool (the C4 PCM pool size), advancing ~1.1 floats per
// iteration, like a PCM voice.  Build: hipcc -O3 --offload-arch=gfx950 tools/ta_rate.hip -o tools/ta_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef float f2a8 __attribute__((ext_vector_type(2), aligned(8)));
typedef float f4a16 __attribute__((ext_vector_type(4), aligned(16)));

// MODE 0: one dword; 1: two dwords (idx, idx+1); 2: dwordx2 dword-aligned; 3: dwordx2 8-byte aligned (idx&~1);
// 4: dwordx4 16-byte aligned (idx&~3); 5: dwordx4 every 4th iteration only (window refill pattern)
template <int MODE>
__global__ __launch_bounds__(256) void walk(const float *__restrict__ pool, int pool_floats, int iters, float *out) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  unsigned h = tid * 2654435761u;
  float pos = (float)(h % (unsigned)(pool_floats - 4096));
  const float inc = 0.25f + 1.75f * (float)((h >> 8) & 1023) / 1024.0f;
  float acc = 0.0f;
  for (int i = 0; i < iters; ++i) {
    pos += inc;
    const int idx = (int)pos;
    if (MODE == 0) acc += pool[idx];
    if (MODE == 1) acc += pool[idx] + pool[idx + 1];
    if (MODE == 2) { const f2a4 v = *reinterpret_cast<const f2a4 *>(pool + idx); acc += v.x + v.y; }
    if (MODE == 3) { const f2a8 v = *reinterpret_cast<const f2a8 *>(pool + (idx & ~1)); acc += v.x + v.y; }
    if (MODE == 4) { const f4a16 v = *reinterpret_cast<const f4a16 *>(pool + (idx & ~3)); acc += v.x + v.w; }
    if (MODE == 6 || MODE == 7) {   // block refill: every B-th iteration, K consecutive dwordx4 (the window)
      const int B = MODE == 6 ? 8 : 16, K = MODE == 6 ? 5 : 9;
      if ((i & (B - 1)) == 0) {
        const f4a16 *w = reinterpret_cast<const f4a16 *>(pool + (idx & ~3));
#pragma unroll
        for (int k = 0; k < K; ++k) { const f4a16 v = w[k]; acc += v.x + v.w; }
      } else acc += pos;
    }
    if (MODE == 5) { if ((i & 3) == 0) { const f4a16 v = *reinterpret_cast<const f4a16 *>(pool + (idx & ~3)); acc += v.x + v.w; } else acc += pos; }
  }
  out[tid] = acc;
}

int main() {
  const int pool_floats = 1176036, iters = 512, threads = 262144;
  float *pool, *out;
  (void)hipMalloc(&pool, (pool_floats + 64) * sizeof(float));
  (void)hipMemset(pool, 0, (pool_floats + 64) * sizeof(float));
  (void)hipMalloc(&out, threads * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const char *names[] = {"dword", "2 x dword", "dwordx2 align4", "dwordx2 align8", "dwordx4 align16", "dwordx4 every 4th iter", "5 x dwordx4 every 8th iter", "9 x dwordx4 every 16th iter"};
  for (int m = 0; m < 8; ++m) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      switch (m) {
        case 0: hipLaunchKernelGGL(walk<0>, dim3(threads / 256), dim3(256), 0, 0, pool, pool_floats, iters, out); break;
        case 1: hipLaunchKernelGGL(walk<1>, dim3(threads / 256), dim3(256), 0, 0, pool, pool_floats, iters, out); break;
        case 2: hipLaunchKernelGGL(walk<2>, dim3(threads / 256), dim3(256), 0, 0, pool, pool_floats, iters, out); break;
        case 3: hipLaunchKernelGGL(walk<3>, dim3(threads / 256), dim3(256), 0, 0, pool, pool_floats, iters, out); break;
        case 4: hipLaunchKernelGGL(walk<4>, dim3(threads / 256), dim3(256), 0, 0, pool, pool_floats, iters, out); break;
        case 5: hipLaunchKernelGGL(walk<5>, dim3(threads / 256), dim3(256), 0, 0, pool, pool_floats, iters, out); break;
        case 6: hipLaunchKernelGGL(walk<6>, dim3(threads / 256), dim3(256), 0, 0, pool, pool_floats, iters, out); break;
        case 7: hipLaunchKernelGGL(walk<7>, dim3(threads / 256), dim3(256), 0, 0, pool, pool_floats, iters, out); break;
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const double lane_ops = (double)threads * iters;
    printf("%-24s %.4f ms  %.3e lane-gathers/s  %.2f cycles/lane/CU @2.4GHz\n", names[m], best, lane_ops / (best * 1e-3),
           best * 1e-3 * 2.4e9 * 256 / lane_ops);
  }
  return 0;
}
