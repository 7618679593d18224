// tools/issue_rate.hip -- microbenchmark: VALU issue rate of a wavefront that has a SIMD to itself (gfx950).
// Each wave runs `iters` x 64 v_add_f32 (or v_pk_add_f32) either as ONE dependent chain or as FOUR independent chains
// interleaved; the grid puts 1, 2 or 4 waves on every SIMD (256 CUs x 4 SIMDs).  Reports cycles per instruction per
// wave from s_memtime-free wall time at the measured shader clock.  Build:
//   hipcc -O3 --offload-arch=gfx950 tools/issue_rate.hip -o tools/issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(X) X X X X
#define REP16(X) REP4(REP4(X))

// MODE 0: one dependent chain; 1: four independent chains; 2: dependent chain of packed adds; 3: four packed chains
template <int MODE>
__global__ __launch_bounds__(256) void spin(int iters, float *out) {
  float a = threadIdx.x, b = a + 1.0f, c = a + 2.0f, d = a + 3.0f;
  const float k = 1.0f;
  typedef float v2 __attribute__((ext_vector_type(2)));
  v2 pa = {a, b}, pb = {c, d}, pc = {a, c}, pd = {b, d};
  const v2 pk = {1.0f, 1.0f};
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) { REP16(asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1" : "+v"(a) : "v"(k));) }
    if (MODE == 1) { REP16(asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(k));) }
    if (MODE == 2) { REP16(asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %0, %0, %1" : "+v"(pa) : "v"(pk));) }
    if (MODE == 3) { REP16(asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4" : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd) : "v"(pk));) }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + pa.x + pa.y + pb.x + pb.y + pc.x + pc.y + pd.x + pd.y;
}

int main() {
  const int iters = 4096;                 // x 64 instructions
  float *out;
  (void)hipMalloc(&out, 4096 * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  int clk_khz = 0;
  (void)hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  printf("reported shader clock: %d MHz\n", clk_khz / 1000);
  const char *names[] = {"v_add_f32, one dependent chain", "v_add_f32, four independent chains", "v_pk_add_f32, one dependent chain", "v_pk_add_f32, four independent chains"};
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  double per_simd[4] = {0, 0, 0, 0};                 // instructions per second and SIMD with 4 waves resident, per mode
  for (int m = 0; m < 4; ++m) {
    for (int wps = 1; wps <= 4; wps *= 2) {          // waves per SIMD: blocks of 4 waves, 256 CUs
      const int blocks = cus * wps;
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        switch (m) {
          case 0: hipLaunchKernelGGL(spin<0>, dim3(blocks), dim3(256), 0, 0, iters, out); break;
          case 1: hipLaunchKernelGGL(spin<1>, dim3(blocks), dim3(256), 0, 0, iters, out); break;
          case 2: hipLaunchKernelGGL(spin<2>, dim3(blocks), dim3(256), 0, 0, iters, out); break;
          case 3: hipLaunchKernelGGL(spin<3>, dim3(blocks), dim3(256), 0, 0, iters, out); break;
        }
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      const double n_inst = (double)iters * 64.0;
      const double cyc = best * 1e-3 * (clk_khz * 1e3) / n_inst;
      printf("%-40s %d wave(s)/SIMD: %.3f ms  -> %.2f cycles per instruction per wave (%.2f per SIMD)\n", names[m], wps, best, cyc, cyc / wps);
      if (wps == 4) per_simd[m] = n_inst * wps / (best * 1e-3);
    }
  }
  // what bench.py prices the kernels' VALU instruction rate against (independent chains: what a SIMD with four waves can issue)
  printf("PEAK plain %.6e packed %.6e simds %d clock_mhz %d\n", per_simd[1], per_simd[3], cus * 4, clk_khz / 1000);
  return 0;
}
