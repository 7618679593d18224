// tools/issue_mix.hip -- microbenchmark: what ONE instruction of each kind the render kernels are made of costs a SIMD (gfx950), with
// 1, 2 and 4 wavefronts resident per SIMD.  Each wave runs `iters` x 64 copies of the instruction (independent operands where the
// form allows); reported: shader cycles per instruction per wave and per SIMD at the measured clock (s_memtime / s_memrealtime).
//   hipcc -O3 --offload-arch=gfx950 tools/issue_mix.hip -o tools/issue_mix
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(X) X X X X
#define REP16(X) REP4(REP4(X))
#define REP64(X) REP16(REP4(X))

template <int MODE>
__global__ __launch_bounds__(256) void spin(int iters, float *out, unsigned long long *clk) {
  __shared__ float lds[4096];
  float a = threadIdx.x, b = a + 1.0f, c = a + 2.0f, d = a + 3.0f;
  int i0 = threadIdx.x, i1 = i0 + 1;
  typedef float v2 __attribute__((ext_vector_type(2)));
  v2 pa = {a, b}, pb = {c, d};
  typedef float v4 __attribute__((ext_vector_type(4)));
  v4 q4 = {a, b, c, d};
  long long w0 = threadIdx.x, w1 = threadIdx.x + 5;
  const float k = 1.0f;
  lds[threadIdx.x] = a; lds[threadIdx.x + 256] = b;
  __syncthreads();
  const unsigned addr = (threadIdx.x & 63) * 4;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) { REP16(asm volatile("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2\n v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(k));) }
    if (MODE == 1) { REP16(asm volatile("v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2\n v_mul_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(k));) }
    if (MODE == 2) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %0, %0, %2\n v_pk_mul_f32 %1, %1, %2" : "+v"(pa), "+v"(pb) : "v"(pa));) }
    if (MODE == 3) { REP16(asm volatile("v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %1, %1, %2, vcc" : "+v"(a), "+v"(b) : "v"(k) : "vcc");) }
    if (MODE == 4) { REP16(asm volatile("v_cmp_ge_f32 vcc, %0, %1\n v_cmp_ge_f32 vcc, %1, %0\n v_cmp_ge_f32 vcc, %0, %1\n v_cmp_ge_f32 vcc, %1, %0" : : "v"(a), "v"(b) : "vcc");) }
    if (MODE == 5) { REP16(asm volatile("v_cvt_i32_f32 %0, %2\n v_cvt_i32_f32 %1, %3\n v_cvt_i32_f32 %0, %2\n v_cvt_i32_f32 %1, %3" : "+v"(i0), "+v"(i1) : "v"(a), "v"(b));) }
    if (MODE == 6) { REP16(asm volatile("v_lshl_add_u32 %0, %0, 2, %2\n v_lshl_add_u32 %1, %1, 2, %2\n v_lshl_add_u32 %0, %0, 2, %2\n v_lshl_add_u32 %1, %1, 2, %2" : "+v"(i0), "+v"(i1) : "v"(i0));) }
    if (MODE == 7) { REP16(asm volatile("v_permlane32_swap_b32 %0, %1\n s_nop 0\n v_permlane32_swap_b32 %2, %3\n s_nop 0\n v_permlane32_swap_b32 %0, %1\n s_nop 0\n v_permlane32_swap_b32 %2, %3\n s_nop 0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
    if (MODE == 8) { REP16(asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf\n s_nop 1\n v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(b));) }
    if (MODE == 9) { REP16(asm volatile("v_mov_b32 %0, %2\n v_mov_b32 %1, %3\n v_mov_b32 %0, %3\n v_mov_b32 %1, %2" : "+v"(a), "+v"(b) : "v"(c), "v"(d));) }
    if (MODE == 10) { REP16(asm volatile("s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0");) }
    if (MODE == 11) { REP16(asm volatile("ds_read_b32 %0, %2\n ds_read_b32 %1, %2 offset:256\n ds_read_b32 %0, %2 offset:512\n ds_read_b32 %1, %2 offset:768\n s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b) : "v"(addr));) }
    if (MODE == 12) { REP16(asm volatile("ds_write_b32 %2, %0\n ds_write_b32 %2, %1 offset:256\n ds_write_b32 %2, %0 offset:512\n ds_write_b32 %2, %1 offset:768" : : "v"(a), "v"(b), "v"(addr));) }
    if (MODE == 13) { REP16(asm volatile("v_add_f32 %0, %0, %2\n s_add_u32 s20, s20, 1\n v_add_f32 %1, %1, %2\n s_add_u32 s21, s21, 1" : "+v"(a), "+v"(b) : "v"(k) : "s20", "s21", "scc");) }   /* VALU + SALU alternating: 4 instructions */
    if (MODE == 14) { REP16(asm volatile("v_add_f32 %0, %0, %2\n v_mul_f32 %1, %1, %2\n v_sub_f32 %0, %0, %2\n v_add_f32 %1, %1, %0" : "+v"(a), "+v"(b) : "v"(k));) }                              /* a dependent mix */
    if (MODE == 16) { REP16(asm volatile("v_cndmask_b32_e64 %0, %0, %2, s[20:21]\n v_cndmask_b32_e64 %1, %1, %2, s[20:21]\n v_cndmask_b32_e64 %0, %0, %2, s[20:21]\n v_cndmask_b32_e64 %1, %1, %2, s[20:21]" : "+v"(a), "+v"(b) : "v"(k) : "s20", "s21");) }
    if (MODE == 17) { REP16(asm volatile("v_min_u32 %0, %0, %2\n v_min_u32 %1, %1, %2\n v_min_u32 %0, %0, %2\n v_min_u32 %1, %1, %2" : "+v"(i0), "+v"(i1) : "v"(i0));) }
    if (MODE == 18) { REP16(asm volatile("v_cmp_ge_f32 vcc, %0, %2\n v_cndmask_b32 %0, %0, %2, vcc\n v_cmp_ge_f32 vcc, %1, %2\n v_cndmask_b32 %1, %1, %2, vcc" : "+v"(a), "+v"(b) : "v"(k) : "vcc");) }
    if (MODE == 19) { REP16(asm volatile("v_min_f32 %0, %0, %2\n v_max_f32 %1, %1, %2\n v_min_f32 %0, %0, %2\n v_max_f32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(k));) }
    if (MODE == 20) { REP16(asm volatile("v_med3_f32 %0, %0, %2, %3\n v_med3_f32 %1, %1, %2, %3\n v_med3_f32 %0, %0, %2, %3\n v_med3_f32 %1, %1, %2, %3" : "+v"(a), "+v"(b) : "v"(k), "v"(c));) }
    if (MODE == 21) { REP16(asm volatile("v_fract_f32 %0, %0\n v_fract_f32 %1, %1\n v_fract_f32 %0, %0\n v_fract_f32 %1, %1" : "+v"(a), "+v"(b));) }
    if (MODE == 22) { REP16(asm volatile("v_cndmask_b32 %0, %2, %3, vcc\n v_cndmask_b32 %1, %3, %2, vcc\n v_cndmask_b32 %0, %3, %2, vcc\n v_cndmask_b32 %1, %2, %3, vcc" : "=v"(a), "=v"(b) : "v"(c), "v"(d) : "vcc");) }   /* independent of their own results */
    if (MODE == 23) { REP16(asm volatile("v_bfi_b32 %0, %2, %0, %3\n v_bfi_b32 %1, %2, %1, %3\n v_bfi_b32 %0, %2, %0, %3\n v_bfi_b32 %1, %2, %1, %3" : "+v"(i0), "+v"(i1) : "v"(i0), "v"(i1));) }
    if (MODE == 24) { REP16(asm volatile("v_ashrrev_i32 %0, 31, %0\n v_and_b32 %1, %1, %0\n v_ashrrev_i32 %0, 31, %0\n v_and_b32 %1, %1, %0" : "+v"(i0), "+v"(i1));) }
    if (MODE == 25) { REP16(asm volatile("ds_read_b128 %0, %1\n ds_read_b128 %0, %1 offset:1024\n ds_read_b128 %0, %1 offset:2048\n ds_read_b128 %0, %1 offset:3072\n s_waitcnt lgkmcnt(0)" : "=v"(q4) : "v"(addr * 4));) }
    if (MODE == 26) { REP16(asm volatile("ds_write_b128 %1, %0\n ds_write_b128 %1, %0 offset:1024\n ds_write_b128 %1, %0 offset:2048\n ds_write_b128 %1, %0 offset:3072" : : "v"(q4), "v"(addr * 4));) }
    if (MODE == 27) { REP16(asm volatile("v_mad_i64_i32 %0, vcc, %2, %3, %0\n v_mad_i64_i32 %1, vcc, %3, %2, %1\n v_mad_i64_i32 %0, vcc, %2, %3, %0\n v_mad_i64_i32 %1, vcc, %3, %2, %1" : "+v"(w0), "+v"(w1) : "v"(i0), "v"(i1) : "vcc");) }
    if (MODE == 28) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %2\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %0, %0, %2\n v_mul_lo_u32 %1, %1, %2" : "+v"(i0), "+v"(i1) : "v"(i0));) }
    if (MODE == 29) { REP16(asm volatile("v_mul_hi_i32 %0, %0, %2\n v_mul_hi_i32 %1, %1, %2\n v_mul_hi_i32 %0, %0, %2\n v_mul_hi_i32 %1, %1, %2" : "+v"(i0), "+v"(i1) : "v"(i0));) }
    if (MODE == 30) { REP16(asm volatile("v_mul_i32_i24 %0, %0, %2\n v_mul_i32_i24 %1, %1, %2\n v_mul_i32_i24 %0, %0, %2\n v_mul_i32_i24 %1, %1, %2" : "+v"(i0), "+v"(i1) : "v"(i0));) }
    if (MODE == 31) { REP16(asm volatile("v_mad_i32_i24 %0, %0, %2, %1\n v_mad_i32_i24 %1, %1, %2, %0\n v_mad_i32_i24 %0, %0, %2, %1\n v_mad_i32_i24 %1, %1, %2, %0" : "+v"(i0), "+v"(i1) : "v"(i0));) }
    if (MODE == 32) { REP16(asm volatile("v_ashrrev_i64 %0, 30, %0\n v_ashrrev_i64 %1, 30, %1\n v_ashrrev_i64 %0, 30, %0\n v_ashrrev_i64 %1, 30, %1" : "+v"(w0), "+v"(w1));) }
    if (MODE == 33) { REP16(asm volatile("v_cmp_lt_i64 vcc, %0, %1\n v_cmp_lt_i64 vcc, %1, %0\n v_cmp_lt_i64 vcc, %0, %1\n v_cmp_lt_i64 vcc, %1, %0" : : "v"(w0), "v"(w1) : "vcc");) }
    if (MODE == 34) { REP16(asm volatile("v_mul_hi_i32_i24 %0, %0, %2\n v_mul_hi_i32_i24 %1, %1, %2\n v_mul_hi_i32_i24 %0, %0, %2\n v_mul_hi_i32_i24 %1, %1, %2" : "+v"(i0), "+v"(i1) : "v"(i0));) }
    if (MODE == 35) { REP16(asm volatile("v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %2, vcc\n v_add_co_u32 %0, vcc, %0, %2\n v_addc_co_u32 %1, vcc, %1, %2, vcc" : "+v"(i0), "+v"(i1) : "v"(i0) : "vcc");) }
    if (MODE == 36) { REP16(asm volatile("v_med3_i32 %0, %0, %2, %3\n v_med3_i32 %1, %1, %2, %3\n v_med3_i32 %0, %0, %2, %3\n v_med3_i32 %1, %1, %2, %3" : "+v"(i0), "+v"(i1) : "v"(i0), "v"(i1));) }
    if (MODE == 15) { REP16(asm volatile("v_sub_f32 %0, %0, %2\n v_sub_f32 %1, %1, %2\n v_sub_f32 %0, %0, %2\n v_sub_f32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(k));) }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
  out[blockIdx.x * 256 + threadIdx.x] = (float)(w0 + w1) + q4.x + q4.w + a + b + c + d + pa.x + pa.y + pb.x + pb.y + (float)i0 + (float)i1 + lds[(threadIdx.x * 7) & 4095];
}

template <int MODE>
static void run(const char *name, int per_rep, float *out, unsigned long long *clk, int cus) {
  const int iters = 1024;
  for (int wps = 1; wps <= 4; wps *= 2) {
    const int blocks = cus * wps;
    double best = 1e30, mhz = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipLaunchKernelGGL(spin<MODE>, dim3(blocks), dim3(256), 0, 0, iters, out, clk);
      hipDeviceSynchronize();
      unsigned long long h[2 * 4096];
      hipMemcpy(h, clk, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
      double cyc = 0, rt = 0;
      for (int i = 0; i < blocks; ++i) { cyc += (double)h[2 * i]; rt += (double)h[2 * i + 1]; }
      const double per = cyc / blocks / ((double)iters * 16.0 * per_rep);
      if (per < best) { best = per; mhz = cyc / rt * 100.0; }
    }
    printf("%-44s %d wave(s)/SIMD: %6.2f cycles per instruction per wave  (%5.2f per SIMD)   clock %4.0f MHz\n", name, wps, best, best / wps, mhz);
  }
}

int main() {
  float *out; unsigned long long *clk;
  (void)hipMalloc(&out, 4096 * 256 * sizeof(float));
  (void)hipMalloc(&clk, 2 * 4096 * sizeof(unsigned long long));
  int cus = 256;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  run<0>("v_add_f32 (two chains)", 4, out, clk, cus);
  run<15>("v_sub_f32", 4, out, clk, cus);
  run<1>("v_mul_f32", 4, out, clk, cus);
  run<2>("v_pk_mul_f32", 4, out, clk, cus);
  run<3>("v_cndmask_b32 (vcc)", 4, out, clk, cus);
  run<4>("v_cmp_ge_f32 -> vcc", 4, out, clk, cus);
  run<5>("v_cvt_i32_f32", 4, out, clk, cus);
  run<6>("v_lshl_add_u32", 4, out, clk, cus);
  run<7>("v_permlane32_swap_b32 + s_nop 0 (per pair)", 4, out, clk, cus);
  run<8>("s_nop 1 + v_add_f32_dpp row_ror:8 (per pair)", 4, out, clk, cus);
  run<9>("v_mov_b32", 4, out, clk, cus);
  run<10>("s_nop 0", 4, out, clk, cus);
  run<11>("ds_read_b32 x4 + s_waitcnt (per read)", 4, out, clk, cus);
  run<12>("ds_write_b32", 4, out, clk, cus);
  run<13>("v_add_f32 / s_add_u32 alternating (per instr)", 4, out, clk, cus);
  run<14>("add, mul, sub, add: dependent mix", 4, out, clk, cus);
  run<16>("v_cndmask_b32_e64 (mask in s[20:21])", 4, out, clk, cus);
  run<22>("v_cndmask_b32 (vcc), no self-dependence", 4, out, clk, cus);
  run<18>("v_cmp_ge_f32 + v_cndmask_b32 (per instr)", 4, out, clk, cus);
  run<17>("v_min_u32", 4, out, clk, cus);
  run<19>("v_min_f32 / v_max_f32", 4, out, clk, cus);
  run<20>("v_med3_f32", 4, out, clk, cus);
  run<21>("v_fract_f32", 4, out, clk, cus);
  run<23>("v_bfi_b32", 4, out, clk, cus);
  run<24>("v_ashrrev_i32 / v_and_b32", 4, out, clk, cus);
  run<25>("ds_read_b128 x4 + s_waitcnt (per read)", 4, out, clk, cus);
  run<26>("ds_write_b128", 4, out, clk, cus);
  run<27>("v_mad_i64_i32", 4, out, clk, cus);
  run<28>("v_mul_lo_u32", 4, out, clk, cus);
  run<29>("v_mul_hi_i32", 4, out, clk, cus);
  run<30>("v_mul_i32_i24", 4, out, clk, cus);
  run<34>("v_mul_hi_i32_i24", 4, out, clk, cus);
  run<31>("v_mad_i32_i24", 4, out, clk, cus);
  run<32>("v_ashrrev_i64", 4, out, clk, cus);
  run<33>("v_cmp_lt_i64 -> vcc", 4, out, clk, cus);
  run<35>("v_add_co_u32 / v_addc_co_u32 (per instr)", 4, out, clk, cus);
  run<36>("v_med3_i32", 4, out, clk, cus);
  return 0;
}
