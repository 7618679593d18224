// skred_fx_kernels.hip -- the FIXED-POINT render path on gfx950 (definition: oracle/cpu_ref_fxpt.c,
// include/skred_amd_fxpt.h).  Same mapping as the float path: one lane per voice, the frame loop
// inside the kernel with phase / smoother state in registers, the int16 LUT pool staged in LDS,
// integer DPP wave sum, wave sums combined as int64 in LDS every 64 frames, per-workgroup int64
// partials, fixed-order reduction.  Every operation is integer, so voices AND the mix are
// bit-exact against the CPU definition whatever the order of the additions.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "skred_fx_layout.h"
#include "skred_kernel_common.hpp"   /* sk_arrive_last: the arrival tickets of the in-kernel mix-down */

// ---------------------------------------------------------------- the block's mix-down + master stage (int64 rows)
//
// The float path's scheme (skred_kernel_common.hpp: sk_finish_block) on integer rows.  The master stage is DEFINED in
// include/skred_amd_fxpt.h ("master"): g (Q31, int64) += ((target - g) * k_q15) >> 15 per frame, out = (mix * (g >> 16)) >> 15.
typedef long long skx_i64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void skx_store_through(long long *p, long long v) {
  __hip_atomic_store((sk_gu64 *)p, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// sum of n rows (step apart) for one (L,R) column, ascending, 8 loads in flight
__device__ __forceinline__ skx_i64x2 skx_add_rows(const skx_i64x2 *rows, size_t ncol, int c, int first, int n, int step) {
  skx_i64x2 acc = rows[(size_t)first * ncol + c];
  int i = 1;
  for (; i + 8 <= n; i += 8) {
    skx_i64x2 t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = rows[(size_t)(first + (i + k) * step) * ncol + c];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += t[k];
  }
  for (; i < n; ++i) acc += rows[(size_t)(first + i * step) * ncol + c];
  return acc;
}
// every workgroup of the render kernel ends here (bid < 0: the gain workgroup); the row was published write-through
__device__ __forceinline__ void skx_finish_block(const skx_args_t &a, int bid, int tid, int nthreads, int *flag_lds) {
  const size_t ncol = (size_t)a.num_frames;
  const bool two_level = a.n_rows > SKX_FINISH_FLAT_MAX;
  const int n_last = two_level ? SKX_FINISH_SLABS : a.n_rows;
  if (bid < 0) {
    if (tid == 0) {                                       // the master gain of every frame: a serial recurrence
      long long g = a.gain_state[0];
      for (int i = 0; i < a.num_frames; ++i) {
        g += ((a.master_target_q31 - g) * (long long)a.master_k_q15) >> 15;
        __hip_atomic_store((sk_gu32 *)(a.gains + i), (uint32_t)(int32_t)(g >> 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      a.gain_commit[0] = g;
    }
  } else if (two_level) {
    const int slab = bid % SKX_FINISH_SLABS;
    const int members = (a.n_rows - slab + SKX_FINISH_SLABS - 1) / SKX_FINISH_SLABS;
    if (!sk_arrive_last(a.tickets + slab, (uint32_t)members, tid, flag_lds)) return;
    long long *dst = a.slab_rows + (size_t)slab * ncol * 2;
    for (int c = tid; c < (int)ncol; c += nthreads) {
      const skx_i64x2 s = skx_add_rows(reinterpret_cast<const skx_i64x2 *>(a.partial), ncol, c, slab, members, SKX_FINISH_SLABS);
      skx_store_through(dst + 2 * c, s.x);
      skx_store_through(dst + 2 * c + 1, s.y);
    }
  }
  if (!sk_arrive_last(a.tickets + SKX_FINISH_SLABS, (uint32_t)n_last + 1u, tid, flag_lds)) return;
  const skx_i64x2 *rows = reinterpret_cast<const skx_i64x2 *>(two_level ? a.slab_rows : a.partial);
  for (int c = tid; c < (int)ncol; c += nthreads) {
    const skx_i64x2 s = skx_add_rows(rows, ncol, c, 0, n_last, 1);
    if (a.sum_out) { a.sum_out[2 * c] = s.x; a.sum_out[2 * c + 1] = s.y; }
    if (a.mix_out) {
      const long long g15 = (long long)a.gains[c];
      a.mix_out[2 * c] = (s.x * g15) >> 15;
      a.mix_out[2 * c + 1] = (s.y * g15) >> 15;
    }
  }
}

// integer wave sum of two values into lane 63 (the float kernels fold L/R first: skred_kernel_common.hpp)
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

// a * b of the definition (int32, wrapping).  NARROW: the caller has proved that both operands fit 24 bits signed, so
// v_mul_i32_i24 (full rate; the 32-bit v_mul_lo_u32 is quarter rate) returns the same low 32 bits
template <bool NARROW>
__device__ __forceinline__ int fxmul(int a, int b) { return NARROW ? __mul24(a, b) : a * b; }
// (a * b) >> 15 where the definition forms the product in 64 bits (amp * e, s * gain: oracle/cpu_ref_fxpt.c:67,74).
// NARROW: both operands fit 24 bits and the product fits 32, so the 24-bit multiply and a 32-bit shift give the same
__device__ __forceinline__ int fxmul64_s15(int a, int b) { return (int)(((long long)a * (long long)b) >> 15); }
template <bool NARROW>
__device__ __forceinline__ int fxmul_s15(int a, int b) { return NARROW ? (__mul24(a, b) >> 15) : fxmul64_s15(a, b); }

__device__ __forceinline__ bool fits24(int x) { return x >= -(1 << 23) && x < (1 << 23); }

// The definition's biquad (include/skred_amd_fxpt.h: "biquad"): Q2.30 coefficients x Q12 delay line, int64 accumulation
// (v_mad_i64_i32), round to nearest, saturate to +-2^29, back to sample units with saturation to int16.  Returns the
// filtered sample and the new newest entries x0 / y0.
struct FxFilt { int b0, b1, b2, a1, a2; };
__device__ __forceinline__ int fx_biquad(const FxFilt &f, int s, int x1, int x2, int y1, int y2, int &x0, int &y0) {
  x0 = s * 4096;                                           // |s| <= 32767: exact, also for negative s
  long long acc = (long long)f.b0 * x0;
  acc += (long long)f.b1 * x1;
  acc += (long long)f.b2 * x2;
  acc -= (long long)f.a1 * y1;
  acc -= (long long)f.a2 * y2;
  long long y = (acc + (1ll << 29)) >> 30;
  y = y < -(1ll << 29) ? -(1ll << 29) : y;
  y = y > (1ll << 29) - 1 ? (1ll << 29) - 1 : y;
  y0 = (int)y;
  const int o = y0 >> 12;
  return o < -32768 ? -32768 : (o > 32767 ? 32767 : o);
}

#define SKX_WAVE_SYNC()                                     \
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
  __builtin_amdgcn_wave_barrier();                          \
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#define SKX_XT 68                  /* ints per tile row: 64 + 4 keeps the 16-byte reads aligned and spreads the rows over the banks */
#define SKX_TILE (8 * SKX_XT)      /* ints per wave: the reduction tile [8 frames][SKX_XT] */

// The pool as a voice reads it: an int16 copy in LDS (pools up to 48 KB; wave-uniform `tm` 1) or gathered from L2 / HBM (0).  The
// LDS pointer carries its address space: a pointer selected between LDS and global memory compiles to FLAT loads.
typedef __attribute__((address_space(3))) const int16_t fx_lds_i16;
struct FxTab { const int16_t *g; fx_lds_i16 *l16; };
__device__ __forceinline__ int fx_tab(const FxTab &t, const int tm, uint32_t idx) { return tm ? (int)t.l16[idx] : (int)t.g[idx]; }

// (L, R) of a lane folded into one register by a lane-half swap (lanes 0..31: l[i] + l[i+32]; lanes 32..63: r[i-32] + r[i]) and
// parked in a 4-byte tile row -- one ds_write_b32 per frame --; lane (f = lane & 7, seg = lane >> 3) then adds the 8 values of segment
// seg of frame f (two ds_read_b128), segments 0..3 (L) and 4..7 (R) meet across lanes (a DPP add inside the 16-lane row, a
// v_permlane16_swap across the row pair), lanes 0..7 / 32..39 store the frame's L / R total: the float kernels' scheme
// (skred_render_fast.hip: SK_FAST_TILE_REDUCE) on integers, where any order of the additions is exact.
__device__ __forceinline__ int fx_fold_lr(int l, int r) {
  const auto p = __builtin_amdgcn_permlane32_swap((unsigned)l, (unsigned)r, false, false);
  return (int)(p[0] + p[1]);
}
__device__ __forceinline__ int fx_row_ror8_add(int x) {
  int y;
  asm volatile("s_nop 1\n\tv_add_u32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "=v"(y) : "v"(x));
  return y;
}
__device__ __forceinline__ int fx_row_pair_add(int x) {       // rows 0+1 and rows 2+3, lane by lane
  const auto p = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);
  return (int)(p[0] + p[1]);
}
#define SKX_TILE_LOAD(TA, TB) { SKX_WAVE_SYNC() TA = tsrc[0]; TB = tsrc[1]; SKX_WAVE_SYNC() }
#define SKX_TILE_FINISH(TA, TB, ROW)                                                                 \
  {                                                                                                  \
    int t_ = ((TA.x + TA.y) + (TA.z + TA.w)) + ((TB.x + TB.y) + (TB.z + TB.w));                      \
    t_ = fx_row_pair_add(fx_row_ror8_add(t_));   /* segments 0..3 -> lanes 0..7 (L), 4..7 -> lanes 32..39 (R) */ \
    if ((lane & 24) == 0) reinterpret_cast<int *>(&(ROW)[lane & 7])[lane >> 5] = t_;                 \
  }

// Launch-constant part of a voice, as the frame code wants it
struct FxVoice {
  uint32_t inc;            // 0 for a skipped voice: its phase stays frozen
  int L;
  uint32_t mask;
  int pan_l, pan_r, k;     // k = 0 for a skipped voice: its smoother stays frozen
  bool smooth, silent;     // silent: skipped or muted -> (0, 0) out
  bool filt;               // runs the biquad
};

// Eight frames of a wave whose envelope levels are constant over the chunk (`target` = (amp * e) >> 15 per lane), the
// definition's arithmetic spelled out: any pool form, full-width products when !NARROW, the delay line's clamp applied.
// (The lean blocks below take the common case; what reaches this one is rare.)
template <bool STEMS, bool NARROW, bool INTERP, bool STALL, bool FILTER>
__device__ __forceinline__ void fx_block(const skx_args_t &a, const FxVoice &vc, const FxFilt &ff, const FxTab &tab, const int tm, uint32_t &phase,
                                         int &sg, int &sample, int &x1, int &x2, int &y1, int &y2, const int target, int *xt,
                                         int2 *wsum_row, const int lane, const int v, const int frame0) {
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    phase += vc.inc;
    const uint32_t idx = phase >> (32 - vc.L);
    int s = fx_tab(tab, tm, idx);
    if (INTERP) {
      const int nxt = fx_tab(tab, tm, (idx + 1) & vc.mask);
      const int frac = (int)((uint32_t)(phase << vc.L) >> 17);
      s = s + (__mul24(nxt - s, frac) >> 15);            // |nxt - s| < 2^17, frac < 2^15: always within 24 bits
    }
    if (FILTER) {                                        // (bank-wide: some voice is filtered; per lane: vc.filt)
      int x0, y0;
      const int fs = fx_biquad(ff, s, x1, x2, y1, y2, x0, y0);
      if (vc.filt) { x2 = x1; x1 = x0; y2 = y1; y1 = y0; s = fs; }
    }
    int gain = target;
    if (vc.smooth) {
      if (!STALL) sg += fxmul<NARROW>(target - sg, vc.k) >> 15;
      gain = sg;
    }
    const int smp = fxmul_s15<NARROW>(s, gain);
    sample = smp;
    int l = fxmul<NARROW>(smp, vc.pan_l) >> 15;
    int r = fxmul<NARROW>(smp, vc.pan_r) >> 15;
    l = vc.silent ? 0 : l;
    r = vc.silent ? 0 : r;
    if (STEMS) {
      if (v < a.n_voices)
        reinterpret_cast<int2 *>(a.stems)[(size_t)(frame0 + q) * (size_t)a.n_voices + (size_t)v] = make_int2(l, r);
    }
    xt[q * SKX_XT + lane] = fx_fold_lr(l, r);
  }
  const int4 *tsrc = reinterpret_cast<const int4 *>(xt + (lane & 7) * SKX_XT + (lane >> 3) * 8);
  int4 ta, tb;
  SKX_TILE_LOAD(ta, tb)
  SKX_TILE_FINISH(ta, tb, wsum_row)                      // |per-voice| < 2^17, 64 of them: fits int32
  SKX_WAVE_SYNC()
}

// ---------------------------------------------------------------- the lean steady blocks (round 4)
//
// The same eight frames for the common case -- NARROW operands, the pool in LDS -- written the way the float kernels'
// steady blocks are (skred_render_fast.hip: SK_FAST_LDS_CHUNK), every result bit-identical to fx_block / the definition:
//   * no per-lane branch around the biquad: a lane that does not filter runs the IDENTITY (b0 = 2^30, the rest 0:
//     (2^30 * (s << 12) + 2^29) >> 30 == s << 12 exactly, back to s) and gets its delay line back behind the block;
//     the feedback coefficients are negated once, so the five products ADD into one v_mad_i64_i32 chain that starts
//     from the rounding constant 2^29;
//   * the delay-line saturation of the definition (clamp to +-2^29) is CHECKED, not applied: (acc >> 30) lies inside the
//     clamp's range iff the accumulator's high word lies in [-2^27, 2^27), a max and a min per frame pair; a block in which some
//     lane leaves that range is rolled back and handed to fx_block (rare: a resonance riding the rail);
//   * a silent lane carries pan gains of 0 instead of two selects per frame;
//   * the tile of block b - 1 is added up while block b runs.

// a * b + c, int32 x int32 + int64: v_mad_i64_i32, spelled out (left to itself hipcc widens the loop-carried delay line to
// 64 bits and multiplies 64 x 64: v_mad_u64_u32 + two v_mul_lo_u32 + the adds, per product)
__device__ __forceinline__ long long fx_mad64(int a, int b, long long c) {
  long long d;
  unsigned long long carry;                                  // (the instruction's carry-out pair: unused)
  asm("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"(a), "v"(b), "v"(c));
  return d;
}

// Up to `nblk` blocks of 8 frames; returns how many were rendered (fewer: the next one saturates its delay line and is
// the caller's, state as the last rendered block left it).  wrow = the wave's row of wsum for this chunk.
template <bool STEMS, bool INTERP, bool STALL, bool FILTER>
__device__ __forceinline__ int fx_chunk_lean(const skx_args_t &a, const FxVoice &vc, const FxFilt &ff, const FxTab &tab, uint32_t &phase,
                                             int &sg, int &sample, int &x1, int &x2, int &y1, int &y2, const int target, int *xt,
                                             int2 *wrow, const int lane, const int v, const int frame0, const int nblk) {
  const int pl = vc.silent ? 0 : vc.pan_l, pr = vc.silent ? 0 : vc.pan_r;
  const int kk = vc.smooth ? vc.k : 0;                       // a lane without the smoother leaves its state alone
  const int fb0 = vc.filt ? ff.b0 : (1 << 30), fb1 = vc.filt ? ff.b1 : 0, fb2 = vc.filt ? ff.b2 : 0;
  const int na1 = vc.filt ? -ff.a1 : 0, na2 = vc.filt ? -ff.a2 : 0;
  const uint32_t sh_idx = 32u - (uint32_t)vc.L, sh_frac = 17u - (uint32_t)vc.L;
  const int4 *tsrc = reinterpret_cast<const int4 *>(xt + (lane & 7) * SKX_XT + (lane >> 3) * 8);
  int4 ta = make_int4(0, 0, 0, 0), tb = ta;
  int pend = -1, b = 0;
  const int gain_c = vc.smooth ? sg : target;                // (STALL: the smoother rests, the gain is one constant)
  const long long half_ = 1ll << 29;                         // the rounding constant the accumulation starts from
#define SKX_LEAN_FRAME(Q, XN, XO, YN, YO)                                                            \
  {                                                                                                  \
    phase += vc.inc;                                                                                 \
    const uint32_t idx_ = phase >> sh_idx;                                                           \
    int s_ = tab.l16[idx_];                                                                          \
    if (INTERP) {                                                                                    \
      const int nxt_ = tab.l16[(idx_ + 1u) & vc.mask];                                               \
      const int frac_ = (int)__builtin_amdgcn_ubfe(phase, sh_frac, 15u);   /* == (phase << L) >> 17 */ \
      s_ = s_ + (__mul24(nxt_ - s_, frac_) >> 15);                                                   \
    }                                                                                                \
    if (FILTER) {                                                                                    \
      const int x0_ = s_ * 4096;                                                                     \
      long long acc_ = fx_mad64(fb0, x0_, half_);                                                    \
      acc_ = fx_mad64(fb1, XN, acc_);                                                                \
      acc_ = fx_mad64(fb2, XO, acc_);                                                                \
      acc_ = fx_mad64(na1, YN, acc_);                                                                \
      acc_ = fx_mad64(na2, YO, acc_);                                                                \
      const int hi_ = (int)(acc_ >> 32);                                                             \
      const int y0_ = (int)(acc_ >> 30);              /* the definition's y0 while it does not clamp */ \
      hi_max = max(hi_max, hi_); hi_min = min(hi_min, hi_);                                          \
      XO = x0_; YO = y0_;                                                                            \
      const int o_ = y0_ >> 12;                                                                      \
      s_ = o_ < -32768 ? -32768 : (o_ > 32767 ? 32767 : o_);                                         \
    }                                                                                                \
    int gain_ = gain_c;                                                                              \
    if (!STALL) { sg += __mul24(target - sg, kk) >> 15; gain_ = vc.smooth ? sg : target; }           \
    smp = __mul24(s_, gain_) >> 15;                                                                  \
    const int l_ = __mul24(smp, pl) >> 15, r_ = __mul24(smp, pr) >> 15;                              \
    if (STEMS) {                                                                                     \
      if (v < a.n_voices)                                                                            \
        reinterpret_cast<int2 *>(a.stems)[(size_t)(frame0 + b * 8 + (Q)) * (size_t)a.n_voices + (size_t)v] = make_int2(l_, r_); \
    }                                                                                                \
    xt[(Q) * SKX_XT + lane] = fx_fold_lr(l_, r_);                                                    \
  }
  for (; b < nblk; ++b) {
    const uint32_t s_phase = phase;
    const int s_sg = sg, s_x1 = x1, s_x2 = x2, s_y1 = y1, s_y2 = y2;
    int hi_max = 0, hi_min = 0;                              // (v_max3_i32 / v_min3_i32: one instruction per frame for the pair)
    int smp = sample;
    if (pend >= 0) SKX_TILE_LOAD(ta, tb)
#pragma unroll
    for (int q = 0; q < 8; q += 2) {
      SKX_LEAN_FRAME(q, x1, x2, y1, y2)
      SKX_LEAN_FRAME(q + 1, x2, x1, y2, y1)
    }
    if (pend >= 0) { SKX_TILE_FINISH(ta, tb, wrow + pend) pend = -1; }
    if (FILTER) {
      if (__any(hi_max >= (1 << 27) || hi_min < -(1 << 27))) {   // some delay line left +-2^29: this block is fx_block's
        phase = s_phase; sg = s_sg; x1 = s_x1; x2 = s_x2; y1 = s_y1; y2 = s_y2;
        break;
      }
      if (!vc.filt) { x1 = s_x1; x2 = s_x2; y1 = s_y1; y2 = s_y2; }
    }
    sample = smp;
    pend = b * 8;
  }
  if (pend >= 0) { SKX_TILE_LOAD(ta, tb) SKX_TILE_FINISH(ta, tb, wrow + pend) }
  SKX_WAVE_SYNC()
#undef SKX_LEAN_FRAME
  return b;
}

template <bool STEMS>
__global__ __launch_bounds__(SKX_GROUP, 4) void sk_fx_render_kernel(const skx_args_t a) {
  extern __shared__ int16_t lut_lds[];                 // [the pool] | int2 wsum[4][SKX_CHUNK] | int tile[4][SKX_TILE]
  int2 *wsum = reinterpret_cast<int2 *>(reinterpret_cast<char *>(lut_lds) + a.lds_bytes_tables);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int *xt = reinterpret_cast<int *>(wsum + 4 * SKX_CHUNK) + wave * SKX_TILE;
  const bool lut_in_lds = a.lds_bytes_tables > 0;
  const int tm = lut_in_lds ? 1 : 0;                   // the pool's form (FxTab)
  const int bid = (int)blockIdx.x - 1;                 // row of the partial mix; -1: the gain workgroup
  if (bid < 0) { skx_finish_block(a, bid, tid, SKX_GROUP, reinterpret_cast<int *>(lut_lds)); return; }
  if (lut_in_lds) {
    const int n4 = a.lds_bytes_tables >> 4;
    const uint4 *src = reinterpret_cast<const uint4 *>(a.tables);
    uint4 *dst = reinterpret_cast<uint4 *>(lut_lds);
    sk_stage_tables<SKX_GROUP>(src, dst, n4, tid);
    __syncthreads();
  }
  const size_t part_base = (size_t)bid * (size_t)a.num_frames * 2;
  bool first_pass = true;

  for (int g = bid; g < a.n_groups; g += a.n_rows) {
    const bool publish = g + a.n_rows >= a.n_groups;   // the pass that completes this workgroup's row: its values leave write-through
    const int v = g * SKX_GROUP + tid;
    const uint4 p0 = *reinterpret_cast<const uint4 *>(&a.ro[SKX_OSC][v]);
    const uint4 p1 = *reinterpret_cast<const uint4 *>(&a.ro[SKX_GAIN][v]);
    const uint4 p2 = *reinterpret_cast<const uint4 *>(&a.ro[SKX_ENV][v]);
    const uint4 p3 = *reinterpret_cast<const uint4 *>(&a.ro[SKX_RECIP][v]);
    const uint4 p4 = *reinterpret_cast<const uint4 *>(&a.ro[SKX_TIME][v]);
    const uint4 st = *reinterpret_cast<const uint4 *>(&a.rw[0][v]);
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
    bool finished = (st.w & 2u) != 0;
    const bool one_shot = (flags & SKXF_ONE_SHOT) != 0;
    FxFilt ff = {0, 0, 0, 0, 0};
    int x1 = 0, x2 = 0, y1 = 0, y2 = 0;
    const bool filt = (flags & SKXF_FILTER) != 0;
    if (a.any_filter) {
      const uint4 f0 = *reinterpret_cast<const uint4 *>(&a.ro[SKX_FILT][v]);
      const uint4 f1 = *reinterpret_cast<const uint4 *>(&a.ro[SKX_FILT2][v]);
      const uint4 fs = *reinterpret_cast<const uint4 *>(&a.rw[1][v]);
      ff.b0 = (int)f0.x; ff.b1 = (int)f0.y; ff.b2 = (int)f0.z; ff.a1 = (int)f0.w; ff.a2 = (int)f1.x;
      x1 = (int)fs.x; x2 = (int)fs.y; y1 = (int)fs.z; y2 = (int)fs.w;
    }
    const bool dead = amp == 0 || (flags & SKXF_INERT) || finished;
    // (a skipped lane runs the block code on inert numbers: entries 0..1 of the pool, whatever its own table fields say)
    const int tbase = dead ? 0 : toff;
    FxTab tab;
    tab.g = a.tables + tbase;
    tab.l16 = (fx_lds_i16 *)lut_lds + tbase;
    const uint32_t mask = (1u << L) - 1u;
    const bool uses_env = (flags & SKXF_USE_ENV) != 0;

    FxVoice vc;
    vc.inc = dead ? 0u : inc; vc.L = dead ? 1 : L; vc.mask = dead ? 1u : mask;
    vc.pan_l = pan_l; vc.pan_r = pan_r; vc.k = dead ? 0 : k;
    vc.smooth = (flags & SKXF_SMOOTH) != 0;
    vc.silent = dead || (flags & SKXF_MUTED);
    vc.filt = filt && !dead;
    // NARROW blocks (24-bit multiplies): pan gains within 24 bits, 0 <= k <= 32768 (the smoother state then stays between
    // itself and its target), and |smoother state|, |target| <= 65535, so that s * gain fits 32 bits like the int16
    // sample times a Q16 gain it is meant to be; anything else takes the blocks with the definition's full-width products
    const bool narrow = __all(dead || (fits24(pan_l) && fits24(pan_r) && k >= 0 && k <= 32768 && sg >= -65535 && sg <= 65535));
    // the lean blocks also want the pool in LDS, table sizes the definition allows (the fraction is a bit field of the phase) and
    // feedback coefficients that can be negated
    const bool lean_ok = lut_in_lds && __all(dead || (L >= 1 && L <= 15 && (!filt || (ff.a1 != INT32_MIN && ff.a2 != INT32_MIN))));

    for (int c0 = 0; c0 < a.num_frames; c0 += SKX_CHUNK) {
      const int cn = min(SKX_CHUNK, a.num_frames - c0);
      // steady: the envelope level of every live lane is one constant over this chunk -- no envelope, an inactive one
      // (it stays inactive), or a held note past its decay (t only grows; note-off arrives between launches)
      const uint32_t t_first = sat32(a.count0 + (uint64_t)c0 + 1 - t_start);
      // (a wave with a live one-shot voice checks for its end frame by frame: no blocks)
      const bool steady = __all(dead || !uses_env || !active || (t_release == 0 && (uint64_t)t_first >= AD)) &&
                          !__any(one_shot && !dead && !finished);
      int j = 0;
      if (steady && cn >= 8) {
        if (finished && !dead) {       // a one-shot that ended earlier in this launch: from here on a skipped lane (state frozen)
          vc.inc = 0u; vc.k = 0; vc.silent = true; vc.filt = false;
        }
        const int lvl = active ? S : 0;
        const int e = uses_env ? (lvl * vel) >> 15 : 32768;
        const int target = fxmul64_s15(amp, e);
        const bool stalled = __all(dead || !vc.smooth || (((target - sg) * vc.k) >> 15) == 0);
        const bool narrow_c = narrow && __all(dead || (target >= -65535 && target <= 65535));
        int2 *row = wsum + wave * SKX_CHUNK;
        if (narrow_c && lean_ok) {       // the lean blocks (fx_chunk_lean); what they leave -- a block that rides the rail -- follows below
          const int nblk = cn >> 3;
          int done;
#define SKX_LEAN(INTERP_, STALL_, FILTER_)                                                                           \
  done = fx_chunk_lean<STEMS, INTERP_, STALL_, FILTER_>(a, vc, ff, tab, phase, sg, sample, x1, x2, y1, y2, target, xt, row, lane, v, c0, nblk);
          if (a.any_filter) {
            if (a.interp) { if (stalled) SKX_LEAN(true, true, true) else SKX_LEAN(true, false, true) }
            else          { if (stalled) SKX_LEAN(false, true, true) else SKX_LEAN(false, false, true) }
          } else {
            if (a.interp) { if (stalled) SKX_LEAN(true, true, false) else SKX_LEAN(true, false, false) }
            else          { if (stalled) SKX_LEAN(false, true, false) else SKX_LEAN(false, false, false) }
          }
#undef SKX_LEAN
          j = done << 3;
        }
#define SKX_BLOCKS(NARROW_, INTERP_, STALL_)                                                                       \
  if (a.any_filter) { for (; j + 8 <= cn; j += 8) fx_block<STEMS, NARROW_, INTERP_, STALL_, true>(a, vc, ff, tab, tm, phase, sg, sample, x1, x2, y1, y2, target, xt, row + j, lane, v, c0 + j); } \
  else { for (; j + 8 <= cn; j += 8) fx_block<STEMS, NARROW_, INTERP_, STALL_, false>(a, vc, ff, tab, tm, phase, sg, sample, x1, x2, y1, y2, target, xt, row + j, lane, v, c0 + j); }
        if (narrow_c) {
          if (a.interp) { if (stalled) SKX_BLOCKS(true, true, true) else SKX_BLOCKS(true, true, false) }
          else          { if (stalled) SKX_BLOCKS(true, false, true) else SKX_BLOCKS(true, false, false) }
        } else {
          if (a.interp) SKX_BLOCKS(false, true, false) else SKX_BLOCKS(false, false, false)
        }
#undef SKX_BLOCKS
        if (finished && j > 0) sample = 0;   // (the blocks leave smp in `sample`; a skipped voice's is 0)
      }
      for (; j < cn; ++j) {
        const uint64_t now = a.count0 + (uint64_t)(c0 + j) + 1;
        int l = 0, r = 0;
        if (dead || finished) {
          if (finished) sample = 0;                      // skipped from the frame after its last one on
        } else {
          uint32_t ph = phase + inc;
          bool ends = false;
          if (one_shot && ph < phase) { ph = 0xFFFFFFFFu; finished = true; ends = true; }   // the add carried: the cycle is over
          phase = ph;
          const uint32_t idx = phase >> (32 - L);
          int s = fx_tab(tab, tm, idx);
          if (a.interp) {
            const int nxt = ends ? s : fx_tab(tab, tm, (idx + 1) & mask);
            const int frac = (int)((uint32_t)(phase << L) >> 17);
            s = s + (((nxt - s) * frac) >> 15);
          }
          if (filt) {
            int x0, y0;
            s = fx_biquad(ff, s, x1, x2, y1, y2, x0, y0);
            x2 = x1; x1 = x0; y2 = y1; y1 = y0;
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
          int gain = fxmul64_s15(amp, e);
          if (flags & SKXF_SMOOTH) {
            sg += ((gain - sg) * k) >> 15;
            gain = sg;
          }
          sample = fxmul64_s15(s, gain);
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
        if (!first_pass) s += *p;
        if (publish) skx_store_through(p, s); else *p = s;
      }
      __syncthreads();
    }
    if (dead) sample = 0;
    uint4 o;
    o.x = phase; o.y = (uint32_t)sg; o.z = (uint32_t)sample; o.w = active | (finished ? 2u : 0u);
    *reinterpret_cast<uint4 *>(&a.rw[0][v]) = o;
    if (a.any_filter && filt) *reinterpret_cast<uint4 *>(&a.rw[1][v]) = make_uint4((uint32_t)x1, (uint32_t)x2, (uint32_t)y1, (uint32_t)y2);
    first_pass = false;
  }
  skx_finish_block(a, bid, tid, SKX_GROUP, reinterpret_cast<int *>(lut_lds));
}

// The master stage as a kernel of its own (multi-GPU form: on the root, after the int64 reduce): the gains of the block were
// walked by the render kernel's gain workgroup; scale, and commit the carried gain.
__global__ __launch_bounds__(256) void sk_fx_master_apply_kernel(const long long *__restrict__ sum, const int32_t *__restrict__ gains,
                                                                 long long *__restrict__ out, int num_frames,
                                                                 const long long *gain_pending, long long *gain_state) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < num_frames) {
    const long long g15 = (long long)gains[i];
    out[2 * i] = (sum[2 * i] * g15) >> 15;
    out[2 * i + 1] = (sum[2 * i + 1] * g15) >> 15;
  }
  if (i == 0) gain_state[0] = gain_pending[0];
}
extern "C" int skx_launch_master_apply(const long long *sum, const int32_t *gains, long long *out, int num_frames,
                                       const long long *gain_pending, long long *gain_state, hipStream_t stream) {
  hipLaunchKernelGGL(sk_fx_master_apply_kernel, dim3((unsigned)((num_frames + 255) / 256)), dim3(256), 0, stream, sum, gains, out, num_frames,
                     gain_pending, gain_state);
  return (int)hipGetLastError();
}

// note-on / note-off stamps on device-resident voices (include/skred_amd_fxpt.h: skred_fxbank_stamp): the integer image of
// amp_envelope_trigger / amp_envelope_release (synth.c:383-395)
__global__ __launch_bounds__(256) void sk_fx_stamp_kernel(const int32_t *__restrict__ ids, int n, int which, skx_plane_t *time_plane,
                                                          skx_plane_t *rw0, uint64_t now) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int v = ids[i];
  uint4 t = *reinterpret_cast<const uint4 *>(&time_plane[v]);
  uint32_t *flags = reinterpret_cast<uint32_t *>(&rw0[v]) + 3;
  uint32_t f = *flags;
  if (which & 1) { t.x = (uint32_t)now; t.y = (uint32_t)(now >> 32); t.z = 0; t.w = 0; f |= 1u; }     // trigger
  if ((which & 2) && (f & 1u)) { t.z = (uint32_t)now; t.w = (uint32_t)(now >> 32); }                 // release, if active
  *reinterpret_cast<uint4 *>(&time_plane[v]) = t;
  *flags = f;
}
extern "C" int skx_launch_stamp(const int32_t *d_ids, int n, int which, skx_plane_t *time_plane, skx_plane_t *rw0, uint64_t now, hipStream_t stream) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(sk_fx_stamp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_ids, n, which, time_plane, rw0, now);
  return (int)hipGetLastError();
}

extern "C" int skx_launch_render(const skx_args_t *args, int n_workgroups, hipStream_t stream) {
  const size_t lds = (size_t)args->lds_bytes_tables + (size_t)4 * SKX_CHUNK * sizeof(int2) + (size_t)4 * SKX_TILE * sizeof(int);
  dim3 grid((unsigned)n_workgroups + 1u), block(SKX_GROUP);   /* + the gain workgroup */
  if (args->stems) hipLaunchKernelGGL((sk_fx_render_kernel<true>), grid, block, lds, stream, *args);
  else             hipLaunchKernelGGL((sk_fx_render_kernel<false>), grid, block, lds, stream, *args);
  return (int)hipGetLastError();
}

