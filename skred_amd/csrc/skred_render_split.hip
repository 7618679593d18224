// skred_render_split.hip -- sk_render_split_kernel: one voice per lane, the frame split between two wavefronts.
#include "skred_fast_common.hpp"
#include "skred_launch.h"

// ---------------------------------------------------------------- why
//
// Small clean banks (BASELINE configs 1 and 2: up to one wavefront of sk_render_fast_kernel per SIMD) are bound by what ONE
// wavefront can issue: about one instruction every 5.4 cycles, whatever the instructions are (tools/issue_rate.hip), plus every
// LDS round trip it has to wait out alone.  The per-voice recurrences are serial in time, so the only way to give such a bank
// more issue slots is to cut the FRAME in two and give each half a wavefront of its own:
//
//   oscillator wave  (wavefronts 4..7 of the workgroup)   phase += inc, wrap, table gather (osc_next, synth.c:217-275), and the
//                                                         feed-forward half of the biquad, (b0*s + b1*x1) + b2*x2, which needs
//                                                         input samples only (mmf_process, synth.c:351-353)
//   post wave        (wavefronts 0..3, owns the voices)   the feedback half, (P - a1*y1) - a2*y2, gain / smoother, pan, the
//                                                         cross-lane sum and the mix-down (synth.c:354-364,580-612)
//
// Nothing in the oscillator half depends on anything behind it, so its wave runs AHEAD and hands 8-frame blocks to the post wave
// through a ring in LDS (SKS_RING slots of 8 frames x 64 lanes).  Both halves call functions of skred_fast_common.hpp that
// restate sk_render_fast_kernel's frame -- same products, same sums, same order -- so every voice renders to the same bits, and
// the wave / workgroup / block sums keep their order too: the mix is the same BYTES as sk_render_fast_kernel's.
//
// Synchronisation is by LDS words only, never by s_barrier inside a block: LDS executes a wave's accesses in order, so a
// producer's `data, then counter` and a consumer's `counter, then data` need no fence beyond keeping the compiler from
// reordering them (all of them DS instructions: see sks_ctrl_t).  Every wait is on a counter that the other side advances
// unconditionally (the oscillator wave produces every block of the launch, the post wave consumes every block), so every wait
// ends.  Neither wave waits where it does not have to: the post wave reads block b + 1 (counter first, then the data, which is
// then valid if the counter says so) BEFORE it renders block b; the oscillator wave issues the gathers of block b + 1 before
// it finishes block b, and looks at the consumer's counter only when its own copy no longer proves a free slot.
//
// Which waves take the split path is decided per pass by the OWNER (post) wave: tame geometry (see fast_advance<TAME>) and a
// constant envelope level on every lane for the whole launch.  Anything else -- an envelope in motion because the host's
// "nothing moves" hint was stale, a huge increment, stems -- is rendered by the owner wave alone, frame by frame, on the
// general path of skred_fast_common.hpp (its oscillator wave idles): slow, rare, and the same bits.  The host launches this
// kernel only while it believes nothing moves (skred_bank.c: render_block), so the hint decides speed, never samples.
//
// Where it pays (tools/ab_split.py, profiles/r04_*): banks of up to one 64-voice group per SIMD.  From two groups per SIMD on
// (the 2^17-voice shard of config 3) the SIMD's VALU is the bound, not the wave's issue rate, and the ring traffic and the
// polling make the split form slower than sk_render_fast_kernel there: the host does not pick it.

#ifndef SKS_RING
#define SKS_RING 4                         /* ring slots per pair (8 frames each) */
#endif
#ifndef SKS_SLEEP
#define SKS_SLEEP 1                        /* s_sleep argument of the polling loops (x 64 cycles) */
#endif
/* timing experiments only (tools/ab_build.sh; the outputs are then wrong): SKS_ABL_FREE_RUN -- the oscillator wave does not wait
 * for free slots; SKS_ABL_POST_ONLY -- the post wave does not wait for the oscillator wave, which leaves at once;
 * SKS_ABL_OSC_ONLY -- the post wave consumes the blocks without rendering them */
#define SKS_SLOT_FLOATS 512                /* [2 halves][64 lanes][4 frames] */
#define SKS_MAIL 10                        /* mailbox rows (owner -> oscillator wave, in the still empty ring): phase inc lo hi toff4 x1 x2 b0 b1 b2 */
#define SKS_PAIR_FLOATS (SKS_RING * SKS_SLOT_FLOATS + 8 * SK_XT + 3 * 64)   /* ring, the post wave's reduction tile, what comes back: phase x1 x2 */
#define SKS_CTRL_INTS 32
// control words: pair p at [4p]: go, produced, consumed, done; [16],[17]: chunk arrival counters; [18],[19]: chunks combined; [24]: finish flag
#define SKS_GO 0
#define SKS_PRODUCED 1
#define SKS_CONSUMED 2
#define SKS_DONE 3

#define SKS_COMPILER_FENCE() asm volatile("" ::: "memory")
/* -DSKS_STAMPS (a diagnostic build: tools/ab_build.sh): every wave leaves, in a.env_list (unused by this family), the shader
 * cycles of its pass loop, the cycles it spent waiting for the other wave, how often it had to, the cycles at the chunk ends, and
 * the 100 MHz real-time ticks of the loop (in-kernel clock = cycles / ticks x 100 MHz).  No stamp executes in the real kernel. */
#ifdef SKS_STAMPS
#define SKS_T() __builtin_amdgcn_s_memtime()
#define SKS_STAMP_DECL unsigned long long st_t0_ = 0, st_wait_ = 0, st_arr_ = 0, st_rt0_ = 0; int st_n_ = 0;
#define SKS_STAMP_BEGIN() { st_t0_ = SKS_T(); st_rt0_ = __builtin_amdgcn_s_memrealtime(); }
#define SKS_STAMP_END()                                                                                  \
  { const unsigned long long t1_ = SKS_T(), rt1_ = __builtin_amdgcn_s_memrealtime();                    \
    if (lane == 0) { int *d_ = a.env_list + ((bid * 8 + wave) * 8);                                      \
      d_[0] = (int)(t1_ - st_t0_); d_[1] = (int)st_wait_; d_[2] = st_n_; d_[3] = (int)st_arr_; d_[4] = (int)(rt1_ - st_rt0_); } }
#define SKS_WAIT_T0() const unsigned long long w0_ = SKS_T();
#define SKS_WAIT_T1() { st_wait_ += SKS_T() - w0_; ++st_n_; }
#define SKS_ARR_T0() const unsigned long long a0_ = SKS_T();
#define SKS_ARR_T1() { st_arr_ += SKS_T() - a0_; }
#else
#define SKS_STAMP_DECL
#define SKS_STAMP_BEGIN()
#define SKS_STAMP_END()
#define SKS_WAIT_T0()
#define SKS_WAIT_T1()
#define SKS_ARR_T0()
#define SKS_ARR_T1()
#endif
#define SKS_WAVE_SYNC()                                     \
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
  __builtin_amdgcn_wave_barrier();                          \
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

// The control words are accessed through LDS-address-space pointers ONLY: a generic pointer would make them FLAT instructions,
// which travel through the vector-memory path and are not ordered against the wave's DS instructions -- the whole protocol
// rests on that order.
typedef __attribute__((address_space(3))) int sks_lds_int;
typedef volatile sks_lds_int *sks_ctrl_t;

__device__ __forceinline__ int sks_peek(sks_ctrl_t p) { return __builtin_amdgcn_readfirstlane(*p); }
__device__ __forceinline__ void sks_wait_ge(sks_ctrl_t p, int need) {
  while (sks_peek(p) < need) __builtin_amdgcn_s_sleep(SKS_SLEEP);
  SKS_COMPILER_FENCE();
}
__device__ __forceinline__ void sks_post(sks_ctrl_t p, int v, int lane) {
  SKS_COMPILER_FENCE();
  if (lane == 0) *p = v;
  SKS_COMPILER_FENCE();
}

// ---- oscillator half: n (<= 8) frames of phase advance + table fetch; then the feed-forward half of the biquad over them
template <int INTERP, bool LOZ>
__device__ __forceinline__ void sks_osc8(FastRegs &r, const char *lds_tab, float (&s)[8], int n) {
  if (n >= 8) {
#pragma unroll
    for (int q = 0; q < 8; ++q) s[q] = fast_fetch<true, INTERP, true>(lds_tab, nullptr, r, fast_advance<true, false, LOZ>(r));
  } else {
#pragma unroll
    for (int q = 0; q < 8; ++q) { s[q] = 0.0f; if (q < n) s[q] = fast_fetch<true, INTERP, true>(lds_tab, nullptr, r, fast_advance<true, false, LOZ>(r)); }
  }
}
template <bool FILTER>
__device__ __forceinline__ void sks_ff8(float b0, const v2f &b12, const v2f &b21, v2f &xx, float (&s)[8], int n) {
  if (!FILTER) return;
  if (n >= 8) {
#pragma unroll
    for (int q = 0; q < 8; q += 2) {
      s[q] = fast_biquad_ff<true>(b0, b12, b21, s[q], xx);
      s[q + 1] = fast_biquad_ff<false>(b0, b12, b21, s[q + 1], xx);
    }
  } else {
#pragma unroll
    for (int q = 0; q < 8; q += 2) {
      if (q < n) s[q] = fast_biquad_ff<true>(b0, b12, b21, s[q], xx);
      if (q + 1 < n) s[q + 1] = fast_biquad_ff<false>(b0, b12, b21, s[q + 1], xx);
    }
    if (n & 1) xx = (v2f){xx.y, xx.x};                // an odd frame count (the launch's last block): the newest entry back into .x
  }
}

// ---- post half: eight frames -> the wave's reduction tile (the image of SK_FAST_POST8_ in skred_render_fast.hip)
template <bool FILTER, bool ENV, bool STALL, bool SEL>
__device__ __forceinline__ void sks_post8(FastRegs &r, const FastPk &pk, v2f &yy, const float (&p)[8], float *xt, int lane, bool silent) {
#pragma unroll
  for (int q = 0; q < 8; q += 2) {
    float s0 = FILTER ? fast_biquad_fb<true>(pk.a12, pk.a21, p[q], yy) : p[q];
    s0 = fast_gain_const<ENV, STALL>(r, s0);
    float s1 = FILTER ? fast_biquad_fb<false>(pk.a12, pk.a21, p[q + 1], yy) : p[q + 1];
    s1 = fast_gain_const<ENV, STALL>(r, s1);
    float f0, f1;
    if (FILTER && !SEL) {
      fast_pan_fold2(s0, s1, pk.pan.x, pk.pan.y, f0, f1);
    } else {
      const v2f lr0 = pk.pan * (v2f){s0, s0}, lr1 = pk.pan * (v2f){s1, s1};
      float l0 = lr0.x, r0 = lr0.y, l1 = lr1.x, r1 = lr1.y;
      if (SEL) { l0 = silent ? 0.0f : l0; r0 = silent ? 0.0f : r0; l1 = silent ? 0.0f : l1; r1 = silent ? 0.0f : r1; }   // a muted voice renders, but stays out of the mix
      f0 = fold_lr(l0, r0); f1 = fold_lr(l1, r1);
    }
    xt[q * SK_XT + lane] = f0;
    xt[(q + 1) * SK_XT + lane] = f1;
  }
}

// the tile of the block before: 8 frames x 64 folded lane values -> 8 (L, R) totals in the wave's chunk row
__device__ __forceinline__ void sks_tile_load(const float *xt, int lane, float4 &ta, float4 &tb) {
  SKS_WAVE_SYNC()
  const float4 *src = reinterpret_cast<const float4 *>(xt + (lane & 7) * SK_XT + (lane >> 3) * 8);
  ta = src[0]; tb = src[1];
  SKS_WAVE_SYNC()
}
__device__ __forceinline__ void sks_tile_finish(const float4 &ta, const float4 &tb, float2 *row, int j, int lane) {
  float t = ((((((ta.x + ta.y) + ta.z) + ta.w) + tb.x) + tb.y) + tb.z) + tb.w;
  t = row_pair_add(row_ror8_add(t));
  if ((lane & 24) == 0) reinterpret_cast<float *>(&row[j + (lane & 7)])[lane >> 5] = t;
}

struct SksChunk {            // what the wave that arrives last at a chunk needs to add the four rows up
  int c0, cn;                // first frame, frames
  bool first_pass, publish;
};

// The owner waves meet at the end of every 64-frame chunk WITHOUT a barrier: each adds to the chunk's arrival counter after its
// row of wsum is written; the one whose add comes last adds the four rows (wave order, as sk_render_fast_kernel does) into the
// workgroup's row of a.partial and publishes `combined`.  Rows alternate between two buffers, and a wave only starts writing
// a buffer again once the chunk that used it before has been combined.
template <int NP>
__device__ __forceinline__ void sks_arrive(const sk_render_args_t &a, sks_ctrl_t ctrl, const float2 *wsum_all, int cs, const SksChunk &ck,
                                           size_t part_base, int lane) {
  const int par = cs & 1;
  SKS_COMPILER_FENCE();
  int old = 0;
  if (lane == 0) old = __hip_atomic_fetch_add((sks_lds_int *)(ctrl + 16 + par), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ds_add_rtn_u32
  old = __builtin_amdgcn_readfirstlane(old);
  if (old != NP - 1) return;
  if (lane == 0) ctrl[16 + par] = 0;
  SKS_COMPILER_FENCE();
  const float *w = reinterpret_cast<const float *>(wsum_all + par * NP * SK_CHUNK);
  for (int i = lane; i < 2 * ck.cn; i += 64) {
    float s = w[0 * 2 * SK_CHUNK + i];
#pragma unroll
    for (int k = 1; k < NP; ++k) s += w[k * 2 * SK_CHUNK + i];
    sk_row_store(a.partial + part_base + (size_t)ck.c0 * 2 + i, s, ck.first_pass, ck.publish);
  }
  sks_post(ctrl + 18 + par, cs + 1, lane);
}

// one block of the ring -> registers: counter first, then the data (valid if the counter says so: LDS keeps the wave's order)
struct SksBlock { float4 a, b; int flag; };
__device__ __forceinline__ void sks_fetch(sks_ctrl_t produced, const float *ring, int gb, int lane, SksBlock &k) {
  SKS_COMPILER_FENCE();
  k.flag = *produced;
  const float4 *slot = reinterpret_cast<const float4 *>(ring + (gb % SKS_RING) * SKS_SLOT_FLOATS);
  k.a = slot[lane]; k.b = slot[64 + lane];
  SKS_COMPILER_FENCE();                               // (the reads are ISSUED here, ahead of whatever the caller does next)
}
// ... make sure it was there; then the slot is free again (the data sits in registers)
__device__ __forceinline__ bool sks_claim(sks_ctrl_t produced, sks_ctrl_t consumed, const float *ring, int gb, int lane, SksBlock &k) {
  const bool late = __builtin_amdgcn_readfirstlane(k.flag) < gb + 1;
  if (late) {
    sks_wait_ge(produced, gb + 1);
    const float4 *slot = reinterpret_cast<const float4 *>(ring + (gb % SKS_RING) * SKS_SLOT_FLOATS);
    k.a = slot[lane]; k.b = slot[64 + lane];
  }
  sks_post(consumed, gb + 1, lane);                   // (behind the reads in the wave's LDS order)
  return late;
}

// NP: 64-voice (post wave, oscillator wave) pairs per workgroup -- 4: 512 threads, 256 voices per pass (wave w and wave w + 4 share
// a SIMD); 2: 256 threads, 128 voices per pass: a workgroup's four waves land on four different SIMDs, so that in a bank of at
// most one such workgroup per CU every wave has a SIMD -- and its full issue rate -- to itself.
template <bool FILTER, bool ENV, int INTERP, int NP>
__global__ __launch_bounds__(NP * 128, NP) void sk_render_split_kernel(const sk_render_args_t a) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int T = a.lds_table_floats;
  float2 *wsum_all = reinterpret_cast<float2 *>(lds + T);                   // [2][NP][SK_CHUNK]
  float *pairs = lds + T + 2 * NP * SK_CHUNK * 2;
  const sks_ctrl_t ctrl = (sks_ctrl_t)(pairs + NP * SKS_PAIR_FLOATS);
  const char *lds_tab = reinterpret_cast<const char *>(lds);
  const char *glb_tab = reinterpret_cast<const char *>(a.tables);
  const int bid = (int)blockIdx.x - a.wg_shift;        // row of the partial mix; -1: the gain workgroup (sk_finish_block)
  constexpr int NTHREADS = NP * 128;
  if (bid < 0) { sk_finish_block(a, bid, tid, NTHREADS, reinterpret_cast<int *>(lds)); return; }

  {
    const int n4 = T >> 2;                             // padded to a multiple of 4 by the host
    const float4 *src4 = reinterpret_cast<const float4 *>(a.tables);
    float4 *dst4 = reinterpret_cast<float4 *>(lds);
    sk_stage_tables<NTHREADS>(src4, dst4, n4, tid);
    if (tid < SKS_CTRL_INTS) ctrl[tid] = 0;
    __syncthreads();
  }

  const int F = a.num_frames;
  const int nchunks = (F + SK_CHUNK - 1) / SK_CHUNK;
  const int nblk_total = (F + 7) >> 3;                 // 8-frame blocks of the launch; the last one may be short
  const int p = wave & (NP - 1);
  const int n_passes_groups = a.n_groups * (4 / NP);   // passes of NP * 64 voices in the bank
  float *ring = pairs + p * SKS_PAIR_FLOATS;           // between passes (empty) the pair's mailbox: SKS_MAIL rows of 64
  float *xt = ring + SKS_RING * SKS_SLOT_FLOATS;       // the post wave's reduction tile
  float *ret = xt + 8 * SK_XT;                         // what the oscillator wave hands back: phase, x1, x2
  const sks_ctrl_t pc = ctrl + 4 * p;
  const size_t part_base = (size_t)bid * (size_t)F * 2;

  if (wave >= NP) {
    // ------------------------------------------------------------ oscillator wave
    int pass_seq = 0;
    for (int g = bid; g < n_passes_groups; g += a.n_rows) {
      ++pass_seq;
      int gv;
      while (((gv = sks_peek(pc + SKS_GO)) >> 1) != pass_seq) __builtin_amdgcn_s_sleep(2);
      SKS_COMPILER_FENCE();
      if (!(gv & 1)) continue;                         // the owner wave renders this pass alone
      FastRegs r;
      r.phase = ring[lane]; r.inc = ring[64 + lane]; r.lo = ring[128 + lane]; r.hi = ring[192 + lane];
      r.toff4 = __float_as_int(ring[256 + lane]);
      r.span = r.hi - r.lo; r.span2 = r.span + r.span; r.tsize_m1 = 0; r.stop = false;
      v2f xx = {0.0f, 0.0f}, b12 = {0.0f, 0.0f}, b21 = {0.0f, 0.0f};
      float b0 = 0.0f;
      if (FILTER) {
        xx = (v2f){ring[320 + lane], ring[384 + lane]};
        b0 = ring[448 + lane];
        const float b1 = ring[512 + lane], b2 = ring[576 + lane];
        b12 = (v2f){b1, b2}; b21 = (v2f){b2, b1};
      }
      SKS_COMPILER_FENCE();                            // (the mailbox is read before this wave writes block 0 over it)
      const int base = (pass_seq - 1) * nblk_total;
#ifdef SKS_ABL_POST_ONLY
      if (nblk_total > 0) { ret[lane] = r.phase; ret[64 + lane] = xx.x; ret[128 + lane] = xx.y; sks_post(pc + SKS_PRODUCED, base + nblk_total, lane); sks_post(pc + SKS_DONE, pass_seq, lane); continue; }
#endif
      int seen = base;                                 // the consumer's counter as last read
      SKS_STAMP_DECL
      SKS_STAMP_BEGIN()
      float cur[8];
      const bool loz = __all(r.lo == 0.0f);            // plain LUTs in every lane: fast_advance<LOZ>
      if (loz) sks_osc8<INTERP, true>(r, lds_tab, cur, F); else sks_osc8<INTERP, false>(r, lds_tab, cur, F);   // block 0: its gathers are in flight
      for (int b = 0; b < nblk_total; ++b) {
        float nxt[8];
        const int n = F - (b << 3);                    // frames of block b (>= 8: a whole block)
        if (b + 1 < nblk_total) { if (loz) sks_osc8<INTERP, true>(r, lds_tab, nxt, n - 8); else sks_osc8<INTERP, false>(r, lds_tab, nxt, n - 8); }   // block b + 1 on its way before block b is finished
#ifndef SKS_ABL_FREE_RUN
        if (base + b + 1 - SKS_RING > seen) {          // no slot known to be free: look (and wait)
          SKS_WAIT_T0()
          while ((seen = sks_peek(pc + SKS_CONSUMED)) < base + b + 1 - SKS_RING) __builtin_amdgcn_s_sleep(SKS_SLEEP);
          SKS_COMPILER_FENCE();
          SKS_WAIT_T1()
        }
#endif
        sks_ff8<FILTER>(b0, b12, b21, xx, cur, n);
        float4 *slot = reinterpret_cast<float4 *>(ring + ((base + b) % SKS_RING) * SKS_SLOT_FLOATS);
        slot[lane] = make_float4(cur[0], cur[1], cur[2], cur[3]);
        slot[64 + lane] = make_float4(cur[4], cur[5], cur[6], cur[7]);
        sks_post(pc + SKS_PRODUCED, base + b + 1, lane);
#pragma unroll
        for (int q = 0; q < 8; ++q) cur[q] = nxt[q];
      }
      SKS_STAMP_END()
      ret[lane] = r.phase; ret[64 + lane] = xx.x; ret[128 + lane] = xx.y;
      sks_post(pc + SKS_DONE, pass_seq, lane);
    }
  } else {
    // ------------------------------------------------------------ post wave: owns 64 voices
    const bool stems_on = a.stems != nullptr;
    int pass_seq = 0;
    for (int g = bid; g < n_passes_groups; g += a.n_rows) {
      ++pass_seq;
      const int v = g * (NP * 64) + tid;               // (tid < NP * 64 here: the owner waves)
      const bool first_pass = g == bid;
      const bool publish = a.finish && g + a.n_rows >= n_passes_groups;   // the pass that completes this workgroup's row
      FastRegs r;
      bool dead, silent, muted, released = false;
      uint64_t t_start = 0, t_release = 0;
      {
        const uint4 osc = *reinterpret_cast<const uint4 *>(&a.ro[SKP_OSC][v]);
        const uint4 tab = *reinterpret_cast<const uint4 *>(&a.ro[SKP_TAB][v]);
        const uint4 gn = *reinterpret_cast<const uint4 *>(&a.ro[SKP_GAIN][v]);
        const uint4 s0 = *reinterpret_cast<const uint4 *>(&a.rw[SKS_OSC][v]);
        const uint4 s1 = *reinterpret_cast<const uint4 *>(&a.rw[SKS_FILT][v]);
        const uint4 s2 = *reinterpret_cast<const uint4 *>(&a.rw[SKS_MISC][v]);
        r.inc = __uint_as_float(osc.x); r.lo = __uint_as_float(osc.y);
        r.hi = __uint_as_float(osc.z);  r.amp = __uint_as_float(osc.w);
        r.span = r.hi - r.lo; r.span2 = r.span + r.span;
        r.toff4 = (int)tab.x << 2; r.tsize_m1 = (int)tab.y - 1;
        const uint32_t flags = tab.z;
        r.vel = __uint_as_float(gn.x); r.k = __uint_as_float(gn.y);
        r.b0 = __uint_as_float(gn.z);  r.b1 = __uint_as_float(gn.w);
        r.phase = __uint_as_float(s0.x); r.sgain = __uint_as_float(s0.y);
        r.x1 = __uint_as_float(s0.z);    r.x2 = __uint_as_float(s0.w);
        r.y1 = __uint_as_float(s1.x);    r.y2 = __uint_as_float(s1.y);
        r.sample = __uint_as_float(s1.z); r.rw = s1.w;
        r.pan_l = __uint_as_float(s2.z); r.pan_r = __uint_as_float(s2.w);
        r.tf = 0.0f; r.trf = 0.0f;
        r.b2 = r.a1 = r.a2 = 0.0f;
        if (FILTER) {
          const uint4 fl = *reinterpret_cast<const uint4 *>(&a.ro[SKP_FILT][v]);
          r.b2 = __uint_as_float(fl.x); r.a1 = __uint_as_float(fl.y); r.a2 = __uint_as_float(fl.z);
        }
        r.att = r.dec = r.sus = r.rel = r.attdec = r.one_m_sus = r.gain_sustain = 0.0f;
        if (ENV) {
          const uint4 et = *reinterpret_cast<const uint4 *>(&a.ro[SKP_ENV_T][v]);
          const uint4 es = *reinterpret_cast<const uint4 *>(&a.ro[SKP_ENV_S][v]);
          r.att = __uint_as_float(et.x); r.dec = __uint_as_float(et.y);
          r.sus = __uint_as_float(et.z); r.rel = __uint_as_float(et.w);
          r.attdec = r.att + r.dec;                    // synth.c:410
          r.one_m_sus = 1.0f - r.sus;                  // synth.c:413
          r.gain_sustain = r.amp * (r.sus * r.vel);    // synth.c:582,588 in the sustain stage
          t_start = ((uint64_t)es.y << 32) | es.x;
          t_release = ((uint64_t)es.w << 32) | es.z;
          released = t_release != 0;                   // synth.c:417
        }
        dead = (r.rw & SKR_FINISHED) || r.amp == 0.0f || (flags & SKF_INERT);
        muted = (flags & SKF_MUTED) != 0;
        silent = dead || muted;
        // (a clean bank: none of the extended features of sk_render_fast_kernel<STOPS>)
        r.stop = false; r.fin = false; r.hi_stop = 0.0f;
        r.fm_addr = -1; r.fm_k = 0.0f; r.fm_depth = 0.0f;
        r.am_addr = -1; r.pm_addr = -1; r.am_depth = 0.0f; r.pm_depth = 0.0f; r.am_prev = 0.0f; r.pm_prev = 0.0f;
        r.pan_dirty = false; r.rev = false; r.hold_max = 0; r.hold_count = 0; r.quant = 0; r.hold = 0.0f;
        r.nosmooth = false; r.noise = false; r.ophase = 0.0f; r.filt = true; r.use_env = true;
        r.ox1 = r.ox2 = r.oy1 = r.oy2 = 0.0f;
      }
      if (dead) {
        // a skipped voice (synth.c:531-542) is never stored back: inert numbers, exact zeros into the mix, table index 0
        r.inc = 0.0f; r.lo = 0.0f; r.hi = 1.0f; r.span = 1.0f; r.span2 = 2.0f; r.phase = 0.0f;
        r.toff4 = 0; r.tsize_m1 = 0;
        r.k = 0.0f; r.sgain = 0.0f; r.amp = 0.0f; r.gain_sustain = 0.0f;
        r.b0 = r.b1 = r.b2 = r.a1 = r.a2 = 0.0f; r.x1 = r.x2 = r.y1 = r.y2 = 0.0f;
        r.pan_l = r.pan_r = 0.0f; r.rw &= ~SKR_ENV_ACTIVE;
      }
      const bool tame_geom = __all(dead || (r.inc >= 0.0f && r.inc <= 0.5f * r.span && r.phase >= r.lo && r.phase <= r.hi &&
                                            r.lo >= 0.0f && r.hi <= (float)(r.tsize_m1 + 1)));
      const bool any_muted = __any(silent && !dead);
      // a constant envelope level on every lane in every chunk of the launch (the per-chunk test of sk_render_fast_kernel)
      bool all_steady = true;
      if (ENV) {
        const bool idle = !(r.rw & SKR_ENV_ACTIVE);
        if (idle && !dead) r.gain_sustain = r.amp * (0.0f * r.vel);    // is_active == 0: e = 0 (synth.c:400-401)
        for (int c0 = 0; c0 < F; c0 += SK_CHUNK) {
          const uint64_t base = a.count0 + (uint64_t)c0;
          const uint64_t d_on = base - t_start;
          const float tf_first = (float)(d_on + 1);
          const bool ahead = !idle && (int64_t)(t_start - (base + 1)) > 0;
          all_steady = all_steady && __all(dead || idle || (!released && !(tf_first < r.attdec) && !ahead));
        }
      }
      const bool split_ok = tame_geom && all_steady && !stems_on;
      const int cs0 = (pass_seq - 1) * nchunks;        // sequence number of this pass's first chunk

      if (split_ok) {
        // hand the oscillator half over (the mailbox lives in the ring, which is empty now)
        ring[lane] = r.phase; ring[64 + lane] = r.inc; ring[128 + lane] = r.lo; ring[192 + lane] = r.hi; ring[256 + lane] = __int_as_float(r.toff4);
        if (FILTER) { ring[320 + lane] = r.x1; ring[384 + lane] = r.x2; ring[448 + lane] = r.b0; ring[512 + lane] = r.b1; ring[576 + lane] = r.b2; }
        sks_post(pc + SKS_GO, (pass_seq << 1) | 1, lane);
        v2f yy = {r.y1, r.y2};
        FastPk pk;
        pk.b12 = (v2f){r.b1, r.b2}; pk.b21 = (v2f){r.b2, r.b1};
        pk.a12 = (v2f){r.a1, r.a2}; pk.a21 = (v2f){r.a2, r.a1};
        pk.pan = (v2f){r.pan_l, r.pan_r};
        const int base = (pass_seq - 1) * nblk_total;
        SksBlock cur, nxt;
        cur.a = cur.b = nxt.a = nxt.b = make_float4(0.0f, 0.0f, 0.0f, 0.0f); cur.flag = nxt.flag = 0;
        SKS_STAMP_DECL
        SKS_STAMP_BEGIN()
        if (nblk_total > 0) {                          // block 0, the only one this wave has to wait for in the open
          sks_fetch(pc + SKS_PRODUCED, ring, base, lane, cur);
          sks_claim(pc + SKS_PRODUCED, pc + SKS_CONSUMED, ring, base, lane, cur);
        }
        for (int c = 0; c < nchunks; ++c) {
          const int c0 = c * SK_CHUNK;
          const int cn = min(SK_CHUNK, F - c0);
          const int cs = cs0 + c;
          float2 *row = wsum_all + (cs & 1) * NP * SK_CHUNK + wave * SK_CHUNK;
          if (cs >= 2) { SKS_ARR_T0() sks_wait_ge(ctrl + 18 + (cs & 1), cs - 1); SKS_ARR_T1() }   // the chunk that used this buffer before has been added up
          const int nblk = cn >> 3;
          const int gb0 = base + (c0 >> 3);
          const int gb_end = base + nblk_total;
          const bool stall = fast_smoother_stalled<ENV>(r);
          /* block gb0 + B sits in `cur`; the one after it is read into `nxt` BEFORE this one is rendered, claimed after */
#define SKS_STEP(B, ...)                                                                                  \
          {                                                                                               \
            const int gn_ = gb0 + (B) + 1;                                                                \
            const bool more_ = gn_ < gb_end;                                                              \
            if (more_) sks_fetch(pc + SKS_PRODUCED, ring, gn_, lane, nxt);                                \
            const float p_[8] = {cur.a.x, cur.a.y, cur.a.z, cur.a.w, cur.b.x, cur.b.y, cur.b.z, cur.b.w}; \
            __VA_ARGS__                                                                                   \
            if (more_) { SKS_WAIT_T0() if (sks_claim(pc + SKS_PRODUCED, pc + SKS_CONSUMED, ring, gn_, lane, nxt)) { SKS_WAIT_T1() } cur = nxt; } \
          }
#ifdef SKS_ABL_OSC_ONLY
#define SKS_BLOCKS(STALL_, SEL_)                                                                          \
          for (int b_ = 0; b_ < nblk; ++b_) SKS_STEP(b_, asm volatile("" ::"v"(p_[0]), "v"(p_[7]));)
#else
#define SKS_BLOCKS(STALL_, SEL_)                                                                          \
          if (nblk > 0) {                                                                                 \
            float4 ta_, tb_;                                                                              \
            SKS_STEP(0, sks_post8<FILTER, ENV, STALL_, SEL_>(r, pk, yy, p_, xt, lane, silent);)           \
            for (int b_ = 1; b_ < nblk; ++b_) {                                                           \
              SKS_STEP(b_, sks_tile_load(xt, lane, ta_, tb_);                                             \
                           sks_post8<FILTER, ENV, STALL_, SEL_>(r, pk, yy, p_, xt, lane, silent);         \
                           sks_tile_finish(ta_, tb_, row, (b_ - 1) * 8, lane);)                           \
            }                                                                                             \
            sks_tile_load(xt, lane, ta_, tb_);                                                            \
            sks_tile_finish(ta_, tb_, row, (nblk - 1) * 8, lane);                                         \
          }
#endif
          if (any_muted) { if (stall) { SKS_BLOCKS(true, true) } else { SKS_BLOCKS(false, true) } }
          else           { if (stall) { SKS_BLOCKS(true, false) } else { SKS_BLOCKS(false, false) } }
          const int rem = cn & 7;                      // (only the launch's last chunk can end in a short block: it sits in `cur`)
          if (rem) {
            const float p_[8] = {cur.a.x, cur.a.y, cur.a.z, cur.a.w, cur.b.x, cur.b.y, cur.b.z, cur.b.w};
#pragma unroll
            for (int q = 0; q < 7; ++q) {              // (unrolled: p_ stays in registers)
              if (q >= rem) break;
              float s = p_[q];
              if (FILTER) s = (q & 1) ? fast_biquad_fb<false>(pk.a12, pk.a21, s, yy) : fast_biquad_fb<true>(pk.a12, pk.a21, s, yy);
              s = fast_gain_const<ENV, false>(r, s);
              const v2f lr = pk.pan * (v2f){s, s};
              float x = fold_lr(silent ? 0.0f : lr.x, silent ? 0.0f : lr.y);
              half_sum_to_lanes_31_63(x);
              if ((lane & 31) == 31) reinterpret_cast<float *>(&row[nblk * 8 + q])[lane >> 5] = x;
            }
            if (FILTER && (rem & 1)) yy = (v2f){yy.y, yy.x};   // an odd frame count: the newest entry back into .x
          }
#undef SKS_BLOCKS
#undef SKS_STEP
          SksChunk ck; ck.c0 = c0; ck.cn = cn; ck.first_pass = first_pass; ck.publish = publish;
          { SKS_ARR_T0() sks_arrive<NP>(a, ctrl, wsum_all, cs, ck, part_base, lane); SKS_ARR_T1() }
        }
        SKS_STAMP_END()
        r.y1 = yy.x; r.y2 = yy.y;
        sks_wait_ge(pc + SKS_DONE, pass_seq);
        r.phase = ret[lane];
        if (FILTER) { r.x1 = ret[64 + lane]; r.x2 = ret[128 + lane]; }
        SKS_COMPILER_FENCE();
      } else {
        // ---- this pass is not splittable: the owner wave alone, general frames (fast_frame: no assumption beyond a clean bank)
        sks_post(pc + SKS_GO, (pass_seq << 1) | 0, lane);
        bool moved = false;
        for (int c = 0; c < nchunks; ++c) {
          const int c0 = c * SK_CHUNK;
          const int cn = min(SK_CHUNK, F - c0);
          const int cs = cs0 + c;
          float2 *row = wsum_all + (cs & 1) * NP * SK_CHUNK + wave * SK_CHUNK;
          if (cs >= 2) sks_wait_ge(ctrl + 18 + (cs & 1), cs - 1);
          bool steady = true;
          if (ENV) {
            const bool idle = !(r.rw & SKR_ENV_ACTIVE);
            if (idle && !dead) r.gain_sustain = r.amp * (0.0f * r.vel);
            const uint64_t base = a.count0 + (uint64_t)c0;
            const uint64_t d_on = base - t_start;
            const float tf_first = (float)(d_on + 1);
            const bool ahead = !idle && (int64_t)(t_start - (base + 1)) > 0;
            steady = __all(dead || idle || (!released && !(tf_first < r.attdec) && !ahead));
            moved = moved || !steady;
          }
          for (int j = 0; j < cn; ++j) {               // integer clocks, synth.c:401,422: exact for every frame count
            float l, rr;
            if (ENV && !steady) {
              const uint64_t now = a.count0 + (uint64_t)(c0 + j) + 1;
              r.tf = (float)(now - t_start); r.trf = (float)(now - t_release);
              fast_frame<true, FILTER, ENV, false, false, INTERP, false>(r, r.x1, r.x2, r.y1, r.y2, released, lds_tab, glb_tab, l, rr);
            } else {
              fast_frame<true, FILTER, ENV, true, false, INTERP, false>(r, r.x1, r.x2, r.y1, r.y2, released, lds_tab, glb_tab, l, rr);
            }
            if (FILTER) { float t_ = r.x1; r.x1 = r.x2; r.x2 = t_; t_ = r.y1; r.y1 = r.y2; r.y2 = t_; }   // the frame wrote the OLDER slots: newest back in x1 / y1
            l = silent ? 0.0f : l; rr = silent ? 0.0f : rr;
            if (stems_on && v < a.n_voices)            // synth.c:607-611
              reinterpret_cast<float2 *>(a.stems)[(size_t)(c0 + j) * (size_t)a.n_voices + (size_t)v] = make_float2(silent ? 0.0f : l, silent ? 0.0f : rr);
            float x = fold_lr(l, rr);
            half_sum_to_lanes_31_63(x);
            if ((lane & 31) == 31) reinterpret_cast<float *>(&row[j])[lane >> 5] = x;
          }
          SksChunk ck; ck.c0 = c0; ck.cn = cn; ck.first_pass = first_pass; ck.publish = publish;
          sks_arrive<NP>(a, ctrl, wsum_all, cs, ck, part_base, lane);
        }
        if (ENV && moved && lane == 0) sk_note_moved(a, bid);   // the host's "nothing moves" hint was stale: it goes back to sk_render_fast_kernel<RAMPK>
      }

      // store the recurrences; skipped voices keep their state and get voice_sample = 0 (synth.c:532,538)
      if (!dead) {
        uint4 s0, s1;
        s0.x = __float_as_uint(r.phase); s0.y = __float_as_uint(r.sgain);
        s0.z = __float_as_uint(r.x1);    s0.w = __float_as_uint(r.x2);
        s1.x = __float_as_uint(r.y1);    s1.y = __float_as_uint(r.y2);
        s1.z = __float_as_uint(r.sample); s1.w = r.rw;
        *reinterpret_cast<uint4 *>(&a.rw[SKS_OSC][v]) = s0;
        *reinterpret_cast<uint4 *>(&a.rw[SKS_FILT][v]) = s1;
      } else {
        reinterpret_cast<uint32_t *>(&a.rw[SKS_FILT][v])[2] = 0u;
      }
      // the rows of a.partial this wave added up in this pass are read back in the next pass, possibly by another wave of the
      // workgroup (whichever arrives last at the chunk): drained before this wave arrives anywhere again
      if (!publish) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  if (a.finish) sk_finish_block(a, bid, tid, NTHREADS, reinterpret_cast<int *>(pairs + NP * SKS_PAIR_FLOATS + 24), true);
}

// ---------------------------------------------------------------- launcher (C linkage)

extern "C" size_t sk_split_lds_bytes(const sk_render_args_t *args, int pairs) {
  return ((size_t)args->lds_table_floats + 2 * (size_t)pairs * SK_CHUNK * 2 + (size_t)pairs * SKS_PAIR_FLOATS + SKS_CTRL_INTS) * sizeof(float);
}

// pairs: 4 (512-thread workgroups, n_workgroups = passes of 256 voices) or 2 (256-thread workgroups, passes of 128 voices)
extern "C" int sk_launch_render_split(const sk_render_args_t *args, int n_workgroups, int pairs, hipStream_t stream) {
  const size_t lds_bytes = sk_split_lds_bytes(args, pairs);
  dim3 grid((unsigned)(n_workgroups + args->wg_shift)), block((unsigned)pairs * 128);
  const int key = ((args->fast_mode & SKM_FILTER_ALL) ? 2 : 0) | ((args->fast_mode & SKM_ENV_ALL) ? 1 : 0);
#define SKS_LAUNCH_(F_, E_, I_)                                                                                              \
  { if (pairs == 2) hipLaunchKernelGGL((sk_render_split_kernel<F_, E_, I_, 2>), grid, block, lds_bytes, stream, *args);      \
    else hipLaunchKernelGGL((sk_render_split_kernel<F_, E_, I_, 4>), grid, block, lds_bytes, stream, *args); }
#define SKS_LAUNCH(F_, E_)                                                                                                   \
  { if (args->interp == 0) SKS_LAUNCH_(F_, E_, 0) else if (args->interp == 2) SKS_LAUNCH_(F_, E_, 2) else SKS_LAUNCH_(F_, E_, 1) }
  switch (key) {
    case 0: SKS_LAUNCH(false, false) break;
    case 1: SKS_LAUNCH(false, true) break;
    case 2: SKS_LAUNCH(true, false) break;
    default: SKS_LAUNCH(true, true) break;
  }
#undef SKS_LAUNCH
#undef SKS_LAUNCH_
  return (int)hipGetLastError();
}
