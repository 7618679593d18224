// skred_render_split.hip -- sk_render_split_kernel: one voice per lane, the frame split between two wavefronts.
#include "skred_fast_common.hpp"
#include "skred_launch.h"

// ---------------------------------------------------------------- why
//
// Small and mid-size clean banks (BASELINE configs 1 and 2, and the 2^17-voice shard config 3 leaves on each of 8 GPUs) give a
// SIMD one or two wavefronts of sk_render_fast_kernel.  A wavefront issues at most one instruction every ~4 cycles and waits
// out every LDS round trip of its own (table gather -> biquad -> reduction tile), so one or two of them leave the SIMD idle
// more than half of the time (profiles/r04_shard17_*: each wave 31-41 % parked on s_waitcnt).  The per-voice recurrences are
// serial in time, so the only way to give the SIMD more independent instruction streams is to cut the FRAME in two:
//
//   oscillator wave  (wavefronts 4..7 of the workgroup)   phase += inc, wrap, table gather          (osc_next, synth.c:217-275)
//   post wave        (wavefronts 0..3, owns the voices)   biquad, gain / smoother, pan, mix-down    (synth.c:349-364,580-612)
//
// The oscillator of a voice does not depend on anything behind it, so its wave runs AHEAD and hands 8-frame blocks of raw table
// samples to the post wave through a ring in LDS (SKS_RING slots of 8 frames x 64 lanes).  Wave w and wave w + 4 of a workgroup
// share a SIMD (a workgroup's waves are dealt 0 -> 2 -> 1 -> 3 over the SIMDs): a 2^17-voice shard runs four streams per SIMD
// instead of two, a 65 536-voice bank two instead of one.  Both halves call the functions of skred_fast_common.hpp that
// sk_render_fast_kernel calls -- same products, same sums, same order -- so every voice renders to the same bits.
//
// Synchronisation is by LDS words only, never by s_barrier inside a block: LDS executes a wave's accesses in order, so a
// producer's `data, then counter` and a consumer's `counter, then data` need no fence beyond keeping the compiler from
// reordering them.  Every wait is on a counter that the other side advances unconditionally (the oscillator wave produces
// every block of the launch, the post wave consumes every block), so every wait ends.
//
// Which waves take the split path is decided per pass by the OWNER (post) wave: tame geometry (see fast_advance<TAME>) and a
// constant envelope level on every lane for the whole launch.  Anything else -- an envelope in motion because the host's
// "nothing moves" hint was stale, a huge increment, stems -- is rendered by the owner wave alone, frame by frame, on the
// general path of skred_fast_common.hpp (its oscillator wave idles): slow, rare, and the same bits.  The host launches this
// kernel only while it believes nothing moves (skred_bank.c: render_block), so the hint decides speed, never samples.

#define SKS_THREADS 512
#define SKS_RING 4                         /* ring slots per pair (8 frames each) */
#define SKS_SLOT_FLOATS 512                /* [2 halves][64 lanes][4 frames] */
#define SKS_PAIR_FLOATS (SKS_RING * SKS_SLOT_FLOATS + 8 * SK_XT + 64)   /* ring, the post wave's reduction tile (also the mailbox), final phases */
#define SKS_CTRL_INTS 32
// control words: pair p at [4p]: go, produced, consumed, done; [16],[17]: chunk arrival counters; [18],[19]: chunks combined; [24]: finish flag
#define SKS_GO 0
#define SKS_PRODUCED 1
#define SKS_CONSUMED 2
#define SKS_DONE 3

#define SKS_COMPILER_FENCE() asm volatile("" ::: "memory")
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
  while (sks_peek(p) < need) __builtin_amdgcn_s_sleep(1);
  SKS_COMPILER_FENCE();
}
__device__ __forceinline__ void sks_post(sks_ctrl_t p, int v, int lane) {
  SKS_COMPILER_FENCE();
  if (lane == 0) *p = v;
  SKS_COMPILER_FENCE();
}

// eight frames of the post stage -> the wave's reduction tile (SK_FAST_POST8_ of skred_render_fast.hip)
template <bool FILTER, bool ENV, bool STALL, bool SEL>
__device__ __forceinline__ void sks_post8(FastRegs &r, const FastPk &pk, v2f &xx, v2f &yy, const float (&s)[8], float *xt, int lane, bool silent) {
#pragma unroll
  for (int q = 0; q < 8; q += 2) {
    float f0, f1;
    if (FILTER && !SEL) {
      float s0, s1, u;
      fast_post_v<FILTER, ENV, STALL, false, true, false>(r, pk, s[q], xx, yy, s0, u);
      fast_post_v<FILTER, ENV, STALL, false, false, false>(r, pk, s[q + 1], xx, yy, s1, u);
      fast_pan_fold2(s0, s1, pk.pan.x, pk.pan.y, f0, f1);
    } else {
      float l0, r0, l1, r1;
      fast_post_v<FILTER, ENV, STALL, false, true>(r, pk, s[q], xx, yy, l0, r0);
      fast_post_v<FILTER, ENV, STALL, false, false>(r, pk, s[q + 1], xx, yy, l1, r1);
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
__device__ __forceinline__ void sks_arrive(const sk_render_args_t &a, sks_ctrl_t ctrl, const float2 *wsum_all, int cs, const SksChunk &ck,
                                           size_t part_base, int lane) {
  const int par = cs & 1;
  SKS_COMPILER_FENCE();
  int old = 0;
  if (lane == 0) old = __hip_atomic_fetch_add((sks_lds_int *)(ctrl + 16 + par), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // ds_add_rtn_u32
  old = __builtin_amdgcn_readfirstlane(old);
  if (old != 3) return;
  if (lane == 0) ctrl[16 + par] = 0;
  SKS_COMPILER_FENCE();
  const float *w = reinterpret_cast<const float *>(wsum_all + par * 4 * SK_CHUNK);
  for (int i = lane; i < 2 * ck.cn; i += 64) {
    float s = w[0 * 2 * SK_CHUNK + i];
    s += w[1 * 2 * SK_CHUNK + i];
    s += w[2 * 2 * SK_CHUNK + i];
    s += w[3 * 2 * SK_CHUNK + i];
    sk_row_store(a.partial + part_base + (size_t)ck.c0 * 2 + i, s, ck.first_pass, ck.publish);
  }
  sks_post(ctrl + 18 + par, cs + 1, lane);
}

template <bool FILTER, bool ENV, int INTERP>
__global__ __launch_bounds__(SKS_THREADS, 4) void sk_render_split_kernel(const sk_render_args_t a) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int T = a.lds_table_floats;
  float2 *wsum_all = reinterpret_cast<float2 *>(lds + T);                   // [2][4][SK_CHUNK]
  float *pairs = lds + T + 2 * 4 * SK_CHUNK * 2;
  const sks_ctrl_t ctrl = (sks_ctrl_t)(pairs + 4 * SKS_PAIR_FLOATS);
  const char *lds_tab = reinterpret_cast<const char *>(lds);
  const char *glb_tab = reinterpret_cast<const char *>(a.tables);
  const int bid = (int)blockIdx.x - a.wg_shift;        // row of the partial mix; -1: the gain workgroup (sk_finish_block)
  if (bid < 0) { sk_finish_block(a, bid, tid, SKS_THREADS, reinterpret_cast<int *>(lds)); return; }

  {
    const int n4 = T >> 2;                             // padded to a multiple of 4 by the host
    const float4 *src4 = reinterpret_cast<const float4 *>(a.tables);
    float4 *dst4 = reinterpret_cast<float4 *>(lds);
    for (int i = tid; i < n4; i += SKS_THREADS) dst4[i] = src4[i];
    if (tid < SKS_CTRL_INTS) ctrl[tid] = 0;
    __syncthreads();
  }

  const int F = a.num_frames;
  const int nchunks = (F + SK_CHUNK - 1) / SK_CHUNK;
  const int nblk_total = (F + 7) >> 3;                 // 8-frame blocks of the launch; the last one may be short
  const int p = wave & 3;
  float *ring = pairs + p * SKS_PAIR_FLOATS;
  float *xt = ring + SKS_RING * SKS_SLOT_FLOATS;       // the post wave's reduction tile; between passes the pair's mailbox
  float *finph = xt + 8 * SK_XT;
  const sks_ctrl_t pc = ctrl + 4 * p;
  const size_t part_base = (size_t)bid * (size_t)F * 2;

  if (wave >= 4) {
    // ------------------------------------------------------------ oscillator wave
    int pass_seq = 0;
    for (int g = bid; g < a.n_groups; g += a.n_rows) {
      ++pass_seq;
      int gv;
      while (((gv = sks_peek(pc + SKS_GO)) >> 1) != pass_seq) __builtin_amdgcn_s_sleep(2);
      SKS_COMPILER_FENCE();
      if (!(gv & 1)) continue;                         // the owner wave renders this pass alone
      FastRegs r;
      r.phase = xt[lane]; r.inc = xt[64 + lane]; r.lo = xt[128 + lane]; r.hi = xt[192 + lane];
      r.toff4 = __float_as_int(xt[256 + lane]);
      r.span = r.hi - r.lo; r.span2 = r.span + r.span; r.tsize_m1 = 0; r.stop = false;
      const int base = (pass_seq - 1) * nblk_total;
      for (int b = 0; b < nblk_total; ++b) {
        sks_wait_ge(pc + SKS_CONSUMED, base + b + 1 - SKS_RING);       // a free slot
        float s[8];
        const int n = F - (b << 3);
        if (n >= 8) {
#pragma unroll
          for (int q = 0; q < 8; ++q) s[q] = fast_fetch<true, INTERP, true>(lds_tab, glb_tab, r, fast_advance<true>(r));
        } else {
#pragma unroll
          for (int q = 0; q < 8; ++q) { s[q] = 0.0f; if (q < n) s[q] = fast_fetch<true, INTERP, true>(lds_tab, glb_tab, r, fast_advance<true>(r)); }
        }
        float4 *slot = reinterpret_cast<float4 *>(ring + ((base + b) % SKS_RING) * SKS_SLOT_FLOATS);
        slot[lane] = make_float4(s[0], s[1], s[2], s[3]);
        slot[64 + lane] = make_float4(s[4], s[5], s[6], s[7]);
        sks_post(pc + SKS_PRODUCED, base + b + 1, lane);
      }
      finph[lane] = r.phase;
      sks_post(pc + SKS_DONE, pass_seq, lane);
    }
  } else {
    // ------------------------------------------------------------ post wave: owns 64 voices
    const bool stems_on = a.stems != nullptr;
    int pass_seq = 0;
    for (int g = bid; g < a.n_groups; g += a.n_rows) {
      ++pass_seq;
      const int v = g * SK_GROUP + tid;                // (tid < 256 here: waves 0..3)
      const bool first_pass = g == bid;
      const bool publish = a.finish && g + a.n_rows >= a.n_groups;   // the pass that completes this workgroup's row
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
        // hand the oscillator half over
        xt[lane] = r.phase; xt[64 + lane] = r.inc; xt[128 + lane] = r.lo; xt[192 + lane] = r.hi; xt[256 + lane] = __int_as_float(r.toff4);
        sks_post(pc + SKS_GO, (pass_seq << 1) | 1, lane);
        v2f xx = {r.x1, r.x2}, yy = {r.y1, r.y2};
        FastPk pk;
        pk.b12 = (v2f){r.b1, r.b2}; pk.b21 = (v2f){r.b2, r.b1};
        pk.a12 = (v2f){r.a1, r.a2}; pk.a21 = (v2f){r.a2, r.a1};
        pk.pan = (v2f){r.pan_l, r.pan_r};
        const int base = (pass_seq - 1) * nblk_total;
        for (int c = 0; c < nchunks; ++c) {
          const int c0 = c * SK_CHUNK;
          const int cn = min(SK_CHUNK, F - c0);
          const int cs = cs0 + c;
          float2 *row = wsum_all + (cs & 1) * 4 * SK_CHUNK + wave * SK_CHUNK;
          if (cs >= 2) sks_wait_ge(ctrl + 18 + (cs & 1), cs - 1);      // the chunk that used this buffer before has been added up
          const int nblk = cn >> 3;
          const int gb0 = base + (c0 >> 3);
          const bool stall = fast_smoother_stalled<ENV>(r);
#define SKS_GET(B)                                                                                        \
          float s_[8];                                                                                    \
          {                                                                                               \
            const int gb_ = gb0 + (B);                                                                    \
            sks_wait_ge(pc + SKS_PRODUCED, gb_ + 1);                                                      \
            const float4 *slot_ = reinterpret_cast<const float4 *>(ring + (gb_ % SKS_RING) * SKS_SLOT_FLOATS); \
            const float4 lo_ = slot_[lane], hi_ = slot_[64 + lane];                                       \
            sks_post(pc + SKS_CONSUMED, gb_ + 1, lane);    /* (behind the reads in the wave's LDS order) */ \
            s_[0] = lo_.x; s_[1] = lo_.y; s_[2] = lo_.z; s_[3] = lo_.w;                                   \
            s_[4] = hi_.x; s_[5] = hi_.y; s_[6] = hi_.z; s_[7] = hi_.w;                                   \
          }
#define SKS_BLOCKS(STALL_, SEL_)                                                                          \
          if (nblk > 0) {                                                                                 \
            float4 ta_, tb_;                                                                              \
            { SKS_GET(0) sks_post8<FILTER, ENV, STALL_, SEL_>(r, pk, xx, yy, s_, xt, lane, silent); }     \
            for (int b_ = 1; b_ < nblk; ++b_) {                                                           \
              SKS_GET(b_)                                                                                 \
              sks_tile_load(xt, lane, ta_, tb_);                                                          \
              sks_post8<FILTER, ENV, STALL_, SEL_>(r, pk, xx, yy, s_, xt, lane, silent);                  \
              sks_tile_finish(ta_, tb_, row, (b_ - 1) * 8, lane);                                         \
            }                                                                                             \
            sks_tile_load(xt, lane, ta_, tb_);                                                            \
            sks_tile_finish(ta_, tb_, row, (nblk - 1) * 8, lane);                                         \
          }
          if (any_muted) { if (stall) { SKS_BLOCKS(true, true) } else { SKS_BLOCKS(false, true) } }
          else           { if (stall) { SKS_BLOCKS(true, false) } else { SKS_BLOCKS(false, false) } }
          const int rem = cn & 7;                      // (only the launch's last chunk can end in a short block)
          if (rem) {
            SKS_GET(nblk)
#pragma unroll
            for (int q = 0; q < 7; ++q) {              // (unrolled: s_ stays in registers)
              if (q >= rem) break;
              float l, rr;
              if (!(q & 1)) fast_post_v<FILTER, ENV, false, false, true>(r, pk, s_[q], xx, yy, l, rr);
              else          fast_post_v<FILTER, ENV, false, false, false>(r, pk, s_[q], xx, yy, l, rr);
              l = silent ? 0.0f : l; rr = silent ? 0.0f : rr;
              float x = fold_lr(l, rr);
              half_sum_to_lanes_31_63(x);
              if ((lane & 31) == 31) reinterpret_cast<float *>(&row[nblk * 8 + q])[lane >> 5] = x;
            }
            if (FILTER && (rem & 1)) { xx = (v2f){xx.y, xx.x}; yy = (v2f){yy.y, yy.x}; }   // an odd frame count: the newest entries back into .x
          }
#undef SKS_BLOCKS
#undef SKS_GET
          SksChunk ck; ck.c0 = c0; ck.cn = cn; ck.first_pass = first_pass; ck.publish = publish;
          sks_arrive(a, ctrl, wsum_all, cs, ck, part_base, lane);
        }
        r.x1 = xx.x; r.x2 = xx.y; r.y1 = yy.x; r.y2 = yy.y;
        sks_wait_ge(pc + SKS_DONE, pass_seq);
        r.phase = finph[lane];
      } else {
        // ---- this pass is not splittable: the owner wave alone, general frames (fast_frame: no assumption beyond a clean bank)
        sks_post(pc + SKS_GO, (pass_seq << 1) | 0, lane);
        bool moved = false;
        for (int c = 0; c < nchunks; ++c) {
          const int c0 = c * SK_CHUNK;
          const int cn = min(SK_CHUNK, F - c0);
          const int cs = cs0 + c;
          float2 *row = wsum_all + (cs & 1) * 4 * SK_CHUNK + wave * SK_CHUNK;
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
          sks_arrive(a, ctrl, wsum_all, cs, ck, part_base, lane);
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
  if (a.finish) sk_finish_block(a, bid, tid, SKS_THREADS, reinterpret_cast<int *>(pairs + 4 * SKS_PAIR_FLOATS + 24), true);
}

// ---------------------------------------------------------------- launcher (C linkage)

extern "C" size_t sk_split_lds_bytes(const sk_render_args_t *args) {
  return ((size_t)args->lds_table_floats + 2 * 4 * SK_CHUNK * 2 + 4 * SKS_PAIR_FLOATS + SKS_CTRL_INTS) * sizeof(float);
}

extern "C" int sk_launch_render_split(const sk_render_args_t *args, int n_workgroups, hipStream_t stream) {
  const size_t lds_bytes = sk_split_lds_bytes(args);
  dim3 grid((unsigned)(n_workgroups + args->wg_shift)), block(SKS_THREADS);
  const int key = ((args->fast_mode & SKM_FILTER_ALL) ? 2 : 0) | ((args->fast_mode & SKM_ENV_ALL) ? 1 : 0);
#define SKS_LAUNCH(F_, E_)                                                                                                   \
  { if (args->interp == 0) hipLaunchKernelGGL((sk_render_split_kernel<F_, E_, 0>), grid, block, lds_bytes, stream, *args);  \
    else if (args->interp == 2) hipLaunchKernelGGL((sk_render_split_kernel<F_, E_, 2>), grid, block, lds_bytes, stream, *args); \
    else hipLaunchKernelGGL((sk_render_split_kernel<F_, E_, 1>), grid, block, lds_bytes, stream, *args); }
  switch (key) {
    case 0: SKS_LAUNCH(false, false) break;
    case 1: SKS_LAUNCH(false, true) break;
    case 2: SKS_LAUNCH(true, false) break;
    default: SKS_LAUNCH(true, true) break;
  }
#undef SKS_LAUNCH
  return (int)hipGetLastError();
}
