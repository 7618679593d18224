// skred_render_fast.hip -- sk_render_fast_kernel: one voice per lane, clean banks.
#include "skred_kernel_common.hpp"
#include "skred_launch.h"

// ---------------------------------------------------------------- fast render kernel
//
// Same arithmetic, per voice bit-identical to the generic kernel above, for "clean" banks -- the
// host only selects it when (bank-wide, skred_bank.c:classify): no one-shot voice that stops at
// its table end, no reverse playback, no sample&hold / bit-crush / noise voices, no modulators,
// smoother on everywhere, filter on for ALL voices or for none, envelope for ALL or none, and
// every phase / increment / loop bound finite (so the !isfinite() branch of osc_next can never
// fire and `voice_finished` cannot change inside a launch).  Under those conditions:
//   * which lanes are skipped (synth.c:531-542) is a launch constant -> one mask, no per-frame
//     branch; their state is left untouched in HBM and their output is select-masked to 0;
//   * the wrap is straight-line: for span <= x < 2*span, fmodf(x, span) == x - span exactly
//     (Sterbenz), anything else (huge increments, NaN) drops into the exact generic path;
//   * envelope time is carried as a float that gains 1.0f per frame: exact below 2^24 frames
//     and then equal to the reference's (float)(uint64) conversion; once every lane of the wave is
//     in its sustain stage (monotone within a launch) the envelope costs nothing per frame and
//     amp*env is a per-lane constant.
// What remains per frame is ~45 VALU instructions instead of ~107, with almost no scalar
// branch bookkeeping.


#include "skred_fast_common.hpp"

// ---- table windows for pools that do not fit in LDS (see skred_render_fast2.hip for the reasoning) ----
// One voice per lane: every 8 frames the lane copies the SK_WIN table samples its voice is about to cross into
// win[row][lane]; a voice that could wrap, fold its second tap or outrun the window in this block takes the
// ordinary gather.  Used for PCM banks too small to fill the machine with two voices per lane.
typedef float fwin4_t __attribute__((ext_vector_type(4), aligned(4)));

struct FastWin {
  int base;
  bool direct, any_direct;
};

__device__ __forceinline__ void fast_win_fill(const FastRegs &r, bool dead, FastWin &w, float *win, int lane,
                                              const char *__restrict__ glb_tab) {
  const float d8 = 8.0f * r.inc;
  const bool fits = d8 <= (float)(SK_WIN - 3) + 0.5f && r.phase + d8 + 2.0f < r.hi;
  w.direct = !dead && !fits;
  w.base = (int)r.phase;
  if (!w.direct) {
    const char *src = glb_tab + (r.toff4 + (w.base << 2));
    float *dst = win + lane;
#pragma unroll
    for (int k = 0; k < SK_WIN / 4; ++k) {
      const fwin4_t t = *reinterpret_cast<const fwin4_t *>(src + 16 * k);
      dst[(4 * k + 0) * 64] = t.x; dst[(4 * k + 1) * 64] = t.y;
      dst[(4 * k + 2) * 64] = t.z; dst[(4 * k + 3) * 64] = t.w;
    }
  }
  w.any_direct = __any(w.direct);
}

template <int INTERP, bool STOPS = false>
__device__ __forceinline__ float fast_fetch_win(const FastRegs &r, const FastWin &w, const float *win, int lane,
                                                const char *__restrict__ glb_tab, float p) {
  const int idx = (int)p;
  const int rel = w.direct ? 0 : idx - w.base;
  // (an LDS-address-space pointer: left generic, the compiler merges this read and the direct gather below into ONE load
  // through a selected pointer -- a FLAT load -- and with packed lanes hipcc 7.2 then dies in instruction selection on one
  // instantiation: "Illegal instruction detected ... V_CMP_NE_U32_e32 0, $src_shared_base")
  typedef __attribute__((address_space(3))) const float lds_cf;
  const lds_cf *src = (const lds_cf *)(win + rel * 64 + lane);
  const float ta = src[0];
  float s = ta;
  if (INTERP != 0) {
    const float tb = src[64];
    s = ta + (p - (float)idx) * (tb - ta);
  }
  if (w.any_direct) {
    if (w.direct) s = fast_fetch<false, INTERP, !STOPS>(nullptr, glb_tab, r, p);   // a finishing phase needs the clamps
  }
  return s;
}

// A voice reached its table end on the frame just rendered (r.fin): from the next frame on the reference skips it
// (synth.c:531-535: voice_sample = 0, nothing advances).  Its state is final now, so it is stored here -- delay
// line in the reference's order whatever role the registers play at this point of the pair -- and the lane turns
// into a skipped one for the rest of the launch (exact zeros, never stored again).
__device__ __forceinline__ void fast_finish(const sk_render_args_t &a, FastRegs &r, int v, bool &dead, bool &silent,
                                            bool &sample_final, bool swapped, bool last_frame, const uint2 misc_xy) {
  if (r.fin) {
    if (r.pan_dirty || r.hold_max) {                      // its pan and hold state, as of this frame
      *reinterpret_cast<uint4 *>(&a.rw[SKS_MISC][v]) =
          r.hold_max ? make_uint4(__float_as_uint(r.hold), (uint32_t)r.hold_count, __float_as_uint(r.pan_l), __float_as_uint(r.pan_r))
                     : make_uint4(misc_xy.x, misc_xy.y, __float_as_uint(r.pan_l), __float_as_uint(r.pan_r));
      r.pan_dirty = false;
    }
    uint4 s0, s1;
    s0.x = __float_as_uint(r.phase); s0.y = __float_as_uint(r.sgain);
    s0.z = __float_as_uint(swapped ? r.x2 : r.x1); s0.w = __float_as_uint(swapped ? r.x1 : r.x2);
    s1.x = __float_as_uint(swapped ? r.y2 : r.y1); s1.y = __float_as_uint(swapped ? r.y1 : r.y2);
    if (!r.filt) { s0.z = __float_as_uint(r.ox1); s0.w = __float_as_uint(r.ox2); s1.x = __float_as_uint(r.oy1); s1.y = __float_as_uint(r.oy2); }
    s1.z = last_frame ? __float_as_uint(r.sample) : 0u;    // a later frame of this launch would have zeroed it
    s1.w = r.rw | SKR_FINISHED;
    *reinterpret_cast<uint4 *>(&a.rw[SKS_OSC][v]) = s0;
    *reinterpret_cast<uint4 *>(&a.rw[SKS_FILT][v]) = s1;
    dead = true; silent = true; sample_final = true;
#ifdef SK_PROBE_TU
    r.probe = nullptr;                                   // (from the next frame on the reference skips it: zeros, which the host put there)
#endif
    r.inc = 0.0f; r.lo = 0.0f; r.hi = 1.0f; r.span = 1.0f; r.span2 = 2.0f; r.phase = 0.0f;
    r.toff4 = 0; r.tsize_m1 = 0;
    r.k = 0.0f; r.sgain = 0.0f; r.amp = 0.0f; r.gain_sustain = 0.0f;
    r.b0 = r.b1 = r.b2 = r.a1 = r.a2 = 0.0f; r.x1 = r.x2 = r.y1 = r.y2 = 0.0f;
    r.pan_l = r.pan_r = 0.0f; r.rw &= ~SKR_ENV_ACTIVE;
    r.stop = false; r.fin = false; r.hi_stop = 0.0f; r.fm_addr = -1; r.am_addr = -1; r.pm_addr = -1; r.rev = false;
    r.hold_max = 0; r.quant = 0; r.nosmooth = false;
  }
}

// one frame of the chunk loop: STEADY_ selects the envelope mode, A/B the delay-line roles
// per-voice stems (synth.c:607-611; a.stems != NULL): frame I of this chunk, this lane's voice; skipped and muted voices
// write exact +0.0f like the reference (an inert lane's product can be -0.0f).  Only the frame-by-frame paths carry
// it: a launch with stems takes those for every chunk, so the block and pair paths stay free of the test (one more
// scalar instruction per frame costs a lone wave 5 % on a small bank)
#define SK_FAST_STEM(I, L, R)                                                                            \
  if (stems_on && v < a.n_voices)                                                                        \
    reinterpret_cast<float2 *>(a.stems)[(size_t)(c0 + (I)) * (size_t)a.n_voices + (size_t)v] =          \
        make_float2(silent ? 0.0f : (L), silent ? 0.0f : (R));
// (STOPS) waves that hold a noise voice: the frame's shared draw, synth.c:525 -- one LCG step per frame of the launch
#define SK_FAST_DRAW()                                                                                   \
  float white_ = 0.0f;                                                                                   \
  if (STOPS && (xf & XF_NOISE)) { rng = rng * LCG_A + LCG_C; white_ = (float)((int32_t)(uint32_t)(rng >> 32)) / 2147483648.0f; }
/* block paths: the oscillator sample S of this frame, or -- a noise lane -- the frame's shared draw (synth.c:543-546); the
   wave advances the LCG once per frame it renders, in frame order, as SK_FAST_DRAW does on the frame paths */
/* NOISE_ is a compile-time constant: a test of xf per frame splits the block into eight scheduling regions (measured: 4-7 % on
   every extended bank), so waves with noise lanes get their own copy of the (non-pipelined) block loops instead */
#define SK_FAST_BLOCK_SAMPLE(S, NOISE_)                                                                  \
  ((STOPS && (NOISE_)) ? (rng = rng * LCG_A + LCG_C, (r.noise ? (float)((int32_t)(uint32_t)(rng >> 32)) / 2147483648.0f : (S))) : (S))
#define SK_FAST_FRAME(J, STEADY_, XN, XO, YN, YO, SWAPPED_)                                              \
  {                                                                                                      \
    float l, rr;                                                                                         \
    SK_FAST_DRAW()                                                                                       \
    fast_frame<TAB_LDS, FILTER, ENV, STEADY_, false, INTERP, STOPS>(r, XN, XO, YN, YO, released, lds_tab, glb_tab, l, rr, xf, muted, white_); \
    l = silent ? 0.0f : l; rr = silent ? 0.0f : rr;                                                      \
    SK_FAST_STEM(J, l, rr)                                                                               \
    if (STOPS && (xf & XF_STOP) && __any(r.fin)) fast_finish(a, r, v, dead, silent, sample_final, SWAPPED_, c0 + (J) == a.num_frames - 1, misc_xy); \
    SK_REDUCE_AND_STORE(J)                                                                               \
  }
// two steady frames (J even, J+1): delay-line roles swap in between, one 4-chain reduction, one 16-byte store.
// Both oscillator halves run first (the phase recurrence does not depend on the samples): the two table reads are
// in flight together and the second frame's read latency hides behind the first frame's biquad / gain chain -- a
// small bank has one wave per SIMD and nothing else to hide it behind.
#define SK_FAST_PAIR_STEADY_(J, TAME_, NOISE_) /* TAME_ loops run only when no live lane is muted: no output select */ \
  {                                                                                                      \
    float l0, r0, l1, r1;                                                                                \
    const float sa_ = fast_fetch<TAB_LDS, INTERP, TAME_>(lds_tab, glb_tab, r, fast_advance<TAME_>(r));    \
    const float sb_ = fast_fetch<TAB_LDS, INTERP, TAME_>(lds_tab, glb_tab, r, fast_advance<TAME_>(r));    \
    fast_post_v<FILTER, ENV, false, STOPS, true>(r, pk, SK_FAST_BLOCK_SAMPLE(sa_, NOISE_), xx, yy, l0, r0, xf_blk); \
    fast_post_v<FILTER, ENV, false, STOPS, false>(r, pk, SK_FAST_BLOCK_SAMPLE(sb_, NOISE_), xx, yy, l1, r1, xf_blk); \
    if (!(TAME_)) { l0 = silent ? 0.0f : l0; r0 = silent ? 0.0f : r0; l1 = silent ? 0.0f : l1; r1 = silent ? 0.0f : r1; } \
    SK_REDUCE4_AND_STORE(J)                                                                              \
  }
#define SK_FAST_PAIR_STEADY(J, TAME_) SK_FAST_PAIR_STEADY_(J, TAME_, false)
// Eight steady frames (J..J+7) of a tame wave of an LDS-table bank with the cross-lane sum through LDS instead of
// the VALU.  Every lane folds its (L,R) of a frame into one float (fold_lr: L pair sums in lanes 0..31, R pair sums in
// lanes 32..63) and parks it in the wave-private tile xt[8 frames][SK_XT]: one ds_write_b32 per frame.  Then lane
// (f = lane&7, seg = lane>>3) adds the 8 floats of segment seg of frame f (two ds_read_b128), segments 0..3 (L) and
// 4..7 (R) are added across lanes in registers (one DPP add inside the 16-lane row, one v_permlane16_swap + add across
// the row pair), and lanes 0..7 / 32..39 store the frame's L / R total.  Per frame 2 VALU + one 4-byte LDS write, per
// block ~12 VALU + 3 LDS instructions -- against 12 v_add_dpp per frame for the butterfly, and half the LDS bytes of
// the (L,R)-pair tile this replaces (with 2 waves per SIMD the CU's one LDS pipe was as busy as its VALUs).
// All traffic stays inside one wavefront (LDS executes a wave's accesses in order): no s_barrier.
#define SK_FAST_WAVE_SYNC()                                 \
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
  __builtin_amdgcn_wave_barrier();                          \
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
// Software pipeline across blocks: the 8 table gathers of block J are issued first (the phase recurrence does not
// depend on the samples), then the tile of the PREVIOUS block is reduced while they are in flight, then the biquad /
// gain chains of block J run and park their outputs.  SK_FAST_LDS_FLUSH reduces the last pending tile of a chunk.
#define SK_FAST_TILE_REDUCE(JP)                                                                          \
  {                                                                                                      \
    SK_FAST_WAVE_SYNC()                                                                                  \
    float t_;                                                                                            \
    {                                                                                                    \
      const float4 *src_ = reinterpret_cast<const float4 *>(xt + (lane & 7) * SK_XT + (lane >> 3) * 8);  \
      const float4 a_ = src_[0], b_ = src_[1];                                                           \
      t_ = ((((((a_.x + a_.y) + a_.z) + a_.w) + b_.x) + b_.y) + b_.z) + b_.w;                            \
    }                                                                                                    \
    t_ = row_pair_add(row_ror8_add(t_));       /* segments 0..3 -> lanes 0..7 (L), 4..7 -> lanes 32..39 (R) */ \
    if ((lane & 24) == 0) reinterpret_cast<float *>(&wsum[wave * SK_CHUNK + (JP) + (lane & 7)])[lane >> 5] = t_; \
    SK_FAST_WAVE_SYNC()                                                                                  \
  }
#define SK_FAST_LDS_FLUSH() if (pend_j >= 0) { SK_FAST_TILE_REDUCE(pend_j) pend_j = -1; }
// All 8-frame blocks of a chunk, software-pipelined one block deep: while the biquad / gain chains of block b run
// (a serial recurrence over its 8 frames), the oscillator of block b+1 advances and gathers (another serial recurrence,
// independent of the first) and the tile of block b-1 -- read at the top, before block b's outputs overwrite it -- is
// added up.  The three strands sit in ONE scheduling region (the wave-level fences only bracket the tile reads), so the
// instruction scheduler can weave them: with one or two waves per SIMD there is nothing else to fill the dependent-issue
// gaps of a single recurrence with.
#define SK_FAST_OSC8(DST)                                                                                \
  _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_)                                                       \
    DST[q_] = fast_fetch<TAB_LDS, INTERP, true>(lds_tab, glb_tab, r, fast_advance<true, false, SK_LOZ_>(r));
#define SK_FAST_POST8_(SRC, STALL_, RAMP_, NOISE_)                                                                      \
  _Pragma("unroll") for (int q_ = 0; q_ < 8; q_ += 2) {                                                  \
    float f0_, f1_;                                                                                      \
    if (FILTER) {              /* two or more waves per SIMD: plain products, swaps spaced by hand */    \
      float s0_, s1_, u_;                                                                                \
      fast_post_v<FILTER, ENV, STALL_, STOPS, true, false, RAMP_>(r, pk, SK_FAST_BLOCK_SAMPLE(SRC[q_], NOISE_), xx, yy, s0_, u_, xf_blk, &ev_);         \
      fast_post_v<FILTER, ENV, STALL_, STOPS, false, false, RAMP_>(r, pk, SK_FAST_BLOCK_SAMPLE(SRC[q_ + 1], NOISE_), xx, yy, s1_, u_, xf_blk, &ev_);    \
      if (SK_PAIRED_) fast_pan_fold2p(s0_, s1_, pk.plA, pk.plB, pk.prA, pk.prB, f0_, f1_);               \
      else fast_pan_fold2(s0_, s1_, pk.pan.x, pk.pan.y, f0_, f1_);                                       \
    } else {                   /* a bare oscillator bank: hipcc weaves the fold into the oscillator steps */     \
      float l0_, r0_, l1_, r1_;                                                                          \
      fast_post_v<FILTER, ENV, STALL_, STOPS, true, true, RAMP_>(r, pk, SK_FAST_BLOCK_SAMPLE(SRC[q_], NOISE_), xx, yy, l0_, r0_, xf_blk, &ev_);               \
      fast_post_v<FILTER, ENV, STALL_, STOPS, false, true, RAMP_>(r, pk, SK_FAST_BLOCK_SAMPLE(SRC[q_ + 1], NOISE_), xx, yy, l1_, r1_, xf_blk, &ev_);          \
      f0_ = fold_lr(l0_, r0_); f1_ = fold_lr(l1_, r1_);                                                  \
    }                                                                                                    \
    xt[q_ * SK_XT + lane] = f0_;                                                                         \
    xt[(q_ + 1) * SK_XT + lane] = f1_;                                                                   \
  }
#define SK_FAST_POST8(SRC, STALL_) SK_FAST_POST8_(SRC, STALL_, false, false)
/* the two strands written frame pair by frame pair, the way they should issue: oscillator of the NEXT block, chains of this one */
#define SK_FAST_OSC_POST8(DST, SRC, STALL_)                                                               \
  _Pragma("unroll") for (int q_ = 0; q_ < 8; q_ += 2) {                                                  \
    float f0_, f1_;                                                                                      \
    DST[q_] = fast_fetch<TAB_LDS, INTERP, true>(lds_tab, glb_tab, r, fast_advance<true, false, SK_LOZ_>(r));             \
    DST[q_ + 1] = fast_fetch<TAB_LDS, INTERP, true>(lds_tab, glb_tab, r, fast_advance<true, false, SK_LOZ_>(r));         \
    if (FILTER) {              /* two or more waves per SIMD: plain products, swaps spaced by hand */    \
      float s0_, s1_, u_;                                                                                \
      fast_post_v<FILTER, ENV, STALL_, STOPS, true, false>(r, pk, SRC[q_], xx, yy, s0_, u_, xf_blk);         \
      fast_post_v<FILTER, ENV, STALL_, STOPS, false, false>(r, pk, SRC[q_ + 1], xx, yy, s1_, u_, xf_blk);    \
      if (SK_PAIRED_) fast_pan_fold2p(s0_, s1_, pk.plA, pk.plB, pk.prA, pk.prB, f0_, f1_);               \
      else fast_pan_fold2(s0_, s1_, pk.pan.x, pk.pan.y, f0_, f1_);                                       \
    } else {                   /* a bare oscillator bank: hipcc weaves the fold into the oscillator steps */     \
      float l0_, r0_, l1_, r1_;                                                                          \
      fast_post_v<FILTER, ENV, STALL_, STOPS, true>(r, pk, SRC[q_], xx, yy, l0_, r0_, xf_blk);               \
      fast_post_v<FILTER, ENV, STALL_, STOPS, false>(r, pk, SRC[q_ + 1], xx, yy, l1_, r1_, xf_blk);          \
      f0_ = fold_lr(l0_, r0_); f1_ = fold_lr(l1_, r1_);                                                  \
    }                                                                                                    \
    xt[q_ * SK_XT + lane] = f0_;                                                                         \
    xt[(q_ + 1) * SK_XT + lane] = f1_;                                                                   \
  }
#define SK_FAST_TILE_LOAD(TA, TB)                                                                        \
  {                                                                                                      \
    SK_FAST_WAVE_SYNC()                                                                                  \
    const float4 *src_ = reinterpret_cast<const float4 *>(xt + (lane & 7) * SK_XT + (lane >> 3) * 8);    \
    TA = src_[0]; TB = src_[1];                                                                          \
    SK_FAST_WAVE_SYNC()                                                                                  \
  }
/* (SK_PAIRED_: tile row q holds (L of frame q | L of frame q + 1), row q + 1 the R's -- fast_pan_fold2p --, so the total of row
   `lane & 7`, half `lane >> 5` belongs to frame (row & 6) + half, channel row & 1) */                   \
#define SK_FAST_TILE_FINISH(TA, TB, JP)                                                                  \
  {                                                                                                      \
    float t_ = ((((((TA.x + TA.y) + TA.z) + TA.w) + TB.x) + TB.y) + TB.z) + TB.w;                        \
    t_ = row_pair_add(row_ror8_add(t_));                                                                 \
    if ((lane & 24) == 0) reinterpret_cast<float *>(&wsum[wave * SK_CHUNK + (JP) + (SK_PAIRED_ ? (lane & 6) + (lane >> 5) : (lane & 7))])[SK_PAIRED_ ? (lane & 1) : (lane >> 5)] = t_; \
  }
#ifndef SK_FAST_NO_PAIRED
#define SK_FAST_PAIRED_OK true
#else
#define SK_FAST_PAIRED_OK false
#endif
#define SK_FAST_LDS_CHUNK(STALL_)                                                                        \
  {                                                                                                      \
    constexpr bool SK_PAIRED_ = FILTER && INTERP == 0 && SK_FAST_PAIRED_OK;   /* (these blocks: one swap per frame pair; with the linear lookup's registers the form measured slower) */ \
    const int nblk_ = cn >> 3;                                                                           \
    if (nblk_ > 0) {                                                                                     \
      float sa_[8], sb_[8];                                                                              \
      float4 ta_, tb_;                                                                                   \
      SK_FAST_OSC8(sa_)                                                                                  \
      if (nblk_ > 1) {                                                                                   \
        SK_FAST_OSC_POST8(sb_, sa_, STALL_)                                             /* block 0 */    \
        int b_ = 1;                                                                                      \
        for (; b_ + 2 < nblk_; b_ += 2) {          /* two blocks per trip: the gathers ping-pong, no moves */ \
          SK_FAST_TILE_LOAD(ta_, tb_)                                                                    \
          SK_FAST_OSC_POST8(sa_, sb_, STALL_) SK_FAST_TILE_FINISH(ta_, tb_, (b_ - 1) * 8)                \
          SK_FAST_TILE_LOAD(ta_, tb_)                                                                    \
          SK_FAST_OSC_POST8(sb_, sa_, STALL_) SK_FAST_TILE_FINISH(ta_, tb_, b_ * 8)                      \
        }                                                                                                \
        if (b_ + 1 < nblk_) {                      /* an odd middle block */                             \
          SK_FAST_TILE_LOAD(ta_, tb_)                                                                    \
          SK_FAST_OSC_POST8(sa_, sb_, STALL_) SK_FAST_TILE_FINISH(ta_, tb_, (b_ - 1) * 8)                \
          _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) sb_[q_] = sa_[q_];                            \
        }                                                                                                \
        SK_FAST_TILE_LOAD(ta_, tb_)                                                                      \
        SK_FAST_POST8(sb_, STALL_) SK_FAST_TILE_FINISH(ta_, tb_, (nblk_ - 2) * 8)       /* last block */ \
      } else {                                                                                           \
        SK_FAST_POST8(sa_, STALL_)                                                                       \
      }                                                                                                  \
      SK_FAST_TILE_LOAD(ta_, tb_)                                                                        \
      SK_FAST_TILE_FINISH(ta_, tb_, (nblk_ - 1) * 8)                                                     \
      SK_FAST_WAVE_SYNC()                                                                                \
    }                                                                                                    \
    j = nblk_ << 3;                                                                                      \
  }
// the steady block paths keep the delay line in register pairs (fast_post_v): in at the start of such a chunk, out at its
// end (an even number of frames later the newest entries are back in .x = x1 / y1)
#define SK_FAST_PACK_IN()                                                                 \
  v2f xx = {r.x1, r.x2}, yy = {r.y1, r.y2};                                               \
  FastPk pk;   /* (built per chunk: a lane that finished meanwhile carries inert numbers in r) */ \
  pk.b12 = (v2f){r.b1, r.b2}; pk.b21 = (v2f){r.b2, r.b1};                                 \
  pk.a12 = (v2f){r.a1, r.a2}; pk.a21 = (v2f){r.a2, r.a1};                                 \
  pk.pan = (v2f){r.pan_l, r.pan_r};   /* (the block paths run only in waves without pan modulation) */ \
  fast_pk_partner(pk, r.pan_l, r.pan_r);
#define SK_FAST_PACK_OUT() { r.x1 = xx.x; r.x2 = xx.y; r.y1 = yy.x; r.y2 = yy.y; }
/* after fast_finish rewrote a finishing lane's numbers (rare): the pairs again */
#define SK_FAST_REPACK()                                                                  \
  { xx = (v2f){r.x1, r.x2}; yy = (v2f){r.y1, r.y2};                                       \
    pk.b12 = (v2f){r.b1, r.b2}; pk.b21 = (v2f){r.b2, r.b1};                               \
    pk.a12 = (v2f){r.a1, r.a2}; pk.a21 = (v2f){r.a2, r.a1};                               \
    pk.pan = (v2f){r.pan_l, r.pan_r}; fast_pk_partner(pk, r.pan_l, r.pan_r); }
// Eight steady frames of the extended frame loop (modulation exchange, finish test, sample & hold ... per frame) with
// the same tile reduction instead of 12 v_add_dpp per frame; the per-wave LDS region is free here (no table windows
// in such a wave).
#define SK_FAST_X_FRAME(Q, XN, XO, YN, YO, SWAPPED_)                                                     \
  {                                                                                                      \
    float l, rr;                                                                                         \
    SK_FAST_DRAW()                                                                                       \
    fast_frame<TAB_LDS, FILTER, ENV, true, false, INTERP, STOPS>(r, XN, XO, YN, YO, released, lds_tab, glb_tab, l, rr, xf, muted, white_); \
    l = silent ? 0.0f : l; rr = silent ? 0.0f : rr;                                                      \
    SK_FAST_STEM(Q, l, rr)                                                                               \
    if (STOPS && (xf & XF_STOP) && __any(r.fin)) fast_finish(a, r, v, dead, silent, sample_final, SWAPPED_, c0 + (Q) == a.num_frames - 1, misc_xy); \
    xt[((Q) & 7) * SK_XT + lane] = fold_lr(l, rr);                                                       \
  }
#define SK_FAST_X_BLOCK(J)                                                                               \
  {                                                                                                      \
    if (pend_j >= 0) SK_FAST_TILE_REDUCE(pend_j)                                                         \
    _Pragma("unroll") for (int q_ = 0; q_ < 8; q_ += 2) {                                                \
      SK_FAST_X_FRAME((J) + q_, r.x1, r.x2, r.y1, r.y2, true)                                            \
      SK_FAST_X_FRAME((J) + q_ + 1, r.x2, r.x1, r.y2, r.y1, false)                                       \
    }                                                                                                    \
    pend_j = (J);                                                                                        \
  }
// Eight steady frames of a wave whose only extended feature is frequency modulation the way the reference's patches write
// it (`v0 ... F1,depth` with the modulator above the carrier: last frame's voice_sample[m], synth.c:548-555).  The exchange
// makes the frames strictly sequential, but most of them are tame: when the modulated increment of every lane still lies in
// [0, span/2] (one wave-wide vote per frame) the frame runs the straight-line oscillator, the unclamped fetch and the
// pair-register chain of the plain blocks; a frame in which deep modulation drives some increment negative or beyond half a
// loop takes the general frame.  (Muted lanes -- the modulators of such patches are `m1` -- are selected away per frame.)
#define SK_FAST_FM_FRAME(Q, NEWEST_X_, XN, XO, YN, YO)                                                   \
  {                                                                                                      \
    float l, rr;                                                                                         \
    const float ms_ = __int_as_float(__builtin_amdgcn_ds_bpermute(r.fm_addr, __float_as_int(r.sample))); \
    const float inc_ = r.fm_addr >= 0 ? r.inc + r.fm_k * (ms_ * r.fm_depth) : r.inc;      /* synth.c:551-554 */ \
    if (__all(inc_ >= 0.0f && inc_ <= half_span)) {                                                      \
      const float s_ = fast_fetch<TAB_LDS, INTERP, true>(lds_tab, glb_tab, r, fast_advance<true, false>(r, inc_)); \
      fast_post_v<FILTER, ENV, false, STOPS, NEWEST_X_>(r, pk, s_, xx, yy, l, rr, xf);                   \
    } else {                                                                                             \
      SK_FAST_PACK_OUT()                                                                                 \
      fast_frame<TAB_LDS, FILTER, ENV, true, false, INTERP, STOPS>(r, XN, XO, YN, YO, released, lds_tab, glb_tab, l, rr, xf, muted, 0.0f); \
      SK_FAST_REPACK()                                                                                   \
    }                                                                                                    \
    l = silent ? 0.0f : l; rr = silent ? 0.0f : rr;                                                      \
    xt[((Q) & 7) * SK_XT + lane] = fold_lr(l, rr);                                                       \
  }
#define SK_FAST_FM_BLOCK(J)                                                                              \
  {                                                                                                      \
    if (pend_j >= 0) SK_FAST_TILE_REDUCE(pend_j)                                                         \
    _Pragma("unroll") for (int q_ = 0; q_ < 8; q_ += 2) {                                                \
      SK_FAST_FM_FRAME((J) + q_, true, r.x1, r.x2, r.y1, r.y2)                                           \
      SK_FAST_FM_FRAME((J) + q_ + 1, false, r.x2, r.x1, r.y2, r.y1)                                      \
    }                                                                                                    \
    pend_j = (J);                                                                                        \
  }
// ---- skewed blocks (round 4, SKRED_OPT_FM_SKEW): previous-frame modulation WITHOUT a per-frame exchange ----
// The modulators of the shipped patches (`v3 w0 f5 a5 m1`) are heard by nobody and read nobody below them, so nothing they
// render depends on their carriers: the per-frame ds_bpermute of SK_FAST_FM_FRAME / fast_frame only exists because all lanes of a
// wave walk the frames in lock step.  Here a lane that is read by others runs 8-FRAME BLOCKS AHEAD of them: `lead` blocks,
// one more than every lane that reads it (3.sk, 1.sk, 37.sk: modulators 1, carriers 0; 7.sk's chain v2 -> v1 -> v0: 2, 1, 0).
// In every step a lane renders the block `lead` ahead of the step's and leaves its eight voice_sample values in the wave's LDS
// ring (ring[q][lane]); a reader takes, at the top of its step, the eight values its source left there in the step before --
// exactly the block it is about to render -- plus, from row 8, the last value of the block before (frame q's increment, gain
// and pan take the modulator's sample of frame q - 1, synth.c:551,586,599).  All of a step's modulation is known at its
// top: the oscillators run eight gathers deep and the tameness vote is one per block.  The lanes meet again before anything
// that counts frames for the whole wave: EXEC-masked lead-in steps open the first skewed step of a pass (the deepest sources
// first), and on the launch's last whole blocks a lane that has rendered them already gets its recurrences put back behind
// the step (`lead > blocks left`).
// Condition (skew_ok, per wave and pass): every lane that is read by another is silent (its (L, R) are zeros whatever frame it
// is on, so the tile rows stay those of the audible lanes' frames; no probe row) -- or it is one block ahead with its pan at
// rest, and its (L, R) are formed a step late from its own ring column (`skew_delay`) --, every source is exactly one block ahead of
// each of its readers, chains at most SK_SKEW_LMAX deep, geometry tame, no reverse / noise / stopping / smoother-off lanes.
// Two forms: LEAN (only frequency modulation, one level: the pair-register chain of the plain blocks, no per-frame feature
// tests) and RICH (amplitude / pan modulation, sample & hold, chains: fast_frame itself with the ring's samples handed in).
// Same products and sums per voice as the exchange forms; a step whose vote fails takes straight-line frames with both wraps when
// its increments stay within half a loop length on either side (fast_frame<BIDIR>), the general frames otherwise.
#define SK_SKEW_RING (9 * 64)    /* floats per wave: rows 0..7 the lanes' samples of their latest block, row 8 the last sample of the block
                                    before it (what frame 0 of a reader's block takes).  (A tenth row cost 3.sk's bank -- 32 KB of
                                    tables -- its third workgroup per CU: 0.38 -> 0.52 ms; the leads live in the pad floats of
                                    the reduction tile's rows, a byte per lane: SK_SKEW_LEAD) */
#define SK_SKEW_LEAD() (reinterpret_cast<unsigned char *>(xt + (lane >> 4) * SK_XT + 64 + ((lane >> 2) & 3))[lane & 3])
#define SK_SKEW_COL(ADDR_) ((ADDR_) >= 0 ? ((ADDR_) >> 2) : lane)   /* the ring column a lane reads: its source's, or (unused) its own */
#define SK_SKEW_LMAX 3
#define SK_FAST_SKEW_RFRAME(Q, MODE_, XN, XO, YN, YO, TILE_, XF_, DL_)     /* MODE_ 0: general frame, 1: tame, 2: bidirectional, 3: tame and no loop windows (LOZ) */ \
  {                                                                                                      \
    float l, rr;                                                                                         \
    /* (amplitude / pan sources: read when the frame needs them -- nobody writes the ring before the end of the step) */ \
    const float aq_ = !ap_ ? 0.0f : ring[((Q) == 0 ? 8 : (Q) - 1) * 64 + SK_SKEW_COL(r.am_addr)];        \
    const float pq_ = !ap_ ? 0.0f : ring[((Q) == 0 ? 8 : (Q) - 1) * 64 + SK_SKEW_COL(r.pm_addr)];        \
    fast_frame<TAB_LDS, FILTER, ENV, true, ((MODE_) & 1) != 0, INTERP, STOPS, true, (MODE_) != 0, (MODE_) == 2, (MODE_) == 3>(r, XN, XO, YN, YO, released, lds_tab, glb_tab, l, rr, XF_, muted, 0.0f, mq_[Q], aq_, pq_); \
    own_[Q] = r.sample;                                                                                  \
    if (TILE_) {                                                                                         \
      if ((DL_) && skew_delay) {     /* an audible source: what it rendered a step ago belongs to THIS frame (same product as fast_post's) */ \
        const float sd_ = ring[(Q) * 64 + lane];                                                         \
        l = dl_ ? sd_ * r.pan_l : l; rr = dl_ ? sd_ * r.pan_r : rr;                                      \
      }                                                                                                  \
      l = silent ? 0.0f : l; rr = silent ? 0.0f : rr;                                                    \
      xt[(Q) * SK_XT + lane] = fold_lr(l, rr);                                                           \
    }                                                                                                    \
  }
/* the top of a step: this block's modulator samples (frame q takes the source's sample of frame q - 1) and the vote.  An
   increment in [+0, span/2] is, as an unsigned integer, at most the bits of span/2 (negative numbers carry the sign bit, NaNs
   sit above every finite number), so the largest of the eight patterns decides for all of them. */
#define SK_FAST_SKEW_TOP()                                                                               \
    float mq_[8];                                                                                        \
    SK_FAST_WAVE_SYNC()                                                                                  \
    {                                                                                                    \
      const float *col_ = ring + SK_SKEW_COL(r.fm_addr);                                                 \
      mq_[0] = col_[8 * 64];                                                                             \
      _Pragma("unroll") for (int q_ = 1; q_ < 8; ++q_) mq_[q_] = col_[(q_ - 1) * 64];                    \
    }                                                                                                    \
    float inc_[8];                                                                                       \
    uint32_t top_ = 0u;                                                                                  \
    _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) {                                                   \
      inc_[q_] = r.fm_addr >= 0 ? r.inc + r.fm_k * (mq_[q_] * r.fm_depth) : r.inc;      /* synth.c:551-554 */ \
      top_ = max(top_, __float_as_uint(inc_[q_]));                                                       \
    }                                                                                                    \
    const bool tame_ = __all(top_ <= __float_as_uint(half_span));                                        \
    /* ... or within half a loop length on either side (deep modulation: the increment goes negative): fast_frame<BIDIR>; voted
       only when the first vote failed */                                                                \
    bool bidir_ = false;                                                                                 \
    if (!tame_) {                                                                                        \
      uint32_t mag_ = 0u;                                                                                \
      _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) mag_ = max(mag_, __float_as_uint(inc_[q_]) & 0x7fffffffu); \
      bidir_ = __all(mag_ <= __float_as_uint(half_span));                                                \
    }                                                                                                    \
    (void)inc_; (void)bidir_;
/* the general frames of a step (delay line in r.x1 ...); the lane's own eight samples go to the ring when every lane of the wave
   has read what it needs of this step */
#define SK_FAST_SKEW_FRAMES(MODE_, TILE_, XF_, DL_)     /* DL_: the copy that also serves audible sources (skew_delay) */ \
    {                                                                                                    \
      const bool ap_ = ((XF_) & XF_AP) != 0;                                                             \
      const bool dl_ = (TILE_) && (DL_) && skew_delay && (int)SK_SKEW_LEAD() == 1 && !silent;            \
      (void)dl_;                                                                                         \
      float own_[8];                                                                                     \
      _Pragma("unroll") for (int q_ = 0; q_ < 8; q_ += 2) {                                              \
        SK_FAST_SKEW_RFRAME(q_, MODE_, r.x1, r.x2, r.y1, r.y2, TILE_, XF_, DL_)                          \
        SK_FAST_SKEW_RFRAME(q_ + 1, MODE_, r.x2, r.x1, r.y2, r.y1, TILE_, XF_, DL_)                      \
      }                                                                                                  \
      const float old7_ = ring[7 * 64 + lane];                                                           \
      SK_FAST_WAVE_SYNC()                                                                                \
      ring[8 * 64 + lane] = old7_;                                                                       \
      _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_) ring[q_ * 64 + lane] = own_[q_];                  \
    }
/* RICH step (delay line in r.x1 ...): fast_frame on both sides of the vote */
#define SK_FAST_SKEW_RSTEP(J, TILE_)                                                                     \
  {                                                                                                      \
    if (TILE_) { if (pend_j >= 0) SK_FAST_TILE_REDUCE(pend_j) }                                          \
    SK_FAST_SKEW_TOP()                                                                                   \
    if (tame_ && (TILE_)) {           /* (the lead-in steps run once per pass: the general frames only) */ \
      /* the shipped shape -- frequency plus amplitude / pan modulation, nothing else -- with the feature mask a literal: the
         frames carry no wave-uniform tests (one scheduling region per frame pair) */                    \
      if (xf == (XF_FM | XF_AP) && !skew_delay) { if (loz) SK_FAST_SKEW_FRAMES(3, TILE_, (XF_FM | XF_AP), false) else SK_FAST_SKEW_FRAMES(1, TILE_, (XF_FM | XF_AP), false) } \
      else SK_FAST_SKEW_FRAMES(1, TILE_, xf, true)                                                       \
    } else if (bidir_ && (TILE_)) {                                                                      \
      if (xf == (XF_FM | XF_AP) && !skew_delay) SK_FAST_SKEW_FRAMES(2, TILE_, (XF_FM | XF_AP), false)    \
      else SK_FAST_SKEW_FRAMES(2, TILE_, xf, true)                                                       \
    } else {                                                                                             \
      SK_FAST_SKEW_FRAMES(0, TILE_, xf, true)                                                            \
    }                                                                                                    \
    if (TILE_) pend_j = (J);                                                                             \
  }
/* LEAN step (delay line in the register pairs xx / yy).  LOZ_: no lane of the wave has a loop window (fast_advance<LOZ>);
   STALL_: no smoother of the wave moves any more (per chunk) */
#define SK_FAST_SKEW_LEAN_BODY(LOZ_, STALL_, DL_)     /* DL_: the copy that also serves audible sources (skew_delay) */ \
    {                                                                                                    \
      float s_[8];                                                                                       \
      const bool dl_ = (DL_) && (int)SK_SKEW_LEAD() == 1 && !silent;                                     \
      (void)dl_;                                                                                         \
      ring[8 * 64 + lane] = ring[7 * 64 + lane];      /* (the lane's own column: readers took row 8 at the top) */ \
      _Pragma("unroll") for (int q_ = 0; q_ < 8; ++q_)                                                   \
        s_[q_] = fast_fetch<TAB_LDS, INTERP, true>(lds_tab, glb_tab, r, fast_advance<true, false, LOZ_>(r, inc_[q_])); \
      _Pragma("unroll") for (int q_ = 0; q_ < 8; q_ += 2) {                                              \
        float s0_, s1_, u_, f0_, f1_;                                                                    \
        float d0_ = 0.0f, d1_ = 0.0f;      /* (an audible source: what it rendered a step ago belongs to THESE frames) */ \
        if (DL_) { d0_ = ring[q_ * 64 + lane]; d1_ = ring[(q_ + 1) * 64 + lane]; }                       \
        fast_post_v<FILTER, ENV, STALL_, STOPS, true, false>(r, pk, s_[q_], xx, yy, s0_, u_, 0);         \
        ring[q_ * 64 + lane] = s0_;                                                                      \
        fast_post_v<FILTER, ENV, STALL_, STOPS, false, false>(r, pk, s_[q_ + 1], xx, yy, s1_, u_, 0);    \
        ring[(q_ + 1) * 64 + lane] = s1_;                                                                \
        if (DL_) { s0_ = dl_ ? d0_ : s0_; s1_ = dl_ ? d1_ : s1_; }                                       \
        s0_ = silent ? 0.0f : s0_; s1_ = silent ? 0.0f : s1_;                                            \
        fast_pan_fold2(s0_, s1_, pk.pan.x, pk.pan.y, f0_, f1_);                                          \
        xt[q_ * SK_XT + lane] = f0_;                                                                     \
        xt[(q_ + 1) * SK_XT + lane] = f1_;                                                               \
      }                                                                                                  \
    }
#define SK_FAST_SKEW_STEP(J)                                                                             \
  {                                                                                                      \
    if (pend_j >= 0) SK_FAST_TILE_REDUCE(pend_j)                                                         \
    SK_FAST_SKEW_TOP()                                                                                   \
    SK_FAST_WAVE_SYNC()                                                                                  \
    if (tame_) {                                                                                         \
      if (skew_delay) SK_FAST_SKEW_LEAN_BODY(false, false, true)                                         \
      else if (loz) { if (stall_) SK_FAST_SKEW_LEAN_BODY(true, true, false) else SK_FAST_SKEW_LEAN_BODY(true, false, false) } \
      else { if (stall_) SK_FAST_SKEW_LEAN_BODY(false, true, false) else SK_FAST_SKEW_LEAN_BODY(false, false, false) } \
    } else {                                                                                             \
      SK_FAST_PACK_OUT()                                                                                 \
      if (bidir_ && !skew_delay) SK_FAST_SKEW_FRAMES(2, true, XF_FM, false)                              \
      else SK_FAST_SKEW_FRAMES(0, true, xf, true)                                                        \
      SK_FAST_REPACK()                                                                                   \
    }                                                                                                    \
    pend_j = (J);                                                                                        \
  }
/* where the skew begins: row 8 takes voice_sample as the frame before left it, then the lead-in steps -- the lanes that must be
   `s_` blocks ahead render a block on their own, the deepest sources first */
#define SK_FAST_SKEW_BEGIN()                                                                             \
  {                                                                                                      \
    SK_FAST_WAVE_SYNC()                                                                                  \
    ring[8 * 64 + lane] = r.sample;       /* what frame 0 of a reader's first block takes ... */           \
    ring[7 * 64 + lane] = r.sample;       /* ... also where a lane's first flush looks for it (row 7 -> row 8) */ \
    SK_FAST_WAVE_SYNC()                                                                                  \
    for (int s_ = lmax; s_ >= 1; --s_) {                                                                 \
      if ((int)SK_SKEW_LEAD() >= s_) SK_FAST_SKEW_RSTEP(0, false)                                        \
    }                                                                                                    \
    skewed = true;                                                                                       \
  }
/* all whole blocks of a chunk.  A lane `lead` blocks ahead has rendered the launch's last `lead` whole blocks already: its state
   is final when the step `lead` blocks before the end begins.  It goes to the state planes there (FREEZE: what the end of the
   pass stores), the lane keeps running on numbers nobody reads (silent; its ring column has no reader left), and behind the last
   step it loads its state again (THAW) -- no registers held across the steps for this. */
#ifdef SK_PROBE_TU   /* (an audible source's probe rows: none while it runs on behind its last block, the pointer back for the tail frames) */
#define SK_SKEW_PROBE_HOLD() { r.probe_hold = r.probe; r.probe = nullptr; }
#define SK_SKEW_PROBE_BACK() { r.probe = r.probe_hold; }
#else
#define SK_SKEW_PROBE_HOLD()
#define SK_SKEW_PROBE_BACK()
#endif
#define SK_FAST_SKEW_FREEZE()                                                                            \
  if ((int)SK_SKEW_LEAD() == left_ + 1 && !dead) {                                                       \
    uint4 s0, s1;                                                                                        \
    s0.x = __float_as_uint(r.phase); s0.y = __float_as_uint(r.sgain);                                    \
    s0.z = __float_as_uint(r.filt ? r.x1 : r.ox1); s0.w = __float_as_uint(r.filt ? r.x2 : r.ox2);        \
    s1.x = __float_as_uint(r.filt ? r.y1 : r.oy1); s1.y = __float_as_uint(r.filt ? r.y2 : r.oy2);        \
    s1.z = __float_as_uint(r.sample); s1.w = r.rw;                                                       \
    *reinterpret_cast<uint4 *>(&a.rw[SKS_OSC][v]) = s0;                                                  \
    *reinterpret_cast<uint4 *>(&a.rw[SKS_FILT][v]) = s1;                                                 \
    if (r.hold_max) *reinterpret_cast<uint2 *>(&a.rw[SKS_MISC][v]) = make_uint2(__float_as_uint(r.hold), (uint32_t)r.hold_count); \
    SK_SKEW_PROBE_HOLD()                                                                                 \
  }
#define SK_FAST_SKEW_THAW()                                                                              \
  {                                                                                                      \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");                                               \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");                                               \
    if (SK_SKEW_LEAD() > 0 && !dead) {                                                                   \
      const uint4 s0 = *reinterpret_cast<const uint4 *>(&a.rw[SKS_OSC][v]);                     \
      const uint4 s1 = *reinterpret_cast<const uint4 *>(&a.rw[SKS_FILT][v]);                    \
      r.phase = __uint_as_float(s0.x); r.sgain = __uint_as_float(s0.y);                                  \
      r.x1 = __uint_as_float(s0.z);    r.x2 = __uint_as_float(s0.w);                                     \
      r.y1 = __uint_as_float(s1.x);    r.y2 = __uint_as_float(s1.y);                                     \
      r.sample = __uint_as_float(s1.z);                                                                  \
      SK_SKEW_PROBE_BACK()                                                                               \
      if (r.hold_max) {                                                                                  \
        const uint2 s2 = *reinterpret_cast<const uint2 *>(&a.rw[SKS_MISC][v]);                  \
        r.hold = __uint_as_float(s2.x); r.hold_count = (int)s2.y;                                        \
      }                                                                                                  \
    }                                                                                                    \
  }
/* OUT_ / IN_: the LEAN form keeps the delay line in the register pairs xx / yy around the steps (SK_FAST_PACK_OUT / _REPACK) */
#define SK_FAST_SKEW_CHUNK(STEP_, OUT_, IN_)                                                             \
  for (; j + 8 <= cn; j += 8) {                                                                          \
    if (!skewed) { OUT_ SK_FAST_SKEW_BEGIN() IN_ }                                                       \
    const int left_ = ((a.num_frames - (c0 + j)) >> 3) - 1;        /* whole blocks of the launch behind this one */ \
    if (left_ < lmax) { OUT_ SK_FAST_SKEW_FREEZE() }                                                     \
    STEP_                                                                                                \
    if (left_ == 0) {                                                                                    \
      OUT_                                                                                               \
      SK_FAST_SKEW_THAW()                                                                                \
      IN_                                                                                                \
      skewed = false;                                                                                    \
    }                                                                                                    \
  }
// eight steady frames of a tame wave of a global-table bank through the table window
#define SK_FAST_WIN_BLOCK_(J, STALL_, RAMP_, NOISE_)                                                                  \
  {                                                                                                      \
    FastWin w_;                                                                                          \
    fast_win_fill(r, dead, w_, win, lane, glb_tab);                                                      \
    _Pragma("unroll") for (int q_ = 0; q_ < 8; q_ += 2) {                                                \
      float l0, r0, l1, r1;                                                                              \
      const float s0_ = fast_fetch_win<INTERP, STOPS>(r, w_, win, lane, glb_tab, fast_advance<true, STOPS>(r)); \
      fast_post_v<FILTER, ENV, STALL_, STOPS, true, true, RAMP_>(r, pk, SK_FAST_BLOCK_SAMPLE(s0_, NOISE_), xx, yy, l0, r0, xf_blk, &ev_);                       \
      if (STOPS && __any(r.fin)) { SK_FAST_PACK_OUT() fast_finish(a, r, v, dead, silent, sample_final, true, false, misc_xy); SK_FAST_REPACK() } \
      const float s1_ = fast_fetch_win<INTERP, STOPS>(r, w_, win, lane, glb_tab, fast_advance<true, STOPS>(r)); \
      fast_post_v<FILTER, ENV, STALL_, STOPS, false, true, RAMP_>(r, pk, SK_FAST_BLOCK_SAMPLE(s1_, NOISE_), xx, yy, l1, r1, xf_blk, &ev_);                       \
      if (STOPS && __any(r.fin)) { SK_FAST_PACK_OUT() fast_finish(a, r, v, dead, silent, sample_final, false, c0 + (J) + q_ + 1 == a.num_frames - 1, misc_xy); SK_FAST_REPACK() } \
      xt[q_ * SK_XT + lane] = fold_lr(l0, r0);     /* (global-table banks: the tile has its own LDS behind the windows) */ \
      xt[(q_ + 1) * SK_XT + lane] = fold_lr(l1, r1);                                                     \
    }                                                                                                    \
    SK_FAST_TILE_REDUCE(J)                                                                               \
  }
#define SK_FAST_WIN_BLOCK(J, STALL_) SK_FAST_WIN_BLOCK_(J, STALL_, false, false)
// after an EVEN frame the newest delay-line entries sit in x2 / y2 (roles swapped), after an ODD one in x1 / y1
#define SK_FAST_EVEN(J, STEADY_) SK_FAST_FRAME(J, STEADY_, r.x1, r.x2, r.y1, r.y2, true)
#define SK_FAST_ODD(J, STEADY_) SK_FAST_FRAME(J, STEADY_, r.x2, r.x1, r.y2, r.y1, false)
/* (only where the biquad runs: in an unfiltered bank the frames never touch the delay line -- mmf_process is skipped,
   synth.c:577 -- and whatever it holds goes back as loaded) */
#define SK_FAST_FIX_ODD_TAIL()                                                     \
  if (FILTER) { float t_ = r.x1; r.x1 = r.x2; r.x2 = t_; t_ = r.y1; r.y1 = r.y2; r.y2 = t_; }

#ifndef SK_FAST_WIN_MIN_WAVES
#define SK_FAST_WIN_MIN_WAVES 4  /* global-table banks: the window refill wants ~20 more registers (6: 12 B of scratch) */
#endif
#ifndef SK_FAST_WIN_EXT_MIN_WAVES
#define SK_FAST_WIN_EXT_MIN_WAVES 4  /* the extended instantiation on global-table banks */
#endif
#ifndef SK_FAST_EXT_MIN_WAVES
#define SK_FAST_EXT_MIN_WAVES 3  /* the extended instantiation (LDS-table banks) carries ~30 more per-lane fields: at 4 waves (128
                                    VGPRs) it spills inside the frame loops, and its LDS footprint allows 3 workgroups per CU in most
                                    banks anyway: one-shot bank 0.46 -> 0.40 ms, FM 1.31 -> 1.22 ms.  PCM banks keep 4: there the
                                    occupancy hides the window refills (0.347 -> 0.372 ms with 3) */
#endif
#ifndef SK_FAST_EXT_BARE_MIN_WAVES
#define SK_FAST_EXT_BARE_MIN_WAVES 4   /* the extended LDS-table instantiation without biquad / envelope, truncating lookup (see the kernel) */
#endif
#ifndef SK_FAST_MIN_WAVES
#define SK_FAST_MIN_WAVES 4      /* waves per SIMD the register allocator must leave room for (LDS-table banks: the table copy,
                                    wsum and the reduction tiles take 38..70 KB per workgroup, i.e. 2..4 workgroups per CU anyway) */
#endif
// STOPS (the "extended" instantiation): the bank holds forward one-shots that play to their table end and finish
// (checked frame by frame) and / or carriers frequency-modulated by a higher-indexed voice of their 64-voice group.
// Such banks run the plain frame loop (no frame pairs); table windows only in waves without carriers.
// RAMPK: the instantiation that also holds the block form of envelopes in motion (fast_env_span2).  It is a separate
// instantiation because its registers perturb the allocation of the steady loops (+4..6 % per frame on C1 / C2 / the
// 2^17-voice shard when both lived in one kernel): the host launches it while envelopes may be moving and the lean one
// once a launch has reported that none did (the general frames are in both, so the choice only decides speed).
// PROBE: the same kernel compiled with -DSK_PROBE_TU (fast_post / fast_post_v then also write the probe rows of
// skred_bank_set_probe): a template parameter only so that those instantiations are symbols of their own.
template <bool TAB_LDS, bool FILTER, bool ENV, int INTERP, bool STOPS, bool RAMPK = false, bool PROBE = false>
// (the extended LDS-table instantiation WITHOUT biquad and envelope, truncating lookup -- what every shipped patch but 18.sk runs on -- needs 129
// registers, three of them holding spilled SGPRs: bounded to 128 it keeps four waves per SIMD where the pool leaves room for
// four workgroups per CU -- 7.sk, 16 KB of tables: 1.29 -> 1.00 ms)
__global__ __launch_bounds__(SK_GROUP, TAB_LDS ? (STOPS ? ((FILTER || ENV || INTERP != 0) ? SK_FAST_EXT_MIN_WAVES : SK_FAST_EXT_BARE_MIN_WAVES) : SK_FAST_MIN_WAVES) : (STOPS ? SK_FAST_WIN_EXT_MIN_WAVES : SK_FAST_WIN_MIN_WAVES)) void sk_render_fast_kernel(const sk_render_args_t a) {
  extern __shared__ float lds[];
  float2 *wsum = reinterpret_cast<float2 *>(lds + (TAB_LDS ? a.lds_table_floats : 0));
  const char *lds_tab = reinterpret_cast<const char *>(lds);
  const char *glb_tab = reinterpret_cast<const char *>(a.tables);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  constexpr bool SK_PAIRED_ = false;   // (the tile layout of the block macros: see SK_FAST_LDS_CHUNK)
  constexpr bool SK_LOZ_ = false;      // (the block macros' wrap form: shadowed where a wave has proved lo == 0, see fast_advance<LOZ>)
  // per wave behind the chunk sums: LDS-table banks -- the reduction tile xt[8][SK_XT]; global-table banks -- the table window
  // (SK_WIN * 64 floats), the tiles behind all four windows
  float *win = reinterpret_cast<float *>(wsum + 4 * SK_CHUNK) + wave * (TAB_LDS ? 8 * SK_XT : SK_WIN * 64);
  // the reduction tile of the 8-frame blocks (8 * SK_XT floats): LDS-table banks keep it in the same per-wave region
  // (no windows there), global-table banks behind the four windows
  float *xt = TAB_LDS ? win : reinterpret_cast<float *>(wsum + 4 * SK_CHUNK) + 4 * (SK_WIN * 64) + wave * (8 * SK_XT);
  // (extended LDS-table instantiation, a.fm_skew) the wave's sample ring of the skewed blocks, behind the four tiles
  float *ring = reinterpret_cast<float *>(wsum + 4 * SK_CHUNK) + 4 * (8 * SK_XT) + wave * SK_SKEW_RING;
  (void)win; (void)xt; (void)ring;
  const int bid = (int)blockIdx.x - a.wg_shift;        // row of the partial mix; -1: the gain workgroup (sk_finish_block)
  if (bid < 0) { sk_finish_block(a, bid, tid, SK_GROUP, reinterpret_cast<int *>(lds)); return; }

  if (TAB_LDS) {
    const int n4 = a.lds_table_floats >> 2;           // padded to a multiple of 4 by the host
    const float4 *src4 = reinterpret_cast<const float4 *>(a.tables);
    float4 *dst4 = reinterpret_cast<float4 *>(lds);
    sk_stage_tables<SK_GROUP>(src4, dst4, n4, tid);
    __syncthreads();
  }

  const size_t part_base = (size_t)bid * (size_t)a.num_frames * 2;
  bool first_pass = true;
  const bool stems_on = a.stems != nullptr;          // (launch-uniform: one scalar branch per frame)

  // packed lanes (sparse banks, extended instantiations only: sk_render_args_t, pack_mask): a wave holds the voices that can
  // sound of several 64-voice groups (an empty lane -- `absent` -- is a dead voice that loads voice 0 and stores nothing)
  const bool packed = STOPS && a.pack_shift < 6;
  const int n_pass = packed ? a.pack_passes : a.n_groups;
  for (int g = bid; g < n_pass; g += a.n_rows) {
    int v = g * SK_GROUP + tid;
    uint64_t pmask = 0;                                // (packed) the lane's group word, and the voice's lane in its own group
    int ppos = lane;
    bool absent = false;
    if (packed) { v = sk_packed_voice(a, g * (SK_GROUP / 64) + wave, lane, pmask, ppos); absent = v < 0; if (absent) { v = 0; ppos = 0; } }
    const bool publish = a.finish && g + a.n_rows >= n_pass;   // the pass that completes this workgroup's row
    FastRegs r;
    bool dead, silent;            // dead: skipped by synth.c:531-542; silent: dead or muted
    bool muted = false;           // voice_disconnect
    uint2 misc_xy = make_uint2(0u, 0u);   // (EXT) sample&hold words of the MISC plane, kept for a pan-modulated store
    bool released = false;
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
      if (FILTER) {
        const uint4 fl = *reinterpret_cast<const uint4 *>(&a.ro[SKP_FILT][v]);
        r.b2 = __uint_as_float(fl.x); r.a1 = __uint_as_float(fl.y); r.a2 = __uint_as_float(fl.z);
      }
      if (ENV) {
        const uint4 et = *reinterpret_cast<const uint4 *>(&a.ro[SKP_ENV_T][v]);
        const uint4 es = *reinterpret_cast<const uint4 *>(&a.ro[SKP_ENV_S][v]);
        r.att = __uint_as_float(et.x); r.dec = __uint_as_float(et.y);
        r.sus = __uint_as_float(et.z); r.rel = __uint_as_float(et.w);
        r.attdec = r.att + r.dec;                    // synth.c:410: decay_start + decay_time
        r.one_m_sus = 1.0f - r.sus;                  // synth.c:413
        r.gain_sustain = r.amp * (r.sus * r.vel);    // synth.c:582,588 in the sustain stage
        t_start = ((uint64_t)es.y << 32) | es.x;
        t_release = ((uint64_t)es.w << 32) | es.z;
        released = t_release != 0;                   // synth.c:417
      }
      dead = absent || (r.rw & SKR_FINISHED) || r.amp == 0.0f || (flags & SKF_INERT);
      silent = dead || (flags & SKF_MUTED);
      r.stop = STOPS && (flags & SKF_ONE_SHOT) && !(flags & SKF_LOOPING);
      r.fin = false;
      r.hi_stop = r.hi - 1e-6f;                       // synth.c:243
      r.fm_addr = -1; r.fm_k = 0.0f; r.fm_depth = 0.0f;
      r.am_addr = -1; r.pm_addr = -1; r.am_depth = 0.0f; r.pm_depth = 0.0f; r.am_prev = 0.0f; r.pm_prev = 0.0f;
      r.pan_dirty = false;
      r.filt = !STOPS || (flags & SKF_FILTER);
      r.use_env = !STOPS || (flags & SKF_USE_ENV);
      r.ox1 = r.x1; r.ox2 = r.x2; r.oy1 = r.y1; r.oy2 = r.y2;
      if (ENV && !r.use_env) { r.gain_sustain = r.amp; released = false; }
      r.rev = STOPS && (flags & SKF_REVERSE);
      r.hold_max = STOPS ? (int)(tab.w >> 8) : 0;
      r.quant = STOPS ? (int)(tab.w & 0xFFu) : 0;
      r.hold = __uint_as_float(s2.x); r.hold_count = (int)s2.y;
      r.nosmooth = STOPS && !(flags & SKF_SMOOTH);
      r.noise = STOPS && (flags & SKF_NOISE);
      r.ophase = r.phase;
      muted = (flags & SKF_MUTED) != 0;
#ifdef SK_PROBE_TU
      r.probe_stride = a.n_probe;
      r.probe = sk_probe_row(a, v, silent || v >= a.n_voices);
      r.probe_any = __any(r.probe != nullptr);
#endif
      if (STOPS && (a.fast_mode & SKM_FM)) {
        const uint4 mi = *reinterpret_cast<const uint4 *>(&a.ro[SKP_MODI][v]);
        const uint4 mf = *reinterpret_cast<const uint4 *>(&a.ro[SKP_MODF][v]);
        int fm_lane = (int)mi.x;
        int am_lane = (int)mi.y, pm_lane = (int)mi.z;
        if (absent) fm_lane = am_lane = pm_lane = -1;
        if (packed) {                                      // the planes number modulators by their lane in the 64-voice group
          if (fm_lane >= 0) fm_lane = sk_packed_lane(a, pmask, lane, fm_lane);
          if (am_lane >= 0) am_lane = am_lane == ppos ? lane : sk_packed_lane(a, pmask, lane, am_lane);
          if (pm_lane >= 0) pm_lane = pm_lane == ppos ? lane : sk_packed_lane(a, pmask, lane, pm_lane);
        }
        if (am_lane >= 0) { r.am_addr = am_lane == lane ? -2 : am_lane << 2; r.am_depth = __uint_as_float(mf.z); }
        if (pm_lane >= 0) { r.pm_addr = pm_lane == lane ? -2 : pm_lane << 2; r.pm_depth = __uint_as_float(mf.w); }
        misc_xy = make_uint2(s2.x, s2.y);
        // voice_phase_inc[m], whatever m's own state is (read before a skipped lane's numbers are neutralised)
        const float inc_m = __int_as_float(__builtin_amdgcn_ds_bpermute(max(fm_lane, 0) << 2, __float_as_int(r.inc)));
        if (fm_lane >= 0) {
          r.fm_addr = fm_lane << 2;
          r.fm_k = inc_m * __uint_as_float(mf.y);
          r.fm_depth = __uint_as_float(mf.x);
        }
      }
    }
    bool sample_final = false;                         // (STOPS) the voice finished in this launch: its planes are final
    // wrap can only ever be the simple one (see fast_frame<TAME>): decided once per pass
    if (dead) {
      // a skipped voice is never stored back (see the end of the pass): give its lane inert numbers
      // so that it contributes exact zeros and its table index stays at 0, whatever its real state is
      r.inc = 0.0f; r.lo = 0.0f; r.hi = 1.0f; r.span = 1.0f; r.span2 = 2.0f; r.phase = 0.0f;
      r.toff4 = 0; r.tsize_m1 = 0;
      r.k = 0.0f; r.sgain = 0.0f; r.amp = 0.0f; r.gain_sustain = 0.0f;
      r.b0 = r.b1 = r.b2 = r.a1 = r.a2 = 0.0f; r.x1 = r.x2 = r.y1 = r.y2 = 0.0f;
      r.pan_l = r.pan_r = 0.0f; r.rw &= ~SKR_ENV_ACTIVE;
      r.stop = false; r.fm_addr = -1; r.am_addr = -1; r.pm_addr = -1; r.rev = false;
      r.hold_max = 0; r.quant = 0; r.nosmooth = false; r.noise = false;
    }
    if (STOPS && r.noise) {
      // osc_next is never called for a noise voice (synth.c:543-546: no phase advance, no finish, frequency modulation
      // ignored): its oscillator idles on inert numbers; everything after the oscillator applies as usual
      r.inc = 0.0f; r.lo = 0.0f; r.hi = 1.0f; r.span = 1.0f; r.span2 = 2.0f; r.phase = 0.0f;
      r.toff4 = 0; r.tsize_m1 = 0;
      r.stop = false; r.fm_addr = -1; r.rev = false;
    }
    // lanes with any extended feature but stopping (modulated, reversed, sample & hold, crush, smoother off, noise) in
    // this wave: no table windows, no tame shortcuts, no stalled-smoother skip
    int xf = 0;                                            // XF_*: wave-uniform
    if (STOPS) {
      if (__any(r.fm_addr >= 0)) xf |= XF_FM;
      if (__any(r.am_addr != -1 || r.pm_addr != -1)) xf |= XF_AP;
      if (__any(r.rev)) xf |= XF_REV;
      if (__any(r.hold_max != 0 || r.quant != 0)) xf |= XF_HOLDQ;
      if (__any(r.nosmooth)) xf |= XF_NOSMOOTH;
      if (__any(r.noise)) xf |= XF_NOISE;
      if (__any(r.stop)) xf |= XF_STOP;                    // (a lane that finishes turns its flag off: the bit stays, harmless)
    }
    // features the block paths do not serve (modulation exchange, reverse, the noise source): such a wave walks frame by
    // frame; sample & hold, bit-crush and smoother-off ride the block paths (fast_post_v), stopping voices while far from their end
    const bool any_fm = (xf & ~(XF_STOP | XF_HOLDQ | XF_NOSMOOTH | XF_NOISE)) != 0;   // (what keeps a wave off the block paths)
    const bool any_stop = (xf & XF_STOP) != 0;
    uint64_t rng = a.rng0;                                // noise LCG state before the first frame of the launch
    (void)any_stop; (void)any_fm; (void)rng;
    // TAME (decided once per pass): the only wrap that can occur is the simple one and the table index
    // needs no clamp -- see fast_frame<TAME> / fast_fetch<NOCLAMP>
    const bool tame_geom = __all(dead || (r.inc >= 0.0f && r.inc <= 0.5f * r.span && r.phase >= r.lo && r.phase <= r.hi &&
                                          r.lo >= 0.0f && r.hi <= (float)(r.tsize_m1 + 1)));
    const bool tame = tame_geom && !__any(silent && !dead);
    // a wave whose only reason for the frame loop is frequency modulation from above (SK_FAST_FM_BLOCK)
    const bool fm_only = STOPS && TAB_LDS && tame_geom && (xf & XF_FM) && (xf & ~(XF_FM | XF_HOLDQ | XF_NOSMOOTH)) == 0;
    const float half_span = 0.5f * r.span;
    (void)fm_only; (void)half_span;
    const bool loz = __all(dead || r.lo == 0.0f);     // (wave-uniform) no lane has a loop window: fast_advance<LOZ>
    (void)loz;
    // skewed blocks (SK_FAST_SKEW_STEP / _RSTEP): every lane's lead, and whether this wave qualifies
    int lmax = 0;
    bool skew_ok = false, skew_lean = false, skewed = false, skew_delay = false;
    if (STOPS && TAB_LDS && a.fm_skew && tame_geom && a.num_frames >= 32 && (xf & (XF_FM | XF_AP)) && (xf & ~(XF_FM | XF_AP | XF_HOLDQ)) == 0) {
      const int fm_src = SK_SKEW_COL(r.fm_addr), am_src = SK_SKEW_COL(r.am_addr), pm_src = SK_SKEW_COL(r.pm_addr);
      const bool self_fm = r.fm_addr >= 0 && fm_src == lane;
      int *lead_w = reinterpret_cast<int *>(ring);          // (row 0 of the ring, as 64 words)
      SK_FAST_WAVE_SYNC()
      lead_w[lane] = 0;
      SK_FAST_WAVE_SYNC()
      for (int it = 0; it <= SK_SKEW_LMAX; ++it) {         // a source runs one block ahead of every lane that reads it
        const int mine_ = lead_w[lane];
        SK_FAST_WAVE_SYNC()
        if (fm_src != lane) atomicMax(&lead_w[fm_src], mine_ + 1);
        if (am_src != lane) atomicMax(&lead_w[am_src], mine_ + 1);
        if (pm_src != lane) atomicMax(&lead_w[pm_src], mine_ + 1);
        SK_FAST_WAVE_SYNC()
      }
      const int lead = lead_w[lane];
      const int lf_ = lead_w[fm_src], la_ = lead_w[am_src], lp_ = lead_w[pm_src];
      SK_FAST_WAVE_SYNC()
      const bool edges_ = (fm_src == lane || lf_ == lead + 1) && (am_src == lane || la_ == lead + 1) && (pm_src == lane || lp_ == lead + 1);
      // a source that is HEARD (37.sk's v4: a modulator without `m1`): one block ahead at most, its pan at rest -- its (L, R) of a
      // frame are formed a step later from the sample it left in its own ring column
      const bool heard_ = lead == 1 && !silent && r.pm_addr == -1;
      skew_ok = __all(edges_ && !self_fm && lead <= SK_SKEW_LMAX && (lead == 0 || silent || heard_));
      skew_delay = skew_ok && __any(heard_);
      lmax = __any(lead >= 3) ? 3 : __any(lead >= 2) ? 2 : __any(lead >= 1) ? 1 : 0;
      SK_SKEW_LEAD() = (unsigned char)lead;
      SK_FAST_WAVE_SYNC()
      skew_lean = skew_ok && xf == XF_FM && lmax == 1;
    }
    (void)lmax; (void)skew_ok; (void)skew_lean; (void)skewed; (void)skew_delay;

    bool moved = false;                               // (wave-uniform) some chunk of this pass had an envelope in motion
    for (int c0 = 0; c0 < a.num_frames; c0 += SK_CHUNK) {
      const int cn = min(SK_CHUNK, a.num_frames - c0);
      // a stopping voice that cannot reach its table end within this chunk (forward, unmodulated: phase + 64*inc,
      // rounding included, stays below it) needs no per-frame finish test yet
      const bool stop_near = STOPS && any_stop && __any(r.stop && !(r.phase + (float)SK_CHUNK * r.inc + 2.0f < r.hi));
      // skewed blocks in this chunk: under way already, or enough whole blocks left for every lead (a lane `lead` blocks ahead
      // renders `lead` blocks in the lead-in steps)
      const bool skew_now = skew_ok && (skewed || lmax <= ((a.num_frames - c0) >> 3));
      (void)skew_now;
      const int xf_blk = xf;                          // the feature mask the block paths test per frame (shadowed by a literal 0
      (void)xf_blk;                                   //  where a wave has nothing to test: one scheduling region per block)
      FastEnv ev_;                                    // (RAMPK) envelopes in motion on the block paths; per chunk
      ev_.clk = ev_.base = ev_.A = ev_.B = ev_.C = ev_.clk2 = ev_.base2 = ev_.A2 = ev_.B2 = ev_.C2 = 0.0f;
      ev_.den = ev_.rinv = ev_.den2 = ev_.rinv2 = 1.0f; ev_.bnd = 0.0f;
      bool steady = true, exact = true, ramp_ok = false;
      float ramp_tf = 0.0f, ramp_trf = 0.0f;
      (void)ramp_tf; (void)ramp_trf; (void)ramp_ok;
      if (ENV) {
        // Envelope clocks for this chunk from the integer timeline: frame c0+j has
        // now = count0 + c0 + j + 1 (synth.c:521).  d_* are the clocks of "frame c0 - 1".
        const uint64_t base = a.count0 + (uint64_t)c0;
        const uint64_t d_on = base - t_start;
        const uint64_t d_off = base - t_release;
        const uint64_t lim = (1ull << 24) - (uint64_t)SK_CHUNK - 2;   // x + 1.0f stays exact below 2^24
        exact = __all(dead || !r.use_env || ((d_on < lim) && (!released || d_off < lim)));
        r.tf = (float)d_on;
        r.trf = released ? (float)d_off : 0.0f;
        // sustain is absorbing within a launch: the clock only grows and note-off arrives between launches
        const float tf_first = (float)(d_on + 1);
        // ... and so is an envelope that is not running (never triggered, or its release has run out: is_active == 0, e = 0,
        // synth.c:400-401): a constant gain of amp * (0 * velocity) until a control action -- notes END all the time in a live
        // bank, and their waves must come back to the steady blocks
        const bool idle = r.use_env && !(r.rw & SKR_ENV_ACTIVE);
        if (idle && !dead) r.gain_sustain = r.amp * (0.0f * r.vel);
        // A note-on AHEAD of the clock (a host that schedules a note by writing sample_start itself): until the clock gets there
        // the reference reads the wrapped difference as a huge elapsed time, i.e. "sustain", and then the attack starts by itself
        // (synth.c:401) -- the one way a constant level ends without a control action.  Such a lane is never steady, and `exact`
        // is false for it (d_on wrapped), so its wave walks the chunk on integer clocks: the reference's own arithmetic.
        const bool ahead = r.use_env && !idle && (int64_t)(t_start - (base + 1)) > 0;
        steady = __all(dead || !r.use_env || idle || (!released && !(tf_first < r.attdec) && !ahead));
        if (RAMPK) moved = moved || !steady;
        // envelopes in motion: can the chunk's 8-frame blocks keep the straight-line form (fast_env_span2)?
        if (RAMPK && !steady && exact && tame && !stems_on && cn >= 8 && (!STOPS || (!any_fm && !stop_near))) {
          const uint64_t n8 = (uint64_t)(cn & ~7);
          bool ok = true, runs_out = false;
          fast_env_span2(r, ev_, dead, released, tf_first, (float)(d_off + 1), (float)(d_on + n8), (float)(d_off + n8), r.tf, r.trf, ok, runs_out);
          ramp_ok = __all(ok);
          if (ramp_ok && runs_out) r.rw &= ~SKR_ENV_ACTIVE;      // synth.c:429 (the frames after the blocks read the flag themselves)
          ramp_tf = (float)(d_on + n8); ramp_trf = released ? (float)(d_off + n8) : 0.0f;   // clocks of the blocks' last frame
        }
      }
      if (STOPS && (!ENV || steady) && tame && !any_fm && !stop_near && TAB_LDS && !stems_on) {
        // an extended bank, but nothing in THIS wave needs the frame loop now (e.g. only some voices filtered, or one-shots
        // still far from their end): frame pairs
        int j = 0;
        SK_FAST_PACK_IN()
        if (xf & XF_NOISE) {          // a wave with noise lanes: its own copy of the block loop (see SK_FAST_BLOCK_SAMPLE)
          if (fast_smoother_stalled<ENV>(r)) { for (; j + 8 <= cn; j += 8) { float s_[8]; SK_FAST_OSC8(s_) SK_FAST_POST8_(s_, true, false, true) SK_FAST_TILE_REDUCE(j) } }
          else { for (; j + 8 <= cn; j += 8) { float s_[8]; SK_FAST_OSC8(s_) SK_FAST_POST8_(s_, false, false, true) SK_FAST_TILE_REDUCE(j) } }
          for (; j + 1 < cn; j += 2) SK_FAST_PAIR_STEADY_(j, true, true)
        } else if (xf & (XF_HOLDQ | XF_NOSMOOTH)) {
          if (fast_smoother_stalled<ENV>(r)) SK_FAST_LDS_CHUNK(true) else SK_FAST_LDS_CHUNK(false)
          for (; j + 1 < cn; j += 2) SK_FAST_PAIR_STEADY(j, true)
        } else {                      // only per-lane filter / envelope flags, or one-shots far from their end: nothing to test per frame
          const int xf_blk = 0;
          if (fast_smoother_stalled<ENV>(r)) SK_FAST_LDS_CHUNK(true) else SK_FAST_LDS_CHUNK(false)
          for (; j + 1 < cn; j += 2) SK_FAST_PAIR_STEADY(j, true)
        }
        SK_FAST_PACK_OUT()
        if (j < cn) { SK_FAST_EVEN(j, true) SK_FAST_FIX_ODD_TAIL() }
      } else if (STOPS && (!ENV || steady)) {
        int j = 0;
        if (!TAB_LDS && tame && !any_fm && !stems_on) {   // a voice about to finish is `direct` in its window block; the block checks per frame
          SK_FAST_PACK_IN()
          if (xf & XF_NOISE) {
            if (fast_smoother_stalled<ENV>(r)) for (; j + 8 <= cn; j += 8) SK_FAST_WIN_BLOCK_(j, true, false, true)
            else for (; j + 8 <= cn; j += 8) SK_FAST_WIN_BLOCK_(j, false, false, true)
          } else {
            if (fast_smoother_stalled<ENV>(r)) for (; j + 8 <= cn; j += 8) SK_FAST_WIN_BLOCK(j, true)
            else for (; j + 8 <= cn; j += 8) SK_FAST_WIN_BLOCK(j, false)
          }
          SK_FAST_PACK_OUT()
        } else if (TAB_LDS && skew_now && !skew_lean && !stems_on) {    // skewed blocks, RICH form
          int pend_j = -1;
          SK_FAST_SKEW_CHUNK(SK_FAST_SKEW_RSTEP(j, true), , )
          SK_FAST_LDS_FLUSH()
        } else if (fm_only && !stems_on) {
          int pend_j = -1;
          if (TAB_LDS && skew_now && skew_lean) {                     // skewed blocks, LEAN form
            const bool stall_ = fast_smoother_stalled<ENV>(r);
            SK_FAST_PACK_IN()
            SK_FAST_SKEW_CHUNK(SK_FAST_SKEW_STEP(j), SK_FAST_PACK_OUT(), SK_FAST_REPACK())
            SK_FAST_LDS_FLUSH()
            SK_FAST_PACK_OUT()
          } else {
            SK_FAST_PACK_IN()
            for (; j + 8 <= cn; j += 8) SK_FAST_FM_BLOCK(j)
            SK_FAST_LDS_FLUSH()
            SK_FAST_PACK_OUT()
          }
        } else {
          int pend_j = -1;
          for (; j + 8 <= cn; j += 8) SK_FAST_X_BLOCK(j)
          SK_FAST_LDS_FLUSH()
        }
        for (; j + 1 < cn; j += 2) { SK_FAST_EVEN(j, true) SK_FAST_ODD(j + 1, true) }
        if (j < cn) { SK_FAST_EVEN(j, true) SK_FAST_FIX_ODD_TAIL() }
      } else if ((!ENV || steady) && tame && !stems_on) {
        int j = 0;
        if (!TAB_LDS) {
          SK_FAST_PACK_IN()
          if (fast_smoother_stalled<ENV>(r)) for (; j + 8 <= cn; j += 8) SK_FAST_WIN_BLOCK(j, true)
          else for (; j + 8 <= cn; j += 8) SK_FAST_WIN_BLOCK(j, false)
          for (; j + 1 < cn; j += 2) SK_FAST_PAIR_STEADY(j, true)
          SK_FAST_PACK_OUT()
        } else {
          SK_FAST_PACK_IN()
#ifndef SK_FAST_NO_LOZ
          if (loz) {                  // plain LUTs (no loop window): the two-instruction wrap
            constexpr bool SK_LOZ_ = true;
            if (fast_smoother_stalled<ENV>(r)) SK_FAST_LDS_CHUNK(true) else SK_FAST_LDS_CHUNK(false)
          } else
#endif
          { if (fast_smoother_stalled<ENV>(r)) SK_FAST_LDS_CHUNK(true) else SK_FAST_LDS_CHUNK(false) }
          for (; j + 1 < cn; j += 2) SK_FAST_PAIR_STEADY(j, true)
          SK_FAST_PACK_OUT()
        }
        if (j < cn) { SK_FAST_EVEN(j, true) SK_FAST_FIX_ODD_TAIL() }
      } else if (!ENV || steady) {
        int j = 0;
        if (stems_on) for (; j + 1 < cn; j += 2) { SK_FAST_EVEN(j, true) SK_FAST_ODD(j + 1, true) }   // frame by frame, stems written
        else { SK_FAST_PACK_IN() for (; j + 1 < cn; j += 2) SK_FAST_PAIR_STEADY(j, false) SK_FAST_PACK_OUT() }
        if (j < cn) { SK_FAST_EVEN(j, true) SK_FAST_FIX_ODD_TAIL() }
      } else if (RAMPK && ramp_ok) {
        // envelopes in motion, block paths: the stalled-smoother skip aside, the steady blocks with a gain that is evaluated
        // per frame (block by block, not the software-pipelined chunk); what is left of the chunk after its last whole block
        // takes the general frames
        int j = 0;
        SK_FAST_PACK_IN()
        if (TAB_LDS) {
          if (STOPS && (xf & XF_NOISE)) { for (; j + 8 <= cn; j += 8) { float s_[8]; SK_FAST_OSC8(s_) SK_FAST_POST8_(s_, false, true, true) SK_FAST_TILE_REDUCE(j) } }
          else { for (; j + 8 <= cn; j += 8) { float s_[8]; SK_FAST_OSC8(s_) SK_FAST_POST8_(s_, false, true, false) SK_FAST_TILE_REDUCE(j) } }
        } else {
          if (STOPS && (xf & XF_NOISE)) for (; j + 8 <= cn; j += 8) SK_FAST_WIN_BLOCK_(j, false, true, true)
          else for (; j + 8 <= cn; j += 8) SK_FAST_WIN_BLOCK_(j, false, true, false)
        }
        SK_FAST_PACK_OUT()
        r.tf = ramp_tf; r.trf = ramp_trf;
        for (; j + 1 < cn; j += 2) {
          r.tf += 1.0f; r.trf += 1.0f;
          SK_FAST_EVEN(j, false)
          r.tf += 1.0f; r.trf += 1.0f;
          SK_FAST_ODD(j + 1, false)
        }
        if (j < cn) { r.tf += 1.0f; r.trf += 1.0f; SK_FAST_EVEN(j, false) SK_FAST_FIX_ODD_TAIL() }
      } else if (exact) {
        int j = 0;
        for (; j + 1 < cn; j += 2) {                 // clocks == (float)(now - sample_start), exact below 2^24
          r.tf += 1.0f; r.trf += 1.0f;
          SK_FAST_EVEN(j, false)
          r.tf += 1.0f; r.trf += 1.0f;
          SK_FAST_ODD(j + 1, false)
        }
        if (j < cn) { r.tf += 1.0f; r.trf += 1.0f; SK_FAST_EVEN(j, false) SK_FAST_FIX_ODD_TAIL() }
      } else {
        for (int j = 0; j < cn; ++j) {               // clocks past 2^24 frames: integer path, synth.c:401,422
          const uint64_t now = a.count0 + (uint64_t)(c0 + j) + 1;
          r.tf = (float)(now - t_start); r.trf = (float)(now - t_release);
          SK_FAST_EVEN(j, false)
          SK_FAST_FIX_ODD_TAIL()
        }
      }
#ifndef SK_ABLATE_BARRIERS   /* (timing experiments only: without the barriers the flush races) */
      __syncthreads();
#endif
      if (tid < 2 * cn) {
        const float *w = reinterpret_cast<const float *>(wsum);
        float s = w[0 * 2 * SK_CHUNK + tid];
        s += w[1 * 2 * SK_CHUNK + tid];
        s += w[2 * 2 * SK_CHUNK + tid];
        s += w[3 * 2 * SK_CHUNK + tid];
        sk_row_store(a.partial + part_base + (size_t)c0 * 2 + tid, s, first_pass, publish);
      }
#ifndef SK_ABLATE_BARRIERS
      __syncthreads();
#endif
    }

    // "an envelope moved in this launch": what the host's choice between the two instantiations rests on
    // (only the RAMPK instantiation reports: while the lean one runs, envelopes can start moving through a control action
    // only, and every control action sends the host back to RAMPK by itself)
    if (RAMPK && moved && lane == 0) sk_note_moved(a, bid);
    // store the recurrences; skipped voices keep their state and get voice_sample = 0 (synth.c:532,538)
    if (absent) {
    } else if (!dead) {
      uint4 s0, s1;
      if (STOPS && !r.filt) { r.x1 = r.ox1; r.x2 = r.ox2; r.y1 = r.oy1; r.y2 = r.oy2; }
      s0.x = __float_as_uint(STOPS && r.noise ? r.ophase : r.phase); s0.y = __float_as_uint(r.sgain);
      s0.z = __float_as_uint(r.x1);    s0.w = __float_as_uint(r.x2);
      s1.x = __float_as_uint(r.y1);    s1.y = __float_as_uint(r.y2);
      s1.z = __float_as_uint(r.sample); s1.w = r.rw;
      *reinterpret_cast<uint4 *>(&a.rw[SKS_OSC][v]) = s0;
      *reinterpret_cast<uint4 *>(&a.rw[SKS_FILT][v]) = s1;
    } else if (!sample_final) {
      reinterpret_cast<uint32_t *>(&a.rw[SKS_FILT][v])[2] = 0u;
    }
    if (absent) {
    } else if (STOPS && !dead && r.hold_max)   // sample & hold state (and the pan next to it)
      *reinterpret_cast<uint4 *>(&a.rw[SKS_MISC][v]) = make_uint4(__float_as_uint(r.hold), (uint32_t)r.hold_count, __float_as_uint(r.pan_l), __float_as_uint(r.pan_r));
    else if (STOPS && r.pan_dirty)      // pan modulation rewrote voice_pan_left / _right (synth.c:600-601)
      *reinterpret_cast<uint4 *>(&a.rw[SKS_MISC][v]) = make_uint4(misc_xy.x, misc_xy.y, __float_as_uint(r.pan_l), __float_as_uint(r.pan_r));
    first_pass = false;
  }
  if (a.finish) sk_finish_block(a, bid, tid, SK_GROUP, reinterpret_cast<int *>(lds), true);
}

// ---------------------------------------------------------------- launcher (C linkage)

// specialisation key: table residency x filter x envelope x interpolation
//
// The template matrix is compiled as FOUR translation units from this one source (-DSK_FAST_PART=n; the extended instantiations
// with the skewed blocks take minutes each): part 0 the clean instantiations and the launcher, part 1 / 2 the extended
// instantiations of LDS-table banks without / with the biquad, part 3 those of global-table banks.  (-DSK_FAST_ONE_CASE=...:
// one instantiation pair in one unit, for compile-time experiments.)
#ifndef SK_FAST_PART
#define SK_FAST_PART 0
#endif
#ifdef SK_PROBE_TU
#define SK_FAST_LAUNCHER sk_launch_render_fastp
#define SK_FAST_CASES(N) sk_fast_cases##N##p
#define SK_PROBE_FLAG true
#else
#define SK_FAST_LAUNCHER sk_launch_render_fast
#define SK_FAST_CASES(N) sk_fast_cases##N
#define SK_PROBE_FLAG false
extern "C" int sk_launch_render_fastp(const sk_render_args_t *args, int n_workgroups, size_t lds_bytes, hipStream_t stream);
#endif
extern "C" int SK_FAST_CASES(1)(const sk_render_args_t *args, unsigned grid_x, size_t lds_bytes, hipStream_t stream, int key, int rampk, int guard);
extern "C" int SK_FAST_CASES(2)(const sk_render_args_t *args, unsigned grid_x, size_t lds_bytes, hipStream_t stream, int key, int rampk, int guard);
extern "C" int SK_FAST_CASES(3)(const sk_render_args_t *args, unsigned grid_x, size_t lds_bytes, hipStream_t stream, int key, int rampk, int guard);

#define SK_FAST_BIG_LDS_(K)                                                                                                 \
  if (lds_bytes > 65536) {   /* (per launch: the attribute belongs to the device the calling thread is on) */              \
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(&K), hipFuncAttributeMaxDynamicSharedMemorySize, 81920) != hipSuccess) return (int)hipGetLastError(); }
#define SK_FAST_LAUNCH_(T, F, E, I, X)                                                                                      \
  { if (E && rampk) { SK_FAST_BIG_LDS_((sk_render_fast_kernel<T, F, E, I, X, E, SK_PROBE_FLAG>)) hipLaunchKernelGGL((sk_render_fast_kernel<T, F, E, I, X, E, SK_PROBE_FLAG>), grid, block, lds_bytes, stream, *args); }   \
    else { SK_FAST_BIG_LDS_((sk_render_fast_kernel<T, F, E, I, X, false, SK_PROBE_FLAG>)) hipLaunchKernelGGL((sk_render_fast_kernel<T, F, E, I, X, false, SK_PROBE_FLAG>), grid, block, lds_bytes, stream, *args); } }
#define SK_FAST_CASE_C(K, T, F, E, I) case K: if (I && guard) SK_FAST_LAUNCH_(T, F, E, (I ? 2 : 0), false) else SK_FAST_LAUNCH_(T, F, E, I, false) break;
#define SK_FAST_CASE_X(K, T, F, E, I) case 16 + K: if (I && guard) SK_FAST_LAUNCH_(T, F, E, (I ? 2 : 0), true) else SK_FAST_LAUNCH_(T, F, E, I, true) break;

#if SK_FAST_PART != 0 && !defined(SK_FAST_ONE_CASE)
// the extended instantiations of this part
#if SK_FAST_PART == 1
extern "C" int SK_FAST_CASES(1)
#elif SK_FAST_PART == 2
extern "C" int SK_FAST_CASES(2)
#else
extern "C" int SK_FAST_CASES(3)
#endif
    (const sk_render_args_t *args, unsigned grid_x, size_t lds_bytes, hipStream_t stream, int key, int rampk, int guard) {
  dim3 grid(grid_x), block(SK_GROUP);
  switch (key) {
#if SK_FAST_PART == 1
    SK_FAST_CASE_X(8, true, false, false, 0)  SK_FAST_CASE_X(9, true, false, false, 1)
    SK_FAST_CASE_X(10, true, false, true, 0)  SK_FAST_CASE_X(11, true, false, true, 1)
#elif SK_FAST_PART == 2
    SK_FAST_CASE_X(12, true, true, false, 0)  SK_FAST_CASE_X(13, true, true, false, 1)
    SK_FAST_CASE_X(14, true, true, true, 0)   SK_FAST_CASE_X(15, true, true, true, 1)
#else
    SK_FAST_CASE_X(0, false, false, false, 0) SK_FAST_CASE_X(1, false, false, false, 1)
    SK_FAST_CASE_X(2, false, false, true, 0)  SK_FAST_CASE_X(3, false, false, true, 1)
    SK_FAST_CASE_X(4, false, true, false, 0)  SK_FAST_CASE_X(5, false, true, false, 1)
    SK_FAST_CASE_X(6, false, true, true, 0)   SK_FAST_CASE_X(7, false, true, true, 1)
#endif
  }
  return (int)hipGetLastError();
}
#else
extern "C" int SK_FAST_LAUNCHER(const sk_render_args_t *args, int n_workgroups, size_t lds_bytes,
                                hipStream_t stream) {
#ifndef SK_PROBE_TU
  if (args->probe_out) return sk_launch_render_fastp(args, n_workgroups, lds_bytes, stream);
#endif
  const bool tab_lds = args->lds_table_floats > 0;
  // per wave: the reduction tile of the block paths; global-table banks: a table window too (LDS-table banks used to get the
  // window's 5 KB per wave as well: 12 KB per workgroup that cost banks with 32 KB of tables their third workgroup per CU)
  lds_bytes += (size_t)4 * (8 * SK_XT) * sizeof(float);
  if (args->lds_table_floats == 0) lds_bytes += (size_t)4 * (SK_WIN * 64) * sizeof(float);
  sk_render_args_t skew_args;    /* the skewed blocks' sample rings */
  if (args->fm_skew) {
    skew_args = *args;
    /* (beyond the 64 KB a launch gets by default the kernel's limit is raised; up to half a CU's LDS, so that a bank with a
       48 KB pool keeps its two workgroups per CU) */
    if (tab_lds && lds_bytes + (size_t)4 * SK_SKEW_RING * sizeof(float) <= 81920) lds_bytes += (size_t)4 * SK_SKEW_RING * sizeof(float);
    else skew_args.fm_skew = 0;
    args = &skew_args;
  }
  dim3 grid((unsigned)(n_workgroups + args->wg_shift)), block(SK_GROUP);
  const int key = (((args->fast_mode & (SKM_STOPS | SKM_FM | SKM_MIXED)) || args->pack_shift < 6) ? 16 : 0) |   /* the extended instantiation */ (tab_lds ? 8 : 0) | ((args->fast_mode & SKM_FILTER_ALL) ? 4 : 0) |
                  ((args->fast_mode & SKM_ENV_ALL) ? 2 : 0) | (args->interp != 0 ? 1 : 0);
  const bool rampk = !args->skip_env2;   /* envelopes may be moving (skip_env2: a launch has reported that none did) */
  const bool guard = args->interp == 2;  /* linear lookup, every live voice on a guarded whole-table loop (SKF_GUARD; the host counts) */
#ifdef SK_FAST_ONE_CASE   /* (compile-time experiments: one instantiation pair) */
#define SK_FAST_CASE_BOTH(...) SK_FAST_CASE_C(__VA_ARGS__) SK_FAST_CASE_X(__VA_ARGS__)
  switch (key) { SK_FAST_CASE_BOTH(SK_FAST_ONE_CASE) }
#else
  if (key >= 24) return (key & 4) ? SK_FAST_CASES(2)(args, grid.x, lds_bytes, stream, key, rampk, guard) : SK_FAST_CASES(1)(args, grid.x, lds_bytes, stream, key, rampk, guard);
  if (key >= 16) return SK_FAST_CASES(3)(args, grid.x, lds_bytes, stream, key, rampk, guard);
  switch (key) {
    SK_FAST_CASE_C(0, false, false, false, 0) SK_FAST_CASE_C(1, false, false, false, 1)
    SK_FAST_CASE_C(2, false, false, true, 0)  SK_FAST_CASE_C(3, false, false, true, 1)
    SK_FAST_CASE_C(4, false, true, false, 0)  SK_FAST_CASE_C(5, false, true, false, 1)
    SK_FAST_CASE_C(6, false, true, true, 0)   SK_FAST_CASE_C(7, false, true, true, 1)
    SK_FAST_CASE_C(8, true, false, false, 0)  SK_FAST_CASE_C(9, true, false, false, 1)
    SK_FAST_CASE_C(10, true, false, true, 0)  SK_FAST_CASE_C(11, true, false, true, 1)
    SK_FAST_CASE_C(12, true, true, false, 0)  SK_FAST_CASE_C(13, true, true, false, 1)
    SK_FAST_CASE_C(14, true, true, true, 0)   SK_FAST_CASE_C(15, true, true, true, 1)
  }
#endif
  return (int)hipGetLastError();
}
#endif
