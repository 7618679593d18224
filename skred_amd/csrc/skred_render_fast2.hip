// skred_render_fast2.hip -- sk_render_fast2_kernel / sk_render_env2_kernel: two voices per lane, large clean banks.
#define SK_TWO_PER_LANE_TU 1   /* skred_kernel_common.hpp: the final arriver also reports and re-arms sk_gain_kernel's row counter */
#include "skred_kernel_common.hpp"
#include "skred_launch.h"

// ---------------------------------------------------------------- fast kernel, two voices per lane
//
// Same per-voice arithmetic as sk_render_fast_kernel, but every lane carries TWO voices (v and v+64
// of a 128-voice wave slice) as 2-vectors, so that
//   * the mul/add backbone (phase add, wrap, biquad, smoother, gain, pan) issues as packed fp32
//     (v_pk_add_f32 / v_pk_mul_f32: two voices per instruction, each lane-op still IEEE fp32,
//     unfused, hence bit-identical), and
//   * the 12-instruction cross-lane DPP reduction is paid once per 128 voices instead of per 64.
// Compares, selects, float->int conversion and the LDS gathers stay per voice.  Used for large
// clean banks (the host decides, skred_bank.c); per-voice results equal the other two kernels'.

typedef float v2f __attribute__((ext_vector_type(2)));

#ifndef SK_LDS_REDUCE
#define SK_LDS_REDUCE 1     /* 1: tame chunks of LDS-table banks sum across lanes through an LDS transposition tile
                               (+11 % measured at 2^20 voices; global-gather banks keep the DPP butterfly: the
                               8-frame block costs them 30 %); 0: always the DPP butterfly */
#endif

// Two kernels share this machinery.  sk_render_fast2_kernel renders the voices that hold a CONSTANT envelope level
// for the whole launch (sustain, inactive, finished release, or no envelope at all): its loop has no envelope code and
// a small register footprint.  The voices that may be in motion -- the MOTION LIST, a bit per voice carried on the
// device from block to block (skred_device_layout.h: mask_cur) -- sit that kernel out and are rendered by
// sk_render_env2_kernel, which carries the envelope machinery (and its registers) alone and runs BESIDE it on a second
// stream: its own rows, its own ticket, and the block's final ticket shared (skred_kernel_common.hpp: sk_finish_env).
// No voice is ever sat out without that kernel being launched in the same block: both read the same, read-only list.
// This source is compiled twice (Makefile): as skred_render_fast2.o with the plain instantiations and the launcher
// sk_launch_render_fast2, and -- with -DSK_FAST2_FMP_TU=1 / =2 -- as skred_render_fm2.o / skred_render_fm2ap.o with the
// FMP (two-operator FM; ... with amplitude / pan modulation) instantiations and sk_launch_render_fm2 / _fm2ap, so that the two halves of the template matrix compile side by side.
#define SK_FAST2_TU_LOCAL static

struct Fast2Regs {
  v2f inc, lo, hi, span, span2;
  int toff4[2], tsize_m1[2];
  v2f k, b0, b1, b2, a1, a2, gain_const;   // gain_const: gain of a constant-level lane
  v2f pan_lr[2];                           // (pan_left, pan_right) of voice 0 / voice 1: packed by channel, not by voice
  // MIXED instantiations: banks in which only some voices run the biquad / use the envelope
  bool filt[2];                            // this voice runs the biquad
  bool fake_active[2];                     // an un-enveloped voice is rendered as a held note at level 1 (amp * (1*1) == amp):
                                           // its ENV_ACTIVE bit was forced on for the launch and is taken back at the store
  v2f ox1, ox2, oy1, oy2;                  // delay lines as loaded: an unfiltered voice gets its own back untouched
  v2f phase, sgain, x1, x2, y1, y2, sample;
  uint32_t rw[2];
  // FMP kernels (two-operator FM banks: the lane's voice 0 is an even voice, voice 1 the voice after it): voice 0 is
  // frequency-modulated by voice 1's sample of the previous frame (synth.c:548-555: the carrier comes first in index order)
  bool fm_on;
  float fm_k, fm_depth;                    // voice_phase_inc[m] * voice_freq_scale[n];  voice_freq_mod_depth[n]
  // ... and its amplitude / pan may be modulated by voice 1's previous sample or by its own sample of this frame
  // (synth.c:583-588 post-filter, synth.c:597-602 post-gain; see fast_post in skred_render_fast.hip)
  bool am_on, am_self, pm_on, pm_self, pan_dirty;
  float am_depth, pm_depth, mprev;         // mprev: voice 1's sample of the previous frame, read before this frame changes it
  uint32_t misc_x, misc_y;                 // voice 0's sample & hold words: they travel back with a modulated pan (one 16-byte store)
  // in-place instantiation (GT): a voice on the motion list stays in its lane and takes the gain of every frame from the row
  // sk_gain_kernel left for it (skred_device_layout.h: env_gain)
  bool lst[2];                             // this voice is listed
  v2f gt;                                  // the current frame's gains of the lane's listed voices (set by the caller)
#ifdef SK_PROBE_TU
  float2 *probe[2];                        // this frame's probe row of the lane's voices (nullptr: not probed / silent / rendered elsewhere)
  int probe_stride;                        // float2 per frame
  bool probe_any;                          // (wave-uniform) some lane of the wave writes probes
#endif
};

struct Env2Regs {                 // sk_render_env2_kernel only
  float att[2], attdec[2], dec[2], rel[2];
  v2f ampv, velv, susv, omsv;     // amp, velocity, sustain_level, 1 - sustain_level
  v2f clk, ebase, eden, erinv, eA, eB, eC;   // "ramp" spans: see fast2_env_span
  v2f clk2, ebase2, eden2, erinv2, eA2, eB2, eC2, bnd;   // blocks with ONE stage change per lane: the stage after it, and
                                                         // the clock value (of `clk`) from which it applies (fast2_env_span2)
  float tf[2], trf[2];            // general frames: this frame's clocks
};

template <bool TAB_LDS, int INTERP, bool NOCLAMP>
__device__ __forceinline__ float fast2_fetch(const char *lds_tab, const char *__restrict__ glb_tab,
                                             int toff4, int tsize_m1, float lo, float hi, float pos) {
  int idx = (int)pos;
  if (!NOCLAMP) idx = max(min(idx, tsize_m1), 0);
  const char *tab = TAB_LDS ? lds_tab : glb_tab;
  if (INTERP == 0) return *reinterpret_cast<const float *>(tab + (toff4 + (idx << 2)));
  if (INTERP == 2 && NOCLAMP) {                                   // linear over guarded whole-table loops: see fast_fetch
    const tap_pair_t pg = TAB_LDS ? *reinterpret_cast<const tap_pair_t *>(tab + (toff4 + (idx << 2)))
                                  : load_tap_pair_global(tab + (toff4 + (idx << 2)));
    return pg.a + __builtin_amdgcn_fractf(pos) * (pg.b - pg.a);
  }
  const tap_pair_t pr = TAB_LDS ? *reinterpret_cast<const tap_pair_t *>(tab + (toff4 + (idx << 2)))   // see fast_fetch
                                : load_tap_pair_global(tab + (toff4 + (idx << 2)));
  const float a = pr.a;
  float b = pr.b;
  int nxt = idx + 1;
  bool special = (float)nxt >= hi;
  if (special) nxt = (int)lo;
  if (!NOCLAMP) { const int c = max(min(nxt, tsize_m1), 0); special = special || (c != nxt); nxt = c; }
  if (special) b = *reinterpret_cast<const float *>(tab + (toff4 + (nxt << 2)));
  const float frac = pos - (float)idx;
  return a + frac * (b - a);
}

// (env_stage_code: skred_kernel_common.hpp: sk_env_stage_code)
#define env_stage_code sk_env_stage_code

// Per-lane envelope constants for a span of frames whose first / last frame have clocks (t1,tr1) / (tN,trN)
// and whose preceding frame has (t0,tr0).  Constant lanes get gain_const; moving lanes get the "ramp" form
//     q = (clk - ebase) / eden,  e = eC * (eA + eB * q)
// which is bit-identical to the reference's stage expressions:
//     attack  q               = 1*(0 + 1*q)            (synth.c:405)
//     decay   1 - q*(1-sus)   = 1*(1 + (-(1-sus))*q)   (synth.c:413)
//     release sus*(1 - q)     = sus*(1 + (-1)*q)       (synth.c:425)
// `st` stays true while every lane sits on a constant level; `same` while no lane changes stage in the span
// (and the denominators are in the range where the short division below equals the full IEEE expansion).
__device__ __forceinline__ void fast2_env_span(Fast2Regs &r, Env2Regs &e, int c, bool dead, bool released,
                                               float t1, float tr1, float tN, float trN, float t0, float tr0,
                                               bool &st, bool &same) {
  const bool act = (r.rw[c] & SKR_ENV_ACTIVE) != 0;
  const int code0 = env_stage_code(act, released, t1, tr1, e.att[c], e.attdec[c], e.rel[c]);
  const int code1 = env_stage_code(act, released, tN, trN, e.att[c], e.attdec[c], e.rel[c]);
  st = st && (dead || code0 == 0 || code0 == 3 || code0 == 5);
  const float level = code0 == 3 ? e.susv[c] : 0.0f;
  e.eA[c] = code0 == 1 ? 0.0f : 1.0f;
  e.eB[c] = code0 == 1 ? 1.0f : (code0 == 2 ? -e.omsv[c] : (code0 == 4 ? -1.0f : 0.0f));
  e.eC[c] = code0 == 4 ? e.susv[c] : ((code0 == 1 || code0 == 2) ? 1.0f : level);
  e.clk[c] = code0 == 4 ? tr0 : t0;
  e.ebase[c] = code0 == 2 ? e.att[c] : 0.0f;
  const float d = code0 == 1 ? e.att[c] : (code0 == 2 ? e.dec[c] : (code0 == 4 ? e.rel[c] : 1.0f));
  e.eden[c] = d;
  const float r0 = __builtin_amdgcn_rcpf(d);
  e.erinv[c] = __builtin_fmaf(__builtin_fmaf(-d, r0, 1.0f), r0, r0);   // refined reciprocal, as v_rcp + one fma step
  r.gain_const[c] = e.ampv[c] * (level * e.velv[c]);                   // amp * (e * velocity), synth.c:582,588
  same = same && (dead || (code0 == code1 && d >= 0x1p-40f && d <= 0x1p40f));
  // a release that has run out: the reference clears is_active on the first frame it notices (synth.c:429)
  if (!dead && code0 == 5) r.rw[c] &= ~SKR_ENV_ACTIVE;
}

// The same for a span in which a lane may change stage ONCE, to the stage that follows its own (attack -> decay, decay ->
// sustain or release, release -> finished): the ramp constants of the stage after the change go to the second set and
// `bnd` is the value of the first set's clock from which they apply -- the comparison the reference makes on that frame
// (synth.c:403,408,420: `t < attack_time`, `t < decay_start + decay_time`, `t_release < release_time`).  A lane that keeps
// its stage gets bnd = +inf.  r.rw is not touched (see `runs_out`).  `ok` stays true while no lane does anything else (two changes inside the span: attack or
// decay shorter than the span, or a release that has already run out when the decay ends).
__device__ __forceinline__ void fast2_env_span2(Fast2Regs &r, Env2Regs &e, int c, bool dead, bool released,
                                                float t1, float tr1, float tN, float trN, float t0, float tr0, bool &ok,
                                                bool &runs_out) {
  const bool act = (r.rw[c] & SKR_ENV_ACTIVE) != 0;
  const int code0 = env_stage_code(act, released, t1, tr1, e.att[c], e.attdec[c], e.rel[c]);
  const int code1 = env_stage_code(act, released, tN, trN, e.att[c], e.attdec[c], e.rel[c]);
  const bool step = code0 != code1;
  const bool next_stage = (code0 == 1 && code1 == 2) || (code0 == 2 && (code1 == 3 || code1 == 4)) || (code0 == 4 && code1 == 5);
  float den[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int code = h ? code1 : code0;
    const float level = code == 3 ? e.susv[c] : 0.0f;
    const float A = code == 1 ? 0.0f : 1.0f;
    const float B = code == 1 ? 1.0f : (code == 2 ? -e.omsv[c] : (code == 4 ? -1.0f : 0.0f));
    const float C = code == 4 ? e.susv[c] : ((code == 1 || code == 2) ? 1.0f : level);
    const float clk = code == 4 ? tr0 : t0;
    const float base = code == 2 ? e.att[c] : 0.0f;
    const float d = code == 1 ? e.att[c] : (code == 2 ? e.dec[c] : (code == 4 ? e.rel[c] : 1.0f));
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float ri = __builtin_fmaf(__builtin_fmaf(-d, r0, 1.0f), r0, r0);
    den[h] = d;
    if (h == 0) { e.eA[c] = A; e.eB[c] = B; e.eC[c] = C; e.clk[c] = clk; e.ebase[c] = base; e.eden[c] = d; e.erinv[c] = ri; }
    else        { e.eA2[c] = A; e.eB2[c] = B; e.eC2[c] = C; e.clk2[c] = clk; e.ebase2[c] = base; e.eden2[c] = d; e.erinv2[c] = ri; }
  }
  e.bnd[c] = !step ? __builtin_huge_valf() : (code0 == 1 ? e.att[c] : (code0 == 2 ? e.attdec[c] : e.rel[c]));
  ok = ok && (dead || ((!step || next_stage) && den[0] >= 0x1p-40f && den[0] <= 0x1p40f && den[1] >= 0x1p-40f && den[1] <= 0x1p40f));
  // a release that runs out in this span: the reference clears is_active on the first frame it notices (synth.c:429).  The
  // caller clears it once it has decided to render the span in this form (the general frames read the flag themselves).
  runs_out = !dead && (code0 == 5 || code1 == 5);
}

// General frames (a lane changes stage inside the block): amp_envelope_step as the reference writes it.
__device__ __forceinline__ float fast2_env_general(Fast2Regs &r, Env2Regs &e, int c, bool released) {
  float lvl = 0.0f;
  if (r.rw[c] & SKR_ENV_ACTIVE) {
    if (e.tf[c] < e.att[c]) {
      lvl = e.tf[c] / e.att[c];
    } else if (e.tf[c] < e.attdec[c]) {
      const float prog = (e.tf[c] - e.att[c]) / e.dec[c];
      lvl = 1.0f - prog * e.omsv[c];
    } else if (!released) {
      lvl = e.susv[c];
    } else if (e.trf[c] < e.rel[c]) {
      const float prog = e.trf[c] / e.rel[c];
      lvl = e.susv[c] * (1.0f - prog);
    } else {
      r.rw[c] &= ~SKR_ENV_ACTIVE;
    }
  }
  return e.ampv[c] * (lvl * e.velv[c]);
}

// EM (envelope mode): 0 every lane has a constant gain; 1 "ramp": every lane keeps one stage, straight-line
// with the short exact division; 2 general.  TAME: see fast_frame.
// Oscillator half of a frame: advance both phases, wrap, fetch the two table samples.
// LOZ: every lane of the wave has lo == 0 (no loop window: the plain LUT case).  Then ph0 - lo == ph0 and
// lo + y == y exactly, so the wrapped phase lo + ((ph0 - lo) - span) is ph0 - span: one packed add instead of three.
template <bool TAB_LDS, bool TAME, int INTERP, bool LOZ = false, int FMP = 0>
__device__ __forceinline__ v2f fast2_osc(Fast2Regs &r, const char *lds_tab, const char *__restrict__ glb_tab) {
  v2f inc = r.inc;
  if (FMP && !TAME) {
    const float ms = r.sample.y;                       // voice_sample[m] as the previous frame left it
    if (FMP == 2) r.mprev = ms;
    inc.x = r.fm_on ? r.inc.x + r.fm_k * (ms * r.fm_depth) : r.inc.x;        // synth.c:551-554
  }
  const v2f ph0 = r.phase + inc;
  const v2f x = LOZ ? ph0 : ph0 - r.lo;
  const v2f phw = LOZ ? x - r.span : r.lo + (x - r.span);
  v2f ph;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const bool over = ph0[c] >= r.hi[c];
    float p;
    if (TAME && LOZ) {
      // lo == 0: hi == span, so `over` is the sign of phw = ph0 - span (exact for span <= ph0 < 2 span, strictly negative below
      // span), and ph0 >= 0.  As unsigned integers a negative float is larger than every non-negative one and non-negative floats
      // keep their order, so min_u32 picks phw when it is >= 0 (then phw < ph0) and ph0 otherwise: the reference's select
      // (synth.c:241-247) in one plain instruction instead of a compare and a v_cndmask (tools/issue_mix.hip: 2.6 against
      // 2.8 + 2.7 SIMD cycles per voice and frame)
      p = __uint_as_float(min(__float_as_uint(ph0[c]), __float_as_uint(phw[c])));
    } else if (TAME) {
      p = over ? phw[c] : ph0[c];
    } else {
      const bool simple = over && (x[c] < r.span2[c]);
      const bool in_range = (ph0[c] >= r.lo[c]) && !over;
      p = simple ? phw[c] : ph0[c];
      if (!(in_range || simple)) p = slow_wrap(ph0[c], r.lo[c], r.hi[c], r.span[c]);
    }
    ph[c] = p;
  }
  r.phase = ph;
  v2f s;
#ifndef SK_FAST2_NO_PAIRED_TAPS
  if (INTERP == 2 && TAME && TAB_LDS) {
    // Linear lookup of both voices with the taps landing where the packed arithmetic wants them: (a0, a1) and (b0, b1) are
    // FOUR 4-byte gathers into two register pairs, and a + fract(pos) * (b - a) is three packed instructions for the lane's
    // two voices (the same products and sums per voice as fast2_fetch).  Fetched as one (a, b) pair per voice -- a
    // ds_read2_b32, which is also what hipcc merges two adjacent 4-byte loads into, hence `volatile` -- the packed operands
    // cost three v_mov per frame to line up: the VALUs are what binds this kernel, the LDS pipe has room.
    typedef __attribute__((address_space(3))) const volatile float lds_vf;
    lds_vf *t0 = (lds_vf *)(lds_tab + (r.toff4[0] + ((int)ph.x << 2)));
    lds_vf *t1 = (lds_vf *)(lds_tab + (r.toff4[1] + ((int)ph.y << 2)));
    v2f ta, tb;
    ta.x = t0[0]; ta.y = t1[0];
    tb.x = t0[1]; tb.y = t1[1];
    const v2f fr = {__builtin_amdgcn_fractf(ph.x), __builtin_amdgcn_fractf(ph.y)};
    return ta + fr * (tb - ta);
  }
#endif
  s.x = fast2_fetch<TAB_LDS, INTERP, TAME>(lds_tab, glb_tab, r.toff4[0], r.tsize_m1[0], r.lo.x, r.hi.x, ph.x);
  s.y = fast2_fetch<TAB_LDS, INTERP, TAME>(lds_tab, glb_tab, r.toff4[1], r.tsize_m1[1], r.lo.y, r.hi.y, ph.y);
  return s;
}

// The rest of the frame: biquad, envelope/gain, smoother, pan, lane-local sum of the two voices.
// EM (envelope mode): 0 every lane has a constant gain; 1 "ramp": every lane keeps one stage, straight-line
// with the short exact division; 2 general; 3 constant gain AND the amp smoother has stalled in every lane
// (fast2_smoother_stalled: its update no longer changes it, so it is skipped); 6: as 0 with the lane's listed voices taking
// this frame's gain from r.gt (in-place instantiation).  TAME: see fast_frame.
template <bool FILTER, int EM, bool TAME, bool MIXED = false, bool MUTESEL = false, int FMP = 0>
__device__ __forceinline__ void fast2_post(Fast2Regs &r, Env2Regs &e, v2f s, v2f &xn, v2f &xo, v2f &yn, v2f &yo,
                                           const bool rel0, const bool rel1, const bool silent0,
                                           const bool silent1, float &out_l, float &out_r) {
  // ---- biquad, packed ----
  if (FILTER) {
    v2f y = r.b0 * s;
    y = y + r.b1 * xn;
    y = y + r.b2 * xo;
    y = y - r.a1 * yn;
    y = y - r.a2 * yo;
    if (MIXED) {                                       // filter_mode 0: the sample passes, the delay line rests (synth.c:577)
      xo.x = r.filt[0] ? s.x : xo.x; xo.y = r.filt[1] ? s.y : xo.y;
      yo.x = r.filt[0] ? y.x : yo.x; yo.y = r.filt[1] ? y.y : yo.y;
      s.x = r.filt[0] ? y.x : s.x;   s.y = r.filt[1] ? y.y : s.y;
    } else {
      xo = s;
      yo = y;
      s = y;
    }
  }
  // ---- gain ----
  v2f gain;
  if (EM == 0 || EM == 3 || EM == 6) {
    gain = r.gain_const;
    if (EM == 6) {                                     // a listed voice: amp * (level * velocity) of this frame from its gain row
      gain.x = r.lst[0] ? r.gt.x : gain.x;
      gain.y = r.lst[1] ? r.gt.y : gain.y;
    }
  } else if (EM == 1) {
    e.clk = e.clk + 1.0f;                              // exact: clocks stay below 2^24 in this mode
    const v2f num = e.clk - e.ebase;                   // t | t - attack_time | t_release
    // q = num / den correctly rounded: the FMA tail of the IEEE fp32 division expansion hipcc itself emits
    // (v_div_scale / v_rcp / fma.. / v_div_fmas / v_div_fixup) with the refined reciprocal prepared once per
    // span; fast2_env_span admits only denominators for which div_scale / div_fixup would not intervene.
    v2f q = num * e.erinv;
    v2f rem = __builtin_elementwise_fma(-e.eden, q, num);
    q = __builtin_elementwise_fma(rem, e.erinv, q);
    rem = __builtin_elementwise_fma(-e.eden, q, num);
    q = __builtin_elementwise_fma(rem, e.erinv, q);
    const v2f lvl = e.eC * (e.eA + e.eB * q);
    gain = e.ampv * (lvl * e.velv);
  } else if (EM == 4) {                                // one stage change per lane at most: fast2_env_span2
    e.clk = e.clk + 1.0f;
    e.clk2 = e.clk2 + 1.0f;
    v2f num, den, rinv, A, B, C;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const bool after = e.clk[c] >= e.bnd[c];           // the reference's `t < limit` failed on this frame
      num[c] = after ? e.clk2[c] - e.ebase2[c] : e.clk[c] - e.ebase[c];
      den[c] = after ? e.eden2[c] : e.eden[c];
      rinv[c] = after ? e.erinv2[c] : e.erinv[c];
      A[c] = after ? e.eA2[c] : e.eA[c];
      B[c] = after ? e.eB2[c] : e.eB[c];
      C[c] = after ? e.eC2[c] : e.eC[c];
    }
    v2f q = num * rinv;
    v2f rem = __builtin_elementwise_fma(-den, q, num);
    q = __builtin_elementwise_fma(rem, rinv, q);
    rem = __builtin_elementwise_fma(-den, q, num);
    q = __builtin_elementwise_fma(rem, rinv, q);
    const v2f lvl = C * (A + B * q);
    gain = e.ampv * (lvl * e.velv);
  } else {
    gain.x = fast2_env_general(r, e, 0, rel0);
    gain.y = fast2_env_general(r, e, 1, rel1);
  }
  if (FMP == 2 && !TAME && r.am_on)                    // final = amp * env * mod (synth.c:583-588): voice 0 only
    gain.x = gain.x * ((r.am_self ? s.x : r.mprev) * r.am_depth);
  if (EM != 3) r.sgain = r.sgain + r.k * (gain - r.sgain);
  s = s * r.sgain;
  r.sample = s;
  if (FMP == 2 && !TAME && r.pm_on && !silent0) {      // synth.c:597-602 (inside the `not disconnected` branch)
    const float q = (r.pm_self ? s.x : r.mprev) * r.pm_depth;
    r.pan_lr[0] = (v2f){(1.0f - q) / 2.0f, (1.0f + q) / 2.0f};
    r.pan_dirty = true;
  }
  // ---- pan, lane-local sum of the two voices ----
  v2f so = s;
  if (!TAME || MUTESEL) {   // plain TAME loops run only when no live lane is muted (dead lanes already yield exact zeros);
                            // MUTESEL: a tame wave with muted live lanes -- their samples are replaced by zero here
    so.x = silent0 ? 0.0f : s.x;
    so.y = silent1 ? 0.0f : s.y;
  }
  // (L, R) of the lane's two voices with the pan gains packed by channel: two packed multiplies by a broadcast
  // sample and one packed add give (out_l, out_r) ready for the tile store -- the same products and the same
  // single add per channel as l = s0*pl0 + s1*pl1, r = s0*pr0 + s1*pr1
#ifdef SK_PROBE_TU
  if (r.probe_any) {                                   // what the reference stores into its stem buffer: sample x pan (synth.c:603-608)
    if (r.probe[0]) { *r.probe[0] = make_float2(s.x * r.pan_lr[0].x, s.x * r.pan_lr[0].y); r.probe[0] += r.probe_stride; }
    if (r.probe[1]) { *r.probe[1] = make_float2(s.y * r.pan_lr[1].x, s.y * r.pan_lr[1].y); r.probe[1] += r.probe_stride; }
  }
#endif
  const v2f lr = r.pan_lr[0] * (v2f){so.x, so.x} + r.pan_lr[1] * (v2f){so.y, so.y};
  out_l = lr.x;
  out_r = lr.y;
}

// The one-pole amp smoother g += k*(gain - g) (synth.c:588-593) towards a CONSTANT gain stops moving after a
// few hundred frames: once k*(gain - g) is below half an ulp of g the sum rounds back to g, and with the same
// inputs it does so on every later frame.  Wave-uniform test of exactly that (the very expression fast2_post
// evaluates, compared bitwise), made once per 64-frame chunk of a constant-gain wave.
__device__ __forceinline__ bool fast2_smoother_stalled(const Fast2Regs &r) {
  const v2f nxt = r.sgain + r.k * (r.gain_const - r.sgain);
  return __all(!r.am_on && __float_as_uint(nxt.x) == __float_as_uint(r.sgain.x) && __float_as_uint(nxt.y) == __float_as_uint(r.sgain.y));
}

template <bool TAB_LDS, bool FILTER, int EM, bool TAME, int INTERP, bool MIXED = false, int FMP = 0>
__device__ __forceinline__ void fast2_frame(Fast2Regs &r, Env2Regs &e, v2f &xn, v2f &xo, v2f &yn, v2f &yo,
                                            const bool rel0, const bool rel1, const bool silent0,
                                            const bool silent1, const char *lds_tab,
                                            const char *__restrict__ glb_tab, float &out_l, float &out_r) {
  const v2f s = fast2_osc<TAB_LDS, TAME, INTERP, false, FMP>(r, lds_tab, glb_tab);
  fast2_post<FILTER, EM, TAME, MIXED, false, FMP>(r, e, s, xn, xo, yn, yo, rel0, rel1, silent0, silent1, out_l, out_r);
}

// ---- table windows for pools that do not fit in LDS (PCM banks) ----
//
// A scattered wave gather from L2 costs one cache-line request per lane (~2.5 cycles per lane and CU measured,
// tools/ta_rate.hip) whatever its width, and a voice that advances ~1 sample per frame asks for the same line
// many frames in a row -- but 16 waves per CU evict it from the 32 KB L1 in between.  So every 8 frames each lane
// copies the SK_WIN table samples its two voices are about to cross (5 dword-aligned global_load_dwordx4 each,
// 1-2 lines) into a wave-private LDS window win[voice][j][lane] (lane-minor: conflict-free whatever j the lanes
// pick) and the 8 frames read their taps with one ds_read2st64_b32.  Line requests drop ~5x.
// A lane whose voice could reach its loop end inside the block (wrap, or the second tap folding back to the loop
// start) or advances more than 2.1875 samples per frame is `direct` for that block: it takes the ordinary
// gather, in a branch the wave only enters when some lane needs it.  Same arithmetic either way.
typedef float win4_t __attribute__((ext_vector_type(4), aligned(4)));

struct WinRegs {
  int base[2];       // table index of win[c][0]
  bool direct[2];    // this lane's voice c bypasses the window in the current block
  bool any_direct;   // some lane of the wave does
};

__device__ __forceinline__ void fast2_win_fill(const Fast2Regs &r, const bool dead[2], WinRegs &w, float *win,
                                               int lane, const char *__restrict__ glb_tab) {
  bool any = false;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const float d8 = 8.0f * r.inc[c];
    // after 8 advances the position is at most phase + d8 (+ rounding << 1): both taps stay inside
    // [base, base + SK_WIN) and below the loop end, so neither the wrap nor the folded tap can occur
    const bool fits = d8 <= (float)(SK_WIN - 3) + 0.5f && r.phase[c] + d8 + 2.0f < r.hi[c];
    w.direct[c] = !dead[c] && !fits;
    w.base[c] = (int)r.phase[c];
    any = any || w.direct[c];
    if (!w.direct[c]) {
      const char *src = glb_tab + (r.toff4[c] + (w.base[c] << 2));
      float *dst = win + (c * SK_WIN) * 64 + lane;
#pragma unroll
      for (int k = 0; k < SK_WIN / 4; ++k) {
        const win4_t t = *reinterpret_cast<const win4_t *>(src + 16 * k);
        dst[(4 * k + 0) * 64] = t.x; dst[(4 * k + 1) * 64] = t.y;
        dst[(4 * k + 2) * 64] = t.z; dst[(4 * k + 3) * 64] = t.w;
      }
    }
  }
  w.any_direct = __any(any);
}

// oscillator half of a frame with the window (tame waves only): same phase arithmetic as fast2_osc<.., true, ..>
template <int INTERP>
__device__ __forceinline__ v2f fast2_osc_win(Fast2Regs &r, const WinRegs &w, const float *win, int lane,
                                             const char *__restrict__ glb_tab) {
  const v2f ph0 = r.phase + r.inc;
  const v2f x = ph0 - r.lo;
  const v2f phw = r.lo + (x - r.span);
  v2f ph, s;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const float p = (ph0[c] >= r.hi[c]) ? phw[c] : ph0[c];
    ph[c] = p;
    const int idx = (int)p;
    const int rel = w.direct[c] ? 0 : idx - w.base[c];
    const float *src = win + (c * SK_WIN + rel) * 64 + lane;
    const float ta = src[0];
    if (INTERP == 0) {
      s[c] = ta;
    } else {
      const float tb = src[64];
      const float frac = p - (float)idx;
      s[c] = ta + frac * (tb - ta);
    }
  }
  r.phase = ph;
  if (w.any_direct) {
#pragma unroll
    for (int c = 0; c < 2; ++c)
      if (w.direct[c])
        s[c] = fast2_fetch<false, INTERP, true>(nullptr, glb_tab, r.toff4[c], r.tsize_m1[c], r.lo[c], r.hi[c], ph[c]);
  }
  return s;
}

#define SK_F2_ARGS released[0], released[1], silent[0], silent[1], lds_tab, glb_tab
// one frame (J) / two frames (J, J+1; delay-line roles swap in between, one 4-chain DPP reduction)
#define SK_FAST2_ONE(J, EM_, TAME_)                                                                      \
  {                                                                                                      \
    float l, rr;                                                                                         \
    fast2_frame<TAB_LDS, FILTER, EM_, TAME_, INTERP, MIXED, FMP>(r, e, r.x1, r.x2, r.y1, r.y2, SK_F2_ARGS, l, rr);   \
    SK_REDUCE_AND_STORE(J)                                                                               \
    { v2f t_ = r.x1; r.x1 = r.x2; r.x2 = t_; t_ = r.y1; r.y1 = r.y2; r.y2 = t_; }                        \
  }
#define SK_FAST2_PAIR(J, EM_, TAME_)                                                                     \
  {                                                                                                      \
    float l0, r0, l1, r1;                                                                                \
    fast2_frame<TAB_LDS, FILTER, EM_, TAME_, INTERP, MIXED, FMP>(r, e, r.x1, r.x2, r.y1, r.y2, SK_F2_ARGS, l0, r0);  \
    fast2_frame<TAB_LDS, FILTER, EM_, TAME_, INTERP, MIXED, FMP>(r, e, r.x2, r.x1, r.y2, r.y1, SK_F2_ARGS, l1, r1);  \
    SK_REDUCE4_AND_STORE(J)                                                                              \
  }
// Eight frames (J..J+7) with the cross-lane sum through LDS instead of the VALU: every lane parks its (L,R) of
// 8 frames in a wave-private transposition tile xp[8][65] (one ds_write_b64 per frame, row stride 65 keeps the
// column reads conflict-free); then lane (f = lane&7, seg = lane>>3) adds the 8 lanes of segment seg for frame f,
// the 8 segment sums go through xq[8][8], and lanes 0..7 finish one frame each.  ~1.75 VALU + 3.25 LDS
// instructions per frame instead of 12 VALU.  All traffic stays inside one wavefront (LDS executes a wave's
// accesses in order): no s_barrier.
#define SK_WAVE_SYNC()                                      \
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
  __builtin_amdgcn_wave_barrier();                          \
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#define SK_XT2 68   /* floats per tile row (see SK_XT in skred_render_fast.hip) */
#define SK_FAST2_LDS_BLOCK_Z(J, EM_, LOZ_, MUTE_)                                                                       \
  {                                                                                                      \
    float *const xt_ = reinterpret_cast<float *>(xp);   /* the wave's tile: [8 frames][SK_XT2] folded (L | R) pair sums */ \
    /* software pipeline: the table gather of the NEXT frame is issued before the biquad/gain chain of the   \
       current one (the source order matters: the compiler may not move an LDS read above the tile write) */  \
    v2f s0_ = fast2_osc<TAB_LDS, true, INTERP, LOZ_>(r, lds_tab, glb_tab);                                      \
    _Pragma("unroll") for (int q_ = 0; q_ < 8; q_ += 2) {                                                \
      float l0, r0, l1, r1;                                                                              \
      const v2f s1_ = fast2_osc<TAB_LDS, true, INTERP, LOZ_>(r, lds_tab, glb_tab);                              \
      fast2_post<FILTER, EM_, true, MIXED, MUTE_>(r, e, s0_, r.x1, r.x2, r.y1, r.y2, released[0], released[1], silent[0], silent[1], l0, r0); \
      if (q_ < 6) s0_ = fast2_osc<TAB_LDS, true, INTERP, LOZ_>(r, lds_tab, glb_tab);                            \
      fast2_post<FILTER, EM_, true, MIXED, MUTE_>(r, e, s1_, r.x2, r.x1, r.y2, r.y1, released[0], released[1], silent[0], silent[1], l1, r1); \
      xt_[q_ * SK_XT2 + lane] = fold_lr(l0, r0);                                                         \
      xt_[(q_ + 1) * SK_XT2 + lane] = fold_lr(l1, r1);                                                   \
    }                                                                                                    \
    SK_WAVE_SYNC()                                                                                       \
    {   /* lane (f, seg) adds 8 floats of tile row f: segments 0..3 hold L pair sums, 4..7 R pair sums */ \
      const float4 *src_ = reinterpret_cast<const float4 *>(xt_ + (lane & 7) * SK_XT2 + (lane >> 3) * 8); \
      const float4 a_ = src_[0], b_ = src_[1];                                                           \
      float t_ = ((((((a_.x + a_.y) + a_.z) + a_.w) + b_.x) + b_.y) + b_.z) + b_.w;                      \
      t_ = row_pair_add(row_ror8_add(t_));       /* -> lanes 0..7 (L of frames 0..7), 32..39 (R) */      \
      if ((lane & 24) == 0) reinterpret_cast<float *>(&wsum[wave * SK_CHUNK + (J) + (lane & 7)])[lane >> 5] = t_; \
    }                                                                                                    \
    SK_WAVE_SYNC()                                                                                       \
  }
// The same eight frames for a wave that is not tame (frequency-modulated carriers of an FMP bank, loops out of range ...):
// general frames, no gather ahead (a carrier's increment needs the modulator's sample of the frame before), same tile.
#define SK_FAST2_LDS_BLOCK_U(J, EM_)                                                                     \
  {                                                                                                      \
    float *const xt_ = reinterpret_cast<float *>(xp);                                                    \
    _Pragma("unroll") for (int q_ = 0; q_ < 8; q_ += 2) {                                                \
      float l0, r0, l1, r1;                                                                              \
      fast2_frame<TAB_LDS, FILTER, EM_, false, INTERP, MIXED, FMP>(r, e, r.x1, r.x2, r.y1, r.y2, SK_F2_ARGS, l0, r0);  \
      fast2_frame<TAB_LDS, FILTER, EM_, false, INTERP, MIXED, FMP>(r, e, r.x2, r.x1, r.y2, r.y1, SK_F2_ARGS, l1, r1);  \
      xt_[q_ * SK_XT2 + lane] = fold_lr(l0, r0);                                                         \
      xt_[(q_ + 1) * SK_XT2 + lane] = fold_lr(l1, r1);                                                   \
    }                                                                                                    \
    SK_WAVE_SYNC()                                                                                       \
    {                                                                                                    \
      const float4 *src_ = reinterpret_cast<const float4 *>(xt_ + (lane & 7) * SK_XT2 + (lane >> 3) * 8); \
      const float4 a_ = src_[0], b_ = src_[1];                                                           \
      float t_ = ((((((a_.x + a_.y) + a_.z) + a_.w) + b_.x) + b_.y) + b_.z) + b_.w;                      \
      t_ = row_pair_add(row_ror8_add(t_));                                                               \
      if ((lane & 24) == 0) reinterpret_cast<float *>(&wsum[wave * SK_CHUNK + (J) + (lane & 7)])[lane >> 5] = t_; \
    }                                                                                                    \
    SK_WAVE_SYNC()                                                                                       \
  }
// `loz` (wave-uniform, set once per pass): see fast2_osc
#define SK_FAST2_LDS_BLOCK(J, EM_)                                                                       \
  { if (loz) SK_FAST2_LDS_BLOCK_Z(J, EM_, true, false) else SK_FAST2_LDS_BLOCK_Z(J, EM_, false, false) }
// the same for a tame wave with muted live lanes (tame_m): their contribution is selected away, synth.c:596
#define SK_FAST2_LDS_BLOCK_M(J, EM_)                                                                     \
  { if (loz) SK_FAST2_LDS_BLOCK_Z(J, EM_, true, true) else SK_FAST2_LDS_BLOCK_Z(J, EM_, false, true) }
// Eight frames of a tame wave that holds listed voices (in-place instantiation): SK_FAST2_LDS_BLOCK_Z with the listed voices'
// gains of the block's frames in gtile[8][SK_GT_RANKS] (rank: the listed voice's number within the wave), parked there by
// the block before.  This block fetches the NEXT eight frames at its start -- lane (frame = lane & 7, rank = (lane >> 3) +
// 8 i) for as many i as the wave has listed voices, srank[rank] = where that voice's row starts -- and parks them behind its
// own frames (one wavefront: its LDS accesses execute in order).
#ifndef SK_GT_RANKS
#define SK_GT_RANKS 32
#endif
#define SK_GT_LDS_FLOATS (8 * SK_GT_RANKS + SK_GT_RANKS)   /* per wave: gtile + srank */
/* (row offsets kept in two registers per lane for the first sixteen ranks instead of read from srank, 16 ranks instead of 32,
   a double-buffered tile: all measured within 1 % of this form -- tools/ab_gt.sh) */
#define SK_GT_FETCH(I, F0)   /* -> pf_[I] */                                                              \
  if (SK_GT_RANKS > 8 * (I) && gcnt > 8 * (I) && (lane >> 3) + 8 * (I) < gcnt)   /* (wave-uniform: any rank of this round; then: this lane's rank exists) */ \
    pf_[I] = a.env_gain[srank[(lane >> 3) + 8 * (I)] + ((F0) + (lane & 7))];
#define SK_GT_PARK(I)                                                                                    \
  if (SK_GT_RANKS > 8 * (I) && gcnt > 8 * (I)) gtile[(lane & 7) * SK_GT_RANKS + (lane >> 3) + 8 * (I)] = pf_[I];
#define SK_FAST2_LDS_BLOCK_G(J, EM_, LOZ_)                                                                  \
  {                                                                                                      \
    float *const xt_ = reinterpret_cast<float *>(xp);                                                    \
    float pf_[4] = {0.0f, 0.0f, 0.0f, 0.0f};                                                             \
    SK_GT_FETCH(0, c0 + (J) + 8) SK_GT_FETCH(1, c0 + (J) + 8) SK_GT_FETCH(2, c0 + (J) + 8) SK_GT_FETCH(3, c0 + (J) + 8) \
    v2f s0_ = fast2_osc<TAB_LDS, true, INTERP, LOZ_>(r, lds_tab, glb_tab);                               \
    _Pragma("unroll") for (int q_ = 0; q_ < 8; q_ += 2) {                                                \
      float l0, r0, l1, r1;                                                                              \
      const v2f s1_ = fast2_osc<TAB_LDS, true, INTERP, LOZ_>(r, lds_tab, glb_tab);                       \
      r.gt.x = gtile[q_ * SK_GT_RANKS + grank[0]]; r.gt.y = gtile[q_ * SK_GT_RANKS + grank[1]];          \
      fast2_post<FILTER, EM_, true, MIXED, false>(r, e, s0_, r.x1, r.x2, r.y1, r.y2, released[0], released[1], silent[0], silent[1], l0, r0); \
      if (q_ < 6) s0_ = fast2_osc<TAB_LDS, true, INTERP, LOZ_>(r, lds_tab, glb_tab);                     \
      r.gt.x = gtile[(q_ + 1) * SK_GT_RANKS + grank[0]]; r.gt.y = gtile[(q_ + 1) * SK_GT_RANKS + grank[1]]; \
      fast2_post<FILTER, EM_, true, MIXED, false>(r, e, s1_, r.x2, r.x1, r.y2, r.y1, released[0], released[1], silent[0], silent[1], l1, r1); \
      xt_[q_ * SK_XT2 + lane] = fold_lr(l0, r0);                                                         \
      xt_[(q_ + 1) * SK_XT2 + lane] = fold_lr(l1, r1);                                                   \
    }                                                                                                    \
    SK_GT_PARK(0) SK_GT_PARK(1) SK_GT_PARK(2) SK_GT_PARK(3)                                              \
    SK_WAVE_SYNC()                                                                                       \
    {                                                                                                    \
      const float4 *src_ = reinterpret_cast<const float4 *>(xt_ + (lane & 7) * SK_XT2 + (lane >> 3) * 8); \
      const float4 a_ = src_[0], b_ = src_[1];                                                           \
      float t_ = ((((((a_.x + a_.y) + a_.z) + a_.w) + b_.x) + b_.y) + b_.z) + b_.w;                      \
      t_ = row_pair_add(row_ror8_add(t_));                                                               \
      if ((lane & 24) == 0) reinterpret_cast<float *>(&wsum[wave * SK_CHUNK + (J) + (lane & 7)])[lane >> 5] = t_; \
    }                                                                                                    \
    SK_WAVE_SYNC()                                                                                       \
  }
// a chunk of a wave with listed voices: staged 8-frame blocks (the wave is tame and holds at most SK_GT_RANKS of them), then --
// or instead -- single general frames with every listed lane reading its own row
#define SK_FAST2_CHUNK_G()                                                                               \
  {                                                                                                      \
    int j = 0;                                                                                           \
    if (gstaged) {                                                                                       \
      if (loz) for (; j + 8 <= cn; j += 8) SK_FAST2_LDS_BLOCK_G(j, 6, true)                              \
      else for (; j + 8 <= cn; j += 8) SK_FAST2_LDS_BLOCK_G(j, 6, false)                                 \
    }                                                                                                    \
    if (j < cn) {                                                                                        \
      const int go0_ = r.lst[0] ? a.env_list[vidx[0]] * a.env_gain_stride : 0;                           \
      const int go1_ = r.lst[1] ? a.env_list[vidx[1]] * a.env_gain_stride : 0;                           \
      for (; j < cn; ++j) {                                                                              \
        r.gt.x = a.env_gain[go0_ + (c0 + j)];                                                            \
        r.gt.y = a.env_gain[go1_ + (c0 + j)];                                                            \
        SK_FAST2_ONE(j, 6, false)                                                                        \
      }                                                                                                  \
      /* (if staged blocks follow in the next chunk -- a ragged chunk is the block's last -- nothing is parked for them) */ \
    }                                                                                                    \
  }
// Eight frames (J..J+7) of a tame wave of a global-table bank through the table windows; DPP pair reductions.
#define SK_FAST2_WIN_BLOCK(J, EM_)                                                                       \
  {                                                                                                      \
    WinRegs wr_;                                                                                         \
    fast2_win_fill(r, dead, wr_, win, lane, glb_tab);                                                    \
    _Pragma("unroll") for (int q_ = 0; q_ < 8; q_ += 2) {                                                \
      float l0, r0, l1, r1;                                                                              \
      const v2f s0_ = fast2_osc_win<INTERP>(r, wr_, win, lane, glb_tab);                                  \
      fast2_post<FILTER, EM_, true, MIXED>(r, e, s0_, r.x1, r.x2, r.y1, r.y2, released[0], released[1], silent[0], silent[1], l0, r0); \
      const v2f s1_ = fast2_osc_win<INTERP>(r, wr_, win, lane, glb_tab);                                  \
      fast2_post<FILTER, EM_, true, MIXED>(r, e, s1_, r.x2, r.x1, r.y2, r.y1, released[0], released[1], silent[0], silent[1], l1, r1); \
      SK_REDUCE4_AND_STORE((J) + q_)                                                                     \
    }                                                                                                    \
  }
// a whole chunk of cn frames in mode EM_: LDS blocks of 8 when tame, DPP pairs otherwise, single-frame tail
#if SK_LDS_REDUCE
#define SK_FAST2_CHUNK(EM_)                                                     \
  {                                                                             \
    int j = 0;                                                                  \
    if (tame) { if (TAB_LDS) { if ((EM_) == 0 && fast2_smoother_stalled(r)) for (; j + 8 <= cn; j += 8) SK_FAST2_LDS_BLOCK(j, (EM_) == 0 ? 3 : (EM_)) \
                               else for (; j + 8 <= cn; j += 8) SK_FAST2_LDS_BLOCK(j, EM_) } \
                else for (; j + 8 <= cn; j += 8) SK_FAST2_WIN_BLOCK(j, EM_)     \
                for (; j + 1 < cn; j += 2) SK_FAST2_PAIR(j, EM_, true)          \
                if (j < cn) SK_FAST2_ONE(j, EM_, true) }                        \
    else      { if (tame_m && TAB_LDS) { if ((EM_) == 0 && fast2_smoother_stalled(r)) for (; j + 8 <= cn; j += 8) SK_FAST2_LDS_BLOCK_M(j, (EM_) == 0 ? 3 : (EM_)) \
                                         else for (; j + 8 <= cn; j += 8) SK_FAST2_LDS_BLOCK_M(j, EM_) } \
                else if (FMP && TAB_LDS) { if ((EM_) == 0 && fast2_smoother_stalled(r)) for (; j + 8 <= cn; j += 8) SK_FAST2_LDS_BLOCK_U(j, (EM_) == 0 ? 3 : (EM_)) \
                                           else for (; j + 8 <= cn; j += 8) SK_FAST2_LDS_BLOCK_U(j, EM_) } \
                for (; j + 1 < cn; j += 2) SK_FAST2_PAIR(j, EM_, false)         \
                if (j < cn) SK_FAST2_ONE(j, EM_, false) }                       \
  }
#else
#define SK_FAST2_CHUNK(EM_)                                                     \
  {                                                                             \
    int j = 0;                                                                  \
    if (tame) { for (; j + 1 < cn; j += 2) SK_FAST2_PAIR(j, EM_, true)          \
                if (j < cn) SK_FAST2_ONE(j, EM_, true) }                        \
    else      { for (; j + 1 < cn; j += 2) SK_FAST2_PAIR(j, EM_, false)         \
                if (j < cn) SK_FAST2_ONE(j, EM_, false) }                       \
  }
#endif
// wave sums of the chunk -> this workgroup's partial-mix row (ACCUM_: add to what is there; `publish`, a bool in scope:
// this pass completes the row, so the values leave as write-through stores -- sk_row_store).  wsum is double
// buffered: the waves go on writing the next chunk into the other half while the first 2*cn threads drain this
// one, so one barrier per chunk is enough (a wave can only reach the half being drained after the NEXT barrier,
// which the draining threads reach after their reads).
#define SK_FAST2_FLUSH(ACCUM_)                                                   \
  __syncthreads();                                                              \
  if (tid < 2 * cn) {                                                           \
    const float *w_ = reinterpret_cast<const float *>(wsum);                    \
    float s_ = w_[0 * 2 * SK_CHUNK + tid];                                      \
    _Pragma("unroll") for (int w2_ = 1; w2_ < NW; ++w2_) s_ += w_[w2_ * 2 * SK_CHUNK + tid]; \
    sk_row_store(row_ptr + (size_t)c0 * 2 + tid, s_, !(ACCUM_), publish);   \
  }                                                                             \
  wsum = (wsum == wsum0) ? wsum0 + NW * SK_CHUNK : wsum0;

// load the two voices of this lane (vbase + lane, vbase + 64 + lane); returns whether the wave is tame
// a voice that contributes nothing to this launch: inert numbers -> exact zeros, table index 0 (see sk_render_fast_kernel)
__device__ __forceinline__ void fast2_make_inert(Fast2Regs &r, Env2Regs &e, int c) {
  r.inc[c] = 0.0f; r.lo[c] = 0.0f; r.hi[c] = 1.0f; r.phase[c] = 0.0f;
  r.toff4[c] = 0; r.tsize_m1[c] = 0;
  r.k[c] = 0.0f; r.sgain[c] = 0.0f; e.ampv[c] = 0.0f; r.gain_const[c] = 0.0f;
  r.b0[c] = r.b1[c] = r.b2[c] = r.a1[c] = r.a2[c] = 0.0f;
  r.x1[c] = r.x2[c] = r.y1[c] = r.y2[c] = 0.0f;
  r.pan_lr[c] = (v2f){0.0f, 0.0f}; r.rw[c] &= ~SKR_ENV_ACTIVE;
}

// vidx[c]: the lane's two voices (sk_render_fast2_kernel: vbase + c*64 + lane of its 128-voice slice; sk_render_env2_kernel:
// two entries of the hand-over list); absent[c]: no voice in that slot (the list's ragged end) -- treated as dead and
// never stored.
template <bool FILTER, bool ENV, bool MIXED, int FMP = 0>
__device__ __forceinline__ bool fast2_load(const sk_render_args_t &a, const int vidx[2], const bool absent[2], int lane, Fast2Regs &r,
                                           Env2Regs &e, bool dead[2], bool silent[2], bool released[2],
                                           uint64_t t_start[2], uint64_t t_release[2], bool &tame_m) {
  float inc_raw1 = 0.0f;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int v = vidx[c];
    const uint4 osc = *reinterpret_cast<const uint4 *>(&a.ro[SKP_OSC][v]);
    const uint4 tab = *reinterpret_cast<const uint4 *>(&a.ro[SKP_TAB][v]);
    const uint4 gn = *reinterpret_cast<const uint4 *>(&a.ro[SKP_GAIN][v]);
    const uint4 s0 = *reinterpret_cast<const uint4 *>(&a.rw[SKS_OSC][v]);
    const uint4 s1 = *reinterpret_cast<const uint4 *>(&a.rw[SKS_FILT][v]);
    const uint4 s2 = *reinterpret_cast<const uint4 *>(&a.rw[SKS_MISC][v]);
    r.inc[c] = __uint_as_float(osc.x); r.lo[c] = __uint_as_float(osc.y);
    r.hi[c] = __uint_as_float(osc.z);
    if (c == 1) inc_raw1 = r.inc[1];                   // voice_phase_inc[m], whatever m's own state is
    e.ampv[c] = __uint_as_float(osc.w);
    r.toff4[c] = (int)tab.x << 2; r.tsize_m1[c] = (int)tab.y - 1;
    const uint32_t flags = tab.z;
    e.velv[c] = __uint_as_float(gn.x); r.k[c] = __uint_as_float(gn.y);
    r.b0[c] = __uint_as_float(gn.z);   r.b1[c] = __uint_as_float(gn.w);
    r.phase[c] = __uint_as_float(s0.x); r.sgain[c] = __uint_as_float(s0.y);
    r.x1[c] = __uint_as_float(s0.z);    r.x2[c] = __uint_as_float(s0.w);
    r.y1[c] = __uint_as_float(s1.x);    r.y2[c] = __uint_as_float(s1.y);
    r.sample[c] = __uint_as_float(s1.z); r.rw[c] = s1.w;
    r.pan_lr[c].x = __uint_as_float(s2.z); r.pan_lr[c].y = __uint_as_float(s2.w);
    if (FMP == 2 && c == 0) { r.misc_x = s2.x; r.misc_y = s2.y; }
    r.b2[c] = 0.0f; r.a1[c] = 0.0f; r.a2[c] = 0.0f;
    if (FILTER) {
      const uint4 fl = *reinterpret_cast<const uint4 *>(&a.ro[SKP_FILT][v]);
      r.b2[c] = __uint_as_float(fl.x); r.a1[c] = __uint_as_float(fl.y); r.a2[c] = __uint_as_float(fl.z);
    }
    released[c] = false; t_start[c] = 0; t_release[c] = 0;
    r.gain_const[c] = e.ampv[c];                       // no envelope: amp * 1.0f * 1.0f
    e.att[c] = e.attdec[c] = e.dec[c] = e.rel[c] = 0.0f;
    e.susv[c] = 0.0f; e.omsv[c] = 0.0f; e.tf[c] = 0.0f; e.trf[c] = 0.0f;
    if (ENV) {
      const uint4 et = *reinterpret_cast<const uint4 *>(&a.ro[SKP_ENV_T][v]);
      const uint4 es = *reinterpret_cast<const uint4 *>(&a.ro[SKP_ENV_S][v]);
      e.att[c] = __uint_as_float(et.x); e.dec[c] = __uint_as_float(et.y);
      e.susv[c] = __uint_as_float(et.z); e.rel[c] = __uint_as_float(et.w);
      e.attdec[c] = e.att[c] + e.dec[c];               // synth.c:410: decay_start + decay_time
      e.omsv[c] = 1.0f - e.susv[c];                    // synth.c:413
      t_start[c] = ((uint64_t)es.y << 32) | es.x;
      t_release[c] = ((uint64_t)es.w << 32) | es.z;
      released[c] = t_release[c] != 0;                 // synth.c:417
    }
    r.filt[c] = !MIXED || (flags & SKF_FILTER);
    r.fake_active[c] = false;
    if (MIXED && ENV && !(flags & SKF_USE_ENV)) {
      // no envelope on this voice: final = amp * 1.0f (synth.c:580-582).  Rendered as a note held at level 1 with
      // velocity 1 since "now": amp * (1 * 1) is the same float, the clocks stay small, the stage never changes
      e.att[c] = e.dec[c] = e.attdec[c] = e.rel[c] = 0.0f;
      e.susv[c] = 1.0f; e.omsv[c] = 0.0f; e.velv[c] = 1.0f;
      t_start[c] = a.count0; t_release[c] = 0; released[c] = false;
      r.fake_active[c] = !(r.rw[c] & SKR_ENV_ACTIVE);
      r.rw[c] |= SKR_ENV_ACTIVE;
    }
    dead[c] = absent[c] || (r.rw[c] & SKR_FINISHED) || e.ampv[c] == 0.0f || (flags & SKF_INERT);
    silent[c] = dead[c] || (flags & SKF_MUTED);
    if (dead[c]) fast2_make_inert(r, e, c);   // never stored back
  }
  r.span = r.hi - r.lo;
  r.span2 = r.span + r.span;
  if (MIXED) { r.ox1 = r.x1; r.ox2 = r.x2; r.oy1 = r.y1; r.oy2 = r.y2; }
  r.fm_on = false; r.fm_k = 0.0f; r.fm_depth = 0.0f;
  r.am_on = r.am_self = r.pm_on = r.pm_self = r.pan_dirty = false; r.am_depth = r.pm_depth = r.mprev = 0.0f;
  if (FMP != 2) { r.misc_x = 0; r.misc_y = 0; }
  if (FMP) {
    const uint4 mi = *reinterpret_cast<const uint4 *>(&a.ro[SKP_MODI][vidx[0]]);
    const uint4 mf = *reinterpret_cast<const uint4 *>(&a.ro[SKP_MODF][vidx[0]]);
    const int own = vidx[0] & 63;                    // (the host vouches that a modulator is the voice after it, or -- amplitude,
    r.fm_on = !dead[0] && (int)mi.x >= 0;            //  pan -- the voice itself)
    r.fm_k = inc_raw1 * __uint_as_float(mf.y);
    r.fm_depth = __uint_as_float(mf.x);
    if (FMP == 2) {
      r.am_on = !dead[0] && (int)mi.y >= 0; r.am_self = (int)mi.y == own; r.am_depth = __uint_as_float(mf.z);
      r.pm_on = !dead[0] && (int)mi.z >= 0; r.pm_self = (int)mi.z == own; r.pm_depth = __uint_as_float(mf.w);
    }
  }
  // a modulated increment may be negative or long (general wrap); a modulated gain or pan needs the general frames too
  bool tame_lane = !(FMP && (r.fm_on || r.am_on || r.pm_on)), muted_lane = false;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    muted_lane = muted_lane || (silent[c] && !dead[c]);
    tame_lane = tame_lane &&
                (dead[c] || (r.inc[c] >= 0.0f && r.inc[c] <= 0.5f * r.span[c] && r.phase[c] >= r.lo[c] &&
                             r.phase[c] <= r.hi[c] && r.lo[c] >= 0.0f && r.hi[c] <= (float)(r.tsize_m1[c] + 1)));
  }
  tame_m = __all(tame_lane);                 // tame but for muted live lanes (the LDS blocks can select those away)
  return tame_m && !__any(muted_lane);
}

// skip[c]: the slot is not this kernel's to store (a voice handed to sk_render_env2_kernel, or no voice at all)
template <bool MIXED>
__device__ __forceinline__ void fast2_store(const sk_render_args_t &a, const Fast2Regs &r, const bool dead[2],
                                            const int vidx[2], const bool skip[2]) {
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int v = vidx[c];
    if (skip[c]) continue;
    if (!dead[c]) {
      uint4 s0, s1;
      const bool keep = MIXED && !r.filt[c];           // an unfiltered voice's delay line goes back as it came
      s0.x = __float_as_uint(r.phase[c]); s0.y = __float_as_uint(r.sgain[c]);
      s0.z = __float_as_uint(keep ? r.ox1[c] : r.x1[c]);    s0.w = __float_as_uint(keep ? r.ox2[c] : r.x2[c]);
      s1.x = __float_as_uint(keep ? r.oy1[c] : r.y1[c]);    s1.y = __float_as_uint(keep ? r.oy2[c] : r.y2[c]);
      s1.z = __float_as_uint(r.sample[c]); s1.w = (MIXED && r.fake_active[c]) ? (r.rw[c] & ~SKR_ENV_ACTIVE) : r.rw[c];
      *reinterpret_cast<uint4 *>(&a.rw[SKS_OSC][v]) = s0;
      *reinterpret_cast<uint4 *>(&a.rw[SKS_FILT][v]) = s1;
      if (c == 0 && r.pan_dirty)   // pan modulation rewrote voice_pan_left / _right (synth.c:600-601)
        *reinterpret_cast<uint4 *>(&a.rw[SKS_MISC][v]) = make_uint4(r.misc_x, r.misc_y, __float_as_uint(r.pan_lr[0].x), __float_as_uint(r.pan_lr[0].y));
    } else {
      reinterpret_cast<uint32_t *>(&a.rw[SKS_FILT][v])[2] = 0u;     // voice_sample = 0, synth.c:532,538
    }
  }
}

// NW (a constexpr in scope): wavefronts per workgroup = 128-voice slices per workgroup pass
#define SK_FAST2_PROLOGUE_(WORKS, ROWS, GAIN_WG)                                                      \
  extern __shared__ float lds[];                                                                     \
  float2 *const wsum0 = reinterpret_cast<float2 *>(lds + (TAB_LDS ? a.lds_table_floats : 0));        \
  float2 *wsum = wsum0;                      /* [2][NW][SK_CHUNK]: see SK_FAST2_FLUSH */              \
  const char *lds_tab = reinterpret_cast<const char *>(lds);                                         \
  const char *glb_tab = reinterpret_cast<const char *>(a.tables);                                    \
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;                                     \
  /* wave-private: the reduction tile [8 frames][SK_XT2] of folded (L | R) pair sums (it used to be a float2 tile [8][65] and a \
     second stage xq[64]: 4.7 KB per wave, of which 2.2 are in use since the fold -- 20 KB per workgroup that kept banks with \
     more than 31 KB of tables at one workgroup per CU) */                                            \
  float2 *xp = reinterpret_cast<float2 *>(reinterpret_cast<float *>(wsum0 + 2 * NW * SK_CHUNK) + wave * (8 * SK_XT2)); \
  /* global-table banks: the same LDS holds the wave's table windows instead (2 voices x SK_WIN x 64 lanes) */ \
  float *win = reinterpret_cast<float *>(wsum0 + 2 * NW * SK_CHUNK) + wave * (2 * SK_WIN * 64);      \
  (void)xp; (void)win;                                                                               \
  const int bid = (int)blockIdx.x - ((GAIN_WG) ? a.wg_shift : 0);   /* row of the partial mix; -1: the gain workgroup */ \
  if ((GAIN_WG) && bid < 0) { sk_finish_block(a, bid, tid, NW * 64, reinterpret_cast<int *>(lds)); return; } \
  const int n_flags = a.n_groups * 2;      /* 128-voice wave slices; env_off[n_flags] = the length of the motion list */ \
  if (TAB_LDS && (WORKS)) {                 /* (a workgroup without a pass needs no tables) */           \
    const int n4 = a.lds_table_floats >> 2;                                                          \
    const float4 *src4 = reinterpret_cast<const float4 *>(a.tables);                                 \
    float4 *dst4 = reinterpret_cast<float4 *>(lds);                                                  \
    sk_stage_tables<NW * 64>(src4, dst4, n4, tid);                                                   \
    __syncthreads();                                                                                 \
  }                                                                                                  \
  float *const row_ptr = (ROWS) + (size_t)bid * (size_t)a.num_frames * 2;   /* this workgroup's row */ \
  const int n_pass = (a.n_groups * SK_GROUP) / (NW * 128);   /* workgroup passes over the (padded) bank */
#define SK_FAST2_PROLOGUE() SK_FAST2_PROLOGUE_(true, a.partial, true)

#ifndef SK_FAST2_MIN_WAVES
#define SK_FAST2_MIN_WAVES 4     /* <= 128 VGPRs */
#endif

// Constant-level groups.  ENV: the bank uses envelopes, so every group is classified first.
// Workgroup shape: LDS-table banks run 8 wavefronts per workgroup (1024 voices per pass) so that two workgroups --
// 16 waves, 4 per SIMD -- share a CU's LDS with ONE copy of the tables each (four per SIMD needs <= 40 KB per
// 256-thread workgroup otherwise, and tables + tiles take ~46 KB); global-table banks keep 4 (their table windows
// scale with the wave count).
/* store what this kernel rendered: not the voices on the motion list (lane masks m0 / m1) */
#define SK_FAST2_STORE_MINE()                                                                        \
  {                                                                                                  \
    const bool skip_[2] = {(bool)((m0 >> lane) & 1), (bool)((m1 >> lane) & 1)};                      \
    fast2_store<MIXED>(a, r, dead, vidx, skip_);                                                     \
  }
template <bool TAB_LDS> struct Fast2Shape { static constexpr int NW = TAB_LDS ? SK_FAST2_NW_LDS : 4; };

// FMP: a two-operator FM bank (SKM_FM_PAIR) -- a lane holds voices 2i and 2i+1 of its slice, carrier and modulator.
// FMP == 2 (SKM_PAIR_AP): some carrier's amplitude or pan is modulated too (by the voice after it or by itself).
// GT: the in-place instantiation (LDS-table banks with envelopes; a.env_gain set): listed voices stay in their lanes, their
// gains come from the rows sk_gain_kernel wrote just before on this stream.  A kernel of its own so that the steady
// instantiation keeps its registers (it sits at 126 of 128).
// PROBE: the same kernel compiled in a translation unit with -DSK_PROBE_TU (fast2_post then also writes the probe rows of
// skred_bank_set_probe): a template parameter only so that its instantiations are symbols of their own.
template <bool TAB_LDS, bool FILTER, bool ENV, int INTERP, bool MIXED, int FMP = 0, bool GT = false, bool PROBE = false>
__global__ __launch_bounds__(Fast2Shape<TAB_LDS>::NW * 64, SK_FAST2_MIN_WAVES) void sk_render_fast2_kernel(const sk_render_args_t a) {
  constexpr int NW = Fast2Shape<TAB_LDS>::NW;
  SK_FAST2_PROLOGUE()
  (void)n_flags;
  // GT: per wave, behind the tiles: gtile[8][SK_GT_RANKS] (the listed voices' gains of one 8-frame block) and
  // srank[SK_GT_RANKS] (rank -> where that voice's row starts in a.env_gain)
  float *const gtile = reinterpret_cast<float *>(wsum0 + 2 * NW * SK_CHUNK) + NW * (8 * SK_XT2) + wave * SK_GT_LDS_FLOATS;
  int *const srank = reinterpret_cast<int *>(gtile + 8 * SK_GT_RANKS);
  (void)gtile; (void)srank;
  bool first_pass = true;
  bool row_published = false;
  for (int g = bid; g < n_pass; g += a.n_rows) {
    const bool publish = a.finish && g + a.n_rows >= n_pass;      // the pass that completes this workgroup's row
    Fast2Regs r;
    Env2Regs e;
    bool dead[2], silent[2], released[2];
    uint64_t t_start[2], t_release[2];
    bool tame_m;
    const int slice = g * NW + wave;
    const int vidx[2] = {FMP ? slice * 128 + 2 * lane : slice * 128 + lane, FMP ? slice * 128 + 2 * lane + 1 : slice * 128 + 64 + lane};
    const bool absent[2] = {false, false};
    uint64_t m0 = 0, m1 = 0;                          // the lanes whose voice 0 / 1 is on the motion list (wave-uniform masks)
    const bool tame = fast2_load<FILTER, ENV, MIXED, FMP>(a, vidx, absent, lane, r, e, dead, silent, released, t_start, t_release, tame_m);
    const bool loz = __all(r.lo.x == 0.0f && r.lo.y == 0.0f);
    (void)loz;
#ifdef SK_PROBE_TU
    r.probe_stride = a.n_probe;
    r.probe[0] = sk_probe_row(a, vidx[0], silent[0]);
    r.probe[1] = sk_probe_row(a, vidx[1], silent[1]);
    r.probe_any = __any(r.probe[0] != nullptr || r.probe[1] != nullptr);
#endif
    bool wave_ok = true;
    r.lst[0] = r.lst[1] = false; r.gt = (v2f){0.0f, 0.0f};
    int grank[2] = {0, 0};            // GT: the lane's listed voices' numbers within the wave (ascending voice order)
    int gcnt = 0;                     // GT: listed voices of the wave
    bool gstaged = false;             // GT: 8-frame blocks through gtile (a tame wave with at most SK_GT_RANKS listed voices)
    (void)grank; (void)gcnt; (void)gstaged;
    if (ENV) {
      // The motion list: voices whose envelope may be in motion sit this launch out here -- pan gains at zero, exact zeros
      // into the mix, nothing stored -- and are rendered by sk_render_env2_kernel, which reads the same bits (collected into
      // full waves) on the other stream.  (Notes start and end all the time in a live bank: almost every 128-voice slice
      // holds a few such voices.)  a.skip_env2: the list is empty, nobody runs beside this kernel.
      if (!a.skip_env2) {
        const int su = __builtin_amdgcn_readfirstlane(slice);
        const uint64_t w0 = a.mask_cur[2 * su], w1 = a.mask_cur[2 * su + 1];     // voices slice*128 + 0..63 / + 64..127
        if (FMP) {                                    // the lane's pair (2 lane, 2 lane + 1) travels together
          const uint64_t w = lane < 32 ? w0 : w1;
          m0 = m1 = __ballot(((w >> ((2 * lane) & 63)) & 3ull) != 0);
        } else {
          m0 = w0; m1 = w1;
        }
      }
      // every voice that is NOT on the list holds a constant level for the whole launch: the stage on its first frame (stages
      // 0 / 3 / 5 are absorbing; a voice leaves them only through a control action, and those put it on the list)
      bool stray = false;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const sk_motion_t mo = sk_env_motion(a.count0 + 1, dead[c], (r.rw[c] & SKR_ENV_ACTIVE) != 0, t_start[c], t_release[c], e.att[c],
                                             e.attdec[c], e.rel[c], e.susv[c], e.ampv[c], e.velv[c], r.k[c], r.sgain[c]);
        r.gain_const[c] = mo.gain_const;                                         // synth.c:582,588
        if (!dead[c] && mo.code == 5 && !(GT && (((c ? m1 : m0) >> lane) & 1))) r.rw[c] &= ~SKR_ENV_ACTIVE;   // synth.c:429 (a listed voice's flag is sk_gain_kernel's)
        stray = stray || (mo.moving && !(((c ? m1 : m0) >> lane) & 1));
      }
      {   // cross-check of the list (unreachable by construction: see DESIGN "The motion list"): counted, the host rebuilds the list
        const uint64_t sb = __ballot(stray);
        if (sb != 0 && lane == 0) atomicAdd(a.violations, (uint32_t)__popcll(sb));
      }
      if (GT && (m0 | m1) != 0) {
        // in place: a listed voice keeps its lane; the gain its smoother is fed (amp * (level * velocity), synth.c:582-588)
        // comes from the row sk_gain_kernel wrote for it, frame by frame
        const uint64_t below = ((uint64_t)1 << lane) - 1;
        const int n0 = __popcll(m0);
        gcnt = n0 + __popcll(m1);
        if (lane == 0) atomicAdd(a.env_count, (uint32_t)gcnt);   // the list's length, for the block's report (sk_final_cols)
        grank[0] = __popcll(m0 & below);
        grank[1] = n0 + __popcll(m1 & below);
        gstaged = tame && TAB_LDS && gcnt <= SK_GT_RANKS;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const bool listed = ((c ? m1 : m0) >> lane) & 1;
          r.lst[c] = listed && !dead[c];
          if (listed && gstaged) srank[grank[c]] = a.env_list[vidx[c]] * a.env_gain_stride;
          if (!listed || !gstaged) grank[c] = 0;
        }
        if (gstaged) {                                  // the first block's gains
          SK_WAVE_SYNC()
          float pf_[4] = {0.0f, 0.0f, 0.0f, 0.0f};
          SK_GT_FETCH(0, 0) SK_GT_FETCH(1, 0) SK_GT_FETCH(2, 0) SK_GT_FETCH(3, 0)
          SK_GT_PARK(0) SK_GT_PARK(1) SK_GT_PARK(2) SK_GT_PARK(3)
          SK_WAVE_SYNC()
        }
      } else
      if ((m0 | m1) != 0) {                           // (wave-uniform: the steady state pays for none of this)
#pragma unroll
        for (int c = 0; c < 2; ++c)
          if (((c ? m1 : m0) >> lane) & 1) {   // (two numbers, not fast2_make_inert's twenty-five: the kernel sits at its register budget)
            r.pan_lr[c] = (v2f){0.0f, 0.0f};           // it walks its oscillator for nothing and adds exact zeros
#ifdef SK_PROBE_TU
            r.probe[c] = nullptr;                      // (its probe rows are the envelope kernel's)
#endif
            r.k[c] = 0.0f;                             // its smoother counts as stalled (fast2_smoother_stalled is a wave vote)
            if (FMP == 2) r.am_on = r.pm_on = false;   // (and its pan stays at zero)
          }
#ifdef SK_PROBE_TU
        r.probe_any = __any(r.probe[0] != nullptr || r.probe[1] != nullptr);
#endif
        wave_ok = __any((!dead[0] && !((m0 >> lane) & 1)) || (!dead[1] && !((m1 >> lane) & 1)));   // (nothing left to render: zeros to the chunk sums)
      }
      if (!GT && !__syncthreads_or(wave_ok ? 1 : 0)) {       // nothing of this pass is rendered here
#pragma unroll
        for (int c = 0; c < 2; ++c)                   // (really dead voices still get voice_sample = 0, synth.c:532,538 --
          if (dead[c] && !(((c ? m1 : m0) >> lane) & 1))   //  unless they are on the list: the envelope kernel does it, and a listed pair's carrier reads it)
            reinterpret_cast<uint32_t *>(&a.rw[SKS_FILT][vidx[c]])[2] = 0u;
        continue;
      }
    }
    for (int c0 = 0; c0 < a.num_frames; c0 += SK_CHUNK) {
      const int cn = min(SK_CHUNK, a.num_frames - c0);
      if (GT && (m0 | m1) != 0) {
        SK_FAST2_CHUNK_G()
      } else if (wave_ok) {
        SK_FAST2_CHUNK(0)
      } else if (lane < cn) {
        wsum[wave * SK_CHUNK + lane] = make_float2(0.0f, 0.0f);             // a wave of listed voices adds nothing to the chunk
      }
      SK_FAST2_FLUSH(!first_pass)
    }
    if (GT) { const bool skip_[2] = {false, false}; fast2_store<MIXED>(a, r, dead, vidx, skip_); }
    else SK_FAST2_STORE_MINE()
    first_pass = false;
    row_published = publish;
  }
  if (first_pass) {   // every pass of this workgroup was skipped: its partial-mix row must still exist
    for (int i = tid; i < 2 * a.num_frames; i += NW * 64) row_ptr[i] = 0.0f;
  }
  // (row_published is false when the completing pass was skipped as a whole -- every voice on the list: the row is then
  // copied out by sk_finish_block)
  if (a.finish) sk_finish_block(a, bid, tid, NW * 64, reinterpret_cast<int *>(lds), row_published);
}

// The voices on the motion list, 128 list entries per wave, on a stream of its own beside sk_render_fast2_kernel (which
// sits exactly these voices out).  Rows, ticket and sum are this kernel's own (sk_finish_env).
#ifndef SK_ENV2_MIN_WAVES
#define SK_ENV2_MIN_WAVES 3      /* the envelope machinery wants ~170 VGPRs: 3 waves per SIMD measured best (2: no spills, 4: 220 B of scratch) */
#endif
template <bool TAB_LDS, bool FILTER, int INTERP, bool MIXED, int FMP = 0, bool PROBE = false>
__global__ __launch_bounds__(SK_GROUP, SK_ENV2_MIN_WAVES) void sk_render_env2_kernel(const sk_render_args_t a) {
  constexpr int NW = 4;              // always 512 voices per pass: its register budget allows 3 waves per SIMD anyway
  // the listed voices in ascending order (sk_collect_scan_kernel + sk_collect_expand_kernel, just before on this stream):
  // every workgroup pass takes 512 of them, so the launch costs what those voices cost; the grid is what the device holds at
  // once (a grid of more rendering workgroups than fit runs in rounds, the last one mostly empty) and strides over the passes
  SK_FAST2_PROLOGUE_((int)blockIdx.x < (a.env_off[a.n_groups * 2] + NW * 128 - 1) / (NW * 128), a.env_rows, false)
  (void)n_pass;
  const int n_mine = a.env_off[n_flags];
  const int n_workers = a.n_env_rows;
  const int n_env_pass = (n_mine + NW * 128 - 1) / (NW * 128);
  const int n_used = min(n_workers, n_env_pass);         // workgroups that render (and publish a row)
  bool first_pass = true;
  for (int g = bid; g < n_env_pass; g += n_workers) {
    const bool publish = g + n_workers >= n_env_pass;    // the pass that completes this workgroup's row
    const int p0 = g * (NW * 128) + wave * 128;          // this wave's first list entry
    const bool mine = p0 < n_mine;
    Fast2Regs r;
    Env2Regs e;
    bool dead[2], silent[2], released[2];
    uint64_t t_start[2], t_release[2];
    int vidx[2] = {0, 0};
    bool absent[2] = {true, true};
    bool tame_m = false, tame = false, loz = false;
    if (mine) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int p = FMP ? p0 + 2 * lane + c : p0 + c * 64 + lane;   // (FMP: the list holds whole pairs, even voice first)
        absent[c] = p >= n_mine;
        vidx[c] = absent[c] ? 0 : a.env_list[p];
      }
      tame = fast2_load<FILTER, true, MIXED, FMP>(a, vidx, absent, lane, r, e, dead, silent, released, t_start, t_release, tame_m);
      loz = __all(r.lo.x == 0.0f && r.lo.y == 0.0f);
#ifdef SK_PROBE_TU
      r.probe_stride = a.n_probe;
      r.probe[0] = sk_probe_row(a, vidx[0], silent[0] || absent[0]);
      r.probe[1] = sk_probe_row(a, vidx[1], silent[1] || absent[1]);
      r.probe_any = __any(r.probe[0] != nullptr || r.probe[1] != nullptr);
#endif
    }
    (void)loz; (void)tame; (void)tame_m;
    bool all_const_from_here = false;
    for (int c0 = 0; c0 < a.num_frames; c0 += SK_CHUNK) {
      const int cn = min(SK_CHUNK, a.num_frames - c0);
      if (!mine) {
        if (lane < cn) wsum[wave * SK_CHUNK + lane] = make_float2(0.0f, 0.0f);
      } else {
      bool steady = true, exact = true, ramp = false, step = false;
      float cb_tf[2] = {0.0f, 0.0f}, cb_trf[2] = {0.0f, 0.0f};     // clocks of the frame BEFORE the chunk
      if (!all_const_from_here) {
        const uint64_t base = a.count0 + (uint64_t)c0;              // frame c0+j has now = base + j + 1 (synth.c:521)
        const uint64_t lim = (1ull << 24) - (uint64_t)SK_CHUNK - 2; // x + 1.0f stays exact below 2^24
        bool ex = true, st = true, same = true;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const uint64_t d_on = base - t_start[c], d_off = base - t_release[c];
          ex = ex && (dead[c] || ((d_on < lim) && (!released[c] || d_off < lim)));
          cb_tf[c] = (float)d_on;
          cb_trf[c] = released[c] ? (float)d_off : 0.0f;
          fast2_env_span(r, e, c, dead[c], released[c], (float)(d_on + 1), (float)(d_off + 1),
                         (float)(d_on + (uint64_t)cn), (float)(d_off + (uint64_t)cn), cb_tf[c], cb_trf[c], st, same);
          // a note-on AHEAD of the clock: the wrapped difference reads as "sustain" until the clock gets there, then the attack
          // starts by itself (synth.c:401) -- not a constant level, and not a chunk for float clocks (ex is false: d_on wrapped)
          if (!dead[c] && (r.rw[c] & SKR_ENV_ACTIVE) && (int64_t)(t_start[c] - (base + 1)) > 0) st = false;
        }
        exact = __all(ex);
        steady = __all(st);
        ramp = !steady && exact && __all(same);
        all_const_from_here = steady;       // constant levels are absorbing within a launch
        if (!steady && !ramp && exact) {
          // lanes change stage in this chunk.  Once each, to the stage that follows?  Then the whole chunk keeps the
          // straight-line form with two constant sets per lane (fast2_env_span2) -- with every lane of the wave in motion
          // (the hand-over is voice by voice) nearly every chunk is of this kind, and deciding per 8-frame block costs
          // as much as the frames.
          bool ok2 = true, out[2];
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const uint64_t d_on = base - t_start[c], d_off = base - t_release[c];
            fast2_env_span2(r, e, c, dead[c], released[c], (float)(d_on + 1), (float)(d_off + 1),
                            (float)(d_on + (uint64_t)cn), (float)(d_off + (uint64_t)cn), cb_tf[c], cb_trf[c], ok2, out[c]);
          }
          step = __all(ok2);
          if (step) { if (out[0]) r.rw[0] &= ~SKR_ENV_ACTIVE; if (out[1]) r.rw[1] &= ~SKR_ENV_ACTIVE; }
        }
      }
      if (steady) {
        SK_FAST2_CHUNK(0)
      } else if (ramp) {
        SK_FAST2_CHUNK(1)
      } else if (step) {
        SK_FAST2_CHUNK(4)
      } else if (exact && tame && !(cn & 7)) {
        // some lane changes stage inside this chunk: re-decide per 8-frame block (clocks are exact floats here)
        for (int jb = 0; jb < cn; jb += 8) {
          bool st = true, same = true;
          const float fb = (float)jb;
#pragma unroll
          for (int c = 0; c < 2; ++c)
            fast2_env_span(r, e, c, dead[c], released[c], cb_tf[c] + fb + 1.0f, cb_trf[c] + fb + 1.0f,
                           cb_tf[c] + fb + 8.0f, cb_trf[c] + fb + 8.0f, cb_tf[c] + fb, cb_trf[c] + fb, st, same);
          const bool b_const = __all(st), b_ramp = __all(same);
          bool b_step = false;
          if (!b_const && !b_ramp) {                  // some lane changes stage in this block: once, to the next stage?
            bool ok2 = true, out[2];
#pragma unroll
            for (int c = 0; c < 2; ++c)
              fast2_env_span2(r, e, c, dead[c], released[c], cb_tf[c] + fb + 1.0f, cb_trf[c] + fb + 1.0f,
                              cb_tf[c] + fb + 8.0f, cb_trf[c] + fb + 8.0f, cb_tf[c] + fb, cb_trf[c] + fb, ok2, out[c]);
            b_step = __all(ok2);
            if (b_step) { if (out[0]) r.rw[0] &= ~SKR_ENV_ACTIVE; if (out[1]) r.rw[1] &= ~SKR_ENV_ACTIVE; }
          }
#if SK_LDS_REDUCE
          if (TAB_LDS && b_const) SK_FAST2_LDS_BLOCK(jb, 0)
          else if (TAB_LDS && b_ramp) SK_FAST2_LDS_BLOCK(jb, 1)
          else if (TAB_LDS && b_step) SK_FAST2_LDS_BLOCK(jb, 4)
          else if (!TAB_LDS && b_const) SK_FAST2_WIN_BLOCK(jb, 0)
          else if (!TAB_LDS && b_ramp) SK_FAST2_WIN_BLOCK(jb, 1)
          else if (!TAB_LDS && b_step) SK_FAST2_WIN_BLOCK(jb, 4)
          else
#endif
          {
#pragma unroll
            for (int c = 0; c < 2; ++c) { e.tf[c] = cb_tf[c] + fb; e.trf[c] = cb_trf[c] + fb; }
            for (int q = 0; q < 8; q += 2) {
              float l0, r0, l1, r1;
              e.tf[0] += 1.0f; e.trf[0] += 1.0f; e.tf[1] += 1.0f; e.trf[1] += 1.0f;
              fast2_frame<TAB_LDS, FILTER, 2, true, INTERP, MIXED, FMP>(r, e, r.x1, r.x2, r.y1, r.y2, SK_F2_ARGS, l0, r0);
              e.tf[0] += 1.0f; e.trf[0] += 1.0f; e.tf[1] += 1.0f; e.trf[1] += 1.0f;
              fast2_frame<TAB_LDS, FILTER, 2, true, INTERP, MIXED, FMP>(r, e, r.x2, r.x1, r.y2, r.y1, SK_F2_ARGS, l1, r1);
              { const int J_ = jb + q; SK_REDUCE4_AND_STORE(J_) }
            }
          }
        }
      } else {
        // clocks past 2^24 frames, ragged chunk lengths, untame waves: integer clocks, one frame at a time
        for (int j = 0; j < cn; ++j) {
          const uint64_t now = a.count0 + (uint64_t)(c0 + j) + 1;
          e.tf[0] = (float)(now - t_start[0]); e.trf[0] = (float)(now - t_release[0]);
          e.tf[1] = (float)(now - t_start[1]); e.trf[1] = (float)(now - t_release[1]);
          SK_FAST2_ONE(j, 2, false)
        }
      }
      }   // mine
      SK_FAST2_FLUSH(!first_pass)
    }
    first_pass = false;
    if (mine) {
      fast2_store<MIXED>(a, r, dead, vidx, absent);
      // Who stays on the list?  A voice whose envelope is still in motion on the next block's first frame, whose note-on is
      // still ahead of the clock, or whose amp smoother has not come to rest (sk_env_motion: the list's one definition).
      bool keep[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const sk_motion_t mo = sk_env_motion(a.count0 + (uint64_t)a.num_frames + 1, dead[c], (r.rw[c] & SKR_ENV_ACTIVE) != 0, t_start[c],
                                             t_release[c], e.att[c], e.attdec[c], e.rel[c], e.susv[c], e.ampv[c], e.velv[c], r.k[c], r.sgain[c]);
        keep[c] = !absent[c] && (mo.moving || mo.settling);
      }
      if (FMP) keep[0] = keep[1] = (keep[0] || keep[1]);        // carrier and modulator stay together
#pragma unroll
      for (int c = 0; c < 2; ++c)
        if (keep[c] && !absent[c]) atomicOr(reinterpret_cast<unsigned long long *>(a.mask_next) + (vidx[c] >> 6), 1ull << (vidx[c] & 63));
    }
  }
  sk_finish_env(a, n_used, tid, NW * 64, reinterpret_cast<int *>(lds));
}

// ---------------------------------------------------------------- the motion list: collect, classify (plain TU only)

#if !defined(SK_FAST2_FMP_TU) && !defined(SK_FAST2_GT_TU) && !defined(SK_PROBE_TU)
#define SK_FAST2_PLAIN_TU 1
#endif
#ifdef SK_FAST2_PLAIN_TU
// mask[2 n] (a bit per voice; a 128-voice wave slice = two words) -> counts[n] (listed voices per slice), off[n] exclusive
// prefix sums, off[n] = their total; and the NEXT block's mask zeroed (sk_render_env2_kernel, next on this stream, ORs its
// survivors in).  pairs: two-operator FM banks list whole (2i, 2i+1) pairs -- either bit lists both voices.
// One workgroup, tiles of 8192 slices staged through LDS (coalesced loads in flight together -- a thread walking its
// contiguous share in global memory pays one memory latency per element): every thread sums its contiguous share of the
// counts, the threads' sums are scanned across lanes (shuffles inside each wavefront, the wave totals through LDS), every
// thread writes its prefixes back into the tile, and the tile leaves coalesced.
#define SK_SCAN_TILE 8192
__device__ __forceinline__ uint64_t sk_pair_bits(uint64_t w) { return (w | (w >> 1)) & 0x5555555555555555ull; }   // bit 2i: pair i listed
#define SK_SCAN_NT 1024          /* threads: 16 wavefronts share the tile (256 threads: 13 us at 2^20 voices, latency of one CU's loads) */
__global__ __launch_bounds__(SK_SCAN_NT) void sk_collect_scan_kernel(const uint64_t *__restrict__ mask, uint64_t *__restrict__ mask_next, int n,
                                                              int pairs, int32_t *__restrict__ counts, int32_t *__restrict__ off) {
  __shared__ int tile[SK_SCAN_TILE + SK_SCAN_TILE / 32];   // element k lives at k + k/32: a thread's 32 counts stay contiguous, the
                                                           // threads' shares start in different banks
#define SK_SCAN_AT(k) ((k) + ((k) >> 5))
  __shared__ int wave_total[SK_SCAN_NT / 64];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int carry = 0;
  for (int t0 = 0; t0 < n; t0 += SK_SCAN_TILE) {
    const int m = min(SK_SCAN_TILE, n - t0);
#pragma unroll 8
    for (int k = t; k < SK_SCAN_TILE; k += SK_SCAN_NT) {
      int cnt = 0;
      if (k < m) {
        const ulonglong2 w = reinterpret_cast<const ulonglong2 *>(mask)[t0 + k];
        cnt = pairs ? 2 * (__popcll(sk_pair_bits(w.x)) + __popcll(sk_pair_bits(w.y))) : __popcll(w.x) + __popcll(w.y);
        counts[t0 + k] = cnt;
        reinterpret_cast<ulonglong2 *>(mask_next)[t0 + k] = make_ulonglong2(0ull, 0ull);
      }
      tile[SK_SCAN_AT(k)] = cnt;
    }
    __syncthreads();
    int c = 0;
#pragma unroll
    for (int i = 0; i < SK_SCAN_TILE / SK_SCAN_NT; ++i) c += tile[SK_SCAN_AT(t * (SK_SCAN_TILE / SK_SCAN_NT) + i)];
    int incl = c;                                      // inclusive scan of the shares inside the wavefront
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(incl, d, 64);
      if (lane >= d) incl += up;
    }
    if (lane == 63) wave_total[wave] = incl;
    __syncthreads();
    int base = carry;
    for (int w = 0; w < wave; ++w) base += wave_total[w];
    int total = 0;
#pragma unroll
    for (int w = 0; w < SK_SCAN_NT / 64; ++w) total += wave_total[w];
    int w = base + incl - c;                           // exclusive prefix of this thread's share
#pragma unroll
    for (int i = 0; i < SK_SCAN_TILE / SK_SCAN_NT; ++i) { int &e_ = tile[SK_SCAN_AT(t * (SK_SCAN_TILE / SK_SCAN_NT) + i)]; const int v = e_; e_ = w; w += v; }
    __syncthreads();
#pragma unroll 8
    for (int k = t; k < m; k += SK_SCAN_NT) off[t0 + k] = tile[SK_SCAN_AT(k)];
    carry += total;
    __syncthreads();
  }
  if (t == 0) off[n] = carry;
}
#undef SK_SCAN_AT

// One wavefront per slice: its listed voices go to list[off[slice] ...] in ascending voice order.
__global__ __launch_bounds__(256) void sk_collect_expand_kernel(const int32_t *__restrict__ counts, const uint64_t *__restrict__ mask,
                                                                const int32_t *__restrict__ off, int n, int32_t *__restrict__ list, int pairs) {
  const int lane = threadIdx.x & 63, slice = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
  if (slice >= n || counts[slice] == 0) return;
  const uint64_t m0 = mask[2 * slice], m1 = mask[2 * slice + 1];
  const int o = off[slice];
  if (pairs) {                                         // lane L: the pair (2L, 2L+1) of the slice
    const uint64_t p0 = sk_pair_bits(m0), p1 = sk_pair_bits(m1);
    const uint64_t p = lane < 32 ? p0 : p1;
    const int sh = (2 * lane) & 63;
    if ((p >> sh) & 1) {
      const int rank = (lane < 32 ? 0 : __popcll(p0)) + __popcll(p & (((uint64_t)1 << sh) - 1));
      list[o + 2 * rank] = slice * 128 + 2 * lane;
      list[o + 2 * rank + 1] = slice * 128 + 2 * lane + 1;
    }
    return;
  }
  const uint64_t below = ((uint64_t)1 << lane) - 1;
  if ((m0 >> lane) & 1) list[o + __popcll(m0 & below)] = slice * 128 + lane;
  if ((m1 >> lane) & 1) list[o + __popcll(m0) + __popcll(m1 & below)] = slice * 128 + 64 + lane;
}

// The list from scratch (after uploads, clock changes, or a stretch on another kernel family): one wavefront per 128-voice
// slice, two voices per lane, sk_env_motion on the block's first frame.  A voice whose envelope rests but whose smoother
// still settles is listed only while its slice holds at most SK_FAST2_MAX_SETTLING of them -- a few such lanes would keep
// their whole wave of the steady kernel off the stalled-smoother blocks; many of them (a bank right after its upload) make
// that wave run its smoothers anyway, and listing them all would send the whole bank through the envelope kernel.
#ifndef SK_FAST2_MAX_SETTLING
#define SK_FAST2_MAX_SETTLING 16
#endif
__global__ __launch_bounds__(256) void sk_classify_kernel(const sk_render_args_t a, uint64_t *__restrict__ mask) {
  const int lane = threadIdx.x & 63, slice = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
  if (slice >= a.n_groups * 2) return;
  bool mv[2], st[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int v = slice * 128 + c * 64 + lane;
    const uint4 osc = *reinterpret_cast<const uint4 *>(&a.ro[SKP_OSC][v]);
    const uint4 tab = *reinterpret_cast<const uint4 *>(&a.ro[SKP_TAB][v]);
    const uint4 gn = *reinterpret_cast<const uint4 *>(&a.ro[SKP_GAIN][v]);
    const uint4 et = *reinterpret_cast<const uint4 *>(&a.ro[SKP_ENV_T][v]);
    const uint4 es = *reinterpret_cast<const uint4 *>(&a.ro[SKP_ENV_S][v]);
    const uint4 s0 = *reinterpret_cast<const uint4 *>(&a.rw[SKS_OSC][v]);
    const uint4 s1 = *reinterpret_cast<const uint4 *>(&a.rw[SKS_FILT][v]);
    const uint32_t flags = tab.z;
    const float amp = __uint_as_float(osc.w);
    const bool dead = (s1.w & SKR_FINISHED) || amp == 0.0f || (flags & SKF_INERT);
    float att = __uint_as_float(et.x), dec = __uint_as_float(et.y), sus = __uint_as_float(et.z), rel = __uint_as_float(et.w);
    float vel = __uint_as_float(gn.x);
    uint64_t t_start = ((uint64_t)es.y << 32) | es.x, t_release = ((uint64_t)es.w << 32) | es.z;
    bool active = (s1.w & SKR_ENV_ACTIVE) != 0;
    if (!(flags & SKF_USE_ENV)) {            // no envelope: a note held at level 1 with velocity 1 (fast2_load does the same)
      att = dec = rel = 0.0f; sus = 1.0f; vel = 1.0f; t_start = a.count0; t_release = 0; active = true;
    }
    const sk_motion_t mo = sk_env_motion(a.count0 + 1, dead, active, t_start, t_release, att, att + dec, rel, sus, amp, vel,
                                         __uint_as_float(gn.y), __uint_as_float(s0.y));
    mv[c] = mo.moving;
    st[c] = mo.settling;
  }
  if (__popcll(__ballot(st[0])) + __popcll(__ballot(st[1])) <= SK_FAST2_MAX_SETTLING) { mv[0] = mv[0] || st[0]; mv[1] = mv[1] || st[1]; }
  const uint64_t m0 = __ballot(mv[0]), m1 = __ballot(mv[1]);
  if (lane == 0) { mask[2 * slice] = m0; mask[2 * slice + 1] = m1; }
}

extern "C" int sk_launch_classify(const sk_render_args_t *args, uint64_t *mask, hipStream_t stream) {
  hipLaunchKernelGGL(sk_classify_kernel, dim3((unsigned)((args->n_groups * 2 + 3) / 4)), dim3(256), 0, stream, *args, mask);
  return (int)hipGetLastError();
}
#endif   // SK_FAST2_PLAIN_TU

// ---------------------------------------------------------------- launchers (C linkage)

// sk_launch_render_fast2: the steady kernel (args->skip_env2: alone; otherwise the host has put sk_launch_render_env2 on
// its second stream first -- skred_bank.c: render_block).  sk_launch_render_env2: collect + the envelope kernel.
#ifdef SK_PROBE_TU
#define SK_PROBE_FLAG true
#else
#define SK_PROBE_FLAG false
#endif
#if defined(SK_FAST2_GT_TU) && defined(SK_PROBE_TU)
#define SK_FAST2_LAUNCHER sk_launch_render_fast2gp
#elif defined(SK_FAST2_GT_TU)
#define SK_FAST2_LAUNCHER sk_launch_render_fast2g
#elif defined(SK_PROBE_TU)
#define SK_FAST2_LAUNCHER sk_launch_render_fast2p
#define SK_ENV2_LAUNCHER sk_launch_env_fast2p
extern "C" int sk_launch_render_fast2gp(const sk_render_args_t *args, int n_workgroups, size_t lds_bytes, hipStream_t stream);
#elif defined(SK_FAST2_FMP_TU) && SK_FAST2_FMP_TU == 2
#define SK_FAST2_LAUNCHER sk_launch_render_fm2ap
#define SK_ENV2_LAUNCHER sk_launch_env_fm2ap
#elif defined(SK_FAST2_FMP_TU)
#define SK_FAST2_LAUNCHER sk_launch_render_fm2
#define SK_ENV2_LAUNCHER sk_launch_env_fm2
#else
#define SK_FAST2_LAUNCHER sk_launch_render_fast2
#define SK_ENV2_LAUNCHER sk_launch_env_fast2
extern "C" int sk_launch_render_fm2(const sk_render_args_t *args, int n_workgroups, size_t lds_bytes, hipStream_t stream);
extern "C" int sk_launch_render_fm2ap(const sk_render_args_t *args, int n_workgroups, size_t lds_bytes, hipStream_t stream);
extern "C" int sk_launch_env_fm2(const sk_render_args_t *args, hipStream_t stream);
extern "C" int sk_launch_env_fm2ap(const sk_render_args_t *args, hipStream_t stream);
extern "C" int sk_launch_render_fast2g(const sk_render_args_t *args, int n_workgroups, size_t lds_bytes, hipStream_t stream);
extern "C" int sk_launch_render_fast2p(const sk_render_args_t *args, int n_workgroups, size_t lds_bytes, hipStream_t stream);
extern "C" int sk_launch_env_fast2p(const sk_render_args_t *args, hipStream_t stream);
#endif

static inline size_t sk_fast2_lds(const sk_render_args_t *args, int nw, bool gt = false) {
  // LDS: [tables] + wsum[2][NW][SK_CHUNK] + per wave the transposition tile and row sums (LDS-table banks) or the table
  // windows (global-table banks)
  const bool tab_lds = args->lds_table_floats > 0;
  const size_t tab_bytes = (size_t)(tab_lds ? args->lds_table_floats : 0) * sizeof(float);
  const size_t per_wave = tab_lds ? (size_t)(8 * SK_XT2) * sizeof(float) : (size_t)(2 * SK_WIN * 64) * sizeof(float);
  return tab_bytes + (size_t)nw * (2 * SK_CHUNK * sizeof(float2) + per_wave + (gt ? (size_t)SK_GT_LDS_FLOATS * sizeof(float) : 0));
}
static inline int sk_fast2_key(const sk_render_args_t *args) {
  return (args->lds_table_floats > 0 ? 8 : 0) | ((args->fast_mode & SKM_FILTER_ALL) ? 4 : 0) |
         ((args->fast_mode & SKM_ENV_ALL) ? 2 : 0) | (args->interp != 0 ? 1 : 0);
}
/* (two-operator FM banks: LDS-table banks only -- T is a constant there, which keeps the instantiations at 24 more; they live
   in the other translation units) */
/* (I = 1: linear; with every live voice on a guarded whole-table loop -- args->interp == 2, SKF_GUARD -- the instantiation
   without the fold test, INTERP == 2: plain translation unit only, two-operator FM banks keep the general form) */
#if defined(SK_FAST2_GT_TU)
/* (the in-place instantiations: LDS-table banks with envelopes) */
#define SK_FAST2_CASE(K, T, F, E, I)                                                                    \
  case K:                                                                                               \
    if (T && E) {                                                                                       \
      if (I && args->interp == 2) { if (mixed) SK_FAST2_LAUNCH(true, F, true, (I ? 2 : 0), true, 0) else SK_FAST2_LAUNCH(true, F, true, (I ? 2 : 0), false, 0) } \
      else if (mixed) SK_FAST2_LAUNCH(true, F, true, I, true, 0)                                        \
      else SK_FAST2_LAUNCH(true, F, true, I, false, 0)                                                  \
    } else return (int)hipErrorInvalidValue;                                                            \
    break;
#elif defined(SK_FAST2_FMP_TU)
#define SK_FAST2_CASE(K, T, F, E, I)                                                                    \
  case K:                                                                                               \
    if (mixed) SK_FAST2_LAUNCH(true, F, E, I, true, SK_FAST2_FMP_TU) else SK_FAST2_LAUNCH(true, F, E, I, false, SK_FAST2_FMP_TU) \
    break;
#else
#define SK_FAST2_CASE(K, T, F, E, I)                                                                    \
  case K:                                                                                               \
    if (I && args->interp == 2) { if (mixed) SK_FAST2_LAUNCH(T, F, E, (I ? 2 : 0), true, 0) else SK_FAST2_LAUNCH(T, F, E, (I ? 2 : 0), false, 0) } \
    else if (mixed) SK_FAST2_LAUNCH(T, F, E, I, true, 0)                                                \
    else SK_FAST2_LAUNCH(T, F, E, I, false, 0)                                                          \
    break;
#endif
#define SK_FAST2_SWITCH()                                                                               \
  switch (key) {                                                                                        \
    SK_FAST2_CASE(0, false, false, false, 0) SK_FAST2_CASE(1, false, false, false, 1)                   \
    SK_FAST2_CASE(2, false, false, true, 0)  SK_FAST2_CASE(3, false, false, true, 1)                    \
    SK_FAST2_CASE(4, false, true, false, 0)  SK_FAST2_CASE(5, false, true, false, 1)                    \
    SK_FAST2_CASE(6, false, true, true, 0)   SK_FAST2_CASE(7, false, true, true, 1)                     \
    SK_FAST2_CASE(8, true, false, false, 0)  SK_FAST2_CASE(9, true, false, false, 1)                    \
    SK_FAST2_CASE(10, true, false, true, 0)  SK_FAST2_CASE(11, true, false, true, 1)                    \
    SK_FAST2_CASE(12, true, true, false, 0)  SK_FAST2_CASE(13, true, true, false, 1)                    \
    SK_FAST2_CASE(14, true, true, true, 0)   SK_FAST2_CASE(15, true, true, true, 1)                     \
  }

extern "C" int SK_FAST2_LAUNCHER(const sk_render_args_t *args, int n_workgroups, size_t lds_bytes,
                                 hipStream_t stream) {
  const bool tab_lds = args->lds_table_floats > 0;
#ifdef SK_FAST2_PLAIN_TU
  if (args->probe_out) return sk_launch_render_fast2p(args, n_workgroups, lds_bytes, stream);   // (the host has ruled FM pairs out)
  if ((args->fast_mode & SKM_FM_PAIR) && tab_lds)
    return (args->fast_mode & SKM_PAIR_AP) ? sk_launch_render_fm2ap(args, n_workgroups, lds_bytes, stream)
                                           : sk_launch_render_fm2(args, n_workgroups, lds_bytes, stream);
  if (args->env_gain) return sk_launch_render_fast2g(args, n_workgroups, lds_bytes, stream);
#endif
#if defined(SK_PROBE_TU) && !defined(SK_FAST2_GT_TU)
  if (args->env_gain) return sk_launch_render_fast2gp(args, n_workgroups, lds_bytes, stream);
#endif
  (void)lds_bytes;
  const int nw = tab_lds ? Fast2Shape<true>::NW : Fast2Shape<false>::NW;
#ifdef SK_FAST2_GT_TU
  const size_t lds_fast2 = sk_fast2_lds(args, nw, true);
#else
  const size_t lds_fast2 = sk_fast2_lds(args, nw);
#endif
  dim3 grid((unsigned)(n_workgroups + args->wg_shift)), block((unsigned)nw * 64);
  const bool mixed = (args->fast_mode & SKM_MIXED) != 0;     // filter / envelope on some voices only: per-lane flags
  const int key = sk_fast2_key(args);
#ifdef SK_FAST2_GT_TU
#define SK_FAST2_LAUNCH(T, F, E, I, M, P) { hipLaunchKernelGGL((sk_render_fast2_kernel<T, F, E, I, M, P, true, SK_PROBE_FLAG>), grid, block, lds_fast2, stream, *args); }
#else
#define SK_FAST2_LAUNCH(T, F, E, I, M, P) { hipLaunchKernelGGL((sk_render_fast2_kernel<T, F, E, I, M, P, false, SK_PROBE_FLAG>), grid, block, lds_fast2, stream, *args); }
#endif
  SK_FAST2_SWITCH()
#undef SK_FAST2_LAUNCH
  return (int)hipGetLastError();
}

// the envelope kernel's workgroups the device holds at once: 3 per CU by registers (SK_ENV2_MIN_WAVES), fewer by LDS
extern "C" int sk_env2_grid(const sk_render_args_t *args);
#ifdef SK_FAST2_PLAIN_TU
extern "C" int sk_env2_grid(const sk_render_args_t *args) {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
  const size_t lds_env2 = sk_fast2_lds(args, 4);
  int per_cu = (int)((size_t)160 * 1024 / (lds_env2 ? lds_env2 : 1));
  if (per_cu > SK_ENV2_MIN_WAVES) per_cu = SK_ENV2_MIN_WAVES;
  if (per_cu < 1) per_cu = 1;
  int n = (cus > 0 ? cus : 64) * per_cu;
  const int most = (args->n_groups * SK_GROUP + 511) / 512;       // passes when every voice is listed
  if (n > most) n = most;
  return n < 1 ? 1 : n;
}
#endif

#ifdef SK_FAST2_PLAIN_TU
// collect the list (this block's mask -> counts, offsets, voices; the next block's mask zeroed): on the block's own stream,
// ahead of both render kernels (an idle machine runs it in ~18 us; beside a full one it took 130)
extern "C" int sk_launch_collect(const sk_render_args_t *args, hipStream_t stream) {
  const bool fmp = (args->fast_mode & SKM_FM_PAIR) != 0 && args->lds_table_floats > 0;
  hipLaunchKernelGGL(sk_collect_scan_kernel, dim3(1), dim3(SK_SCAN_NT), 0, stream, args->mask_cur, args->mask_next, args->n_groups * 2, fmp ? 1 : 0,
                     args->group_flag, args->env_off);
  hipLaunchKernelGGL(sk_collect_expand_kernel, dim3((unsigned)((args->n_groups * 2 + 3) / 4)), dim3(256), 0, stream,
                     args->group_flag, args->mask_cur, args->env_off, args->n_groups * 2, args->env_list, fmp ? 1 : 0);
  return (int)hipGetLastError();
}
#endif

#ifndef SK_FAST2_GT_TU
// render the list: on the SECOND stream of the block, beside the steady kernel; args->n_env_rows workgroups (sk_env2_grid)
extern "C" int SK_ENV2_LAUNCHER(const sk_render_args_t *args, hipStream_t stream) {
  const bool tab_lds = args->lds_table_floats > 0;
#ifdef SK_FAST2_PLAIN_TU
  if (args->probe_out) return sk_launch_env_fast2p(args, stream);
  if ((args->fast_mode & SKM_FM_PAIR) != 0 && tab_lds)
    return (args->fast_mode & SKM_PAIR_AP) ? sk_launch_env_fm2ap(args, stream) : sk_launch_env_fm2(args, stream);
#endif
  const size_t lds_env2 = sk_fast2_lds(args, 4);
  dim3 grid((unsigned)args->n_env_rows), block_env(SK_GROUP);
  const bool mixed = (args->fast_mode & SKM_MIXED) != 0;
  const int key = sk_fast2_key(args) | 2;                    // (the envelope kernel exists for enveloped banks only)
  (void)tab_lds;
#define SK_FAST2_LAUNCH(T, F, E, I, M, P) { if (E) hipLaunchKernelGGL((sk_render_env2_kernel<T, F, I, M, P, SK_PROBE_FLAG>), grid, block_env, lds_env2, stream, *args); }
  SK_FAST2_SWITCH()
#undef SK_FAST2_LAUNCH
  return (int)hipGetLastError();
}
#endif   // !SK_FAST2_GT_TU
#undef SK_FAST2_CASE
#undef SK_FAST2_SWITCH
