// skred_fast_common.hpp -- the per-voice, per-frame arithmetic of the one-voice-per-lane family (gfx950 / CDNA4).
//
// Included by skred_render_fast.hip (sk_render_fast_kernel: the whole frame in one wave) and skred_render_split.hip
// (sk_render_split_kernel: the same frame split between an oscillator wave and a post wave).  Both call the SAME functions --
// same products, same sums, same order -- so a voice renders to the same bits whichever kernel the host picks.
#pragma once
#include "skred_kernel_common.hpp"

struct FastRegs {
  // launch constants
  float inc, lo, hi, span, span2, amp;
  int toff4, tsize_m1;          // byte offset of the table in the pool, table_size - 1
  float att, attdec, dec, sus, one_m_sus, rel, vel;
  float k, b0, b1, b2, a1, a2, pan_l, pan_r;
  float gain_sustain;           // amp * (sustain_level * velocity)
  // recurrences
  float phase, sgain, x1, x2, y1, y2, sample;
  float tf, trf;                // frames since note-on / note-off for the CURRENT frame
  uint32_t rw;
  // STOPS kernels: forward one-shots without loop play to the table end and finish (synth.c:242-244)
  bool stop;                    // this voice stops instead of wrapping
  bool fin;                     // ... and did so on the frame just advanced
  float hi_stop;                // loop_end - 1e-6f, the phase it is left at
  // ... and frequency modulation by a higher-indexed voice of the same 64-voice group (synth.c:548-555: the
  // carrier reads the modulator's voice_sample of the PREVIOUS frame, so no ordering inside a frame is needed)
  int fm_addr;                  // modulator lane * 4 (ds_bpermute address), -1: none
  float fm_k, fm_depth;         // voice_phase_inc[m] * voice_freq_scale[n];  voice_freq_mod_depth[n]
  // amplitude and pan modulation by a higher-indexed voice of the group (previous frame's voice_sample) or by the
  // voice itself (its own sample of THIS frame: synth.c:584-587 post-filter, synth.c:597-602 post-gain)
  int am_addr, pm_addr;         // lane * 4; -1: none; -2: the voice itself
  float am_depth, pm_depth;
  float am_prev, pm_prev;       // this frame's modulator samples, fetched before the voice's own sample changes
  bool pan_dirty;               // pan modulation rewrote voice_pan_left / _right: the MISC plane is stored back
  bool rev;                     // voice_direction: the (modulated) increment is negated (synth.c:224)
  // sample & hold (synth.c:560-571), bit-crush (synth.c:574), smoother off (synth.c:589)
  int hold_max, hold_count, quant;
  float hold;
  bool nosmooth;
  // the noise source (w6, synth.c:543-546): the voice takes the frame's shared LCG draw instead of running its
  // oscillator; its phase is never touched (ophase is what goes back)
  bool noise;
  float ophase;
  // banks in which only some voices run the biquad / the envelope
  bool filt, use_env;
  float ox1, ox2, oy1, oy2;     // delay line of an UNfiltered voice as loaded: it is stored back untouched
#ifdef SK_PROBE_TU
  float2 *probe;                // this frame's probe row of the voice (nullptr: not probed / skipped / muted); skred_bank_set_probe
  float2 *probe_hold;           // (skewed blocks) the row pointer of a lane that has rendered its last block already, while it runs on
  int probe_stride;             // float2 per frame
  bool probe_any;               // (wave-uniform) some lane of the wave writes probes
#endif
};

// NOCLAMP: the caller guarantees 0 <= lo <= pos < hi <= table_size (TAME loops), so the reference's
// index clamps (synth.c:271-272) can never act and are dropped.
template <bool TAB_LDS, int INTERP, bool NOCLAMP>
__device__ __forceinline__ float fast_fetch(const char *lds_tab, const char *__restrict__ glb_tab,
                                            const FastRegs &r, float pos) {
  int idx = (int)pos;
  if (!NOCLAMP) idx = max(min(idx, r.tsize_m1), 0);          // clamp, synth.c:271-272
  const char *tab = TAB_LDS ? lds_tab : glb_tab;
  if (INTERP == 0) return *reinterpret_cast<const float *>(tab + (r.toff4 + (idx << 2)));
  if (INTERP == 2 && NOCLAMP) {
    // linear, and EVERY live voice of the bank loops over its whole table with a guard sample behind it (SKF_GUARD; the host
    // picks this instantiation): the second tap is always the next float -- no fold test, no third gather; the fraction is
    // v_fract_f32 (pos - floor(pos): the same exact difference as pos - (float)(int)pos for pos >= 0).  Same products, same
    // sums as the form below and as oracle/cpu_ref.c: table_fetch.
    const tap_pair_t pg = TAB_LDS ? *reinterpret_cast<const tap_pair_t *>(tab + (r.toff4 + (idx << 2)))
                                  : load_tap_pair_global(tab + (r.toff4 + (idx << 2)));
    return pg.a + __builtin_amdgcn_fractf(pos) * (pg.b - pg.a);
  }
  // linear: oracle/cpu_ref.c:table_fetch.  Every voice of a fast bank wraps (no stopping one-shots).
  // Both taps come from ONE 8-byte gather (4-byte aligned pair) -- the neighbour is idx+1 except on
  // the last sample before the loop end, where a second (rare) gather fetches the loop start.
  const tap_pair_t pr = TAB_LDS ? *reinterpret_cast<const tap_pair_t *>(tab + (r.toff4 + (idx << 2)))
                                : load_tap_pair_global(tab + (r.toff4 + (idx << 2)));
  const float a = pr.a;
  float b = pr.b;
  int nxt = idx + 1;
  bool special = !r.stop && (float)nxt >= r.hi;      // a stopping voice does not fold: its neighbour clamps below
  if (special) nxt = (int)r.lo;
  if (!NOCLAMP) { const int c = max(min(nxt, r.tsize_m1), 0); special = special || (c != nxt); nxt = c; }
  if (special) b = *reinterpret_cast<const float *>(tab + (r.toff4 + (nxt << 2)));
  const float frac = pos - (float)idx;
  return a + frac * (b - a);
}


// One voice, one frame.  STEADY: every lane of the wave sits in its sustain stage.  When !STEADY the
// caller has set r.tf / r.trf to this frame's envelope clocks.
// TAME: the caller has proved for every lane that lo <= phase <= hi and 0 <= inc <= span/2, which
// by induction keeps phase+inc in [lo, hi + span/2]: the only wrap that can occur is the simple one.
// Oscillator half of a frame (osc_next, synth.c:217-275): advance and wrap the phase.
// LOZ (with TAME, without STOPS): the caller has also proved lo == 0 for every lane of the wave (the plain LUT case: no loop
// window), so hi == span and the wrapped phase lo + ((ph0 - lo) - span) is ph0 - span, whose SIGN is the reference's test
// `phase >= loop_end` (exact for span <= ph0 < 2 span, strictly negative below span; ph0 >= 0).  As unsigned integers a negative
// float is larger than every non-negative one and non-negative floats keep their order: min_u32 picks ph0 - span when it is
// >= 0 and ph0 otherwise -- the select of synth.c:241-247 in two instructions (sub, min) where the general form takes five.
template <bool TAME, bool STOPS, bool LOZ>
__device__ __forceinline__ float fast_advance(FastRegs &r, float inc);
template <bool TAME, bool STOPS = false, bool LOZ = false>
__device__ __forceinline__ float fast_advance(FastRegs &r) { return fast_advance<TAME, STOPS, LOZ>(r, r.inc); }
template <bool TAME, bool STOPS>
__device__ __forceinline__ float fast_advance(FastRegs &r, float inc) { return fast_advance<TAME, STOPS, false>(r, inc); }

template <bool TAME, bool STOPS, bool LOZ>
__device__ __forceinline__ float fast_advance(FastRegs &r, float inc) {
  if (TAME && LOZ && !STOPS) {
    const float p0 = r.phase + inc;
    const float d = p0 - r.span;
    const float p = __uint_as_float(min(__float_as_uint(p0), __float_as_uint(d)));
    r.phase = p;
    return p;
  }
  const float ph0 = r.phase + inc;
  const float x = ph0 - r.lo;
  const bool over = ph0 >= r.hi;
  float ph;
  if (TAME) {
    ph = over ? r.lo + (x - r.span) : ph0;
  } else {
    const bool simple = over && (x < r.span2);          // one loop length past the end: x - span exact
    const bool in_range = (ph0 >= r.lo) && !over;
    ph = simple ? r.lo + (x - r.span) : ph0;
    if (!(in_range || simple)) ph = slow_wrap(ph0, r.lo, r.hi, r.span);
  }
  if (STOPS) {                                          // synth.c:242-244,248-250: clamp and finish instead of wrapping
    const bool under = ph0 < r.lo;
    r.fin = r.stop && (over || under);
    if (r.stop) ph = over ? r.hi_stop : (under ? r.lo : ph0);
  }
  r.phase = ph;
  return ph;
}

// (extended instantiation) which per-lane features occur in the wave at all: wave-uniform, decided once per pass, so
// that a wave only pays for the exchanges / tests of the features it holds
#define XF_FM 1        /* a frequency-modulated lane: the modulator's previous sample comes through ds_bpermute */
#define XF_AP 2        /* amplitude or pan modulation */
#define XF_REV 4       /* reverse playback */
#define XF_HOLDQ 8     /* sample & hold, bit-crush */
#define XF_NOSMOOTH 16 /* smoother off */
#define XF_NOISE 32    /* the noise source */
#define XF_STOP 64     /* a one-shot that can finish */
#define XF_ALL 127

// The rest of the frame: biquad, envelope / gain, smoother, pan.
// STALL (steady waves only): the smoother no longer moves in any lane (fast_smoother_stalled) and is skipped.
// PLAIN (EXT only): the caller has proved that no lane of the wave is modulated or runs without the smoother
// (`!any_fm`): only the per-lane filter / envelope flags of an extended bank remain.
template <bool FILTER, bool ENV, bool STEADY, bool STALL = false, bool EXT = false, bool PLAIN = false>
__device__ __forceinline__ void fast_post(FastRegs &r, float s, float &xn, float &xo, float &yn, float &yo,
                                          const bool released, float &out_l, float &out_r, const bool muted = false,
                                          const int xf = XF_ALL) {
  constexpr bool XMOD = EXT && !PLAIN;
  // ---- biquad (mmf_process, synth.c:349-364) ----
  if (FILTER) {
    // xn/yn: newest delay-line entries, xo/yo: the older ones.  The new values overwrite the OLD
    // slots, so the caller alternates the argument order frame by frame instead of shifting
    // registers (x2 = x1; x1 = s costs four v_mov per frame).
    float y = r.b0 * s;
    y = y + r.b1 * xn;
    y = y + r.b2 * xo;
    y = y - r.a1 * yn;
    y = y - r.a2 * yo;
    if (!EXT || r.filt) {                           // (EXT: a voice with filter_mode 0 passes through, synth.c:577)
      xo = s;
      yo = y;
      s = y;
    }
  }
  // ---- envelope (amp_envelope_step, synth.c:398-431) and gain (synth.c:580-588) ----
  float gain;
  if (!ENV || (EXT && !r.use_env)) {
    gain = r.amp;                                   // amp * 1.0f * 1.0f
  } else if (STEADY) {
    gain = r.gain_sustain;                          // e = sustain_level on every lane
  } else {
    float e = 0.0f;
    if (r.rw & SKR_ENV_ACTIVE) {
      if (r.tf < r.att) {
        e = r.tf / r.att;
      } else if (r.tf < r.attdec) {
        const float prog = (r.tf - r.att) / r.dec;
        e = 1.0f - prog * r.one_m_sus;
      } else if (!released) {
        e = r.sus;
      } else if (r.trf < r.rel) {
        const float prog = r.trf / r.rel;
        e = r.sus * (1.0f - prog);
      } else {
        r.rw &= ~SKR_ENV_ACTIVE;
      }
    }
    gain = r.amp * (e * r.vel);
  }
  // ---- smoother + apply (synth.c:589-593), pan (synth.c:603-604) ----
  if (XMOD && (xf & XF_AP)) {                            // final = amp * env * mod (synth.c:583-588)
    // (selects, not a branch per lane: hipcc turned `if (am_addr != -1) gain *= ...` into two nested EXEC-masked branches per frame)
    const float am_ = r.am_addr == -2 ? s : r.am_prev;
    const float g_am_ = gain * (am_ * r.am_depth);
    gain = r.am_addr != -1 ? g_am_ : gain;
  }
  if (XMOD && (xf & XF_NOSMOOTH) && r.nosmooth) {
    s *= gain;                                           // voice_smoother_gain is left alone (synth.c:589-593)
  } else {
    if (!STALL) r.sgain += r.k * (gain - r.sgain);
    s *= r.sgain;
  }
  r.sample = s;
  if (XMOD && (xf & XF_AP) && r.pm_addr != -1 && !muted) {               // synth.c:597-602 (inside the `not disconnected` branch)
    const float q = (r.pm_addr == -2 ? s : r.pm_prev) * r.pm_depth;
    r.pan_l = (1.0f - q) / 2.0f;
    r.pan_r = (1.0f + q) / 2.0f;
    r.pan_dirty = true;
  }
  out_l = s * r.pan_l;
  out_r = s * r.pan_r;
#ifdef SK_PROBE_TU
  if (r.probe_any && r.probe) { *r.probe = make_float2(out_l, out_r); r.probe += r.probe_stride; }   // synth.c:603-608
#endif
}

// The same rest-of-frame for the steady block paths, with the delay line and its coefficients held as register PAIRS so
// that the two feed-forward and the two feedback products are one v_pk_mul_f32 each and the pan is a third: 7 + 1 + 1
// instructions for biquad, gain and pan, no register moves (left to itself hipcc also SLP-packs these products, but
// pays 2-3 v_mov per frame to line the operands up).  Same products, same order of the four additions, hence the same
// bits as fast_post.  NEWEST_X: the newest delay-line entries sit in .x (frames alternate, as in fast_post's role swap).
typedef float v2f __attribute__((ext_vector_type(2)));
struct FastPk {
  v2f b12, b21, a12, a21;       // (b1,b2) (b2,b1) (a1,a2) (a2,a1)
  v2f pan;                      // (pan_left, pan_right)
  // fast_pan_fold2p: the pan gains of the lane's own voice and of its partner 32 lanes away, arranged so that ONE lane-half swap
  // of two frames' samples feeds both channels (lanes 0..31: A = own, B = the voice 32 lanes up; lanes 32..63: A = the voice 32
  // lanes down, B = own)
  float plA, plB, prA, prB;
};

// sample & hold and bit-crush of a lane (synth.c:560-574), for the block paths of extended banks
__device__ __forceinline__ float fast_holdq(FastRegs &r, float s) {
  if (r.hold_max) {
    if (r.hold_count == 0) r.hold = s;
    s = r.hold;
    if (++r.hold_count >= r.hold_max) r.hold_count = 0;
  }
  if (r.quant) s = crush(s, r.quant);
  return s;
}

// Envelopes in motion on the block paths (round 2; the two-per-lane envelope kernel's form, skred_render_fast2.hip:
// fast2_env_span2, one voice per lane).  Over a span of frames in which a lane keeps its stage or changes it ONCE, to the
// stage that follows, the level is e = C * (A + B * q), q = (clk - base) / den, bit-identical to the reference's stage
// expressions (synth.c:405,413,425): attack q = 1*(0 + 1*q), decay 1 - q*(1-sus) = 1*(1 + (-(1-sus))*q), release
// sus*(1 - q) = sus*(1 + (-1)*q), a held or silent stage level*(1 + 0*q).  Two constant sets per lane: the stage on the
// span's first frame and the one after the change; `bnd` is the value of the first set's clock from which the second
// applies -- the comparison the reference makes on that frame (`t < attack_time`, `t < decay_start + decay_time`,
// `t_release < release_time`); +inf when the lane keeps its stage.  q is the correctly rounded quotient: the FMA tail of
// the IEEE fp32 division expansion with a reciprocal refined once per span (denominators in [2^-40, 2^40], where
// div_scale / div_fixup would not intervene).
struct FastEnv {
  float clk, base, den, rinv, A, B, C;
  float clk2, base2, den2, rinv2, A2, B2, C2, bnd;
};

__device__ __forceinline__ int fast_stage_code(bool active, bool released, float t, float tr, float att, float attdec, float rel) {
  if (!active) return 0;
  if (t < att) return 1;
  if (t < attdec) return 2;
  if (!released) return 3;
  return (tr < rel) ? 4 : 5;
}

// (t1,tr1) / (tN,trN): clocks of the span's first / last frame; (t0,tr0): of the frame before it.  `ok` stays true while
// the lane can be rendered in this form; `runs_out`: its release ends in the span (the caller clears is_active, synth.c:429).
__device__ __forceinline__ void fast_env_span2(const FastRegs &r, FastEnv &e, bool dead, bool released, float t1, float tr1,
                                               float tN, float trN, float t0, float tr0, bool &ok, bool &runs_out) {
  const bool act = (r.rw & SKR_ENV_ACTIVE) != 0 && r.use_env;   // (a voice without envelope takes `amp`: fast_env_gain; its sets are inert)
  const int code0 = fast_stage_code(act, released, t1, tr1, r.att, r.attdec, r.rel);
  const int code1 = fast_stage_code(act, released, tN, trN, r.att, r.attdec, r.rel);
  const bool step = code0 != code1;
  const bool next_stage = (code0 == 1 && code1 == 2) || (code0 == 2 && (code1 == 3 || code1 == 4)) || (code0 == 4 && code1 == 5);
  float den[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int code = h ? code1 : code0;
    const float level = code == 3 ? r.sus : 0.0f;
    const float A = code == 1 ? 0.0f : 1.0f;
    const float B = code == 1 ? 1.0f : (code == 2 ? -r.one_m_sus : (code == 4 ? -1.0f : 0.0f));
    const float C = code == 4 ? r.sus : ((code == 1 || code == 2) ? 1.0f : level);
    const float clk = code == 4 ? tr0 : t0;
    const float base = code == 2 ? r.att : 0.0f;
    const float d = code == 1 ? r.att : (code == 2 ? r.dec : (code == 4 ? r.rel : 1.0f));
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float ri = __builtin_fmaf(__builtin_fmaf(-d, r0, 1.0f), r0, r0);
    den[h] = d;
    if (h == 0) { e.A = A; e.B = B; e.C = C; e.clk = clk; e.base = base; e.den = d; e.rinv = ri; }
    else        { e.A2 = A; e.B2 = B; e.C2 = C; e.clk2 = clk; e.base2 = base; e.den2 = d; e.rinv2 = ri; }
  }
  e.bnd = !step ? __builtin_huge_valf() : (code0 == 1 ? r.att : (code0 == 2 ? r.attdec : r.rel));
  ok = ok && (dead || ((!step || next_stage) && den[0] >= 0x1p-40f && den[0] <= 0x1p40f && den[1] >= 0x1p-40f && den[1] <= 0x1p40f));
  runs_out = !dead && r.use_env && (code0 == 5 || code1 == 5);
}

// this frame's gain amp * (e * velocity) (synth.c:582,588) from the two constant sets
__device__ __forceinline__ float fast_env_gain(const FastRegs &r, FastEnv &e) {
  e.clk += 1.0f;                                       // exact: the clocks stay below 2^24 in this form
  e.clk2 += 1.0f;
  const bool after = e.clk >= e.bnd;                   // the reference's `t < limit` failed on this frame
  const float num = after ? e.clk2 - e.base2 : e.clk - e.base;
  const float den = after ? e.den2 : e.den, rinv = after ? e.rinv2 : e.rinv;
  const float A = after ? e.A2 : e.A, B = after ? e.B2 : e.B, C = after ? e.C2 : e.C;
  float q = num * rinv;
  float rem = __builtin_fmaf(-den, q, num);
  q = __builtin_fmaf(rem, rinv, q);
  rem = __builtin_fmaf(-den, q, num);
  q = __builtin_fmaf(rem, rinv, q);
  const float lvl = C * (A + B * q);
  return r.use_env ? r.amp * (lvl * r.vel) : r.amp;    // (extended banks: a voice without envelope, synth.c:580-582)
}

template <bool FILTER, bool ENV, bool STALL, bool EXT, bool NEWEST_X, bool PAN = true, bool RAMP = false>
__device__ __forceinline__ void fast_post_v(FastRegs &r, const FastPk &k, float s, v2f &xx, v2f &yy, float &out_l, float &out_r,
                                            const int xf = 0, FastEnv *ev = nullptr) {
  if (EXT && (xf & XF_HOLDQ)) s = fast_holdq(r, s);           // (wave-uniform: some lane of the wave holds or crushes)
  if (FILTER) {
    const v2f t = (NEWEST_X ? k.b12 : k.b21) * xx;
    const v2f u = (NEWEST_X ? k.a12 : k.a21) * yy;
    float y = r.b0 * s;
    y = y + (NEWEST_X ? t.x : t.y);                 // + b1 * newest x
    y = y + (NEWEST_X ? t.y : t.x);                 // + b2 * older x
    y = y - (NEWEST_X ? u.x : u.y);                 // - a1 * newest y
    y = y - (NEWEST_X ? u.y : u.x);                 // - a2 * older y
    if (!EXT || r.filt) {                           // the new entries overwrite the OLDER slots
      if (NEWEST_X) { xx.y = s; yy.y = y; } else { xx.x = s; yy.x = y; }
      s = y;
    }
  }
  const float gain = RAMP ? fast_env_gain(r, *ev) : (ENV ? r.gain_sustain : r.amp);   // (an un-enveloped voice of a mixed bank carries amp in gain_sustain)
  if (EXT && (xf & XF_NOSMOOTH)) {                     // some lane runs without the smoother: its gain applies directly,
    if (!STALL) { const float nx = r.sgain + r.k * (gain - r.sgain); r.sgain = r.nosmooth ? r.sgain : nx; }   // voice_smoother_gain rests
    s *= r.nosmooth ? gain : r.sgain;
  } else {
    if (!STALL) r.sgain += r.k * (gain - r.sgain);
    s *= r.sgain;
  }
  r.sample = s;
#ifdef SK_PROBE_TU
  if (r.probe_any && r.probe) { *r.probe = make_float2(s * k.pan.x, s * k.pan.y); r.probe += r.probe_stride; }   // synth.c:603-608: the same products the mix takes
#endif
  if (!PAN) { out_l = s; return; }                     // the caller pans and folds two frames at once (fast_pan_fold2)
  const v2f lr = k.pan * (v2f){s, s};
  out_l = lr.x;
  out_r = lr.y;
}

// The two halves of fast_post_v's biquad, for a kernel that evaluates them in different wavefronts (skred_render_split.hip):
// the feed-forward partial sum (b0*s + b1*x1) + b2*x2 needs input samples only, the feedback half finishes
// y = (P - a1*y1) - a2*y2 -- the same products and the same left-to-right sums as fast_post_v (mmf_process, synth.c:349-364),
// hence the same bits.  NEWEST_X as there: the newest delay-line entry sits in .x, the new one overwrites the older slot.
template <bool NEWEST_X>
__device__ __forceinline__ float fast_biquad_ff(float b0, const v2f &b12, const v2f &b21, float s, v2f &xx) {
  const v2f t = (NEWEST_X ? b12 : b21) * xx;
  float y = b0 * s;
  y = y + (NEWEST_X ? t.x : t.y);                 // + b1 * newest x
  y = y + (NEWEST_X ? t.y : t.x);                 // + b2 * older x
  if (NEWEST_X) xx.y = s; else xx.x = s;
  return y;
}
template <bool NEWEST_X>
__device__ __forceinline__ float fast_biquad_fb(const v2f &a12, const v2f &a21, float p, v2f &yy) {
  const v2f u = (NEWEST_X ? a12 : a21) * yy;
  float y = p;
  y = y - (NEWEST_X ? u.x : u.y);                 // - a1 * newest y
  y = y - (NEWEST_X ? u.y : u.x);                 // - a2 * older y
  if (NEWEST_X) yy.y = y; else yy.x = y;
  return y;
}
// ... and what follows the biquad in fast_post_v for a clean bank at a constant envelope level: gain, smoother, voice_sample
template <bool ENV, bool STALL>
__device__ __forceinline__ float fast_gain_const(FastRegs &r, float s) {
  const float gain = ENV ? r.gain_sustain : r.amp;
  if (!STALL) r.sgain += r.k * (gain - r.sgain);
  s *= r.sgain;
  r.sample = s;
  return s;
}

// Pan and L/R fold of two frames in one go: four plain products (a v_pk_mul_f32 of a splat makes hipcc treat the
// unused upper register of the pair as a source -- if an LDS gather is in flight into it the wave stalls on it -- and
// costs a v_mov + s_nop 1 per frame to get its halves into the swap), the two swaps spaced by hand (>= 2 wait states
// behind the product they read; hipcc pads nothing inside an asm statement), the two adds.  Same products as
// `s * pan_left`, `s * pan_right`, same sum as fold_lr.
__device__ __forceinline__ void fast_pan_fold2(float s0, float s1, float pan_l, float pan_r, float &f0, float &f1) {
  float a0, b0, a1, b1;
  asm("v_mul_f32_e32 %0, %4, %6\n\t"
      "v_mul_f32_e32 %1, %4, %7\n\t"
      "v_mul_f32_e32 %2, %5, %6\n\t"
      "v_mul_f32_e32 %3, %5, %7\n\t"
      "v_permlane32_swap_b32_e32 %0, %1\n\t"
      "s_nop 0\n\t"
      "v_permlane32_swap_b32_e32 %2, %3\n\t"
      "v_add_f32_e32 %0, %0, %1\n\t"
      "v_add_f32_e32 %2, %2, %3"
      : "=&v"(a0), "=&v"(b0), "=&v"(a1), "=&v"(b1)
      : "v"(s0), "v"(s1), "v"(pan_l), "v"(pan_r));
  f0 = a0;
  f1 = a1;
}

// the partner's pan gains for fast_pan_fold2p (once per chunk; the pan of a clean bank does not move inside a launch)
__device__ __forceinline__ void fast_pk_partner(FastPk &k, float pan_l, float pan_r) {
  const auto pl = __builtin_amdgcn_permlane32_swap(__float_as_uint(pan_l), __float_as_uint(pan_l), false, false);
  const auto pr = __builtin_amdgcn_permlane32_swap(__float_as_uint(pan_r), __float_as_uint(pan_r), false, false);
  k.plA = __uint_as_float(pl[0]); k.plB = __uint_as_float(pl[1]);   // (pl[0]: pan_left of lanes 0..31 in both halves; pl[1]: of lanes 32..63)
  k.prA = __uint_as_float(pr[0]); k.prB = __uint_as_float(pr[1]);
}

// The same pan + fold for two frames with ONE swap instead of two: v_permlane32_swap(s0, s1) leaves (s0 of lanes 0..31 | s1 of
// lanes 0..31) in one register and (s0 of lanes 32..63 | s1 of lanes 32..63) in the other, so with the partner's pan gains at
// hand (FastPk::plA ...) every lane forms a pair sum directly: lanes 0..31 the pair sums of FRAME 0 (u: left, v: right), lanes
// 32..63 those of FRAME 1 -- the products s * pan_left, s * pan_right of both lanes of a pair and their sum, exactly what
// fast_pan_fold2 adds, one v_permlane32_swap (8.8 cycles of a lone wave, tools/issue_mix.hip) and a wait state fewer per frame
// pair.  The tile rows then hold (L of frame q | L of frame q + 1) and (R of frame q | R of frame q + 1): the reduction's
// last step stores accordingly (SK_PAIRED_ in skred_render_fast.hip).
__device__ __forceinline__ void fast_pan_fold2p(float s0, float s1, float plA, float plB, float prA, float prB, float &u, float &v) {
  // (the builtin, not an asm island: hipcc spaces the swap behind the products it reads by itself and fills the wait states
  // with the other strand's instructions -- a lone wave has nothing else to fill them with)
  const auto p = __builtin_amdgcn_permlane32_swap(__float_as_uint(s0), __float_as_uint(s1), false, false);
  const float a = __uint_as_float(p[0]), b = __uint_as_float(p[1]);
  u = a * plA + b * plB;
  v = a * prA + b * prB;
}

// A one-pole smoother towards a constant gain stops moving once k*(gain - g) rounds away (see
// skred_render_fast2.hip: fast2_smoother_stalled); wave-uniform, tested on the expression fast_post evaluates.
template <bool ENV>
__device__ __forceinline__ bool fast_smoother_stalled(const FastRegs &r) {
  const float gain = ENV ? r.gain_sustain : r.amp;
  const float nxt = r.sgain + r.k * (gain - r.sgain);
  return __all(r.nosmooth || __float_as_uint(nxt) == __float_as_uint(r.sgain));   // (a lane without the smoother never moves it)
}

// One voice, one frame.  STEADY: every lane of the wave sits in its sustain stage.  When !STEADY the
// caller has set r.tf / r.trf to this frame's envelope clocks.
// TAME: the caller has proved for every lane that lo <= phase <= hi and 0 <= inc <= span/2, which
// by induction keeps phase+inc in [lo, hi + span/2]: the only wrap that can occur is the simple one.
// EXTMS: the caller hands in the modulators' previous samples (`ms_ext`, `am_ext`, `pm_ext`: the skewed blocks of
// skred_render_fast.hip, whose modulator lanes run ahead and leave their samples in an LDS ring) instead of the ds_bpermute
// exchange.
// NOSTOP (with STOPS): the caller has proved that no lane of the wave is a stopping one-shot (XF_STOP clear), so the finish test and
// the clamps that only a finishing phase needs are dropped -- r.stop is false in every lane, the results are the same.
// BIDIR (with NOSTOP, instead of TAME): the caller has proved lo <= phase <= hi and |inc| <= span/2 for every lane -- deep
// frequency modulation that drives the increment below zero (7.sk, 0.sk) --, so phase + inc lies within half a loop length of
// the loop on either side and both wraps are straight-line: beyond the end lo + ((ph0 - lo) - span) as in TAME; before the start
// the reference's loop_end - fmodf(loop_start - phase, loop_length) (synth.c:253) with an argument below one loop length, where
// fmodf returns it unchanged: hi - (lo - ph0).  (That difference can round to hi itself: the fetch keeps its index clamp.)
// LOZ (with TAME and NOSTOP): no lane of the wave has a loop window -- fast_advance<LOZ>'s two-instruction wrap.
template <bool TAB_LDS, bool FILTER, bool ENV, bool STEADY, bool TAME, int INTERP, bool STOPS = false, bool EXTMS = false, bool NOSTOP = false,
          bool BIDIR = false, bool LOZ = false>
__device__ __forceinline__ void fast_frame(FastRegs &r, float &xn, float &xo, float &yn, float &yo,
                                           const bool released, const char *lds_tab,
                                           const char *__restrict__ glb_tab, float &out_l, float &out_r,
                                           const int xf = 0, const bool muted = false, const float white = 0.0f,
                                           const float ms_ext = 0.0f, const float am_ext = 0.0f, const float pm_ext = 0.0f) {
  float inc = r.inc;
  if (STOPS && (xf & (XF_FM | XF_AP))) {                // wave-uniform: some lane of the wave is modulated
    // voice_sample[m] as the previous frame left it (a modulator that is skipped this frame holds exact zero)
    const int mine = __float_as_int(r.sample);
    (void)mine;
    if (xf & XF_FM) {
      const float ms = EXTMS ? ms_ext : __int_as_float(__builtin_amdgcn_ds_bpermute(r.fm_addr, mine));
      if (r.fm_addr >= 0) inc = r.inc + r.fm_k * (ms * r.fm_depth);      // synth.c:551-554
    }
    if (xf & XF_AP) {
      r.am_prev = EXTMS ? am_ext : __int_as_float(__builtin_amdgcn_ds_bpermute(max(r.am_addr, 0), mine));
      r.pm_prev = EXTMS ? pm_ext : __int_as_float(__builtin_amdgcn_ds_bpermute(max(r.pm_addr, 0), mine));
    }
  }
  if (STOPS && (xf & XF_REV) && r.rev) inc = -inc;      // reverse playback, applied to the modulated increment
  float ph;
  if (BIDIR) {
    const float ph0 = r.phase + inc;
    const float over_ = r.lo + ((ph0 - r.lo) - r.span), under_ = r.hi - (r.lo - ph0);   // (both formed: two selects, no branch per lane)
    ph = ph0 >= r.hi ? over_ : (ph0 < r.lo ? under_ : ph0);
    r.phase = ph;
  } else {
    ph = fast_advance<TAME, STOPS && !NOSTOP, LOZ && TAME && (!STOPS || NOSTOP)>(r, inc);
  }
  float s = fast_fetch<TAB_LDS, INTERP, TAME && !BIDIR && (!STOPS || NOSTOP)>(lds_tab, glb_tab, r, ph);   // a finishing phase needs the index clamp
  if (STOPS && (xf & XF_NOISE) && r.noise) s = white;   // synth.c:543-546 (the lane's oscillator idles on inert numbers)
  if (STOPS && (xf & XF_HOLDQ)) {
    if (r.hold_max) {                                    // sample & hold, synth.c:560-571
      if (r.hold_count == 0) r.hold = s;
      s = r.hold;
      if (++r.hold_count >= r.hold_max) r.hold_count = 0;
    }
    if (r.quant) s = crush(s, r.quant);                  // synth.c:574
  }
  fast_post<FILTER, ENV, STEADY, false, STOPS>(r, s, xn, xo, yn, yo, released, out_l, out_r, muted, xf);
}

// the wave-private reduction tile of the 8-frame block paths: xt[8 frames][SK_XT]
#define SK_XT 68   /* floats per tile row: 64 + 4 keeps the 16-byte reads aligned and spreads the rows over the banks */
