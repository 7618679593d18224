/*
 * skred_synth_persample.c -- the eight per-sample entry points that synth.h declares besides synth()
 * itself (reference synth.h:24-26,29,30,33,38,43), exported from libskred_synth.so so that a program
 * written against synth.h finds every symbol the header promises.
 *
 *      audio_rng_init / audio_rng_next / audio_rng_float   synth.c:105-123
 *      cz_phasor                                           synth.c:149-215  (+ fast_pow 140-147)
 *      osc_next                                            synth.c:217-275
 *      quantize_bits_int                                   synth.c:341-345
 *      mmf_process                                         synth.c:349-364
 *      amp_envelope_step                                   synth.c:398-431
 *
 * They act on ONE voice of the 64-voice global arrays for ONE sample, host side.  synth() of this
 * library never calls them: the render loop is the HIP kernels' (skred_synth_dropin.c: synth()).  They
 * exist for header completeness -- control code that wants to step a single voice by hand (the
 * reference's `:m` style probes) -- and tests/test_dropin.py pins each against the compiled reference
 * on the fixtures' voice states, bit for bit.  Compiled with -ffp-contract=off like everything that
 * must agree with the reference to the last bit.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "skred_synth_abi.h"

/* ------------------------------------------------------------------ the LCG behind the noise slots */

#define LCG_MUL 6364136223846793005ULL
#define LCG_ADD 1442695040888963407ULL

void audio_rng_init(uint64_t *rng, uint64_t seed) { *rng = seed == 0 ? 1 : seed; }

uint64_t audio_rng_next(uint64_t *rng) { return *rng = *rng * LCG_MUL + LCG_ADD; }

/* upper word as a signed fraction of 2^31: [-1, 1) */
float audio_rng_float(uint64_t *rng) {
  const int32_t hi = (int32_t)(uint32_t)(audio_rng_next(rng) >> 32);
  return (float)hi / 2147483648.0f;
}

/* ------------------------------------------------------------------ bit-crush */

float quantize_bits_int(float v, int bits) {
  const int top = (1 << bits) - 1;
  const double scaled = (double)(v * (float)top) + 0.5;    /* the reference's +0.5 is a double add */
  return (float)(int)scaled * (1.0f / (float)top);
}

/* ------------------------------------------------------------------ phase distortion */

/* a^b through the exponent field: the reference's approximation, kept because its error IS the sound */
static float pow_by_exponent_bits(float a, float b) {
  if (a <= 0.0f) return 0.0f;              /* a NaN goes through the bit arithmetic, as in the reference */
  int32_t bits;
  memcpy(&bits, &a, sizeof bits);
  const int32_t one = 0x3f800000;
  bits = (int32_t)(b * (float)(bits - one) + (float)one);
  float r;
  memcpy(&r, &bits, sizeof r);
  return r;
}

/* the two-segment warps: below `knee` the phase is stretched by `lo`; above it the remainder is
 * stretched by `hi` and re-based at 0.5 */
static float warp_two_slopes(float x, float knee, float lo, float hi) {
  return x < knee ? x * lo : 0.5f + (x - knee) * hi;
}

float cz_phasor(int n, float p, float d, int table_size) {
  if (n < 1 || n > 7) return p;
  const float size = (float)table_size;
  float x = p / size;
  if (d < 0.0f) d = 0.0f; else if (d > 0.999f) d = 0.999f;
  const float narrow = 0.5f / (0.5f - d * 0.5f);           /* slope of a half compressed by d */
  switch (n) {
    case 1: x = warp_two_slopes(x, d, 0.5f / d, 0.5f / (1.0f - d)); break;
    case 2: x = x < 0.5f ? x * narrow : 1.0f - (1.0f - x) * narrow; break;
    case 3: x = warp_two_slopes(x, 0.5f, narrow, narrow); break;
    case 4: x = fmodf(x * 2.0f, 1.0f); break;
    case 5: x = warp_two_slopes(x, 0.5f, narrow, 0.5f / (0.5f + d * 0.5f)); break;
    case 6: x = pow_by_exponent_bits(x, 1.0f + 4.0f * d); break;
    default: x = pow_by_exponent_bits(x, 1.0f + 8.0f * d); break;
  }
  return x * size;
}

/* ------------------------------------------------------------------ oscillator */

float osc_next(int voice, float phase_inc) {
  if (voice_finished[voice]) return 0.0f;
  const int size = voice_table_size[voice];
  const int stops = voice_one_shot[voice] && !voice_loop_enabled[voice];   /* plays once, then finishes */
  const int window = voice_loop_enabled[voice] && voice_loop_valid[voice];
  const float lo = window ? voice_loop_start_f[voice] : 0.0f;
  const float hi = window ? voice_loop_end_f[voice] : (float)size;
  const float span = hi - lo;

  float ph = voice_phase[voice] + (voice_direction[voice] ? -phase_inc : phase_inc);
  if (!isfinite(ph)) {                       /* a voice that ran away is parked at 0 */
    voice_phase[voice] = 0.0f;
    voice_finished[voice] = voice_one_shot[voice] ? 1 : 0;
    return 0.0f;
  }
  if (ph >= hi) {
    if (stops) { ph = hi - 1e-6f; voice_finished[voice] = 1; }
    else ph = lo + fmodf(ph - lo, span);
  } else if (ph < lo) {
    if (stops) { ph = lo; voice_finished[voice] = 1; }
    else ph = hi - fmodf(lo - ph, span);
  }
  voice_phase[voice] = ph;

  int idx;
  const int cz = voice_cz_mode[voice];
  if (cz) {
    const int m = voice_cz_mod_osc[voice];
    const float wobble = m >= 0 ? voice_sample[m] * voice_cz_mod_depth[voice] : 1.0f;
    idx = (int)cz_phasor(cz, ph, voice_cz_distortion[voice] + wobble, size);
  } else {
    idx = (int)ph;
  }
  if (idx > size - 1) idx = size - 1;
  if (idx < 0) idx = 0;
  return voice_table[voice][idx];
}

/* ------------------------------------------------------------------ biquad, one sample */

float mmf_process(int n, float input) {
  skred_mmf_t *q = &voice_filter[n];
  float y = q->b0 * input;                 /* left to right: the association is part of the contract */
  y = y + q->b1 * q->x1;
  y = y + q->b2 * q->x2;
  y = y - q->a1 * q->y1;
  y = y - q->a2 * q->y2;
  q->x2 = q->x1; q->x1 = input;
  q->y2 = q->y1; q->y1 = y;
  return y;
}

/* ------------------------------------------------------------------ linear ADSR on the global clock */

float amp_envelope_step(int v) {
  skred_envelope_t *e = &voice_amp_envelope[v];
  if (!e->is_active) return 0.0f;
  const uint64_t now = synth_sample_count;
  const float t = (float)(now - e->sample_start);
  const float A = e->attack_time, D = e->decay_time, S = e->sustain_level;
  if (t < A) return t / A;
  if (t < A + D) return 1.0f - ((t - A) / D) * (1.0f - S);
  if (e->sample_release == 0) return S;
  const float tr = (float)(now - e->sample_release);
  if (tr < e->release_time) return S * (1.0f - tr / e->release_time);
  e->is_active = 0;
  return 0.0f;
}
