/*
 * oracle/cpu_ref_fxpt.c -- TEST INFRASTRUCTURE: the DEFINITION of the fixed-point render path.
 *
 * The reference (octetta/skred) has no fixed-point path, so there is nothing
 * upstream to pin this to: parity status "unpinned upstream".  This scalar C file is the
 * specification (restated in include/skred_amd_fxpt.h); the HIP kernel sk_fx_render_kernel must
 * match it bit for bit, including the integer mix.  Only tests/, smoke() and bench.py's
 * cpu_baseline leg may load it.
 */
#include <stdint.h>
#include <stddef.h>

#include "skred_amd.h"
#include "skred_amd_fxpt.h"

static inline uint32_t sat32(uint64_t x) { return x > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)x; }
static inline uint32_t recip32(uint32_t x) { return x ? (uint32_t)(0x100000000ull / x) : 0u; }

/* ADSR level in Q15 at global time `now`; clears is_active when the release has run out. */
static int32_t fx_envelope(skred_fxpt_bank_t *b, int v, uint64_t now) {
  if (!b->is_active[v]) return 0;
  const uint32_t A = b->attack_frames[v], D = b->decay_frames[v], R = b->release_frames[v];
  const int32_t S = b->sustain_q15[v];
  const uint32_t t = sat32(now - b->sample_start[v]);
  if (t < A) return (int32_t)((uint32_t)(t * recip32(A)) >> 17);
  if ((uint64_t)t < (uint64_t)A + D) {
    const int32_t prog = (int32_t)((uint32_t)((t - A) * recip32(D)) >> 17);
    return 32768 - ((prog * (32768 - S)) >> 15);
  }
  if (b->sample_release[v] == 0) return S;
  const uint32_t tr = sat32(now - b->sample_release[v]);
  if (tr < R) {
    const int32_t prog = (int32_t)((uint32_t)(tr * recip32(R)) >> 17);
    return S - ((prog * S) >> 15);
  }
  b->is_active[v] = 0;
  return 0;
}

/* direct form I biquad in Q2.30 x Q12 with int64 accumulation (include/skred_amd_fxpt.h: "biquad"); s in, s out */
static int32_t fx_biquad(skred_fxpt_bank_t *b, int v, int32_t s) {
  const int64_t x0 = (int64_t)s * 4096;                       /* s << 12, also for negative s */
  int64_t acc = (int64_t)b->b0_q30[v] * x0;
  acc += (int64_t)b->b1_q30[v] * b->x1[v];
  acc += (int64_t)b->b2_q30[v] * b->x2[v];
  acc -= (int64_t)b->a1_q30[v] * b->y1[v];
  acc -= (int64_t)b->a2_q30[v] * b->y2[v];
  int64_t y0 = (acc + ((int64_t)1 << 29)) >> 30;
  const int64_t lim = (int64_t)1 << 29;
  if (y0 < -lim) y0 = -lim;
  if (y0 > lim - 1) y0 = lim - 1;
  b->x2[v] = b->x1[v]; b->x1[v] = (int32_t)x0;
  b->y2[v] = b->y1[v]; b->y1[v] = (int32_t)y0;
  int32_t o = (int32_t)(y0 >> 12);
  if (o < -32768) o = -32768;
  if (o > 32767) o = 32767;
  return o;
}

/* mix: int64 [F][2]; stems: int32 [F][n][2] or NULL.  *count is synth_sample_count (advanced). */
int skred_cpuref_fx_render(skred_fxpt_bank_t *b, const int16_t *pool, uint64_t *count, int num_frames,
                           int interp, int64_t *mix, int32_t *stems) {
  if (!b || !pool || !count || !mix || num_frames < 0) return SKRED_E_BAD_ARG;
  uint64_t now = *count;
  const int n = b->n_voices;
  for (int i = 0; i < num_frames; i++) {
    now++;
    int64_t sum_l = 0, sum_r = 0;
    for (int v = 0; v < n; v++) {
      int32_t l = 0, r = 0;
      if (b->amp_q15[v] == 0 || b->finished[v]) {
        b->voice_sample[v] = 0;
      } else {
        const int L = b->log2_size[v];
        const int16_t *lut = pool + b->table_offset[v];
        uint32_t ph = b->phase[v] + b->phase_inc[v];
        int ends = 0;
        if (b->one_shot[v] && ph < b->phase[v]) {        /* the add carried: the one cycle is over */
          ph = 0xFFFFFFFFu;
          b->finished[v] = 1;
          ends = 1;
        }
        b->phase[v] = ph;
        const uint32_t idx = ph >> (32 - L);
        int32_t s = lut[idx];
        if (interp) {
          const int32_t nxt = ends ? s : lut[(idx + 1) & ((1u << L) - 1)];
          const int32_t frac = (int32_t)((uint32_t)(ph << L) >> 17);
          s = s + (((nxt - s) * frac) >> 15);
        }
        if (b->filter_mode[v]) s = fx_biquad(b, v, s);
        int32_t e = 32768;
        if (b->use_envelope[v]) e = (fx_envelope(b, v, now) * b->velocity_q15[v]) >> 15;
        int32_t gain = (int32_t)(((int64_t)b->amp_q15[v] * e) >> 15);
        if (b->smoother_enable[v]) {
          int32_t g = b->smoother_gain_q15[v];
          g += ((gain - g) * b->smoother_k_q15[v]) >> 15;
          b->smoother_gain_q15[v] = g;
          gain = g;
        }
        const int32_t out = (int32_t)(((int64_t)s * gain) >> 15);
        b->voice_sample[v] = out;
        if (!b->disconnect[v]) {
          l = (out * b->pan_left_q15[v]) >> 15;
          r = (out * b->pan_right_q15[v]) >> 15;
        }
      }
      sum_l += l; sum_r += r;
      if (stems) { stems[((size_t)i * n + v) * 2] = l; stems[((size_t)i * n + v) * 2 + 1] = r; }
    }
    mix[2 * i] = sum_l; mix[2 * i + 1] = sum_r;
  }
  *count = now;
  return SKRED_OK;
}

/* The master stage of the definition (include/skred_amd_fxpt.h: "master"): *gain_q31 is carried. */
int skred_cpuref_fx_master(int64_t target_q31, int32_t k_q15, int64_t *gain_q31, const int64_t *mix, int num_frames, int64_t *out) {
  if (!gain_q31 || !mix || !out || num_frames < 0) return SKRED_E_BAD_ARG;
  int64_t g = *gain_q31;
  for (int i = 0; i < num_frames; i++) {
    g += ((target_q31 - g) * (int64_t)k_q15) >> 15;
    const int64_t g15 = g >> 16;
    out[2 * i] = (mix[2 * i] * g15) >> 15;
    out[2 * i + 1] = (mix[2 * i + 1] * g15) >> 15;
  }
  *gain_q31 = g;
  return SKRED_OK;
}
