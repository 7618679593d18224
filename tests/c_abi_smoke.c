/*
 * tests/c_abi_smoke.c -- the bank-mode C ABI (include/skred_amd.h) used from plain C, the way a C host would:
 * build a voice bank of sines, render blocks through skred_bank_render_host, push a parameter change and a
 * stamped note-off with skred_bank_update, queue another one, read the state back.  Compiled and run by
 * tests/test_c_abi.py on the GPU box; prints "OK" and exits 0 when every check holds.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "skred_amd.h"

#define N 1000
#define F 256
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "check failed: %s (line %d): %s\n", #c, __LINE__, skred_amd_last_error()); return 1; } } while (0)

static double rms(const float *x, int n) { double s = 0; for (int i = 0; i < n; i++) s += (double)x[i] * x[i]; return sqrt(s / n); }

int main(void) {
  CHECK(skred_amd_abi_version() == SKRED_AMD_ABI_VERSION);
  CHECK(skred_amd_device_count() > 0);
  /* one 4096-sample sine table, as wave_table_init builds it (synth.c:1231-1248) */
  enum { T = 4096 };
  float *table = malloc(T * sizeof(float));
  float ph = 0.0f;
  for (int i = 0; i < T; i++) { table[i] = sinf(2.0f * (float)M_PI * ph); ph += 1.0f / T; }

  skred_voice_bank_t v;
  memset(&v, 0, sizeof(v));
  v.n_voices = N;
#define ARR(field, type) v.field = calloc(N, sizeof(type))
  ARR(voice_phase, float); ARR(voice_phase_inc, float); ARR(voice_table_offset, int64_t); ARR(voice_table_size, int32_t);
  ARR(voice_one_shot, int32_t); ARR(voice_finished, int32_t); ARR(voice_loop_enabled, int32_t); ARR(voice_loop_valid, int32_t);
  ARR(voice_loop_start_f, float); ARR(voice_loop_end_f, float); ARR(voice_direction, int32_t); ARR(voice_wave_table_index, int32_t);
  ARR(voice_sample, float); ARR(voice_sample_hold, float); ARR(voice_sample_hold_count, int32_t); ARR(voice_sample_hold_max, int32_t);
  ARR(voice_quantize, int32_t); ARR(voice_amp, float); ARR(voice_use_amp_envelope, int32_t); ARR(voice_smoother_enable, int32_t);
  ARR(voice_smoother_gain, float); ARR(voice_smoother_smoothing, float); ARR(voice_filter_mode, int32_t);
  ARR(voice_filter, skred_mmf_t); ARR(voice_amp_envelope, skred_envelope_t);
  ARR(voice_pan_left, float); ARR(voice_pan_right, float); ARR(voice_disconnect, int32_t);
  ARR(voice_freq_mod_osc, int32_t); ARR(voice_freq_mod_depth, float); ARR(voice_freq_scale, float);
  ARR(voice_amp_mod_osc, int32_t); ARR(voice_amp_mod_depth, float); ARR(voice_pan_mod_osc, int32_t); ARR(voice_pan_mod_depth, float);
  ARR(voice_cz_mod_osc, int32_t); ARR(voice_cz_mod_depth, float); ARR(voice_cz_mode, int32_t); ARR(voice_cz_distortion, float);
  for (int i = 0; i < N; i++) {
    const float hz = 55.0f * powf(2.0f, (float)i / 120.0f);
    v.voice_phase_inc[i] = hz * (float)T / 48000.0f;
    v.voice_table_size[i] = T;
    v.voice_amp[i] = 0.5f;
    v.voice_smoother_enable[i] = 1; v.voice_smoother_smoothing[i] = 0.02f;
    v.voice_pan_left[i] = 0.5f; v.voice_pan_right[i] = 0.5f;
    v.voice_freq_mod_osc[i] = v.voice_amp_mod_osc[i] = v.voice_pan_mod_osc[i] = v.voice_cz_mod_osc[i] = -1;
    v.voice_use_amp_envelope[i] = 1;
    skred_envelope_t *e = &v.voice_amp_envelope[i];
    e->attack_time = 480.0f; e->decay_time = 4800.0f; e->sustain_level = 0.7f; e->release_time = 2400.0f;
    e->velocity = 1.0f; e->is_active = 1; e->sample_start = 0;
  }

  skred_bank_t *b = NULL;
  CHECK(skred_bank_create(0, N, &b) == SKRED_OK);
  CHECK(skred_bank_n_voices(b) == N);
  CHECK(skred_bank_set_tables_f32(b, table, T) == SKRED_OK);
  CHECK(skred_bank_upload(b, &v, 0, 0, N) == SKRED_OK);
  float *out = malloc(F * 2 * sizeof(float));
  double level[6];
  for (int k = 0; k < 6; k++) {
    if (k == 2) {                       /* a control action: voice 7 an octave up; note-off on voices 0..499 */
      int32_t seven = 7, half[500];
      v.voice_phase_inc[7] *= 2.0f;
      CHECK(skred_bank_update(b, &v, &seven, 1, SKRED_DIRTY_PARAMS, NULL) == SKRED_OK);
      for (int i = 0; i < 500; i++) half[i] = i;
      CHECK(skred_bank_defer(b, (uint64_t)(3 * F), &v, half, 500, SKRED_STAMP_RELEASE) == SKRED_OK);
      CHECK(skred_bank_queue_pending(b) == 1);
    }
    CHECK(skred_bank_run_queue(b, F, NULL) >= 0);                 /* seq() of the previous callback */
    CHECK(skred_bank_render_host(b, out, F, 2, SKRED_INTERP_TRUNCATE, NULL) == SKRED_OK);
    for (int i = 0; i < 2 * F; i++) CHECK(isfinite(out[i]));
    level[k] = rms(out, 2 * F);
  }
  CHECK(level[0] > 0.0 && level[5] > 0.0);
  CHECK(skred_bank_queue_pending(b) == 0);
  CHECK(skred_bank_last_kernel(b) == SKRED_KERNEL_FAST);
  skred_globals_t g;
  CHECK(skred_bank_get_globals(b, &g) == SKRED_OK && g.synth_sample_count == 6 * F);
  CHECK(skred_bank_download(b, &v, 0, 0, N) == SKRED_OK);
  int moved = 0;
  for (int i = 0; i < N; i++) moved += v.voice_phase[i] != 0.0f;
  CHECK(moved > N / 2);
  /* errors are reported, never papered over */
  CHECK(skred_bank_render_host(b, out, 0, 2, SKRED_INTERP_TRUNCATE, NULL) == SKRED_E_BAD_ARG);
  CHECK(skred_bank_upload(b, &v, 0, 1, N) == SKRED_E_RANGE);
  skred_bank_destroy(b);
  printf("OK rms per block: %.5f %.5f %.5f %.5f %.5f %.5f\n", level[0], level[1], level[2], level[3], level[4], level[5]);
  return 0;
}
