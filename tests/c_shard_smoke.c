/*
 * tests/c_shard_smoke.c -- the sharded render (include/skred_amd.h: skred_shard_*) used from plain C with the one GPU a
 * test box has: a two-rank partition is checked on paper, a one-rank shard renders blocks (a) without a collective,
 * (b) through a host-supplied reduce step and (c) through the library's own RCCL communicator with one rank, and every
 * form must deliver the bytes skred_bank_render_host() delivers for the same bank.  Needs hipMemcpy only to look at the
 * device output.  Compiled and run by tests/test_c_abi.py; prints "OK" and exits 0 when every check holds.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include "skred_amd.h"

#define N 3000
#define F 300
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "check failed: %s (line %d): %s\n", #c, __LINE__, skred_amd_last_error()); return 1; } } while (0)

static int reduce_calls;
static int one_rank_reduce(void *ctx, float *partial, size_t n, int root, void *stream) {
  (void)ctx; (void)partial; (void)stream;
  reduce_calls += (n == (size_t)F * 2 && root == 0);      /* one rank: the sum over ranks is the buffer itself */
  return SKRED_OK;
}

int main(void) {
  CHECK(skred_amd_device_count() > 0);
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
    v.voice_phase_inc[i] = 55.0f * powf(2.0f, (float)i / 400.0f) * (float)T / 48000.0f;
    v.voice_table_size[i] = T;
    v.voice_amp[i] = 0.25f;
    v.voice_smoother_enable[i] = 1; v.voice_smoother_smoothing[i] = 0.02f;
    v.voice_pan_left[i] = 0.25f + 0.5f * (float)(i % 7) / 7.0f; v.voice_pan_right[i] = 1.0f - v.voice_pan_left[i];
    v.voice_freq_mod_osc[i] = v.voice_amp_mod_osc[i] = v.voice_pan_mod_osc[i] = v.voice_cz_mod_osc[i] = -1;
  }

  /* the partition rule and the legality of a cut, no device involved */
  int lo, hi;
  CHECK(skred_shard_partition(N, 2, 0, &lo, &hi) == SKRED_OK && lo == 0 && hi == N / 2);
  CHECK(skred_shard_partition(N, 2, 1, &lo, &hi) == SKRED_OK && lo == N / 2 && hi == N);
  CHECK(skred_shard_cut_ok(&v, 0, N / 2) == 1);
  v.voice_amp_mod_osc[5] = N - 1;                            /* carrier on rank 0, modulator on rank 1 */
  CHECK(skred_shard_cut_ok(&v, 0, N / 2) == 0 && skred_shard_cut_ok(&v, N / 2, N) == 1);
  v.voice_amp_mod_osc[5] = -1;

  /* the same bank through the plain bank ABI: the bytes every sharded form must reproduce */
  float *want = malloc(3 * F * 2 * sizeof(float)), *got = malloc(F * 2 * sizeof(float));
  skred_bank_t *b = NULL;
  CHECK(skred_bank_create(0, N, &b) == SKRED_OK);
  CHECK(skred_bank_set_tables_f32(b, table, T) == SKRED_OK);
  CHECK(skred_bank_upload(b, &v, 0, 0, N) == SKRED_OK);
  for (int k = 0; k < 3; k++) CHECK(skred_bank_render_host(b, want + k * F * 2, F, 2, SKRED_INTERP_TRUNCATE, NULL) == SKRED_OK);
  skred_bank_destroy(b);

  skred_shard_t *s = NULL;
  CHECK(skred_shard_create(0, 0, 2, 0, N, &s) == SKRED_OK);  /* rank 0 of 2 owns half the bank */
  CHECK(skred_shard_range(s, &lo, &hi) == SKRED_OK && hi - lo == N / 2 && skred_bank_n_voices(skred_shard_bank(s)) == N / 2);
  CHECK(skred_shard_render_mix(s, F, SKRED_INTERP_TRUNCATE, NULL, NULL, 2, NULL) == SKRED_E_BAD_ARG);   /* the root needs an output */
  skred_shard_destroy(s);

  CHECK(skred_shard_create(0, 0, 1, 0, N, &s) == SKRED_OK);
  CHECK(skred_bank_set_tables_f32(skred_shard_bank(s), table, T) == SKRED_OK);
  v.voice_amp_mod_osc[5] = N + 7;                            /* modulator outside the bank: the upload must refuse */
  CHECK(skred_shard_upload(s, &v) == SKRED_E_UNSUPPORTED);
  v.voice_amp_mod_osc[5] = -1;
  CHECK(skred_shard_upload(s, &v) == SKRED_OK);
  float *d_out = NULL;
  CHECK(hipMalloc((void **)&d_out, F * 2 * sizeof(float)) == hipSuccess);
  /* (a) one rank, no collective */
  CHECK(skred_shard_render_mix(s, F, SKRED_INTERP_TRUNCATE, NULL, d_out, 2, NULL) == SKRED_OK);
  CHECK(hipMemcpy(got, d_out, F * 2 * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess);
  CHECK(memcmp(got, want, F * 2 * sizeof(float)) == 0);
  /* (b) the N>1 sequence with a host-supplied reduce step */
  skred_shard_ops_t ops;
  memset(&ops, 0, sizeof(ops));
  ops.reduce = one_rank_reduce;
  CHECK(skred_shard_set_ops(s, &ops, 1) == SKRED_OK);
  CHECK(skred_shard_render_mix(s, F, SKRED_INTERP_TRUNCATE, NULL, d_out, 2, NULL) == SKRED_OK);
  CHECK(hipMemcpy(got, d_out, F * 2 * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess);
  CHECK(reduce_calls == 1 && memcmp(got, want + F * 2, F * 2 * sizeof(float)) == 0);
  skred_shard_destroy(s);
  /* (c) the library's own RCCL communicator (one rank): ncclReduce in place on the root */
  CHECK(skred_shard_create(0, 0, 1, 0, N, &s) == SKRED_OK);
  CHECK(skred_bank_set_tables_f32(skred_shard_bank(s), table, T) == SKRED_OK);
  CHECK(skred_shard_upload(s, &v) == SKRED_OK);
  char id[128];
  CHECK(skred_shard_rccl_unique_id(id) == SKRED_OK);
  CHECK(skred_shard_init_rccl(s, id) == SKRED_OK);
  CHECK(skred_shard_set_ops(s, NULL, 1) == SKRED_OK);
  for (int k = 0; k < 3; k++) {
    CHECK(skred_shard_render_mix(s, F, SKRED_INTERP_TRUNCATE, NULL, d_out, 2, NULL) == SKRED_OK);
    CHECK(hipMemcpy(got, d_out, F * 2 * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess);
    CHECK(memcmp(got, want + k * F * 2, F * 2 * sizeof(float)) == 0);
  }
  skred_shard_destroy(s);
  /* (d) the PIPELINED form (the collective of block k beside the render of block k + 1) through the same one-rank RCCL
   * communicator: two output buffers in alternation, block k's output complete once call k + 2 has returned (host-paced) or
   * on the stream after a flush; same bytes */
  CHECK(skred_shard_create(0, 0, 1, 0, N, &s) == SKRED_OK);
  CHECK(skred_bank_set_tables_f32(skred_shard_bank(s), table, T) == SKRED_OK);
  CHECK(skred_shard_upload(s, &v) == SKRED_OK);
  CHECK(skred_shard_rccl_unique_id(id) == SKRED_OK);
  CHECK(skred_shard_init_rccl(s, id) == SKRED_OK);
  CHECK(skred_shard_set_ops(s, NULL, 1) == SKRED_OK);
  float *d_out2[3] = {NULL, NULL, NULL};   /* (three: the test also reads block 0 while block 2 is in flight) */
  hipStream_t st = NULL;
  CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess);
  for (int i = 0; i < 3; i++) CHECK(hipMalloc((void **)&d_out2[i], F * 2 * sizeof(float)) == hipSuccess);
  CHECK(skred_shard_render_mix_pipelined(s, F, SKRED_INTERP_TRUNCATE, NULL, 2, st) == SKRED_E_BAD_ARG);   /* the root needs an output */
  hipStream_t other = NULL;                 /* (a stream that has nothing to do with the shard: only the host's pacing orders it) */
  CHECK(hipStreamCreateWithFlags(&other, hipStreamNonBlocking) == hipSuccess);
  for (int k = 0; k < 3; k++) {
    CHECK(skred_shard_render_mix_pipelined(s, F, SKRED_INTERP_TRUNCATE, d_out2[k], 2, st) == SKRED_OK);
    if (k == 1) {            /* the latest block on `st` itself: after a flush */
      CHECK(skred_shard_flush(s, st) == SKRED_OK);
      CHECK(hipMemcpyAsync(got, d_out2[1], F * 2 * sizeof(float), hipMemcpyDeviceToHost, st) == hipSuccess);
      CHECK(hipStreamSynchronize(st) == hipSuccess);
      CHECK(memcmp(got, want + 1 * F * 2, F * 2 * sizeof(float)) == 0);
    }
    if (k == 2) {            /* call k has waited for block k - 2 on the host: its output is complete for every stream */
      CHECK(hipMemcpyAsync(got, d_out2[0], F * 2 * sizeof(float), hipMemcpyDeviceToHost, other) == hipSuccess);
      CHECK(hipStreamSynchronize(other) == hipSuccess);
      CHECK(memcmp(got, want + 0 * F * 2, F * 2 * sizeof(float)) == 0);
    }
  }
  CHECK(skred_shard_flush(s, st) == SKRED_OK);
  CHECK(hipMemcpyAsync(got, d_out2[2], F * 2 * sizeof(float), hipMemcpyDeviceToHost, st) == hipSuccess);
  CHECK(hipStreamSynchronize(st) == hipSuccess);
  CHECK(memcmp(got, want + 2 * F * 2, F * 2 * sizeof(float)) == 0);
  (void)hipStreamDestroy(other);
  skred_shard_destroy(s);
  for (int i = 0; i < 3; i++) (void)hipFree(d_out2[i]);
  (void)hipStreamDestroy(st);
  (void)hipFree(d_out);
  printf("OK sharded forms equal the single bank, RCCL one-rank reduce and the pipelined form included\n");
  return 0;
}
