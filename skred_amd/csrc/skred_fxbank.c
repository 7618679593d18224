/*
 * skred_fxbank.c -- C host shim of the fixed-point path (include/skred_amd_fxpt.h).  Mirrors
 * skred_bank.c: owns the HBM planes, packs the host arrays, sequences sk_fx_render_kernel and the
 * int64 partial reduction.  No CPU rendering.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "skred_amd.h"
#include "skred_amd_fxpt.h"
#include "skred_fx_layout.h"

int skx_launch_render(const skx_args_t *args, int n_workgroups, hipStream_t stream);
int skx_launch_master_apply(const long long *sum, const int32_t *gains, long long *out, int num_frames, const long long *gain_pending,
                            long long *gain_state, hipStream_t stream);
int skx_launch_stamp(const int32_t *d_ids, int n, int which, skx_plane_t *time_plane, skx_plane_t *rw0, uint64_t now, hipStream_t stream);
int skred_amd_set_error(int code, const char *fmt, ...);   /* skred_bank.c */

struct skred_fxbank {
  int device, n_voices, n_padded, n_groups;
  skx_plane_t *d_ro[SKX_COUNT];
  skx_plane_t *d_rw[SKX_RW_COUNT];
  int n_filter;                 /* voices with filter_mode != 0 (recounted on whole-bank uploads, grown otherwise) */
  int16_t *d_tables;
  size_t table_entries, table_bytes_padded;
  long long *d_partial; size_t partial_cap;  /* [n_wg][F][2] rows, [SKX_FINISH_SLABS][F][2] slab sums, then int32 gains[F] */
  uint32_t *d_tickets;          /* [SKX_FINISH_SLABS + 1] arrival counters of the in-kernel mix-down */
  long long *d_gain_state;      /* [0] Q31 master gain carried between blocks; [1] the gain a sum-only render prepared for skred_fxbank_master */
  long long master_target_q31;  /* default: 0.025 (the float path's volume_final) in Q31 */
  int32_t master_k_q15;         /* default: 0.002 in Q15 */
  int gains_frames;             /* > 0: the latest sum-only render left the gains of a block of this many frames */
  size_t gains_offset;          /* ... at this int64 offset into d_partial */
  int32_t *d_ids; int32_t *h_ids; size_t ids_cap;   /* staging of skred_fxbank_stamp */
  long long *d_mix; size_t mix_cap;
  int32_t *d_stems; size_t stems_cap;
  uint64_t count;
  hipEvent_t ev0, ev1;
  int timed;
};

#define HIP_TRY(call)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess) return skred_amd_set_error(SKRED_E_NO_DEVICE, "%s -> %s", #call, hipGetErrorString(e_)); \
  } while (0)

static int grow_bytes(void **buf, size_t *cap, size_t need) {
  if (*cap >= need) return SKRED_OK;
  if (*buf) { hipFree(*buf); *buf = NULL; *cap = 0; }
  HIP_TRY(hipMalloc(buf, need));
  *cap = need;
  return SKRED_OK;
}

int skred_fxbank_create(int device, int n_voices, skred_fxbank_t **out) {
  if (!out || n_voices <= 0) return skred_amd_set_error(SKRED_E_BAD_ARG, "skred_fxbank_create: bad arguments");
  *out = NULL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return skred_amd_set_error(SKRED_E_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
  if (device < 0 || device >= ndev) return skred_amd_set_error(SKRED_E_BAD_ARG, "device %d of %d", device, ndev);
  HIP_TRY(hipSetDevice(device));
  skred_fxbank_t *fx = (skred_fxbank_t *)calloc(1, sizeof(*fx));
  if (!fx) return skred_amd_set_error(SKRED_E_NO_MEM, "calloc");
  fx->device = device; fx->n_voices = n_voices;
  fx->n_groups = (n_voices + SKX_GROUP - 1) / SKX_GROUP;
  fx->n_padded = fx->n_groups * SKX_GROUP;
  const size_t bytes = (size_t)fx->n_padded * sizeof(skx_plane_t);
  for (int p = 0; p < SKX_COUNT; p++) { HIP_TRY(hipMalloc((void **)&fx->d_ro[p], bytes)); HIP_TRY(hipMemset(fx->d_ro[p], 0, bytes)); }
  for (int p = 0; p < SKX_RW_COUNT; p++) { HIP_TRY(hipMalloc((void **)&fx->d_rw[p], bytes)); HIP_TRY(hipMemset(fx->d_rw[p], 0, bytes)); }
  skx_plane_t *inert = (skx_plane_t *)calloc((size_t)fx->n_padded, sizeof(skx_plane_t));
  if (!inert) return skred_amd_set_error(SKRED_E_NO_MEM, "calloc");
  for (int v = 0; v < fx->n_padded; v++) inert[v].w[2] = 3u | (SKXF_INERT << 8);   /* log2_size 3, amp 0 */
  hipError_t e = hipMemcpy(fx->d_ro[SKX_OSC], inert, bytes, hipMemcpyHostToDevice);
  free(inert);
  HIP_TRY(e);
  HIP_TRY(hipEventCreate(&fx->ev0)); HIP_TRY(hipEventCreate(&fx->ev1));
  HIP_TRY(hipMalloc((void **)&fx->d_tickets, (SKX_FINISH_SLABS + 1) * sizeof(uint32_t)));
  HIP_TRY(hipMemset(fx->d_tickets, 0, (SKX_FINISH_SLABS + 1) * sizeof(uint32_t)));
  HIP_TRY(hipMalloc((void **)&fx->d_gain_state, 4 * sizeof(long long)));
  HIP_TRY(hipMemset(fx->d_gain_state, 0, 4 * sizeof(long long)));
  fx->master_target_q31 = (long long)(0.025 * 2147483648.0);    /* volume_user 1 * AMY_FACTOR (synth.c:96-100) */
  fx->master_k_q15 = 66;                                        /* 0.002 (synth.c:92) */
  *out = fx;
  return SKRED_OK;
}

void skred_fxbank_destroy(skred_fxbank_t *fx) {
  if (!fx) return;
  hipSetDevice(fx->device);
  for (int p = 0; p < SKX_COUNT; p++) if (fx->d_ro[p]) hipFree(fx->d_ro[p]);
  for (int p = 0; p < SKX_RW_COUNT; p++) if (fx->d_rw[p]) hipFree(fx->d_rw[p]);
  if (fx->d_tables) hipFree(fx->d_tables);
  if (fx->d_partial) hipFree(fx->d_partial);
  if (fx->d_tickets) hipFree(fx->d_tickets);
  if (fx->d_gain_state) hipFree(fx->d_gain_state);
  if (fx->d_ids) hipFree(fx->d_ids);
  if (fx->h_ids) hipHostFree(fx->h_ids);
  if (fx->d_mix) hipFree(fx->d_mix);
  if (fx->d_stems) hipFree(fx->d_stems);
  if (fx->ev0) hipEventDestroy(fx->ev0);
  if (fx->ev1) hipEventDestroy(fx->ev1);
  free(fx);
}

int skred_fxbank_set_tables_i16(skred_fxbank_t *fx, const int16_t *pool, size_t n) {
  if (!fx || !pool || n == 0) return skred_amd_set_error(SKRED_E_BAD_ARG, "fx set_tables: bad arguments");
  HIP_TRY(hipSetDevice(fx->device));
  if (fx->d_tables) { hipFree(fx->d_tables); fx->d_tables = NULL; }
  fx->table_entries = n;
  fx->table_bytes_padded = (n * sizeof(int16_t) + 15) & ~(size_t)15;
  HIP_TRY(hipMalloc((void **)&fx->d_tables, fx->table_bytes_padded));
  HIP_TRY(hipMemset(fx->d_tables, 0, fx->table_bytes_padded));
  HIP_TRY(hipMemcpy(fx->d_tables, pool, n * sizeof(int16_t), hipMemcpyHostToDevice));
  return SKRED_OK;
}

static uint32_t recip32(uint32_t x) { return x ? (uint32_t)(0x100000000ull / x) : 0u; }
/* the fields the biquad and the one-shots brought (filter_mode .. y2): a caller that zero-initialises the struct and leaves
 * them NULL gets what it got before they existed -- no filter, no one-shot, a delay line at rest */
#define FX_OPT(arr, v) ((arr) ? (arr)[v] : 0)

int skred_fxbank_upload(skred_fxbank_t *fx, const skred_fxpt_bank_t *h, int src_first, int dst_first, int count) {
  if (!fx || !h || count < 0) return skred_amd_set_error(SKRED_E_BAD_ARG, "fx upload: bad arguments");
  if (src_first < 0 || src_first + count > h->n_voices || dst_first < 0 || dst_first + count > fx->n_voices)
    return skred_amd_set_error(SKRED_E_RANGE, "fx upload window outside bank");
  if (count == 0) return SKRED_OK;
  HIP_TRY(hipSetDevice(fx->device));
  skx_plane_t *st = (skx_plane_t *)calloc((size_t)(SKX_COUNT + SKX_RW_COUNT) * (size_t)count, sizeof(skx_plane_t));
  if (!st) return skred_amd_set_error(SKRED_E_NO_MEM, "fx upload staging");
  if (dst_first == 0 && count == fx->n_voices) fx->n_filter = 0;          /* whole bank replaced */
  for (int i = 0; i < count; i++) {
    const int v = src_first + i;
    const int L = h->log2_size[v];
    if (L < 3 || L > 15 || h->table_offset[v] < 0 ||
        (size_t)h->table_offset[v] + ((size_t)1 << L) > fx->table_entries) {
      free(st);
      return skred_amd_set_error(SKRED_E_RANGE, "fx voice %d: table [%d,+2^%d) outside pool of %zu entries", v, h->table_offset[v], L, fx->table_entries);
    }
    if (h->amp_q15[v] < 0 || h->amp_q15[v] > 65535) { free(st); return skred_amd_set_error(SKRED_E_RANGE, "fx voice %d: amp_q15 %d outside 0..65535", v, h->amp_q15[v]); }
    uint32_t flags = 0;
    if (h->use_envelope[v]) flags |= SKXF_USE_ENV;
    if (h->smoother_enable[v]) flags |= SKXF_SMOOTH;
    if (h->disconnect[v]) flags |= SKXF_MUTED;
    if (FX_OPT(h->filter_mode, v)) { flags |= SKXF_FILTER; fx->n_filter++; }
    if (FX_OPT(h->one_shot, v)) flags |= SKXF_ONE_SHOT;
    {
      const int64_t lim = (int64_t)1 << 29;      /* the delay line the definition can produce: |x|, |y| < 2^29 */
      const int32_t d[4] = { FX_OPT(h->x1, v), FX_OPT(h->x2, v), FX_OPT(h->y1, v), FX_OPT(h->y2, v) };
      for (int k = 0; k < 4; k++)
        if (d[k] < -lim || d[k] >= lim) { free(st); return skred_amd_set_error(SKRED_E_RANGE, "fx voice %d: filter state %d outside +-2^29", v, d[k]); }
    }
#define P(p) st[(size_t)(p) * count + i]
    P(SKX_OSC).w[0] = h->phase_inc[v]; P(SKX_OSC).w[1] = (uint32_t)h->table_offset[v];
    P(SKX_OSC).w[2] = (uint32_t)L | (flags << 8); P(SKX_OSC).w[3] = (uint32_t)h->amp_q15[v];
    P(SKX_GAIN).w[0] = (uint32_t)h->pan_left_q15[v]; P(SKX_GAIN).w[1] = (uint32_t)h->pan_right_q15[v];
    P(SKX_GAIN).w[2] = (uint32_t)h->smoother_k_q15[v]; P(SKX_GAIN).w[3] = (uint32_t)h->velocity_q15[v];
    P(SKX_ENV).w[0] = h->attack_frames[v]; P(SKX_ENV).w[1] = h->decay_frames[v];
    P(SKX_ENV).w[2] = h->release_frames[v]; P(SKX_ENV).w[3] = (uint32_t)h->sustain_q15[v];
    P(SKX_RECIP).w[0] = recip32(h->attack_frames[v]); P(SKX_RECIP).w[1] = recip32(h->decay_frames[v]);
    P(SKX_RECIP).w[2] = recip32(h->release_frames[v]);
    P(SKX_TIME).w[0] = (uint32_t)h->sample_start[v]; P(SKX_TIME).w[1] = (uint32_t)(h->sample_start[v] >> 32);
    P(SKX_TIME).w[2] = (uint32_t)h->sample_release[v]; P(SKX_TIME).w[3] = (uint32_t)(h->sample_release[v] >> 32);
    P(SKX_FILT).w[0] = (uint32_t)FX_OPT(h->b0_q30, v); P(SKX_FILT).w[1] = (uint32_t)FX_OPT(h->b1_q30, v);
    P(SKX_FILT).w[2] = (uint32_t)FX_OPT(h->b2_q30, v); P(SKX_FILT).w[3] = (uint32_t)FX_OPT(h->a1_q30, v);
    P(SKX_FILT2).w[0] = (uint32_t)FX_OPT(h->a2_q30, v);
    P(SKX_COUNT).w[0] = h->phase[v]; P(SKX_COUNT).w[1] = (uint32_t)h->smoother_gain_q15[v];
    P(SKX_COUNT).w[2] = (uint32_t)h->voice_sample[v]; P(SKX_COUNT).w[3] = (h->is_active[v] ? 1u : 0u) | (FX_OPT(h->finished, v) ? 2u : 0u);
    P(SKX_COUNT + 1).w[0] = (uint32_t)FX_OPT(h->x1, v); P(SKX_COUNT + 1).w[1] = (uint32_t)FX_OPT(h->x2, v);
    P(SKX_COUNT + 1).w[2] = (uint32_t)FX_OPT(h->y1, v); P(SKX_COUNT + 1).w[3] = (uint32_t)FX_OPT(h->y2, v);
#undef P
  }
  const size_t bytes = (size_t)count * sizeof(skx_plane_t);
  hipError_t e = hipSuccess;
  for (int p = 0; p < SKX_COUNT && e == hipSuccess; p++)
    e = hipMemcpy(fx->d_ro[p] + dst_first, st + (size_t)p * count, bytes, hipMemcpyHostToDevice);
  for (int p = 0; p < SKX_RW_COUNT && e == hipSuccess; p++)
    e = hipMemcpy(fx->d_rw[p] + dst_first, st + (size_t)(SKX_COUNT + p) * count, bytes, hipMemcpyHostToDevice);
  free(st);
  HIP_TRY(e);
  return SKRED_OK;
}

int skred_fxbank_download(skred_fxbank_t *fx, skred_fxpt_bank_t *h, int src_first, int dst_first, int count) {
  if (!fx || !h || count < 0) return skred_amd_set_error(SKRED_E_BAD_ARG, "fx download: bad arguments");
  if (src_first < 0 || src_first + count > fx->n_voices || dst_first < 0 || dst_first + count > h->n_voices)
    return skred_amd_set_error(SKRED_E_RANGE, "fx download window outside bank");
  if (count == 0) return SKRED_OK;
  HIP_TRY(hipSetDevice(fx->device));
  skx_plane_t *st = (skx_plane_t *)malloc((size_t)SKX_RW_COUNT * (size_t)count * sizeof(skx_plane_t));
  if (!st) return skred_amd_set_error(SKRED_E_NO_MEM, "fx download staging");
  HIP_TRY(hipDeviceSynchronize());
  hipError_t e = hipSuccess;
  for (int p = 0; p < SKX_RW_COUNT && e == hipSuccess; p++)
    e = hipMemcpy(st + (size_t)p * count, fx->d_rw[p] + src_first, (size_t)count * sizeof(skx_plane_t), hipMemcpyDeviceToHost);
  if (e != hipSuccess) { free(st); HIP_TRY(e); }
  for (int i = 0; i < count; i++) {
    const int v = dst_first + i;
    const skx_plane_t *f = &st[(size_t)count + i];
    h->phase[v] = st[i].w[0];
    h->smoother_gain_q15[v] = (int32_t)st[i].w[1];
    h->voice_sample[v] = (int32_t)st[i].w[2];
    h->is_active[v] = (int32_t)(st[i].w[3] & 1u);
    if (h->finished) h->finished[v] = (int32_t)((st[i].w[3] >> 1) & 1u);
    if (h->x1) h->x1[v] = (int32_t)f->w[0];
    if (h->x2) h->x2[v] = (int32_t)f->w[1];
    if (h->y1) h->y1[v] = (int32_t)f->w[2];
    if (h->y2) h->y2[v] = (int32_t)f->w[3];
  }
  free(st);
  return SKRED_OK;
}

int skred_fxbank_set_sample_count(skred_fxbank_t *fx, uint64_t c) { if (!fx) return SKRED_E_BAD_ARG; fx->count = c; return SKRED_OK; }
uint64_t skred_fxbank_get_sample_count(const skred_fxbank_t *fx) { return fx ? fx->count : 0; }

/* one block: the render kernel, whose last-arriving workgroups add the rows up into `d_sum` (pre-master, may be NULL) and / or,
 * scaled by the master gain of each frame, into `d_out` */
static int fx_block(skred_fxbank_t *fx, int num_frames, int interp, int64_t *d_sum, int64_t *d_out, int32_t *d_stems, hipStream_t s) {
  if (!fx->d_tables) return skred_amd_set_error(SKRED_E_BAD_ARG, "fx render: no table pool set");
  HIP_TRY(hipSetDevice(fx->device));
  const int n_wg = fx->n_groups < SKX_MAX_WORKGROUPS ? fx->n_groups : SKX_MAX_WORKGROUPS;
  const size_t row = (size_t)num_frames * 2;
  const size_t gains_at = ((size_t)n_wg + SKX_FINISH_SLABS) * row;                   /* int64 units; the gains are int32 behind */
  int rc = grow_bytes((void **)&fx->d_partial, &fx->partial_cap, gains_at * sizeof(long long) + (size_t)num_frames * sizeof(int32_t));
  if (rc) return rc;
  skx_args_t a;
  memset(&a, 0, sizeof(a));
  for (int p = 0; p < SKX_COUNT; p++) a.ro[p] = fx->d_ro[p];
  for (int p = 0; p < SKX_RW_COUNT; p++) a.rw[p] = fx->d_rw[p];
  a.any_filter = fx->n_filter > 0;
  a.tables = fx->d_tables; a.partial = fx->d_partial; a.stems = d_stems;
  a.count0 = fx->count; a.n_voices = fx->n_voices; a.n_groups = fx->n_groups;
  a.num_frames = num_frames; a.interp = interp ? 1 : 0;
  a.lds_bytes_tables = fx->table_bytes_padded <= SKX_LDS_TABLE_MAX_BYTES ? (int32_t)fx->table_bytes_padded : 0;
  a.n_rows = n_wg;
  a.slab_rows = fx->d_partial + (size_t)n_wg * row;
  a.gains = (int32_t *)(fx->d_partial + gains_at);
  a.tickets = fx->d_tickets;
  a.sum_out = (long long *)d_sum;
  a.mix_out = (long long *)d_out;
  a.gain_state = fx->d_gain_state;
  a.gain_commit = d_out ? fx->d_gain_state : fx->d_gain_state + 1;
  a.master_target_q31 = fx->master_target_q31;
  a.master_k_q15 = fx->master_k_q15;
  fx->gains_frames = d_out ? 0 : num_frames;
  fx->gains_offset = gains_at;
  HIP_TRY(hipEventRecord(fx->ev0, s));
  hipError_t e = (hipError_t)skx_launch_render(&a, n_wg, s);
  if (e != hipSuccess) return skred_amd_set_error(SKRED_E_NO_DEVICE, "fx render launch -> %s", hipGetErrorString(e));
  HIP_TRY(hipEventRecord(fx->ev1, s));
  fx->timed = 1;
  fx->count += (uint64_t)num_frames;
  return SKRED_OK;
}

int skred_fxbank_render(skred_fxbank_t *fx, int num_frames, int interp, int64_t *d_mix, int32_t *d_stems, void *stream) {
  if (!fx || !d_mix || num_frames <= 0) return skred_amd_set_error(SKRED_E_BAD_ARG, "fx render: bad arguments");
  return fx_block(fx, num_frames, interp, d_mix, NULL, d_stems, (hipStream_t)stream);
}

int skred_fxbank_render_mix(skred_fxbank_t *fx, int num_frames, int interp, int64_t *d_out, int32_t *d_stems, void *stream) {
  if (!fx || !d_out || num_frames <= 0) return skred_amd_set_error(SKRED_E_BAD_ARG, "fx render_mix: bad arguments");
  return fx_block(fx, num_frames, interp, NULL, d_out, d_stems, (hipStream_t)stream);
}

int skred_fxbank_master(skred_fxbank_t *fx, const int64_t *d_sum, int num_frames, int64_t *d_out, void *stream) {
  if (!fx || !d_sum || !d_out || num_frames <= 0) return skred_amd_set_error(SKRED_E_BAD_ARG, "fx master: bad arguments");
  if (fx->gains_frames != num_frames || !fx->d_partial)
    return skred_amd_set_error(SKRED_E_BAD_ARG, "fx master: no skred_fxbank_render of %d frames precedes it (the render walks the block's gains)", num_frames);
  HIP_TRY(hipSetDevice(fx->device));
  const hipError_t e = (hipError_t)skx_launch_master_apply((const long long *)d_sum, (const int32_t *)(fx->d_partial + fx->gains_offset), (long long *)d_out,
                                                           num_frames, fx->d_gain_state + 1, fx->d_gain_state, (hipStream_t)stream);
  fx->gains_frames = 0;
  if (e != hipSuccess) return skred_amd_set_error(SKRED_E_NO_DEVICE, "fx master launch -> %s", hipGetErrorString(e));
  return SKRED_OK;
}

int skred_fxbank_set_master(skred_fxbank_t *fx, int64_t target_q31, int32_t k_q15, int64_t gain_q31) {
  if (!fx || target_q31 < 0 || target_q31 > 0x7FFFFFFFll || k_q15 < 0 || k_q15 > 32768 || gain_q31 < 0 || gain_q31 > 0x7FFFFFFFll)
    return skred_amd_set_error(SKRED_E_BAD_ARG, "fx set_master: target and gain are Q31 in [0, 2^31), k is Q15 in [0, 32768]");
  HIP_TRY(hipSetDevice(fx->device));
  HIP_TRY(hipDeviceSynchronize());
  fx->master_target_q31 = target_q31;
  fx->master_k_q15 = k_q15;
  fx->gains_frames = 0;
  const long long g = gain_q31;
  HIP_TRY(hipMemcpy(fx->d_gain_state, &g, sizeof(g), hipMemcpyHostToDevice));
  return SKRED_OK;
}

int64_t skred_fxbank_get_master_gain(skred_fxbank_t *fx) {
  if (!fx || hipSetDevice(fx->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return -1;
  long long g = -1;
  if (hipMemcpy(&g, fx->d_gain_state, sizeof(g), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return g;
}

/* note-ons / note-offs on device-resident voices, stamped with the bank's synth_sample_count when they run (the integer image of
 * amp_envelope_trigger / amp_envelope_release, synth.c:383-395; the float path: SKRED_STAMP_TRIGGER / _RELEASE) */
int skred_fxbank_stamp(skred_fxbank_t *fx, const int32_t *voices, int n, int which, void *stream) {
  if (!fx || n < 0 || (n > 0 && !voices) || !(which & 3) || (which & ~3)) return skred_amd_set_error(SKRED_E_BAD_ARG, "fx stamp: bad arguments");
  if (n == 0) return SKRED_OK;
  for (int i = 0; i < n; i++)
    if (voices[i] < 0 || voices[i] >= fx->n_voices) return skred_amd_set_error(SKRED_E_RANGE, "fx stamp: voice %d outside the bank", voices[i]);
  HIP_TRY(hipSetDevice(fx->device));
  const size_t bytes = (size_t)n * sizeof(int32_t);
  if (bytes > fx->ids_cap) {
    HIP_TRY(hipDeviceSynchronize());
    if (fx->d_ids) { (void)hipFree(fx->d_ids); fx->d_ids = NULL; }
    if (fx->h_ids) { (void)hipHostFree(fx->h_ids); fx->h_ids = NULL; }
    size_t cap = 4096;
    while (cap < bytes) cap *= 2;
    HIP_TRY(hipMalloc((void **)&fx->d_ids, cap));
    HIP_TRY(hipHostMalloc((void **)&fx->h_ids, cap, hipHostMallocDefault));
    fx->ids_cap = cap;
  } else {
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));   /* (one staging buffer: the previous batch must have been read) */
  }
  memcpy(fx->h_ids, voices, bytes);
  HIP_TRY(hipMemcpyAsync(fx->d_ids, fx->h_ids, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
  const hipError_t e = (hipError_t)skx_launch_stamp(fx->d_ids, n, which, fx->d_ro[SKX_TIME], fx->d_rw[0], fx->count, (hipStream_t)stream);
  if (e != hipSuccess) return skred_amd_set_error(SKRED_E_NO_DEVICE, "fx stamp launch -> %s", hipGetErrorString(e));
  return SKRED_OK;
}

int skred_fxbank_render_host(skred_fxbank_t *fx, int num_frames, int interp, int64_t *mix, int32_t *stems) {
  if (!fx || !mix || num_frames <= 0) return skred_amd_set_error(SKRED_E_BAD_ARG, "fx render_host: bad arguments");
  HIP_TRY(hipSetDevice(fx->device));
  int rc;
  const size_t mix_bytes = (size_t)num_frames * 2 * sizeof(int64_t);
  const size_t stem_bytes = (size_t)num_frames * (size_t)fx->n_voices * 2 * sizeof(int32_t);
  if ((rc = grow_bytes((void **)&fx->d_mix, &fx->mix_cap, mix_bytes))) return rc;
  if (stems && (rc = grow_bytes((void **)&fx->d_stems, &fx->stems_cap, stem_bytes))) return rc;
  if ((rc = skred_fxbank_render(fx, num_frames, interp, (int64_t *)fx->d_mix, stems ? fx->d_stems : NULL, NULL))) return rc;
  HIP_TRY(hipMemcpy(mix, fx->d_mix, mix_bytes, hipMemcpyDeviceToHost));
  if (stems) HIP_TRY(hipMemcpy(stems, fx->d_stems, stem_bytes, hipMemcpyDeviceToHost));
  return SKRED_OK;
}

float skred_fxbank_last_render_ms(skred_fxbank_t *fx) {
  if (!fx || !fx->timed) return -1.0f;
  float ms = -1.0f;
  if (hipSetDevice(fx->device) != hipSuccess) return -1.0f;
  if (hipEventSynchronize(fx->ev1) != hipSuccess) return -1.0f;
  if (hipEventElapsedTime(&ms, fx->ev0, fx->ev1) != hipSuccess) return -1.0f;
  return ms;
}
