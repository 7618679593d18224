/*
 * skred_bank.c -- C host shim behind include/skred_amd.h (bank mode).
 *
 * Plain C on purpose (the reference's host code is C, north_star: "Host code stays in C and
 * reaches the HIP kernels through a thin C-ABI shim").  It owns the HBM copy of a voice bank,
 * packs the reference-named host arrays (synth.def:12-89) into the 16-byte device planes of
 * skred_device_layout.h, and sequences the kernels of skred_render_*.hip and
 * skred_mix_kernels.hip (through skred_launch.h).  There is no CPU rendering here: every failure
 * to reach the GPU is reported, never papered over.
 */

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "skred_bank_priv.h"

static __thread char g_err[512];

const char *skred_amd_last_error(void) { return g_err; }

/* every translation unit of the library reports through this (skred_bank_priv.h: fail) */
int skred_amd_set_error(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
int skred_amd_abi_version(void) { return SKRED_AMD_ABI_VERSION; }

int skred_amd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

/* ------------------------------------------------------------------ create / destroy */

static int grow(float **buf, size_t *cap, size_t need) {
  if (*cap >= need) return SKRED_OK;
  if (*buf) { hipFree(*buf); *buf = NULL; *cap = 0; }
  HIP_TRY(hipMalloc((void **)buf, need * sizeof(float)));
  *cap = need;
  return SKRED_OK;
}

/* everything after the calloc: on any failure the caller destroys the partly built bank (skred_bank_destroy
 * tolerates one), so neither the struct nor the HBM already allocated leaks */
static int bank_build(skred_bank_t *b) {
  const size_t plane_bytes = (size_t)b->n_padded * sizeof(sk_plane_t);
  /* one slab, planes back to back (read-only planes first): a window of voices across all planes is one pitched copy */
  HIP_TRY(hipMalloc((void **)&b->d_planes, (size_t)(SKP_COUNT + SKS_COUNT) * plane_bytes));
  HIP_TRY(hipMemset(b->d_planes, 0, (size_t)(SKP_COUNT + SKS_COUNT) * plane_bytes));
  for (int p = 0; p < SKP_COUNT; p++) b->d_ro[p] = b->d_planes + (size_t)p * (size_t)b->n_padded;
  for (int p = 0; p < SKS_COUNT; p++) b->d_rw[p] = b->d_planes + (size_t)(SKP_COUNT + p) * (size_t)b->n_padded;
  /* every slot starts inert (skipped by the kernel) until a voice is uploaded into it */
  sk_plane_t *inert = (sk_plane_t *)calloc((size_t)b->n_padded, sizeof(sk_plane_t));
  if (!inert) return fail(SKRED_E_NO_MEM, "calloc");
  for (int v = 0; v < b->n_padded; v++) { inert[v].w[1] = 1; inert[v].w[2] = SKF_INERT; }   /* table_size 1: fetch stays in bounds */
  hipError_t e = hipMemcpy(b->d_ro[SKP_TAB], inert, plane_bytes, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    const float one = 1.0f;                          /* loop window [0,1): phase 0 + inc 0 stays in range */
    for (int v = 0; v < b->n_padded; v++) { memset(&inert[v], 0, sizeof(inert[v])); memcpy(&inert[v].w[2], &one, 4); }
    e = hipMemcpy(b->d_ro[SKP_OSC], inert, plane_bytes, hipMemcpyHostToDevice);
  }
  free(inert);
  HIP_TRY(e);
  b->h_class = (uint16_t *)calloc((size_t)b->n_padded, sizeof(uint16_t));
  b->h_mod = (int8_t *)malloc((size_t)b->n_padded * 4);
  b->h_level = (int *)calloc((size_t)b->n_padded, sizeof(int));
  if (!b->h_class || !b->h_mod || !b->h_level) return fail(SKRED_E_NO_MEM, "calloc");
  memset(b->h_mod, -1, (size_t)b->n_padded * 4);
  {
    const size_t g64 = (size_t)b->n_padded / 64;
    b->h_pack_mask = (uint64_t *)calloc(g64, sizeof(uint64_t));
    b->h_pack_dirty = (uint8_t *)calloc(g64, 1);
    if (!b->h_pack_mask || !b->h_pack_dirty) return fail(SKRED_E_NO_MEM, "calloc");
    b->pack_hist[0] = (int)g64;
    HIP_TRY(hipMalloc((void **)&b->d_pack_mask, g64 * sizeof(uint64_t)));
    HIP_TRY(hipMemset(b->d_pack_mask, 0, g64 * sizeof(uint64_t)));
  }
  HIP_TRY(hipMalloc((void **)&b->d_level, (size_t)b->n_padded * sizeof(int)));
  HIP_TRY(hipMemset(b->d_level, 0, (size_t)b->n_padded * sizeof(int)));
  /* per 128-voice wave slice of the two-per-lane kernels: listed voices; one more slot: the one-voice family's ticket */
  HIP_TRY(hipMalloc((void **)&b->d_group_flag, (size_t)(b->n_groups * 2 + 1) * sizeof(int32_t)));
  HIP_TRY(hipMemset(b->d_group_flag, 0, (size_t)(b->n_groups * 2 + 1) * sizeof(int32_t)));
  HIP_TRY(hipMalloc((void **)&b->d_env_list, (size_t)b->n_groups * SK_GROUP * sizeof(int32_t)));
  HIP_TRY(hipMemset(b->d_env_list, 0, (size_t)b->n_groups * SK_GROUP * sizeof(int32_t)));
  HIP_TRY(hipMalloc((void **)&b->d_env_off, (size_t)(b->n_groups * 2 + 1) * sizeof(int32_t)));
  HIP_TRY(hipMemset(b->d_env_off, 0, (size_t)(b->n_groups * 2 + 1) * sizeof(int32_t)));
  /* the motion list, double-buffered (a bit per voice), and behind it the violation counter and sk_gain_kernel's two counters
   * (skred_device_layout.h: env_count): one allocation */
  {
    const size_t words = (size_t)b->n_groups * 4;
    HIP_TRY(hipMalloc((void **)&b->d_mask[0], (2 * words + 2) * sizeof(uint64_t)));   /* + violations, env_count[0], env_count[1], pad */
    HIP_TRY(hipMemset(b->d_mask[0], 0, (2 * words + 2) * sizeof(uint64_t)));
    b->d_mask[1] = b->d_mask[0] + words;
    b->d_violations = (uint32_t *)(b->d_mask[0] + 2 * words);
    b->mask_dirty = 1;
  }
  HIP_TRY(hipStreamCreateWithPriority(&b->side, hipStreamNonBlocking, -1));   /* (the envelope kernel's few workgroups should not queue behind a full machine) */
  HIP_TRY(hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&b->ev_join, hipEventDisableTiming));
  HIP_TRY(hipMalloc((void **)&b->d_gain_state, 4 * sizeof(float)));
  HIP_TRY(hipMemset(b->d_gain_state, 0, 4 * sizeof(float)));
  /* arrival tickets of the in-kernel mix-down: zero once, every last arriver re-arms its own */
  /* slabs, final, the envelope kernel's own; behind them one "an envelope moved" word per workgroup row (one-voice family) */
  HIP_TRY(hipMalloc((void **)&b->d_tickets, (SK_FINISH_SLABS + 2 + SK_MAX_WORKGROUPS) * sizeof(uint32_t)));
  HIP_TRY(hipMemset(b->d_tickets, 0, (SK_FINISH_SLABS + 2 + SK_MAX_WORKGROUPS) * sizeof(uint32_t)));
  for (int i = 0; i < SK_TIMING_RING; i++) {
    HIP_TRY(hipEventCreate(&b->ev0[i]));
    HIP_TRY(hipEventCreate(&b->ev1[i]));
  }
  return SKRED_OK;
}

int skred_bank_create(int device, int n_voices, skred_bank_t **out) {
  if (!out || n_voices <= 0) return fail(SKRED_E_BAD_ARG, "skred_bank_create: bad arguments");
  *out = NULL;
  /* voice, slice and list indices inside the kernels are 32-bit ints (scaled by up to 128 before they are widened): the
   * documented ceiling keeps every such product below 2^31 with room to spare; tests/test_gpu_parity.py renders a bank of
   * exactly this size */
  if (n_voices > SKRED_MAX_VOICES)
    return fail(SKRED_E_RANGE, "skred_bank_create: %d voices exceed SKRED_MAX_VOICES (%d)", n_voices, SKRED_MAX_VOICES);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(SKRED_E_NO_DEVICE, "no HIP device visible (this library has no CPU path)");
  if (device < 0 || device >= ndev) return fail(SKRED_E_BAD_ARG, "device %d of %d", device, ndev);
  HIP_TRY(hipSetDevice(device));
  skred_bank_t *b = (skred_bank_t *)calloc(1, sizeof(*b));
  if (!b) return fail(SKRED_E_NO_MEM, "calloc");
  b->device = device;
  if (hipDeviceGetAttribute(&b->n_cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || b->n_cus <= 0) b->n_cus = 256;
  b->n_voices = n_voices;
  b->n_groups = ((n_voices + 4 * SK_GROUP - 1) / (4 * SK_GROUP)) * 4;   /* multiple of 4: the two-per-lane kernel takes up to 1024 voices per pass */
  b->fast2_min_voices = SK_FAST2_MIN_VOICES;
  b->fm2_min_voices = SK_FM2_MIN_VOICES;
  b->in_place_mode = 1;
  b->split_mode = 0;
  b->pack_mode = 1;
  b->fm_skew = 1;
  b->timing_every = 1;
  b->pp_parity = -1;
  b->n_padded = b->n_groups * SK_GROUP;
  b->class_dirty = 1;
  b->mod_dirty = 1;
  /* synth.c:85-92 defaults: volume_user 1 * AMY_FACTOR, LCG seeded with 1 (synth.c:508) */
  b->g.synth_sample_count = 0;
  b->g.noise_rng = 1;
  b->g.volume_final = 0.025f;
  b->g.volume_smoother_gain = 0.0f;
  b->g.volume_smoother_smoothing = 0.002f;
  const int rc = bank_build(b);
  if (rc) { skred_bank_destroy(b); return rc; }      /* (the error text set by the failing step survives: destroy reports nothing) */
  *out = b;
  return SKRED_OK;
}

void skred_bank_destroy(skred_bank_t *b) {
  if (!b) return;
  hipSetDevice(b->device);
  if (b->d_planes) hipFree(b->d_planes);
  if (b->d_tables) hipFree(b->d_tables);
  free(b->h_tables);
  if (b->d_partial) hipFree(b->d_partial);
  if (b->d_tickets) hipFree(b->d_tickets);
  if (b->d_gain_state) hipFree(b->d_gain_state);
  if (b->d_pp_gains) hipFree(b->d_pp_gains);
  if (b->d_probe_ids) hipFree(b->d_probe_ids);
  if (b->d_out) hipFree(b->d_out);
  if (b->d_stems) hipFree(b->d_stems);
  free(b->h_class); free(b->h_mod); free(b->h_level);
  free(b->h_pack_mask); free(b->h_pack_dirty);
  if (b->d_pack_mask) hipFree(b->d_pack_mask);
  sk_queue_free(b);
  sk_patterns_free(b);
  for (int i = 0; i < SK_UPD_RING; i++) {
    if (b->upd[i].d) hipFree(b->upd[i].d);
    if (b->upd[i].h) hipHostFree(b->upd[i].h);
  }
  if (b->h_upd_done) hipHostFree((void *)b->h_upd_done);
  if (b->d_upd_cnt) hipFree(b->d_upd_cnt);
  free(b->upd_mark);
  if (b->h_report) hipHostFree((void *)b->h_report);
  if (b->d_level) hipFree(b->d_level);
  if (b->d_group_flag) hipFree(b->d_group_flag);
  if (b->d_env_list) hipFree(b->d_env_list);
  if (b->d_env_gain) hipFree(b->d_env_gain);
  if (b->d_env_off) hipFree(b->d_env_off);
  if (b->d_mask[0]) hipFree(b->d_mask[0]);
  if (b->side) hipStreamDestroy(b->side);
  if (b->ev_fork) hipEventDestroy(b->ev_fork);
  if (b->ev_join) hipEventDestroy(b->ev_join);
  for (int i = 0; i < SK_TIMING_RING; i++) {
    if (b->ev0[i]) hipEventDestroy(b->ev0[i]);
    if (b->ev1[i]) hipEventDestroy(b->ev1[i]);
  }
  free(b);
}

int skred_bank_n_voices(const skred_bank_t *b) { return b ? b->n_voices : 0; }

/* diagnostic builds only (tools/ab_split.py --stamps; not part of the ABI): the words a -DSKS_STAMPS build of
 * sk_render_split_kernel leaves in the bank's list buffer */
int sk_debug_env_list(skred_bank_t *b, int32_t *dst, int n) {
  if (!b || !dst || n <= 0 || n > b->n_groups * SK_GROUP) return SKRED_E_BAD_ARG;
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(dst, b->d_env_list, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
  return SKRED_OK;
}

/* ------------------------------------------------------------------ tables */

int skred_bank_set_tables_f32(skred_bank_t *b, const float *pool, size_t n_floats) {
  if (!b || !pool || n_floats == 0) return fail(SKRED_E_BAD_ARG, "set_tables: bad arguments");
  if (n_floats > 0x7FFFFFF0u) return fail(SKRED_E_RANGE, "table pool too large");
  HIP_TRY(hipSetDevice(b->device));
  if (b->d_tables) { hipFree(b->d_tables); b->d_tables = NULL; }
  b->table_floats = n_floats;
  b->table_floats_padded = (n_floats + SK_TABLE_PAD + 3) & ~(size_t)3;   /* readable past the last table: see SK_TABLE_PAD */
  HIP_TRY(hipMalloc((void **)&b->d_tables, b->table_floats_padded * sizeof(float)));
  HIP_TRY(hipMemset(b->d_tables, 0, b->table_floats_padded * sizeof(float)));
  HIP_TRY(hipMemcpy(b->d_tables, pool, n_floats * sizeof(float), hipMemcpyHostToDevice));
  free(b->h_tables);
  b->h_tables = (float *)malloc(n_floats * sizeof(float));
  if (!b->h_tables) return fail(SKRED_E_NO_MEM, "set_tables: host copy of the pool");
  memcpy(b->h_tables, pool, n_floats * sizeof(float));
  b->tables_epoch++;                    /* voices packed against the old pool carry its guard flags: see render_block */
  return SKRED_OK;
}

/* ------------------------------------------------------------------ pack / unpack */

static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

int skred_bank_upload(skred_bank_t *b, const skred_voice_bank_t *h, int src_first, int dst_first, int count) {
  if (!b || !h || count < 0) return fail(SKRED_E_BAD_ARG, "upload: bad arguments");
  if (src_first < 0 || src_first + count > h->n_voices || dst_first < 0 || dst_first + count > b->n_voices)
    return fail(SKRED_E_RANGE, "upload window [%d,+%d) -> [%d,+%d) outside bank", src_first, count, dst_first, count);
  if (count == 0) return SKRED_OK;
  HIP_TRY(hipSetDevice(b->device));
  /* synchronous by contract: the planes are overwritten with a blocking copy, and a render still running on a
   * non-blocking stream (the caller's, or the bank's own tail stream) is not ordered against that by itself */
  HIP_TRY(hipDeviceSynchronize());
  const int NP = SKP_COUNT + SKS_COUNT;
  sk_plane_t *st = (sk_plane_t *)calloc((size_t)NP * (size_t)count, sizeof(sk_plane_t));
  sk_voice_meta_t *meta = (sk_voice_meta_t *)malloc((size_t)count * sizeof(sk_voice_meta_t));
  if (!st || !meta) { free(st); free(meta); return fail(SKRED_E_NO_MEM, "upload staging"); }
  for (int i = 0; i < count; i++) {
    sk_plane_t ro[SKP_COUNT], rw[SKS_COUNT];
    const int rc = sk_pack_voice(b, h, src_first + i, dst_first + i, 1, ro, rw, &meta[i]);
    if (rc) { free(st); free(meta); return rc; }
    for (int p = 0; p < SKP_COUNT; p++) st[(size_t)p * count + i] = ro[p];
    for (int p = 0; p < SKS_COUNT; p++) st[(size_t)(SKP_COUNT + p) * count + i] = rw[p];
  }
  const size_t bytes = (size_t)count * sizeof(sk_plane_t);
  /* all planes of the window in one pitched copy (rows = planes) */
  const hipError_t e = hipMemcpy2D(b->d_planes + dst_first, (size_t)b->n_padded * sizeof(sk_plane_t), st, bytes, bytes, (size_t)NP,
                                   hipMemcpyHostToDevice);
  free(st);
  if (e != hipSuccess) { free(meta); HIP_TRY(e); }
  if (dst_first == 0 && count == b->n_voices) { b->features = 0; b->guard_epoch = b->tables_epoch; }   /* whole bank replaced */
  for (int i = 0; i < count; i++) sk_apply_meta(b, dst_first + i, &meta[i], 1);
  free(meta);
  sk_control_changed(b);
  b->mask_dirty = 1;                    /* the motion list is rebuilt from the planes before the next two-per-lane block */
  b->pack_zero = 1;                     /* (uploaded state may hold a voice_sample on a voice that is skipped) */
  return SKRED_OK;
}

/* Packed lanes: the words of the groups whose voices changed, and the histogram that sizes the lane slots.  A group's word
 * holds the voices that can sound (SKC_LIVE) and the voices those name as modulators (they keep a lane so that the exchange of
 * the one-voice kernel finds them; at run time they are skipped like any dead voice).  Returns the most lanes a group needs. */
static int pack_refresh(skred_bank_t *b) {
  if (b->pack_any_dirty) {
    const int g64 = b->n_padded / 64;
    for (int g = 0; g < g64; g++) {
      if (!b->h_pack_dirty[g]) continue;
      b->h_pack_dirty[g] = 0;
      uint64_t w = 0;
      for (int l = 0; l < 64; l++) {
        const int v = g * 64 + l;
        if (!(b->h_class[v] & SKC_LIVE)) continue;
        w |= (uint64_t)1 << l;
        for (int k = 0; k < 4; k++) {
          const int m = b->h_mod[(size_t)k * b->n_padded + v];
          if (m >= 0) w |= (uint64_t)1 << m;
        }
      }
      const uint64_t old = b->h_pack_mask[g];
      if (w == old) continue;
      b->pack_hist[__builtin_popcountll(old)]--;
      b->pack_hist[__builtin_popcountll(w)]++;
      b->h_pack_mask[g] = w;
      b->pack_upload = 1;
      if (old & ~w) b->pack_zero = 1;     /* a voice lost its lane: the reference would clear its voice_sample on the next frame */
    }
    b->pack_any_dirty = 0;
  }
  int most = 64;
  while (most > 0 && b->pack_hist[most] == 0) most--;
  return most;
}

/* Pick the kernel.  The fast kernel (skred_render_fast.hip: sk_render_fast_kernel) is valid when, over
 * all voices that can sound: none is "exotic" (stopping one-shot, reverse, sample&hold, bit-crush,
 * noise, modulated, smoother off, non-finite phase data), and the biquad / the envelope are each used
 * by all of them or by none.  Anything else runs the generic kernel; both give identical samples. */
static int classify(skred_bank_t *b) {
  if (!b->class_dirty) return SKRED_OK;
  const int real = b->cnt_real, filt = b->cnt_filter, env = b->cnt_env, exotic = b->cnt_exotic;
  uint32_t m = 0;
  if (real > 0 && !exotic) {
    m = SKM_FAST;
    if (filt) m |= SKM_FILTER_ALL;
    if (env) m |= SKM_ENV_ALL;
    if ((filt && filt != real) || (env && env != real)) m |= SKM_MIXED;   /* some voices only: per-lane flags */
    if (b->cnt_stops) m |= SKM_STOPS;
    if (b->cnt_fm) m |= SKM_FM;
    if (b->cnt_fm && !b->cnt_fm_odd && !b->cnt_stops) m |= SKM_FM_PAIR;   /* every carrier: an even voice modulated by the next one */
    if ((m & SKM_FM_PAIR) && b->cnt_pair_ap) m |= SKM_PAIR_AP;
  }
  b->fast_mode = m;
  b->class_dirty = 0;
  /* dependency levels for modulated banks (skred_render_generic.hip: sk_render_mod_kernel) */
  if (!b->mod_dirty) return SKRED_OK;
  b->mod_dirty = 0;
  b->max_level = 0;
  if (b->features & (SKB_ANY_MOD | SKB_ANY_FM)) {
    for (int g0 = 0; g0 < b->n_padded; g0 += 64) {
      for (int l = 0; l < 64; l++) {
        int lvl = 0;
        for (int k = 0; k < 4; k++) {
          const int src = b->h_mod[(size_t)k * b->n_padded + g0 + l];
          if (src >= 0 && src < l && b->h_level[g0 + src] + 1 > lvl) lvl = b->h_level[g0 + src] + 1;
        }
        b->h_level[g0 + l] = lvl;
        if (lvl > b->max_level) b->max_level = lvl;
      }
    }
    const hipError_t e = hipMemcpy(b->d_level, b->h_level, (size_t)b->n_padded * sizeof(int), hipMemcpyHostToDevice);
    if (e != hipSuccess) { b->mod_dirty = 1; return fail(SKRED_E_NO_DEVICE, "dependency levels -> %s", hipGetErrorString(e)); }
  }
  return SKRED_OK;
}

int skred_bank_set_option(skred_bank_t *b, int option, int value) {
  if (!b) return fail(SKRED_E_BAD_ARG, "set_option");
  switch (option) {
    case SKRED_OPT_FORCE_GENERIC: b->force_generic = value != 0; return SKRED_OK;
    case SKRED_OPT_FAST2_MIN_VOICES: b->fast2_min_voices = value; b->fast2_min_user = 1; return SKRED_OK;
    case SKRED_OPT_FM2_MIN_VOICES: b->fm2_min_voices = value; return SKRED_OK;
    case SKRED_OPT_IN_PLACE: b->in_place_mode = value < 0 ? 0 : value > 2 ? 2 : value; return SKRED_OK;
    case SKRED_OPT_KERNEL_TIMING: b->timing_every = value < 0 ? 0 : value; return SKRED_OK;
    case SKRED_OPT_SPLIT_PAIRS: b->split_pairs = (value == 2 || value == 4) ? value : 0; return SKRED_OK;
    case SKRED_OPT_SPLIT: b->split_mode = value < 0 ? 0 : value > 3 ? 3 : value; return SKRED_OK;
    case SKRED_OPT_PACK: b->pack_mode = value < 0 ? 0 : value > 2 ? 2 : value; return SKRED_OK;
    case SKRED_OPT_FM_SKEW: b->fm_skew = value != 0; return SKRED_OK;
    default: return fail(SKRED_E_BAD_ARG, "unknown option %d", option);
  }
}

int skred_bank_set_probe(skred_bank_t *b, const int32_t *voices, int n, float *d_probe) {
  if (!b || n < 0 || n > SK_PROBE_MAX || (n > 0 && (!voices || !d_probe))) return fail(SKRED_E_BAD_ARG, "set_probe: bad arguments");
  for (int i = 0; i < n; i++)
    if (voices[i] < 0 || voices[i] >= b->n_voices) return fail(SKRED_E_RANGE, "set_probe: voice %d outside the bank", voices[i]);
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipDeviceSynchronize());                   /* (a block in flight may still read the old list) */
  if (n > 0) {
    if (!b->d_probe_ids) HIP_TRY(hipMalloc((void **)&b->d_probe_ids, SK_PROBE_MAX * sizeof(int32_t)));
    HIP_TRY(hipMemcpy(b->d_probe_ids, voices, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice));
  }
  b->n_probe = n;
  b->d_probe_out = n > 0 ? d_probe : NULL;
  return SKRED_OK;
}

int skred_bank_last_kernel(const skred_bank_t *b) { return b ? b->last_kernel : -1; }
int skred_bank_last_in_place(const skred_bank_t *b) { return b ? b->last_in_place : 0; }
int skred_bank_last_split(const skred_bank_t *b) { return b ? b->last_split : 0; }
int skred_bank_last_pack(const skred_bank_t *b) { return b ? b->last_pack : 0; }
unsigned skred_bank_list_violations(const skred_bank_t *b) { return b ? b->violations_seen : 0u; }

int skred_bank_download(skred_bank_t *b, skred_voice_bank_t *h, int src_first, int dst_first, int count) {
  if (!b || !h || count < 0) return fail(SKRED_E_BAD_ARG, "download: bad arguments");
  if (src_first < 0 || src_first + count > b->n_voices || dst_first < 0 || dst_first + count > h->n_voices)
    return fail(SKRED_E_RANGE, "download window outside bank");
  if (count == 0) return SKRED_OK;
  HIP_TRY(hipSetDevice(b->device));
  sk_plane_t *st = (sk_plane_t *)malloc((size_t)SKS_COUNT * (size_t)count * sizeof(sk_plane_t));
  if (!st) return fail(SKRED_E_NO_MEM, "download staging");
  HIP_TRY(hipDeviceSynchronize());
  const size_t bytes = (size_t)count * sizeof(sk_plane_t);
  const hipError_t e = hipMemcpy2D(st, bytes, b->d_rw[0] + src_first, (size_t)b->n_padded * sizeof(sk_plane_t), bytes, (size_t)SKS_COUNT,
                                   hipMemcpyDeviceToHost);      /* the read-write planes of the window: one pitched copy */
  if (e != hipSuccess) { free(st); HIP_TRY(e); }
  for (int i = 0; i < count; i++) {
    const int v = dst_first + i;
    const sk_plane_t *s0 = &st[(size_t)SKS_OSC * count + i];
    const sk_plane_t *s1 = &st[(size_t)SKS_FILT * count + i];
    const sk_plane_t *s2 = &st[(size_t)SKS_MISC * count + i];
    h->voice_phase[v] = u2f(s0->w[0]);
    h->voice_smoother_gain[v] = u2f(s0->w[1]);
    h->voice_filter[v].x1 = u2f(s0->w[2]); h->voice_filter[v].x2 = u2f(s0->w[3]);
    h->voice_filter[v].y1 = u2f(s1->w[0]); h->voice_filter[v].y2 = u2f(s1->w[1]);
    h->voice_sample[v] = u2f(s1->w[2]);
    h->voice_finished[v] = (s1->w[3] & SKR_FINISHED) ? 1 : 0;
    h->voice_amp_envelope[v].is_active = (s1->w[3] & SKR_ENV_ACTIVE) ? 1 : 0;
    h->voice_sample_hold[v] = u2f(s2->w[0]);
    h->voice_sample_hold_count[v] = (int32_t)s2->w[1];
    h->voice_pan_left[v] = u2f(s2->w[2]);
    h->voice_pan_right[v] = u2f(s2->w[3]);
  }
  free(st);
  return SKRED_OK;
}

/* ------------------------------------------------------------------ globals */

int skred_bank_set_globals(skred_bank_t *b, const skred_globals_t *g) {
  if (!b || !g) return fail(SKRED_E_BAD_ARG, "set_globals");
  HIP_TRY(hipSetDevice(b->device));
  /* the carried master gain lives on the device and a block's tail may still be writing it (on the caller's
   * non-blocking stream): wait for the device first */
  HIP_TRY(hipDeviceSynchronize());
  b->g = *g;
  b->gains_frames = 0;                /* gains prepared for a pending skred_bank_master no longer hold */
  HIP_TRY(hipMemcpy(b->d_gain_state, &g->volume_smoother_gain, sizeof(float), hipMemcpyHostToDevice));
  sk_control_changed(b);              /* the clock may have moved: envelope stages are a function of it */
  b->mask_dirty = 1;
  return SKRED_OK;
}

int skred_bank_get_globals(skred_bank_t *b, skred_globals_t *g) {
  if (!b || !g) return fail(SKRED_E_BAD_ARG, "get_globals");
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(&b->g.volume_smoother_gain, b->d_gain_state, sizeof(float), hipMemcpyDeviceToHost));
  *g = b->g;
  return SKRED_OK;
}

/* ------------------------------------------------------------------ render */

/* What earlier launches found, as far as their answers have arrived (never waits): the block's final arriver stores
 * (launch ticket << 32 | finding) into two pinned host words (skred_kernel_common.hpp: sk_final_cols). */
static void poll_reports(skred_bank_t *b) {
  if (!b->h_report) return;
  const uint64_t w0 = __atomic_load_n(&b->h_report[0], __ATOMIC_RELAXED), w1 = __atomic_load_n(&b->h_report[1], __ATOMIC_RELAXED);
  const uint32_t t0 = (uint32_t)(w0 >> 32);
  if (t0 != 0 && t0 != b->report_seen && t0 == (uint32_t)(w1 >> 32)) {     /* a new report, both words of the same launch */
    b->report_seen = t0;
    const int slot = (int)(t0 % SK_REPORT_RING);
    if (b->report_ticket[slot] == t0) {                                    /* (else: asked so long ago that its slot was re-used) */
      const int fresh = b->report_epoch[slot] == b->control_epoch;         /* no control action reached the bank since it was issued */
      const uint32_t found = (uint32_t)w0;
      if (b->report_kind[slot] == 1) {
        /* one-voice family: did an envelope move in that launch */
        if (!found && fresh) b->env_quiet = 1;
        if (found) b->env_quiet = 0;
      } else {
        /* two-per-lane family: the length of the list that block rendered.  Empty, and nothing added since: every later list is
         * empty too (a list is the survivors of the one before plus what control actions add) -- a structural fact, not an
         * inference about envelopes.  And the cross-check counter of sk_render_fast2_kernel: should it ever move, rebuild. */
        if (b->report_kind[slot] == 2 && found == 0 && fresh) b->list_empty = 1;
        if (b->report_kind[slot] == 2 && (int32_t)(t0 - b->bound_min_ticket) >= 0) {   /* (not a list from before the last rebuild) */
          b->bound_len = found;
          b->bound_touched = b->report_touched[slot];
          b->bound_valid = 1;
        }
        if ((uint32_t)w1 != b->violations_seen) {
          b->violations_seen = (uint32_t)w1;
          b->mask_dirty = 1;
          b->list_empty = 0;
          (void)fail(SKRED_E_UNSUPPORTED, "launch %u or one before it rendered a voice at a constant level whose envelope was in motion (not on the motion list): list rebuilt", t0);
        }
      }
    }
  }
}

/* this launch will report: remember what its ticket means (kind 1: one-voice "moved"; 2: list length; 3: violations only) */
static int expect_report(skred_bank_t *b, sk_render_args_t *a, int kind) {
  if (!b->h_report) {
    HIP_TRY(hipHostMalloc((void **)&b->h_report, 2 * sizeof(uint64_t), hipHostMallocCoherent));   /* (polled by the host while kernels run) */
    b->h_report[0] = b->h_report[1] = 0;
  }
  const int slot = (int)(a->launch_ticket % SK_REPORT_RING);
  b->report_ticket[slot] = a->launch_ticket;
  b->report_epoch[slot] = b->control_epoch;
  b->report_kind[slot] = (uint8_t)kind;
  b->report_touched[slot] = b->touched_total;
  a->report = (unsigned long long *)b->h_report;
  return SKRED_OK;
}

/* One block: picks and launches the render kernel(s), whose last-arriving workgroups also add the per-workgroup rows
 * up (skred_kernel_common.hpp: sk_finish_block) into `d_sum` (pre-master, may be NULL) and / or, scaled by the master
 * gain of each frame, into `d_out`; advances the timeline. */
static int render_block(skred_bank_t *b, int num_frames, int interp, float *d_stems, float *d_sum, float *d_out,
                        int num_channels, hipStream_t s) {
  if (interp != SKRED_INTERP_TRUNCATE && interp != SKRED_INTERP_LINEAR) return fail(SKRED_E_BAD_ARG, "render: interp %d", interp);
  if (!b->d_tables) return fail(SKRED_E_BAD_ARG, "render: no table pool set");
  if ((b->features & (SKB_ANY_MOD | SKB_ANY_FM)) && b->cnt_escapes > 0)
    return fail(SKRED_E_UNSUPPORTED, "a voice is modulated by a voice outside its aligned 64-voice group: "
                                     "keep modulator and carrier in the same group (SURVEY 8e)");
  HIP_TRY(hipSetDevice(b->device));
  int rc = classify(b);
  if (rc) return rc;
  poll_reports(b);
  /* the modulated kernel serves every kind of modulation; banks whose only modulation is previous-frame FM stay on
   * the one-per-lane kernel when they are otherwise clean */
  const int fast_ok = (b->fast_mode & SKM_FAST) && !b->force_generic;
  const int modulated = (b->features & SKB_ANY_MOD) != 0 || ((b->features & SKB_ANY_FM) && b->cnt_fm > 0 && !fast_ok);
  int n_wg = b->n_groups < SK_MAX_WORKGROUPS ? b->n_groups : SK_MAX_WORKGROUPS;   /* workgroups stride over 256-voice passes */

  sk_render_args_t a;
  memset(&a, 0, sizeof(a));
  for (int p = 0; p < SKP_COUNT; p++) a.ro[p] = b->d_ro[p];
  for (int p = 0; p < SKS_COUNT; p++) a.rw[p] = b->d_rw[p];
  a.tables = b->d_tables;
  a.stems = d_stems;
  a.group_flag = b->d_group_flag;
  a.env_list = b->d_env_list;
  a.env_off = b->d_env_off;
  a.mask_cur = b->d_mask[b->mask_p];
  a.mask_next = b->d_mask[b->mask_p ^ 1];
  a.violations = b->d_violations;
  a.count0 = b->g.synth_sample_count;
  a.rng0 = b->g.noise_rng;
  a.n_voices = b->n_voices;
  a.n_groups = b->n_groups;
  a.num_frames = num_frames;
  a.table_floats = (int32_t)b->table_floats;
  a.lds_table_floats = b->table_floats_padded <= SK_LDS_TABLE_MAX_FLOATS ? (int32_t)b->table_floats_padded : 0;
  a.interp = interp;
  a.features = b->features;
  a.fast_mode = b->force_generic ? 0u : b->fast_mode;
  /* two voices per lane pay off for large LDS-table banks (packed fp32); banks whose tables stay in L2 / HBM do
   * better with one voice per lane at every size measured (2^16 .. 2^20: twice the waves to hide the window
   * refills behind) unless the caller set the threshold explicitly */
  if ((a.fast_mode & SKM_FAST) && !(a.fast_mode & (SKM_STOPS | SKM_FM)) && b->n_voices >= b->fast2_min_voices &&
      (a.lds_table_floats > 0 || b->fast2_min_user) && !d_stems)      /* (per-voice stems: the one-voice kernel writes them) */
    a.fast_mode |= SKM_TWO_PER_LANE;        /* (voices that finish mid-launch are handled by the one-per-lane kernel only) */
  /* ... except where its 1024-voice passes fill the machine unevenly: one pass per CU up to n_cus passes, then SOME CUs with two
   * (the block takes as long as a full second layer: 262 144 voices 87 us, 294 912 voices 132 us, 524 288 voices 146 us), where
   * the one-voice kernel's finer grain wins until the second layer is about five eighths full (294 912 voices 106 us, 360 448
   * voices 121 vs 132, 393 216 voices 124 vs 133, 458 752 voices 140 vs 135; tools/measure_banks.py mid -- since the one-voice
   * kernel's LDS-table instantiations stopped reserving a table window per wave they fit four workgroups per CU) */
  if ((a.fast_mode & SKM_TWO_PER_LANE) && !b->fast2_min_user && a.lds_table_floats > 0) {
    const int passes = b->n_groups * 2 / SK_FAST2_NW_LDS;
    /* (only while nothing moves -- with envelopes in motion the two-per-lane kernel and the envelope kernel beside it are
     * ahead at these sizes, 202 vs 214..255 us --: the family that rendered the previous block knows: an empty motion list,
     * or a one-voice launch that saw no envelope move) */
    const int quiet = !(a.fast_mode & SKM_ENV_ALL) || (b->last_family == SKRED_KERNEL_FAST2 ? b->list_empty : b->env_quiet);
    if (quiet && passes > b->n_cus && passes <= b->n_cus + b->n_cus * 5 / 8) a.fast_mode &= ~SKM_TWO_PER_LANE;
  }
  /* ... but while envelopes move the one-voice kernel's block form of them beats the two-per-lane kernel + envelope kernel on
   * mid-size banks (tools/ab_env_mid.py): such banks change kernels with their state (both families read and write the same
   * planes).  "Envelopes move": from a control action until a one-voice launch has reported that none did -- a speed hint. */
  if ((a.fast_mode & SKM_TWO_PER_LANE) && (a.fast_mode & SKM_ENV_ALL) && !(a.fast_mode & SKM_MIXED) && !b->env_quiet &&
      !b->fast2_min_user && b->n_voices < SK_FAST2_MOTION_MIN_VOICES)
    a.fast_mode &= ~SKM_TWO_PER_LANE;
  /* two-operator FM (every carrier an even voice, modulated by the voice after it): carrier and modulator share a lane of
   * the two-per-lane kernel, so the per-frame exchange of the one-per-lane kernel disappears.  LDS-table banks. */
  if (fast_ok && (a.fast_mode & SKM_FM_PAIR) && a.lds_table_floats > 0 && !d_stems && b->n_voices >= b->fm2_min_voices)
    a.fast_mode |= SKM_TWO_PER_LANE;
  else
    a.fast_mode &= ~(SKM_FM_PAIR | SKM_PAIR_AP);
  /* linear lookup on a bank whose every real voice loops over its whole table with a guard sample behind it: the specialised
   * kernels' instantiations without the fold test (two-operator FM banks keep the general form) */
  if (interp == SKRED_INTERP_LINEAR && (a.fast_mode & SKM_FAST) && !(a.fast_mode & SKM_FM_PAIR) && !modulated && b->cnt_real > 0 &&
      b->cnt_guard == b->cnt_real && b->guard_epoch == b->tables_epoch)
    a.interp = 2;
  /* Sparse banks (most voices skipped by the reference's own rule, synth.c:537 -- the shipped patches use 3 to 6 voices of 64):
   * the one-voice family with the lanes PACKED -- a wave takes the voices that can sound of 64 / S aligned 64-voice groups, S = the
   * most lanes any group needs, rounded up to a power of two (skred_device_layout.h: pack_mask).  The extended instantiation
   * renders them (it holds every per-lane feature test), so the rule asks for at least half the waves to disappear; a bank the
   * two-per-lane kernel would take, for a quarter of them. */
  a.pack_shift = 6;
  a.fm_skew = b->fm_skew && (a.fast_mode & SKM_FM) && a.lds_table_floats > 0 && !d_stems;   /* (the launcher drops it when the ring does not fit) */
  if (modulated) a.fm_skew = b->fm_skew && !d_stems;        /* (the modulated kernel: its frame-lag form, same option) */
  int pack_s = 0;
  if (b->pack_mode && (modulated || ((a.fast_mode & SKM_FAST) && !(a.fast_mode & SKM_FM_PAIR))) && !d_stems) {   /* (the modulated kernel packs the same way) */
    const int most = pack_refresh(b);
    int sh = 0;
    while ((1 << sh) < most) sh++;
    /* ... on a bank that fills the machine several times over: up to two 256-voice passes per CU the block's time is one pass's
     * latency whatever the waves hold, and the extended instantiation's is the longer one (131 072 voices, 5 % in use: 54 us
     * packed, 50 not; 2^20 voices: 117 against 228) */
    const int big = b->n_groups >= 3 * b->n_cus;        /* (196 608 voices: 55 us packed, 63 not; 262 144: 55 against 79) */
    if ((big && (1 << sh) <= ((!modulated && (a.fast_mode & SKM_TWO_PER_LANE)) ? 16 : 32)) || (b->pack_mode == 2 && sh < 6)) {
      if (!modulated) a.fast_mode &= ~SKM_TWO_PER_LANE;
      a.pack_shift = sh;
      a.pack_groups = b->n_padded / 64;
      const int per_pass = 4 << (6 - sh);               /* groups per 4-wave workgroup pass */
      a.pack_passes = (a.pack_groups + per_pass - 1) / per_pass;
      a.pack_mask = b->d_pack_mask;
      pack_s = 1 << sh;
    }
  }
  b->last_kernel = !(a.fast_mode & SKM_FAST) ? SKRED_KERNEL_GENERIC
                   : (a.fast_mode & SKM_TWO_PER_LANE) ? SKRED_KERNEL_FAST2 : SKRED_KERNEL_FAST;
  if (modulated) b->last_kernel = SKRED_KERNEL_MODULATED;
  if (!modulated && (a.fast_mode & SKM_TWO_PER_LANE)) {
    /* passes of sk_render_fast2_kernel: 1024 voices each for LDS-table banks, 512 otherwise (skred_render_fast2.hip) */
    const int passes = a.lds_table_floats > 0 ? b->n_groups * 2 / SK_FAST2_NW_LDS : b->n_groups / 2;
    n_wg = passes < SK_MAX_WORKGROUPS ? passes : SK_MAX_WORKGROUPS;
  }
  if (pack_s) n_wg = a.pack_passes < SK_MAX_WORKGROUPS ? a.pack_passes : SK_MAX_WORKGROUPS;
  /* two-per-lane banks with envelopes: the voices on the motion list are rendered by sk_render_env2_kernel BESIDE the steady
   * kernel, on the bank's second stream (its own rows, its own ticket; skred_kernel_common.hpp: sk_finish_env) */
  const int two_env = !modulated && (a.fast_mode & SKM_TWO_PER_LANE) && (a.fast_mode & SKM_ENV_ALL);
  /* one-per-lane banks with envelopes: a launch's report picks between the instantiation that also holds the block form of
   * envelopes in motion and the lean one (skred_render_fast.hip: RAMPK); both render everything, so a stale answer costs
   * speed, never samples */
  const int one_env = !modulated && (a.fast_mode & SKM_FAST) && !(a.fast_mode & SKM_TWO_PER_LANE) && (a.fast_mode & SKM_ENV_ALL);
  /* SKRED_OPT_SPLIT (off by default): the one-voice family on a clean LDS-table bank that is believed steady (no envelope: always;
   * envelopes: a launch has reported that none moved and no control action arrived since) with every frame split between an
   * oscillator wave and a post wave (skred_render_split.hip).  A wave whose voices are not steady after all renders itself on the
   * general path of the same kernel, so the belief decides speed only.  Built on the previous review's advice to give small and
   * mid-size banks more instruction streams per SIMD; measured (tools/ab_split.py, tools/issue_mix.hip, profiles/r04_split_*): the
   * LDS instructions of the hand-over cost a wave about what the moved arithmetic saves, and from two 64-voice groups per SIMD on
   * the SIMD's own throughput binds -- 1.4 % faster than sk_render_fast_kernel at 65 536 voices, 3 % slower at 4 096, 20 % slower at
   * 131 072 -- so the library never picks it by itself; values 1 / 2 / 3 keep it reachable (the rule of value 1: banks of 32 768 ..
   * 65 536 filtered voices on a 256-CU device).  (Decided here, ahead of the row layout: the two-pair form has twice the rows.) */
  int split = 0;
  if (!modulated && (a.fast_mode & SKM_FAST) && !(a.fast_mode & (SKM_TWO_PER_LANE | SKM_STOPS | SKM_FM | SKM_MIXED)) && a.lds_table_floats > 0 &&
      !d_stems && !pack_s && b->split_mode && (!one_env || b->env_quiet || b->split_mode == 3) && sk_split_lds_bytes(&a, 4) <= SK_SPLIT_MAX_LDS) {
    const int per_cu = (int)(SK_SPLIT_MAX_LDS / sk_split_lds_bytes(&a, 4)) >= 2 ? 2 : 1;
    if (b->split_mode >= 2 || ((a.fast_mode & SKM_FILTER_ALL) && b->n_groups * 2 >= b->n_cus && b->n_groups <= b->n_cus)) split = 4;
    (void)per_cu;
    if (split && b->split_pairs && (b->split_pairs == 4 || b->n_groups * 2 <= SK_MAX_WORKGROUPS)) split = b->split_pairs;   /* (tests) */
  }
  if (b->n_probe > 0) {
    /* probes are written by the probe instantiations of the specialised kernels only */
    if (modulated || !(a.fast_mode & SKM_FAST) || (a.fast_mode & SKM_FM_PAIR) || d_stems)
      return fail(SKRED_E_UNSUPPORTED, "a probe is set, but this block would run a kernel without probe instantiations "
                                       "(generic / modulated / two-operator FM pairs, or a launch with the full stem buffer)");
    split = 0;
    a.probe_ids = b->d_probe_ids;
    a.n_probe = b->n_probe;
    a.probe_out = b->d_probe_out;
    HIP_TRY(hipMemsetAsync(b->d_probe_out, 0, (size_t)num_frames * (size_t)b->n_probe * 2 * sizeof(float), s));   /* (a skipped / muted voice writes nothing) */
  }
  if (split) a.fast_mode |= SKM_SPLIT | (split == 2 ? SKM_SPLIT2 : 0u);
  if (split == 2) n_wg = b->n_groups * 2;
  if (two_env) {
    if (b->last_family != SKRED_KERNEL_FAST2) b->mask_dirty = 1;     /* another family rendered meanwhile: it does not keep the list */
    if (b->mask_dirty) {
      const hipError_t ec = (hipError_t)sk_launch_classify(&a, b->d_mask[b->mask_p], s);
      if (ec != hipSuccess) return fail(SKRED_E_NO_DEVICE, "classify launch -> %s", hipGetErrorString(ec));
      b->mask_dirty = 0;
      b->list_empty = 0;
      b->bound_valid = 0;                              /* the rebuilt list's length is not known until this block reports it */
      b->bound_min_ticket = b->launch_ticket + 1;
    }
  }
  /* Sparse lists of LDS-table banks: the listed voices stay in their lanes (skred_gain_kernels.hip ahead of the steady kernel's
   * in-place instantiations, same stream) instead of going through the envelope kernel beside it -- a second kernel costs the
   * steady one a third round of workgroups however few voices it holds (DESIGN "The motion list").  The gain rows are a
   * buffer of fixed capacity, so this path is only taken under a PROVEN bound on the list's length: the length a launch
   * reported (sk_final_cols) plus every voice a control action has touched since that launch was issued -- a list is the
   * survivors of the one before plus what control actions add. */
  int inplace = 0;
  if (two_env && !b->list_empty && b->bound_valid && b->in_place_mode && a.lds_table_floats > 0 && !(a.fast_mode & SKM_FM_PAIR)) {
    const uint64_t bound = (uint64_t)b->bound_len + (b->touched_total - b->bound_touched);
    const size_t stride = (size_t)num_frames + 8;
    /* rows: SK_INPLACE_WORD_ROWS per 64-voice word of the list (handed out without an atomic), then an overflow area for words
     * that hold more -- as large as the bound must be small, so the rows cannot run out */
    const size_t own = (size_t)b->n_groups * 4 * SK_INPLACE_WORD_ROWS;
    const size_t over = (size_t)b->n_voices / (b->in_place_mode == 2 ? SK_INPLACE_DENOM : 64) + 64;   /* (mode 1 never takes lists beyond n / 128) */
    const size_t rows = own + over;
    /* ... and only where it is the faster of the two (tools/ab_inplace.py, MI355X; DESIGN "The motion list"): every wave
     * of the steady kernel that holds a listed voice runs its smoothers and reads gains (~ +30 %), so the list must be sparse;
     * and the envelope kernel beside the steady one is cheap when the steady kernel's last round of workgroups leaves slots
     * free -- it costs a whole extra round when that round is full (2^19, 2^20 voices on 256 CUs) */
    uint64_t limit = over;
    if (b->in_place_mode == 1) {
      const int slots = 2 * b->n_cus, passes = b->n_groups * 2 / SK_FAST2_NW_LDS;
      const int rounds = passes / slots, last = passes % slots;
      if (last == 0) limit = (uint64_t)b->n_voices / (128u * (unsigned)(rounds > 0 ? rounds : 1));
      else if (rounds == 0 || last * 20 <= slots * 11) limit = (uint64_t)b->n_voices / 600u;
      else limit = 0;
    }
    if (bound <= limit && bound <= over && rows * stride * sizeof(float) <= SK_INPLACE_MAX_BYTES) {
      if (rows * stride + 8 > b->env_gain_cap) {
        HIP_TRY(hipDeviceSynchronize());               /* (a block on another stream may still read the old rows) */
        if ((rc = grow(&b->d_env_gain, &b->env_gain_cap, rows * stride + 8))) return rc;
      }
      a.env_gain = b->d_env_gain;
      a.env_gain_stride = (int32_t)stride;
      a.env_gain_cap = (int32_t)rows;
      a.env_word_rows = SK_INPLACE_WORD_ROWS;
      a.env_count = b->d_violations + 1;
      inplace = 1;
    }
  }
  const int env_beside = two_env && !b->list_empty && !inplace;
  const int n_env = env_beside ? sk_env2_grid(&a) : 0;
  /* rows of the partial mix, the slab sums of the two-level mix-down, the per-frame master gains, and (when the envelope kernel
   * runs beside) its rows and its sum: one allocation */
  const size_t row = (size_t)num_frames * 2;
  const size_t gains_at = ((size_t)n_wg + SK_FINISH_SLABS) * row;
  const size_t env_at = (gains_at + (size_t)num_frames + 3) & ~(size_t)3;          /* 16-byte aligned */
  if (env_at + ((size_t)n_env + 1) * row > b->partial_cap && b->pp_parity >= 0) HIP_TRY(hipDeviceSynchronize());   /* (a master stage of the pipelined form may still read the old rows) */
  if ((rc = grow(&b->d_partial, &b->partial_cap, env_at + ((size_t)n_env + 1) * row))) return rc;
  const int pp = b->pp_parity >= 0 && !d_out;      /* pipelined sum-only form: skred_shard_render_mix_pipelined */
  a.partial = b->d_partial;
  a.slab_rows = b->d_partial + (size_t)n_wg * row;
  a.gains = b->d_partial + gains_at;
  if (pp) {
    /* the pipelined form's master stage of block k reads its gain row on another stream while block k + 1 renders: the two rows
     * live in an allocation of their own, at offsets that depend on nothing but the parity -- not behind the rows of d_partial,
     * whose number follows the kernel family and whose length follows num_frames, both free to change from block to block */
    if ((size_t)num_frames > b->pp_gains_cap) {
      HIP_TRY(hipDeviceSynchronize());               /* (a master stage may still read the old rows) */
      if (b->d_pp_gains) { hipFree(b->d_pp_gains); b->d_pp_gains = NULL; b->pp_gains_cap = 0; }
      const size_t cap = ((size_t)num_frames + 1023) & ~(size_t)1023;
      HIP_TRY(hipMalloc((void **)&b->d_pp_gains, 2 * cap * sizeof(float)));
      b->pp_gains_cap = cap;
    }
    a.gains = b->d_pp_gains + (b->pp_parity ? b->pp_gains_cap : 0);
  }
  a.env_rows = b->d_partial + env_at;
  a.env_sum = a.env_rows + (size_t)n_env * row;
  a.env_ticket = b->d_tickets + SK_FINISH_SLABS + 1;
  a.moved = b->d_tickets + SK_FINISH_SLABS + 2;
  a.n_env_rows = n_env;
  a.env_beside = env_beside;
  b->gains_offset = gains_at;
  a.n_rows = n_wg;
  a.finish = 1;
  /* The gain workgroup walks the master gain of every frame beside the renderers.  Single-GPU form: the last arriver applies
   * it.  Sum-only form (the multi-GPU render): the gains are left for skred_bank_master(), which runs after the RCCL sum
   * and then has nothing serial left to do; the carried gain is committed there (slot 1 holds it meanwhile). */
  a.wg_shift = 1;
  a.sum_out = d_sum;
  a.mix_out = d_out;
  a.num_channels = num_channels;
  a.gain_state = b->d_gain_state;
  /* (pipelined form: the NEXT block's render starts before this block's master stage has run, so the render commits the
   * carried gain itself and the master stage, sk_bank_master_pp, only scales) */
  a.gain_commit = (d_out || pp) ? b->d_gain_state : b->d_gain_state + 1;
  b->gains_frames = (d_out || pp) ? 0 : num_frames;  /* gains for a block of this many frames are waiting for skred_bank_master */
  a.tickets = b->d_tickets;
  a.vol_target = b->g.volume_final;
  a.vol_k = b->g.volume_smoother_smoothing;
#ifdef SK_ABLATE_FINISH   /* timing experiments only (tools/ab_finish.py says how such a library is built): the block's output is then garbage */
  a.finish = 0; a.wg_shift = 0; b->gains_frames = 0;
#endif

  if (pack_s) {
    /* the words the kernel reads, and voice_sample = 0 where the reference's skip rule would have left it and no lane does */
    if (b->pack_upload) {
      HIP_TRY(hipMemcpyAsync(b->d_pack_mask, b->h_pack_mask, (size_t)(b->n_padded / 64) * sizeof(uint64_t), hipMemcpyHostToDevice, s));
      b->pack_upload = 0;
    }
    if (b->pack_zero) {
      const hipError_t ez = (hipError_t)sk_launch_pack_zero(b->d_pack_mask, b->d_rw[SKS_FILT], b->n_padded, s);
      if (ez != hipSuccess) return fail(SKRED_E_NO_DEVICE, "pack_zero launch -> %s", hipGetErrorString(ez));
      b->pack_zero = 0;
    }
  }
  const int tslot = b->n_timed % SK_TIMING_RING;
  a.launch_ticket = ++b->launch_ticket;
  a.skip_env2 = two_env ? (uint32_t)b->list_empty : (uint32_t)(one_env && b->env_quiet);
  const int timed = b->timing_every > 0 && (b->launch_ticket % (uint32_t)b->timing_every) == 0;
  if (timed) HIP_TRY(hipEventRecord(b->ev0[tslot], s));
  hipError_t e;
  if (two_env && (env_beside || inplace || (a.launch_ticket & 63u) == 0)) { if ((rc = expect_report(b, &a, (env_beside || inplace) ? 2 : 3))) return rc; }
  else if (one_env && (!a.skip_env2 || (a.launch_ticket & 15u) == 0)) { if ((rc = expect_report(b, &a, 1))) return rc; }
  if (env_beside) {
    /* the list of this block, then fork: everything queued on `s` so far (updates, the classify pass, the list) is ahead of
     * the envelope kernel too; both render kernels become ready together */
    e = (hipError_t)sk_launch_collect(&a, s);
    if (e != hipSuccess) return fail(SKRED_E_NO_DEVICE, "collect launch -> %s", hipGetErrorString(e));
    HIP_TRY(hipEventRecord(b->ev_fork, s));
    HIP_TRY(hipStreamWaitEvent(b->side, b->ev_fork, 0));
    e = (hipError_t)sk_launch_env_fast2(&a, b->side);   /* (with a probe set it forwards to the probe instantiations) */
    if (e != hipSuccess) return fail(SKRED_E_NO_DEVICE, "envelope kernel launch -> %s", hipGetErrorString(e));
    HIP_TRY(hipEventRecord(b->ev_join, b->side));
  }
  if (inplace) {
    e = (hipError_t)sk_launch_gain(&a, s);
    if (e != hipSuccess) return fail(SKRED_E_NO_DEVICE, "gain kernel launch -> %s", hipGetErrorString(e));
    b->mask_p ^= 1;                                                  /* it wrote the next block's list */
  }
  if (modulated) {
    e = (hipError_t)sk_launch_render_mod(&a, n_wg, b->d_level, b->max_level, s);
  } else {
    e = (hipError_t)sk_launch_render(&a, n_wg, s);
  }
  if (e != hipSuccess) return fail(SKRED_E_NO_DEVICE, "render launch -> %s", hipGetErrorString(e));
  if (env_beside) {
    HIP_TRY(hipStreamWaitEvent(s, b->ev_join, 0));                   /* join: the block is complete on `s` */
    b->mask_p ^= 1;                                                  /* the survivors are the next block's list */
  }

  if (timed) {
    HIP_TRY(hipEventRecord(b->ev1[tslot], s));
    b->n_timed++;
  }
  b->last_family = b->last_kernel;
  b->last_in_place = inplace;
  b->last_split = split;
  b->last_pack = pack_s;

  /* advance the timeline exactly as synth.c:521,525 do: one count and one LCG draw per frame */
  b->g.synth_sample_count += (uint64_t)num_frames;
  uint64_t r = b->g.noise_rng;
  for (int i = 0; i < num_frames; i++) r = r * 6364136223846793005ULL + 1442695040888963407ULL;
  b->g.noise_rng = r;
  return SKRED_OK;
}

int skred_bank_render(skred_bank_t *b, int num_frames, int interp, float *d_partial, float *d_stems, void *stream) {
  if (!b || !d_partial || num_frames <= 0) return fail(SKRED_E_BAD_ARG, "render: bad arguments");
  return render_block(b, num_frames, interp, d_stems, d_partial, NULL, 0, (hipStream_t)stream);
}

/* The two halves of a block in the PIPELINED multi-GPU form (skred_shard.c: skred_shard_render_mix_pipelined): the render of block
 * k + 1 runs while the collective and the master stage of block k are still under way on another stream, so the per-frame gains
 * live in two alternating rows (`parity`) and the carried gain is committed by the render. */
int sk_bank_render_sum_pp(skred_bank_t *b, int num_frames, int interp, float *d_sum, int parity, void *stream) {
  if (!b || !d_sum || num_frames <= 0) return fail(SKRED_E_BAD_ARG, "render_sum_pp: bad arguments");
  b->pp_parity = parity & 1;
  const int rc = render_block(b, num_frames, interp, NULL, d_sum, NULL, 0, (hipStream_t)stream);
  b->pp_parity = -1;
  return rc;
}

int sk_bank_master_pp(skred_bank_t *b, const float *d_sum, int num_frames, int num_channels, float *d_out, int parity, void *stream) {
  if (!b || !d_sum || !d_out || num_frames <= 0 || num_channels < 2 || !b->d_pp_gains || (size_t)num_frames > b->pp_gains_cap)
    return fail(SKRED_E_BAD_ARG, "master_pp: bad arguments");
  HIP_TRY(hipSetDevice(b->device));
  const float *gains = b->d_pp_gains + ((parity & 1) ? b->pp_gains_cap : 0);
  /* (nothing to commit: slots 2 and 3 of the gain state are scratch) */
  const hipError_t e = (hipError_t)sk_launch_master_apply(d_sum, gains, d_out, num_frames, num_channels, b->d_gain_state + 2, b->d_gain_state + 3, (hipStream_t)stream);
  if (e != hipSuccess) return fail(SKRED_E_NO_DEVICE, "master launch -> %s", hipGetErrorString(e));
  return SKRED_OK;
}

int skred_bank_render_mix(skred_bank_t *b, int num_frames, int interp, float *d_out, int num_channels,
                          float *d_stems, void *stream) {
  if (!b || !d_out || num_frames <= 0 || num_channels < 2) return fail(SKRED_E_BAD_ARG, "render_mix: bad arguments");
  return render_block(b, num_frames, interp, d_stems, NULL, d_out, num_channels, (hipStream_t)stream);
}

int skred_bank_master(skred_bank_t *b, const float *d_sum, int num_frames, int num_channels, float *d_out, void *stream) {
  if (!b || !d_sum || !d_out || num_frames <= 0 || num_channels < 2) return fail(SKRED_E_BAD_ARG, "master: bad arguments");
  HIP_TRY(hipSetDevice(b->device));
  hipError_t e;
  if (b->gains_frames == num_frames && b->d_partial) {
    /* the render of this block (skred_bank_render, same stream) already walked the gains: scale and commit */
    const float *gains = b->d_partial + b->gains_offset;
    e = (hipError_t)sk_launch_master_apply(d_sum, gains, d_out, num_frames, num_channels, b->d_gain_state + 1, b->d_gain_state, (hipStream_t)stream);
    b->gains_frames = 0;
  } else {
    e = (hipError_t)sk_launch_master(d_sum, d_out, num_frames, num_channels, b->g.volume_final,
                                     b->g.volume_smoother_smoothing, b->d_gain_state, (hipStream_t)stream);
  }
  if (e != hipSuccess) return fail(SKRED_E_NO_DEVICE, "master launch -> %s", hipGetErrorString(e));
  return SKRED_OK;
}

int skred_bank_render_host(skred_bank_t *b, float *buffer, int num_frames, int num_channels, int interp, float *stems) {
  if (!b || !buffer || num_frames <= 0 || num_channels < 2) return fail(SKRED_E_BAD_ARG, "render_host: bad arguments");
  HIP_TRY(hipSetDevice(b->device));
  int rc;
  if ((rc = grow(&b->d_out, &b->out_cap, (size_t)num_frames * (size_t)num_channels))) return rc;
  const size_t stem_floats = (size_t)num_frames * (size_t)b->n_voices * 2;
  if (stems && (rc = grow(&b->d_stems, &b->stems_cap, stem_floats))) return rc;
  if (num_channels > 2) HIP_TRY(hipMemsetAsync(b->d_out, 0, (size_t)num_frames * num_channels * sizeof(float), NULL));
  if ((rc = skred_bank_render_mix(b, num_frames, interp, b->d_out, num_channels, stems ? b->d_stems : NULL, NULL))) return rc;
  HIP_TRY(hipMemcpy(buffer, b->d_out, (size_t)num_frames * num_channels * sizeof(float), hipMemcpyDeviceToHost));
  if (stems) HIP_TRY(hipMemcpy(stems, b->d_stems, stem_floats * sizeof(float), hipMemcpyDeviceToHost));
  return SKRED_OK;
}

float skred_bank_last_render_ms(skred_bank_t *b) {
  if (!b || b->n_timed == 0) return -1.0f;
  const int slot = (b->n_timed - 1) % SK_TIMING_RING;
  float ms = -1.0f;
  if (hipSetDevice(b->device) != hipSuccess) return -1.0f;
  if (hipEventSynchronize(b->ev1[slot]) != hipSuccess) return -1.0f;
  if (hipEventElapsedTime(&ms, b->ev0[slot], b->ev1[slot]) != hipSuccess) return -1.0f;
  return ms;
}

void skred_bank_timing_reset(skred_bank_t *b) { if (b) b->n_timed = 0; }

int skred_bank_timing_summary(skred_bank_t *b, float *mean_ms, float *min_ms, int *count) {
  if (!b) return fail(SKRED_E_BAD_ARG, "timing_summary");
  HIP_TRY(hipSetDevice(b->device));
  const int n = b->n_timed < SK_TIMING_RING ? b->n_timed : SK_TIMING_RING;
  double sum = 0.0;
  float mn = 0.0f;
  for (int i = 0; i < n; i++) {
    float ms = 0.0f;
    HIP_TRY(hipEventSynchronize(b->ev1[i]));
    HIP_TRY(hipEventElapsedTime(&ms, b->ev0[i], b->ev1[i]));
    sum += ms;
    if (i == 0 || ms < mn) mn = ms;
  }
  if (mean_ms) *mean_ms = n ? (float)(sum / n) : -1.0f;
  if (min_ms) *min_ms = n ? mn : -1.0f;
  if (count) *count = n;
  return SKRED_OK;
}
