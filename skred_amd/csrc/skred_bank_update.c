/*
 * skred_bank_update.c -- block-granular parameter updates on device-resident voices, and the deferred
 * queue that drives them (SURVEY 8f "next" #4; include/skred_amd.h: skred_bank_update / _defer / _run_queue).
 *
 * In the reference every control action -- typed, received over UDP, fired by a pattern step or by a deferred
 * item (seq.c:164-213,243-257; wire.c:869-892) -- ends in a few stores into the per-voice arrays, and the
 * audio callback sees them at its next block.  With the voices resident in HBM those stores have to travel:
 * this file is the protocol.  The host view (skred_voice_bank_t, the reference's array names) stays the place
 * control code writes to; it then names the voices it touched and WHICH KIND of field (SKRED_DIRTY_*), and
 * only those voices' planes / words are rewritten on the device, by one scatter kernel per batch, ordered on
 * the render stream before the next block.  Everything not named keeps the value the GPU last computed (a
 * full skred_bank_upload would overwrite the running phase, filter memory and smoother with stale host copies).
 */
#include <stdlib.h>
#include <pthread.h>
#include <sched.h>
#include <time.h>
#include <string.h>

#include "skred_bank_priv.h"

_Static_assert(SKU_PARAMS == SKRED_DIRTY_PARAMS && SKU_PHASE == SKRED_DIRTY_PHASE && SKU_ENV_STATE == SKRED_DIRTY_ENV_STATE &&
               SKU_PAN == SKRED_DIRTY_PAN && SKU_FILTER_STATE == SKRED_DIRTY_FILTER_STATE && SKU_SMOOTHER == SKRED_DIRTY_SMOOTHER &&
               SKU_HOLD == SKRED_DIRTY_HOLD && SKU_SAMPLE == SKRED_DIRTY_SAMPLE && SKU_STAMP_TRIGGER == SKRED_STAMP_TRIGGER &&
               SKU_STAMP_RELEASE == SKRED_STAMP_RELEASE && SKU_ENV_CLOCK == SKRED_DIRTY_ENV_CLOCK, "device update bits must equal the public SKRED_DIRTY_* values");

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* ------------------------------------------------------------------ one voice -> planes */

int sk_pack_voice(const skred_bank_t *b, const skred_voice_bank_t *h, int v, int dst, int phase_known,
                  sk_plane_t ro[SKP_COUNT], sk_plane_t rw[SKS_COUNT], sk_voice_meta_t *meta) {
  memset(ro, 0, SKP_COUNT * sizeof(sk_plane_t));
  memset(rw, 0, SKS_COUNT * sizeof(sk_plane_t));
  memset(meta, 0, sizeof(*meta));
  const int size = h->voice_table_size[v];
  const int64_t off = h->voice_table_offset[v];
  const int noise = h->voice_wave_table_index[v] == SKRED_WAVE_TABLE_NOISE_ALT;
  uint32_t flags = 0, features = 0;
  int usable = 1;
  if (!noise) {
    if (size <= 0) usable = 0;   /* the reference would dereference a NULL/empty table here */
    else if (off < 0 || (uint64_t)off + (uint64_t)size > b->table_floats)
      return fail(SKRED_E_RANGE, "voice %d: table [%lld,+%d) outside pool of %zu floats", v, (long long)off, size, b->table_floats);
  }
  if (!usable) flags |= SKF_INERT;
  const int windowed = h->voice_loop_enabled[v] && h->voice_loop_valid[v];   /* synth.c:235-238 */
  const float lo = windowed ? h->voice_loop_start_f[v] : 0.0f;
  const float hi = windowed ? h->voice_loop_end_f[v] : (float)size;
  if (h->voice_one_shot[v]) flags |= SKF_ONE_SHOT;
  if (h->voice_loop_enabled[v]) flags |= SKF_LOOPING;
  if (h->voice_direction[v]) flags |= SKF_REVERSE;
  if (h->voice_use_amp_envelope[v]) { flags |= SKF_USE_ENV; features |= SKB_ANY_ENV; }
  if (h->voice_filter_mode[v]) { flags |= SKF_FILTER; features |= SKB_ANY_FILTER; }
  if (h->voice_smoother_enable[v]) flags |= SKF_SMOOTH;
  if (h->voice_disconnect[v]) flags |= SKF_MUTED;
  if (noise) { flags |= SKF_NOISE; features |= SKB_ANY_NOISE; }
  /* a guard sample behind a table that is looped as a whole: the linear lookup's second tap never folds (SKF_GUARD) */
  if (usable && !noise && !windowed && b->h_tables && (uint64_t)off + (uint64_t)size < b->table_floats &&
      !(h->voice_one_shot[v] && !h->voice_loop_enabled[v]) &&     /* (a one-shot that plays to its end does not fold: its neighbour clamps) */
      memcmp(&b->h_tables[off + size], &b->h_tables[off], sizeof(float)) == 0)
    flags |= SKF_GUARD;
  const int has_mod = h->voice_freq_mod_osc[v] >= 0 || h->voice_amp_mod_osc[v] >= 0 ||
                      h->voice_pan_mod_osc[v] >= 0 || h->voice_cz_mode[v] != 0;
  /* Modulation the one-per-lane kernel can serve: FM / AM / pan by a HIGHER-indexed voice of the same aligned
   * 64-voice group (the carrier then reads last frame's voice_sample of its modulator: synth.c:548-555,584-587,
   * 597-602 in index order, so no ordering inside a frame is involved), AM / pan by the voice itself, no CZ. */
  int fm_only = 0;
  if (has_mod && h->voice_cz_mode[v] == 0) {
    const int src[3] = { h->voice_freq_mod_osc[v] == v ? -1 : h->voice_freq_mod_osc[v], h->voice_amp_mod_osc[v], h->voice_pan_mod_osc[v] };
    fm_only = 1;
    for (int k = 0; k < 3; k++) {
      if (src[k] < 0) continue;
      const int md = src[k] + (dst - v);
      const int above = md >= 0 && md < b->n_voices && (md >> 6) == (dst >> 6) && md > dst;
      const int self_ok = k > 0 && md == dst;
      if (!above && !self_ok) fm_only = 0;
    }
  }
  if (has_mod) { flags |= SKF_HAS_MOD; features |= fm_only ? SKB_ANY_FM : SKB_ANY_MOD; }
  int quant = h->voice_quantize[v], hold = h->voice_sample_hold_max[v];
  if (quant < 0 || quant > 30) quant = quant < 0 ? 0 : 30;
  if (hold < 0) hold = 0;
  if (hold > 0xFFFFFF) hold = 0xFFFFFF;
  if (quant || hold) features |= SKB_ANY_HOLDQ;
  {
    uint16_t c = 0;
    if (usable || noise) c |= SKC_REAL;
    if ((usable || noise) && !(h->voice_amp[v] == 0.0f)) c |= SKC_LIVE;
    if (flags & SKF_GUARD) c |= SKC_GUARD;
    if (h->voice_filter_mode[v]) c |= SKC_FILTER;
    if (h->voice_use_amp_envelope[v]) c |= SKC_ENV;
    const int stops = h->voice_one_shot[v] && !h->voice_loop_enabled[v];
    /* a one-shot that plays to its end, and reverse playback, are still "clean" voices for the one-per-lane kernel's
     * extended instantiation */
    if (stops || h->voice_direction[v] || quant || hold || !h->voice_smoother_enable[v] || noise) c |= SKC_STOPS;
    /* a phase that lives on the device was finite when it was uploaded and the kernels keep it so */
    const float ph = phase_known ? h->voice_phase[v] : 0.0f, pi = h->voice_phase_inc[v];
    const int finite = (ph - ph == 0.0f) && (pi - pi == 0.0f) && (lo - lo == 0.0f) && (hi - hi == 0.0f) && hi > lo;
    if ((has_mod && !fm_only) || (!finite && !noise))        /* (a noise voice never runs its oscillator: synth.c:543-546) */
      c |= SKC_EXOTIC;
    if (fm_only) c |= SKC_FM;
    {
      /* the pair shape: an even voice whose frequency, amplitude and pan modulators are each the voice after it -- or, amplitude
       * and pan, the voice itself (`F1`, `A1`, `P1`, `A0`, `P0` on voice 0) */
      const int fo = h->voice_freq_mod_osc[v] == v ? -1 : h->voice_freq_mod_osc[v], ao = h->voice_amp_mod_osc[v], po = h->voice_pan_mod_osc[v];
      const int pair = (dst & 1) == 0 && (fo < 0 || fo - v == 1) && (ao < 0 || ao - v == 1 || ao == v) && (po < 0 || po - v == 1 || po == v);
      if (fm_only && !pair) c |= SKC_FM_ODD;
      if (fm_only && pair && (ao >= 0 || po >= 0)) c |= SKC_PAIR_AP;
    }
    meta->cls = c;
  }
  meta->features = features;
  const skred_envelope_t *e = &h->voice_amp_envelope[v];
  const skred_mmf_t *f = &h->voice_filter[v];

  ro[SKP_OSC].w[0] = f2u(h->voice_phase_inc[v]); ro[SKP_OSC].w[1] = f2u(usable ? lo : 0.0f);
  ro[SKP_OSC].w[2] = f2u(usable ? hi : 1.0f);    ro[SKP_OSC].w[3] = f2u(h->voice_amp[v]);
  ro[SKP_TAB].w[0] = (uint32_t)(int32_t)(usable && !noise ? off : 0);
  ro[SKP_TAB].w[1] = (uint32_t)(usable && !noise ? size : 1); ro[SKP_TAB].w[2] = flags;
  ro[SKP_TAB].w[3] = (uint32_t)quant | ((uint32_t)hold << 8);
  ro[SKP_ENV_T].w[0] = f2u(e->attack_time);   ro[SKP_ENV_T].w[1] = f2u(e->decay_time);
  ro[SKP_ENV_T].w[2] = f2u(e->sustain_level); ro[SKP_ENV_T].w[3] = f2u(e->release_time);
  ro[SKP_ENV_S].w[0] = (uint32_t)(e->sample_start & 0xFFFFFFFFu);
  ro[SKP_ENV_S].w[1] = (uint32_t)(e->sample_start >> 32);
  ro[SKP_ENV_S].w[2] = (uint32_t)(e->sample_release & 0xFFFFFFFFu);
  ro[SKP_ENV_S].w[3] = (uint32_t)(e->sample_release >> 32);
  ro[SKP_GAIN].w[0] = f2u(e->velocity); ro[SKP_GAIN].w[1] = f2u(h->voice_smoother_smoothing[v]);
  ro[SKP_GAIN].w[2] = f2u(f->b0);       ro[SKP_GAIN].w[3] = f2u(f->b1);
  ro[SKP_FILT].w[0] = f2u(f->b2); ro[SKP_FILT].w[1] = f2u(f->a1);
  ro[SKP_FILT].w[2] = f2u(f->a2); ro[SKP_FILT].w[3] = f2u(h->voice_cz_distortion[v]);
  {
    /* modulator indices: host index -> lane inside the carrier's 64-voice device group, -1 = unused
     * (FM ignores a self reference, synth.c:549; the CZ source only matters with CZ on, synth.c:262) */
    int src[4] = { h->voice_freq_mod_osc[v], h->voice_amp_mod_osc[v], h->voice_pan_mod_osc[v],
                   h->voice_cz_mode[v] ? h->voice_cz_mod_osc[v] : -1 };
    if (src[0] == v) src[0] = -1;
    for (int k = 0; k < 4; k++) {
      int lane_k = -1;
      if (src[k] >= 0) {
        const int md = src[k] + (dst - v);
        if (md < 0 || md >= b->n_voices || (md >> 6) != (dst >> 6)) meta->cls |= SKC_ESCAPES;
        else lane_k = md & 63;
      }
      meta->mod_lane[k] = (int8_t)lane_k;
      ro[SKP_MODI].w[k] = (uint32_t)(int32_t)lane_k;
    }
  }
  ro[SKP_MODF].w[0] = f2u(h->voice_freq_mod_depth[v]); ro[SKP_MODF].w[1] = f2u(h->voice_freq_scale[v]);
  ro[SKP_MODF].w[2] = f2u(h->voice_amp_mod_depth[v]);  ro[SKP_MODF].w[3] = f2u(h->voice_pan_mod_depth[v]);
  ro[SKP_MODX].w[0] = f2u(h->voice_cz_mod_depth[v]); ro[SKP_MODX].w[1] = (uint32_t)h->voice_cz_mode[v];

  rw[SKS_OSC].w[0] = f2u(h->voice_phase[v]); rw[SKS_OSC].w[1] = f2u(h->voice_smoother_gain[v]);
  rw[SKS_OSC].w[2] = f2u(f->x1);             rw[SKS_OSC].w[3] = f2u(f->x2);
  rw[SKS_FILT].w[0] = f2u(f->y1); rw[SKS_FILT].w[1] = f2u(f->y2);
  rw[SKS_FILT].w[2] = f2u(h->voice_sample[v]);
  rw[SKS_FILT].w[3] = (h->voice_finished[v] ? SKR_FINISHED : 0u) | (e->is_active ? SKR_ENV_ACTIVE : 0u);
  rw[SKS_MISC].w[0] = f2u(h->voice_sample_hold[v]); rw[SKS_MISC].w[1] = (uint32_t)h->voice_sample_hold_count[v];
  rw[SKS_MISC].w[2] = f2u(h->voice_pan_left[v]);    rw[SKS_MISC].w[3] = f2u(h->voice_pan_right[v]);
  return SKRED_OK;
}

/* what voice `dst` now means for kernel selection (counters instead of a scan: a bank has up to millions of voices) */
void sk_apply_meta(skred_bank_t *b, int dst, const sk_voice_meta_t *m, int params_travel) {
  if (!params_travel) return;
  const uint16_t old = b->h_class[dst], now = m->cls;
  if (old != now) {
    if (old & SKC_REAL) { b->cnt_real--; if (old & SKC_FILTER) b->cnt_filter--; if (old & SKC_ENV) b->cnt_env--; if (old & SKC_EXOTIC) b->cnt_exotic--; if (old & SKC_STOPS) b->cnt_stops--; if (old & SKC_FM) b->cnt_fm--; if (old & SKC_FM_ODD) b->cnt_fm_odd--; if (old & SKC_PAIR_AP) b->cnt_pair_ap--; if (old & SKC_GUARD) b->cnt_guard--; }
    if (now & SKC_REAL) { b->cnt_real++; if (now & SKC_FILTER) b->cnt_filter++; if (now & SKC_ENV) b->cnt_env++; if (now & SKC_EXOTIC) b->cnt_exotic++; if (now & SKC_STOPS) b->cnt_stops++; if (now & SKC_FM) b->cnt_fm++; if (now & SKC_FM_ODD) b->cnt_fm_odd++; if (now & SKC_PAIR_AP) b->cnt_pair_ap++; if (now & SKC_GUARD) b->cnt_guard++; }
    /* per-voice bits that are not kernel classes: counted whether or not the voice can sound, and recounted whenever the
     * voice is written again -- a routing that escaped its group stops blocking the bank once it is fixed */
    b->cnt_escapes += ((now & SKC_ESCAPES) != 0) - ((old & SKC_ESCAPES) != 0);
    b->h_class[dst] = now;
    b->class_dirty = 1;
    if ((old ^ now) & SKC_LIVE) { b->h_pack_dirty[dst >> 6] = 1; b->pack_any_dirty = 1; }
  }
  for (int k = 0; k < 4; k++) {
    int8_t *slot = &b->h_mod[(size_t)k * b->n_padded + dst];
    if (*slot != m->mod_lane[k]) { *slot = m->mod_lane[k]; b->mod_dirty = 1; b->class_dirty = 1; b->h_pack_dirty[dst >> 6] = 1; b->pack_any_dirty = 1; }
  }
  if ((b->features | m->features) != b->features) { b->features |= m->features; b->class_dirty = 1; b->mod_dirty = 1; }
}

/* ------------------------------------------------------------------ batches of voice updates */

typedef struct sk_queue_item {
  struct sk_queue_item *next;
  uint64_t when;
  int n;
  sk_update_t *rec;          /* device-format records */
  sk_voice_meta_t *meta;     /* NULL unless the batch carries SKRED_DIRTY_PARAMS */
} sk_queue_item_t;

#define SK_STAMP_ONLY(d) (((d) & ~(uint32_t)(SKRED_STAMP_TRIGGER | SKRED_STAMP_RELEASE)) == 0)

/* Packing is pure per voice (sk_pack_voice reads the host view and the bank's table geometry, writes its own record), so a
 * large batch -- a chord of thousands of notes, a preset change -- is packed by a few threads side by side: ~300 ns per
 * re-parameterised voice on one core was what a block with 2 % of a 2^20-voice bank re-triggered spent most of its time on. */
#define SK_PACK_PER_THREAD 512     /* voices per thread below which splitting further does not pay (a thread costs ~25 us to start) */
#define SK_PACK_THREADS 8

typedef struct {
  const skred_bank_t *b; const skred_voice_bank_t *h; const int32_t *voices; int lo, hi; uint32_t dirty;
  sk_update_t *rec; sk_voice_meta_t *meta; int stamp_only; int rc; int bad_voice; int has_bad;
} sk_pack_job_t;

static void *pack_range(void *arg) {
  sk_pack_job_t *j = (sk_pack_job_t *)arg;
  j->rc = SKRED_OK;
  j->has_bad = 0;
  for (int i = j->lo; i < j->hi; i++) {
    const int v = j->voices[i];
    if (v < 0 || v >= j->b->n_voices || v >= j->h->n_voices) { j->rc = SKRED_E_RANGE; j->bad_voice = v; j->has_bad = 1; return NULL; }
    if (!j->stamp_only) {
      sk_voice_meta_t m;
      const int rc = sk_pack_voice(j->b, j->h, v, v, (j->dirty & SKRED_DIRTY_PHASE) != 0, j->rec[i].ro, j->rec[i].rw, &m);
      if (rc) { j->rc = rc; return NULL; }
      if (j->meta) j->meta[i] = m;
    }
    j->rec[i].voice = v;
    j->rec[i].dirty = j->dirty;
  }
  return NULL;
}

static int build_batch(const skred_bank_t *b, const skred_voice_bank_t *h, const int32_t *voices, int n,
                       uint32_t dirty, sk_update_t **rec_out, sk_voice_meta_t **meta_out) {
  *rec_out = NULL;
  *meta_out = NULL;
  if ((dirty & ~(uint32_t)SKRED_DIRTY_VALID_MASK) || !dirty) return fail(SKRED_E_BAD_ARG, "update: dirty mask 0x%x", dirty);
  sk_update_t *rec = (sk_update_t *)calloc((size_t)n, sizeof(sk_update_t));
  const int wants_meta = (dirty & SKRED_DIRTY_PARAMS) != 0;
  sk_voice_meta_t *meta = wants_meta ? (sk_voice_meta_t *)calloc((size_t)n, sizeof(sk_voice_meta_t)) : NULL;
  if (!rec || (wants_meta && !meta)) { free(rec); free(meta); return fail(SKRED_E_NO_MEM, "update staging"); }
  const int stamp_only = SK_STAMP_ONLY(dirty);        /* note-on / note-off stamps carry no values: nothing to pack */
  sk_pack_job_t job[SK_PACK_THREADS];
  pthread_t th[SK_PACK_THREADS];
  int n_jobs = stamp_only ? 1 : n / SK_PACK_PER_THREAD, started = 0;
  if (n_jobs < 1) n_jobs = 1;
  if (n_jobs > SK_PACK_THREADS) n_jobs = SK_PACK_THREADS;
  for (int k = 0; k < n_jobs; k++) {
    job[k] = (sk_pack_job_t){ b, h, voices, (int)((int64_t)n * k / n_jobs), (int)((int64_t)n * (k + 1) / n_jobs), dirty, rec, meta, stamp_only, SKRED_OK, 0, 0 };
  }
  for (int k = 1; k < n_jobs; k++) {                  /* job 0 runs on the caller's thread */
    if (pthread_create(&th[k], NULL, pack_range, &job[k]) != 0) break;
    started = k;
  }
  pack_range(&job[0]);
  for (int k = started + 1; k < n_jobs; k++) pack_range(&job[k]);   /* (threads that could not be started: their share here) */
  for (int k = 1; k <= started; k++) pthread_join(th[k], NULL);
  for (int k = 0; k < n_jobs; k++) {
    if (job[k].rc == SKRED_OK) continue;
    const int rc = job[k].rc, bad = job[k].bad_voice, has_bad = job[k].has_bad;
    free(rec); free(meta);
    return has_bad ? fail(SKRED_E_RANGE, "update: voice %d outside the bank", bad) : rc;   /* (sk_pack_voice has said why itself) */
  }
  *rec_out = rec;
  *meta_out = meta;
  return SKRED_OK;
}

/* One staging slot of the ring: pinned host buffer, device buffer, and the number of the batch after which both may be
 * reused -- the batch's last kernel stores it into pinned memory (sk_batch_done), the host looks there.  With a ring the
 * host only ever waits for a batch issued SK_UPD_RING batches ago (a host that queues blocks far ahead of the device is
 * held back here, as an event would hold it); a single slot would stall every update behind the render that precedes it in
 * the stream. */
static int staging_slot(skred_bank_t *b, size_t bytes, hipStream_t s, sk_upd_slot_t **out) {
  const int idx = (int)(b->upd_head++ % SK_UPD_RING);
  sk_upd_slot_t *sl = &b->upd[idx];
  if (!b->h_upd_done) {
    HIP_TRY(hipHostMalloc((void **)&b->h_upd_done, SK_UPD_RING * sizeof(uint32_t), hipHostMallocCoherent));   /* (the host polls words the device stores while kernels run) */
    for (int i = 0; i < SK_UPD_RING; i++) b->h_upd_done[i] = 0;
    HIP_TRY(hipMalloc((void **)&b->d_upd_cnt, SK_UPD_RING * sizeof(uint32_t)));
    HIP_TRY(hipMemset(b->d_upd_cnt, 0, SK_UPD_RING * sizeof(uint32_t)));
  }
  if (sl->seq) {
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (unsigned spins = 0; __atomic_load_n(&b->h_upd_done[idx], __ATOMIC_ACQUIRE) != sl->seq; spins++) {
      if ((spins & 63) == 63) {
        sched_yield();
        clock_gettime(CLOCK_MONOTONIC, &t1);
        /* Five seconds without the batch's kernel having run: its stream is blocked -- e.g. behind an event the caller means to record
         * only after this call returns.  Waiting for the device here would turn that into a deadlock, so the call fails instead; the
         * slot stays marked in flight (its buffers are still the batch's), and the ring moves on to the next slot. */
        if (t1.tv_sec - t0.tv_sec > 5)
          return fail(SKRED_E_NO_DEVICE, "update: staging slot %d is still in flight after 5 s (batch %u of a stream that has not run): "
                                         "the bank is %d update batches ahead of the device", idx, sl->seq, SK_UPD_RING);
      }
    }
    sl->seq = 0;
  }
  if (bytes > sl->cap) {
    if (sl->d) { (void)hipFree(sl->d); sl->d = NULL; }
    if (sl->h) { (void)hipHostFree(sl->h); sl->h = NULL; }
    sl->cap = 0;
    size_t cap = 64 * 1024;
    while (cap < bytes) cap *= 2;
    HIP_TRY(hipMalloc(&sl->d, cap));
    HIP_TRY(hipHostMalloc(&sl->h, cap, hipHostMallocDefault));
    sl->cap = cap;
  }
  (void)s;
  *out = sl;
  return SKRED_OK;
}

/* Where the scatter kernel reads a staged batch from.  A small batch -- the notes of one audio block -- is read straight out
 * of the pinned host buffer (it is mapped into the device's address space): a copy engine transfer in front of a 5 us kernel
 * cost the stream ~12 us each (the copy and the gap behind it), four times per block under note traffic.  A large batch goes
 * through the copy engine into the slot's device twin: bandwidth matters there, not latency. */
#define SK_UPD_ZERO_COPY_MAX (256 * 1024)
static const void *sk_stage(sk_upd_slot_t *sl, size_t bytes, hipStream_t s) {
  if (bytes <= SK_UPD_ZERO_COPY_MAX) return sl->h;
  const hipError_t e = hipMemcpyAsync(sl->d, sl->h, bytes, hipMemcpyHostToDevice, s);
  if (e != hipSuccess) { (void)fail(SKRED_E_NO_DEVICE, "update copy -> %s", hipGetErrorString(e)); return NULL; }
  return sl->d;
}

/* push records to the device and scatter them; a voice named twice is applied in order (one launch per run
 * of distinct voices, found with a per-voice epoch mark: linear in the batch) */
static int apply_batch(skred_bank_t *b, const sk_update_t *rec, const sk_voice_meta_t *meta, int n, hipStream_t s) {
  HIP_TRY(hipSetDevice(b->device));
  const uint32_t dirty = rec[0].dirty;                 /* a batch has one mask */
  sk_upd_slot_t *sl;
  if (SK_STAMP_ONLY(dirty)) {
    /* note-ons / note-offs: voice ids only (stamping the same voice twice with the same clock is idempotent) */
    int rc = staging_slot(b, (size_t)n * sizeof(int32_t), s, &sl);
    if (rc) return rc;
    int32_t *ids = (int32_t *)sl->h;
    for (int i = 0; i < n; i++) ids[i] = rec[i].voice;
    const void *src = sk_stage(sl, (size_t)n * sizeof(int32_t), s);
    if (!src) return SKRED_E_NO_DEVICE;
    const int idx = (int)(sl - b->upd);
    if (++b->upd_seq == 0) b->upd_seq = 1;
    const hipError_t e = (hipError_t)sk_launch_stamp((const int32_t *)src, n, dirty, b->d_ro, b->d_rw, b->g.synth_sample_count, b->d_mask[b->mask_p],
                                                     b->d_upd_cnt + idx, (uint32_t *)b->h_upd_done + idx, b->upd_seq, s);
    if (e != hipSuccess) return fail(SKRED_E_NO_DEVICE, "stamp launch -> %s", hipGetErrorString(e));
    sl->seq = b->upd_seq;
    b->touched_total += (uint64_t)n;
    sk_control_changed(b);
    return SKRED_OK;
  }
  int rc = staging_slot(b, (size_t)n * sizeof(sk_update_t), s, &sl);
  if (rc) return rc;
  memcpy(sl->h, rec, (size_t)n * sizeof(sk_update_t));
  const sk_update_t *src = (const sk_update_t *)sk_stage(sl, (size_t)n * sizeof(sk_update_t), s);
  if (!src) return SKRED_E_NO_DEVICE;
  const int idx = (int)(sl - b->upd);
  if (!b->upd_mark) {
    b->upd_mark = (uint32_t *)calloc((size_t)b->n_voices, sizeof(uint32_t));
    if (!b->upd_mark) return fail(SKRED_E_NO_MEM, "update marks");
  }
  int start = 0;
  while (start < n) {
    if (++b->upd_epoch == 0) { memset(b->upd_mark, 0, (size_t)b->n_voices * sizeof(uint32_t)); b->upd_epoch = 1; }
    int end = start;
    for (; end < n; end++) {
      uint32_t *m = &b->upd_mark[rec[end].voice];
      if (*m == b->upd_epoch) break;                    /* named before in this run: the next launch takes it */
      *m = b->upd_epoch;
    }
    const int last = end == n;                          /* the batch's last launch reports the slot free */
    if (last && ++b->upd_seq == 0) b->upd_seq = 1;
    const hipError_t e = (hipError_t)sk_launch_update(src + start, end - start, b->d_ro, b->d_rw,
                                                      b->g.synth_sample_count, b->d_mask[b->mask_p],
                                                      last ? b->d_upd_cnt + idx : NULL, last ? (uint32_t *)b->h_upd_done + idx : NULL, b->upd_seq, s);
    if (e != hipSuccess) {                              /* (launches of this batch already queued still read the slot) */
      (void)hipStreamSynchronize(s);
      return fail(SKRED_E_NO_DEVICE, "update launch -> %s", hipGetErrorString(e));
    }
    if (last) sl->seq = b->upd_seq;
    start = end;
  }
  if (meta && (dirty & SKRED_DIRTY_PARAMS)) for (int i = 0; i < n; i++) sk_apply_meta(b, rec[i].voice, &meta[i], 1);
  if (dirty & SKRED_DIRTY_SAMPLE) b->pack_zero = 1;    /* (a voice_sample written onto a skipped voice: cleared again before a packed block) */
  b->touched_total += (uint64_t)n;
  sk_control_changed(b);
  return SKRED_OK;
}

int skred_bank_update(skred_bank_t *b, const skred_voice_bank_t *h, const int32_t *voices, int n_voices,
                      uint32_t dirty, void *stream) {
  if (!b || !h || !voices || n_voices < 0) return fail(SKRED_E_BAD_ARG, "update: bad arguments");
  if (n_voices == 0) return SKRED_OK;
  sk_update_t *rec;
  sk_voice_meta_t *meta;
  int rc = build_batch(b, h, voices, n_voices, dirty, &rec, &meta);
  if (rc) return rc;
  rc = apply_batch(b, rec, meta, n_voices, (hipStream_t)stream);
  free(rec);
  free(meta);
  return rc;
}

/* ------------------------------------------------------------------ the deferred queue (seq.c:243-257, 170-177) */

int skred_bank_defer(skred_bank_t *b, uint64_t when, const skred_voice_bank_t *h, const int32_t *voices,
                     int n_voices, uint32_t dirty) {
  if (!b || !h || !voices || n_voices <= 0) return fail(SKRED_E_BAD_ARG, "defer: bad arguments");
  if (b->queue_len >= SKRED_QUEUE_SIZE) return fail(SKRED_E_RANGE, "defer: queue full (%d items)", SKRED_QUEUE_SIZE);
  sk_queue_item_t *it = (sk_queue_item_t *)calloc(1, sizeof(*it));
  if (!it) return fail(SKRED_E_NO_MEM, "defer");
  const int rc = build_batch(b, h, voices, n_voices, dirty, &it->rec, &it->meta);
  if (rc) { free(it); return rc; }
  it->when = when;
  it->n = n_voices;
  if (b->queue_tail) b->queue_tail->next = it; else b->queue = it;
  b->queue_tail = it;
  b->queue_len++;
  return SKRED_OK;
}

int skred_bank_queue_pending(const skred_bank_t *b) { return b ? b->queue_len : 0; }

/* ------------------------------------------------------------------ pattern steps (seq.c:179-213 through skred_seq.c) */

typedef struct sk_pat_step { int n; sk_update_t *rec; sk_voice_meta_t *meta; } sk_pat_step_t;

skred_seq_t *skred_bank_seq(skred_bank_t *b) {
  if (!b) return NULL;
  if (!b->seq && skred_seq_create(&b->seq) != SKRED_OK) return NULL;
  return b->seq;
}

int skred_bank_set_sample_rate(skred_bank_t *b, float rate) {
  if (!b || !(rate > 0.0f)) return fail(SKRED_E_BAD_ARG, "set_sample_rate");
  b->seq_rate = rate;
  return SKRED_OK;
}

static void pat_step_free(sk_pat_step_t *st) { free(st->rec); free(st->meta); st->rec = NULL; st->meta = NULL; st->n = 0; }

static int pat_slot(skred_bank_t *b, int pattern, int step, sk_pat_step_t **out) {
  if (pattern < 0 || pattern >= SKRED_PATTERNS_MAX || step < 0 || step >= SKRED_SEQ_STEPS_MAX)
    return fail(SKRED_E_BAD_ARG, "pattern %d step %d", pattern, step);
  if (!skred_bank_seq(b)) return fail(SKRED_E_NO_MEM, "pattern clock");
  if (!b->pat) {
    b->pat = (sk_pat_step_t *)calloc((size_t)SKRED_PATTERNS_MAX * SKRED_SEQ_STEPS_MAX, sizeof(sk_pat_step_t));
    if (!b->pat) return fail(SKRED_E_NO_MEM, "pattern steps");
  }
  *out = &b->pat[(size_t)pattern * SKRED_SEQ_STEPS_MAX + step];
  return SKRED_OK;
}

/* What the reference stores as a line of wire text (seq_step_set, seq.c:267-270) is stored here as what that line would
 * DO: the `dirty` parts of the listed voices as the host view has them now.  n_voices == 0: a rest -- the step is
 * occupied (the pattern does not wrap here) but stores nothing. */
int skred_bank_pattern_step_set(skred_bank_t *b, int pattern, int step, const skred_voice_bank_t *h, const int32_t *voices,
                                int n_voices, uint32_t dirty) {
  if (!b || n_voices < 0 || (n_voices > 0 && (!h || !voices))) return fail(SKRED_E_BAD_ARG, "pattern_step_set: bad arguments");
  sk_pat_step_t *st;
  int rc = pat_slot(b, pattern, step, &st);
  if (rc) return rc;
  sk_update_t *rec = NULL;
  sk_voice_meta_t *meta = NULL;
  if (n_voices > 0 && (rc = build_batch(b, h, voices, n_voices, dirty, &rec, &meta))) return rc;
  pat_step_free(st);
  st->n = n_voices; st->rec = rec; st->meta = meta;
  return skred_seq_step_set(b->seq, pattern, step, 1);
}

int skred_bank_pattern_step_clear(skred_bank_t *b, int pattern, int step) {
  if (!b) return fail(SKRED_E_BAD_ARG, "pattern_step_clear");
  sk_pat_step_t *st;
  const int rc = pat_slot(b, pattern, step, &st);
  if (rc) return rc;
  pat_step_free(st);
  return skred_seq_step_set(b->seq, pattern, step, 0);
}

void sk_patterns_free(skred_bank_t *b) {
  if (b->pat) {
    for (size_t i = 0; i < (size_t)SKRED_PATTERNS_MAX * SKRED_SEQ_STEPS_MAX; i++) pat_step_free(&b->pat[i]);
    free(b->pat);
    b->pat = NULL;
  }
  skred_seq_destroy(b->seq);
  b->seq = NULL;
}

int skred_bank_run_queue(skred_bank_t *b, int frame_count, void *stream) {
  if (!b || frame_count < 0) return fail(SKRED_E_BAD_ARG, "run_queue: bad arguments");
  const uint64_t horizon = b->g.synth_sample_count + (uint64_t)frame_count;      /* seq.c:173 */
  int applied = 0;
  sk_queue_item_t **link = &b->queue, *prev = NULL;
  while (*link) {
    sk_queue_item_t *it = *link;
    if (it->when <= horizon) {
      const int rc = apply_batch(b, it->rec, it->meta, it->n, (hipStream_t)stream);
      if (rc) return rc;
      *link = it->next;
      if (b->queue_tail == it) b->queue_tail = prev;
      free(it->rec); free(it->meta); free(it);
      b->queue_len--;
      applied++;
    } else {
      prev = it;
      link = &it->next;
    }
  }
  /* the second half of seq(): the tempo clock and the steps it fires (seq.c:179-213) */
  if (b->seq) {
    int32_t fired[SKRED_PATTERNS_MAX];
    const int n = skred_seq_tick(b->seq, frame_count, b->seq_rate > 0.0f ? b->seq_rate : 44100.0f, fired, SKRED_PATTERNS_MAX);
    if (n < 0) return n;
    for (int i = 0; i < n; i++) {
      /* a step switched on through the bank's clock itself (skred_seq_step_set on skred_bank_seq()) holds no batch: a rest */
      if (!b->pat) { applied++; continue; }
      const sk_pat_step_t *st = &b->pat[(size_t)(fired[i] >> 16) * SKRED_SEQ_STEPS_MAX + (fired[i] & 0xFFFF)];
      if (st->n > 0) {
        const int rc = apply_batch(b, st->rec, st->meta, st->n, (hipStream_t)stream);
        if (rc) return rc;
      }
      applied++;
    }
  }
  return applied;
}

void sk_queue_free(skred_bank_t *b) {
  sk_queue_item_t *it = b->queue;
  while (it) { sk_queue_item_t *n = it->next; free(it->rec); free(it->meta); free(it); it = n; }
  b->queue = b->queue_tail = NULL;
  b->queue_len = 0;
}
