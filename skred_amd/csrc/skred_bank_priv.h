/*
 * skred_bank_priv.h -- internals shared by the translation units behind include/skred_amd.h
 * (skred_bank.c: lifecycle, upload / download, render; skred_bank_update.c: block-granular updates
 * and the deferred queue).  Not installed.
 */
#ifndef SKRED_BANK_PRIV_H
#define SKRED_BANK_PRIV_H

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdint.h>

#include "skred_amd.h"
#include "skred_device_layout.h"
#include "skred_launch.h"

#define SK_TIMING_RING 256
#define SK_REPORT_RING 64
#define SK_UPD_RING 8

typedef struct {
  void *h, *d;                /* pinned host buffer and its device twin */
  size_t cap;                 /* bytes */
  uint32_t seq;               /* != 0: a batch read from this slot is in flight; its last kernel stores this number into the bank's
                                 h_upd_done[slot] when it has run (no event: skred_update_kernels.hip, sk_batch_done) */
} sk_upd_slot_t;
#define SK_SPLIT_MAX_LDS (160u * 1024u)   /* LDS of a CU: sk_render_split_kernel's workgroup must fit (twice, for two workgroups per CU) */
#define SK_FM2_MIN_VOICES 1024      /* two-operator FM banks at least this large keep each (carrier, modulator) pair in one lane */
#define SK_FAST2_MOTION_MIN_VOICES 278528   /* ... while envelopes move: banks smaller than this stay on the one-voice kernel (round 3, the envelope kernel
                                               beside the steady one: 262 144 voices 184 vs 198 us per block, 294 912 voices 214 vs 202; tools/ab_env_mid.py) */
#define SK_FAST2_MIN_VOICES 212992   /* banks at least this large use two voices per lane (measured crossover, 512-frame blocks, C2 recipe: 196608 voices 86 vs 95 us, 262144 voices 108 vs 101 us; profiles/r02_v1_measure_banks.txt) */

struct skred_bank {
  int device;
  int n_cus;                  /* compute units of the device (the two-per-lane kernel's passes come one per CU, then two) */
  int n_voices, n_padded, n_groups;
  sk_plane_t *d_planes;       /* the slab behind d_ro[] / d_rw[] */
  sk_plane_t *d_ro[SKP_COUNT];
  sk_plane_t *d_rw[SKS_COUNT];
  float *d_tables;
  float *h_tables;            /* host copy of the pool (sk_pack_voice looks for guard samples: SKF_GUARD) */
  size_t table_floats;        /* real pool size       */
  size_t table_floats_padded; /* rounded up to 4      */
  float *d_partial;           /* [n_wg][F][2] workgroup rows, then [SK_FINISH_SLABS][F][2] slab sums, then [F] master gains */
  size_t partial_cap;         /* floats */
  uint32_t *d_tickets;        /* [SK_FINISH_SLABS + 1] arrival counters of the in-kernel mix-down */
  int timing_every;           /* SKRED_OPT_KERNEL_TIMING: bracket every n-th launch's render kernels with an event pair (0: none) */
  float *d_gain_state;        /* [0] master smoother gain carried between blocks; [1] the gain a sum-only render prepared for skred_bank_master */
  float *d_pp_gains;          /* [2][pp_gains_cap]: the per-frame master gains of the pipelined sum-only form's two blocks in flight (fixed rows:
                                 block k's master stage reads its row while block k + 1 renders, whatever kernel and block length that one has) */
  size_t pp_gains_cap;        /* floats per row */
  int pp_parity;              /* >= 0 only inside sk_bank_render_sum_pp: which of the two gain rows the sum-only render fills */
  int gains_frames;           /* > 0: the latest skred_bank_render() left the master gains of a block of this many frames in d_partial */
  size_t gains_offset;        /* ... at this float offset */
  float *d_out, *d_stems;     /* scratch of skred_bank_render_host */
  size_t out_cap, stems_cap;
  uint16_t *h_class;          /* per-voice SKC_* bits, shadow used to pick the kernel */
  int8_t *h_mod;              /* [4][n_padded] modulator lane inside the 64-voice group (fm, am, pm, cz) or -1 */
  int *h_level;               /* [n_padded] dependency level of each voice (modulated banks) */
  int *d_level;
  int32_t *d_env_list;        /* the voices on the motion list, in ascending order (sk_collect_expand_kernel) */
  int32_t *d_env_off;         /* per 128-voice wave slice: where its voices start in d_env_list; one more slot: their number */
  int32_t *d_group_flag;      /* per 128-voice wave slice: listed voices; one more slot: the one-voice family's "an envelope moved" ticket */
  /* THE MOTION LIST of the two-per-lane family (skred_device_layout.h: mask_cur): a bit per voice, double-buffered -- the
   * block reads d_mask[mask_p] and builds d_mask[mask_p ^ 1] (survivors), control actions OR into d_mask[mask_p] */
  uint64_t *d_mask[2];
  int mask_p;
  int mask_dirty;             /* the list must be rebuilt from the planes before the next two-per-lane block (upload, clock change,
                                 a stretch on another kernel family, a violation report) */
  int list_empty;             /* STRUCTURAL: the last list the device built was empty and nothing was added since (no control action);
                                 the envelope kernel is then not launched -- with an empty list it has nothing to render */
  uint32_t *d_violations;     /* sticky device counter: sk_render_fast2_kernel found a moving voice that was not listed; the word behind
                                 it: sk_gain_kernel's row counter (skred_device_layout.h: env_count) */
  /* listed voices rendered in place (render_block): the gain rows, and a proven upper bound on the current list's length --
   * the length launch t reported plus the voices control actions have touched since launch t was issued */
  float *d_env_gain;
  size_t env_gain_cap;        /* floats */
  uint64_t touched_total;     /* voices named by control actions so far (every one of them goes on the list: sk_list_voice) */
  uint64_t bound_touched;     /* ... when the launch that reported bound_len was issued */
  uint32_t bound_len, bound_min_ticket;   /* reports of launches before bound_min_ticket (the last rebuild of the list) do not count */
  int bound_valid;
  int last_in_place;          /* the latest block took that path */
  int32_t *d_probe_ids;       /* skred_bank_set_probe: the probed voices on the device, their number, the caller's buffer */
  int n_probe;
  float *d_probe_out;
  int split_mode;             /* SKRED_OPT_SPLIT: 0 never, 1 where it is the faster form (default), 2 whenever the bank qualifies */
  int split_pairs;            /* SKRED_OPT_SPLIT_PAIRS: 0 the library's choice, 2 / 4 forced (tests) */
  int last_split;             /* the latest block ran sk_render_split_kernel */
  /* packed lanes of sparse banks (skred_device_layout.h: pack_mask; skred_bank.c: pack_refresh, render_block) */
  int pack_mode;              /* SKRED_OPT_PACK: 0 never, 1 where it pays (default) */
  int fm_skew;                /* SKRED_OPT_FM_SKEW: modulator lanes a block ahead of their carriers (default 1) */
  uint64_t *h_pack_mask;      /* [n_padded / 64] per aligned 64-voice group: voices that can sound, and the modulators they name */
  uint64_t *d_pack_mask;
  uint8_t *h_pack_dirty;      /* [n_padded / 64] a voice of the group changed class or modulator: its word is recomputed */
  int pack_any_dirty;
  int pack_hist[65];          /* groups per number of set bits in their word (the highest non-empty bin sizes the lane slots) */
  int pack_upload;            /* the device copy of the words is stale */
  int pack_zero;              /* a voice without a lane may hold a voice_sample the reference's skip rule would have cleared */
  int last_pack;              /* lanes per 64-voice group in the latest block (0: not packed) */
  int in_place_mode;          /* SKRED_OPT_IN_PLACE: 0 never, 1 where it is the faster path (default), 2 wherever the rows provably suffice */
  uint32_t violations_seen;   /* ... as last read back */
  hipStream_t side;           /* the envelope kernel's stream, beside the caller's */
  hipEvent_t ev_fork, ev_join;
  int last_family;            /* SKRED_KERNEL_* of the previous block (the list is only maintained while the two-per-lane family renders) */
  int max_level;
  int class_dirty;
  int cnt_real, cnt_filter, cnt_env, cnt_exotic, cnt_stops, cnt_fm;   /* voices per SKC_* bit (kept incrementally) */
  int cnt_guard;              /* real voices with SKC_GUARD */
  uint32_t tables_epoch, guard_epoch;   /* pools set so far; the pool the whole bank was last packed against (guard flags are a property of the pool) */
  int cnt_pair_ap;            /* pair-shaped carriers whose amplitude or pan is modulated too (SKC_PAIR_AP) */
  int cnt_fm_odd;             /* SKC_FM voices that are not the even half of a (carrier, next voice) pair (SKC_FM_ODD) */
  int cnt_escapes;            /* voices naming a modulator outside their aligned 64-voice group (SKC_ESCAPES) */
  int mod_dirty;              /* modulator lanes changed: dependency levels must be recomputed */
  uint32_t fast_mode;         /* SKM_* from classify() */
  int force_generic;          /* SKRED_OPT_FORCE_GENERIC */
  int fast2_min_voices;       /* SKRED_OPT_FAST2_MIN_VOICES */
  int fm2_min_voices;         /* SKRED_OPT_FM2_MIN_VOICES */
  int fast2_min_user;         /* ... was set by the caller (then it also applies to global-table banks) */
  int last_kernel;            /* SKRED_KERNEL_* used by the most recent render */
  skred_globals_t g;
  uint32_t features;
  hipEvent_t ev0[SK_TIMING_RING], ev1[SK_TIMING_RING]; /* around the render kernel of each call */
  int n_timed;                /* render calls since the last timing reset */
  /* what a launch found, reported by its final arriver into two pinned host words (no copy, no event): one-voice family --
   * "did an envelope move" (picks the lean instantiation: a speed hint); two-per-lane family -- the length of the list it
   * rendered, and the violation counter */
  uint32_t launch_ticket;     /* one per render launch */
  uint32_t control_epoch;     /* bumped by every upload / update / globals change */
  int env_quiet;              /* one-voice family: the last answered launch saw no envelope move and nothing changed since */
  volatile uint64_t *h_report;            /* pinned: [0] ticket << 32 | finding, [1] ticket << 32 | violations */
  uint32_t report_seen;                   /* ticket of the last report taken */
  uint32_t report_ticket[SK_REPORT_RING], report_epoch[SK_REPORT_RING];   /* what the launches that will report were issued under */
  uint8_t report_kind[SK_REPORT_RING];
  uint64_t report_touched[SK_REPORT_RING];   /* touched_total when the launch was issued */
  skred_seq_t *seq;             /* the pattern step clock (skred_seq.c), created on first use */
  struct sk_pat_step *pat;      /* [SKRED_PATTERNS_MAX][SKRED_SEQ_STEPS_MAX] batches the steps apply (skred_bank_update.c) */
  float seq_rate;               /* sample rate the step clock counts blocks in (0: the reference's 44100) */
  struct sk_queue_item *queue;  /* deferred updates (skred_bank_update.c), singly linked in arrival order */
  struct sk_queue_item *queue_tail;
  int queue_len;
  sk_upd_slot_t upd[SK_UPD_RING];   /* staging ring of the update path (skred_bank_update.c) */
  uint32_t upd_head;
  volatile uint32_t *h_upd_done;    /* pinned [SK_UPD_RING]: the sequence number of the last batch each slot's kernels have finished with */
  uint32_t *d_upd_cnt;              /* device [SK_UPD_RING]: arrival counters of those kernels' workgroups */
  uint32_t upd_seq;
  uint32_t *upd_mark, upd_epoch;    /* per-voice epoch marks: duplicate voices inside one batch */
};

/* per-voice classification (host shadow) */
#define SK_INPLACE_WORD_ROWS 8                   /* gain rows every 64-voice word of the list owns */
#define SK_INPLACE_DENOM 6                       /* lists up to n_voices / 6 are rendered in place (a 128-voice wave stages at most 32) */
#define SK_INPLACE_MAX_BYTES ((size_t)4 << 30)   /* ... while the gain rows stay below this */
#define SKC_REAL   1u   /* a voice was uploaded into this slot and it can sound (has a table) */
#define SKC_FILTER 2u
#define SKC_ENV    4u
#define SKC_EXOTIC 8u   /* needs the generic kernel: see classify() */
#define SKC_FM    32u   /* carrier of a higher-indexed modulator of its 64-voice group, nothing else modulated */
#define SKC_STOPS 16u   /* one-shot without loop (plays to its table end and finishes) or reverse playback: the one-per-lane kernel's extended instantiation */
#define SKC_FM_ODD 256u /* an SKC_FM voice that is anything but: even index, frequency-modulated by the voice after it and by nothing else
                          -- the shape sk_render_fast2_kernel<FMP> renders with carrier and modulator in one lane */
#define SKC_PAIR_AP 512u /* a pair-shaped carrier whose amplitude or pan is modulated (by the voice after it or by itself) */
#define SKC_GUARD 64u   /* loops over its whole table with a guard sample behind it (SKF_GUARD): when every real voice does, the
                           linear lookup runs the instantiations without the fold test */
#define SKC_LIVE 1024u  /* can sound as far as its parameters say: a usable table and voice_amp != 0 (synth.c:537; a voice that has
                           finished is a matter of state, not of class) */
#define SKC_ESCAPES 128u /* names a modulator outside its aligned 64-voice group: the bank cannot be rendered until that is fixed */


int skred_amd_set_error(int code, const char *fmt, ...);
#define fail skred_amd_set_error
#define HIP_TRY(call)                                                              \
  do {                                                                             \
    hipError_t e_ = (call);                                                        \
    if (e_ != hipSuccess)                                                          \
      return fail(SKRED_E_NO_DEVICE, "%s -> %s", #call, hipGetErrorString(e_));   \
  } while (0)

/* One voice of the host view -> its device planes.  Pure: nothing in the bank changes; what the voice
 * means for kernel selection comes back in `meta` and is applied by sk_apply_meta() when the planes are
 * written.  phase_known: the host's voice_phase is the voice's current phase (an upload); 0 when only
 * parameters are being pushed and the phase lives on the device. */
typedef struct {
  uint16_t cls;         /* SKC_* */
  int8_t mod_lane[4];   /* modulator lane inside the 64-voice group (fm, am, pan, cz) or -1 */
  uint32_t features;    /* SKB_* this voice needs */
} sk_voice_meta_t;

int sk_pack_voice(const skred_bank_t *b, const skred_voice_bank_t *h, int v, int dst, int phase_known,
                  sk_plane_t ro[SKP_COUNT], sk_plane_t rw[SKS_COUNT], sk_voice_meta_t *meta);
/* params_travel: the parameter planes were written, so `meta` applies (a pure clock / state update leaves the classes alone) */
void sk_apply_meta(skred_bank_t *b, int dst, const sk_voice_meta_t *meta, int params_travel);
/* the pipelined multi-GPU form's two halves of a block (skred_bank.c; used by skred_shard.c) */
int sk_bank_render_sum_pp(skred_bank_t *b, int num_frames, int interp, float *d_sum, int parity, void *stream);
int sk_bank_master_pp(skred_bank_t *b, const float *d_sum, int num_frames, int num_channels, float *d_out, int parity, void *stream);
void sk_queue_free(skred_bank_t *b);
void sk_patterns_free(skred_bank_t *b);
/* a control action reached the bank: what earlier launches reported about envelope activity no longer holds (the voices it
 * touched went on the motion list on the device: skred_update_kernels.hip) */
static inline void sk_control_changed(skred_bank_t *b) { b->control_epoch++; b->env_quiet = 0; b->list_empty = 0; }

#endif
