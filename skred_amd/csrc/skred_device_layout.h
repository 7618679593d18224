/*
 * skred_device_layout.h -- how a voice bank is laid out in HBM.
 *
 * The host view (include/skred_amd.h: skred_voice_bank_t) keeps one array per reference
 * field (synth.def:12-89).  On the device the ~60 hot fields of a voice are packed into
 * 16-byte "planes": plane p is an array uint4[n_padded], so a wavefront of 64 consecutive
 * voices reads or writes one plane with ONE dwordx4 access per lane = 1 KiB contiguous.
 * Read-only planes (parameters) are loaded once per launch; read-write planes (the serial
 * recurrences: phase, smoother, biquad delay line ...) are loaded, kept in registers for
 * all frames of the launch, and stored back once.
 *
 * Shared between the C host shim (packing) and the HIP kernels (unpacking).
 */
#ifndef SKRED_DEVICE_LAYOUT_H
#define SKRED_DEVICE_LAYOUT_H

#include <stdint.h>

/* ---- read-only planes ---- */
enum {
  SKP_OSC = 0,   /* f32 phase_inc | f32 loop_lo | f32 loop_hi | f32 amp                              */
  SKP_TAB,       /* i32 table_offset | i32 table_size | u32 flags | u32 quantize | hold_max<<8        */
  SKP_ENV_T,     /* f32 attack_time | f32 decay_time | f32 sustain_level | f32 release_time          */
  SKP_ENV_S,     /* u32 sample_start lo,hi | u32 sample_release lo,hi                                 */
  SKP_GAIN,      /* f32 velocity | f32 smoother_smoothing | f32 b0 | f32 b1                           */
  SKP_FILT,      /* f32 b2 | f32 a1 | f32 a2 | f32 cz_distortion                                      */
  SKP_MODI,      /* i32 freq_mod_osc | i32 amp_mod_osc | i32 pan_mod_osc | i32 cz_mod_osc             */
  SKP_MODF,      /* f32 freq_mod_depth | f32 freq_scale | f32 amp_mod_depth | f32 pan_mod_depth       */
  SKP_MODX,      /* f32 cz_mod_depth | i32 cz_mode | 0 | 0                                            */
  SKP_COUNT
};

/* ---- read-write planes ---- */
enum {
  SKS_OSC = 0,   /* f32 phase | f32 smoother_gain | f32 x1 | f32 x2                                   */
  SKS_FILT,      /* f32 y1 | f32 y2 | f32 voice_sample | u32 rwflags                                  */
  SKS_MISC,      /* f32 sample_hold | i32 sample_hold_count | f32 pan_left | f32 pan_right            */
  SKS_COUNT
};

/* loop_lo / loop_hi are the window osc_next() actually uses (synth.c:235-238):
 *   loop_enabled && loop_valid ? loop_start_f : 0      and      ... ? loop_end_f : (float)table_size
 * -- a pure select, resolved on upload. */

/* flags word of SKP_TAB */
#define SKF_ONE_SHOT   (1u << 0)   /* voice_one_shot                                   */
#define SKF_LOOPING    (1u << 1)   /* voice_loop_enabled                               */
#define SKF_REVERSE    (1u << 2)   /* voice_direction                                  */
#define SKF_USE_ENV    (1u << 3)   /* voice_use_amp_envelope                           */
#define SKF_FILTER     (1u << 4)   /* voice_filter_mode != 0                           */
#define SKF_SMOOTH     (1u << 5)   /* voice_smoother_enable                            */
#define SKF_MUTED      (1u << 6)   /* voice_disconnect                                 */
#define SKF_NOISE      (1u << 7)   /* voice_wave_table_index == WAVE_TABLE_NOISE_ALT   */
#define SKF_INERT      (1u << 8)   /* padding voice, or a voice without a table: always skipped */
#define SKF_HAS_MOD    (1u << 9)   /* names another voice as FM/AM/pan/CZ modulator, or uses CZ */
#define SKF_GUARD      (1u << 10)  /* the voice loops over its WHOLE table (lo == 0, hi == table_size) and the pool holds a guard sample behind
                                      the table, equal to its first one: the second tap of the linear lookup is always the next float, the
                                      fold at the loop end (oracle/cpu_ref.c: table_fetch) needs no test.  A property of the caller's pool,
                                      checked per voice when it is packed (skred_bank_update.c) */

/* rwflags word of SKS_FILT */
#define SKR_FINISHED   (1u << 0)   /* voice_finished                 */
#define SKR_ENV_ACTIVE (1u << 1)   /* voice_amp_envelope.is_active   */

/* bank-wide feature summary (OR over voices), selects kernel paths */
#define SKB_ANY_NOISE  (1u << 0)
#define SKB_ANY_FILTER (1u << 1)
#define SKB_ANY_ENV    (1u << 2)
#define SKB_ANY_HOLDQ  (1u << 3)   /* sample&hold or bit-crush somewhere */
#define SKB_ANY_MOD    (1u << 4)   /* modulation that needs sk_render_mod_kernel (same-frame dependencies, AM, pan, CZ) */
#define SKB_ANY_FM     (1u << 5)   /* frequency modulation by a higher-indexed voice of the group only (previous-frame
                                      semantics): the one-per-lane kernel can do it; sk_render_mod_kernel otherwise */

/* fast_mode word (host -> launcher) */
#define SKM_FAST        (1u << 0)  /* bank qualifies for sk_render_fast_kernel (see skred_bank.c:classify) */
#define SKM_FILTER_ALL  (1u << 1)  /* every live voice runs the biquad (with SKM_MIXED: some do) */
#define SKM_ENV_ALL     (1u << 2)  /* every live voice uses the amp envelope (with SKM_MIXED: some do) */
#define SKM_TWO_PER_LANE (1u << 3) /* large bank: sk_render_fast2_kernel (two voices per lane, packed fp32) */
#define SKM_FM          (1u << 5)  /* some voice is frequency-modulated by a higher-indexed voice of its 64-voice group and
                                      nothing else is modulated (sk_render_fast_kernel<STOPS>: previous-frame exchange by ds_bpermute) */
#define SKM_MIXED       (1u << 6)  /* the biquad and / or the envelope is used by some voices only: per-lane flags
                                      (sk_render_fast_kernel's extended instantiation) */
#define SKM_FM_PAIR     (1u << 7)  /* SKM_FM and every carrier is an even voice modulated by the next voice only, no SKM_STOPS features:
                                      sk_render_fast2_kernel<FMP> keeps carrier and modulator in one lane (no exchange, no votes) */
#define SKM_PAIR_AP     (1u << 8)  /* SKM_FM_PAIR and some carrier's amplitude or pan is modulated too (by the voice after it or by itself) */
#define SKM_SPLIT       (1u << 9)  /* one-voice family, clean LDS-table bank believed steady: sk_render_split_kernel (skred_render_split.hip) --
                                      every 64 voices get an oscillator wave and a post wave, so that small and mid-size banks give a SIMD
                                      twice the independent instruction streams */
#define SKM_SPLIT2      (1u << 10) /* ... with SKM_SPLIT: two pairs per workgroup (256 threads, 128 voices per pass: args->n_rows counts those) */
#define SKM_STOPS       (1u << 4)  /* some voice is a forward one-shot that finishes at its table end (sk_render_fast_kernel<STOPS>) */

#define SK_GROUP 256               /* voices per workgroup pass (4 wavefronts) */
#define SK_CHUNK 64                /* frames between two workgroup-level mix flushes */
#ifndef SK_FAST2_NW_LDS
#define SK_FAST2_NW_LDS 8   /* sk_render_fast2_kernel, LDS-table banks: wavefronts (128-voice slices) per workgroup pass */
#endif
#define SK_LDS_TABLE_MAX_FLOATS 12320  /* 48 KiB + the pad the host appends (SK_TABLE_PAD): pools up to this size are staged in LDS --
                                          three of the reference's 4096-entry built-in waves still fit */
#define SK_MAX_WORKGROUPS 2048
#define SK_WIN 20                  /* floats of one voice's table window (skred_render_fast2.hip: 8 frames at up to
                                      2.1875 table samples per frame, plus the second tap) */
#define SK_TABLE_PAD 24            /* zero floats the host appends to the pool: a window (or the two-tap gather)
                                      may start at the last sample of the last table */

typedef struct {
  uint32_t w[4];
} sk_plane_t;

/* ---- block-granular voice updates (skred_bank_update.c -> skred_update_kernels.hip) ----
 * One record per touched voice: all planes as the host packed them, and which parts to write.  The bits equal
 * SKRED_DIRTY_* of include/skred_amd.h (checked at compile time in skred_bank_update.c). */
#define SKU_PARAMS        (1u << 0)   /* all read-only planes except SKP_ENV_S */
#define SKU_PHASE         (1u << 1)   /* SKS_OSC.w0, SKR_FINISHED */
#define SKU_ENV_STATE     (1u << 2)   /* SKR_ENV_ACTIVE */
#define SKU_PAN           (1u << 3)   /* SKS_MISC.w2, w3 */
#define SKU_FILTER_STATE  (1u << 4)   /* SKS_OSC.w2, w3; SKS_FILT.w0, w1 */
#define SKU_SMOOTHER      (1u << 5)   /* SKS_OSC.w1 */
#define SKU_HOLD          (1u << 6)   /* SKS_MISC.w0, w1 */
#define SKU_SAMPLE        (1u << 7)   /* SKS_FILT.w2 */
#define SKU_STAMP_TRIGGER (1u << 8)   /* envelope: sample_start = now, sample_release = 0, is_active = 1 (synth.c:383-388) */
#define SKU_STAMP_RELEASE (1u << 9)   /* envelope: if is_active, sample_release = now (synth.c:391-395) */
#define SKU_ENV_CLOCK     (1u << 10)  /* SKP_ENV_S: sample_start, sample_release as the host has them */

typedef struct {
  int32_t voice;
  uint32_t dirty;
  uint32_t pad[2];
  sk_plane_t ro[SKP_COUNT];
  sk_plane_t rw[SKS_COUNT];
} sk_update_t;

/* kernel arguments of the render kernel */
typedef struct {
  const sk_plane_t *ro[SKP_COUNT];
  sk_plane_t *rw[SKS_COUNT];
  const float *tables;     /* HBM pool */
  float *partial;          /* [n_workgroups][num_frames][2] pre-master partial sums */
  float *stems;            /* [num_frames][n_voices][2] or NULL */
  int32_t *group_flag;     /* [n_groups*2]: voices of each 128-voice wave slice on the motion list (written by sk_collect_scan_kernel) */
  const uint64_t *mask_cur;/* [n_groups*4]: THE MOTION LIST as a bit per voice (voice v: word v >> 6, bit v & 63): voices whose
                              envelope may be in motion during this block (attack / decay / release, a note-on ahead of the clock,
                              a smoother still settling).  sk_render_fast2_kernel sits these voices out, sk_render_env2_kernel
                              renders them -- beside it, on a second stream.  Carried from block to block on the device:
                              sk_update_kernel / sk_stamp_kernel set the bit of every voice they touch, sk_classify_kernel
                              rebuilds it after uploads and clock changes, and sk_render_env2_kernel carries its voices
                              over into mask_next until they have come to rest.  Read-only during a block. */
  uint64_t *mask_next;     /* the list of the NEXT block: zeroed by sk_collect_scan_kernel, survivors OR-ed in by sk_render_env2_kernel */
  int32_t *env_off;        /* [n_groups*2 + 1]: exclusive prefix sums of the counts, [n_groups*2] = their total
                              (sk_collect_scan_kernel, ahead of sk_render_env2_kernel on its stream) */
  int32_t *env_list;       /* [n_groups*SK_GROUP]: the listed voices in ascending order (sk_collect_expand_kernel) */
  uint32_t *moved;         /* [n_rows] one-voice family: the ticket of the last launch in which workgroup (row) i saw an envelope move */
  unsigned long long *report; /* [2] in pinned HOST memory, written by the block's final arriver (one writer, one 8-byte store each):
                              (launch_ticket << 32) | what the launch found -- one-voice family: 1 when an envelope moved; two-per-lane
                              family: [0] the length of the motion list it rendered, [1] the violation counter.  The host polls the
                              words: no copy, no event */
  uint32_t *violations;    /* [1] sticky: voices sk_render_fast2_kernel found with an envelope in motion that were NOT on the list
                              (unreachable by construction; the host rebuilds the list when it ever reads non-zero) */
  uint64_t count0;         /* synth_sample_count before the first frame */
  uint64_t rng0;           /* noise LCG state before the first frame */
  int32_t n_voices;        /* real voices (stems indexing) */
  int32_t n_groups;        /* n_padded / SK_GROUP */
  int32_t num_frames;
  int32_t table_floats;    /* pool size */
  int32_t lds_table_floats;/* floats staged in LDS (0 or == table_floats) */
  int32_t interp;
  uint32_t features;       /* SKB_* */
  uint32_t fast_mode;      /* SKM_* : which specialised kernel the host picked */
  uint32_t launch_ticket;  /* this launch's number (see group_flag) */
  uint32_t skip_env2;      /* two-per-lane family: the motion list is EMPTY (a structural fact: the last list the device built was
                              empty and nothing was added since), the masks are not read and sk_render_env2_kernel is not
                              launched.  One-voice family: a launch reported that no envelope moved (picks the lean
                              instantiation; both render everything, so this one is a pure speed hint) */
  /* ---- the envelope kernel beside the steady kernel (two-per-lane family): its own rows, its own arrival ticket; its last
   * arriver adds the rows that were used into env_sum and then arrives at the block's final ticket like one more slab ---- */
  int32_t env_beside;      /* this block has an envelope kernel: the final ticket expects one more arrival, the final sum adds env_sum */
  int32_t n_env_rows;      /* workgroups of sk_render_env2_kernel (each strides over the list's 512-voice passes) */
  float *env_rows;         /* [n_env_rows][num_frames][2] */
  float *env_sum;          /* [num_frames][2] */
  uint32_t *env_ticket;    /* [1] */
  /* ---- listed voices rendered IN PLACE (two-per-lane family, sparse lists): sk_gain_kernel, ahead of the steady kernel on the
   * same stream, evaluates the envelope of every listed voice for every frame of the block and leaves the gains in a row of
   * env_gain; the steady kernel's in-place instantiation keeps such a voice in its lane and feeds its smoother from that row
   * instead of a constant.  No second kernel beside it, no list to collect. ---- */
  float *env_gain;         /* [env_gain_cap][env_gain_stride]: row r = amp * (level * velocity) (synth.c:582; what the voice's amp smoother
                              is fed) of every frame of the block for the voice that holds row r (env_list[voice] = r, handed out
                              by sk_gain_kernel); NULL: the envelope kernel renders the list */
  int32_t env_gain_stride; /* floats per row: num_frames + 8 (the steady kernel fetches one 8-frame block ahead) */
  int32_t env_gain_cap;    /* rows: n_groups * 4 * env_word_rows that the 64-voice words of the list own, then the overflow area */
  int32_t env_word_rows;   /* rows every 64-voice word owns (a word with more listed voices takes rows from the overflow area) */
  uint32_t *env_count;     /* [0] the length of the list this block rendered (counted by the steady kernel's waves), [1] overflow rows
                              handed out by sk_gain_kernel; the block's final arriver reports [0] and sets both back to 0 */
  /* ---- the block's mix-down, inside the last render kernel of the block (skred_kernel_common.hpp: sk_finish_block) ----
   * Every workgroup leaves its partial-mix row in `partial`; with `finish` set the workgroup that arrives LAST at a
   * ticket adds the rows up in a fixed order (slab by slab when there are many), writes the pre-master sum and, in
   * the single-GPU form, applies the master volume -- one launch per block, no reduction kernels behind it. */
  int32_t n_rows;          /* workgroups that render (rows of `partial`); the grid has wg_shift more */
  int32_t wg_shift;        /* 1: workgroup 0 is the gain workgroup (walks the master-volume recurrence), renderers are blockIdx.x - 1 */
  int32_t finish;          /* this kernel is the last one of the block that writes rows */
  int32_t num_channels;    /* of mix_out */
  float *slab_rows;        /* [SK_FINISH_SLABS][num_frames][2]: sums of the rows w = slab (mod SK_FINISH_SLABS), ascending */
  float *sum_out;          /* [num_frames][2] pre-master sum of all rows (the operand of the multi-GPU reduce), or NULL */
  float *mix_out;          /* [num_frames][num_channels] post-master output (channels 0, 1 written), or NULL */
  float *gains;            /* [num_frames] master gain per frame (synth.c:616-620), written by the gain workgroup */
  float *gain_state;       /* the smoother's carried gain as of frame 0 (read) */
  float *gain_commit;      /* where the gain after the last frame goes: gain_state itself when this launch applies the master
                              stage; a pending slot when it only prepares `gains` for skred_bank_master (multi-GPU form), which
                              commits it */
  uint32_t *tickets;       /* [SK_FINISH_SLABS + 1] arrival counters; the last arriver re-arms its counter to 0 */
  float vol_target, vol_k; /* volume_final, volume_smoother_smoothing */
  /* ---- per-frame probes (tests; skred_bank_set_probe): the (L, R) of every frame of up to SK_PROBE_MAX voices, written from INSIDE
   * the kernels' block paths by their probe instantiations (translation units compiled with -DSK_PROBE_TU; the ordinary
   * instantiations contain none of this) ---- */
  const int32_t *probe_ids; /* [n_probe] voice numbers */
  float *probe_out;         /* [num_frames][n_probe][2], zeroed by the host before the launch (a skipped / muted voice writes nothing) */
  int32_t n_probe;
  /* ---- packed lanes (sparse banks; skred_bank.c: render_block decides): most voices of the bank are skipped by the reference's own
   * rule (voice_amp == 0, synth.c:537) for as long as nobody changes them, so a wave of the one-voice family takes the voices
   * that CAN sound of 2^(6 - pack_shift) aligned 64-voice groups instead of all 64 voices of one -- group j of the wave owns lanes
   * [j << pack_shift, (j + 1) << pack_shift), the k-th set bit of pack_mask[group] is the voice in its k-th lane.  A bit is set for
   * every voice that can sound and for every voice a voice that can sound names as a modulator (those keep a lane so that the
   * exchange finds them: skipped at run time like any dead voice).  pack_shift == 6: off. ---- */
  const uint64_t *pack_mask; /* [pack_groups] */
  int32_t pack_shift;        /* log2 of the lanes per group */
  int32_t pack_groups;       /* aligned 64-voice groups of the bank */
  int32_t pack_passes;       /* workgroup passes: ceil(pack_groups / (4 << (6 - pack_shift))) */
  /* ---- skewed blocks of frequency-modulated wavefronts (one-voice family, extended instantiation; SKRED_OPT_FM_SKEW) ---- */
  int32_t fm_skew;           /* 1: the launch carries the per-wave sample ring (SK_SKEW_RING floats) behind the reduction tiles */
} sk_render_args_t;
#define SK_PROBE_MAX 64

#define SK_FINISH_SLABS 32     /* slabs of the two-level mix-down (a multiple of 8: the workgroups of a slab share an XCD's L2
                                  under round-robin placement -- speed only) */
#define SK_FINISH_FLAT_MAX 64  /* up to this many rows the last arriver adds the rows directly */

#endif
