/*
 * include/skred_amd.h -- C ABI of the MI355X-native skred render path.
 *
 * One hot path is implemented: the per-voice render loop of skred's audio
 * callback, `synth()` (reference synth.c:502-630, declared synth.h:8, called
 * only from the miniaudio data_callback `synth_callback`, skred.c:107-116).
 *
 * Two boundaries are exported by libskred_amd.so:
 *
 *  (1) BANK MODE (this header): a runtime-N "voice bank" whose per-voice fields
 *      carry the reference's own names and types (synth.def:12-89; structs
 *      synth-types.h:13-38) but with N voices instead of VOICE_MAX=64
 *      (skred.h:9).  State lives in HBM between calls; every call renders
 *      `num_frames` frames for all voices with hand-written HIP kernels.
 *
 *  (2) DROP-IN MODE (include/skred_synth_abi.h): the literal synth.h surface
 *      -- `synth()`, the setters and the 75 global arrays -- backed by (1).
 *
 * Plain pointers and sizes only: no C++ or torch types cross this boundary.
 * All functions return 0 on success or a negative SKRED_E_* code; the product
 * path has NO CPU fallback: without a usable HIP device every entry point that
 * would touch the GPU fails with SKRED_E_NO_DEVICE.
 */
#ifndef SKRED_AMD_H
#define SKRED_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SKRED_AMD_ABI_VERSION 1

/* ---- error codes ------------------------------------------------------- */
enum {
  SKRED_OK = 0,
  SKRED_E_NO_DEVICE = -1,   /* no HIP device / HIP call failed (see skred_amd_last_error) */
  SKRED_E_BAD_ARG = -2,
  SKRED_E_NO_MEM = -3,
  SKRED_E_RANGE = -4,       /* voice window or table offset outside the bank / pool */
  SKRED_E_UNSUPPORTED = -5, /* feature of synth() not implemented by the kernels (see flags) */
  SKRED_E_IO = -6,          /* file cannot be opened / written (skred_wav.h) */
};

/* ---- structs kept from the reference (layout-identical) ---------------- */

/* == mmf_t, synth-types.h:13-23 (48 bytes): RBJ biquad state + coefficients */
typedef struct {
  float x1, x2;             /* input delay line  */
  float y1, y2;             /* output delay line */
  float b0, b1, b2;         /* feed-forward      */
  float a1, a2;             /* feedback          */
  float last_freq, last_resonance;
  int32_t last_mode;
} skred_mmf_t;

/* == envelope_t, synth-types.h:25-38 (56 bytes): linear ADSR keyed on the global sample counter */
typedef struct {
  float a, d, s, r;         /* seconds (control path only) */
  float attack_time;        /* samples */
  float decay_time;         /* samples */
  float sustain_level;      /* 0..1    */
  float release_time;       /* samples */
  uint64_t sample_start;    /* synth_sample_count at note-on  */
  uint64_t sample_release;  /* synth_sample_count at note-off, 0 = held */
  int32_t is_active;
  float velocity;
} skred_envelope_t;

/* ---- the voice bank (host view) --------------------------------------- */

/*
 * Host-side structure-of-arrays view of N voices.  Every member is the array
 * of the same name in synth.def:12-89 with `VOICE_MAX` replaced by
 * `n_voices`; the one exception is `voice_table`, a raw `float*` in the
 * reference (synth.def:14), which becomes `voice_table_offset`: the index of
 * the table's first sample inside the table pool handed to
 * skred_bank_set_tables_f32().
 *
 * Only the fields synth() reads or writes are present.
 * A bank view may simply point at the 64-entry global arrays of the drop-in
 * facade -- that is how drop-in mode is implemented.
 */
typedef struct skred_voice_bank {
  int32_t n_voices;

  /* oscillator (osc_next, synth.c:217-275) */
  float   *voice_phase;            /* rw */
  float   *voice_phase_inc;
  int64_t *voice_table_offset;     /* replaces float *voice_table[] */
  int32_t *voice_table_size;
  int32_t *voice_one_shot;
  int32_t *voice_finished;         /* rw */
  int32_t *voice_loop_enabled;
  int32_t *voice_loop_valid;
  float   *voice_loop_start_f;
  float   *voice_loop_end_f;
  int32_t *voice_direction;
  int32_t *voice_wave_table_index; /* only compared with WAVE_TABLE_NOISE_ALT (synth.c:543) */

  /* sample chain (synth.c:560-593) */
  float   *voice_sample;           /* rw */
  float   *voice_sample_hold;      /* rw */
  int32_t *voice_sample_hold_count;/* rw */
  int32_t *voice_sample_hold_max;
  int32_t *voice_quantize;
  float   *voice_amp;
  int32_t *voice_use_amp_envelope;
  int32_t *voice_smoother_enable;
  float   *voice_smoother_gain;    /* rw */
  float   *voice_smoother_smoothing;
  int32_t *voice_filter_mode;
  skred_mmf_t      *voice_filter;        /* rw: x1 x2 y1 y2 */
  skred_envelope_t *voice_amp_envelope;  /* rw: is_active   */

  /* pan / mix (synth.c:595-612) */
  float   *voice_pan_left;         /* rw only under pan modulation */
  float   *voice_pan_right;
  int32_t *voice_disconnect;

  /* cross-voice modulation + phase distortion (synth.c:548-558,584-587,597-602,262-267) */
  int32_t *voice_freq_mod_osc;
  float   *voice_freq_mod_depth;
  float   *voice_freq_scale;
  int32_t *voice_amp_mod_osc;
  float   *voice_amp_mod_depth;
  int32_t *voice_pan_mod_osc;
  float   *voice_pan_mod_depth;
  int32_t *voice_cz_mod_osc;
  float   *voice_cz_mod_depth;
  int32_t *voice_cz_mode;
  float   *voice_cz_distortion;
} skred_voice_bank_t;

/* Scalars synth() keeps outside the per-voice arrays. */
typedef struct skred_globals {
  uint64_t synth_sample_count;       /* synth.c:85; pre-incremented per frame (synth.c:521) */
  uint64_t noise_rng;                /* synth()'s static LCG state (synth.c:504,508,525)  */
  float volume_final;                /* synth.c:90  = volume_user * AMY_FACTOR            */
  float volume_smoother_gain;        /* synth.c:91  rw                                     */
  float volume_smoother_smoothing;   /* synth.c:92                                         */
  float reserved;
} skred_globals_t;

#define SKRED_WAVE_TABLE_NOISE_ALT 6 /* skred.h:30 */

/* render flags */
enum {
  SKRED_INTERP_TRUNCATE = 0, /* table[(int)phase] -- what the reference does (synth.c:268-274) */
  SKRED_INTERP_LINEAR   = 1, /* north-star mode; defined by oracle/cpu_ref.c, not by the reference */
};

/* ---- device-side bank -------------------------------------------------- */

typedef struct skred_bank skred_bank_t; /* opaque; owns HBM state for n voices on one GPU */

int  skred_amd_abi_version(void);
int  skred_amd_device_count(void);                 /* <=0: no usable GPU */
const char *skred_amd_last_error(void);            /* thread-local text of the last failure */

/* The largest bank a GPU holds: voice and list indices inside the kernels are 32-bit (a bank of this size takes 3.2 GB of
 * HBM); skred_bank_create() refuses more with SKRED_E_RANGE before it touches the device. */
#define SKRED_MAX_VOICES (1 << 24)
int  skred_bank_create(int device, int n_voices, skred_bank_t **out);
void skred_bank_destroy(skred_bank_t *bank);
int  skred_bank_n_voices(const skred_bank_t *bank);

/* Table pool: every table a voice can name, concatenated (floats).  Replaces the
 * malloc'd wave_table_data[] tables (synth.def:1, synth.c:1224).  Tables that fit
 * are staged into LDS by the kernel; larger pools are gathered from HBM/L2. */
/* Guard samples (optional, linear interpolation only): when a table is followed in the pool by one more float equal to its
 * first sample, a voice that loops over the whole table (no loop window) finds the second tap of the linear lookup in the next
 * float at every index; a bank in which every voice does runs the lookup without the fold test at the loop end (same samples,
 * ~1.3x the throughput on LUT banks).  Pools without guards render the same, through the general form. */
int  skred_bank_set_tables_f32(skred_bank_t *bank, const float *pool, size_t n_floats);

/* Pack `count` voices starting at host index `src_first` into device slots
 * [dst_first, dst_first+count).  Host arrays stay the source of truth.
 * Synchronous: waits for all work queued on the device (renders on any stream) before the planes are overwritten;
 * so do skred_bank_set_globals, skred_bank_get_globals and skred_bank_download. */
int  skred_bank_upload(skred_bank_t *bank, const skred_voice_bank_t *host,
                       int src_first, int dst_first, int count);
/* Copy the read-write fields (marked rw above) back into the host view. */
int  skred_bank_download(skred_bank_t *bank, skred_voice_bank_t *host,
                         int src_first, int dst_first, int count);

int  skred_bank_set_globals(skred_bank_t *bank, const skred_globals_t *g);
int  skred_bank_get_globals(skred_bank_t *bank, skred_globals_t *g);

/*
 * Render `num_frames` frames of every voice (the two nested loops of
 * synth.c:520-613) and leave this GPU's PRE-master-volume stereo sum in
 * `d_partial` (device pointer, float[num_frames][2]): one kernel launch (plus,
 * while notes ramp on the two-voices-per-lane path, the envelope kernel beside it on
 * a stream of the bank's own, joined back into `stream` before the call returns).  `d_stems` (device,
 * float[num_frames][n_voices][2], the `user` buffer layout of synth.c:533-534,
 * 607-611) may be NULL.  Advances synth_sample_count and the noise LCG.
 * `stream` is a hipStream_t (NULL = default stream).  Asynchronous.
 */
int  skred_bank_render(skred_bank_t *bank, int num_frames, int interp,
                       float *d_partial, float *d_stems, void *stream);

/*
 * Master volume stage (synth.c:616-624): serial one-pole smoothing of the
 * gain, multiply, and interleave into `d_out` (device, float[num_frames]
 * [num_channels], channels 0 and 1 written).  In a multi-GPU run this is
 * called on the root after the RCCL sum of the partials.  Asynchronous.
 */
int  skred_bank_master(skred_bank_t *bank, const float *d_sum, int num_frames,
                       int num_channels, float *d_out, void *stream);

/* Single-GPU form of render + master in ONE launch: the render kernel's last-arriving workgroups add the
 * per-workgroup rows up and apply the master gain (same samples as skred_bank_render + skred_bank_master:
 * both forms add the rows in the same fixed order).  `d_out` as for skred_bank_master.  Asynchronous. */
int  skred_bank_render_mix(skred_bank_t *bank, int num_frames, int interp, float *d_out, int num_channels,
                           float *d_stems_or_null, void *stream);

/* Whole synth() contract on host buffers: render + master + D2H (+ stems). Synchronous. */
int  skred_bank_render_host(skred_bank_t *bank, float *buffer, int num_frames,
                            int num_channels, int interp, float *stems_or_null);

/* ---- block-granular updates of device-resident voices (SURVEY 8f "next" #4) -------------------------
 *
 * The reference's control path (wire.c:606-716 commands, seq.c pattern steps, deferred items) stores into the
 * per-voice arrays and the next audio block sees the change.  With the voices in HBM the host view
 * (skred_voice_bank_t, same array names) stays what control code writes to; afterwards it names the voices it
 * touched and which KIND of field, and only those travel.  Everything not named keeps the value the GPU last
 * computed -- a full skred_bank_upload() would overwrite the running phase, filter memory and smoother with
 * the host's stale copies.
 */
enum {
  SKRED_DIRTY_PARAMS        = 1u << 0, /* every parameter: phase_inc, amp, table / loop window, direction, flags, quantize,
                                          hold_max, envelope times + velocity, filter coefficients, smoother k,
                                          modulation routing and depths, cz mode -- but not the envelope clock */
  SKRED_DIRTY_PHASE         = 1u << 1, /* voice_phase, voice_finished                    (osc_trigger, synth.c:316-339) */
  SKRED_DIRTY_ENV_STATE     = 1u << 2, /* voice_amp_envelope.is_active */
  SKRED_DIRTY_PAN           = 1u << 3, /* voice_pan_left, voice_pan_right                (pan_set, synth.c:838-847) */
  SKRED_DIRTY_FILTER_STATE  = 1u << 4, /* voice_filter.x1 x2 y1 y2                       (mmf_init, synth.c:1015-1030) */
  SKRED_DIRTY_SMOOTHER      = 1u << 5, /* voice_smoother_gain */
  SKRED_DIRTY_HOLD          = 1u << 6, /* voice_sample_hold, voice_sample_hold_count */
  SKRED_DIRTY_SAMPLE        = 1u << 7, /* voice_sample */
  /* the two control actions whose stores depend on WHEN they run: the library stamps the bank's
   * synth_sample_count at application time, as the reference's functions read the global */
  SKRED_STAMP_TRIGGER       = 1u << 8, /* amp_envelope_trigger (synth.c:383-388): sample_start = now, sample_release = 0,
                                          is_active = 1; send the velocity with SKRED_DIRTY_PARAMS */
  SKRED_STAMP_RELEASE       = 1u << 9, /* amp_envelope_release (synth.c:391-395): if is_active (device state), sample_release = now */
  SKRED_DIRTY_ENV_CLOCK     = 1u << 10, /* voice_amp_envelope.sample_start / .sample_release as the host has them (kept out
                                           of PARAMS so that a later parameter change cannot undo a stamped note-on / -off) */
  SKRED_DIRTY_VALID_MASK    = 0x7FF
};
#define SKRED_QUEUE_SIZE 1024          /* skred.h:86 QUEUE_SIZE */

/* Rewrite the `dirty` parts of the listed voices (indices into both the host view and the bank) from the host
 * view, on `stream` (a hipStream_t; pass the render stream: the update is ordered before the next render on it).
 * A voice may be listed more than once; the copies are applied in order. */
int  skred_bank_update(skred_bank_t *bank, const skred_voice_bank_t *host, const int32_t *voices, int n_voices,
                       uint32_t dirty, void *stream);

/* The deferred queue: seq.c:243-257 queue_item(when, what, voice) + the first loop of seq() (seq.c:170-177).
 * The reference stores command TEXT and runs it when due; here the values are captured from the host view when
 * the item is queued (only the STAMP actions read the clock when they run).  skred_bank_run_queue() is what
 * seq() does after synth(): every item with when <= synth_sample_count + frame_count is applied, in arrival
 * order, i.e. an item takes effect at the start of the block that contains its time.  Returns the number of
 * items applied (>= 0) or a SKRED_E_* code. */
int  skred_bank_defer(skred_bank_t *bank, uint64_t when, const skred_voice_bank_t *host, const int32_t *voices,
                      int n_voices, uint32_t dirty);
int  skred_bank_run_queue(skred_bank_t *bank, int frame_count, void *stream);
int  skred_bank_queue_pending(const skred_bank_t *bank);

/* ---- the pattern step clock: the other half of seq() (seq.c:179-213) ---------------------------------------------
 *
 * skred_seq_t is the reference's sequencer state on the host, with the same arithmetic: a double clock that gains
 * (float)frame_count / (float)rate per call and fires when it reaches tempo_time_per_step (tempo_set: 1 / (bpm / 60) / 4
 * seconds, four steps per beat; 60 s until a tempo is set), one step per call at most; per RUNNING pattern the modulo
 * (default 4) divides the step rate, a muted step advances silently, the pointer wraps at the first empty step.  No
 * device is involved: tests hold it against the compiled reference call by call.
 * A bank owns one (skred_bank_seq); a step of a bank's pattern is a batch of voice updates captured when it is written,
 * and skred_bank_run_queue() -- called once per block after the render, like seq() at skred.c:119 -- applies the
 * deferred items that are due and then the steps the clock fires, in pattern order, on `stream`. */
#define SKRED_PATTERNS_MAX 16          /* skred.h:75 */
#define SKRED_SEQ_STEPS_MAX 256        /* skred.h:76 */
enum { SKRED_SEQ_STOPPED = 0, SKRED_SEQ_RUNNING = 1, SKRED_SEQ_PAUSED = 2 };   /* skred.h:80-82 */
typedef struct skred_seq skred_seq_t;
int   skred_seq_create(skred_seq_t **out);
void  skred_seq_destroy(skred_seq_t *seq);
int   skred_seq_tempo_set(skred_seq_t *seq, float bpm);                       /* tempo_set, seq.c:21-28 */
float skred_seq_time_per_step(const skred_seq_t *seq);
int   skred_seq_step_set(skred_seq_t *seq, int pattern, int step, int occupied);   /* seq_step_set: empty text = not occupied */
int   skred_seq_mute_set(skred_seq_t *seq, int pattern, int step, int mute);  /* seq_mute_set */
int   skred_seq_modulo_set(skred_seq_t *seq, int pattern, int modulo);        /* seq_modulo_set */
int   skred_seq_state_set(skred_seq_t *seq, int pattern, int state);          /* seq_state_set: 0 stop, 1 start, 2 pause, 3 resume */
int   skred_seq_pattern_reset(skred_seq_t *seq, int pattern);                 /* pattern_reset */
int   skred_seq_pointer(const skred_seq_t *seq, int pattern);
int   skred_seq_counter(const skred_seq_t *seq, int pattern);
/* one call per block: fired[] receives (pattern << 16) | step of every step whose text the reference would run */
int   skred_seq_tick(skred_seq_t *seq, int frame_count, float sample_rate, int32_t *fired, int max_fired);

skred_seq_t *skred_bank_seq(skred_bank_t *bank);                    /* the bank's own clock (tempo, mute, modulo, state through skred_seq_*) */
int  skred_bank_set_sample_rate(skred_bank_t *bank, float rate);   /* the rate the clock counts blocks in; default 44100 (MAIN_SAMPLE_RATE, skred.h:6) */
int  skred_bank_pattern_step_set(skred_bank_t *bank, int pattern, int step, const skred_voice_bank_t *host,
                                 const int32_t *voices, int n_voices, uint32_t dirty);   /* n_voices == 0: a rest */
int  skred_bank_pattern_step_clear(skred_bank_t *bank, int pattern, int step);           /* empty step: the pattern wraps here */

/* Options.  The render loop has full-featured kernels (generic; modulated for banks with cross-voice modulation) and
 * specialised ones chosen per launch from what the bank holds (one voice per lane, two per lane);
 * their per-voice results, stems included, are bit-identical.  FORCE_GENERIC pins the full-featured kernels (the parity
 * tests use it to cross-check the specialised ones). */
enum { SKRED_OPT_FORCE_GENERIC = 1, SKRED_OPT_FAST2_MIN_VOICES = 2 /* bank size from which the two-voices-per-lane kernel is used */,
       SKRED_OPT_KERNEL_TIMING = 4 /* n: an event pair brackets the render kernels of every n-th launch (default 1: every
                                      launch; 0: none).  skred_bank_last_render_ms / _timing_summary report the bracketed
                                      launches; an event pair costs ~6 us of stream time on an MI355X, hence the knob */,
       SKRED_OPT_FM2_MIN_VOICES = 5 /* bank size from which a two-operator FM bank (every carrier an even voice, frequency-
                                       modulated by the voice after it and by nothing else) keeps carrier and modulator in
                                       one lane of the two-voices-per-lane kernel */,
       SKRED_OPT_IN_PLACE = 6 /* how the motion list of a two-voices-per-lane LDS-table bank is rendered while it is short: in the
                                 steady kernel's own lanes, from per-frame gain rows written by sk_gain_kernel just ahead of it
                                 ("in place"), or by the envelope kernel beside the steady one.  1 (default): in place where that
                                 is the faster path (sparse lists; bank sizes at which a second kernel costs the steady one a
                                 whole round of workgroups); 0: never; 2: whenever the rows provably suffice (tests).  Same
                                 per-voice results either way */,
       SKRED_OPT_SPLIT = 7 /* small clean LDS-table banks, while nothing moves: the one-voice-per-lane kernel with every frame split
                              between an oscillator wave and a post wave (sk_render_split_kernel).  0 (default): never -- measured at
                              best 1.4 % faster than the unsplit kernel; 1: banks of half a 256-voice group to one
                              group per CU with filters; 2: whenever the bank qualifies; 3: even while envelopes may be moving (tests:
                              the kernel then renders the waves concerned on its general path).  Same per-voice results either way */,
       SKRED_OPT_SPLIT_PAIRS = 8 /* (tests) the split form's workgroup shape: 0 (default) four (oscillator wave, post wave)
                                    pairs per workgroup; 2 / 4: forced (two: 256-thread workgroups whose four waves land on four SIMDs) */,
       SKRED_OPT_PACK = 9 /* sparse banks -- most voices skipped by the reference's own rule, voice_amp == 0 (synth.c:537), as in every
                             shipped patch (3 to 6 voices of 64 in use): the one-voice-per-lane kernel with the lanes PACKED, a
                             wavefront holding the voices that can sound of several aligned 64-voice groups (and the modulators
                             they name) instead of all 64 voices of one.  1 (default): on banks of at least 768 voices per CU, where at least half of
                             the wavefronts disappear (three quarters for banks the two-voices-per-lane kernel would take); 0: never;
                             2: whenever any disappear (tests, small banks).  The modulated kernel packs the same way.  Launches with the full stem buffer and banks on the generic / modulated / FM-pair
                             kernels are never packed.  Per-voice state is bit-identical either way; the mix differs by
                             summation order only */,
       SKRED_OPT_FM_SKEW = 10 /* previous-frame modulation on the one-voice-per-lane kernel (`v0 ... F3,1` / `A` / `P` with the modulator
                             above its carriers, synth.c:548-555,584-587,597-602, as in 3.sk / 1.sk / 7.sk / 37.sk / 0.sk): 1 (default) a lane
                             that is read by others runs 8-frame blocks AHEAD of its readers (chains up to three levels) and hands
                             its samples over through an LDS ring, so the readers' blocks have no per-frame exchange -- wavefronts
                             whose sources are silent (`m1`) or heard with their pan at rest, each exactly one block ahead of every
                             lane that reads it, and without reverse / noise / stopping / smoother-off lanes; other wavefronts keep
                             the exchange.  0: the per-frame ds_bpermute exchange everywhere.  The same option
                             governs the modulated kernel's FRAME-LAG form (a modulator BELOW its carrier -- a same-frame dependency,
                             18.sk -- with one dependency level: the dependent lanes run one frame behind instead of every frame being
                             rendered once per level).  Same bits either way */ };
enum { SKRED_KERNEL_GENERIC = 0, SKRED_KERNEL_FAST = 1, SKRED_KERNEL_MODULATED = 2, SKRED_KERNEL_FAST2 = 3 };
int  skred_bank_set_option(skred_bank_t *bank, int option, int value);
int  skred_bank_last_kernel(const skred_bank_t *bank);   /* SKRED_KERNEL_* of the latest render */
int  skred_bank_last_in_place(const skred_bank_t *bank);  /* 1: the latest block rendered its motion list in place (SKRED_OPT_IN_PLACE) */
int  skred_bank_last_pack(const skred_bank_t *bank);      /* lanes per 64-voice group in the latest block (SKRED_OPT_PACK), 0: not packed */
int  skred_bank_last_split(const skred_bank_t *bank);     /* 1: the latest block ran the split form of the one-voice kernel (SKRED_OPT_SPLIT) */

/* Per-frame evidence from INSIDE the fast paths (tests).  A launch with the full stem buffer takes the kernels' frame-by-frame
 * paths, so the 8-frame block paths the benchmarks time were only ever seen through end-of-block state and the mix.  A probe names
 * up to SKRED_PROBE_MAX voices; every later block writes, for each frame, what the reference stores into its stem buffer for them
 * (synth.c:607-611: voice_sample x pan_left, x pan_right; exact zeros for a skipped or muted voice) into
 * d_probe[frame][i][2] (device memory, num_frames x n x 2 floats per block, overwritten block by block) -- from the same launch,
 * on the same kernel paths, that render the block without it (probe instantiations of the one-voice-per-lane, two-voices-per-lane,
 * in-place and envelope kernels; other kernel families return SKRED_E_UNSUPPORTED while a probe is set).  n = 0 ends it. */
#define SKRED_PROBE_MAX 64
int  skred_bank_set_probe(skred_bank_t *bank, const int32_t *voices, int n, float *d_probe);

/* Cross-check of the motion list of the two-voices-per-lane path ("The motion list"): voices whose envelope may be
 * in motion are kept on a per-voice list ON THE DEVICE (every control action lists the voices it touches, the envelope kernel
 * keeps its voices listed until they rest) and rendered by the envelope kernel beside the steady kernel, which never has to be
 * told by the host whether anything moves.  The steady kernel still classifies every voice it renders; this is the number of
 * voices it ever found in motion without being listed, as far as reported (asynchronously).  0 by construction; should it move,
 * the list is rebuilt from the voice state by itself and skred_amd_last_error() names the launch. */
unsigned skred_bank_list_violations(const skred_bank_t *bank);

/* ---- voices sharded over the GPUs of one node (BASELINE config 3) -----------------------------------
 *
 * One process per GPU.  Rank r of `world` owns the contiguous block [lo, hi) of the bank's voices and renders its
 * PRE-master partial mix float[F][2]; the one exchange step of the path is the sum of those partials on `root` (one
 * RCCL reduce of 8*F bytes per block, over xGMI), after which the root applies the master volume stage once.  This is
 * the host side of BASELINE config 3 in C: what the reference's audio callback (skred.c:107-116: synth() then the
 * output) becomes when the voices of one bank live on several GPUs.  A cut is only legal where no voice is modulated
 * across it (skred_shard_cut_ok; skred_shard_upload refuses otherwise).
 *
 * The three steps of a block are function pointers so that the same sequencing runs elsewhere: skred_shard_create()
 * wires them to the bank-mode entry points and to ncclReduce (after skred_shard_init_rccl); a program with its own
 * communicator replaces `reduce`; the CPU tests (tests/test_sharded_gloo.py) supply all three. */
typedef struct skred_shard skred_shard_t;
typedef struct skred_shard_ops {
  void *ctx;            /* passed to render and master */
  int (*render)(void *ctx, int num_frames, int interp, float *partial, void *stream);   /* this rank's pre-master sum -> partial[F][2] */
  int (*master)(void *ctx, const float *sum, int num_frames, int num_channels, float *out, void *stream);   /* root only: synth.c:616-624 */
  void *reduce_ctx;     /* passed to reduce */
  int (*reduce)(void *reduce_ctx, float *partial, size_t n_floats, int root, void *stream);   /* sum over ranks, in place on the root */
} skred_shard_ops_t;

int  skred_shard_partition(int total_voices, int world, int rank, int *lo, int *hi);   /* blocks differ by at most one voice */
int  skred_shard_cut_ok(const skred_voice_bank_t *whole_bank, int lo, int hi);         /* 1: no modulation crosses the cut */
int  skred_shard_create(int device, int rank, int world, int root, int total_voices, skred_shard_t **out);   /* + this rank's bank on `device` */
int  skred_shard_create_custom(int rank, int world, int root, int total_voices, const skred_shard_ops_t *ops, skred_shard_t **out);
void skred_shard_destroy(skred_shard_t *shard);
skred_bank_t *skred_shard_bank(skred_shard_t *shard);        /* tables, globals, options, updates: through the bank ABI above */
int  skred_shard_range(const skred_shard_t *shard, int *lo, int *hi);
int  skred_shard_upload(skred_shard_t *shard, const skred_voice_bank_t *whole_bank);   /* this rank's block of the WHOLE bank */
int  skred_shard_set_ops(skred_shard_t *shard, const skred_shard_ops_t *ops, int always_reduce);   /* NULL members keep the current step */
/* RCCL owned by the library: rank 0 draws an id, the host program carries its 128 bytes to the other ranks, every
 * rank initialises with it (collective call). */
int  skred_shard_rccl_unique_id(void *out128);
int  skred_shard_init_rccl(skred_shard_t *shard, const void *unique_id128);
/* One block: render -> reduce -> master on the root.  `partial`: float[num_frames][2] scratch in the memory the steps
 * work on (NULL: the shard's own device scratch); `out` ([num_frames][num_channels]) is written on the root only.
 * Asynchronous on `stream` with the bank-backed steps. */
int  skred_shard_render_mix(skred_shard_t *shard, int num_frames, int interp, float *partial, float *out, int num_channels, void *stream);

/* The same block with the collective of block k overlapped with the render of block k + 1 (two partial buffers, the
 * collective and the root's master stage on a stream of the shard's own): throughput is bounded by max(render,
 * reduce + master) instead of their sum.  Host-paced: call k first waits (on the host) until block k - 2 has left the
 * collective's stream, so `out` of call k is complete -- for the host and for every stream -- once call k + 2 has
 * returned: alternate between two output buffers.  `stream` itself never waits for the collective's stream; to have
 * everything issued so far complete ON `stream` (e.g. to consume the latest block with a kernel or a copy queued
 * there) call skred_shard_flush(shard, stream).  Same samples as skred_shard_render_mix, bit for bit.  Custom steps
 * (host memory) run synchronously, in order. */
int  skred_shard_render_mix_pipelined(skred_shard_t *shard, int num_frames, int interp, float *out, int num_channels, void *stream);
int  skred_shard_flush(skred_shard_t *shard, void *stream);

/* Timing of the most recent skred_bank_render() on its stream, via hipEvents
 * recorded around the render kernel itself (ms; <0 if unavailable). Synchronises. */
float skred_bank_last_render_ms(skred_bank_t *bank);

/* The render kernel of every skred_bank_render() call is bracketed by a hipEvent pair on the
 * call's stream (ring of 256).  reset() starts a measurement window; summary() synchronises and
 * reports mean / min kernel duration (ms) over the calls since the reset (at most the last 256). */
void skred_bank_timing_reset(skred_bank_t *bank);
int  skred_bank_timing_summary(skred_bank_t *bank, float *mean_ms, float *min_ms, int *count);

#ifdef __cplusplus
}
#endif
#endif /* SKRED_AMD_H */
