/*
 * include/skred_synth_abi.h -- DROP-IN MODE: the reference's synth.h surface, re-declared.
 *
 * libskred_synth.so (skred_amd/csrc/skred_synth_dropin.c) defines every data symbol and every
 * control-path function that the rest of skred links against in place of synth.o
 * (reference Makefile:88-100), and a synth() whose render loop runs on the GPU through
 * libskred_amd.so.  The reference's own wire.c / seq.c / skred.c objects link against it
 * unchanged (tests/test_dropin.py does exactly that).
 *
 * Names, element types and array lengths below ARE the ABI: they restate
 *   synth.h:8-85      function prototypes
 *   synth.h:12-22     scalar globals
 *   synth.def:1-10    per-wave-slot arrays   [WAVE_TABLE_MAX = 1200, skred.h:78-81]
 *   synth.def:12-89   per-voice arrays       [VOICE_MAX = 64, skred.h:9]
 *   synth-types.h     mmf_t / envelope_t (here skred_mmf_t / skred_envelope_t, same layout)
 *
 * The per-sample functions synth.h also declares (osc_next, cz_phasor, quantize_bits_int,
 * mmf_process, amp_envelope_step, audio_rng_init/next/float; synth.h:24-26,29,30,33,38,43) are
 * exported too (skred_amd/csrc/skred_synth_persample.c): one voice, one sample, on the host arrays.
 * synth() never calls them -- the render loop is the HIP kernels' -- and no other translation unit
 * of the reference does; they complete the header's surface.
 *
 * Imported from the host program, as in the reference (all optional here, weak):
 *   int debug (skred.c:53), float *mw_free(float*) (miniwav.c:94),
 *   pcm_map_t pcm_map[], int16_t pcm[] (amysamples.h:8-9).
 */
#ifndef SKRED_SYNTH_ABI_H
#define SKRED_SYNTH_ABI_H

#include <stdint.h>
#include <time.h>

#include "skred_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

#define SKRED_VOICE_MAX 64            /* skred.h:9   */
#define SKRED_WAVE_TABLE_MAX 1200     /* skred.h:81  (EXT_SAMPLE_999 + 1) */
#define SKRED_MAIN_SAMPLE_RATE 44100  /* skred.h:6   */
#define SKRED_AMY_FACTOR 0.025f       /* skred.h:11  */

/* ---- wave slots (synth.def:1-10) ---- */
extern float *wave_table_data[SKRED_WAVE_TABLE_MAX];
extern int   wave_size[SKRED_WAVE_TABLE_MAX];
extern float wave_rate[SKRED_WAVE_TABLE_MAX];
extern int   wave_one_shot[SKRED_WAVE_TABLE_MAX];
extern int   wave_loop_enabled[SKRED_WAVE_TABLE_MAX];
extern int   wave_loop_start[SKRED_WAVE_TABLE_MAX];
extern int   wave_loop_end[SKRED_WAVE_TABLE_MAX];
extern float wave_midi_note[SKRED_WAVE_TABLE_MAX];
extern float wave_offset_hz[SKRED_WAVE_TABLE_MAX];
extern int   wave_is_miniwav[SKRED_WAVE_TABLE_MAX];

/* ---- voices: oscillator (synth.def:12-29) ---- */
extern float  voice_phase[SKRED_VOICE_MAX];
extern float  voice_phase_inc[SKRED_VOICE_MAX];
extern float *voice_table[SKRED_VOICE_MAX];
extern int    voice_table_size[SKRED_VOICE_MAX];
extern int    voice_one_shot[SKRED_VOICE_MAX];
extern int    voice_finished[SKRED_VOICE_MAX];
extern int    voice_loop_enabled[SKRED_VOICE_MAX];
extern float  voice_table_rate[SKRED_VOICE_MAX];
extern int    voice_loop_start[SKRED_VOICE_MAX];
extern int    voice_loop_end[SKRED_VOICE_MAX];
extern float  voice_midi_note[SKRED_VOICE_MAX];
extern float  voice_midi_transpose[SKRED_VOICE_MAX];
extern float  voice_link_midi_a[SKRED_VOICE_MAX];
extern float  voice_link_midi_b[SKRED_VOICE_MAX];
extern float  voice_link_velo_a[SKRED_VOICE_MAX];
extern float  voice_link_velo_b[SKRED_VOICE_MAX];
extern float  voice_link_trig[SKRED_VOICE_MAX];
extern float  voice_offset_hz[SKRED_VOICE_MAX];
/* ---- voices: level, pan, sample chain (synth.def:31-42) ---- */
extern float voice_freq[SKRED_VOICE_MAX];
extern float voice_note[SKRED_VOICE_MAX];
extern float voice_sample[SKRED_VOICE_MAX];
extern float voice_sample_hold[SKRED_VOICE_MAX];
extern int   voice_sample_hold_count[SKRED_VOICE_MAX];
extern int   voice_sample_hold_max[SKRED_VOICE_MAX];
extern float voice_amp[SKRED_VOICE_MAX];
extern float voice_user_amp[SKRED_VOICE_MAX];
extern float voice_pan_left[SKRED_VOICE_MAX];
extern float voice_pan_right[SKRED_VOICE_MAX];
extern float voice_pan[SKRED_VOICE_MAX];
extern int   voice_use_amp_envelope[SKRED_VOICE_MAX];
/* ---- voices: modulation routing (synth.def:44-59) ---- */
extern int   voice_freq_mod_osc[SKRED_VOICE_MAX];
extern float voice_freq_mod_depth[SKRED_VOICE_MAX];
extern float voice_freq_scale[SKRED_VOICE_MAX];
extern int   voice_pan_mod_osc[SKRED_VOICE_MAX];
extern int   voice_amp_mod_osc[SKRED_VOICE_MAX];
extern int   voice_cz_mod_osc[SKRED_VOICE_MAX];
extern float voice_pan_mod_depth[SKRED_VOICE_MAX];
extern float voice_amp_mod_depth[SKRED_VOICE_MAX];
extern float voice_cz_mod_depth[SKRED_VOICE_MAX];
extern int   voice_disconnect[SKRED_VOICE_MAX];
extern int   voice_quantize[SKRED_VOICE_MAX];
extern int   voice_direction[SKRED_VOICE_MAX];
extern int   voice_phase_reset[SKRED_VOICE_MAX];
extern int   voice_record[SKRED_VOICE_MAX];
/* ---- voices: table slot, CZ, smoother, glissando (synth.def:61-72) ---- */
extern int   voice_wave_table_index[SKRED_VOICE_MAX];
extern int   voice_cz_mode[SKRED_VOICE_MAX];
extern float voice_cz_distortion[SKRED_VOICE_MAX];
extern int   voice_smoother_enable[SKRED_VOICE_MAX];
extern float voice_smoother_gain[SKRED_VOICE_MAX];
extern float voice_smoother_smoothing[SKRED_VOICE_MAX];
extern int   voice_glissando_enable[SKRED_VOICE_MAX];
extern float voice_glissando_speed[SKRED_VOICE_MAX];
extern float voice_glissando_target[SKRED_VOICE_MAX];
/* ---- voices: filter, envelope, loop window (synth.def:74-84) ---- */
extern float voice_filter_freq[SKRED_VOICE_MAX];
extern float voice_filter_res[SKRED_VOICE_MAX];
extern int   voice_filter_mode[SKRED_VOICE_MAX];
extern skred_mmf_t      voice_filter[SKRED_VOICE_MAX];
extern skred_envelope_t voice_amp_envelope[SKRED_VOICE_MAX];
extern int   voice_loop_valid[SKRED_VOICE_MAX];
extern int   voice_loop_length[SKRED_VOICE_MAX];
extern float voice_loop_start_f[SKRED_VOICE_MAX];
extern float voice_loop_end_f[SKRED_VOICE_MAX];
/* ---- voices: latency marks (synth.def:88-90) ---- */
extern int             voice_mark_go[SKRED_VOICE_MAX];
extern struct timespec voice_mark_a[SKRED_VOICE_MAX];
extern struct timespec voice_mark_b[SKRED_VOICE_MAX];

/* ---- scalars (synth.h:12-22) ---- */
extern int requested_synth_frames_per_callback;
extern int synth_frames_per_callback;
extern volatile uint64_t synth_sample_count;
extern float volume_user, volume_final, volume_smoother_gain, volume_smoother_smoothing;
extern float volume_threshold, volume_smoother_higher_smoothing;

/* ---- the render entry (synth.h:8) ---- */
void synth(float *buffer, float *input, int num_frames, int num_channels, void *user);

/* ---- lifecycle / tables (synth.h:9-10,79-82) ---- */
void synth_init(void);
void synth_free(void);
void wave_table_init(void);
void wave_free(void);
void voice_init(void);
void voice_reset(int voice);

/* ---- control path, same signatures and return codes as synth.h:27-78 ---- */
float osc_get_phase_inc(int v, float f);
void  osc_set_freq(int v, float f);
void  osc_set_wave_table_index(int voice, int wave);
void  osc_trigger(int voice);
void  mmf_init(int n, float f, float resonance);
void  mmf_set_params(int n, float f, float resonance);
int   mmf_set_freq(int n, float f);
int   mmf_set_res(int n, float res);
void  envelope_init(int v, float attack_time, float decay_time, float sustain_level, float release_time);
void  amp_envelope_trigger(int v, float f);
void  amp_envelope_release(int v);
int   volume_set(float v);
int   envelope_is_flat(int v);
int   cz_set(int v, int n, float f);
int   cmod_set(int voice, int o, float f);
int   amp_set(int v, float f);
int   pan_set(int voice, float f);
int   wave_quant(int voice, int n);
int   freq_set(int v, float f);
int   voice_set(int n, int *old_voice);
int   voice_copy(int v, int n);
int   wave_set(int voice, int wave);
int   wave_mute(int voice, int state);
int   wave_dir(int voice, int state);
int   freq_midi(int voice, float f);
int   amp_mod_set(int voice, int o, float f);
int   envelope_velocity(int voice, float f);
int   envelope_set(int voice, float a, float d, float s, float r);
int   wave_reset(int voice, int n);
int   freq_mod_set(int voice, int o, float f);
int   pan_mod_set(int voice, int o, float f);
int   voice_trigger(int voice);
int   wave_default(int voice);
int   wave_loop(int voice, int state);
float midi2hz(float f);
char *voice_format(int v, char *out, int verbose);
void  voice_show(int v, char c, int verbose);
int   voice_show_all(int voice, int verbose);
char *synth_stats(void);
void  synth_voice_bench(int voice);

/* ---- per-sample entry points, same signatures as synth.h:24-26,29,30,33,38,43; host side, one
 *      voice of the global arrays, one sample (skred_synth_persample.c).  Not used by synth(). ---- */
void     audio_rng_init(uint64_t *rng, uint64_t seed);     /* synth.c:105-107  */
uint64_t audio_rng_next(uint64_t *rng);                    /* synth.c:110-114  */
float    audio_rng_float(uint64_t *rng);                   /* synth.c:117-123  */
float    cz_phasor(int n, float p, float d, int table_size);   /* synth.c:149-215 */
float    osc_next(int voice, float phase_inc);             /* synth.c:217-275  */
float    quantize_bits_int(float v, int bits);             /* synth.c:341-345  */
float    mmf_process(int n, float input);                  /* synth.c:349-364  */
float    amp_envelope_step(int v);                         /* synth.c:398-431  */

/* ---- `.sk` patch subset -> voice state (skred_patch.c; SURVEY 8f "next" #1) ---- */
typedef struct { int voice; int unsupported; int errors; } skred_patch_t;
void skred_patch_init(skred_patch_t *p);
int  skred_patch_line(skred_patch_t *p, const char *line);  /* returns unsupported tokens met in this line */
int  skred_patch_load(const char *path, skred_patch_t *p);  /* -1: cannot open; else total unsupported tokens */

/* ---- `:wN,slot,ch` -- N.wav -> EXT slot (wire.c:406-441 wave_load, with include/skred_wav.h's reader) ----
 * 0 on success, SKRED_ERR_INVALID_EXT_SAMPLE when the slot is outside [200, 1199) or the file cannot be
 * decoded (the reference returns ERR_INVALID_EXT_SAMPLE, wire.h:152, in both cases). */
#define SKRED_ERR_INVALID_EXT_SAMPLE 17   /* position of ERR_INVALID_EXT_SAMPLE in wire.h:134-155 */
int skred_wave_load(int which, int where, int ch);

/* ---- additions of this build (no counterpart in the reference) ---- */
int         skred_synth_last_rc(void);          /* SKRED_E_* of the most recent synth() call */
const char *skred_synth_last_error(void);
void        skred_synth_set_device(int device); /* GPU used by synth(); default 0 */
void        skred_synth_shutdown(void);         /* release the GPU bank */

#ifdef __cplusplus
}
#endif
#endif
