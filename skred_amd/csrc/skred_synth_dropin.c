/*
 * skred_synth_dropin.c -- DROP-IN MODE (include/skred_synth_abi.h): what the rest of skred links
 * instead of synth.o.  Two halves:
 *
 *   1. the control path -- the global voice / wave-slot arrays and the setters that wire.c, seq.c
 *      and skred.c call (reference synth.c:96-136,277-339,367-395,632-1169,1199-1307).  Scalar C on
 *      the host, exactly as in the reference: it runs once per user command, not per sample.
 *      tests/test_dropin.py (oracle/Makefile: dropin_check) links the reference's own wire.o/seq.o/skred.o against this file and
 *      checks that every patch line leaves bit-identical state to the reference's synth.o.
 *
 *   2. synth() -- the per-sample render loop (synth.c:502-630) is NOT here: the call snapshots the
 *      arrays into a 64-voice GPU bank (include/skred_amd.h), renders there, and copies the frames,
 *      stems and recurrences back.  No CPU rendering exists in this file; if the GPU path fails the
 *      callback outputs silence and records the error (an audio callback must not abort).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "skred_synth_abi.h"

#define NV SKRED_VOICE_MAX
#define NW SKRED_WAVE_TABLE_MAX
#define RATE SKRED_MAIN_SAMPLE_RATE
#define SMOOTH_K 0.02f               /* synth.c:87 */
#define ERR_ARG 100                  /* the reference's catch-all codes, synth.c:834,844,894,661 */
#define ERR_FREQ 101                 /* synth.c:860 */

enum { SLOT_SINE = 0, SLOT_SQR, SLOT_SAW_DOWN, SLOT_SAW_UP, SLOT_TRI, SLOT_NOISE, SLOT_NOISE_ALT,
       SLOT_KORG_FIRST = 32, SLOT_KORG_LAST = 62,          /* WAVE_TABLE_KRG1 .. KRG32-1, synth.c:1253 */
       SLOT_AMY_FIRST = 100, SLOT_AMY_LAST = 199 };        /* skred.h:24-76 */

/* ------------------------------------------------------------------ imported (weak) symbols */
typedef struct {            /* == pcm_map_t, notamy/pcm.h:4-12 */
  int32_t offset; uint32_t length, loopstart, loopend; uint8_t midinote; int32_t rate; int16_t *external;
} amy_region_t;
extern int debug __attribute__((weak));
extern float *mw_free(float *) __attribute__((weak));
extern amy_region_t pcm_map[] __attribute__((weak));
extern int16_t pcm[] __attribute__((weak));
#define AMY_REGIONS 67        /* amysamples.h:5 */
#define AMY_RATE 22050        /* amysamples.h:6 */

/* ------------------------------------------------------------------ data symbols */
float *wave_table_data[NW]; int wave_size[NW]; float wave_rate[NW]; int wave_one_shot[NW];
int wave_loop_enabled[NW]; int wave_loop_start[NW]; int wave_loop_end[NW];
float wave_midi_note[NW]; float wave_offset_hz[NW]; int wave_is_miniwav[NW];

float voice_phase[NV], voice_phase_inc[NV]; float *voice_table[NV]; int voice_table_size[NV];
int voice_one_shot[NV], voice_finished[NV], voice_loop_enabled[NV]; float voice_table_rate[NV];
int voice_loop_start[NV], voice_loop_end[NV];
float voice_midi_note[NV], voice_midi_transpose[NV], voice_link_midi_a[NV], voice_link_midi_b[NV];
float voice_link_velo_a[NV], voice_link_velo_b[NV], voice_link_trig[NV], voice_offset_hz[NV];
float voice_freq[NV], voice_note[NV], voice_sample[NV], voice_sample_hold[NV];
int voice_sample_hold_count[NV], voice_sample_hold_max[NV];
float voice_amp[NV], voice_user_amp[NV], voice_pan_left[NV], voice_pan_right[NV], voice_pan[NV];
int voice_use_amp_envelope[NV];
int voice_freq_mod_osc[NV]; float voice_freq_mod_depth[NV], voice_freq_scale[NV];
int voice_pan_mod_osc[NV], voice_amp_mod_osc[NV], voice_cz_mod_osc[NV];
float voice_pan_mod_depth[NV], voice_amp_mod_depth[NV], voice_cz_mod_depth[NV];
int voice_disconnect[NV], voice_quantize[NV], voice_direction[NV], voice_phase_reset[NV], voice_record[NV];
int voice_wave_table_index[NV], voice_cz_mode[NV]; float voice_cz_distortion[NV];
int voice_smoother_enable[NV]; float voice_smoother_gain[NV], voice_smoother_smoothing[NV];
int voice_glissando_enable[NV]; float voice_glissando_speed[NV], voice_glissando_target[NV];
float voice_filter_freq[NV], voice_filter_res[NV]; int voice_filter_mode[NV];
skred_mmf_t voice_filter[NV]; skred_envelope_t voice_amp_envelope[NV];
int voice_loop_valid[NV], voice_loop_length[NV]; float voice_loop_start_f[NV], voice_loop_end_f[NV];
int voice_mark_go[NV]; struct timespec voice_mark_a[NV], voice_mark_b[NV];

int requested_synth_frames_per_callback = 512;     /* SYNTH_FRAMES_PER_CALLBACK, skred.h:12 */
int synth_frames_per_callback = 0;
volatile uint64_t synth_sample_count = 0;
float volume_user = 1.0f;
float volume_final = SKRED_AMY_FACTOR;
float volume_smoother_gain = 0.0f;
float volume_smoother_smoothing = 0.002f;
float volume_threshold = 0.05f;
float volume_smoother_higher_smoothing = 0.3f;

static inline int bad_voice(int v) { return v < 0 || v >= NV; }
static inline int flip_or_set(int state, int current) { return state < 0 ? !current : state; }

/* ================================================================== control path */

/* ---- master volume (synth.c:96-100) ---- */
int volume_set(float v) {
  volume_user = v;
  volume_final = v * SKRED_AMY_FACTOR;
  return 0;
}

/* ---- oscillator set-up ---- */

/* table samples advanced per output frame (synth.c:125-132); the association of the products and
 * quotients is the reference's, kept because phase_inc feeds the render bit for bit */
float osc_get_phase_inc(int v, float f) {
  const float pitch = voice_one_shot[v] ? f / voice_offset_hz[v] : f;
  const float tr = voice_table_rate[v];
  return (pitch * (float)voice_table_size[v]) / tr * (tr / RATE);
}

void osc_set_freq(int v, float f) { voice_phase_inc[v] = osc_get_phase_inc(v, f); }

/* what a voice inherits from the wave slot it is pointed at (synth.c:277-314) */
static void adopt_slot(int voice, int wave) {
  voice_wave_table_index[voice] = wave;
  voice_table[voice]        = wave_table_data[wave];
  voice_table_size[voice]   = wave_size[wave];
  voice_table_rate[voice]   = wave_rate[wave];
  voice_one_shot[voice]     = wave_one_shot[wave];
  voice_finished[voice]     = wave_one_shot[wave] ? 1 : 0;     /* a one-shot waits for its trigger */
  voice_loop_enabled[voice] = wave_loop_enabled[wave];
  voice_loop_start[voice]   = wave_loop_start[wave];
  voice_loop_end[voice]     = wave_loop_end[wave];
  voice_midi_note[voice]    = wave_midi_note[wave];
  voice_offset_hz[voice]    = wave_offset_hz[wave];
}

/* the loop window in the form the render loop reads (floats + a validity bit) */
static void derive_loop_window(int voice) {
  const int lo = voice_loop_start[voice], hi = voice_loop_end[voice];
  const int ok = hi > lo;
  voice_loop_start_f[voice] = (float)lo;
  voice_loop_end_f[voice] = (float)hi;
  voice_loop_valid[voice] = ok;
  voice_loop_length[voice] = (int)(float)(ok ? hi - lo : voice_table_size[voice]);
}

void osc_set_wave_table_index(int voice, int wave) {
  const int usable = wave_table_data[wave] && wave_size[wave] && wave_rate[wave] > 0.0;
  if (!usable) return;                                        /* empty slot: the voice keeps its table */
  const int retune = voice_table_rate[voice] != wave_rate[wave] || voice_table_size[voice] != wave_size[wave];
  adopt_slot(voice, wave);
  derive_loop_window(voice);
  if (retune) osc_set_freq(voice, voice_freq[voice]);         /* same pitch on the new geometry */
}

/* (re)start playback at the end that matches direction and looping (synth.c:316-339) */
void osc_trigger(int voice) {
  const int reversed = voice_direction[voice];
  const int whole_table = voice_one_shot[voice] || !voice_loop_enabled[voice];
  float at;
  if (whole_table) at = reversed ? (float)(voice_table_size[voice] - 1) : 0.0f;
  else             at = reversed ? (float)voice_loop_end[voice] - 1e-6f : (float)voice_loop_start[voice];
  voice_phase[voice] = at;
  voice_finished[voice] = 0;
}

/* ---- biquad coefficients: RBJ cookbook, cached on (freq, resonance, mode) (synth.c:929-1008) ---- */

typedef struct { float b0, b1, b2; } rbj_numerator_t;

static rbj_numerator_t rbj_numerator(int mode, float cs, float alpha) {
  switch (mode) {
    case 2:  return (rbj_numerator_t){ (1.0f + cs) / 2.0f, -(1.0f + cs), (1.0f + cs) / 2.0f };   /* high-pass */
    case 3:  return (rbj_numerator_t){ alpha, 0.0f, -alpha };                                    /* band-pass */
    case 4:  return (rbj_numerator_t){ 1.0f, -2.0f * cs, 1.0f };                                 /* notch */
    case 5:  return (rbj_numerator_t){ 1.0f - alpha, -2.0f * cs, 1.0f + alpha };                 /* all-pass */
    default: return (rbj_numerator_t){ (1.0f - cs) / 2.0f, 1.0f - cs, (1.0f - cs) / 2.0f };      /* low-pass; also any unknown mode */
  }
}

void mmf_set_params(int n, float f, float resonance) {
  skred_mmf_t *q = &voice_filter[n];
  const int mode = voice_filter_mode[n];
  if (q->last_freq == f && q->last_resonance == resonance && q->last_mode == mode) return;
  q->last_freq = f; q->last_resonance = resonance; q->last_mode = mode;
  if (mode == 0) return;                                      /* bypassed: keys remembered, nothing computed */
  const float w = 2.0f * (float)M_PI * f / (float)RATE;
  const float sn = sinf(w), cs = cosf(w);
  const float alpha = sn / (2.0f * resonance);
  const float a0 = 1.0f + alpha;
  const rbj_numerator_t num = rbj_numerator(mode, cs, alpha);
  q->b0 = num.b0 / a0; q->b1 = num.b1 / a0; q->b2 = num.b2 / a0;
  q->a1 = (-2.0f * cs) / a0;
  q->a2 = (1.0f - alpha) / a0;
  voice_filter_freq[n] = f;
  voice_filter_res[n] = resonance;
}

void mmf_init(int n, float f, float resonance) {              /* synth.c:1015-1030 */
  skred_mmf_t *q = &voice_filter[n];
  q->x1 = q->x2 = q->y1 = q->y2 = 0.0f;
  q->last_freq = q->last_resonance = -1.0f;                   /* impossible keys: the next set recomputes */
  q->last_mode = -1;
  voice_filter_freq[n] = f;
  voice_filter_res[n] = resonance;
  mmf_set_params(n, f, resonance);
}

int mmf_set_freq(int n, float f) { mmf_set_params(n, f, voice_filter_res[n]); return 0; }
int mmf_set_res(int n, float res) { if (res > 0) mmf_set_params(n, voice_filter_freq[n], res); return 0; }

/* ---- envelope (synth.c:367-395,632-638,1146-1159) ---- */

void envelope_init(int v, float a, float d, float s, float r) {
  skred_envelope_t *e = &voice_amp_envelope[v];
  e->a = a; e->d = d; e->s = s; e->r = r;                     /* seconds, as typed */
  e->attack_time = a * RATE; e->decay_time = d * RATE; e->release_time = r * RATE;   /* frames */
  e->sustain_level = fmaxf(0, fminf(1.0f, s));
  e->sample_start = e->sample_release = 0;
  e->is_active = 0;
}

void amp_envelope_trigger(int v, float velocity) {
  skred_envelope_t *e = &voice_amp_envelope[v];
  e->sample_start = synth_sample_count;
  e->sample_release = 0;
  e->velocity = velocity;
  e->is_active = 1;
}

void amp_envelope_release(int v) {
  skred_envelope_t *e = &voice_amp_envelope[v];
  if (e->is_active) e->sample_release = synth_sample_count;
}

int envelope_is_flat(int v) {
  const skred_envelope_t *e = &voice_amp_envelope[v];
  return e->a == 0.0f && e->d == 0.0f && e->s == 1.0f && e->r == 0.0f;
}

int envelope_set(int voice, float a, float d, float s, float r) { envelope_init(voice, a, d, s, r); return 0; }

/* `l<velocity>`: 0 is note-off, anything else (re)starts the note */
int envelope_velocity(int voice, float f) {
  if (bad_voice(voice)) return ERR_ARG;
  if (f == 0) { amp_envelope_release(voice); return 0; }
  voice_use_amp_envelope[voice] = 1;
  if (voice_one_shot[voice]) osc_trigger(voice);
  amp_envelope_trigger(voice, f);
  return 0;
}

/* ---- one-line setters (synth.c:829-911) ---- */

int cz_set(int v, int n, float f) { voice_cz_mode[v] = n; voice_cz_distortion[v] = f; return 0; }
int cmod_set(int voice, int o, float f) { voice_cz_mod_osc[voice] = o; voice_cz_mod_depth[voice] = f; return 0; }
int wave_quant(int voice, int n) { voice_quantize[voice] = n; return 0; }
int wave_mute(int voice, int state) { voice_disconnect[voice] = flip_or_set(state, voice_disconnect[voice]); return 0; }
int wave_dir(int voice, int state) { voice_direction[voice] = flip_or_set(state, voice_direction[voice]); return 0; }
int wave_loop(int voice, int state) { voice_loop_enabled[voice] = flip_or_set(state, voice_loop_enabled[voice]); return 0; }
int voice_trigger(int voice) { osc_trigger(voice); return 0; }

int amp_set(int voice, float f) {
  if (!(f >= 0)) return ERR_ARG;
  voice_use_amp_envelope[voice] = 0;                          /* a plain level switches the envelope off */
  voice_amp[voice] = voice_user_amp[voice] = f;
  return 0;
}

int pan_set(int voice, float f) {                             /* linear pan law */
  if (!(f >= -1.0f && f <= 1.0f)) return ERR_ARG;
  voice_pan[voice] = f;
  voice_pan_left[voice] = (1.0f - f) / 2.0f;
  voice_pan_right[voice] = (1.0f + f) / 2.0f;
  return 0;
}

int freq_set(int voice, float f) {
  if (!(f >= 0 && f < (double)RATE)) return ERR_FREQ;
  voice_freq[voice] = f;
  osc_set_freq(voice, f);
  return 0;
}

int wave_set(int voice, int wave) {
  if (wave < 0 || wave >= NW) return ERR_ARG;
  osc_set_wave_table_index(voice, wave);
  return 0;
}

/* the three modulator routings share one argument check */
static int route(int voice, int source, int *osc, float *depth, float amount) {
  if (bad_voice(voice) || bad_voice(source)) return ERR_ARG;
  osc[voice] = source;
  depth[voice] = amount;
  return 0;
}

int pan_mod_set(int voice, int o, float f) { return route(voice, o, voice_pan_mod_osc, voice_pan_mod_depth, f); }
int amp_mod_set(int voice, int o, float f) { return route(voice, o, voice_amp_mod_osc, voice_amp_mod_depth, f); }

int freq_mod_set(int voice, int o, float f) {
  const int rc = route(voice, o, voice_freq_mod_osc, voice_freq_mod_depth, f);
  if (!rc) voice_freq_scale[voice] = (float)voice_table_size[voice] / (float)voice_table_size[o];
  return rc;
}

/* ---- notes (synth.c:1056-1088) ---- */

float midi2hz(float f) { return 440.0f * powf(2.0f, (f - 69.0f) / 12.0f); }

int voice_set(int n, int *old_voice) {
  if (bad_voice(n)) return ERR_ARG;
  if (old_voice) *old_voice = n;
  return 0;
}

int wave_default(int voice) {                                 /* play the slot at its own root note */
  const float note = (float)voice_midi_note[voice];
  const float hz = midi2hz(note);
  voice_freq[voice] = hz;
  voice_note[voice] = note;
  osc_set_freq(voice, hz);
  return 0;
}

int freq_midi(int voice, float f) {
  if (!(f >= 0.0 && f <= 127.0)) return ERR_ARG;
  if (voice_midi_transpose[voice]) f += voice_midi_transpose[voice];
  return freq_set(voice, midi2hz(f));
}

/* ---- whole-voice operations ---- */

/* `>n`: voice v's patch onto voice n through the setters, in the reference's order of effects
 * (synth.c:1033-1054): each setter's side effects (retune on a new table, envelope switch-off on a
 * plain level, filter memory cleared) must land in that order for the states to agree */
int voice_copy(int v, int n) {
  const skred_envelope_t *e = &voice_amp_envelope[v];
  wave_set(n, voice_wave_table_index[v]);
  amp_set(n, voice_user_amp[v]);
  freq_set(n, voice_freq[v]);
  pan_set(n, voice_pan[v]);
  amp_mod_set(n, voice_amp_mod_osc[v], voice_amp_mod_depth[v]);
  freq_mod_set(n, voice_freq_mod_osc[v], voice_freq_mod_depth[v]);
  pan_mod_set(n, voice_pan_mod_osc[v], voice_pan_mod_depth[v]);
  wave_loop(n, voice_loop_enabled[v]);
  wave_dir(n, voice_direction[v]);
  wave_quant(n, voice_quantize[v]);
  voice_sample_hold_max[n] = voice_sample_hold_max[v];
  voice_sample_hold_count[n] = voice_sample_hold_count[v];
  voice_sample_hold[n] = voice_sample_hold[v];
  envelope_set(n, e->a, e->d, e->s, e->r);
  cz_set(n, voice_cz_mode[v], voice_cz_distortion[v]);
  cmod_set(n, voice_cz_mod_osc[v], voice_cz_mod_depth[v]);
  voice_filter_mode[n] = voice_filter_mode[v];
  mmf_init(n, voice_filter_freq[v], voice_filter_res[v]);
  return 0;
}

/* Power-on values of the plain per-voice fields (synth.c:1090-1132) as a list; voice_reset() stores
 * them and then runs the three initialisers whose results depend on them. */
typedef struct { void *array; char type; float value; } field_default_t;     /* type: 'i' int, 'f' float */
static const field_default_t POWER_ON[] = {
  { voice_wave_table_index, 'i', 0 },   { voice_table_rate, 'f', 0 },        { voice_table_size, 'i', 0 },
  { voice_sample, 'f', 0 },             { voice_amp, 'f', 0 },               { voice_user_amp, 'f', 0 },
  { voice_pan, 'f', 0 },                { voice_pan_left, 'f', 0.5f },       { voice_pan_right, 'f', 0.5f },
  { voice_use_amp_envelope, 'i', 0 },
  { voice_amp_mod_osc, 'i', -1 },       { voice_freq_mod_osc, 'i', -1 },     { voice_pan_mod_osc, 'i', -1 },
  { voice_freq_mod_depth, 'f', 0 },     { voice_freq_scale, 'f', 1.0f },
  { voice_disconnect, 'i', 0 },         { voice_quantize, 'i', 0 },          { voice_direction, 'i', 0 },
  { voice_freq, 'f', 440.0f },          { voice_midi_note, 'f', 69.0f },     { voice_midi_transpose, 'f', 0 },
  { voice_link_midi_a, 'f', -1 },       { voice_link_midi_b, 'f', -1 },
  { voice_link_velo_a, 'f', -1 },       { voice_link_velo_b, 'f', -1 },      { voice_link_trig, 'f', -1 },
  { voice_filter_mode, 'i', 0 },
  { voice_smoother_enable, 'i', 1 },    { voice_smoother_gain, 'f', 0 },     { voice_smoother_smoothing, 'f', SMOOTH_K },
  { voice_glissando_enable, 'i', 0 },   { voice_glissando_speed, 'f', 0 },   { voice_glissando_target, 'f', 440.0f },
  { voice_record, 'i', 0 },
};

void voice_reset(int i) {
  for (size_t k = 0; k < sizeof(POWER_ON) / sizeof(POWER_ON[0]); k++) {
    const field_default_t *d = &POWER_ON[k];
    if (d->type == 'i') ((int *)d->array)[i] = (int)d->value; else ((float *)d->array)[i] = d->value;
  }
  envelope_init(i, 0.0f, 0.0f, 1.0f, 0.0f);                   /* flat: level 1, no ramps */
  osc_set_wave_table_index(i, SLOT_SINE);                     /* geometry was zeroed above, so this retunes to 440 Hz */
  mmf_init(i, 8000.0f, 0.707f);
}

void voice_init(void) { for (int i = 0; i < NV; i++) voice_reset(i); }

/* `S<n>`: one voice, or every voice when n names none (synth.c:1140-1144) */
int wave_reset(int voice, int n) {
  (void)voice;
  if (bad_voice(n)) voice_init(); else voice_reset(n);
  return 0;
}

/* ------------------------------------------------------------------ tables */

void synth_init(void) { printf("# synth_init :: GPU render path (libskred_amd), static arrays\n"); }
void synth_free(void) { skred_synth_shutdown(); printf("# synth_free\n"); }

static void scale_to_unit_peak(float *d, int n) {             /* synth.c:1175-1197 */
  float peak = 0.0f;
  for (int i = 0; i < n; i++) if (fabsf(d[i]) > peak) peak = fabsf(d[i]);
  if (peak == 0.0) return;
  const float k = 1.0f / peak;
  for (int i = 0; i < n; i++) d[i] *= k;
}

static void fill_slot(int slot, float *data, int n, float rate, int one_shot, int loop_lo, int loop_hi) {
  wave_table_data[slot] = data; wave_size[slot] = n; wave_rate[slot] = rate;
  wave_one_shot[slot] = one_shot; wave_loop_start[slot] = loop_lo; wave_loop_end[slot] = loop_hi;
}

/* slots 0-6: 4096-entry single cycles generated from one running float phase (synth.c:1208-1249);
 * the two noise slots draw from one LCG stream seeded 1 */
static void make_builtin_cycles(void) {
  enum { N = 4096 };
  uint64_t rng;
  audio_rng_init(&rng, 1);
  const float step = 1.0f / (float)N;
  for (int w = SLOT_SINE; w <= SLOT_NOISE_ALT; w++) {
    float *t = (float *)malloc(sizeof(float) * N);
    int k = 0;
    for (float ph = 0.0f; ph < 1.0f; ph += step, k++) {
      float v;
      switch (w) {
        case SLOT_SINE:     v = sinf(2.0f * (float)M_PI * ph); break;
        case SLOT_SQR:      v = (ph < 0.5) ? 1.0f : -1.0f; break;
        case SLOT_SAW_DOWN: v = 2.0f * ph - 1.0f; break;
        case SLOT_SAW_UP:   v = 1.0f - 2.0f * ph; break;
        case SLOT_TRI:      v = (ph < 0.5f) ? (4.0f * ph - 1.0f) : (3.0f - 4.0f * ph); break;
        default:            v = audio_rng_float(&rng); break;
      }
      t[k] = v;
    }
    fill_slot(w, t, N, RATE, 0, 0, N - 1);
  }
}

/* Where this library's data files live: $SKRED_AMD_DATA, else data/ beside the shared object. */
static const char *data_path(const char *name, char *buf, size_t cap) {
  const char *dir = getenv("SKRED_AMD_DATA");
  if (dir && *dir) { snprintf(buf, cap, "%s/%s", dir, name); return buf; }
  Dl_info info;
  if (dladdr((void *)&data_path, &info) && info.dli_fname) {
    snprintf(buf, cap, "%s", info.dli_fname);
    char *slash = strrchr(buf, '/');
    if (slash) { snprintf(slash + 1, cap - (size_t)(slash + 1 - buf), "data/%s", name); return buf; }
  }
  snprintf(buf, cap, "data/%s", name);
  return buf;
}

/* slots 32-62: the Korg DW-8000 single cycles (synth.c:1251-1268: int16/32767, MAIN_SAMPLE_RATE, not
 * one-shot, loop 0..size-1) from the package's data blob (skred_amd/data/korg_waves.bin: "SKKORG1\0",
 * u32 count, u32 size[count], int16 samples; written by skred_amd/data/extract_korg_waves.py).
 * Returns the number of slots filled; a missing or damaged blob is reported, never papered over. */
static int load_korg_cycles(void) {
  char path[1024];
  FILE *f = fopen(data_path("korg_waves.bin", path, sizeof path), "rb");
  if (!f) { fprintf(stderr, "# skred_synth: %s not found: wave slots %d-%d stay empty\n", path, SLOT_KORG_FIRST, SLOT_KORG_LAST); return 0; }
  char magic[8];
  uint32_t count = 0, sizes[SLOT_KORG_LAST - SLOT_KORG_FIRST + 1];
  int filled = 0;
  if (fread(magic, 1, 8, f) == 8 && !memcmp(magic, "SKKORG1", 8) && fread(&count, 4, 1, f) == 1 &&
      count <= SLOT_KORG_LAST - SLOT_KORG_FIRST + 1 && fread(sizes, 4, count, f) == count) {
    for (uint32_t k = 0; k < count; k++) {
      const int n = (int)sizes[k];
      int16_t *raw = (int16_t *)malloc(sizeof(int16_t) * (size_t)n);
      float *t = (float *)malloc(sizeof(float) * (size_t)n);
      if (fread(raw, sizeof(int16_t), (size_t)n, f) != (size_t)n) { free(raw); free(t); break; }
      for (int j = 0; j < n; j++) t[j] = (float)raw[j] / (float)32767;
      free(raw);
      fill_slot(SLOT_KORG_FIRST + (int)k, t, n, RATE, 0, 0, n - 1);
      filled++;
    }
  }
  fclose(f);
  if (filled != SLOT_KORG_LAST - SLOT_KORG_FIRST + 1)
    fprintf(stderr, "# skred_synth: %s is damaged: only %d of %d Korg waves loaded\n", path, filled, SLOT_KORG_LAST - SLOT_KORG_FIRST + 1);
  return filled;
}

/* slots 100..: the AMY PCM regions, when the host program provides pcm / pcm_map (synth.c:1270-1292) */
static void load_amy_regions(void) {
  if (!(pcm_map && pcm)) return;
  for (int i = 0; i < AMY_REGIONS && SLOT_AMY_FIRST + i < SLOT_AMY_LAST; i++) {
    const int slot = SLOT_AMY_FIRST + i;
    const amy_region_t *m = &pcm_map[i];
    float *t = (float *)malloc(sizeof(float) * (size_t)m->length);
    for (uint32_t k = 0; k < m->length; k++) t[k] = (float)pcm[m->offset + k] / 32767.0f;
    scale_to_unit_peak(t, (int)m->length);
    fill_slot(slot, t, (int)m->length, AMY_RATE, 1, (int)m->loopstart, (int)m->loopend);
    wave_loop_enabled[slot] = 0;
    wave_midi_note[slot] = (int)m->midinote;
    wave_offset_hz[slot] = midi2hz((float)m->midinote);
  }
}

void wave_table_init(void) {                                  /* synth.c:1199-1294 */
  for (int i = 0; i < NW; i++) { wave_table_data[i] = NULL; wave_size[i] = 0; wave_is_miniwav[i] = 0; }
  make_builtin_cycles();
  load_korg_cycles();
  load_amy_regions();
}

void wave_free(void) {                                        /* synth.c:1296-1307 */
  for (int i = 0; i < NW; i++) {
    if (!wave_table_data[i]) continue;
    if (wave_is_miniwav[i] && mw_free) mw_free(wave_table_data[i]); else free(wave_table_data[i]);
    wave_size[i] = 0;
  }
}

/* ------------------------------------------------------------------ text (wire `?`, `:s`) */

static long long ns_between(const struct timespec *a, const struct timespec *b) {
  return ((long long)b->tv_sec - a->tv_sec) * 1000000000LL + ((long long)b->tv_nsec - a->tv_nsec);
}

/* voice_format() writes a voice back as the wire text that would recreate it (synth.c:663-808).
 * The token list below is that format: each token is up to four values, each with the text in front of
 * it, printed when its condition holds (always in verbose mode, except where noted). */
typedef struct { const char *lead; char type; const void *base; size_t stride; } value_t;   /* 'i' int, 'f' float */
typedef enum {
  WHEN_NONZERO,         /* cond[0] != 0                                   */
  WHEN_ANY_ASSIGNED,    /* cond[0] >= 0 || cond[1] >= 0  (floats; -1 = unassigned) */
  WHEN_ROUTED,          /* cond[0] (int osc) >= 0 && cond[1] (float depth) > 0 */
  WHEN_CUSTOM_SMOOTHING,/* (verbose || enable) && smoothing != default: verbose alone does not force it */
  WHEN_SHAPED_ENVELOPE, /* !envelope_is_flat(v)                            */
} when_t;
typedef struct { when_t when; value_t cond[2]; value_t val[4]; } token_t;

#define INTS(a)   'i', (a), sizeof(int)
#define FLTS(a)   'f', (a), sizeof(float)
#define ENVF(m)   'f', &voice_amp_envelope[0].m, sizeof(skred_envelope_t)
static const token_t TOKENS[] = {
  { WHEN_NONZERO,      {{0, FLTS(voice_midi_transpose)}},                         {{" N", FLTS(voice_midi_transpose)}} },
  { WHEN_ANY_ASSIGNED, {{0, FLTS(voice_link_midi_a)}, {0, FLTS(voice_link_midi_b)}}, {{" G", FLTS(voice_link_midi_a)}, {",", FLTS(voice_link_midi_b)}} },
  { WHEN_ANY_ASSIGNED, {{0, FLTS(voice_link_velo_a)}, {0, FLTS(voice_link_velo_b)}}, {{" H", FLTS(voice_link_velo_a)}, {",", FLTS(voice_link_velo_b)}} },
  { WHEN_ANY_ASSIGNED, {{0, FLTS(voice_link_trig)}, {0, FLTS(voice_link_trig)}},  {{" L", FLTS(voice_link_trig)}} },
  { WHEN_NONZERO,      {{0, INTS(voice_direction)}},                              {{" b", INTS(voice_direction)}} },
  { WHEN_NONZERO,      {{0, INTS(voice_loop_enabled)}},                           {{" B", INTS(voice_loop_enabled)}} },
  { WHEN_NONZERO,      {{0, FLTS(voice_pan)}},                                    {{" p", FLTS(voice_pan)}} },
  { WHEN_NONZERO,      {{0, FLTS(voice_note)}},                                   {{" n", FLTS(voice_note)}} },
  { WHEN_NONZERO,      {{0, INTS(voice_filter_mode)}},                            {{" J", INTS(voice_filter_mode)}, {" K", FLTS(voice_filter_freq)}, {" Q", FLTS(voice_filter_res)}} },
  { WHEN_NONZERO,      {{0, INTS(voice_cz_mode)}},                                {{" c", INTS(voice_cz_mode)}, {",", FLTS(voice_cz_distortion)}} },
  { WHEN_NONZERO,      {{0, INTS(voice_quantize)}},                               {{" q", INTS(voice_quantize)}} },
  { WHEN_NONZERO,      {{0, INTS(voice_sample_hold_max)}},                        {{" h", INTS(voice_sample_hold_max)}} },
  { WHEN_ROUTED,       {{0, INTS(voice_amp_mod_osc)}, {0, FLTS(voice_amp_mod_depth)}},   {{" A", INTS(voice_amp_mod_osc)}, {",", FLTS(voice_amp_mod_depth)}} },
  { WHEN_ROUTED,       {{0, INTS(voice_cz_mod_osc)}, {0, FLTS(voice_cz_mod_depth)}},     {{" C", INTS(voice_cz_mod_osc)}, {",", FLTS(voice_cz_mod_depth)}} },
  { WHEN_ROUTED,       {{0, INTS(voice_freq_mod_osc)}, {0, FLTS(voice_freq_mod_depth)}}, {{" F", INTS(voice_freq_mod_osc)}, {",", FLTS(voice_freq_mod_depth)}} },
  { WHEN_ROUTED,       {{0, INTS(voice_pan_mod_osc)}, {0, FLTS(voice_pan_mod_depth)}},   {{" P", INTS(voice_pan_mod_osc)}, {",", FLTS(voice_pan_mod_depth)}} },
  { WHEN_NONZERO,      {{0, INTS(voice_disconnect)}},                             {{" m", INTS(voice_disconnect)}} },
  { WHEN_NONZERO,      {{0, INTS(voice_record)}},                                 {{" r", INTS(voice_record)}} },
  { WHEN_CUSTOM_SMOOTHING, {{0, INTS(voice_smoother_enable)}, {0, FLTS(voice_smoother_smoothing)}}, {{" s", FLTS(voice_smoother_smoothing)}} },
  { WHEN_NONZERO,      {{0, INTS(voice_glissando_enable)}},                       {{" g", FLTS(voice_glissando_speed)}} },
  { WHEN_SHAPED_ENVELOPE, {{0}},                                                  {{" t", ENVF(a)}, {",", ENVF(d)}, {",", ENVF(s)}, {",", ENVF(r)}} },
};
/* the read-only tail of the verbose form: runtime state, not part of a patch */
static const value_t VERBOSE_TAIL[] = {
  {" freq_scale:", FLTS(voice_freq_scale)}, {" finished:", INTS(voice_finished)}, {" one_shot:", INTS(voice_one_shot)},
  {" sample:", FLTS(voice_sample)}, {" smoother:", FLTS(voice_smoother_gain)},
  {" phase:", FLTS(voice_phase)}, {" phase_inc:", FLTS(voice_phase_inc)}, {" offset_hz:", FLTS(voice_offset_hz)},
};

static double value_at(const value_t *x, int v) {
  const char *p = (const char *)x->base + (size_t)v * x->stride;
  return x->type == 'i' ? (double)*(const int *)p : (double)*(const float *)p;
}

static char *put_value(char *p, const value_t *x, int v) {
  const char *at = (const char *)x->base + (size_t)v * x->stride;
  if (x->type == 'i') return p + sprintf(p, "%s%d", x->lead, *(const int *)at);
  return p + sprintf(p, "%s%g", x->lead, *(const float *)at);
}

static int token_due(const token_t *t, int v, int verbose) {
  switch (t->when) {
    case WHEN_NONZERO:          return verbose || value_at(&t->cond[0], v) != 0;
    case WHEN_ANY_ASSIGNED:     return verbose || value_at(&t->cond[0], v) >= 0 || value_at(&t->cond[1], v) >= 0;
    case WHEN_ROUTED:           return verbose || (value_at(&t->cond[0], v) >= 0 && value_at(&t->cond[1], v) > 0);
    case WHEN_CUSTOM_SMOOTHING: return (verbose || value_at(&t->cond[0], v) != 0) && *((const float *)t->cond[1].base + v) != SMOOTH_K;
    case WHEN_SHAPED_ENVELOPE:  return verbose || !envelope_is_flat(v);
  }
  return 0;
}

char *voice_format(int v, char *out, int verbose) {
  if (!out) return "(NULL)";
  if (bad_voice(v)) { out[0] = '\0'; return out; }
  char *p = out + sprintf(out, "v%d w%d f%g a%g", v, voice_wave_table_index[v], voice_freq[v], voice_user_amp[v]);
  for (size_t k = 0; k < sizeof(TOKENS) / sizeof(TOKENS[0]); k++) {
    const token_t *t = &TOKENS[k];
    if (!token_due(t, v, verbose)) continue;
    for (int j = 0; j < 4 && t->val[j].lead; j++) p = put_value(p, &t->val[j], v);
  }
  if (verbose) {
    p += sprintf(p, "\n#");
    for (size_t k = 0; k < sizeof(VERBOSE_TAIL) / sizeof(VERBOSE_TAIL[0]); k++) p = put_value(p, &VERBOSE_TAIL[k], v);
    p += sprintf(p, " latency:%gms", (double)ns_between(&voice_mark_a[v], &voice_mark_b[v]) / 1000000.0);
  }
  return out;
}

void voice_show(int v, char c, int verbose) {                 /* synth.c:811-817 */
  char s[1024];
  voice_format(v, s, verbose);
  if (s[0]) printf("; %s%s\n", s, c != ' ' ? " # *" : "");
}

int voice_show_all(int voice, int verbose) {                  /* every sounding voice; the console voice starred */
  for (int i = 0; i < NV; i++)
    if (voice_amp[i] != 0) voice_show(i, i == voice ? '*' : ' ', verbose);
  return 0;
}

void synth_voice_bench(int voice) {                           /* `:m`: stamp now, synth() stamps the next callback */
  voice_mark_b[voice].tv_sec = 0; voice_mark_b[voice].tv_nsec = 0;
  clock_gettime(CLOCK_MONOTONIC_COARSE, &voice_mark_a[voice]);
  voice_mark_go[voice] = 1;
}

#define PUT(...) (p += sprintf(p, __VA_ARGS__))

/* ================================================================== synth(): GPU render */

#define TIMING_RING 16                                      /* BENLEN, synth.c:444 */
static struct { struct timespec t0, t1; int frames; long long order; int filled; float gpu_ms; } g_ring[TIMING_RING];
static long long g_calls;

static skred_bank_t *g_bank;
static int g_device;
static int g_last_rc;
static char g_last_err[512];
static uint64_t g_noise_rng;
static float *g_stems;                 /* `user`, latched on the first call (synth.c:503-511) */
static int g_first = 1;

static int64_t g_offset[NV];
static float *g_pool;
static size_t g_pool_floats, g_pool_cap;
static struct { const float *ptr; int size; } g_seen[NV];

const char *skred_synth_last_error(void) { return g_last_err; }
int skred_synth_last_rc(void) { return g_last_rc; }
void skred_synth_set_device(int device) { g_device = device; }
void skred_synth_shutdown(void) {
  if (g_bank) { skred_bank_destroy(g_bank); g_bank = NULL; }
  free(g_pool); g_pool = NULL; g_pool_floats = g_pool_cap = 0;
  memset(g_seen, 0, sizeof(g_seen));
}

static int note(int rc, const char *what) {
  g_last_rc = rc;
  if (rc) snprintf(g_last_err, sizeof(g_last_err), "%s: %s", what, skred_amd_last_error());
  return rc;
}

/* Concatenate the tables the voices point at into one float pool (the bank's replacement for raw
 * table pointers).  Returns 1 when the pool had to be rebuilt: some voice_table[]/size changed. */
static int pool_refresh(void) {
  int same = g_pool != NULL;
  for (int v = 0; v < NV && same; v++)
    same = g_seen[v].ptr == voice_table[v] && g_seen[v].size == voice_table_size[v];
  if (same) return 0;
  size_t need = 0;
  for (int v = 0; v < NV; v++)
    if (voice_table[v] && voice_table_size[v] > 0) need += (size_t)voice_table_size[v];
  if (need == 0) need = 1;
  if (need > g_pool_cap) { free(g_pool); g_pool = (float *)malloc(need * sizeof(float)); g_pool_cap = need; }
  size_t pos = 0;
  if (pos == 0 && need == 1) g_pool[0] = 0.0f;
  for (int v = 0; v < NV; v++) {
    g_seen[v].ptr = voice_table[v]; g_seen[v].size = voice_table_size[v];
    g_offset[v] = 0;
    if (!(voice_table[v] && voice_table_size[v] > 0)) continue;
    int shared = -1;                        /* voices on the same table share one copy */
    for (int u = 0; u < v; u++)
      if (voice_table[u] == voice_table[v] && voice_table_size[u] == voice_table_size[v]) { shared = u; break; }
    if (shared >= 0) { g_offset[v] = g_offset[shared]; continue; }
    memcpy(g_pool + pos, voice_table[v], sizeof(float) * (size_t)voice_table_size[v]);
    g_offset[v] = (int64_t)pos;
    pos += (size_t)voice_table_size[v];
  }
  g_pool_floats = pos ? pos : 1;
  return 1;
}

static void bank_view(skred_voice_bank_t *b) {
  memset(b, 0, sizeof(*b));
  b->n_voices = NV;
  b->voice_phase = voice_phase; b->voice_phase_inc = voice_phase_inc;
  b->voice_table_offset = g_offset; b->voice_table_size = voice_table_size;
  b->voice_one_shot = voice_one_shot; b->voice_finished = voice_finished;
  b->voice_loop_enabled = voice_loop_enabled; b->voice_loop_valid = voice_loop_valid;
  b->voice_loop_start_f = voice_loop_start_f; b->voice_loop_end_f = voice_loop_end_f;
  b->voice_direction = voice_direction; b->voice_wave_table_index = voice_wave_table_index;
  b->voice_sample = voice_sample; b->voice_sample_hold = voice_sample_hold;
  b->voice_sample_hold_count = voice_sample_hold_count; b->voice_sample_hold_max = voice_sample_hold_max;
  b->voice_quantize = voice_quantize; b->voice_amp = voice_amp;
  b->voice_use_amp_envelope = voice_use_amp_envelope;
  b->voice_smoother_enable = voice_smoother_enable; b->voice_smoother_gain = voice_smoother_gain;
  b->voice_smoother_smoothing = voice_smoother_smoothing; b->voice_filter_mode = voice_filter_mode;
  b->voice_filter = voice_filter; b->voice_amp_envelope = voice_amp_envelope;
  b->voice_pan_left = voice_pan_left; b->voice_pan_right = voice_pan_right;
  b->voice_disconnect = voice_disconnect;
  b->voice_freq_mod_osc = voice_freq_mod_osc; b->voice_freq_mod_depth = voice_freq_mod_depth;
  b->voice_freq_scale = voice_freq_scale;
  b->voice_amp_mod_osc = voice_amp_mod_osc; b->voice_amp_mod_depth = voice_amp_mod_depth;
  b->voice_pan_mod_osc = voice_pan_mod_osc; b->voice_pan_mod_depth = voice_pan_mod_depth;
  b->voice_cz_mod_osc = voice_cz_mod_osc; b->voice_cz_mod_depth = voice_cz_mod_depth;
  b->voice_cz_mode = voice_cz_mode; b->voice_cz_distortion = voice_cz_distortion;
}

static void silence(float *buffer, int num_frames, int num_channels) {
  for (int i = 0; i < num_frames; i++) buffer[(size_t)i * num_channels] = buffer[(size_t)i * num_channels + 1] = 0.0f;
}

/* The miniaudio data callback's only callee (skred.c:116).  Same contract as synth.c:502-630:
 * fills channels 0/1 of `buffer`, writes the per-voice stems into the `user` buffer latched on the
 * first call, advances synth_sample_count and every voice's recurrences. */
void synth(float *buffer, float *input, int num_frames, int num_channels, void *user) {
  (void)input;
  if (g_first) {
    synth_frames_per_callback = num_frames;
    g_noise_rng = 1;
    g_stems = (float *)user;
    g_first = 0;
  }
  const int slot = (int)(g_calls % TIMING_RING);
  clock_gettime(CLOCK_MONOTONIC, &g_ring[slot].t0);
  g_ring[slot].frames = num_frames; g_ring[slot].order = g_calls; g_ring[slot].filled = 0;
  for (int v = 0; v < NV; v++)                       /* `:m` latency marker, synth.c:527-530 */
    if (voice_mark_go[v]) { clock_gettime(CLOCK_MONOTONIC_COARSE, &voice_mark_b[v]); voice_mark_go[v] = 0; }

  int rc = SKRED_OK;
  if (num_frames <= 0 || num_channels < 2 || !buffer) { note(SKRED_E_BAD_ARG, "synth arguments"); return; }
  if (!g_bank) rc = note(skred_bank_create(g_device, NV, &g_bank), "skred_bank_create");
  if (!rc && pool_refresh()) rc = note(skred_bank_set_tables_f32(g_bank, g_pool, g_pool_floats), "skred_bank_set_tables_f32");
  skred_voice_bank_t view;
  bank_view(&view);
  skred_globals_t g = { synth_sample_count, g_noise_rng, volume_final, volume_smoother_gain, volume_smoother_smoothing, 0.0f };
  if (!rc) rc = note(skred_bank_upload(g_bank, &view, 0, 0, NV), "skred_bank_upload");
  if (!rc) rc = note(skred_bank_set_globals(g_bank, &g), "skred_bank_set_globals");
  if (!rc) rc = note(skred_bank_render_host(g_bank, buffer, num_frames, num_channels, SKRED_INTERP_TRUNCATE, g_stems),
                     "skred_bank_render_host");
  if (!rc) rc = note(skred_bank_download(g_bank, &view, 0, 0, NV), "skred_bank_download");
  if (!rc) rc = note(skred_bank_get_globals(g_bank, &g), "skred_bank_get_globals");
  if (!rc) {
    synth_sample_count = g.synth_sample_count;
    g_noise_rng = g.noise_rng;
    volume_smoother_gain = g.volume_smoother_gain;
    g_ring[slot].gpu_ms = skred_bank_last_render_ms(g_bank);
  } else {
    silence(buffer, num_frames, num_channels);
    synth_sample_count += (uint64_t)num_frames;      /* keep the control-path clock moving */
  }
  clock_gettime(CLOCK_MONOTONIC, &g_ring[slot].t1);
  g_ring[slot].filled = 1;
  g_calls++;
}

/* callback timing ring as text: "# order frames elapsed-ms budget-ms", synth.c:462-480 (+ GPU kernel ms) */
char *synth_stats(void) {
  static char text[65536];
  char *p = text;
  *p = '\0';
  for (int i = 0; i < TIMING_RING; i++) {
    if (!g_ring[i].filled) continue;
    const double budget = (double)g_ring[i].frames / (double)RATE * 1000.0;
    const double ms = ns_between(&g_ring[i].t0, &g_ring[i].t1) / 1000000.0;
    PUT("# %lld %d %gms %gms (gpu %gms)\n", g_ring[i].order, g_ring[i].frames, ms, budget, g_ring[i].gpu_ms);
    g_ring[i].filled = 0;
  }
  return text;
}
