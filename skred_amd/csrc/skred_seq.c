/*
 * skred_seq.c -- the pattern step clock of the reference's sequencer (seq.c:179-213) for device-resident voices
 * (include/skred_amd.h: skred_seq_* and skred_bank_pattern_*; SURVEY 8f "next" #4).
 *
 * In the reference seq(frame_count) runs after synth() in the audio callback (skred.c:119).  Its second half is a
 * tempo accumulator -- the block's duration is added to a clock, and when the clock reaches the time of one step every
 * RUNNING pattern advances: its modulo divides the step rate, a muted step is skipped silently, the pointer wraps at the
 * first empty step -- and each step that fires runs a line of wire text, i.e. a few stores into the voice arrays.
 * Here the clock and the pattern bookkeeping are the same arithmetic on the host (skred_seq_t: no device involved, so
 * the CPU tests can hold it against the compiled reference call by call), and what a step "runs" is a batch of voice
 * updates captured when the step was written (skred_bank_pattern_step_set), applied through the dirty-field protocol of
 * skred_bank_update.c by skred_bank_run_queue() -- which is thereby all of seq(): deferred items first, then the clock.
 */
#include <stdlib.h>
#include <string.h>

#include "skred_bank_priv.h"

struct skred_seq {
  double clock_sec;                                   /* seq.c:182 `static double clock_sec` */
  float time_per_step;                                /* tempo_time_per_step, skred.c:47 (60 s until a tempo is set) */
  float tempo_base, tempo_bpm;
  int pointer[SKRED_PATTERNS_MAX], counter[SKRED_PATTERNS_MAX], state[SKRED_PATTERNS_MAX], modulo[SKRED_PATTERNS_MAX];
  uint8_t occupied[SKRED_PATTERNS_MAX][SKRED_SEQ_STEPS_MAX];   /* the step's text is not empty */
  uint8_t mute[SKRED_PATTERNS_MAX][SKRED_SEQ_STEPS_MAX];
};

static int bad_pattern(int p) { return p < 0 || p >= SKRED_PATTERNS_MAX; }
static int bad_step(int s) { return s < 0 || s >= SKRED_SEQ_STEPS_MAX; }

int skred_seq_pattern_reset(skred_seq_t *s, int p) {        /* pattern_reset, seq.c:216-225 */
  if (!s || bad_pattern(p)) return fail(SKRED_E_BAD_ARG, "seq_pattern_reset: pattern %d", p);
  s->pointer[p] = 0; s->counter[p] = 0;
  s->state[p] = SKRED_SEQ_STOPPED;
  s->modulo[p] = 4;
  memset(s->occupied[p], 0, sizeof(s->occupied[p]));
  memset(s->mute[p], 0, sizeof(s->mute[p]));
  return SKRED_OK;
}

int skred_seq_create(skred_seq_t **out) {
  if (!out) return fail(SKRED_E_BAD_ARG, "seq_create");
  skred_seq_t *s = (skred_seq_t *)calloc(1, sizeof(*s));
  if (!s) return fail(SKRED_E_NO_MEM, "calloc");
  s->time_per_step = 60.0f;                           /* skred.c:47 */
  s->tempo_bpm = 120.0f / 4.0f;                       /* skred.c:48 */
  for (int p = 0; p < SKRED_PATTERNS_MAX; p++) (void)skred_seq_pattern_reset(s, p);   /* seq_init, seq.c:227-232 */
  *out = s;
  return SKRED_OK;
}

void skred_seq_destroy(skred_seq_t *s) { free(s); }

int skred_seq_tempo_set(skred_seq_t *s, float m) {          /* tempo_set, seq.c:21-28: four steps per beat */
  if (!s) return fail(SKRED_E_BAD_ARG, "seq_tempo_set");
  s->tempo_base = m;
  s->tempo_bpm = m / 4.0;
  const float bps = m / 60.f;
  s->time_per_step = 1.0f / bps / 4.0f;
  return SKRED_OK;
}

float skred_seq_time_per_step(const skred_seq_t *s) { return s ? s->time_per_step : 0.0f; }

int skred_seq_step_set(skred_seq_t *s, int p, int step, int occupied) {   /* seq_step_set, seq.c:267-270: "" empties the step */
  if (!s || bad_pattern(p) || bad_step(step)) return fail(SKRED_E_BAD_ARG, "seq_step_set: pattern %d step %d", p, step);
  s->occupied[p][step] = occupied ? 1 : 0;
  return SKRED_OK;
}

int skred_seq_mute_set(skred_seq_t *s, int p, int step, int m) {
  if (!s || bad_pattern(p) || bad_step(step)) return fail(SKRED_E_BAD_ARG, "seq_mute_set: pattern %d step %d", p, step);
  s->mute[p][step] = m ? 1 : 0;
  return SKRED_OK;
}

int skred_seq_modulo_set(skred_seq_t *s, int p, int m) {
  if (!s || bad_pattern(p)) return fail(SKRED_E_BAD_ARG, "seq_modulo_set: pattern %d", p);
  s->modulo[p] = m;
  return SKRED_OK;
}

int skred_seq_state_set(skred_seq_t *s, int p, int state) {   /* seq_state_set, seq.c:273-291: 0 stop, 1 start, 2 pause, 3 resume */
  if (!s || bad_pattern(p)) return fail(SKRED_E_BAD_ARG, "seq_state_set: pattern %d", p);
  switch (state) {
    case 0: s->state[p] = SKRED_SEQ_STOPPED; s->pointer[p] = 0; break;
    case 1: s->state[p] = SKRED_SEQ_RUNNING; s->pointer[p] = 0; break;
    case 2: s->state[p] = SKRED_SEQ_PAUSED; break;
    case 3: s->state[p] = SKRED_SEQ_RUNNING; break;
    default: return fail(SKRED_E_BAD_ARG, "seq_state_set: state %d", state);
  }
  return SKRED_OK;
}

int skred_seq_pointer(const skred_seq_t *s, int p) { return (s && !bad_pattern(p)) ? s->pointer[p] : -1; }
int skred_seq_counter(const skred_seq_t *s, int p) { return (s && !bad_pattern(p)) ? s->counter[p] : -1; }

/* The second half of seq(frame_count), seq.c:179-213.  fired[] receives (pattern << 16) | step for every step whose text
 * the reference would run in this call, in pattern order; returns their number (at most SKRED_PATTERNS_MAX: the clock
 * advances one step per call at most, however long the block). */
int skred_seq_tick(skred_seq_t *s, int frame_count, float sample_rate, int32_t *fired, int max_fired) {
  if (!s || frame_count < 0 || !(sample_rate > 0.0f)) return fail(SKRED_E_BAD_ARG, "seq_tick: bad arguments");
  const float frame_time_sec = (float)frame_count / sample_rate;
  s->clock_sec += frame_time_sec;
  if (!(s->clock_sec >= s->time_per_step)) return 0;
  s->clock_sec -= s->time_per_step;
  int n = 0;
  for (int p = 0; p < SKRED_PATTERNS_MAX; p++) {
    if (s->state[p] != SKRED_SEQ_RUNNING) continue;
    if (s->modulo[p] > 1 && (s->counter[p] % s->modulo[p]) != 0) { s->counter[p]++; continue; }   /* this pattern sits the step out */
    s->counter[p]++;
    const int at = s->pointer[p];
    if (!s->mute[p][at] && s->occupied[p][at] && fired && n < max_fired) fired[n++] = (p << 16) | at;
    s->pointer[p] = at + 1;
    /* wrap at the first empty step.  (With all 256 steps written the reference reads one step past its array here;
     * this build wraps.) */
    if (s->pointer[p] >= SKRED_SEQ_STEPS_MAX || !s->occupied[p][s->pointer[p]]) s->pointer[p] = 0;
  }
  return n;
}
