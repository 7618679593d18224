/*
 * skred_patch.c -- `.sk` patch subset -> voice state (SURVEY §8f "next" #1).
 *
 * A .sk file is wire text: one-to-four character atoms followed by comma-separated numbers
 * (`v0 w0 f440 a4 F1,10`), tokenised by skode.c:283-429 and dispatched by the big switch of
 * wire.c:591-867.  This is a small from-scratch reader for the subset that describes VOICES --
 * exactly the atoms that end in a call of the control path in skred_synth_dropin.c:
 *
 *     v w f a p n t l J K Q b B T m s S h q c C F A P V N g G H L r / >
 *
 * Everything else (sequencer patterns x y z Z M % ! @, strings {..}, arrays (..), deferred
 * chunks +n ~n, system atoms :x /x, variables $n) is counted as unsupported and skipped, never
 * half-executed.  Same dispatch rules as the reference: an atom fires when the next atom (or the
 * end of the chunk) arrives, with the numbers collected since; numbers are [0-9.-]+ read by strtod.
 * tests/test_patch_loader.py compares the resulting state with reference wire() on every reference
 * patch that stays inside the subset.
 */
#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "skred_synth_abi.h"
#include "skred_wav.h"

#define MAX_ARGS 8                      /* ARG_MAX, skode.c:33 */

static int is_number_char(int c) { return isdigit(c) || c == '-' || c == '.'; }
static int is_atom_char(int c) { return isalpha(c) || (c && strchr("!@%^&*_=:\"'<>?/", c)); }

void skred_patch_init(skred_patch_t *p) { p->voice = 0; p->unsupported = 0; p->errors = 0; }

/* one atom with its arguments: wire.c:606-759 for the voice subset */
/* `:wN,slot,ch` -- load N.wav from the current directory into an EXT slot: the reference's wave_load
 * (wire.c:406-441) on this library's wave arrays, with skred_wav_get in place of mw_get.  A table that
 * is replaced may still be some voice's voice_table, so it is parked in a small ring and freed eight
 * loads later, as the reference does (wire.c:370-390,417-426). */
#define SKRED_EXT_SAMPLE_000 200                              /* skred.h:70 */
#define SKRED_EXT_SAMPLE_999 (200 + 999)                      /* skred.h:71 */
#define SKRED_RETIRED 8
static float *retired[SKRED_RETIRED];
static int retired_ptr = 0;

int skred_wave_load(int which, int where, int ch) {
  if (where < SKRED_EXT_SAMPLE_000 || where >= SKRED_EXT_SAMPLE_999) return SKRED_ERR_INVALID_EXT_SAMPLE;
  char name[64];
  snprintf(name, sizeof(name), "%d.wav", which);
  skred_wav_info_t info;
  int len = 0;
  float *table = skred_wav_get(name, &len, &info, ch);
  if (!table) return SKRED_ERR_INVALID_EXT_SAMPLE;
  if (wave_table_data[where]) {
    if (retired_ptr >= SKRED_RETIRED) retired_ptr = 0;
    free(retired[retired_ptr]);
    retired[retired_ptr++] = wave_table_data[where];
  }
  wave_is_miniwav[where] = 1;
  wave_table_data[where] = table;
  wave_size[where] = len;
  wave_rate[where] = (float)info.sample_rate;
  wave_one_shot[where] = 1;
  wave_loop_enabled[where] = 0;
  wave_loop_start[where] = 1;
  wave_loop_end[where] = len;
  wave_midi_note[where] = 69;
  wave_offset_hz[where] = (float)len / (float)info.sample_rate * 440.0f;
  return 0;
}

static void dispatch(skred_patch_t *p, const char *atom, const double *arg, int argc) {
  const int v = p->voice;
  const int x = argc ? (int)arg[0] : 0;
  int rc = 0;
  if ((atom[0] == ':' || atom[0] == '/') && atom[1] == 'w' && atom[2] == '\0') {   /* wire.c:801-814 */
    const int which = argc >= 1 ? (int)arg[0] : 0;
    const int where = argc >= 2 ? (int)arg[1] : SKRED_EXT_SAMPLE_000;
    const int ch = argc > 2 ? (int)arg[2] : -1;
    if (skred_wave_load(which, where, ch)) p->errors++;
    return;
  }
  if (atom[1] != '\0') { p->unsupported++; return; }          /* other multi-character atoms: system / sequencer */
  switch (atom[0]) {
    case 'a': if (argc) rc = amp_set(v, (float)arg[0]); break;
    case 'A': if (argc == 1) rc = amp_mod_set(v, -1, 0); else if (argc > 1) rc = amp_mod_set(v, x, (float)arg[1]); break;
    case 'b': rc = wave_dir(v, argc ? x : -1); break;
    case 'B': rc = wave_loop(v, argc ? x : -1); break;
    case 'c': rc = cz_set(v, argc ? x : 0, argc > 1 ? (float)arg[1] : .5f); break;
    case 'C': rc = cmod_set(v, x, argc > 1 ? (float)arg[1] : -1.0f); break;
    case 'f': if (argc) rc = freq_set(v, (float)arg[0]); break;
    case 'F': rc = freq_mod_set(v, x, argc > 1 ? (float)arg[1] : -1.0f); break;
    case 'g': if (argc) { if (arg[0] <= 0) voice_glissando_enable[v] = 0;
                          else { voice_glissando_enable[v] = 1; voice_glissando_speed[v] = (float)arg[0]; } } break;
    case 'G': if (argc) { voice_link_midi_a[v] = (float)x; if (argc > 1) voice_link_midi_b[v] = (float)(int)arg[1]; } break;
    case 'h': if (argc) voice_sample_hold_max[v] = x; break;
    case 'H': if (argc) { voice_link_velo_a[v] = (float)x; if (argc > 1) voice_link_velo_b[v] = (float)(int)arg[1]; } break;
    case 'J': if (argc) { voice_filter_mode[v] = x; mmf_set_params(v, voice_filter_freq[v], voice_filter_res[v]); } break;
    case 'K': if (argc) rc = mmf_set_freq(v, (float)arg[0]); break;
    case 'l': if (argc) {
        rc = envelope_velocity(v, (float)arg[0]);
        if (voice_link_velo_a[v] >= 0) envelope_velocity((int)voice_link_velo_a[v], (float)arg[0]);
        if (voice_link_velo_b[v] >= 0) envelope_velocity((int)voice_link_velo_b[v], (float)arg[0]);
      } break;
    case 'L': if (argc) voice_link_trig[v] = (float)x; break;
    case 'm': if (argc) rc = wave_mute(v, x); break;
    case 'n': if (argc) {
        rc = freq_midi(v, (float)arg[0]);
        if (voice_link_midi_a[v] >= 0) freq_midi((int)voice_link_midi_a[v], (float)arg[0]);
        if (voice_link_midi_b[v] >= 0) freq_midi((int)voice_link_midi_b[v], (float)arg[0]);
      } break;
    case 'N': if (argc) voice_midi_transpose[v] = (float)arg[0]; break;
    case 'p': if (argc) rc = pan_set(v, (float)arg[0]); break;
    case 'P': rc = pan_mod_set(v, x, argc > 1 ? (float)arg[1] : -1.0f); break;
    case 'q': if (argc) rc = wave_quant(v, x); break;
    case 'Q': if (argc) rc = mmf_set_res(v, (float)arg[0]); break;
    case 'r': if (argc) voice_record[v] = x; break;
    case 's': if (argc) { if (arg[0] <= 0) voice_smoother_enable[v] = 0;
                          else { voice_smoother_enable[v] = 1; voice_smoother_smoothing[v] = (float)arg[0]; } } break;
    case 'S': if (argc) rc = wave_reset(v, x); break;
    case 't': if (argc > 3) rc = envelope_set(v, (float)arg[0], (float)arg[1], (float)arg[2], (float)arg[3]); break;
    case 'T': rc = voice_trigger(v); if (voice_link_trig[v] > 0) voice_trigger((int)voice_link_trig[v]); break;
    case 'v': if (argc) rc = voice_set(x, &p->voice); break;
    case 'V': if (argc) rc = volume_set((float)arg[0]); break;
    case 'w': if (argc) rc = wave_set(v, x); break;
    case '/': rc = wave_default(v); break;
    case '>': if (argc) rc = voice_copy(v, x); break;
    default: p->unsupported++; return;
  }
  if (rc) p->errors++;
}

/* Feed one line (or several chunks separated by ';').  Returns the number of unsupported tokens met. */
int skred_patch_line(skred_patch_t *p, const char *line) {
  const int before = p->unsupported;
  char atom[5] = "";
  int have_atom = 0, argc = 0;
  double arg[MAX_ARGS];
  const char *s = line;
#define FIRE() do { if (have_atom) dispatch(p, atom, arg, argc); have_atom = 0; argc = 0; } while (0)
  while (*s) {
    const unsigned char c = (unsigned char)*s;
    if (c == '#') { while (*s && *s != '\n' && *s != ';') s++; continue; }       /* comment */
    if (c == ';' || c == 0x04) { FIRE(); s++; continue; }                          /* chunk end */
    if (isspace(c) || c == ',') { s++; continue; }
    if (is_number_char(c)) {
      char buf[64]; int n = 0;
      while (*s && is_number_char((unsigned char)*s)) { if (n < 63) buf[n++] = *s; s++; }
      buf[n] = '\0';
      if (argc < MAX_ARGS) arg[argc++] = strtod(buf, NULL);
      continue;
    }
    if (c == '{') { FIRE(); p->unsupported++; while (*s && *s != '}') s++; if (*s) s++; continue; }
    if (c == '(') { FIRE(); p->unsupported++; while (*s && *s != ')') s++; if (*s) s++; continue; }
    if (c == '+' || c == '~') { FIRE(); p->unsupported++; while (*s && *s != ';') s++; continue; } /* deferred chunk */
    if (c == '$' || c == '[' || c == ']') { p->unsupported++; s++; continue; }
    if (is_atom_char(c)) {
      FIRE();                                          /* the previous atom fires when the next one starts */
      int n = 0;
      while (*s && is_atom_char((unsigned char)*s)) { if (n < 4) atom[n++] = *s; s++; }
      atom[n] = '\0';
      have_atom = 1;
      continue;
    }
    s++;                                               /* anything else: ignore, like the reference's tokenizer */
  }
  FIRE();
#undef FIRE
  return p->unsupported - before;
}

/* Load a patch file line by line (wire.c:342-368 does the same through wire()). */
int skred_patch_load(const char *path, skred_patch_t *p) {
  FILE *f = fopen(path, "r");
  if (!f) return -1;
  char line[4096];
  while (fgets(line, sizeof(line), f)) skred_patch_line(p, line);
  fclose(f);
  return p->unsupported;
}
