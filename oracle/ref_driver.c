/*
 * oracle/ref_driver.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Thin driver linked next to the UNMODIFIED reference sources, which are
 * compiled where they lie under /root/reference by oracle/Makefile into
 * oracle/_ref/libskred_ref.so (never copied into this repository).  It only
 * exists so that tests/golden/gen_golden.py can drive the reference's own
 * synth()/synth_callback()/wire() from Python (ctypes) and dump golden
 * vectors.  Nothing under skred_amd/ may link or load it.
 *
 * What the reference itself provides inside the .so:
 *   synth.c seq.c wire.c skode.c udp.c miniwav.c util.c skred-mem.c
 *   miniaudio.c bestline.c and skred.c (built with -Dmain=skred_main so its
 *   globals -- debug, scope, tempo_*, rec_*, one_skred_frame -- and its
 *   synth_callback() (skred.c:107-152) are the reference's own).
 *
 * What this driver has to define, and why:
 *   pcm[], pcm_map[]  (declared amysamples.h:8-9).  amysamples.c cannot be
 *   built here: its sample blob notamy/pcm_samples_large.h is absent from the
 *   mount (/root/reference/.MISSING_LARGE_BLOBS).  The driver therefore owns
 *   an EMPTY sample map: wave_table_init() (synth.c:1270-1292) then registers
 *   zero-length AMY slots, i.e. slots 100..166 are unusable in the oracle.
 *   PCM-style cases use EXT slots (200+) filled through ref_ext_table(), the
 *   same direct assignment wire.c:381-399 performs.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "skred.h"
#include "synth-types.h"
#include "synth.h"
#include "wire.h"
#include "seq.h"
#include "miniaudio.h"
/* amysamples.h:10 ends in a dangling `extern`; the prototype that follows completes it. */
#include "amysamples.h"
int ref_voice_max(void);

/* empty AMY sample ROM: see header comment */
pcm_map_t pcm_map[PCM_SAMPLES];
int16_t pcm[PCM_LENGTH];

/* reference symbols defined in skred.c */
extern float one_skred_frame[];
void synth_callback(ma_device *pDevice, void *output, const void *input, ma_uint32 frame_count);

static ma_device g_dev;
static wire_t g_wire = WIRE();
static int g_booted = 0;

/* Same init order as skred.c:230-234 minus the recorder malloc and the
 * audio device (skred.c:237-251). */
void ref_boot(void) {
  if (g_booted) return;
  synth_init();
  wave_table_init();
  voice_init();
  seq_init();
  memset(&g_dev, 0, sizeof(g_dev));
  g_dev.playback.channels = AUDIO_CHANNELS;
  g_dev.pUserData = one_skred_frame;       /* skred.c:248 */
  g_wire.output = 0;
  g_booted = 1;
}

/* One control line through the reference's own parser (skred.c:341). */
int ref_wire(const char *line) {
  char buf[4096];
  strncpy(buf, line, sizeof(buf) - 1);
  buf[sizeof(buf) - 1] = '\0';
  return wire(buf, &g_wire);
}

/* Load N.sk from the reference tree via wire.c:342 (it fopen()s "N.sk"
 * relative to the cwd, so hop there and back). */
int ref_load_patch(const char *refdir, int n) {
  char old[4096];
  if (!getcwd(old, sizeof(old))) return -1;
  if (chdir(refdir) != 0) return -2;
  int r = sk_load(NULL, 0, n, 0);
  if (chdir(old) != 0) return -3;
  return r;
}

/* Exactly what miniaudio's device thread does: one data_callback call.
 * Runs synth() then seq() (skred.c:116,119). */
void ref_callback(float *out, unsigned frames) {
  synth_callback(&g_dev, out, NULL, (ma_uint32)frames);
}

/* synth() alone (no sequencer tick), for block-size experiments. */
void ref_synth_only(float *out, int frames) {
  synth(out, NULL, frames, AUDIO_CHANNELS, one_skred_frame);
}

float *ref_stems(void) { return one_skred_frame; }

/* Install a float table into an EXT slot the way wire.c:381-399 /
 * wire.c:427-436 do (direct assignment of the wave_* arrays). */
int ref_ext_table(int slot, const float *data, int len, float rate,
                  int one_shot, int loop_enabled, int loop_start, int loop_end,
                  float midi_note, float offset_hz) {
  if (slot < EXT_SAMPLE_000 || slot >= EXT_SAMPLE_999) return 1;
  float *t = (float *)malloc(sizeof(float) * (size_t)len);
  if (!t) return 2;
  memcpy(t, data, sizeof(float) * (size_t)len);
  wave_table_data[slot] = t;
  wave_size[slot] = len;
  wave_rate[slot] = rate;
  wave_one_shot[slot] = one_shot;
  wave_loop_enabled[slot] = loop_enabled;
  wave_loop_start[slot] = loop_start;
  wave_loop_end[slot] = loop_end;
  wave_midi_note[slot] = midi_note;
  wave_offset_hz[slot] = offset_hz;
  return 0;
}

/* sizes so the Python side can sanity-check its struct mirrors */
int ref_sizeof_mmf(void) { return (int)sizeof(mmf_t); }
int ref_sizeof_envelope(void) { return (int)sizeof(envelope_t); }
int ref_voice_max(void) { return VOICE_MAX; }
int ref_wave_table_max(void) { return WAVE_TABLE_MAX; }
int ref_sample_rate(void) { return MAIN_SAMPLE_RATE; }
