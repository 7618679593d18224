/*
 * sk_render -- BASELINE config 0 as a tiny C harness on the drop-in library: load a .sk patch,
 * call synth() in callback-sized blocks exactly as miniaudio would (skred.c:107-116, 512 frames,
 * skred.h:12), write the frames to a WAV file.  The reference has no offline render mode;
 * this is the build-side harness for it, running the render loop on the GPU.
 *
 *   sk_render [--patch FILE.sk | --patch-0sk] [--seconds S] [--block N] OUT.wav
 *
 * WAV: IEEE float32, stereo, 44100 Hz (MAIN_SAMPLE_RATE, skred.h:6) -- the callback's own sample format,
 * so the file holds the callback output bit for bit.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "skred_synth_abi.h"

static const char *PATCH_0SK[] = { "S100", "v0 w0 f440 a4 F1,10", "v1 w0 f1 a50 m1" };  /* reference patch 0.sk */

static void put32(FILE *f, uint32_t v) { fwrite(&v, 4, 1, f); }
static void put16(FILE *f, uint16_t v) { fwrite(&v, 2, 1, f); }

static int write_wav_f32(const char *path, const float *frames, uint32_t n_frames, uint32_t rate) {
  FILE *f = fopen(path, "wb");
  if (!f) return -1;
  const uint32_t bytes = n_frames * 2 * 4;
  fwrite("RIFF", 1, 4, f); put32(f, 36 + bytes); fwrite("WAVE", 1, 4, f);
  fwrite("fmt ", 1, 4, f); put32(f, 16); put16(f, 3 /* IEEE float */); put16(f, 2);
  put32(f, rate); put32(f, rate * 8); put16(f, 8); put16(f, 32);
  fwrite("data", 1, 4, f); put32(f, bytes);
  fwrite(frames, 4, (size_t)n_frames * 2, f);
  fclose(f);
  return 0;
}

int main(int argc, char **argv) {
  const char *patch = NULL, *out = NULL;
  int builtin = 0, block = 512;
  double seconds = 1.0;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--patch") && i + 1 < argc) patch = argv[++i];
    else if (!strcmp(argv[i], "--patch-0sk")) builtin = 1;
    else if (!strcmp(argv[i], "--seconds") && i + 1 < argc) seconds = atof(argv[++i]);
    else if (!strcmp(argv[i], "--block") && i + 1 < argc) block = atoi(argv[++i]);
    else out = argv[i];
  }
  if (!out || (!patch && !builtin) || block <= 0) {
    fprintf(stderr, "usage: sk_render [--patch FILE.sk | --patch-0sk] [--seconds S] [--block N] OUT.wav\n");
    return 2;
  }
  synth_init();
  wave_table_init();
  voice_init();
  skred_patch_t p;
  skred_patch_init(&p);
  if (builtin) {
    for (size_t i = 0; i < sizeof(PATCH_0SK) / sizeof(PATCH_0SK[0]); i++) skred_patch_line(&p, PATCH_0SK[i]);
  } else if (skred_patch_load(patch, &p) < 0) {
    fprintf(stderr, "cannot open %s\n", patch);
    return 1;
  }
  if (p.unsupported) fprintf(stderr, "# %d token(s) outside the voice subset were skipped\n", p.unsupported);

  const uint32_t total = (uint32_t)(seconds * SKRED_MAIN_SAMPLE_RATE);
  float *frames = (float *)calloc((size_t)total * 2, sizeof(float));
  float *stems = (float *)calloc((size_t)block * 2 * SKRED_VOICE_MAX, sizeof(float));   /* the `user` buffer */
  if (!frames || !stems) return 1;
  for (uint32_t done = 0; done < total;) {
    const int n = (int)(total - done < (uint32_t)block ? total - done : (uint32_t)block);
    synth(frames + (size_t)done * 2, NULL, n, 2, stems);
    if (skred_synth_last_rc() != 0) {
      fprintf(stderr, "synth() failed: %s\n", skred_synth_last_error());   /* no GPU -> no audio: never a CPU fallback */
      return 1;
    }
    done += (uint32_t)n;
  }
  if (write_wav_f32(out, frames, total, SKRED_MAIN_SAMPLE_RATE) != 0) { fprintf(stderr, "cannot write %s\n", out); return 1; }
  printf("# wrote %s: %u frames, stereo f32, %d Hz\n%s", out, total, SKRED_MAIN_SAMPLE_RATE, synth_stats());
  synth_free();
  free(frames); free(stems);
  return 0;
}
