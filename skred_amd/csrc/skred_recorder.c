/*
 * skred_recorder.c -- the per-voice stem recorder, device resident (include/skred_wav.h).
 *
 * State machine of the reference, kept one to one (skred.c:84-104,120-131; wire.c:816-849):
 *   create  = synth_callback_init(max_sec)      rec buffer sized for the longest take
 *   start   = `<sec`                            rec_ptr = 0, rec_state = 1, optional shorter limit
 *   append  = the loop in synth_callback        copy this callback's stems while rec_state, stop when full
 *   save    = `*` -> save_wav                   two 16-bit channels per selected voice, min/max-scaled
 * with the buffer in HBM and the two passes of save_wav as kernels (skred_rec_kernels.hip).  Frames
 * are counted whole: the reference can stop in the middle of a frame when rec_max is not a multiple
 * of 2*VOICE_MAX, but save_wav only ever uses rec_ptr / VOICE_MAX / AUDIO_CHANNELS whole frames.
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "skred_amd.h"
#include "skred_launch.h"
#include "skred_wav.h"

int skred_amd_set_error(int code, const char *fmt, ...);
#define fail skred_amd_set_error
#define HIP_TRY(call)                                                              \
  do {                                                                             \
    hipError_t e_ = (call);                                                        \
    if (e_ != hipSuccess)                                                          \
      return fail(SKRED_E_NO_DEVICE, "%s -> %s", #call, hipGetErrorString(e_));   \
  } while (0)

struct skred_recorder {
  int device, n_voices;
  long capacity, limit, frames;   /* in frames */
  int recording;
  float *d_rec;                   /* [capacity][n_voices][2] */
  float *d_partial;               /* min/max partials */
  int *d_sel;                     /* selected voice ids */
  int16_t *d_pcm;                 /* conversion output, grown on demand */
  size_t pcm_cap;
};

int skred_recorder_create(skred_recorder_t **out, int device, int n_voices, long capacity_frames) {
  if (!out || n_voices <= 0 || capacity_frames <= 0) return fail(SKRED_E_BAD_ARG, "skred_recorder_create: bad arguments");
  *out = NULL;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(SKRED_E_NO_DEVICE, "no HIP device visible (the recorder has no CPU path)");
  if (device < 0 || device >= ndev) return fail(SKRED_E_BAD_ARG, "device %d of %d", device, ndev);
  HIP_TRY(hipSetDevice(device));
  skred_recorder_t *r = (skred_recorder_t *)calloc(1, sizeof(*r));
  if (!r) return fail(SKRED_E_NO_MEM, "calloc");
  r->device = device;
  r->n_voices = n_voices;
  r->capacity = r->limit = capacity_frames;
  const size_t bytes = (size_t)capacity_frames * n_voices * 2 * sizeof(float);
  if (hipMalloc((void **)&r->d_rec, bytes) != hipSuccess || hipMalloc((void **)&r->d_partial, (size_t)sk_rec_partial_floats() * sizeof(float)) != hipSuccess ||
      hipMalloc((void **)&r->d_sel, (size_t)n_voices * sizeof(int)) != hipSuccess) {
    skred_recorder_destroy(r);
    return fail(SKRED_E_NO_MEM, "hipMalloc of %zu bytes for the recording", bytes);
  }
  *out = r;
  return SKRED_OK;
}

void skred_recorder_destroy(skred_recorder_t *r) {
  if (!r) return;
  (void)hipSetDevice(r->device);
  if (r->d_rec) (void)hipFree(r->d_rec);
  if (r->d_partial) (void)hipFree(r->d_partial);
  if (r->d_sel) (void)hipFree(r->d_sel);
  if (r->d_pcm) (void)hipFree(r->d_pcm);
  free(r);
}

int skred_recorder_start(skred_recorder_t *r, long max_frames) {
  if (!r) return fail(SKRED_E_BAD_ARG, "skred_recorder_start");
  if (max_frames > 0) r->limit = max_frames < r->capacity ? max_frames : r->capacity;   /* wire.c:819-826 */
  r->frames = 0;
  r->recording = 1;
  return SKRED_OK;
}

void skred_recorder_stop(skred_recorder_t *r) { if (r) r->recording = 0; }
int skred_recorder_recording(const skred_recorder_t *r) { return r ? r->recording : 0; }
long skred_recorder_frames(const skred_recorder_t *r) { return r ? r->frames : 0; }

int skred_recorder_append(skred_recorder_t *r, const float *d_stems, int frames, void *stream) {
  if (!r || !d_stems || frames < 0) return fail(SKRED_E_BAD_ARG, "skred_recorder_append");
  if (!r->recording || frames == 0) return SKRED_OK;
  long n = r->limit - r->frames;
  if (n > frames) n = frames;
  if (n > 0) {
    HIP_TRY(hipSetDevice(r->device));
    const size_t stride = (size_t)r->n_voices * 2;
    HIP_TRY(hipMemcpyAsync(r->d_rec + (size_t)r->frames * stride, d_stems, (size_t)n * stride * sizeof(float),
                           hipMemcpyDeviceToDevice, (hipStream_t)stream));
    r->frames += n;
  }
  if (r->frames >= r->limit) r->recording = 0;        /* skred.c:126-129 */
  return SKRED_OK;
}

/* device passes of save_wav; on success *n_sel_out voices x frames x 2 int16 sit in r->d_pcm */
static int convert_on_device(skred_recorder_t *r, const int *record, int *n_sel_out) {
  int *sel = (int *)malloc((size_t)r->n_voices * sizeof(int));
  if (!sel) return fail(SKRED_E_NO_MEM, "malloc");
  int n_sel = 0;
  for (int v = 0; v < r->n_voices; ++v)
    if (record[v]) sel[n_sel++] = v;
  *n_sel_out = n_sel;
  if (n_sel == 0 || r->frames == 0) { free(sel); return SKRED_OK; }
  hipError_t e = hipSetDevice(r->device);
  /* the appends (and the renders that produced the stems) were queued on the caller's stream, which may be
   * non-blocking: nothing else orders the two passes below, on the null stream, behind them */
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpy(r->d_sel, sel, (size_t)n_sel * sizeof(int), hipMemcpyHostToDevice);
  free(sel);
  if (e != hipSuccess) return fail(SKRED_E_NO_DEVICE, "selection upload -> %s", hipGetErrorString(e));
  const size_t need = (size_t)r->frames * n_sel * 2;
  if (need > r->pcm_cap) {
    if (r->d_pcm) (void)hipFree(r->d_pcm);
    r->d_pcm = NULL;
    r->pcm_cap = 0;
    HIP_TRY(hipMalloc((void **)&r->d_pcm, need * sizeof(int16_t)));
    r->pcm_cap = need;
  }
  /* pass 1: min / max over every voice's samples of the take (wire.c:150-156) */
  int n_blocks = 0;
  e = (hipError_t)sk_launch_rec_minmax(r->d_rec, (size_t)r->frames * r->n_voices * 2, r->d_partial, &n_blocks, 0);
  if (e != hipSuccess) return fail(SKRED_E_NO_DEVICE, "minmax launch -> %s", hipGetErrorString(e));
  float *part = (float *)malloc((size_t)n_blocks * 2 * sizeof(float));
  if (!part) return fail(SKRED_E_NO_MEM, "malloc");
  e = hipMemcpy(part, r->d_partial, (size_t)n_blocks * 2 * sizeof(float), hipMemcpyDeviceToHost);
  if (e != hipSuccess) { free(part); return fail(SKRED_E_NO_DEVICE, "minmax readback -> %s", hipGetErrorString(e)); }
  float fbig = 0.0f, fsmall = 0.0f;
  for (int i = 0; i < n_blocks; ++i) {
    if (part[2 * i] > fbig) fbig = part[2 * i];
    if (part[2 * i + 1] < fsmall) fsmall = part[2 * i + 1];
  }
  free(part);
  /* wire.c:161-166 */
  float scale;
  if (fabsf(fsmall) > fabsf(fbig)) scale = -1.0f / fsmall;
  else scale = 1.0f / fbig;
  /* pass 2 (wire.c:170-180) */
  e = (hipError_t)sk_launch_rec_convert(r->d_rec, r->frames, r->n_voices, r->d_sel, n_sel, scale, r->d_pcm, 0);
  if (e != hipSuccess) return fail(SKRED_E_NO_DEVICE, "convert launch -> %s", hipGetErrorString(e));
  HIP_TRY(hipDeviceSynchronize());
  return SKRED_OK;
}

long skred_recorder_convert(skred_recorder_t *r, const int *record, int16_t *out, long out_capacity) {
  if (!r || !record || !out) return fail(SKRED_E_BAD_ARG, "skred_recorder_convert");
  int n_sel = 0;
  const int rc = convert_on_device(r, record, &n_sel);
  if (rc) return rc;
  const long n = r->frames * n_sel * 2;
  if (n == 0) return 0;
  if (n > out_capacity) return fail(SKRED_E_RANGE, "convert: %ld samples, room for %ld", n, out_capacity);
  if (hipMemcpy(out, r->d_pcm, (size_t)n * sizeof(int16_t), hipMemcpyDeviceToHost) != hipSuccess)
    return fail(SKRED_E_NO_DEVICE, "pcm readback failed");
  return n;
}

static void put32(FILE *f, uint32_t v) { fwrite(&v, 4, 1, f); }
static void put16(FILE *f, uint16_t v) { fwrite(&v, 2, 1, f); }

int skred_recorder_save_wav(skred_recorder_t *r, const char *filename, const int *record, int sample_rate) {
  if (!r || !filename || !record || sample_rate <= 0) return fail(SKRED_E_BAD_ARG, "skred_recorder_save_wav");
  int n_sel = 0;
  const int rc = convert_on_device(r, record, &n_sel);
  if (rc) return rc;
  if (n_sel == 0 || r->frames == 0) return SKRED_OK;          /* nothing to record: no file (wire.c:106-109) */
  const size_t n = (size_t)r->frames * n_sel * 2;
  int16_t *pcm = (int16_t *)malloc(n * sizeof(int16_t));
  if (!pcm) return fail(SKRED_E_NO_MEM, "malloc of %zu bytes", n * sizeof(int16_t));
  if (hipMemcpy(pcm, r->d_pcm, n * sizeof(int16_t), hipMemcpyDeviceToHost) != hipSuccess) {
    free(pcm);
    return fail(SKRED_E_NO_DEVICE, "pcm readback failed");
  }
  FILE *f = fopen(filename, "wb");
  if (!f) { free(pcm); return fail(SKRED_E_IO, "cannot open %s", filename); }
  /* header fields and their order: wire.c:117-146 */
  const int num_channels = 2 * n_sel, bits = 16;
  const int block_align = num_channels * bits / 8;
  const int byte_rate = sample_rate * block_align;
  const int data_size = (int)(r->frames * block_align);
  fwrite("RIFF", 1, 4, f); put32(f, 36u + (uint32_t)data_size); fwrite("WAVE", 1, 4, f);
  fwrite("fmt ", 1, 4, f); put32(f, 16); put16(f, 1); put16(f, (uint16_t)num_channels);
  put32(f, (uint32_t)sample_rate); put32(f, (uint32_t)byte_rate); put16(f, (uint16_t)block_align); put16(f, bits);
  fwrite("data", 1, 4, f); put32(f, (uint32_t)data_size);
  const size_t w = fwrite(pcm, sizeof(int16_t), n, f);
  free(pcm);
  if (fclose(f) != 0 || w != n) return fail(SKRED_E_IO, "short write to %s", filename);
  return SKRED_OK;
}
