/*
 * include/skred_wav.h -- the steps either side of the render path (SURVEY 8f "next" #3):
 * WAV files in (sample tables for EXT slots) and the per-voice stem recorder out.
 *
 * Reference interfaces restated here:
 *   miniwav.h:48  float *mw_get(char *name, int *frames_out, wav_t *w, int ch)   -> skred_wav_get
 *   miniwav.h:50  float *mw_free(float *f)                                       -> skred_wav_free
 *   wire.c:406-441 int wave_load(wire_t*, int which, int where, int ch)          -> skred_wave_load
 *                                                              (libskred_synth.so, skred_synth_abi.h)
 *   skred.c:84-104,120-131  rec_state / rec_ptr / rec_max / recording, synth_callback_init/_free,
 *                           the append loop of the audio callback                -> skred_recorder_*
 *   wire.c:94-185  save_wav(w, filename, samples, num_samples, record, max)      -> skred_recorder_save_wav
 *
 * The reference decodes with miniaudio (any container); this build reads RIFF/WAVE only
 * (PCM 8/16/24/32 bit, IEEE float 32/64, plain or WAVE_FORMAT_EXTENSIBLE) and converts to f32 with
 * the same arithmetic (tests/golden/wav_samples.npz holds the reference's decoded tables).
 *
 * The recorder keeps the stems in HBM: appending is a device-to-device copy on the render stream, the
 * min/max scan and the float -> int16 conversion of save_wav run as HIP kernels; the host only writes
 * the 44-byte header and the converted bytes.  There is no host fallback.
 */
#ifndef SKRED_WAV_H
#define SKRED_WAV_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* what mw_get reports through wav_t (miniwav.h:5-20: Channels, SamplesRate), plus the sample format */
typedef struct {
  uint16_t channels;
  uint32_t sample_rate;
  uint16_t bits_per_sample;
  uint16_t format_tag;     /* 1 PCM, 3 IEEE float (after WAVE_FORMAT_EXTENSIBLE translation) */
  uint32_t frames;
} skred_wav_info_t;

/*
 * One channel of a WAV file as a malloc'ed float table of *frames_out samples; NULL (and
 * *frames_out = 0) when the file cannot be read or decoded (miniwav.c:113-117,128-131).
 *
 * Channel selection follows miniwav.c:130-139 as compiled: `ch` is compared with the unsigned
 * channel count, so any ch < 0 (the parser's default is -1, wire.c:804) or ch > channels selects
 * index `channels`, i.e. channel 0 of the NEXT frame: table[j] = frame[j+1].ch0.  The reference
 * then reads its last element past the decoded buffer; this build defines that element as 0.0f.
 * 0 <= ch < channels selects that channel.
 */
float *skred_wav_get(const char *filename, int *frames_out, skred_wav_info_t *info, int ch);
float *skred_wav_get_mem(const void *bytes, size_t n_bytes, int *frames_out, skred_wav_info_t *info, int ch);
float *skred_wav_free(float *f);   /* frees, returns NULL (miniwav.c:94-97) */

/* ---------------------------------------------------------------- stem recorder (device resident) */

typedef struct skred_recorder skred_recorder_t;

/* skred.c:91-99 synth_callback_init(max_sec): room for `capacity_frames` frames of n_voices stereo stems
 * (float[frames][n_voices][2], the layout synth() writes through `user`, synth.c:607-611) on `device`. */
int  skred_recorder_create(skred_recorder_t **out, int device, int n_voices, long capacity_frames);
void skred_recorder_destroy(skred_recorder_t *r);      /* skred.c:101-105 */

/* `<sec` (wire.c:816-830): rec_ptr = 0, rec_state = 1; max_frames > 0 lowers the limit (clamped to capacity) */
int  skred_recorder_start(skred_recorder_t *r, long max_frames);
void skred_recorder_stop(skred_recorder_t *r);         /* rec_state = 0 */
int  skred_recorder_recording(const skred_recorder_t *r);   /* rec_state */
long skred_recorder_frames(const skred_recorder_t *r);      /* rec_ptr / VOICE_MAX / AUDIO_CHANNELS */

/* skred.c:120-131: append one callback's stems while recording; stops (rec_state = 0) when the limit is
 * reached, keeping the frames that fit.  d_stems is a DEVICE pointer, `stream` a hipStream_t (0 = default);
 * the copy is ordered after the render that produced the stems when both use the same stream. */
int  skred_recorder_append(skred_recorder_t *r, const float *d_stems, int frames, void *stream);

/*
 * wire.c:94-185 save_wav: 16-bit PCM, two channels per voice with record[v] != 0, all recorded frames.
 * As in the reference the scale factor comes from the min and max over ALL voices' samples in the
 * recorded range (recorded or not): scale = |min| > |max| ? -1/min : 1/max; each sample is
 * clamp(g*scale, -1, 1) * 32767 truncated toward zero.  Nothing is written (return SKRED_OK) when no
 * voice is selected or nothing was recorded (wire.c:106-109,833).  `sample_rate` goes into the header
 * (the reference hard-codes 44100, wire.c:117).
 */
int  skred_recorder_save_wav(skred_recorder_t *r, const char *filename, const int *record, int sample_rate);

/* the same conversion into a caller buffer of frames * 2 * n_selected int16 (no file); returns the number
 * of int16 written or a negative SKRED_E_* code */
long skred_recorder_convert(skred_recorder_t *r, const int *record, int16_t *out, long out_capacity);

#ifdef __cplusplus
}
#endif
#endif
