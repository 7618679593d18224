/*
 * skred_wav.c -- RIFF/WAVE reader for sample tables (include/skred_wav.h).
 *
 * Replaces, for this build's own tools and for bank mode, what the reference gets from
 * mw_get (miniwav.c:103-147) = miniaudio's decoder asked for f32 output.  Sample conversion:
 *   u8   x * (2/255) - 1          s16  x * 2^-15          s24  x * 2^-23
 *   s32  (float)(x / 2^31)        f32  as stored          f64  (float)x
 * (power-of-two scalings are exact, so s16/s24/s32 do not depend on the order of convert and
 * scale; the u8 constant is miniaudio's 0.00784313725490196078f).  tests/test_wav.py compares
 * every table with the one the compiled reference decoded from the same bytes.
 *
 * Host code by nature (file parsing); nothing here renders audio.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "skred_wav.h"

static uint32_t rd32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint16_t rd16(const uint8_t *p) { return (uint16_t)(p[0] | (p[1] << 8)); }

static float sample_at(const uint8_t *d, uint16_t tag, uint16_t bits, size_t i) {
  if (tag == 3) {
    if (bits == 32) { float f; memcpy(&f, d + 4 * i, 4); return f; }
    double g; memcpy(&g, d + 8 * i, 8); return (float)g;
  }
  switch (bits) {
    case 8: { float x = (float)d[i]; x = x * 0.00784313725490196078f; return x - 1; }
    case 16: return (float)(int16_t)rd16(d + 2 * i) * 0.000030517578125f;
    case 24: {
      const uint8_t *p = d + 3 * i;
      const int32_t v = (int32_t)(((uint32_t)p[0] << 8) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 24)) >> 8;
      return (float)v * 0.00000011920928955078125f;
    }
    default: return (float)((double)(int32_t)rd32(d + 4 * i) / 2147483648.0);
  }
}

float *skred_wav_get_mem(const void *bytes, size_t n, int *frames_out, skred_wav_info_t *info, int ch) {
  const uint8_t *b = (const uint8_t *)bytes;
  if (frames_out) *frames_out = 0;
  if (!b || n < 12 || memcmp(b, "RIFF", 4) || memcmp(b + 8, "WAVE", 4)) return NULL;
  uint16_t tag = 0, channels = 0, bits = 0;
  uint32_t rate = 0;
  const uint8_t *data = NULL;
  size_t data_bytes = 0;
  for (size_t pos = 12; pos + 8 <= n;) {
    const uint32_t len = rd32(b + pos + 4);
    const uint8_t *body = b + pos + 8;
    const size_t room = n - (pos + 8);
    if (!memcmp(b + pos, "fmt ", 4)) {
      if (len < 16 || room < 16) return NULL;
      tag = rd16(body);
      channels = rd16(body + 2);
      rate = rd32(body + 4);
      bits = rd16(body + 14);
      if (tag == 0xFFFE) {                       /* WAVE_FORMAT_EXTENSIBLE: the sub-format GUID starts with the tag */
        if (len < 40 || room < 40) return NULL;
        tag = rd16(body + 24);
      }
    } else if (!memcmp(b + pos, "data", 4)) {
      data = body;
      data_bytes = len < room ? len : room;      /* a truncated file plays what is there */
      break;
    }
    pos += 8 + (size_t)len + (len & 1);          /* chunks are word aligned */
  }
  if (!data || channels == 0) return NULL;
  if (!((tag == 1 && (bits == 8 || bits == 16 || bits == 24 || bits == 32)) || (tag == 3 && (bits == 32 || bits == 64))))
    return NULL;
  const size_t frame_bytes = (size_t)channels * (bits / 8);
  const size_t frames = data_bytes / frame_bytes;
  if (frames == 0 || frames > 0x7FFFFFFF) return NULL;
  float *t = (float *)malloc(frames * sizeof(float));
  if (!t) return NULL;
  /* miniwav.c:130 with ch promoted to unsigned: everything outside [0, channels] lands on `channels` */
  const size_t sel = (ch < 0 || (unsigned)ch > channels) ? channels : (size_t)ch;
  const size_t total = frames * channels;
  for (size_t j = 0; j < frames; ++j) {
    const size_t i = j * channels + sel;
    t[j] = i < total ? sample_at(data, tag, bits, i) : 0.0f;   /* the reference over-reads here */
  }
  if (info) {
    info->channels = channels;
    info->sample_rate = rate;
    info->bits_per_sample = bits;
    info->format_tag = tag;
    info->frames = (uint32_t)frames;
  }
  if (frames_out) *frames_out = (int)frames;
  return t;
}

float *skred_wav_get(const char *filename, int *frames_out, skred_wav_info_t *info, int ch) {
  if (frames_out) *frames_out = 0;
  FILE *f = filename ? fopen(filename, "rb") : NULL;
  if (!f) return NULL;
  float *t = NULL;
  if (fseek(f, 0, SEEK_END) == 0) {
    const long n = ftell(f);
    if (n > 0 && fseek(f, 0, SEEK_SET) == 0) {
      uint8_t *buf = (uint8_t *)malloc((size_t)n);
      if (buf && fread(buf, 1, (size_t)n, f) == (size_t)n) t = skred_wav_get_mem(buf, (size_t)n, frames_out, info, ch);
      free(buf);
    }
  }
  fclose(f);
  return t;
}

float *skred_wav_free(float *f) {
  free(f);
  return NULL;
}
