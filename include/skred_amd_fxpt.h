/*
 * include/skred_amd_fxpt.h -- C ABI of the FIXED-POINT render path.
 *
 * The reference has no fixed-point render path at all (the notamy <name>_lutset_fxpt.h files are
 * data-only headers nobody includes), so this path is DEFINED by this project: the definition is
 * oracle/cpu_ref_fxpt.c (scalar C) and the HIP kernel must reproduce it bit for bit, mix included
 * (integer sums do not depend on the order of addition).  Parity against the reference: unpinned
 * upstream, by construction.
 *
 * Arithmetic (all integer, two's complement, `>>` on signed values is arithmetic):
 *   LUT      int16 single-cycle tables of 2^L entries (the notamy *_fxpt pyramids), one pool
 *   phase    uint32, 2^32 = one table cycle; per frame  phase += phase_inc  (mod 2^32), THEN sampled
 *   index    i = phase >> (32-L);  truncate: s = lut[i]
 *            linear:   f = (phase << L) >> 17  (Q15),  s = a + (((b - a) * f) >> 15),  b = lut[(i+1) & (2^L-1)]
 *   ADSR     t = sat32(now - sample_start), linear stages on integer frame counts A, D, R and Q15
 *            sustain S with reciprocals rX = floor(2^32 / X):
 *              t < A        e = (t * rA) >> 17
 *              t < A + D    e = 32768 - ((((t-A) * rD) >> 17) * (32768 - S) >> 15)
 *              held         e = S                      (sample_release == 0)
 *              tr < R       e = S - ((((tr * rR) >> 17) * S) >> 15),  tr = sat32(now - sample_release)
 *              else         e = 0, is_active = 0
 *   gain     target = (amp_q15 * ((e * velocity_q15) >> 15)) >> 15        (amp_q15 <= 65535; this product in 64 bits)
 *            smoother (optional): g += ((target - g) * k_q15) >> 15, gain = g
 *   output   v = (s * gain) >> 15 (product in 64 bits) ; L = (v * pan_left_q15) >> 15 ; R = (v * pan_right_q15) >> 15
 *            every other product is an int32 one (keep it inside 32 bits: Q15 gains, velocity <= 65535)
 *   mix      int64 sum of L and of R over all voices, per frame
 *   skipped  amp_q15 == 0 or finished: v = 0, state frozen (as synth.c:531-542 does for the float path)
 *   one-shot a voice with one_shot != 0 plays ONE cycle of its table: in the frame where phase + phase_inc carries out of
 *            32 bits the phase is left at 0xFFFFFFFF, that frame is still rendered from the table's last entry (linear:
 *            the neighbour does not fold back to entry 0, b = a), `finished` is set and from the next frame on the voice
 *            is skipped -- the image of osc_next()'s finish rule for forward playback (synth.c:241-256)
 *   biquad   filter_mode != 0: direct form I on the (interpolated) table sample s, |s| <= 32767, in the order of the
 *            float path (synth.c:349-364: after the oscillator, before the gain).  Coefficients Q2.30 in int32 (the RBJ
 *            b/a0, a/a0 of mmf_set_params, synth.c:929-1008: all inside (-2, 2)); delay line in Q12 sample units, int32,
 *            |x|, |y| < 2^29; accumulation in int64 (five products < 2^60 each):
 *              x0  = s << 12
 *              acc = b0*x0 + b1*x1 + b2*x2 - a1*y1 - a2*y2
 *              y0  = clamp((acc + 2^29) >> 30, -2^29, 2^29 - 1)       round to nearest, saturating
 *              x2 = x1, x1 = x0, y2 = y1, y1 = y0
 *              s   = clamp(y0 >> 12, -32768, 32767)                   back to sample units: a resonant overshoot saturates
 *   master   the image of the master-volume stage (synth.c:616-624: vg += k * (target - vg); out = sum * vg), after the mix:
 *            g is Q31 held in int64, target in [0, 2^31), k_q15 in [0, 32768]; per frame
 *              g   += ((target_q31 - g) * k_q15) >> 15
 *              out  = (mix * (g >> 16)) >> 15                         per channel, int64 (|mix| < 2^38 for 2^20 voices: 53 bits)
 *            defaults: target 0.025 (volume_user 1 x AMY_FACTOR), k 66/32768 (0.002), g 0 -- the float path's
 *   stamps   note-on: sample_start = now, sample_release = 0, is_active = 1; note-off: if is_active, sample_release = now
 *            (amp_envelope_trigger / _release, synth.c:383-395), now = the bank's synth_sample_count when the stamp runs
 */
#ifndef SKRED_AMD_FXPT_H
#define SKRED_AMD_FXPT_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct skred_fxpt_bank {
  int32_t n_voices;
  uint32_t *phase;              /* rw */
  uint32_t *phase_inc;
  int32_t  *table_offset;       /* first entry of the voice's table inside the int16 pool */
  int32_t  *log2_size;          /* 3..15 */
  int32_t  *amp_q15;            /* 0..65535 */
  int32_t  *pan_left_q15, *pan_right_q15;
  int32_t  *disconnect;
  int32_t  *use_envelope;
  uint32_t *attack_frames, *decay_frames, *release_frames;
  int32_t  *sustain_q15, *velocity_q15;
  uint64_t *sample_start, *sample_release;
  int32_t  *is_active;          /* rw */
  int32_t  *smoother_enable;
  int32_t  *smoother_k_q15;
  int32_t  *smoother_gain_q15;  /* rw */
  int32_t  *voice_sample;       /* rw: v of the last rendered frame */
  int32_t  *one_shot;           /* plays one cycle, then finishes */
  int32_t  *finished;           /* rw */
  int32_t  *filter_mode;        /* 0 = no filter (the kind of filter is in the coefficients) */
  int32_t  *b0_q30, *b1_q30, *b2_q30, *a1_q30, *a2_q30;
  int32_t  *x1, *x2, *y1, *y2;  /* rw: delay line, Q12 sample units */
} skred_fxpt_bank_t;

typedef struct skred_fxbank skred_fxbank_t;   /* opaque device-side bank */

int  skred_fxbank_create(int device, int n_voices, skred_fxbank_t **out);
void skred_fxbank_destroy(skred_fxbank_t *fx);
int  skred_fxbank_set_tables_i16(skred_fxbank_t *fx, const int16_t *pool, size_t n_entries);
int  skred_fxbank_upload(skred_fxbank_t *fx, const skred_fxpt_bank_t *host, int src_first, int dst_first, int count);
int  skred_fxbank_download(skred_fxbank_t *fx, skred_fxpt_bank_t *host, int src_first, int dst_first, int count);
int  skred_fxbank_set_sample_count(skred_fxbank_t *fx, uint64_t synth_sample_count);
uint64_t skred_fxbank_get_sample_count(const skred_fxbank_t *fx);

/* Render num_frames frames; d_mix = device int64[num_frames][2]; d_stems = device int32
 * [num_frames][n_voices][2] or NULL.  interp: 0 truncate, 1 linear.  Asynchronous on `stream`. */
int  skred_fxbank_render(skred_fxbank_t *fx, int num_frames, int interp, int64_t *d_mix, int32_t *d_stems, void *stream);
/* Render + mix-down + master stage in ONE launch (the render kernel's last-arriving workgroups add the rows up and apply the
 * gain of each frame): d_out = device int64[num_frames][2], post-master. */
int  skred_fxbank_render_mix(skred_fxbank_t *fx, int num_frames, int interp, int64_t *d_out, int32_t *d_stems, void *stream);
/* The master stage alone, for a sum that travelled (multi-GPU: on the root, after the int64 reduce of the ranks'
 * skred_fxbank_render outputs -- an exact sum, whatever the order): must follow a skred_fxbank_render of the same block on
 * this bank, which walked the block's gains.  d_out may equal d_sum. */
int  skred_fxbank_master(skred_fxbank_t *fx, const int64_t *d_sum, int num_frames, int64_t *d_out, void *stream);
int  skred_fxbank_set_master(skred_fxbank_t *fx, int64_t target_q31, int32_t k_q15, int64_t gain_q31);   /* synchronous */
int64_t skred_fxbank_get_master_gain(skred_fxbank_t *fx);                                                  /* synchronous; < 0: error */
/* Note-ons / note-offs on device-resident voices (the float path's SKRED_STAMP_TRIGGER / _RELEASE), on `stream`. */
enum { SKRED_FX_STAMP_TRIGGER = 1, SKRED_FX_STAMP_RELEASE = 2 };
int  skred_fxbank_stamp(skred_fxbank_t *fx, const int32_t *voices, int n_voices, int which, void *stream);
/* Same on host buffers (synchronous). */
int  skred_fxbank_render_host(skred_fxbank_t *fx, int num_frames, int interp, int64_t *mix, int32_t *stems_or_null);
float skred_fxbank_last_render_ms(skred_fxbank_t *fx);

/* ---- the fixed-point bank sharded over the GPUs of one node (include/skred_amd.h: skred_shard_*) ----
 * skred_fxshard_create() makes an skred_shard_t whose steps are this path's: every rank renders its block of the bank into an
 * int64 pre-master sum, the one collective is ncclReduce(ncclSum, ncclInt64) -- an exact sum: the sharded render equals the
 * unsharded one BIT FOR BIT for every number of ranks --, the root applies the master stage.  Drive it with
 * skred_shard_render_mix (partial / out: int64[num_frames][2] behind the float pointers), skred_shard_init_rccl,
 * skred_shard_set_ops, skred_shard_destroy. */
struct skred_shard;
int  skred_fxshard_create(int device, int rank, int world, int root, int total_voices, struct skred_shard **out);
skred_fxbank_t *skred_fxshard_bank(struct skred_shard *shard);
int  skred_fxshard_upload(struct skred_shard *shard, const skred_fxpt_bank_t *whole_bank);

#ifdef __cplusplus
}
#endif
#endif
