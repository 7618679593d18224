/*
 * skred_launch.h -- C-linkage launch entry points of the HIP translation units (internal to
 * libskred_amd.so; the public ABI is include/skred_amd.h).
 *
 *   skred_render_generic.hip  sk_launch_render (dispatcher), sk_launch_render_mod
 *   skred_render_fast.hip     sk_launch_render_fast
 *   skred_render_split.hip    sk_launch_render_split, sk_split_lds_bytes
 *   skred_render_fast2.hip    sk_launch_render_fast2, sk_launch_env_fast2, sk_env2_grid, sk_launch_classify
 *   skred_gain_kernels.hip    sk_launch_gain
 *   skred_mix_kernels.hip     sk_launch_master, sk_launch_master_apply
 *   skred_update_kernels.hip  sk_launch_update, sk_launch_stamp, sk_launch_pack_zero
 *   skred_rec_kernels.hip     sk_launch_rec_minmax, sk_rec_partial_floats, sk_launch_rec_convert
 *
 * Every launcher returns the hipError_t of the launch as an int.
 */
#ifndef SKRED_LAUNCH_H
#define SKRED_LAUNCH_H

#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1   /* C hosts (gcc): hipcc defines it itself */
#endif
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

#include "skred_device_layout.h"

#ifdef __cplusplus
extern "C" {
#endif

/* picks the kernel family from args->fast_mode / args->stems and launches it on `stream` */
int sk_launch_render(const sk_render_args_t *args, int n_workgroups, hipStream_t stream);
/* banks with modulators: one aligned 64-voice group per wavefront, dependency levels in `levels` */
int sk_launch_render_mod(const sk_render_args_t *args, int n_workgroups, const int *levels, int max_level,
                         hipStream_t stream);
/* the two specialised families (called by sk_launch_render only) */
int sk_launch_render_fast(const sk_render_args_t *args, int n_workgroups, size_t lds_bytes, hipStream_t stream);
int sk_launch_render_fast2(const sk_render_args_t *args, int n_workgroups, size_t lds_bytes, hipStream_t stream);
/* args->fast_mode & SKM_SPLIT: the one-voice family with the frame split between an oscillator wave and a post wave (512-thread
 * workgroups, 256 voices per pass as sk_launch_render_fast); sk_split_lds_bytes: the LDS one of its workgroups takes */
int sk_launch_render_split(const sk_render_args_t *args, int n_workgroups, int pairs, hipStream_t stream);
size_t sk_split_lds_bytes(const sk_render_args_t *args, int pairs);
/* the two-per-lane family's motion list (skred_render_fast2.hip): the list collected from args->mask_cur and rendered by
 * sk_render_env2_kernel on `stream` -- the block's SECOND stream, beside sk_launch_render_fast2 -- with args->n_env_rows
 * workgroups (sk_env2_grid: what the device holds at once); sk_launch_classify rebuilds `mask` from the planes */
int sk_launch_collect(const sk_render_args_t *args, hipStream_t stream);      /* mask_cur -> group_flag (counts), env_off, env_list; mask_next zeroed */
int sk_launch_env_fast2(const sk_render_args_t *args, hipStream_t stream);
int sk_env2_grid(const sk_render_args_t *args);
int sk_launch_classify(const sk_render_args_t *args, uint64_t *mask, hipStream_t stream);
/* sparse lists of LDS-table banks: the listed voices stay in their lanes.  sk_launch_gain (skred_gain_kernels.hip) writes their
 * per-frame gains into args->env_gain and the next block's list into args->mask_next, then sk_launch_render_fast2 -- with
 * args->env_gain set it launches the in-place instantiations -- renders the whole bank; same stream, in this order */
int sk_launch_gain(const sk_render_args_t *args, hipStream_t stream);

int sk_launch_master(const float *sum, float *out, int num_frames, int num_channels, float target, float k,
                     float *gain_state, hipStream_t stream);
/* the same with the block's gains already walked by the render kernel (gains[num_frames], the gain to carry on in gain_pending) */
int sk_launch_master_apply(const float *sum, const float *gains, float *out, int num_frames, int num_channels,
                           const float *gain_pending, float *gain_state, hipStream_t stream);

/* scatter n voice updates into the planes; `now` = synth_sample_count for the STAMP bits; every touched voice goes on the
 * motion list (`mask`: a bit per voice, skred_device_layout.h: mask_cur) */
/* cnt / done / seq: when `done` is not NULL the workgroup that finishes last stores `seq` into *done (pinned host memory the
 * host polls before it reuses the staging buffer the batch was read from); `cnt` is that slot's arrival counter on the device */
int sk_launch_update(const sk_update_t *d_updates, int n, sk_plane_t *const ro[SKP_COUNT], sk_plane_t *const rw[SKS_COUNT],
                     uint64_t now, uint64_t *mask, uint32_t *cnt, uint32_t *done, uint32_t seq, hipStream_t stream);

/* note-on / note-off stamps only: a list of voice ids */
int sk_launch_stamp(const int32_t *d_ids, int n, uint32_t dirty, sk_plane_t *const ro[SKP_COUNT], sk_plane_t *const rw[SKS_COUNT],
                    uint64_t now, uint64_t *mask, uint32_t *cnt, uint32_t *done, uint32_t seq, hipStream_t stream);

/* packed lanes: voice_sample = 0 for every voice without a bit in d_mask (one bit per voice of the padded bank) */
int sk_launch_pack_zero(const uint64_t *d_mask, sk_plane_t *filt, int n_voices_padded, hipStream_t stream);

/* stem recorder (skred_recorder.c): min/max partials of rec[n_floats]; selected voices -> int16 pairs */
int sk_rec_partial_floats(void);
int sk_launch_rec_minmax(const float *rec, size_t n_floats, float *partial, int *n_blocks_out, hipStream_t stream);
int sk_launch_rec_convert(const float *rec, long frames, int n_voices, const int *sel, int n_sel, float scale,
                          int16_t *out, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif
