/* skred_fx_layout.h -- device layout of a fixed-point voice bank (see skred_device_layout.h for the
 * float path; same idea: 16-byte planes, one coalesced dwordx4 per lane per plane). */
#ifndef SKRED_FX_LAYOUT_H
#define SKRED_FX_LAYOUT_H
#include <stdint.h>

enum {
  SKX_OSC = 0,  /* u32 phase_inc | i32 table_offset | u32 log2_size + (flags << 8) | i32 amp_q15 */
  SKX_GAIN,     /* i32 pan_left_q15 | i32 pan_right_q15 | i32 smoother_k_q15 | i32 velocity_q15   */
  SKX_ENV,      /* u32 attack_frames | u32 decay_frames | u32 release_frames | i32 sustain_q15    */
  SKX_RECIP,    /* u32 floor(2^32/attack) | floor(2^32/decay) | floor(2^32/release) | 0           */
  SKX_TIME,     /* u32 sample_start lo,hi | u32 sample_release lo,hi                               */
  SKX_FILT,     /* i32 b0 | b1 | b2 | a1   (Q2.30)                                                 */
  SKX_FILT2,    /* i32 a2 (Q2.30) | 0 | 0 | 0                                                       */
  SKX_COUNT
};
/* read-write planes: [0] u32 phase | i32 smoother_gain_q15 | i32 voice_sample | u32 bit 0 is_active, bit 1 finished
 *                    [1] i32 x1 | x2 | y1 | y2   (biquad delay line, Q12 sample units) */
#define SKX_RW_COUNT 2

#define SKXF_USE_ENV (1u << 0)
#define SKXF_SMOOTH  (1u << 1)
#define SKXF_MUTED   (1u << 2)
#define SKXF_INERT   (1u << 3)
#define SKXF_FILTER  (1u << 4)
#define SKXF_ONE_SHOT (1u << 5)

#define SKX_GROUP 256
#define SKX_CHUNK 64
#define SKX_LDS_TABLE_MAX_BYTES 49152
#define SKX_MAX_WORKGROUPS 2048

typedef struct { uint32_t w[4]; } skx_plane_t;

typedef struct {
  const skx_plane_t *ro[SKX_COUNT];
  skx_plane_t *rw[SKX_RW_COUNT];
  const int16_t *tables;
  long long *partial;     /* [n_workgroups][num_frames][2] */
  int32_t *stems;         /* [num_frames][n_voices][2] or NULL */
  uint64_t count0;
  int32_t n_voices, n_groups, num_frames, interp;
  int32_t lds_bytes_tables;   /* bytes of the pool staged in LDS (multiple of 16), 0 = gather from L2/HBM */
  int32_t any_filter;         /* some voice of the bank runs the biquad (selects the block instantiation) */
  /* ---- the block's mix-down and master stage inside the render kernel (the float path's scheme, skred_kernel_common.hpp:
   * sk_finish_block, on int64 rows: every workgroup publishes its row write-through, the last arriver of a slab adds the slab,
   * the last slab adds the slabs -- integer sums, any order is exact -- and applies the master gain of each frame, which
   * workgroup 0 of the grid walked meanwhile) ---- */
  int32_t n_rows;             /* rendering workgroups; the grid has one more (blockIdx 0: the gain workgroup) */
  long long *slab_rows;       /* [SKX_FINISH_SLABS][num_frames][2] */
  uint32_t *tickets;          /* [SKX_FINISH_SLABS + 1], zero between launches */
  long long *sum_out;         /* [num_frames][2] pre-master sum (the operand of the multi-GPU reduce), or NULL */
  long long *mix_out;         /* [num_frames][2] post-master output, or NULL */
  int32_t *gains;             /* [num_frames] Q15 master gain per frame */
  long long *gain_state;      /* [0] Q31 gain carried between blocks (read) */
  long long *gain_commit;     /* where the gain after the last frame goes ([0] when this launch applies the stage; [1]: pending for skred_fxbank_master) */
  long long master_target_q31;
  int32_t master_k_q15;
} skx_args_t;

#define SKX_FINISH_SLABS 32
#define SKX_FINISH_FLAT_MAX 64
#endif
