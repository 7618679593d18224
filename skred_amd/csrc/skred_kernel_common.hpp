// skred_kernel_common.hpp -- device helpers shared by the render kernels (gfx950 / CDNA4).
//
// Included by skred_render_generic.hip, skred_render_fast.hip and skred_render_fast2.hip; each is its
// own translation unit (no relocatable device code), so everything here is __forceinline__.
//
// Arithmetic contract (must match oracle/cpu_ref.c bit for bit per voice): compiled with
// -ffp-contract=off (no FMA fusion), fp32 subnormals kept (hipcc default), IEEE-rounded
// divide (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt), exact fmod.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "skred_device_layout.h"

#define LCG_A 6364136223846793005ULL
#define LCG_C 1442695040888963407ULL

// ---------------------------------------------------------------- wave reduction (fold L/R, then DPP)
//
// Per frame every lane holds its voice's (L, R).  v_permlane32_swap exchanges the upper half of one register with
// the lower half of another, so ONE swap + ONE add leave, in a single register, L pair sums (lane i + lane i+32) in
// lanes 0..31 and R pair sums in lanes 32..63.  From there one 5-step DPP butterfly over 32 lanes
//   quad_perm [1,0,3,2]; quad_perm [2,3,0,1]; row_half_mirror; row_mirror (every lane now holds its 16-lane row
//   sum); row_bcast:15 into rows 1 and 3
// sums both channels at once: the frame's L total ends in lane 31, its R total in lane 63 -- 7 VALU instructions per
// frame where two separate 6-step butterflies took 12.  Fixed association order: bit-reproducible run to run.
//
// The butterflies are asm: every DPP read must sit >= 2 wait states behind the VALU write of its source and hipcc pads
// nothing inside an asm statement, so single chains are spaced with s_nop 1 and the two-chain form interleaves
// (s_nop 0 between stages).  Rows masked off by row_mask keep their value.
__device__ __forceinline__ float fold_lr(float l, float r) {
  const auto p = __builtin_amdgcn_permlane32_swap(__float_as_uint(l), __float_as_uint(r), false, false);
  return __uint_as_float(p[0]) + __uint_as_float(p[1]);      // lanes 0..31: l[i] + l[i+32]; lanes 32..63: r[i-32] + r[i]
}

// x: folded value of one frame; afterwards lane 31 holds the L total, lane 63 the R total
__device__ __forceinline__ void half_sum_to_lanes_31_63(float &x) {
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(x));
}

// x + (x of the lane 8 further in the same 16-lane row, cyclically): one DPP add
__device__ __forceinline__ float row_ror8_add(float x) {
  float y;
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1" : "=v"(y) : "v"(x));
  return y;
}

// rows 0+1 and rows 2+3 (lane by lane): every lane of rows 0,1 ends with x[row 0] + x[row 1], of rows 2,3 with x[row 2] + x[row 3]
__device__ __forceinline__ float row_pair_add(float x) {
  const auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(p[0]) + __uint_as_float(p[1]);
}

// two frames at once (two independent chains)
__device__ __forceinline__ void half_sum2_to_lanes_31_63(float &x0, float &x1) {
#define SK_DPP2(CTRL)                                                \
  "v_add_f32_dpp %0, %0, %0 " CTRL " bank_mask:0xf\n\t"              \
  "v_add_f32_dpp %1, %1, %1 " CTRL " bank_mask:0xf\n\t"              \
  "s_nop 0\n\t"
  asm volatile("s_nop 1\n\t"
               SK_DPP2("quad_perm:[1,0,3,2] row_mask:0xf")
               SK_DPP2("quad_perm:[2,3,0,1] row_mask:0xf")
               SK_DPP2("row_half_mirror row_mask:0xf")
               SK_DPP2("row_mirror row_mask:0xf")
               SK_DPP2("row_bcast:15 row_mask:0xa")
               "s_nop 0"
               : "+v"(x0), "+v"(x1));
#undef SK_DPP2
}

// ---------------------------------------------------------------- the block's mix-down, in the render kernel
//
// Every workgroup of a render kernel owns one row of a.partial ([n_rows][F][2], pre-master).  Instead of reduction
// kernels behind the render, the workgroup that ARRIVES LAST does the adding -- no workgroup ever waits for another,
// so nothing here can spin or hang.  Many rows go in two levels: rows w = s (mod 32) form slab s, the last arriver
// of a slab adds its rows (ascending w) into slab_rows[s]; the last slab to finish adds the slab rows (ascending s),
// writes the pre-master sum and, when asked (single-GPU form), multiplies by the master gain of each frame.  The
// order of the additions is fixed by the indices, not by who arrives when: the output is bit-reproducible.
// The master gain (synth.c:616-620: vg += k * (target - vg) per frame, a serial float recurrence) is walked by one
// extra workgroup (blockIdx.x == 0 when a.wg_shift) while the others render; it arrives at the last ticket too.
//
// Visibility follows the guide's write-through recipe (cdna_hip_programming.md, Guideline 16, R1): the bytes another
// workgroup will read are stored WRITE-THROUGH (sc1: 8-byte agent-scope relaxed atomic stores), every storing wave
// drains them (s_waitcnt vmcnt(0)), the workgroup meets at a barrier and ONE lane adds to the ticket -- no release
// fence, which would write back the XCD's whole L2 (full of voice-state lines here) once per workgroup.  The last
// arriver acquires at agent scope (its CU's L1 lines dropped), drains, barrier, then plain loads.
// The workgroup's copy of the table pool into LDS (n16 16-byte words, NTHREADS threads): every thread keeps up to eight loads in
// flight before it stores the first.  The plain loop `dst[i] = src[i]` compiles to load, wait, store per trip -- seven round trips
// to L2 for the C2 recipe's 25 KB pool, ~5 us of a block that takes 44 (round 4: tools/ab.py frames, the fixed part of a block).
template <int NTHREADS>
__device__ __forceinline__ void sk_stage_tables(const void *__restrict__ src_, void *dst_, int n16, int tid) {
  const uint4 *__restrict__ src = reinterpret_cast<const uint4 *>(src_);
  uint4 *dst = reinterpret_cast<uint4 *>(dst_);
  for (int base = 0; base < n16; base += 8 * NTHREADS) {
    uint4 t[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) t[k] = src[min(base + k * NTHREADS + tid, n16 - 1)];
    // (indices clamped, nothing predicated: behind the pool's end a thread copies the last word again -- the same value to the same
    // place --; a store under `if (i < n16)` makes hipcc sink the load into the branch, and the round trips are back)
#pragma unroll
    for (int k = 0; k < 8; ++k) dst[min(base + k * NTHREADS + tid, n16 - 1)] = t[k];
  }
}

typedef __attribute__((address_space(1))) unsigned long long sk_gu64;
typedef __attribute__((address_space(1))) unsigned int sk_gu32;
__device__ __forceinline__ void sk_store_through(float2 *p, float2 v) {
  const unsigned long long bits = ((unsigned long long)__float_as_uint(v.y) << 32) | (unsigned long long)__float_as_uint(v.x);
  __hip_atomic_store((sk_gu64 *)p, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sk_store_through(float *p, float v) {
  __hip_atomic_store((sk_gu32 *)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Every wave of the workgroup calls this after its write-through stores; true in the workgroup that arrived last.
__device__ __forceinline__ bool sk_arrive_last(uint32_t *ticket, uint32_t expected, int tid, int *flag_lds) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's published bytes have left the CU
  __syncthreads();                                            // (also: every wave is done with the LDS word used below)
  if (tid == 0) {
    const uint32_t before = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = before + 1u == expected;
    if (last) {
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed for the next launch (stream-ordered)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    *flag_lds = last ? 1 : 0;
  }
  __syncthreads();
  return *flag_lds != 0;
}

// Packed lanes (sk_render_args_t: pack_mask): the voice of `lane` of packed wave `wave_g`, -1 when the lane is empty.
// `mask`: the lane's group word; `pos`: the voice's lane in its own 64-voice group (what the MODI plane's modulator lanes
// are numbered in).  Once per pass and lane.
__device__ __forceinline__ int sk_packed_voice(const sk_render_args_t &a, int wave_g, int lane, uint64_t &mask, int &pos) {
  const int sh = a.pack_shift;
  const int grp = (wave_g << (6 - sh)) + (lane >> sh);
  const int rank = lane & ((1 << sh) - 1);
  mask = grp < a.pack_groups ? a.pack_mask[grp] : 0ull;
  if (rank == 0 && __popcll(mask) > (1 << sh)) __hip_atomic_fetch_add((sk_gu32 *)a.violations, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the host sized the slots from these very words)
  pos = -1;
  if (rank < __popcll(mask)) {
    uint64_t t = mask;
    for (int i = 0; i < rank; ++i) t &= t - 1;
    pos = __builtin_ctzll(t);
  }
  return pos < 0 ? -1 : grp * 64 + pos;
}
// a modulator's lane in its 64-voice group -> its lane in the packed wave (the host gave every modulator of a voice that can
// sound a lane: a missing one is counted and reads the carrier itself)
__device__ __forceinline__ int sk_packed_lane(const sk_render_args_t &a, uint64_t mask, int lane, int mod_lane) {
  if (!((mask >> mod_lane) & 1)) { __hip_atomic_fetch_add((sk_gu32 *)a.violations, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return lane; }
  return (lane & ~((1 << a.pack_shift) - 1)) + __popcll(mask & (((uint64_t)1 << mod_lane) - 1));
}

// One float of the workgroup's own row, at the end of a chunk: overwritten in the workgroup's first pass over the bank,
// accumulated in later ones (plain accesses: this CU's own lines); in the pass that completes the row (`publish`) the
// value leaves as a write-through store -- that is the copy another workgroup may read (sk_finish_block).
__device__ __forceinline__ void sk_row_store(float *p, float s, bool first_pass, bool publish) {
  const float v = first_pass ? s : *p + s;
  if (publish) sk_store_through(p, v); else *p = v;
}

// Sum of `n` rows (`step` rows apart, starting at row `first`) for one column group, ascending, up to 16 loads in
// flight: the rows sit behind the memory side (write-through stores drop them from the writers' L2), so what this
// costs is round trips, not bytes.  V = float4 (two (L,R) columns) or float2.
template <typename V> __device__ __forceinline__ void sk_acc(V &a, const V &t);
template <> __device__ __forceinline__ void sk_acc<float4>(float4 &a, const float4 &t) { a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w; }
template <> __device__ __forceinline__ void sk_acc<float2>(float2 &a, const float2 &t) { a.x += t.x; a.y += t.y; }
template <typename V>
__device__ __forceinline__ V sk_add_rows(const V *__restrict__ rows, size_t ncolv, int c, int first, int n, int step) {
  V acc = rows[(size_t)first * ncolv + c];
  int i = 1;
  for (; i + 16 <= n; i += 16) {
    V t[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) t[k] = rows[(size_t)(first + (i + k) * step) * ncolv + c];
#pragma unroll
    for (int k = 0; k < 16; ++k) sk_acc<V>(acc, t[k]);
  }
  if (i < n) {
    V t[15];
#pragma unroll
    for (int k = 0; k < 15; ++k) if (i + k < n) t[k] = rows[(size_t)(first + (i + k) * step) * ncolv + c];
#pragma unroll
    for (int k = 0; k < 15; ++k) if (i + k < n) sk_acc<V>(acc, t[k]);
  }
  return acc;
}

// The block's last step, run by whichever workgroup arrived last at the final ticket: the rows (or the slab sums) added up in
// index order, plus -- when an envelope kernel ran beside this one -- its sum; then the pre-master sum and / or the frames.
template <typename V>
__device__ __forceinline__ void sk_final_cols(const sk_render_args_t &a, int tid, int nthreads) {
  constexpr int PER = sizeof(V) / sizeof(float2);             // (L,R) columns per V
  const size_t ncolv = (size_t)a.num_frames / PER;
  const bool two_level = a.n_rows > SK_FINISH_FLAT_MAX;
  const int n_last = two_level ? SK_FINISH_SLABS : a.n_rows;
  const V *rows = reinterpret_cast<const V *>(two_level ? a.slab_rows : a.partial);
  for (int c = tid; c < (int)ncolv; c += nthreads) {
    V s = sk_add_rows<V>(rows, ncolv, c, 0, n_last, 1);
    if (a.env_beside) sk_acc<V>(s, reinterpret_cast<const V *>(a.env_sum)[c]);
    const float2 *h = reinterpret_cast<const float2 *>(&s);
    if (a.sum_out) reinterpret_cast<V *>(a.sum_out)[c] = s;
    if (a.mix_out) {                                          // synth.c:621-624
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        const size_t f = (size_t)c * PER + k;
        const float vg = a.gains[f];
        a.mix_out[f * a.num_channels + 0] = h[k].x * vg;
        a.mix_out[f * a.num_channels + 1] = h[k].y * vg;
      }
    }
  }
  // what this launch found, straight into the host's pinned words (skred_bank.c: poll_reports)
  if (a.report) {
    uint32_t w0 = 0, w1 = 0;
    if (a.fast_mode & SKM_TWO_PER_LANE) {
      w0 = a.env_beside ? (uint32_t)__hip_atomic_load(a.env_off + a.n_groups * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#ifdef SK_TWO_PER_LANE_TU
      if (a.env_gain) w0 = __hip_atomic_load(a.env_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (the listed voices this kernel's waves counted)
#endif
      w1 = __hip_atomic_load(a.violations, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      // one-voice family: did any workgroup see an envelope move (sk_note_moved: one word per row, written through)
      int any = 0;
      for (int i = tid; i < a.n_rows; i += nthreads)
        any |= __hip_atomic_load(a.moved + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == a.launch_ticket;
      w0 = __syncthreads_or(any) ? 1u : 0u;
    }
    if (tid == 0) {
      const unsigned long long t = (unsigned long long)a.launch_ticket << 32;
      __hip_atomic_store(a.report + 1, t | w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      __hip_atomic_store(a.report, t | w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
#ifdef SK_TWO_PER_LANE_TU
  if (a.env_gain && tid == 0) { a.env_count[0] = 0u; a.env_count[1] = 0u; }   // re-armed for the next block (later kernels on the stream)
#endif
}
// one-voice family, RAMPK instantiation: this workgroup saw an envelope in motion in this launch (one word per row, written
// through: the block's final arriver, possibly on another XCD, collects them for the host's report)
__device__ __forceinline__ void sk_note_moved(const sk_render_args_t &a, int bid) {
  __hip_atomic_store(a.moved + bid, a.launch_ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// arrivals the final ticket waits for: the rows (or slabs), the gain workgroup, the envelope kernel's sum
__device__ __forceinline__ uint32_t sk_final_arrivals(const sk_render_args_t &a) {
  const int n_last = a.n_rows > SK_FINISH_FLAT_MAX ? SK_FINISH_SLABS : a.n_rows;
  return (uint32_t)n_last + (a.wg_shift ? 1u : 0u) + (a.env_beside ? 1u : 0u);
}

// Called by EVERY workgroup of the block's steady (or only) render kernel, after its last store into a.partial (renderers:
// bid = their row; the gain workgroup: bid < 0).  `published`: the row's final values already left as write-through
// stores (sk_row_store in the completing pass); otherwise it is copied in place that way here.
// `flag_lds`: one LDS word the workgroup no longer needs.
template <typename V>
__device__ __forceinline__ void sk_finish_cols(const sk_render_args_t &a, int bid, int tid, int nthreads, int *flag_lds, bool published) {
  constexpr int PER = sizeof(V) / sizeof(float2);             // (L,R) columns per V
  const size_t ncolv = (size_t)a.num_frames / PER;
  const bool two_level = a.n_rows > SK_FINISH_FLAT_MAX;
  V *rows = reinterpret_cast<V *>(a.partial);
  if (bid < 0) {
    if (tid == 0) {                                           // the master gain of every frame, serially as the reference does
      float vg = a.gain_state[0];
      for (int i = 0; i < a.num_frames; ++i) {
        vg += a.vol_k * (a.vol_target - vg);
        sk_store_through(&a.gains[i], vg);
      }
      a.gain_commit[0] = vg;                                  // (read by a later launch only: the next block's, or the master kernel's)
    }
  } else {
    if (!published) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      float2 *mine = reinterpret_cast<float2 *>(a.partial) + (size_t)bid * a.num_frames;
      for (int c = tid; c < a.num_frames; c += nthreads) sk_store_through(&mine[c], mine[c]);
    }
    if (two_level) {
      const int slab = bid % SK_FINISH_SLABS;
      const int members = (a.n_rows - slab + SK_FINISH_SLABS - 1) / SK_FINISH_SLABS;
      if (!sk_arrive_last(a.tickets + slab, (uint32_t)members, tid, flag_lds)) return;
      float2 *dst = reinterpret_cast<float2 *>(a.slab_rows) + (size_t)slab * a.num_frames;
      for (int c = tid; c < (int)ncolv; c += nthreads) {
        const V s = sk_add_rows<V>(rows, ncolv, c, slab, members, SK_FINISH_SLABS);
        const float2 *h = reinterpret_cast<const float2 *>(&s);
#pragma unroll
        for (int k = 0; k < PER; ++k) sk_store_through(&dst[(size_t)c * PER + k], h[k]);
      }
    }
  }
  if (!sk_arrive_last(a.tickets + SK_FINISH_SLABS, sk_final_arrivals(a), tid, flag_lds)) return;
  sk_final_cols<V>(a, tid, nthreads);
}

// two columns per load when the rows (num_frames * 8 bytes each) and sum_out keep 16-byte alignment
__device__ __forceinline__ bool sk_finish_wide(const sk_render_args_t &a) {
  return (a.num_frames & 1) == 0 && ((reinterpret_cast<uintptr_t>(a.sum_out) & 15) == 0);
}

__device__ __forceinline__ void sk_finish_block(const sk_render_args_t &a, int bid, int tid, int nthreads, int *flag_lds,
                                                bool published = false) {
  if (sk_finish_wide(a)) sk_finish_cols<float4>(a, bid, tid, nthreads, flag_lds, published);
  else sk_finish_cols<float2>(a, bid, tid, nthreads, flag_lds, published);
}

// The same for the envelope kernel that runs BESIDE the steady kernel (its own stream): every one of its n_env_rows
// workgroups arrives at the kernel's own ticket -- the first `n_used` of them rendered something and have published a row
// of a.env_rows --, the last arriver adds those rows into a.env_sum (zeros when the list was empty) and then arrives at the
// block's final ticket like one more slab.  Whichever workgroup of either kernel is last there finishes the block.
template <typename V>
__device__ __forceinline__ void sk_finish_env_cols(const sk_render_args_t &a, int n_used, int tid, int nthreads, int *flag_lds) {
  constexpr int PER = sizeof(V) / sizeof(float2);
  const size_t ncolv = (size_t)a.num_frames / PER;
  if (!sk_arrive_last(a.env_ticket, (uint32_t)a.n_env_rows, tid, flag_lds)) return;
  float2 *dst = reinterpret_cast<float2 *>(a.env_sum);
  for (int c = tid; c < (int)ncolv; c += nthreads) {
    V s;
    if (n_used > 0) s = sk_add_rows<V>(reinterpret_cast<const V *>(a.env_rows), ncolv, c, 0, n_used, 1);
    else { float2 *z = reinterpret_cast<float2 *>(&s); for (int k = 0; k < PER; ++k) z[k] = make_float2(0.0f, 0.0f); }
    const float2 *h = reinterpret_cast<const float2 *>(&s);
#pragma unroll
    for (int k = 0; k < PER; ++k) sk_store_through(&dst[(size_t)c * PER + k], h[k]);
  }
  if (!sk_arrive_last(a.tickets + SK_FINISH_SLABS, sk_final_arrivals(a), tid, flag_lds)) return;
  sk_final_cols<V>(a, tid, nthreads);
}
__device__ __forceinline__ void sk_finish_env(const sk_render_args_t &a, int n_used, int tid, int nthreads, int *flag_lds) {
  if (sk_finish_wide(a) && (reinterpret_cast<uintptr_t>(a.env_sum) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.env_rows) & 15) == 0)
    sk_finish_env_cols<float4>(a, n_used, tid, nthreads, flag_lds);
  else sk_finish_env_cols<float2>(a, n_used, tid, nthreads, flag_lds);
}

// ---------------------------------------------------------------- per-frame probes (translation units built with -DSK_PROBE_TU)
#ifdef SK_PROBE_TU
// where voice v's (L, R) of the block's first frame go (or nullptr: not probed, or `off`: a skipped / muted voice, whose rows
// keep the zeros the host put there, as the reference's stems do, synth.c:533-534,609-611); rows are a.n_probe float2 apart
__device__ __forceinline__ float2 *sk_probe_row(const sk_render_args_t &a, int v, bool off) {
  int slot = -1;
  for (int i = 0; i < a.n_probe; ++i) slot = a.probe_ids[i] == v ? i : slot;
  return (slot < 0 || off) ? nullptr : reinterpret_cast<float2 *>(a.probe_out) + slot;
}
#endif

// ---------------------------------------------------------------- small exact helpers

// fmodf for x >= 0, y > 0, exact.  x - y is exact for y <= x < 2y (Sterbenz), which is the
// case whenever the phase increment is below one loop length.
__device__ __forceinline__ float fmod_pos(float x, float y) {
  if (x < y) return x;
  if (x < y + y) return x - y;
  return fmodf(x, y);
}

// == quantize_bits_int, synth.c:341-345 (the +0.5 is a double add there)
__device__ __forceinline__ float crush(float v, int bits) {
  const int levels = (1 << bits) - 1;
  const int q = (int)((double)(v * (float)levels) + 0.5);
  return (float)q * (1.0f / (float)levels);
}

// ---------------------------------------------------------------- is a voice's gain at rest?
//
// Which branch of amp_envelope_step (synth.c:398-431) a voice takes at envelope clocks (t, tr):
// 0 inactive, 1 attack, 2 decay, 3 sustain (held), 4 release, 5 release finished (is_active -> 0).
// With both clocks behind the bank's the code is monotone in time (note-off only arrives between launches), so a voice
// whose code is the same on the first and the last frame of a span keeps it for the whole span, and codes 0 / 3 / 5
// (constant level) are absorbing.
__device__ __forceinline__ int sk_env_stage_code(bool active, bool released, float t, float tr, float att, float attdec, float rel) {
  if (!active) return 0;
  if (t < att) return 1;
  if (t < attdec) return 2;
  if (!released) return 3;
  return (tr < rel) ? 4 : 5;
}

// The ONE definition of "this voice may be in motion" that the motion list of the two-per-lane family is built from
// (sk_classify_kernel after uploads and clock changes, sk_render_env2_kernel when it decides which of its voices stay on
// the list, sk_render_fast2_kernel as a cross-check of the voices it was NOT told to sit out).  `first_now`: the clock of
// the first frame that would be rendered (synth.c:521 pre-increments: count + 1).
//   moving    the envelope is in attack / decay / release on that frame, or the note-on lies AHEAD of the clock: until the
//             clock reaches sample_start the reference reads the wrapped difference as a huge elapsed time (synth.c:401), i.e.
//             "sustain", and then starts the attack by itself -- the one way a constant level ends without a control action
//   settling  constant level, but the one-pole amp smoother (synth.c:588-593) still moves: g + k*(gain - g) != g
//   neither   at rest: constant level for every later block until a control action reaches the voice (stages 0 / 3 / 5 are
//             absorbing, and every control action puts the voice back on the list)
struct sk_motion_t {
  int code;          // stage on the first frame
  bool moving, settling;
  float gain_const;  // amp * (level * velocity) of a constant stage (synth.c:582,588); 0-level for the moving ones
};
__device__ __forceinline__ sk_motion_t sk_env_motion(uint64_t first_now, bool dead, bool active, uint64_t t_start, uint64_t t_release,
                                                     float att, float attdec, float rel, float sus, float amp, float vel, float k, float sgain) {
  sk_motion_t m;
  const bool released = t_release != 0;                        // synth.c:417
  const uint64_t d_on = first_now - t_start, d_off = first_now - t_release;
  m.code = sk_env_stage_code(active, released, (float)d_on, (float)d_off, att, attdec, rel);
  const bool ahead = active && (int64_t)(t_start - first_now) > 0;
  m.moving = !dead && (ahead || !(m.code == 0 || m.code == 3 || m.code == 5));
  const float level = m.code == 3 ? sus : 0.0f;
  m.gain_const = amp * (level * vel);
  const float nxt = sgain + k * (m.gain_const - sgain);
  m.settling = !dead && !m.moving && __float_as_uint(nxt) != __float_as_uint(sgain);
  return m;
}

// ---------------------------------------------------------------- per-voice registers

struct VoiceRegs {
  // read-only
  float inc, lo, hi, amp;
  int toff, tsize;
  uint32_t flags;
  int quant, hold_max;
  float att, dec, sus, rel;
  uint64_t t_start, t_release;
  float vel, smooth_k, b0, b1, b2, a1, a2;
  // read-write
  float phase, sgain, x1, x2, y1, y2, sample, hold, pan_l, pan_r;
  int hold_count;
  uint32_t rw;
};

__device__ __forceinline__ void load_voice(const sk_render_args_t &a, int v, VoiceRegs &r) {
  const uint4 osc = *reinterpret_cast<const uint4 *>(&a.ro[SKP_OSC][v]);
  const uint4 tab = *reinterpret_cast<const uint4 *>(&a.ro[SKP_TAB][v]);
  const uint4 et = *reinterpret_cast<const uint4 *>(&a.ro[SKP_ENV_T][v]);
  const uint4 es = *reinterpret_cast<const uint4 *>(&a.ro[SKP_ENV_S][v]);
  const uint4 gn = *reinterpret_cast<const uint4 *>(&a.ro[SKP_GAIN][v]);
  const uint4 fl = *reinterpret_cast<const uint4 *>(&a.ro[SKP_FILT][v]);
  const uint4 s0 = *reinterpret_cast<const uint4 *>(&a.rw[SKS_OSC][v]);
  const uint4 s1 = *reinterpret_cast<const uint4 *>(&a.rw[SKS_FILT][v]);
  const uint4 s2 = *reinterpret_cast<const uint4 *>(&a.rw[SKS_MISC][v]);
  r.inc = __uint_as_float(osc.x); r.lo = __uint_as_float(osc.y);
  r.hi = __uint_as_float(osc.z);  r.amp = __uint_as_float(osc.w);
  r.toff = (int)tab.x; r.tsize = (int)tab.y; r.flags = tab.z;
  r.quant = (int)(tab.w & 0xFFu); r.hold_max = (int)(tab.w >> 8);
  r.att = __uint_as_float(et.x); r.dec = __uint_as_float(et.y);
  r.sus = __uint_as_float(et.z); r.rel = __uint_as_float(et.w);
  r.t_start = ((uint64_t)es.y << 32) | es.x;
  r.t_release = ((uint64_t)es.w << 32) | es.z;
  r.vel = __uint_as_float(gn.x); r.smooth_k = __uint_as_float(gn.y);
  r.b0 = __uint_as_float(gn.z);  r.b1 = __uint_as_float(gn.w);
  r.b2 = __uint_as_float(fl.x);  r.a1 = __uint_as_float(fl.y); r.a2 = __uint_as_float(fl.z);
  r.phase = __uint_as_float(s0.x); r.sgain = __uint_as_float(s0.y);
  r.x1 = __uint_as_float(s0.z);    r.x2 = __uint_as_float(s0.w);
  r.y1 = __uint_as_float(s1.x);    r.y2 = __uint_as_float(s1.y);
  r.sample = __uint_as_float(s1.z); r.rw = s1.w;
  r.hold = __uint_as_float(s2.x);  r.hold_count = (int)s2.y;
  r.pan_l = __uint_as_float(s2.z); r.pan_r = __uint_as_float(s2.w);
  if (r.flags & SKF_REVERSE) r.inc = -r.inc;   // synth.c:224
}

__device__ __forceinline__ void store_voice(const sk_render_args_t &a, int v, const VoiceRegs &r) {
  uint4 s0, s1, s2;
  s0.x = __float_as_uint(r.phase); s0.y = __float_as_uint(r.sgain);
  s0.z = __float_as_uint(r.x1);    s0.w = __float_as_uint(r.x2);
  s1.x = __float_as_uint(r.y1);    s1.y = __float_as_uint(r.y2);
  s1.z = __float_as_uint(r.sample); s1.w = r.rw;
  s2.x = __float_as_uint(r.hold);  s2.y = (uint32_t)r.hold_count;
  s2.z = __float_as_uint(r.pan_l); s2.w = __float_as_uint(r.pan_r);
  *reinterpret_cast<uint4 *>(&a.rw[SKS_OSC][v]) = s0;
  *reinterpret_cast<uint4 *>(&a.rw[SKS_FILT][v]) = s1;
  *reinterpret_cast<uint4 *>(&a.rw[SKS_MISC][v]) = s2;
}

// ---------------------------------------------------------------- one voice, one frame

// Table fetch at table-domain position `pos` (>= 0).  truncate == synth.c:261-274;
// linear == oracle/cpu_ref.c:table_fetch (defined by this project, not by the reference).
template <bool TAB_LDS>
__device__ __forceinline__ float table_fetch(const float *lds_tab, const float *__restrict__ glb_tab,
                                             const VoiceRegs &r, float pos, int interp, bool wraps) {
  int idx = (int)pos;
  if (idx >= r.tsize) idx = r.tsize - 1;
  if (idx < 0) idx = 0;
  const float *tab = TAB_LDS ? lds_tab : glb_tab;
  const float a = tab[r.toff + idx];
  if (interp != 1) return a;
  int nxt = idx + 1;
  if (wraps && (float)nxt >= r.hi) nxt = (int)r.lo;
  if (nxt >= r.tsize) nxt = r.tsize - 1;
  if (nxt < 0) nxt = 0;
  const float frac = pos - (float)idx;
  return a + frac * (tab[r.toff + nxt] - a);
}

// Everything synth.c:531-612 does for one voice in one frame.  Returns this voice's L/R
// contribution to the mix (0,0 when skipped or muted).
template <bool TAB_LDS>
__device__ __forceinline__ void voice_frame(VoiceRegs &r, const float *lds_tab,
                                            const float *__restrict__ glb_tab, uint64_t now,
                                            float white, int interp, float &out_l, float &out_r) {
  out_l = 0.0f; out_r = 0.0f;
  // skip tests, synth.c:531-542: finished or silent voices keep their state frozen
  if ((r.rw & SKR_FINISHED) || r.amp == 0.0f || (r.flags & SKF_INERT)) {
    r.sample = 0.0f;
    return;
  }
  float raw;
  if (r.flags & SKF_NOISE) {
    raw = white;                                            // synth.c:543-546
  } else {
    // osc_next, synth.c:217-275
    float ph = r.phase + r.inc;
    if (!__builtin_isfinite(ph)) {
      r.phase = 0.0f;
      if (r.flags & SKF_ONE_SHOT) r.rw |= SKR_FINISHED;
      raw = 0.0f;
    } else {
      const bool stops = (r.flags & SKF_ONE_SHOT) && !(r.flags & SKF_LOOPING);
      const float span = r.hi - r.lo;
      if (ph >= r.hi) {
        if (stops) { ph = r.hi - 1e-6f; r.rw |= SKR_FINISHED; }
        else ph = r.lo + fmod_pos(ph - r.lo, span);
      } else if (ph < r.lo) {
        if (stops) { ph = r.lo; r.rw |= SKR_FINISHED; }
        else ph = r.hi - fmod_pos(r.lo - ph, span);
      }
      r.phase = ph;
      raw = table_fetch<TAB_LDS>(lds_tab, glb_tab, r, ph, interp, !stops);
    }
  }
  // sample & hold, synth.c:560-571
  if (r.hold_max) {
    if (r.hold_count == 0) r.hold = raw;
    raw = r.hold;
    if (++r.hold_count >= r.hold_max) r.hold_count = 0;
  }
  float s = raw;
  if (r.quant) s = crush(s, r.quant);                        // synth.c:574
  if (r.flags & SKF_FILTER) {                                // mmf_process, synth.c:349-364
    float y = r.b0 * s;
    y = y + r.b1 * r.x1;
    y = y + r.b2 * r.x2;
    y = y - r.a1 * r.y1;
    y = y - r.a2 * r.y2;
    r.x2 = r.x1; r.x1 = s;
    r.y2 = r.y1; r.y1 = y;
    s = y;
  }
  // amp_envelope_step, synth.c:398-431
  float env = 1.0f;
  if (r.flags & SKF_USE_ENV) {
    float e = 0.0f;
    if (r.rw & SKR_ENV_ACTIVE) {
      const float t = (float)(now - r.t_start);
      if (t < r.att) {
        e = t / r.att;
      } else if (t < r.att + r.dec) {
        const float prog = (t - r.att) / r.dec;
        e = 1.0f - prog * (1.0f - r.sus);
      } else if (r.t_release == 0) {
        e = r.sus;
      } else {
        const float tr = (float)(now - r.t_release);
        if (tr < r.rel) {
          const float prog = tr / r.rel;
          e = r.sus * (1.0f - prog);
        } else {
          r.rw &= ~SKR_ENV_ACTIVE;
        }
      }
    }
    env = e * r.vel;
  }
  // amp, smoother, apply: synth.c:580-593 (no amplitude modulator in this kernel: mod == 1)
  float gain = r.amp * env;
  if (r.flags & SKF_SMOOTH) {
    r.sgain += r.smooth_k * (gain - r.sgain);
    gain = r.sgain;
  }
  s *= gain;
  r.sample = s;
  // pan + mix, synth.c:595-612
  if (!(r.flags & SKF_MUTED)) {
    out_l = s * r.pan_l;
    out_r = s * r.pan_r;
  }
}

// ---------------------------------------------------------------- shared by the fast kernels

// two adjacent table samples, fetched with one 4-byte-aligned 8-byte access (global_load_dwordx2 /
// ds_read2_b32).  The pool is padded by the host so that reading one float past any table is in bounds.
struct __attribute__((packed, aligned(4))) tap_pair_t { float a, b; };
// The same access for HBM/L2-resident pools, kept as ONE <2 x float> load of alignment 4 so that it is selected
// as a single global_load_dwordx2 (multi-dword global loads only need dword alignment).  The struct form above is
// split into two global_load_dword by the optimiser, and the texture-address unit processes a scattered wave
// gather at about one lane-dword per cycle, so two loads cost twice one (C4: TA busy 83 %).
typedef float tap_pair_v __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ tap_pair_t load_tap_pair_global(const char *__restrict__ p) {
  const tap_pair_v v = *reinterpret_cast<const tap_pair_v *>(p);
  tap_pair_t r;
  r.a = v.x;
  r.b = v.y;
  return r;
}

// exact wrap for the cases the straight-line code does not cover (synth.c:241-256, looping voice)
__device__ __forceinline__ float slow_wrap(float ph, float lo, float hi, float span) {
  if (!__builtin_isfinite(ph)) return 0.0f;     // unreachable for a fast bank; kept total
  if (ph >= hi) return lo + fmod_pos(ph - lo, span);
  if (ph < lo) return hi - fmod_pos(lo - ph, span);
  return ph;
}

// One frame / two frames of a wave -> wsum[wave][J] (float2 per frame: lane 31 stores .x, lane 63 stores .y).
// (timing experiments only: -DSK_ABLATE_REDUCE drops the cross-lane sum; outputs are then wrong)
#ifdef SK_ABLATE_REDUCE
#define SK_REDUCE_AND_STORE(J) asm volatile("" ::"v"(l), "v"(rr));
#define SK_REDUCE4_AND_STORE(J) asm volatile("" ::"v"(l0), "v"(r0), "v"(l1), "v"(r1));
#else
#define SK_REDUCE_AND_STORE(J)                                                                    \
  {                                                                                               \
    float x_ = fold_lr(l, rr);                                                                    \
    half_sum_to_lanes_31_63(x_);                                                                  \
    if ((lane & 31) == 31) reinterpret_cast<float *>(&wsum[wave * SK_CHUNK + (J)])[lane >> 5] = x_; \
  }
#define SK_REDUCE4_AND_STORE(J)                                                                   \
  {                                                                                               \
    float x0_ = fold_lr(l0, r0), x1_ = fold_lr(l1, r1);                                           \
    half_sum2_to_lanes_31_63(x0_, x1_);                                                           \
    if ((lane & 31) == 31) {                                                                      \
      float *w_ = reinterpret_cast<float *>(&wsum[wave * SK_CHUNK + (J)]) + (lane >> 5);          \
      w_[0] = x0_;                                                                                \
      w_[2] = x1_;                                                                                \
    }                                                                                             \
  }
#endif
