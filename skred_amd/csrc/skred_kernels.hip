// skred_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4) for skred's render loop.
//
// What is computed: the two nested loops of the reference's synth() (synth.c:520-613):
// for every frame, for every voice: phase-accumulator oscillator with table lookup
// (osc_next, synth.c:217-275), sample&hold / bit-crush (synth.c:560-574,341-345), RBJ biquad
// (mmf_process, synth.c:349-364), linear ADSR keyed on the global sample counter
// (amp_envelope_step, synth.c:398-431), one-pole amp smoother (synth.c:588-593), pan and the
// polyphonic stereo sum (synth.c:595-612); then the master volume stage (synth.c:616-624).
//
// Mapping to the machine:
//   * one lane per voice, 64 voices per wavefront, 256 per workgroup pass; the time loop runs
//     INSIDE the kernel with all recurrences (phase, smoother, biquad delay line) in registers;
//   * voice parameters/state are 16-byte planes (skred_device_layout.h): one coalesced
//     dwordx4 load per plane per launch, one dwordx4 store per read-write plane;
//   * wavetables are staged into LDS once per workgroup when the pool fits (gather = ds_read);
//     larger pools (PCM) are gathered from L2/HBM;
//   * per frame the 64 lanes' L/R are summed with a fixed-order DPP reduction (no LDS traffic,
//     no atomics: results are bit-reproducible run to run); wave sums meet in LDS every
//     SK_CHUNK frames and leave as coalesced stores into a per-workgroup partial mix;
//   * no MFMA: this is gather + multiply-add along a serial recurrence, not a contraction.
//
// Arithmetic contract (must match oracle/cpu_ref.c bit for bit per voice): compiled with
// -ffp-contract=off (no FMA fusion), fp32 subnormals kept (hipcc default), IEEE-rounded
// divide (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt), exact fmod.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "skred_device_layout.h"

#define LCG_A 6364136223846793005ULL
#define LCG_C 1442695040888963407ULL

// ---------------------------------------------------------------- wave reduction (DPP)

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_take(float v) {
  // lanes whose row is masked out receive 0.0f (the `old` operand)
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}

// Sum over the 64 lanes of a wavefront; the total lands in lane 63.  Fixed association order.
// (reference form; the kernels use wave_sum2_to_lane63 below, which performs the same additions)
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v += dpp_take<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
  v += dpp_take<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
  v += dpp_take<0x141, 0xF>(v);   // row_half_mirror
  v += dpp_take<0x140, 0xF>(v);   // row_mirror           -> every lane holds its 16-lane row sum
  v += dpp_take<0x142, 0xA>(v);   // row_bcast:15 into rows 1,3
  v += dpp_take<0x143, 0xC>(v);   // row_bcast:31 into rows 2,3 -> lane 63 holds the wave sum
  return v;
}

// The same two reductions (L and R) as 12 v_add_f32 with the lane permutation folded into the
// add's DPP operand.  Written as one asm block because hipcc otherwise SLP-packs L/R into
// v_pk_add_f32, which cannot take a DPP operand, and then spends 5 instructions per stage
// (2 x v_mov 0, 2 x v_mov_dpp, v_pk_add) = 30 per frame.  The two chains are interleaved and padded
// with s_nop so that every DPP read sits >= 2 wait states behind the VALU write of its source
// (hipcc pads nothing inside an asm statement).  Rows masked off by row_mask keep their value.
__device__ __forceinline__ void wave_sum2_to_lane63(float &l, float &r) {
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 0\n\t"
      "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(l), "+v"(r));
}

// Two frames at once: four independent chains (L0, R0, L1, R1), so every DPP read already sits three
// instructions behind the write of its source and no s_nop padding is needed between the stages.
__device__ __forceinline__ void wave_sum4_to_lane63(float &l0, float &r0, float &l1, float &r1) {
#define SK_DPP4(CTRL)                                                \
  "v_add_f32_dpp %0, %0, %0 " CTRL " bank_mask:0xf\n\t"              \
  "v_add_f32_dpp %1, %1, %1 " CTRL " bank_mask:0xf\n\t"              \
  "v_add_f32_dpp %2, %2, %2 " CTRL " bank_mask:0xf\n\t"              \
  "v_add_f32_dpp %3, %3, %3 " CTRL " bank_mask:0xf\n\t"
  asm volatile("s_nop 1\n\t"
               SK_DPP4("quad_perm:[1,0,3,2] row_mask:0xf")
               SK_DPP4("quad_perm:[2,3,0,1] row_mask:0xf")
               SK_DPP4("row_half_mirror row_mask:0xf")
               SK_DPP4("row_mirror row_mask:0xf")
               SK_DPP4("row_bcast:15 row_mask:0xa")
               SK_DPP4("row_bcast:31 row_mask:0xc")
               "s_nop 1"
               : "+v"(l0), "+v"(r0), "+v"(l1), "+v"(r1));
#undef SK_DPP4
}

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

// ---------------------------------------------------------------- render kernel

// LDS: [lds_table_floats] staged table pool, then [4 waves][SK_CHUNK][2] wave sums.
template <bool TAB_LDS, bool STEMS>
__global__ __launch_bounds__(SK_GROUP) void sk_render_kernel(const sk_render_args_t a) {
  extern __shared__ float lds[];
  float *lds_tab = lds;
  float2 *wsum = reinterpret_cast<float2 *>(lds + (TAB_LDS ? a.lds_table_floats : 0));

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  if (TAB_LDS) {
    // stage the whole pool; float4 when aligned, coalesced
    const int n4 = a.lds_table_floats >> 2;
    const float4 *src4 = reinterpret_cast<const float4 *>(a.tables);
    float4 *dst4 = reinterpret_cast<float4 *>(lds_tab);
    for (int i = tid; i < n4; i += SK_GROUP) dst4[i] = src4[i];
    for (int i = (n4 << 2) + tid; i < a.lds_table_floats; i += SK_GROUP) lds_tab[i] = a.tables[i];
    __syncthreads();
  }

  const size_t part_base = (size_t)blockIdx.x * (size_t)a.num_frames * 2;
  bool first_pass = true;

  for (int g = blockIdx.x; g < a.n_groups; g += gridDim.x) {
    const int v = g * SK_GROUP + tid;
    VoiceRegs r;
    load_voice(a, v, r);

    uint64_t rng = a.rng0;
    for (int c0 = 0; c0 < a.num_frames; c0 += SK_CHUNK) {
      const int cn = min(SK_CHUNK, a.num_frames - c0);
      for (int j = 0; j < cn; ++j) {
        const int i = c0 + j;
        const uint64_t now = a.count0 + (uint64_t)i + 1;       // synth.c:521 (pre-increment)
        float white = 0.0f;
        if (a.features & SKB_ANY_NOISE) {                        // synth.c:525, one draw per frame
          rng = rng * LCG_A + LCG_C;
          white = (float)((int32_t)(uint32_t)(rng >> 32)) / 2147483648.0f;
        }
        float l, rr;
        voice_frame<TAB_LDS>(r, lds_tab, a.tables, now, white, a.interp, l, rr);
        if (STEMS) {
          if (v < a.n_voices)
            reinterpret_cast<float2 *>(a.stems)[(size_t)i * (size_t)a.n_voices + (size_t)v] =
                make_float2(l, rr);
        }
        float sl = l, sr = rr;
        wave_sum2_to_lane63(sl, sr);
        if (lane == 63) wsum[wave * SK_CHUNK + j] = make_float2(sl, sr);
      }
      __syncthreads();
      if (tid < 2 * cn) {
        const float *w = reinterpret_cast<const float *>(wsum);
        float s = w[0 * 2 * SK_CHUNK + tid];
        s += w[1 * 2 * SK_CHUNK + tid];
        s += w[2 * 2 * SK_CHUNK + tid];
        s += w[3 * 2 * SK_CHUNK + tid];
        float *p = a.partial + part_base + (size_t)c0 * 2 + tid;
        if (first_pass) *p = s; else *p += s;
      }
      __syncthreads();
    }
    store_voice(a, v, r);
    first_pass = false;
  }
}

// ---------------------------------------------------------------- modulated banks
//
// Cross-voice modulation (FM synth.c:548-558, AM 584-587, pan 597-602, CZ amount 262-267) and CZ
// phase distortion (cz_phasor, synth.c:149-215).  The reference walks voices in index order inside a
// frame, so a carrier n sees THIS frame's voice_sample[m] of a modulator m < n and the PREVIOUS
// frame's of a modulator m > n.  Here a workgroup is one wavefront = one aligned group of 64 voices
// (the reference's VOICE_MAX; the host refuses banks whose modulators leave their group).  The host
// assigns every voice a dependency level (0 = needs no same-frame value; else 1 + max level of its
// modulators with a lower index); per frame the levels run one after the other, exchanging
// voice_sample through two LDS arrays (previous / current frame).  Cost = (max level + 1) passes per
// frame -- irrelevant for the 64-voice drop-in, and banks without modulators never come here.

// == fast_pow, synth.c:140-147.  The float->int cast saturates on the GPU exactly where the x86
// conversion of the reference returns INT_MIN (large negative products), so the bits agree.
__device__ __forceinline__ float pow_bits(float base, float expo) {
  if (base <= 0.0f) return 0.0f;
  int i = __float_as_int(base);
  i = (int)(expo * (float)(i - 1065353216) + 1065353216.0f);
  return __int_as_float(i);
}

// == cz_phasor, synth.c:149-215
__device__ float cz_warp(int mode, float table_phase, float amount, int table_size) {
  const float size_f = (float)table_size;
  float x = table_phase / size_f;
  float d = amount;
  if (d < 0.0f) d = 0.0f; else if (d > 0.999f) d = 0.999f;
  switch (mode) {
    case 1: {
      const float k_lo = 0.5f / d, k_hi = 0.5f / (1.0f - d);
      x = (x < d) ? x * k_lo : 0.5f + (x - d) * k_hi;
      break;
    }
    case 2: {
      const float k = 0.5f / (0.5f - d * 0.5f);
      x = (x < 0.5f) ? x * k : 1.0f - (1.0f - x) * k;
      break;
    }
    case 3: {
      const float k = 0.5f / (0.5f - d * 0.5f);
      x = (x < 0.5f) ? x * k : 0.5f + (x - 0.5f) * k;
      break;
    }
    case 4: x = fmodf(x * 2.0f, 1.0f); break;
    case 5: {
      const float h = d * 0.5f;
      const float k_lo = 0.5f / (0.5f - h), k_hi = 0.5f / (0.5f + h);
      x = (x < 0.5f) ? x * k_lo : 0.5f + (x - 0.5f) * k_hi;
      break;
    }
    case 6: x = pow_bits(x, 1.0f + 4.0f * d); break;
    case 7: x = pow_bits(x, 1.0f + 8.0f * d); break;
    default: return table_phase;
  }
  return x * size_f;
}

struct ModRegs {
  int fm, am, pm, cz;            // modulator lane inside the 64-voice group, or -1
  float fm_depth, freq_scale, am_depth, pm_depth, cz_depth, cz_dist;
  int cz_mode, level;
  float inc_raw;                 // voice_phase_inc before the direction sign
};

// One voice, one frame, with modulation: synth.c:531-612.  `prev`/`cur` are the LDS exchange arrays.
__device__ __forceinline__ void voice_frame_mod(VoiceRegs &r, const ModRegs &m, int lane,
                                                const float *prev, float *cur, const float *incs,
                                                const float *__restrict__ tab, uint64_t now, float white,
                                                int interp, float &out_l, float &out_r) {
  auto other = [&](int src) -> float { return src < lane ? cur[src] : prev[src]; };
  out_l = 0.0f; out_r = 0.0f;
  float raw;
  if (r.flags & SKF_NOISE) {
    raw = white;
  } else {
    float inc = m.inc_raw;
    if (m.fm >= 0 && m.fm != lane) {                                   // synth.c:548-555
      const float g = other(m.fm) * m.fm_depth;
      inc = inc + (incs[m.fm] * m.freq_scale * g);
    }
    if (r.flags & SKF_REVERSE) inc = -inc;
    float ph = r.phase + inc;
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
      float pos = ph;
      if (m.cz_mode) {                                                 // synth.c:262-267
        // the carrier's own voice_sample still holds last frame's value at this point
        const float dm = (m.cz >= 0) ? (m.cz == lane ? prev[lane] : other(m.cz)) * m.cz_depth : 1.0f;
        pos = cz_warp(m.cz_mode, ph, m.cz_dist + dm, r.tsize);
      }
      raw = table_fetch<false>(nullptr, tab, r, pos, interp, !stops);
    }
  }
  if (r.hold_max) {
    if (r.hold_count == 0) r.hold = raw;
    raw = r.hold;
    if (++r.hold_count >= r.hold_max) r.hold_count = 0;
  }
  float s = raw;
  if (r.quant) s = crush(s, r.quant);
  if (r.flags & SKF_FILTER) {
    float y = r.b0 * s;
    y = y + r.b1 * r.x1;
    y = y + r.b2 * r.x2;
    y = y - r.a1 * r.y1;
    y = y - r.a2 * r.y2;
    r.x2 = r.x1; r.x1 = s;
    r.y2 = r.y1; r.y1 = y;
    s = y;
  }
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
  float am = 1.0f;                                                     // synth.c:583-587
  if (m.am >= 0) am = (m.am == lane ? s : other(m.am)) * m.am_depth;   // own slot holds the post-filter sample
  float gain = r.amp * env * am;
  if (r.flags & SKF_SMOOTH) {
    r.sgain += r.smooth_k * (gain - r.sgain);
    gain = r.sgain;
  }
  s *= gain;
  r.sample = s;
  if (!(r.flags & SKF_MUTED)) {
    if (m.pm >= 0) {                                                   // synth.c:597-602
      const float q = (m.pm == lane ? s : other(m.pm)) * m.pm_depth;
      r.pan_l = (1.0f - q) / 2.0f;
      r.pan_r = (1.0f + q) / 2.0f;
    }
    out_l = s * r.pan_l;
    out_r = s * r.pan_r;
  }
}

template <bool STEMS>
__global__ __launch_bounds__(64) void sk_render_mod_kernel(const sk_render_args_t a, const int *__restrict__ levels,
                                                           int max_level) {
  __shared__ float xch[2][64];
  __shared__ float incs[64];
  const int lane = threadIdx.x;
  const int v = blockIdx.x * 64 + lane;
  VoiceRegs r;
  load_voice(a, v, r);
  ModRegs m;
  {
    const uint4 mi = *reinterpret_cast<const uint4 *>(&a.ro[SKP_MODI][v]);
    const uint4 mf = *reinterpret_cast<const uint4 *>(&a.ro[SKP_MODF][v]);
    const uint4 mx = *reinterpret_cast<const uint4 *>(&a.ro[SKP_MODX][v]);
    const uint4 fl = *reinterpret_cast<const uint4 *>(&a.ro[SKP_FILT][v]);
    m.fm = (int)mi.x; m.am = (int)mi.y; m.pm = (int)mi.z; m.cz = (int)mi.w;
    m.fm_depth = __uint_as_float(mf.x); m.freq_scale = __uint_as_float(mf.y);
    m.am_depth = __uint_as_float(mf.z); m.pm_depth = __uint_as_float(mf.w);
    m.cz_depth = __uint_as_float(mx.x); m.cz_mode = (int)mx.y;
    m.cz_dist = __uint_as_float(fl.w);
    m.level = levels[v];
    m.inc_raw = (r.flags & SKF_REVERSE) ? -r.inc : r.inc;    // load_voice applied the direction sign
  }
  incs[lane] = m.inc_raw;
  xch[0][lane] = r.sample;                                   // voice_sample[] as the last callback left it
  __syncthreads();

  uint64_t rng = a.rng0;
  int cur_i = 1;
  float *part = a.partial + (size_t)blockIdx.x * (size_t)a.num_frames * 2;
  for (int i = 0; i < a.num_frames; ++i) {
    const uint64_t now = a.count0 + (uint64_t)i + 1;
    rng = rng * LCG_A + LCG_C;
    const float white = (float)((int32_t)(uint32_t)(rng >> 32)) / 2147483648.0f;
    float *cur = xch[cur_i];
    const float *prev = xch[cur_i ^ 1];
    const bool live = !((r.rw & SKR_FINISHED) || r.amp == 0.0f || (r.flags & SKF_INERT));
    if (!live) { r.sample = 0.0f; cur[lane] = 0.0f; }         // synth.c:531-542
    __syncthreads();
    float l = 0.0f, rr = 0.0f;
    for (int lev = 0; lev <= max_level; ++lev) {
      if (live && m.level == lev) {
        voice_frame_mod(r, m, lane, prev, cur, incs, a.tables, now, white, a.interp, l, rr);
        cur[lane] = r.sample;
      }
      __syncthreads();
    }
    if (STEMS) {
      if (v < a.n_voices)
        reinterpret_cast<float2 *>(a.stems)[(size_t)i * (size_t)a.n_voices + (size_t)v] = make_float2(l, rr);
    }
    float sl = l, sr = rr;
    wave_sum2_to_lane63(sl, sr);
    if (lane == 63) reinterpret_cast<float2 *>(part)[i] = make_float2(sl, sr);
    cur_i ^= 1;
  }
  store_voice(a, v, r);
}

// ---------------------------------------------------------------- fast render kernel
//
// Same arithmetic, per voice bit-identical to the generic kernel above, for "clean" banks -- the
// host only selects it when (bank-wide, skred_bank.c:classify): no one-shot voice that stops at
// its table end, no reverse playback, no sample&hold / bit-crush / noise voices, no modulators,
// smoother on everywhere, filter on for ALL voices or for none, envelope for ALL or none, and
// every phase / increment / loop bound finite (so the !isfinite() branch of osc_next can never
// fire and `voice_finished` cannot change inside a launch).  Under those conditions:
//   * which lanes are skipped (synth.c:531-542) is a launch constant -> one mask, no per-frame
//     branch; their state is left untouched in HBM and their output is select-masked to 0;
//   * the wrap is straight-line: for span <= x < 2*span, fmodf(x, span) == x - span exactly
//     (Sterbenz), anything else (huge increments, NaN) drops into the exact generic path;
//   * envelope time is carried as a float that gains 1.0f per frame: exact below 2^24 frames
//     and then equal to the reference's (float)(uint64) conversion; once every lane of the wave is
//     in its sustain stage (monotone within a launch) the envelope costs nothing per frame and
//     amp*env is a per-lane constant.
// What remains per frame is ~45 VALU instructions instead of ~107, with almost no scalar
// branch bookkeeping.

// two adjacent table samples, fetched with one 4-byte-aligned 8-byte access (global_load_dwordx2 /
// ds_read2_b32).  The pool is padded by the host so that reading one float past any table is in bounds.
struct __attribute__((packed, aligned(4))) tap_pair_t { float a, b; };

struct FastRegs {
  // launch constants
  float inc, lo, hi, span, span2, amp;
  int toff4, tsize_m1;          // byte offset of the table in the pool, table_size - 1
  float att, attdec, dec, sus, one_m_sus, rel, vel;
  float k, b0, b1, b2, a1, a2, pan_l, pan_r;
  float gain_sustain;           // amp * (sustain_level * velocity)
  // recurrences
  float phase, sgain, x1, x2, y1, y2, sample;
  float tf, trf;                // frames since note-on / note-off for the CURRENT frame
  uint32_t rw;
};

// NOCLAMP: the caller guarantees 0 <= lo <= pos < hi <= table_size (TAME loops), so the reference's
// index clamps (synth.c:271-272) can never act and are dropped.
template <bool TAB_LDS, int INTERP, bool NOCLAMP>
__device__ __forceinline__ float fast_fetch(const char *lds_tab, const char *__restrict__ glb_tab,
                                            const FastRegs &r, float pos) {
  int idx = (int)pos;
  if (!NOCLAMP) idx = max(min(idx, r.tsize_m1), 0);          // clamp, synth.c:271-272
  const char *tab = TAB_LDS ? lds_tab : glb_tab;
  if (INTERP == 0) return *reinterpret_cast<const float *>(tab + (r.toff4 + (idx << 2)));
  // linear: oracle/cpu_ref.c:table_fetch.  Every voice of a fast bank wraps (no stopping one-shots).
  // Both taps come from ONE 8-byte gather (4-byte aligned pair) -- the neighbour is idx+1 except on
  // the last sample before the loop end, where a second (rare) gather fetches the loop start.
  const tap_pair_t pr = *reinterpret_cast<const tap_pair_t *>(tab + (r.toff4 + (idx << 2)));
  const float a = pr.a;
  float b = pr.b;
  int nxt = idx + 1;
  bool special = (float)nxt >= r.hi;
  if (special) nxt = (int)r.lo;
  if (!NOCLAMP) { const int c = max(min(nxt, r.tsize_m1), 0); special = special || (c != nxt); nxt = c; }
  if (special) b = *reinterpret_cast<const float *>(tab + (r.toff4 + (nxt << 2)));
  const float frac = pos - (float)idx;
  return a + frac * (b - a);
}

// exact wrap for the cases the straight-line code does not cover (synth.c:241-256, looping voice)
__device__ __forceinline__ float slow_wrap(float ph, float lo, float hi, float span) {
  if (!__builtin_isfinite(ph)) return 0.0f;     // unreachable for a fast bank; kept total
  if (ph >= hi) return lo + fmod_pos(ph - lo, span);
  if (ph < lo) return hi - fmod_pos(lo - ph, span);
  return ph;
}

// One voice, one frame.  STEADY: every lane of the wave sits in its sustain stage.  When !STEADY the
// caller has set r.tf / r.trf to this frame's envelope clocks.
// TAME: the caller has proved for every lane that lo <= phase <= hi and 0 <= inc <= span/2, which
// by induction keeps phase+inc in [lo, hi + span/2]: the only wrap that can occur is the simple one.
template <bool TAB_LDS, bool FILTER, bool ENV, bool STEADY, bool TAME, int INTERP>
__device__ __forceinline__ void fast_frame(FastRegs &r, float &xn, float &xo, float &yn, float &yo,
                                           const bool released, const char *lds_tab,
                                           const char *__restrict__ glb_tab, float &out_l, float &out_r) {
  // ---- oscillator (osc_next, synth.c:217-275) ----
  const float ph0 = r.phase + r.inc;
  const float x = ph0 - r.lo;
  const bool over = ph0 >= r.hi;
  float ph;
  if (TAME) {
    ph = over ? r.lo + (x - r.span) : ph0;
  } else {
    const bool simple = over && (x < r.span2);          // one loop length past the end: x - span exact
    const bool in_range = (ph0 >= r.lo) && !over;
    ph = simple ? r.lo + (x - r.span) : ph0;
    if (!(in_range || simple)) ph = slow_wrap(ph0, r.lo, r.hi, r.span);
  }
  r.phase = ph;
  float s = fast_fetch<TAB_LDS, INTERP, TAME>(lds_tab, glb_tab, r, ph);
  // ---- biquad (mmf_process, synth.c:349-364) ----
  if (FILTER) {
    // xn/yn: newest delay-line entries, xo/yo: the older ones.  The new values overwrite the OLD
    // slots, so the caller alternates the argument order frame by frame instead of shifting
    // registers (x2 = x1; x1 = s costs four v_mov per frame).
    float y = r.b0 * s;
    y = y + r.b1 * xn;
    y = y + r.b2 * xo;
    y = y - r.a1 * yn;
    y = y - r.a2 * yo;
    xo = s;
    yo = y;
    s = y;
  }
  // ---- envelope (amp_envelope_step, synth.c:398-431) and gain (synth.c:580-588) ----
  float gain;
  if (!ENV) {
    gain = r.amp;                                   // amp * 1.0f * 1.0f
  } else if (STEADY) {
    gain = r.gain_sustain;                          // e = sustain_level on every lane
  } else {
    float e = 0.0f;
    if (r.rw & SKR_ENV_ACTIVE) {
      if (r.tf < r.att) {
        e = r.tf / r.att;
      } else if (r.tf < r.attdec) {
        const float prog = (r.tf - r.att) / r.dec;
        e = 1.0f - prog * r.one_m_sus;
      } else if (!released) {
        e = r.sus;
      } else if (r.trf < r.rel) {
        const float prog = r.trf / r.rel;
        e = r.sus * (1.0f - prog);
      } else {
        r.rw &= ~SKR_ENV_ACTIVE;
      }
    }
    gain = r.amp * (e * r.vel);
  }
  // ---- smoother + apply (synth.c:589-593), pan (synth.c:603-604) ----
  r.sgain += r.k * (gain - r.sgain);
  s *= r.sgain;
  r.sample = s;
  out_l = s * r.pan_l;
  out_r = s * r.pan_r;
}

// (timing experiments only: -DSK_ABLATE_REDUCE drops the cross-lane sum; outputs are then wrong)
#ifdef SK_ABLATE_REDUCE
#define SK_REDUCE_AND_STORE(J) asm volatile("" ::"v"(l), "v"(rr));
#else
#define SK_REDUCE_AND_STORE(J)       \
  wave_sum2_to_lane63(l, rr);        \
  if (lane == 63) wsum[wave * SK_CHUNK + (J)] = make_float2(l, rr);
#endif
#ifdef SK_ABLATE_REDUCE
#define SK_REDUCE4_AND_STORE(J) asm volatile("" ::"v"(l0), "v"(r0), "v"(l1), "v"(r1));
#else
#define SK_REDUCE4_AND_STORE(J)                      \
  wave_sum4_to_lane63(l0, r0, l1, r1);               \
  if (lane == 63) *reinterpret_cast<float4 *>(&wsum[wave * SK_CHUNK + (J)]) = make_float4(l0, r0, l1, r1);
#endif
// one frame of the chunk loop: STEADY_ selects the envelope mode, A/B the delay-line roles
#define SK_FAST_FRAME(J, STEADY_, XN, XO, YN, YO)                                                        \
  {                                                                                                      \
    float l, rr;                                                                                         \
    fast_frame<TAB_LDS, FILTER, ENV, STEADY_, false, INTERP>(r, XN, XO, YN, YO, released, lds_tab, glb_tab, l, rr); \
    l = silent ? 0.0f : l; rr = silent ? 0.0f : rr;                                                      \
    SK_REDUCE_AND_STORE(J)                                                                               \
  }
// two steady frames (J even, J+1): delay-line roles swap in between, one 4-chain reduction, one 16-byte store
#define SK_FAST_PAIR_STEADY(J, TAME_) /* TAME_ loops run only when no live lane is muted: no output select */ \
  {                                                                                                      \
    float l0, r0, l1, r1;                                                                                \
    fast_frame<TAB_LDS, FILTER, ENV, true, TAME_, INTERP>(r, r.x1, r.x2, r.y1, r.y2, released, lds_tab, glb_tab, l0, r0); \
    fast_frame<TAB_LDS, FILTER, ENV, true, TAME_, INTERP>(r, r.x2, r.x1, r.y2, r.y1, released, lds_tab, glb_tab, l1, r1); \
    if (!(TAME_)) { l0 = silent ? 0.0f : l0; r0 = silent ? 0.0f : r0; l1 = silent ? 0.0f : l1; r1 = silent ? 0.0f : r1; } \
    SK_REDUCE4_AND_STORE(J)                                                                              \
  }
#define SK_FAST_EVEN(J, STEADY_) SK_FAST_FRAME(J, STEADY_, r.x1, r.x2, r.y1, r.y2)
#define SK_FAST_ODD(J, STEADY_) SK_FAST_FRAME(J, STEADY_, r.x2, r.x1, r.y2, r.y1)
#define SK_FAST_FIX_ODD_TAIL()                                                     \
  { float t_ = r.x1; r.x1 = r.x2; r.x2 = t_; t_ = r.y1; r.y1 = r.y2; r.y2 = t_; }

#ifndef SK_FAST_MIN_WAVES
#define SK_FAST_MIN_WAVES 6      /* waves per SIMD the register allocator must leave room for */
#endif
template <bool TAB_LDS, bool FILTER, bool ENV, int INTERP>
__global__ __launch_bounds__(SK_GROUP, SK_FAST_MIN_WAVES) void sk_render_fast_kernel(const sk_render_args_t a) {
  extern __shared__ float lds[];
  float2 *wsum = reinterpret_cast<float2 *>(lds + (TAB_LDS ? a.lds_table_floats : 0));
  const char *lds_tab = reinterpret_cast<const char *>(lds);
  const char *glb_tab = reinterpret_cast<const char *>(a.tables);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  if (TAB_LDS) {
    const int n4 = a.lds_table_floats >> 2;           // padded to a multiple of 4 by the host
    const float4 *src4 = reinterpret_cast<const float4 *>(a.tables);
    float4 *dst4 = reinterpret_cast<float4 *>(lds);
    for (int i = tid; i < n4; i += SK_GROUP) dst4[i] = src4[i];
    __syncthreads();
  }

  const size_t part_base = (size_t)blockIdx.x * (size_t)a.num_frames * 2;
  bool first_pass = true;

  for (int g = blockIdx.x; g < a.n_groups; g += gridDim.x) {
    const int v = g * SK_GROUP + tid;
    FastRegs r;
    bool dead, silent;            // dead: skipped by synth.c:531-542; silent: dead or muted
    bool released = false;
    uint64_t t_start = 0, t_release = 0;
    {
      const uint4 osc = *reinterpret_cast<const uint4 *>(&a.ro[SKP_OSC][v]);
      const uint4 tab = *reinterpret_cast<const uint4 *>(&a.ro[SKP_TAB][v]);
      const uint4 gn = *reinterpret_cast<const uint4 *>(&a.ro[SKP_GAIN][v]);
      const uint4 s0 = *reinterpret_cast<const uint4 *>(&a.rw[SKS_OSC][v]);
      const uint4 s1 = *reinterpret_cast<const uint4 *>(&a.rw[SKS_FILT][v]);
      const uint4 s2 = *reinterpret_cast<const uint4 *>(&a.rw[SKS_MISC][v]);
      r.inc = __uint_as_float(osc.x); r.lo = __uint_as_float(osc.y);
      r.hi = __uint_as_float(osc.z);  r.amp = __uint_as_float(osc.w);
      r.span = r.hi - r.lo; r.span2 = r.span + r.span;
      r.toff4 = (int)tab.x << 2; r.tsize_m1 = (int)tab.y - 1;
      const uint32_t flags = tab.z;
      r.vel = __uint_as_float(gn.x); r.k = __uint_as_float(gn.y);
      r.b0 = __uint_as_float(gn.z);  r.b1 = __uint_as_float(gn.w);
      r.phase = __uint_as_float(s0.x); r.sgain = __uint_as_float(s0.y);
      r.x1 = __uint_as_float(s0.z);    r.x2 = __uint_as_float(s0.w);
      r.y1 = __uint_as_float(s1.x);    r.y2 = __uint_as_float(s1.y);
      r.sample = __uint_as_float(s1.z); r.rw = s1.w;
      r.pan_l = __uint_as_float(s2.z); r.pan_r = __uint_as_float(s2.w);
      r.tf = 0.0f; r.trf = 0.0f;
      if (FILTER) {
        const uint4 fl = *reinterpret_cast<const uint4 *>(&a.ro[SKP_FILT][v]);
        r.b2 = __uint_as_float(fl.x); r.a1 = __uint_as_float(fl.y); r.a2 = __uint_as_float(fl.z);
      }
      if (ENV) {
        const uint4 et = *reinterpret_cast<const uint4 *>(&a.ro[SKP_ENV_T][v]);
        const uint4 es = *reinterpret_cast<const uint4 *>(&a.ro[SKP_ENV_S][v]);
        r.att = __uint_as_float(et.x); r.dec = __uint_as_float(et.y);
        r.sus = __uint_as_float(et.z); r.rel = __uint_as_float(et.w);
        r.attdec = r.att + r.dec;                    // synth.c:410: decay_start + decay_time
        r.one_m_sus = 1.0f - r.sus;                  // synth.c:413
        r.gain_sustain = r.amp * (r.sus * r.vel);    // synth.c:582,588 in the sustain stage
        t_start = ((uint64_t)es.y << 32) | es.x;
        t_release = ((uint64_t)es.w << 32) | es.z;
        released = t_release != 0;                   // synth.c:417
      }
      dead = (r.rw & SKR_FINISHED) || r.amp == 0.0f || (flags & SKF_INERT);
      silent = dead || (flags & SKF_MUTED);
    }
    // wrap can only ever be the simple one (see fast_frame<TAME>): decided once per pass
    if (dead) {
      // a skipped voice is never stored back (see the end of the pass): give its lane inert numbers
      // so that it contributes exact zeros and its table index stays at 0, whatever its real state is
      r.inc = 0.0f; r.lo = 0.0f; r.hi = 1.0f; r.span = 1.0f; r.span2 = 2.0f; r.phase = 0.0f;
      r.toff4 = 0; r.tsize_m1 = 0;
      r.k = 0.0f; r.sgain = 0.0f; r.amp = 0.0f; r.gain_sustain = 0.0f;
      r.b0 = r.b1 = r.b2 = r.a1 = r.a2 = 0.0f; r.x1 = r.x2 = r.y1 = r.y2 = 0.0f;
      r.pan_l = r.pan_r = 0.0f; r.rw &= ~SKR_ENV_ACTIVE;
    }
    // TAME (decided once per pass): the only wrap that can occur is the simple one and the table index
    // needs no clamp -- see fast_frame<TAME> / fast_fetch<NOCLAMP>
    const bool tame = __all(dead || (r.inc >= 0.0f && r.inc <= 0.5f * r.span && r.phase >= r.lo && r.phase <= r.hi &&
                                     r.lo >= 0.0f && r.hi <= (float)(r.tsize_m1 + 1))) && !__any(silent && !dead);

    for (int c0 = 0; c0 < a.num_frames; c0 += SK_CHUNK) {
      const int cn = min(SK_CHUNK, a.num_frames - c0);
      bool steady = true, exact = true;
      if (ENV) {
        // Envelope clocks for this chunk from the integer timeline: frame c0+j has
        // now = count0 + c0 + j + 1 (synth.c:521).  d_* are the clocks of "frame c0 - 1".
        const uint64_t base = a.count0 + (uint64_t)c0;
        const uint64_t d_on = base - t_start;
        const uint64_t d_off = base - t_release;
        const uint64_t lim = (1ull << 24) - (uint64_t)SK_CHUNK - 2;   // x + 1.0f stays exact below 2^24
        exact = __all(dead || ((d_on < lim) && (!released || d_off < lim)));
        r.tf = (float)d_on;
        r.trf = released ? (float)d_off : 0.0f;
        // sustain is absorbing within a launch: the clock only grows and note-off arrives between launches
        const float tf_first = (float)(d_on + 1);
        steady = __all(dead || ((r.rw & SKR_ENV_ACTIVE) && !released && !(tf_first < r.attdec)));
      }
      if ((!ENV || steady) && tame) {
        int j = 0;
        for (; j + 1 < cn; j += 2) SK_FAST_PAIR_STEADY(j, true)
        if (j < cn) { SK_FAST_EVEN(j, true) SK_FAST_FIX_ODD_TAIL() }
      } else if (!ENV || steady) {
        int j = 0;
        for (; j + 1 < cn; j += 2) SK_FAST_PAIR_STEADY(j, false)
        if (j < cn) { SK_FAST_EVEN(j, true) SK_FAST_FIX_ODD_TAIL() }
      } else if (exact) {
        int j = 0;
        for (; j + 1 < cn; j += 2) {                 // clocks == (float)(now - sample_start), exact below 2^24
          r.tf += 1.0f; r.trf += 1.0f;
          SK_FAST_EVEN(j, false)
          r.tf += 1.0f; r.trf += 1.0f;
          SK_FAST_ODD(j + 1, false)
        }
        if (j < cn) { r.tf += 1.0f; r.trf += 1.0f; SK_FAST_EVEN(j, false) SK_FAST_FIX_ODD_TAIL() }
      } else {
        for (int j = 0; j < cn; ++j) {               // clocks past 2^24 frames: integer path, synth.c:401,422
          const uint64_t now = a.count0 + (uint64_t)(c0 + j) + 1;
          r.tf = (float)(now - t_start); r.trf = (float)(now - t_release);
          SK_FAST_EVEN(j, false)
          SK_FAST_FIX_ODD_TAIL()
        }
      }
      __syncthreads();
      if (tid < 2 * cn) {
        const float *w = reinterpret_cast<const float *>(wsum);
        float s = w[0 * 2 * SK_CHUNK + tid];
        s += w[1 * 2 * SK_CHUNK + tid];
        s += w[2 * 2 * SK_CHUNK + tid];
        s += w[3 * 2 * SK_CHUNK + tid];
        float *p = a.partial + part_base + (size_t)c0 * 2 + tid;
        if (first_pass) *p = s; else *p += s;
      }
      __syncthreads();
    }

    // store the recurrences; skipped voices keep their state and get voice_sample = 0 (synth.c:532,538)
    if (!dead) {
      uint4 s0, s1;
      s0.x = __float_as_uint(r.phase); s0.y = __float_as_uint(r.sgain);
      s0.z = __float_as_uint(r.x1);    s0.w = __float_as_uint(r.x2);
      s1.x = __float_as_uint(r.y1);    s1.y = __float_as_uint(r.y2);
      s1.z = __float_as_uint(r.sample); s1.w = r.rw;
      *reinterpret_cast<uint4 *>(&a.rw[SKS_OSC][v]) = s0;
      *reinterpret_cast<uint4 *>(&a.rw[SKS_FILT][v]) = s1;
    } else {
      reinterpret_cast<uint32_t *>(&a.rw[SKS_FILT][v])[2] = 0u;
    }
    first_pass = false;
  }
}

// ---------------------------------------------------------------- fast kernel, two voices per lane
//
// Same per-voice arithmetic as sk_render_fast_kernel, but every lane carries TWO voices (v and v+64
// of a 128-voice wave slice) as 2-vectors, so that
//   * the mul/add backbone (phase add, wrap, biquad, smoother, gain, pan) issues as packed fp32
//     (v_pk_add_f32 / v_pk_mul_f32: two voices per instruction, each lane-op still IEEE fp32,
//     unfused, hence bit-identical), and
//   * the 12-instruction cross-lane DPP reduction is paid once per 128 voices instead of per 64.
// Compares, selects, float->int conversion and the LDS gathers stay per voice.  Used for large
// clean banks (the host decides, skred_bank.c); per-voice results equal the other two kernels'.

typedef float v2f __attribute__((ext_vector_type(2)));

#ifndef SK_LDS_REDUCE
#define SK_LDS_REDUCE 1     /* 1: tame chunks sum across lanes through an LDS transposition tile (+11 % measured);
                               0: always the DPP butterfly */
#endif

// Two kernels share this machinery.  sk_render_fast2_kernel renders the 512-voice groups in which every
// voice holds a CONSTANT envelope level for the whole launch (sustain, inactive, finished release, or no
// envelope at all): its loop has no envelope code and a small register footprint.  Groups with any voice
// in attack / decay / release are only flagged by it and rendered right afterwards by
// sk_render_env2_kernel, which carries the envelope machinery (and its registers) alone.
struct Fast2Regs {
  v2f inc, lo, hi, span, span2;
  int toff4[2], tsize_m1[2];
  v2f k, b0, b1, b2, a1, a2, pan_l, pan_r, gain_const;   // gain_const: gain of a constant-level lane
  v2f phase, sgain, x1, x2, y1, y2, sample;
  uint32_t rw[2];
};

struct Env2Regs {                 // sk_render_env2_kernel only
  float att[2], attdec[2], dec[2], rel[2];
  v2f ampv, velv, susv, omsv;     // amp, velocity, sustain_level, 1 - sustain_level
  v2f clk, ebase, eden, erinv, eA, eB, eC;   // "ramp" spans: see fast2_env_span
  float tf[2], trf[2];            // general frames: this frame's clocks
};

template <bool TAB_LDS, int INTERP, bool NOCLAMP>
__device__ __forceinline__ float fast2_fetch(const char *lds_tab, const char *__restrict__ glb_tab,
                                             int toff4, int tsize_m1, float lo, float hi, float pos) {
  int idx = (int)pos;
  if (!NOCLAMP) idx = max(min(idx, tsize_m1), 0);
  const char *tab = TAB_LDS ? lds_tab : glb_tab;
  if (INTERP == 0) return *reinterpret_cast<const float *>(tab + (toff4 + (idx << 2)));
  const tap_pair_t pr = *reinterpret_cast<const tap_pair_t *>(tab + (toff4 + (idx << 2)));   // see fast_fetch
  const float a = pr.a;
  float b = pr.b;
  int nxt = idx + 1;
  bool special = (float)nxt >= hi;
  if (special) nxt = (int)lo;
  if (!NOCLAMP) { const int c = max(min(nxt, tsize_m1), 0); special = special || (c != nxt); nxt = c; }
  if (special) b = *reinterpret_cast<const float *>(tab + (toff4 + (nxt << 2)));
  const float frac = pos - (float)idx;
  return a + frac * (b - a);
}

// Which branch of amp_envelope_step (synth.c:398-431) a voice takes at envelope clocks (t, tr):
// 0 inactive, 1 attack, 2 decay, 3 sustain (held), 4 release, 5 release finished (is_active -> 0).
// The code is monotone in time within a launch (note-off only arrives between launches), so a voice
// whose code is the same on the first and the last frame of a span keeps it for the whole span, and
// codes 0 / 3 / 5 (constant level) are absorbing.
__device__ __forceinline__ int env_stage_code(bool active, bool released, float t, float tr, float att,
                                              float attdec, float rel) {
  if (!active) return 0;
  if (t < att) return 1;
  if (t < attdec) return 2;
  if (!released) return 3;
  return (tr < rel) ? 4 : 5;
}

// Per-lane envelope constants for a span of frames whose first / last frame have clocks (t1,tr1) / (tN,trN)
// and whose preceding frame has (t0,tr0).  Constant lanes get gain_const; moving lanes get the "ramp" form
//     q = (clk - ebase) / eden,  e = eC * (eA + eB * q)
// which is bit-identical to the reference's stage expressions:
//     attack  q               = 1*(0 + 1*q)            (synth.c:405)
//     decay   1 - q*(1-sus)   = 1*(1 + (-(1-sus))*q)   (synth.c:413)
//     release sus*(1 - q)     = sus*(1 + (-1)*q)       (synth.c:425)
// `st` stays true while every lane sits on a constant level; `same` while no lane changes stage in the span
// (and the denominators are in the range where the short division below equals the full IEEE expansion).
__device__ __forceinline__ void fast2_env_span(Fast2Regs &r, Env2Regs &e, int c, bool dead, bool released,
                                               float t1, float tr1, float tN, float trN, float t0, float tr0,
                                               bool &st, bool &same) {
  const bool act = (r.rw[c] & SKR_ENV_ACTIVE) != 0;
  const int code0 = env_stage_code(act, released, t1, tr1, e.att[c], e.attdec[c], e.rel[c]);
  const int code1 = env_stage_code(act, released, tN, trN, e.att[c], e.attdec[c], e.rel[c]);
  st = st && (dead || code0 == 0 || code0 == 3 || code0 == 5);
  const float level = code0 == 3 ? e.susv[c] : 0.0f;
  e.eA[c] = code0 == 1 ? 0.0f : 1.0f;
  e.eB[c] = code0 == 1 ? 1.0f : (code0 == 2 ? -e.omsv[c] : (code0 == 4 ? -1.0f : 0.0f));
  e.eC[c] = code0 == 4 ? e.susv[c] : ((code0 == 1 || code0 == 2) ? 1.0f : level);
  e.clk[c] = code0 == 4 ? tr0 : t0;
  e.ebase[c] = code0 == 2 ? e.att[c] : 0.0f;
  const float d = code0 == 1 ? e.att[c] : (code0 == 2 ? e.dec[c] : (code0 == 4 ? e.rel[c] : 1.0f));
  e.eden[c] = d;
  const float r0 = __builtin_amdgcn_rcpf(d);
  e.erinv[c] = __builtin_fmaf(__builtin_fmaf(-d, r0, 1.0f), r0, r0);   // refined reciprocal, as v_rcp + one fma step
  r.gain_const[c] = e.ampv[c] * (level * e.velv[c]);                   // amp * (e * velocity), synth.c:582,588
  same = same && (dead || (code0 == code1 && d >= 0x1p-40f && d <= 0x1p40f));
  // a release that has run out: the reference clears is_active on the first frame it notices (synth.c:429)
  if (!dead && code0 == 5) r.rw[c] &= ~SKR_ENV_ACTIVE;
}

// General frames (a lane changes stage inside the block): amp_envelope_step as the reference writes it.
__device__ __forceinline__ float fast2_env_general(Fast2Regs &r, Env2Regs &e, int c, bool released) {
  float lvl = 0.0f;
  if (r.rw[c] & SKR_ENV_ACTIVE) {
    if (e.tf[c] < e.att[c]) {
      lvl = e.tf[c] / e.att[c];
    } else if (e.tf[c] < e.attdec[c]) {
      const float prog = (e.tf[c] - e.att[c]) / e.dec[c];
      lvl = 1.0f - prog * e.omsv[c];
    } else if (!released) {
      lvl = e.susv[c];
    } else if (e.trf[c] < e.rel[c]) {
      const float prog = e.trf[c] / e.rel[c];
      lvl = e.susv[c] * (1.0f - prog);
    } else {
      r.rw[c] &= ~SKR_ENV_ACTIVE;
    }
  }
  return e.ampv[c] * (lvl * e.velv[c]);
}

// EM (envelope mode): 0 every lane has a constant gain; 1 "ramp": every lane keeps one stage, straight-line
// with the short exact division; 2 general.  TAME: see fast_frame.
// Oscillator half of a frame: advance both phases, wrap, fetch the two table samples.
template <bool TAB_LDS, bool TAME, int INTERP>
__device__ __forceinline__ v2f fast2_osc(Fast2Regs &r, const char *lds_tab, const char *__restrict__ glb_tab) {
  const v2f ph0 = r.phase + r.inc;
  const v2f x = ph0 - r.lo;
  const v2f phw = r.lo + (x - r.span);
  v2f ph;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const bool over = ph0[c] >= r.hi[c];
    float p;
    if (TAME) {
      p = over ? phw[c] : ph0[c];
    } else {
      const bool simple = over && (x[c] < r.span2[c]);
      const bool in_range = (ph0[c] >= r.lo[c]) && !over;
      p = simple ? phw[c] : ph0[c];
      if (!(in_range || simple)) p = slow_wrap(ph0[c], r.lo[c], r.hi[c], r.span[c]);
    }
    ph[c] = p;
  }
  r.phase = ph;
  v2f s;
  s.x = fast2_fetch<TAB_LDS, INTERP, TAME>(lds_tab, glb_tab, r.toff4[0], r.tsize_m1[0], r.lo.x, r.hi.x, ph.x);
  s.y = fast2_fetch<TAB_LDS, INTERP, TAME>(lds_tab, glb_tab, r.toff4[1], r.tsize_m1[1], r.lo.y, r.hi.y, ph.y);
  return s;
}

// The rest of the frame: biquad, envelope/gain, smoother, pan, lane-local sum of the two voices.
// EM (envelope mode): 0 every lane has a constant gain; 1 "ramp": every lane keeps one stage, straight-line
// with the short exact division; 2 general.  TAME: see fast_frame.
template <bool FILTER, int EM, bool TAME>
__device__ __forceinline__ void fast2_post(Fast2Regs &r, Env2Regs &e, v2f s, v2f &xn, v2f &xo, v2f &yn, v2f &yo,
                                           const bool rel0, const bool rel1, const bool silent0,
                                           const bool silent1, float &out_l, float &out_r) {
  // ---- biquad, packed ----
  if (FILTER) {
    v2f y = r.b0 * s;
    y = y + r.b1 * xn;
    y = y + r.b2 * xo;
    y = y - r.a1 * yn;
    y = y - r.a2 * yo;
    xo = s;
    yo = y;
    s = y;
  }
  // ---- gain ----
  v2f gain;
  if (EM == 0) {
    gain = r.gain_const;
  } else if (EM == 1) {
    e.clk = e.clk + 1.0f;                              // exact: clocks stay below 2^24 in this mode
    const v2f num = e.clk - e.ebase;                   // t | t - attack_time | t_release
    // q = num / den correctly rounded: the FMA tail of the IEEE fp32 division expansion hipcc itself emits
    // (v_div_scale / v_rcp / fma.. / v_div_fmas / v_div_fixup) with the refined reciprocal prepared once per
    // span; fast2_env_span admits only denominators for which div_scale / div_fixup would not intervene.
    v2f q = num * e.erinv;
    v2f rem = __builtin_elementwise_fma(-e.eden, q, num);
    q = __builtin_elementwise_fma(rem, e.erinv, q);
    rem = __builtin_elementwise_fma(-e.eden, q, num);
    q = __builtin_elementwise_fma(rem, e.erinv, q);
    const v2f lvl = e.eC * (e.eA + e.eB * q);
    gain = e.ampv * (lvl * e.velv);
  } else {
    gain.x = fast2_env_general(r, e, 0, rel0);
    gain.y = fast2_env_general(r, e, 1, rel1);
  }
  r.sgain = r.sgain + r.k * (gain - r.sgain);
  s = s * r.sgain;
  r.sample = s;
  // ---- pan, lane-local sum of the two voices ----
  v2f so = s;
  if (!TAME) {          // TAME loops run only when no live lane is muted (dead lanes already yield exact zeros)
    so.x = silent0 ? 0.0f : s.x;
    so.y = silent1 ? 0.0f : s.y;
  }
  const v2f l2 = so * r.pan_l;
  const v2f r2 = so * r.pan_r;
  out_l = l2.x + l2.y;
  out_r = r2.x + r2.y;
}

template <bool TAB_LDS, bool FILTER, int EM, bool TAME, int INTERP>
__device__ __forceinline__ void fast2_frame(Fast2Regs &r, Env2Regs &e, v2f &xn, v2f &xo, v2f &yn, v2f &yo,
                                            const bool rel0, const bool rel1, const bool silent0,
                                            const bool silent1, const char *lds_tab,
                                            const char *__restrict__ glb_tab, float &out_l, float &out_r) {
  const v2f s = fast2_osc<TAB_LDS, TAME, INTERP>(r, lds_tab, glb_tab);
  fast2_post<FILTER, EM, TAME>(r, e, s, xn, xo, yn, yo, rel0, rel1, silent0, silent1, out_l, out_r);
}

#define SK_F2_ARGS released[0], released[1], silent[0], silent[1], lds_tab, glb_tab
// one frame (J) / two frames (J, J+1; delay-line roles swap in between, one 4-chain DPP reduction)
#define SK_FAST2_ONE(J, EM_, TAME_)                                                                      \
  {                                                                                                      \
    float l, rr;                                                                                         \
    fast2_frame<TAB_LDS, FILTER, EM_, TAME_, INTERP>(r, e, r.x1, r.x2, r.y1, r.y2, SK_F2_ARGS, l, rr);   \
    SK_REDUCE_AND_STORE(J)                                                                               \
    { v2f t_ = r.x1; r.x1 = r.x2; r.x2 = t_; t_ = r.y1; r.y1 = r.y2; r.y2 = t_; }                        \
  }
#define SK_FAST2_PAIR(J, EM_, TAME_)                                                                     \
  {                                                                                                      \
    float l0, r0, l1, r1;                                                                                \
    fast2_frame<TAB_LDS, FILTER, EM_, TAME_, INTERP>(r, e, r.x1, r.x2, r.y1, r.y2, SK_F2_ARGS, l0, r0);  \
    fast2_frame<TAB_LDS, FILTER, EM_, TAME_, INTERP>(r, e, r.x2, r.x1, r.y2, r.y1, SK_F2_ARGS, l1, r1);  \
    SK_REDUCE4_AND_STORE(J)                                                                              \
  }
// Eight frames (J..J+7) with the cross-lane sum through LDS instead of the VALU: every lane parks its (L,R) of
// 8 frames in a wave-private transposition tile xp[8][65] (one ds_write_b64 per frame, row stride 65 keeps the
// column reads conflict-free); then lane (f = lane&7, seg = lane>>3) adds the 8 lanes of segment seg for frame f,
// the 8 segment sums go through xq[8][8], and lanes 0..7 finish one frame each.  ~1.75 VALU + 3.25 LDS
// instructions per frame instead of 12 VALU.  All traffic stays inside one wavefront (LDS executes a wave's
// accesses in order): no s_barrier.
#define SK_WAVE_SYNC()                                      \
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
  __builtin_amdgcn_wave_barrier();                          \
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#define SK_FAST2_LDS_BLOCK(J, EM_)                                                                       \
  {                                                                                                      \
    /* software pipeline: the table gather of the NEXT frame is issued before the biquad/gain chain of the   \
       current one (the source order matters: the compiler may not move an LDS read above the tile write) */  \
    v2f s0_ = fast2_osc<TAB_LDS, true, INTERP>(r, lds_tab, glb_tab);                                      \
    _Pragma("unroll") for (int q_ = 0; q_ < 8; q_ += 2) {                                                \
      float l0, r0, l1, r1;                                                                              \
      const v2f s1_ = fast2_osc<TAB_LDS, true, INTERP>(r, lds_tab, glb_tab);                              \
      fast2_post<FILTER, EM_, true>(r, e, s0_, r.x1, r.x2, r.y1, r.y2, released[0], released[1], silent[0], silent[1], l0, r0); \
      if (q_ < 6) s0_ = fast2_osc<TAB_LDS, true, INTERP>(r, lds_tab, glb_tab);                            \
      fast2_post<FILTER, EM_, true>(r, e, s1_, r.x2, r.x1, r.y2, r.y1, released[0], released[1], silent[0], silent[1], l1, r1); \
      xp[q_ * 65 + lane] = make_float2(l0, r0);                                                          \
      xp[(q_ + 1) * 65 + lane] = make_float2(l1, r1);                                                    \
    }                                                                                                    \
    SK_WAVE_SYNC()                                                                                       \
    {                                                                                                    \
      const float2 *src_ = xp + (lane & 7) * 65 + (lane >> 3) * 8;                                       \
      float2 a0_ = src_[0];                                                                              \
      _Pragma("unroll") for (int i_ = 1; i_ < 8; ++i_) { const float2 t_ = src_[i_]; a0_.x += t_.x; a0_.y += t_.y; } \
      xq[lane] = a0_; /* == xq[seg * 8 + f] */                                                           \
    }                                                                                                    \
    SK_WAVE_SYNC()                                                                                       \
    if (lane < 8) {                                                                                      \
      float2 t0_ = xq[lane];                                                                             \
      _Pragma("unroll") for (int g_ = 1; g_ < 8; ++g_) { const float2 t_ = xq[g_ * 8 + lane]; t0_.x += t_.x; t0_.y += t_.y; } \
      wsum[wave * SK_CHUNK + (J) + lane] = t0_;                                                          \
    }                                                                                                    \
    SK_WAVE_SYNC()                                                                                       \
  }
// a whole chunk of cn frames in mode EM_: LDS blocks of 8 when tame, DPP pairs otherwise, single-frame tail
#if SK_LDS_REDUCE
#define SK_FAST2_CHUNK(EM_)                                                     \
  {                                                                             \
    int j = 0;                                                                  \
    if (tame) { for (; j + 8 <= cn; j += 8) SK_FAST2_LDS_BLOCK(j, EM_)          \
                for (; j + 1 < cn; j += 2) SK_FAST2_PAIR(j, EM_, true)          \
                if (j < cn) SK_FAST2_ONE(j, EM_, true) }                        \
    else      { for (; j + 1 < cn; j += 2) SK_FAST2_PAIR(j, EM_, false)         \
                if (j < cn) SK_FAST2_ONE(j, EM_, false) }                       \
  }
#else
#define SK_FAST2_CHUNK(EM_)                                                     \
  {                                                                             \
    int j = 0;                                                                  \
    if (tame) { for (; j + 1 < cn; j += 2) SK_FAST2_PAIR(j, EM_, true)          \
                if (j < cn) SK_FAST2_ONE(j, EM_, true) }                        \
    else      { for (; j + 1 < cn; j += 2) SK_FAST2_PAIR(j, EM_, false)         \
                if (j < cn) SK_FAST2_ONE(j, EM_, false) }                       \
  }
#endif
// wave sums of the chunk -> this workgroup's partial-mix row (ACCUM_: add to what is there)
#define SK_FAST2_FLUSH(ACCUM_)                                                   \
  __syncthreads();                                                              \
  if (tid < 2 * cn) {                                                           \
    const float *w_ = reinterpret_cast<const float *>(wsum);                    \
    float s_ = w_[0 * 2 * SK_CHUNK + tid];                                      \
    s_ += w_[1 * 2 * SK_CHUNK + tid];                                           \
    s_ += w_[2 * 2 * SK_CHUNK + tid];                                           \
    s_ += w_[3 * 2 * SK_CHUNK + tid];                                           \
    float *p_ = a.partial + part_base + (size_t)c0 * 2 + tid;                   \
    if (ACCUM_) *p_ += s_; else *p_ = s_;                                       \
  }                                                                             \
  __syncthreads();

// load the two voices of this lane for workgroup pass g; returns whether the wave is tame
template <bool FILTER, bool ENV>
__device__ __forceinline__ bool fast2_load(const sk_render_args_t &a, int g, int wave, int lane, Fast2Regs &r,
                                           Env2Regs &e, bool dead[2], bool silent[2], bool released[2],
                                           uint64_t t_start[2], uint64_t t_release[2], int vidx[2]) {
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int v = g * (2 * SK_GROUP) + wave * 128 + c * 64 + lane;
    vidx[c] = v;
    const uint4 osc = *reinterpret_cast<const uint4 *>(&a.ro[SKP_OSC][v]);
    const uint4 tab = *reinterpret_cast<const uint4 *>(&a.ro[SKP_TAB][v]);
    const uint4 gn = *reinterpret_cast<const uint4 *>(&a.ro[SKP_GAIN][v]);
    const uint4 s0 = *reinterpret_cast<const uint4 *>(&a.rw[SKS_OSC][v]);
    const uint4 s1 = *reinterpret_cast<const uint4 *>(&a.rw[SKS_FILT][v]);
    const uint4 s2 = *reinterpret_cast<const uint4 *>(&a.rw[SKS_MISC][v]);
    r.inc[c] = __uint_as_float(osc.x); r.lo[c] = __uint_as_float(osc.y);
    r.hi[c] = __uint_as_float(osc.z);
    e.ampv[c] = __uint_as_float(osc.w);
    r.toff4[c] = (int)tab.x << 2; r.tsize_m1[c] = (int)tab.y - 1;
    const uint32_t flags = tab.z;
    e.velv[c] = __uint_as_float(gn.x); r.k[c] = __uint_as_float(gn.y);
    r.b0[c] = __uint_as_float(gn.z);   r.b1[c] = __uint_as_float(gn.w);
    r.phase[c] = __uint_as_float(s0.x); r.sgain[c] = __uint_as_float(s0.y);
    r.x1[c] = __uint_as_float(s0.z);    r.x2[c] = __uint_as_float(s0.w);
    r.y1[c] = __uint_as_float(s1.x);    r.y2[c] = __uint_as_float(s1.y);
    r.sample[c] = __uint_as_float(s1.z); r.rw[c] = s1.w;
    r.pan_l[c] = __uint_as_float(s2.z); r.pan_r[c] = __uint_as_float(s2.w);
    r.b2[c] = 0.0f; r.a1[c] = 0.0f; r.a2[c] = 0.0f;
    if (FILTER) {
      const uint4 fl = *reinterpret_cast<const uint4 *>(&a.ro[SKP_FILT][v]);
      r.b2[c] = __uint_as_float(fl.x); r.a1[c] = __uint_as_float(fl.y); r.a2[c] = __uint_as_float(fl.z);
    }
    released[c] = false; t_start[c] = 0; t_release[c] = 0;
    r.gain_const[c] = e.ampv[c];                       // no envelope: amp * 1.0f * 1.0f
    e.att[c] = e.attdec[c] = e.dec[c] = e.rel[c] = 0.0f;
    e.susv[c] = 0.0f; e.omsv[c] = 0.0f; e.tf[c] = 0.0f; e.trf[c] = 0.0f;
    if (ENV) {
      const uint4 et = *reinterpret_cast<const uint4 *>(&a.ro[SKP_ENV_T][v]);
      const uint4 es = *reinterpret_cast<const uint4 *>(&a.ro[SKP_ENV_S][v]);
      e.att[c] = __uint_as_float(et.x); e.dec[c] = __uint_as_float(et.y);
      e.susv[c] = __uint_as_float(et.z); e.rel[c] = __uint_as_float(et.w);
      e.attdec[c] = e.att[c] + e.dec[c];               // synth.c:410: decay_start + decay_time
      e.omsv[c] = 1.0f - e.susv[c];                    // synth.c:413
      t_start[c] = ((uint64_t)es.y << 32) | es.x;
      t_release[c] = ((uint64_t)es.w << 32) | es.z;
      released[c] = t_release[c] != 0;                 // synth.c:417
    }
    dead[c] = (r.rw[c] & SKR_FINISHED) || e.ampv[c] == 0.0f || (flags & SKF_INERT);
    silent[c] = dead[c] || (flags & SKF_MUTED);
    if (dead[c]) {   // never stored back: inert numbers -> exact zeros, table index 0 (see sk_render_fast_kernel)
      r.inc[c] = 0.0f; r.lo[c] = 0.0f; r.hi[c] = 1.0f; r.phase[c] = 0.0f;
      r.toff4[c] = 0; r.tsize_m1[c] = 0;
      r.k[c] = 0.0f; r.sgain[c] = 0.0f; e.ampv[c] = 0.0f; r.gain_const[c] = 0.0f;
      r.b0[c] = r.b1[c] = r.b2[c] = r.a1[c] = r.a2[c] = 0.0f;
      r.x1[c] = r.x2[c] = r.y1[c] = r.y2[c] = 0.0f;
      r.pan_l[c] = r.pan_r[c] = 0.0f; r.rw[c] &= ~SKR_ENV_ACTIVE;
    }
  }
  r.span = r.hi - r.lo;
  r.span2 = r.span + r.span;
  bool tame_lane = true;
#pragma unroll
  for (int c = 0; c < 2; ++c)
    tame_lane = tame_lane && !(silent[c] && !dead[c]) &&
                (dead[c] || (r.inc[c] >= 0.0f && r.inc[c] <= 0.5f * r.span[c] && r.phase[c] >= r.lo[c] &&
                             r.phase[c] <= r.hi[c] && r.lo[c] >= 0.0f && r.hi[c] <= (float)(r.tsize_m1[c] + 1)));
  return __all(tame_lane);
}

__device__ __forceinline__ void fast2_store(const sk_render_args_t &a, const Fast2Regs &r, const bool dead[2],
                                            const int vidx[2]) {
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int v = vidx[c];
    if (!dead[c]) {
      uint4 s0, s1;
      s0.x = __float_as_uint(r.phase[c]); s0.y = __float_as_uint(r.sgain[c]);
      s0.z = __float_as_uint(r.x1[c]);    s0.w = __float_as_uint(r.x2[c]);
      s1.x = __float_as_uint(r.y1[c]);    s1.y = __float_as_uint(r.y2[c]);
      s1.z = __float_as_uint(r.sample[c]); s1.w = r.rw[c];
      *reinterpret_cast<uint4 *>(&a.rw[SKS_OSC][v]) = s0;
      *reinterpret_cast<uint4 *>(&a.rw[SKS_FILT][v]) = s1;
    } else {
      reinterpret_cast<uint32_t *>(&a.rw[SKS_FILT][v])[2] = 0u;     // voice_sample = 0, synth.c:532,538
    }
  }
}

#define SK_FAST2_PROLOGUE()                                                                          \
  extern __shared__ float lds[];                                                                     \
  float2 *wsum = reinterpret_cast<float2 *>(lds + (TAB_LDS ? a.lds_table_floats : 0));               \
  const char *lds_tab = reinterpret_cast<const char *>(lds);                                         \
  const char *glb_tab = reinterpret_cast<const char *>(a.tables);                                    \
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;                                     \
  float2 *xp = wsum + 4 * SK_CHUNK + wave * (8 * 65 + 64);   /* wave-private: tile [8][65] then xq [64] */ \
  float2 *xq = xp + 8 * 65;                                                                          \
  (void)xp; (void)xq;                                                                                \
  if (TAB_LDS) {                                                                                     \
    const int n4 = a.lds_table_floats >> 2;                                                          \
    const float4 *src4 = reinterpret_cast<const float4 *>(a.tables);                                 \
    float4 *dst4 = reinterpret_cast<float4 *>(lds);                                                  \
    for (int i = tid; i < n4; i += SK_GROUP) dst4[i] = src4[i];                                      \
    __syncthreads();                                                                                 \
  }                                                                                                  \
  const size_t part_base = (size_t)blockIdx.x * (size_t)a.num_frames * 2;                            \
  const int n_groups2 = a.n_groups >> 1;   /* 512 voices per workgroup pass (host pads to 512) */

#ifndef SK_FAST2_MIN_WAVES
#define SK_FAST2_MIN_WAVES 4     /* <= 128 VGPRs */
#endif

// Constant-level groups.  ENV: the bank uses envelopes, so every group is classified first.
template <bool TAB_LDS, bool FILTER, bool ENV, int INTERP>
__global__ __launch_bounds__(SK_GROUP, SK_FAST2_MIN_WAVES) void sk_render_fast2_kernel(const sk_render_args_t a) {
  SK_FAST2_PROLOGUE()
  bool first_pass = true;
  for (int g = blockIdx.x; g < n_groups2; g += gridDim.x) {
    Fast2Regs r;
    Env2Regs e;
    bool dead[2], silent[2], released[2];
    uint64_t t_start[2], t_release[2];
    int vidx[2];
    const bool tame = fast2_load<FILTER, ENV>(a, g, wave, lane, r, e, dead, silent, released, t_start, t_release, vidx);
    if (ENV) {
      // constant envelope level on the first frame of the launch <=> for the whole launch (absorbing codes)
      bool ok = true;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const uint64_t d_on = a.count0 + 1 - t_start[c], d_off = a.count0 + 1 - t_release[c];
        const int code = env_stage_code((r.rw[c] & SKR_ENV_ACTIVE) != 0, released[c], (float)d_on, (float)d_off,
                                        e.att[c], e.attdec[c], e.rel[c]);
        ok = ok && (dead[c] || code == 0 || code == 3 || code == 5);
        const float level = code == 3 ? e.susv[c] : 0.0f;
        r.gain_const[c] = e.ampv[c] * (level * e.velv[c]);                       // synth.c:582,588
        if (!dead[c] && code == 5) r.rw[c] &= ~SKR_ENV_ACTIVE;                   // synth.c:429
      }
      const int group_ok = __syncthreads_and(ok ? 1 : 0);
      if (tid == 0) a.group_flag[g] = group_ok ? 0 : 1;
      if (!group_ok) continue;                          // sk_render_env2_kernel renders this group
    }
    for (int c0 = 0; c0 < a.num_frames; c0 += SK_CHUNK) {
      const int cn = min(SK_CHUNK, a.num_frames - c0);
      SK_FAST2_CHUNK(0)
      SK_FAST2_FLUSH(!first_pass)
    }
    fast2_store(a, r, dead, vidx);
    first_pass = false;
  }
  if (first_pass) {   // every group of this workgroup was deferred: its partial-mix row must still exist
    for (int i = tid; i < 2 * a.num_frames; i += SK_GROUP) a.partial[part_base + i] = 0.0f;
  }
}

// Groups with envelopes in motion (flagged by sk_render_fast2_kernel, which ran just before on the stream).
#ifndef SK_ENV2_MIN_WAVES
#define SK_ENV2_MIN_WAVES 3      /* the envelope machinery wants ~170 VGPRs: 3 waves per SIMD measured best (2: no spills, 4: 220 B of scratch) */
#endif
template <bool TAB_LDS, bool FILTER, int INTERP>
__global__ __launch_bounds__(SK_GROUP, SK_ENV2_MIN_WAVES) void sk_render_env2_kernel(const sk_render_args_t a) {
  SK_FAST2_PROLOGUE()
  for (int g = blockIdx.x; g < n_groups2; g += gridDim.x) {
    if (a.group_flag[g] == 0) continue;
    Fast2Regs r;
    Env2Regs e;
    bool dead[2], silent[2], released[2];
    uint64_t t_start[2], t_release[2];
    int vidx[2];
    const bool tame = fast2_load<FILTER, true>(a, g, wave, lane, r, e, dead, silent, released, t_start, t_release, vidx);
    bool all_const_from_here = false;
    for (int c0 = 0; c0 < a.num_frames; c0 += SK_CHUNK) {
      const int cn = min(SK_CHUNK, a.num_frames - c0);
      bool steady = true, exact = true, ramp = false;
      float cb_tf[2] = {0.0f, 0.0f}, cb_trf[2] = {0.0f, 0.0f};     // clocks of the frame BEFORE the chunk
      if (!all_const_from_here) {
        const uint64_t base = a.count0 + (uint64_t)c0;              // frame c0+j has now = base + j + 1 (synth.c:521)
        const uint64_t lim = (1ull << 24) - (uint64_t)SK_CHUNK - 2; // x + 1.0f stays exact below 2^24
        bool ex = true, st = true, same = true;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const uint64_t d_on = base - t_start[c], d_off = base - t_release[c];
          ex = ex && (dead[c] || ((d_on < lim) && (!released[c] || d_off < lim)));
          cb_tf[c] = (float)d_on;
          cb_trf[c] = released[c] ? (float)d_off : 0.0f;
          fast2_env_span(r, e, c, dead[c], released[c], (float)(d_on + 1), (float)(d_off + 1),
                         (float)(d_on + (uint64_t)cn), (float)(d_off + (uint64_t)cn), cb_tf[c], cb_trf[c], st, same);
        }
        exact = __all(ex);
        steady = __all(st);
        ramp = !steady && exact && __all(same);
        all_const_from_here = steady;       // constant levels are absorbing within a launch
      }
      if (steady) {
        SK_FAST2_CHUNK(0)
      } else if (ramp) {
        SK_FAST2_CHUNK(1)
      } else if (exact && tame && !(cn & 7)) {
        // some lane changes stage inside this chunk: re-decide per 8-frame block (clocks are exact floats here)
        for (int jb = 0; jb < cn; jb += 8) {
          bool st = true, same = true;
          const float fb = (float)jb;
#pragma unroll
          for (int c = 0; c < 2; ++c)
            fast2_env_span(r, e, c, dead[c], released[c], cb_tf[c] + fb + 1.0f, cb_trf[c] + fb + 1.0f,
                           cb_tf[c] + fb + 8.0f, cb_trf[c] + fb + 8.0f, cb_tf[c] + fb, cb_trf[c] + fb, st, same);
          const bool b_const = __all(st), b_ramp = __all(same);
#if SK_LDS_REDUCE
          if (b_const) SK_FAST2_LDS_BLOCK(jb, 0)
          else if (b_ramp) SK_FAST2_LDS_BLOCK(jb, 1)
          else
#endif
          {
#pragma unroll
            for (int c = 0; c < 2; ++c) { e.tf[c] = cb_tf[c] + fb; e.trf[c] = cb_trf[c] + fb; }
            for (int q = 0; q < 8; q += 2) {
              float l0, r0, l1, r1;
              e.tf[0] += 1.0f; e.trf[0] += 1.0f; e.tf[1] += 1.0f; e.trf[1] += 1.0f;
              fast2_frame<TAB_LDS, FILTER, 2, true, INTERP>(r, e, r.x1, r.x2, r.y1, r.y2, SK_F2_ARGS, l0, r0);
              e.tf[0] += 1.0f; e.trf[0] += 1.0f; e.tf[1] += 1.0f; e.trf[1] += 1.0f;
              fast2_frame<TAB_LDS, FILTER, 2, true, INTERP>(r, e, r.x2, r.x1, r.y2, r.y1, SK_F2_ARGS, l1, r1);
              { const int J_ = jb + q; wave_sum4_to_lane63(l0, r0, l1, r1);
                if (lane == 63) *reinterpret_cast<float4 *>(&wsum[wave * SK_CHUNK + J_]) = make_float4(l0, r0, l1, r1); }
            }
          }
        }
      } else {
        // clocks past 2^24 frames, ragged chunk lengths, untame waves: integer clocks, one frame at a time
        for (int j = 0; j < cn; ++j) {
          const uint64_t now = a.count0 + (uint64_t)(c0 + j) + 1;
          e.tf[0] = (float)(now - t_start[0]); e.trf[0] = (float)(now - t_release[0]);
          e.tf[1] = (float)(now - t_start[1]); e.trf[1] = (float)(now - t_release[1]);
          SK_FAST2_ONE(j, 2, false)
        }
      }
      SK_FAST2_FLUSH(true)
    }
    fast2_store(a, r, dead, vidx);
  }
}

// ---------------------------------------------------------------- partial-mix reduction

// Stage 1: partial[W][ncols] -> tmp[SK_RED_SLABS][ncols]; stage 2: tmp -> out[ncols].
// Rows are added in a fixed order in both stages (bit-reproducible, no atomics).
#define SK_RED_SLABS 16
__global__ __launch_bounds__(256) void sk_reduce_kernel(const float *__restrict__ partial,
                                                        float *__restrict__ out, int W, int ncols) {
  __shared__ float part[4][64];
  const int c = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + c;
  const int slabs = gridDim.y;
  const int w0 = (int)((long long)W * blockIdx.y / slabs), w1 = (int)((long long)W * (blockIdx.y + 1) / slabs);
  float s = 0.0f;
  if (col < ncols)
    for (int w = w0 + slice; w < w1; w += 4) s += partial[(size_t)w * ncols + col];
  part[slice][c] = s;
  __syncthreads();
  if (slice == 0 && col < ncols)
    out[(size_t)blockIdx.y * ncols + col] = ((part[0][c] + part[1][c]) + part[2][c]) + part[3][c];
}

// ---------------------------------------------------------------- master volume

// synth.c:616-624.  The gain is a serial one-pole recurrence over frames, so one lane walks it
// (bit parity forbids a parallel scan); the other lanes then scale and interleave.
// gain_state[0] is the smoother state carried between launches.
#define SK_MASTER_TILE 1024
__global__ __launch_bounds__(256) void sk_master_kernel(const float *__restrict__ sum,
                                                        float *__restrict__ out, int num_frames,
                                                        int num_channels, float target, float k,
                                                        float *gain_state) {
  __shared__ float gains[SK_MASTER_TILE];
  __shared__ float carry;
  if (threadIdx.x == 0) carry = gain_state[0];
  __syncthreads();
  for (int f0 = 0; f0 < num_frames; f0 += SK_MASTER_TILE) {
    const int n = min(SK_MASTER_TILE, num_frames - f0);
    if (threadIdx.x == 0) {
      float vg = carry;
      for (int i = 0; i < n; ++i) {
        vg += k * (target - vg);
        gains[i] = vg;
      }
      carry = vg;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) {
      const float vg = gains[i];
      const float2 s = reinterpret_cast<const float2 *>(sum)[f0 + i];
      out[(size_t)(f0 + i) * num_channels + 0] = s.x * vg;
      out[(size_t)(f0 + i) * num_channels + 1] = s.y * vg;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) gain_state[0] = carry;
}

// ---------------------------------------------------------------- launchers (C linkage)

extern "C" int sk_launch_render(const sk_render_args_t *args, int n_workgroups, hipStream_t stream) {
  const bool tab_lds = args->lds_table_floats > 0;
  const bool stems = args->stems != nullptr;
  size_t lds_bytes = (size_t)(tab_lds ? args->lds_table_floats : 0) * sizeof(float) +
                     (size_t)4 * SK_CHUNK * sizeof(float2);
  if (args->fast_mode & SKM_TWO_PER_LANE) lds_bytes += (size_t)4 * (8 * 65 + 64) * sizeof(float2);
  dim3 grid((unsigned)n_workgroups), block(SK_GROUP);
  if ((args->fast_mode & SKM_FAST) && (args->fast_mode & SKM_TWO_PER_LANE) && !stems) {
    const int key = (tab_lds ? 8 : 0) | ((args->fast_mode & SKM_FILTER_ALL) ? 4 : 0) |
                    ((args->fast_mode & SKM_ENV_ALL) ? 2 : 0) | (args->interp == 1 ? 1 : 0);
#define SK_FAST2_CASE(K, T, F, E, I)                                                                     \
  case K:                                                                                                \
    hipLaunchKernelGGL((sk_render_fast2_kernel<T, F, E, I>), grid, block, lds_bytes, stream, *args);     \
    if (E) hipLaunchKernelGGL((sk_render_env2_kernel<T, F, I>), grid, block, lds_bytes, stream, *args);  \
    break;
    switch (key) {
      SK_FAST2_CASE(0, false, false, false, 0) SK_FAST2_CASE(1, false, false, false, 1)
      SK_FAST2_CASE(2, false, false, true, 0)  SK_FAST2_CASE(3, false, false, true, 1)
      SK_FAST2_CASE(4, false, true, false, 0)  SK_FAST2_CASE(5, false, true, false, 1)
      SK_FAST2_CASE(6, false, true, true, 0)   SK_FAST2_CASE(7, false, true, true, 1)
      SK_FAST2_CASE(8, true, false, false, 0)  SK_FAST2_CASE(9, true, false, false, 1)
      SK_FAST2_CASE(10, true, false, true, 0)  SK_FAST2_CASE(11, true, false, true, 1)
      SK_FAST2_CASE(12, true, true, false, 0)  SK_FAST2_CASE(13, true, true, false, 1)
      SK_FAST2_CASE(14, true, true, true, 0)   SK_FAST2_CASE(15, true, true, true, 1)
    }
#undef SK_FAST2_CASE
    return (int)hipGetLastError();
  }
  if ((args->fast_mode & SKM_FAST) && !stems) {
    // clean bank: specialised kernel (table residency x filter x envelope x interpolation)
    const int key = (tab_lds ? 8 : 0) | ((args->fast_mode & SKM_FILTER_ALL) ? 4 : 0) |
                    ((args->fast_mode & SKM_ENV_ALL) ? 2 : 0) | (args->interp == 1 ? 1 : 0);
#define SK_FAST_CASE(K, T, F, E, I)                                                                      \
  case K: hipLaunchKernelGGL((sk_render_fast_kernel<T, F, E, I>), grid, block, lds_bytes, stream, *args); break;
    switch (key) {
      SK_FAST_CASE(0, false, false, false, 0) SK_FAST_CASE(1, false, false, false, 1)
      SK_FAST_CASE(2, false, false, true, 0)  SK_FAST_CASE(3, false, false, true, 1)
      SK_FAST_CASE(4, false, true, false, 0)  SK_FAST_CASE(5, false, true, false, 1)
      SK_FAST_CASE(6, false, true, true, 0)   SK_FAST_CASE(7, false, true, true, 1)
      SK_FAST_CASE(8, true, false, false, 0)  SK_FAST_CASE(9, true, false, false, 1)
      SK_FAST_CASE(10, true, false, true, 0)  SK_FAST_CASE(11, true, false, true, 1)
      SK_FAST_CASE(12, true, true, false, 0)  SK_FAST_CASE(13, true, true, false, 1)
      SK_FAST_CASE(14, true, true, true, 0)   SK_FAST_CASE(15, true, true, true, 1)
    }
#undef SK_FAST_CASE
    return (int)hipGetLastError();
  }
  if (tab_lds) {
    if (stems) hipLaunchKernelGGL((sk_render_kernel<true, true>), grid, block, lds_bytes, stream, *args);
    else       hipLaunchKernelGGL((sk_render_kernel<true, false>), grid, block, lds_bytes, stream, *args);
  } else {
    if (stems) hipLaunchKernelGGL((sk_render_kernel<false, true>), grid, block, lds_bytes, stream, *args);
    else       hipLaunchKernelGGL((sk_render_kernel<false, false>), grid, block, lds_bytes, stream, *args);
  }
  return (int)hipGetLastError();
}

extern "C" int sk_launch_render_mod(const sk_render_args_t *args, int n_groups64, const int *levels,
                                    int max_level, hipStream_t stream) {
  dim3 grid((unsigned)n_groups64), block(64);
  if (args->stems) hipLaunchKernelGGL((sk_render_mod_kernel<true>), grid, block, 0, stream, *args, levels, max_level);
  else             hipLaunchKernelGGL((sk_render_mod_kernel<false>), grid, block, 0, stream, *args, levels, max_level);
  return (int)hipGetLastError();
}

// `tmp` holds SK_RED_SLABS * ncols floats.
extern "C" int sk_launch_reduce(const float *partial, float *tmp, float *out, int W, int ncols,
                                hipStream_t stream) {
  const unsigned cols = (unsigned)((ncols + 63) / 64);
  if (W <= 4 * SK_RED_SLABS) {
    hipLaunchKernelGGL(sk_reduce_kernel, dim3(cols, 1), dim3(256), 0, stream, partial, out, W, ncols);
  } else {
    hipLaunchKernelGGL(sk_reduce_kernel, dim3(cols, SK_RED_SLABS), dim3(256), 0, stream, partial, tmp, W, ncols);
    hipLaunchKernelGGL(sk_reduce_kernel, dim3(cols, 1), dim3(256), 0, stream, tmp, out, SK_RED_SLABS, ncols);
  }
  return (int)hipGetLastError();
}

extern "C" int sk_reduce_tmp_floats(int ncols) { return SK_RED_SLABS * ncols; }

extern "C" int sk_launch_master(const float *sum, float *out, int num_frames, int num_channels,
                                float target, float k, float *gain_state, hipStream_t stream) {
  hipLaunchKernelGGL(sk_master_kernel, dim3(1), dim3(256), 0, stream, sum, out, num_frames,
                     num_channels, target, k, gain_state);
  return (int)hipGetLastError();
}
