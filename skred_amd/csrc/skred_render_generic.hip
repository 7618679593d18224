// skred_render_generic.hip -- the general render kernels (any bank; modulated banks) + the launch dispatcher.
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
//   * per frame the 64 lanes' L/R are folded into one register (v_permlane32_swap) and summed with a fixed-order
//     DPP butterfly (no LDS traffic, no atomics: results are bit-reproducible run to run); wave sums meet in LDS every
//     SK_CHUNK frames and leave as coalesced stores into a per-workgroup partial mix;
//   * no MFMA: this is gather + multiply-add along a serial recurrence, not a contraction.
//
// Arithmetic contract (must match oracle/cpu_ref.c bit for bit per voice): compiled with
// -ffp-contract=off (no FMA fusion), fp32 subnormals kept (hipcc default), IEEE-rounded
// divide (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt), exact fmod.
#include <hip/hip_runtime.h>
#include "skred_kernel_common.hpp"
#include "skred_launch.h"

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
  const int bid = (int)blockIdx.x - a.wg_shift;             // this workgroup's row of the partial mix; -1: the gain workgroup
  if (bid < 0) { sk_finish_block(a, bid, tid, SK_GROUP, reinterpret_cast<int *>(lds)); return; }

  if (TAB_LDS) {
    // stage the whole pool; float4 when aligned, coalesced
    const int n4 = a.lds_table_floats >> 2;
    const float4 *src4 = reinterpret_cast<const float4 *>(a.tables);
    float4 *dst4 = reinterpret_cast<float4 *>(lds_tab);
    sk_stage_tables<SK_GROUP>(src4, dst4, n4, tid);
    for (int i = (n4 << 2) + tid; i < a.lds_table_floats; i += SK_GROUP) lds_tab[i] = a.tables[i];
    __syncthreads();
  }

  const size_t part_base = (size_t)bid * (size_t)a.num_frames * 2;
  bool first_pass = true;

  for (int g = bid; g < a.n_groups; g += a.n_rows) {
    const int v = g * SK_GROUP + tid;
    const bool publish = a.finish && g + a.n_rows >= a.n_groups;   // the pass that completes this workgroup's row
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
        SK_REDUCE_AND_STORE(j)
      }
      __syncthreads();
      if (tid < 2 * cn) {
        const float *w = reinterpret_cast<const float *>(wsum);
        float s = w[0 * 2 * SK_CHUNK + tid];
        s += w[1 * 2 * SK_CHUNK + tid];
        s += w[2 * 2 * SK_CHUNK + tid];
        s += w[3 * 2 * SK_CHUNK + tid];
        sk_row_store(a.partial + part_base + (size_t)c0 * 2 + tid, s, first_pass, publish);
      }
      __syncthreads();
    }
    store_voice(a, v, r);
    first_pass = false;
  }
  if (a.finish) sk_finish_block(a, bid, tid, SK_GROUP, reinterpret_cast<int *>(lds), true);
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
template <bool TAB_LDS>
__device__ __forceinline__ void voice_frame_mod(VoiceRegs &r, const ModRegs &m, int lane,
                                                const float *prev, float *cur, const float *incs,
                                                const float *lds_tab, const float *__restrict__ tab, uint64_t now,
                                                float white, int interp, float &out_l, float &out_r, const bool lag = false) {
  // (lag: the frame-lag form of sk_render_mod_kernel -- every modulator's sample this lane needs is the one the PREVIOUS
  // iteration left, whichever side of the lane it sits on)
  auto other = [&](int src) -> float { return (src < lane && !lag) ? cur[src] : prev[src]; };
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
      raw = table_fetch<TAB_LDS>(lds_tab, tab, r, pos, interp, !stops);
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

// Four wavefronts per workgroup, each rendering one aligned 64-voice group per pass; the workgroup strides over
// the bank (grid <= SK_MAX_WORKGROUPS) so that the table pool is staged into LDS once per workgroup and the
// partial mix has one row per workgroup.  Modulators live in their carrier's group (the host refuses anything
// else), so the per-frame exchange of voice_sample between dependency levels never leaves a wavefront: the
// arrays are wave-private and a wave barrier (LDS executes a wave's accesses in order) replaces the workgroup one.
#define SK_MOD_WAVE_SYNC()                                  \
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
  __builtin_amdgcn_wave_barrier();                          \
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
template <bool TAB_LDS, bool STEMS>
__global__ __launch_bounds__(SK_GROUP) void sk_render_mod_kernel(const sk_render_args_t a, const int *__restrict__ levels,
                                                                 int max_level) {
  extern __shared__ float lds[];
  const float *lds_tab = lds;
  float2 *wsum = reinterpret_cast<float2 *>(lds + (TAB_LDS ? a.lds_table_floats : 0));   // [4][SK_CHUNK]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float *xch0 = reinterpret_cast<float *>(wsum + 4 * SK_CHUNK) + wave * 192;               // wave-private: xch[2][64], incs[64]
  float *incs = xch0 + 128;
  const int bid = (int)blockIdx.x - a.wg_shift;        // row of the partial mix; -1: the gain workgroup (sk_finish_block)
  if (bid < 0) { sk_finish_block(a, bid, tid, SK_GROUP, reinterpret_cast<int *>(lds)); return; }
  if (TAB_LDS) {
    const int n4 = a.lds_table_floats >> 2;
    const float4 *src4 = reinterpret_cast<const float4 *>(a.tables);
    float4 *dst4 = reinterpret_cast<float4 *>(lds);
    sk_stage_tables<SK_GROUP>(src4, dst4, n4, tid);
    __syncthreads();
  }
  const size_t part_base = (size_t)bid * (size_t)a.num_frames * 2;
  // packed lanes (sparse banks; sk_render_args_t: pack_mask): a wave holds the voices that can sound, and the modulators they
  // name, of several 64-voice groups -- the exchange arrays are indexed by wave lane either way, a modulator's lane is translated
  // once per pass, and lanes keep the order of their voices, so "below the carrier" (same frame) stays "a lower lane"
  const bool packed = !STEMS && a.pack_shift < 6;
  const int n_pass = packed ? a.pack_passes : a.n_groups;   // passes of 256 voices = 4 groups of 64 (packed: 4 waves of several groups)
  bool first_pass = true;
  for (int g = bid; g < n_pass; g += a.n_rows) {
    int v = g * SK_GROUP + tid;
    uint64_t pmask = 0;
    int ppos = lane;
    bool absent = false;
    if (packed) { v = sk_packed_voice(a, g * (SK_GROUP / 64) + wave, lane, pmask, ppos); absent = v < 0; if (absent) v = 0; }
    const bool publish = a.finish && g + a.n_rows >= n_pass;       // the pass that completes this workgroup's row
    VoiceRegs r;
    load_voice(a, v, r);
    if (absent) r.flags |= SKF_INERT;                          // an empty lane: skipped on every frame, nothing stored
    ModRegs m;
    {
      const uint4 mi = *reinterpret_cast<const uint4 *>(&a.ro[SKP_MODI][v]);
      const uint4 mf = *reinterpret_cast<const uint4 *>(&a.ro[SKP_MODF][v]);
      const uint4 mx = *reinterpret_cast<const uint4 *>(&a.ro[SKP_MODX][v]);
      const uint4 fl = *reinterpret_cast<const uint4 *>(&a.ro[SKP_FILT][v]);
      m.fm = (int)mi.x; m.am = (int)mi.y; m.pm = (int)mi.z; m.cz = (int)mi.w;
      if (packed && !absent) {                                 // (the planes number modulators by their lane in the 64-voice group)
        if (m.fm >= 0) m.fm = sk_packed_lane(a, pmask, lane, m.fm);
        if (m.am >= 0) m.am = sk_packed_lane(a, pmask, lane, m.am);
        if (m.pm >= 0) m.pm = sk_packed_lane(a, pmask, lane, m.pm);
        if (m.cz >= 0) m.cz = sk_packed_lane(a, pmask, lane, m.cz);
      }
      m.fm_depth = __uint_as_float(mf.x); m.freq_scale = __uint_as_float(mf.y);
      m.am_depth = __uint_as_float(mf.z); m.pm_depth = __uint_as_float(mf.w);
      m.cz_depth = __uint_as_float(mx.x); m.cz_mode = (int)mx.y;
      m.cz_dist = __uint_as_float(fl.w);
      m.level = levels[v];
      m.inc_raw = (r.flags & SKF_REVERSE) ? -r.inc : r.inc;    // load_voice applied the direction sign
    }
    incs[lane] = m.inc_raw;
    xch0[lane] = r.sample;                                     // voice_sample[] as the last callback left it
    SK_MOD_WAVE_SYNC()

    uint64_t rng = a.rng0;
    int cur_i = 1;
    // FRAME-LAG form (round 4; 18.sk's shape: `v10 ... F0,70`, a modulator BELOW its carrier -- synth.c:548-555 in index order makes
    // that a same-frame dependency).  The level loop below renders every frame once per dependency level, each pass under the
    // EXEC mask of its level.  When the wave has exactly one level of same-frame dependencies and every edge fits -- a source
    // below its reader one level lower, a source above it on the reader's own level -- the level-1 lanes simply run ONE FRAME
    // BEHIND the level-0 lanes: in iteration i a level-0 lane renders frame i, a level-1 lane frame i - 1, and every sample a
    // lane needs from another is the one the previous iteration left in the exchange array.  One pass per frame for all lanes;
    // the level-0 lanes' (L, R) are held one iteration so that a frame's sum still takes all lanes' values of THAT frame; one
    // iteration more per launch (the first without the level-1 lanes, the last without the level-0 lanes).  Same arithmetic per
    // voice, same sums.  Waves that do not fit (deeper chains, stems) keep the level loop.
    bool lag = false;
    if (!STEMS && a.fm_skew && max_level >= 1 && a.num_frames >= 2) {
      xch0[64 + lane] = (float)m.level;                        // (the second exchange array is first written in iteration 0)
      SK_MOD_WAVE_SYNC()
      auto fits = [&](int s_) -> bool { return s_ < 0 || s_ == lane || (int)xch0[64 + (s_ & 63)] == (s_ < lane ? m.level - 1 : m.level); };
      const bool ok = m.level <= 1 && fits(m.fm) && fits(m.am) && fits(m.pm) && fits(m.cz);
      lag = __all(ok) && __any(m.level == 1);
      SK_MOD_WAVE_SYNC()
    }
    float hl = 0.0f, hr = 0.0f, white_prev = 0.0f;            // (lag) the level-0 lanes' (L, R) and the noise draw of the frame before
    if (lag) {                                                 // iteration 0: the level-0 lanes render frame 0
      rng = rng * LCG_A + LCG_C;
      const float white = (float)((int32_t)(uint32_t)(rng >> 32)) / 2147483648.0f;
      float *cur = xch0 + cur_i * 64;
      const float *prev = xch0 + (cur_i ^ 1) * 64;
      const bool live = !((r.rw & SKR_FINISHED) || r.amp == 0.0f || (r.flags & SKF_INERT));
      if (!live) r.sample = 0.0f;
      SK_MOD_WAVE_SYNC()
      float l = 0.0f, rr = 0.0f;
      if (live && m.level == 0) voice_frame_mod<TAB_LDS>(r, m, lane, prev, cur, incs, lds_tab, a.tables, a.count0 + 1, white, a.interp, l, rr, true);
      cur[lane] = r.sample;                                    // (a lane that sat out carries its sample along)
      SK_MOD_WAVE_SYNC()
      hl = l; hr = rr; white_prev = white;
      cur_i ^= 1;
    }
    for (int c0 = 0; c0 < a.num_frames; c0 += SK_CHUNK) {
      const int cn = min(SK_CHUNK, a.num_frames - c0);
      if (lag) {
        for (int j = 0; j < cn; ++j) {
          const int i = c0 + j + 1;                            // level-0 lanes: frame i (while there is one); level-1 lanes: frame i - 1
          const bool more = i < a.num_frames;
          float white = 0.0f;
          if (more) { rng = rng * LCG_A + LCG_C; white = (float)((int32_t)(uint32_t)(rng >> 32)) / 2147483648.0f; }
          float *cur = xch0 + cur_i * 64;
          const float *prev = xch0 + (cur_i ^ 1) * 64;
          const bool live = !((r.rw & SKR_FINISHED) || r.amp == 0.0f || (r.flags & SKF_INERT));
          if (!live) r.sample = 0.0f;
          SK_MOD_WAVE_SYNC()
          float l = 0.0f, rr = 0.0f;
          if (live && (m.level == 1 || more))
            voice_frame_mod<TAB_LDS>(r, m, lane, prev, cur, incs, lds_tab, a.tables, a.count0 + (uint64_t)(i - m.level) + 1,
                                     m.level == 0 ? white : white_prev, a.interp, l, rr, true);
          cur[lane] = r.sample;
          SK_MOD_WAVE_SYNC()
          if (m.level == 0) { const float tl = l, tr = rr; l = hl; rr = hr; hl = tl; hr = tr; }   // frame c0 + j: what was held
          SK_REDUCE_AND_STORE(j)
          white_prev = white;
          cur_i ^= 1;
        }
      } else
      for (int j = 0; j < cn; ++j) {
        const int i = c0 + j;
        const uint64_t now = a.count0 + (uint64_t)i + 1;
        rng = rng * LCG_A + LCG_C;
        const float white = (float)((int32_t)(uint32_t)(rng >> 32)) / 2147483648.0f;
        float *cur = xch0 + cur_i * 64;
        const float *prev = xch0 + (cur_i ^ 1) * 64;
        const bool live = !((r.rw & SKR_FINISHED) || r.amp == 0.0f || (r.flags & SKF_INERT));
        if (!live) { r.sample = 0.0f; cur[lane] = 0.0f; }         // synth.c:531-542
        SK_MOD_WAVE_SYNC()
        float l = 0.0f, rr = 0.0f;
        for (int lev = 0; lev <= max_level; ++lev) {
          if (live && m.level == lev) {
            voice_frame_mod<TAB_LDS>(r, m, lane, prev, cur, incs, lds_tab, a.tables, now, white, a.interp, l, rr);
            cur[lane] = r.sample;
          }
          SK_MOD_WAVE_SYNC()
        }
        if (STEMS) {
          if (v < a.n_voices)
            reinterpret_cast<float2 *>(a.stems)[(size_t)i * (size_t)a.n_voices + (size_t)v] = make_float2(l, rr);
        }
        SK_REDUCE_AND_STORE(j)
        cur_i ^= 1;
      }
      __syncthreads();
      if (tid < 2 * cn) {
        const float *w = reinterpret_cast<const float *>(wsum);
        float s = w[0 * 2 * SK_CHUNK + tid];
        s += w[1 * 2 * SK_CHUNK + tid];
        s += w[2 * 2 * SK_CHUNK + tid];
        s += w[3 * 2 * SK_CHUNK + tid];
        sk_row_store(a.partial + part_base + (size_t)c0 * 2 + tid, s, first_pass, publish);
      }
      __syncthreads();
    }
    if (!absent) store_voice(a, v, r);
    first_pass = false;
  }
  if (a.finish) sk_finish_block(a, bid, tid, SK_GROUP, reinterpret_cast<int *>(lds), true);
}
// ---------------------------------------------------------------- launchers (C linkage)

extern "C" int sk_launch_render(const sk_render_args_t *args, int n_workgroups, hipStream_t stream) {
  const bool tab_lds = args->lds_table_floats > 0;
  const bool stems = args->stems != nullptr;
  const size_t lds_bytes = (size_t)(tab_lds ? args->lds_table_floats : 0) * sizeof(float) +
                           (size_t)4 * SK_CHUNK * sizeof(float2);
  // clean banks (skred_bank.c:classify) have specialised kernels; with stems the one-voice kernel (or the generic one)
  if ((args->fast_mode & SKM_FAST) && (args->fast_mode & SKM_SPLIT) && !stems && tab_lds && !(args->fast_mode & SKM_TWO_PER_LANE))
    return sk_launch_render_split(args, n_workgroups, (args->fast_mode & SKM_SPLIT2) ? 2 : 4, stream);
  if ((args->fast_mode & SKM_FAST) && (!stems || !(args->fast_mode & SKM_TWO_PER_LANE)))   // (stems: the one-voice kernel has them)
    return (args->fast_mode & SKM_TWO_PER_LANE) ? sk_launch_render_fast2(args, n_workgroups, lds_bytes, stream)
                                                : sk_launch_render_fast(args, n_workgroups, lds_bytes, stream);
  dim3 grid((unsigned)(n_workgroups + args->wg_shift)), block(SK_GROUP);
  if (tab_lds) {
    if (stems) hipLaunchKernelGGL((sk_render_kernel<true, true>), grid, block, lds_bytes, stream, *args);
    else       hipLaunchKernelGGL((sk_render_kernel<true, false>), grid, block, lds_bytes, stream, *args);
  } else {
    if (stems) hipLaunchKernelGGL((sk_render_kernel<false, true>), grid, block, lds_bytes, stream, *args);
    else       hipLaunchKernelGGL((sk_render_kernel<false, false>), grid, block, lds_bytes, stream, *args);
  }
  return (int)hipGetLastError();
}

extern "C" int sk_launch_render_mod(const sk_render_args_t *args, int n_workgroups, const int *levels,
                                    int max_level, hipStream_t stream) {
  const bool tab_lds = args->lds_table_floats > 0;
  const size_t lds_bytes = (size_t)(tab_lds ? args->lds_table_floats : 0) * sizeof(float) +
                           (size_t)4 * SK_CHUNK * sizeof(float2) + (size_t)4 * 192 * sizeof(float);
  dim3 grid((unsigned)(n_workgroups + args->wg_shift)), block(SK_GROUP);
#define SK_MOD_LAUNCH(T, S) hipLaunchKernelGGL((sk_render_mod_kernel<T, S>), grid, block, lds_bytes, stream, *args, levels, max_level)
  if (tab_lds) { if (args->stems) SK_MOD_LAUNCH(true, true); else SK_MOD_LAUNCH(true, false); }
  else         { if (args->stems) SK_MOD_LAUNCH(false, true); else SK_MOD_LAUNCH(false, false); }
#undef SK_MOD_LAUNCH
  return (int)hipGetLastError();
}
