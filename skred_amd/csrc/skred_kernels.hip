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
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v += dpp_take<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
  v += dpp_take<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
  v += dpp_take<0x141, 0xF>(v);   // row_half_mirror
  v += dpp_take<0x140, 0xF>(v);   // row_mirror           -> every lane holds its 16-lane row sum
  v += dpp_take<0x142, 0xA>(v);   // row_bcast:15 into rows 1,3
  v += dpp_take<0x143, 0xC>(v);   // row_bcast:31 into rows 2,3 -> lane 63 holds the wave sum
  return v;
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
        const float sl = wave_sum_to_lane63(l);
        const float sr = wave_sum_to_lane63(rr);
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

// ---------------------------------------------------------------- partial-mix reduction

// partial[W][ncols] -> out[ncols], rows added in a fixed order (bit-reproducible).
__global__ __launch_bounds__(256) void sk_reduce_kernel(const float *__restrict__ partial,
                                                        float *__restrict__ out, int W, int ncols) {
  __shared__ float part[4][64];
  const int c = threadIdx.x & 63, slice = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + c;
  float s = 0.0f;
  if (col < ncols)
    for (int w = slice; w < W; w += 4) s += partial[(size_t)w * ncols + col];
  part[slice][c] = s;
  __syncthreads();
  if (slice == 0 && col < ncols) out[col] = ((part[0][c] + part[1][c]) + part[2][c]) + part[3][c];
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
  const size_t lds_bytes = (size_t)(tab_lds ? args->lds_table_floats : 0) * sizeof(float) +
                           (size_t)4 * SK_CHUNK * sizeof(float2);
  dim3 grid((unsigned)n_workgroups), block(SK_GROUP);
  if (tab_lds) {
    if (stems) hipLaunchKernelGGL((sk_render_kernel<true, true>), grid, block, lds_bytes, stream, *args);
    else       hipLaunchKernelGGL((sk_render_kernel<true, false>), grid, block, lds_bytes, stream, *args);
  } else {
    if (stems) hipLaunchKernelGGL((sk_render_kernel<false, true>), grid, block, lds_bytes, stream, *args);
    else       hipLaunchKernelGGL((sk_render_kernel<false, false>), grid, block, lds_bytes, stream, *args);
  }
  return (int)hipGetLastError();
}

extern "C" int sk_launch_reduce(const float *partial, float *out, int W, int ncols, hipStream_t stream) {
  hipLaunchKernelGGL(sk_reduce_kernel, dim3((unsigned)((ncols + 63) / 64)), dim3(256), 0, stream,
                     partial, out, W, ncols);
  return (int)hipGetLastError();
}

extern "C" int sk_launch_master(const float *sum, float *out, int num_frames, int num_channels,
                                float target, float k, float *gain_state, hipStream_t stream) {
  hipLaunchKernelGGL(sk_master_kernel, dim3(1), dim3(256), 0, stream, sum, out, num_frames,
                     num_channels, target, k, gain_state);
  return (int)hipGetLastError();
}
