// skred_gain_kernels.hip -- sk_gain_kernel: the envelopes of the voices on the motion list, one block ahead of the samples.
//
// Two-per-lane family, sparse lists (skred_bank.c: render_block decides).  The steady kernel has no room for envelope code
// (it sits at its register budget, and a voice in motion rendered by a second kernel BESIDE it costs a third round of
// workgroups: DESIGN "The motion list"), but the envelope of a voice needs nothing from its samples: amp_envelope_step
// (synth.c:398-431) is a function of the clock, the note's two time stamps and four constants.  So this kernel, ahead of the
// steady kernel on the same stream, evaluates it for every frame of the block and every listed voice -- lanes are FRAMES
// here: 64 consecutive frames of one voice per step, on the integer clocks the reference uses, one IEEE division where the
// reference has one -- and leaves amp * (level * velocity) (synth.c:582: what the voice's amp smoother is fed) of every frame
// in a row of a.env_gain.  The steady kernel's in-place instantiation (skred_render_fast2.hip: GT) keeps the voice in its
// lane, feeds the row to the voice's smoother frame by frame and renders it like any other.  (The smoother stays there: a
// 512-step serial walk per wave with a listed voice costs this kernel more than the three packed instructions per frame
// cost that one -- measured both ways.)  One wavefront per 64-voice word of the list:
//   * rows: every word owns a.env_word_rows rows (row = word * env_word_rows + the voice's rank in the word: no atomic --
//     thousands of waves asking one counter for rows took 70 us); a word with more listed voices than that takes consecutive
//     rows from the overflow area behind them (one atomic per such wave).  Each voice's row number goes into
//     a.env_list[voice].  (The list's length is counted by the steady kernel, wave by wave: a.env_count[0].)
//   * is_active: the first frame that finds the release run out clears it for the frames behind it (synth.c:429), and the
//     voice's flag word gets the end value before the steady kernel loads it;
//   * the NEXT block's list: the voices that are still moving on its first frame (sk_env_motion) -- this wave writes the
//     whole word, so nothing has to be zeroed or OR-ed.  (A voice whose envelope has come to rest but whose smoother still
//     settles leaves the list: the steady kernel runs smoothers by itself.)
#include "skred_kernel_common.hpp"
#include "skred_launch.h"

__global__ __launch_bounds__(256) void sk_gain_kernel(const sk_render_args_t a) {
  const int lane = threadIdx.x & 63;
  const int word = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
  if (word >= a.n_groups * 4) return;
  const uint64_t wv = a.mask_cur[word];
  const uint64_t w = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(wv >> 32)) << 32) |
                     (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)wv);   // (wave-uniform for the compiler too)
  if (w == 0) {
    if (lane == 0) a.mask_next[word] = 0;
    return;
  }
  const int v = word * 64 + lane;
  const bool listed = (w >> lane) & 1;
  // this lane's voice (listed lanes only)
  float amp = 0.0f, vel = 0.0f, att = 0.0f, dec = 0.0f, sus = 0.0f, rel = 0.0f, k = 0.0f, sgain = 0.0f;
  uint64_t t_start = 0, t_release = 0;
  uint32_t rw = 0;
  bool dead = true, act = false, fake = false;
  if (listed) {
    const uint4 osc = *reinterpret_cast<const uint4 *>(&a.ro[SKP_OSC][v]);
    const uint4 tab = *reinterpret_cast<const uint4 *>(&a.ro[SKP_TAB][v]);
    const uint4 gn = *reinterpret_cast<const uint4 *>(&a.ro[SKP_GAIN][v]);
    const uint4 et = *reinterpret_cast<const uint4 *>(&a.ro[SKP_ENV_T][v]);
    const uint4 es = *reinterpret_cast<const uint4 *>(&a.ro[SKP_ENV_S][v]);
    const uint4 s0 = *reinterpret_cast<const uint4 *>(&a.rw[SKS_OSC][v]);
    const uint4 s1 = *reinterpret_cast<const uint4 *>(&a.rw[SKS_FILT][v]);
    const uint32_t flags = tab.z;
    amp = __uint_as_float(osc.w);
    vel = __uint_as_float(gn.x); k = __uint_as_float(gn.y);
    att = __uint_as_float(et.x); dec = __uint_as_float(et.y); sus = __uint_as_float(et.z); rel = __uint_as_float(et.w);
    t_start = ((uint64_t)es.y << 32) | es.x;
    t_release = ((uint64_t)es.w << 32) | es.z;
    sgain = __uint_as_float(s0.y);
    rw = s1.w;
    act = (rw & SKR_ENV_ACTIVE) != 0;
    dead = (rw & SKR_FINISHED) || amp == 0.0f || (flags & SKF_INERT);
    if (!(flags & SKF_USE_ENV)) {   // no envelope on this voice: final = amp * 1.0f (synth.c:580-582), i.e. a note held at level 1
      att = dec = rel = 0.0f; sus = 1.0f; vel = 1.0f; t_start = a.count0; t_release = 0; act = true; fake = true;   // (as fast2_load)
    }
  }
  // rows
  const int cnt = __popcll(w);
  int base = word * a.env_word_rows;
  if (cnt > a.env_word_rows) {
    const int own = a.n_groups * 4 * a.env_word_rows;       // rows the words own; the overflow area follows
    int ov = 0;
    if (lane == 0) ov = (int)atomicAdd(a.env_count + 1, (uint32_t)cnt);
    ov = __builtin_amdgcn_readfirstlane(ov);
    if (own + ov + cnt > a.env_gain_cap) {   // (the host takes this path only with a proven bound on the list's length below the overflow
      if (lane == 0) atomicAdd(a.violations, 1u);   //  area's size: unreachable, counted, and the rows stay inside the buffer)
      ov = max(0, a.env_gain_cap - own - cnt);
    }
    base = own + ov;
  }
  const int slot = base + __popcll(w & (((uint64_t)1 << lane) - 1));
  if (listed) a.env_list[v] = slot;
  const bool was_act = act;
  // one listed voice at a time, 64 frames per step
  for (uint64_t rem = w; rem != 0; rem &= rem - 1) {
    const int l = __builtin_ctzll(rem);                  // (wave-uniform)
    if (__shfl((int)dead, l, 64)) continue;
    const float o_amp = __shfl(amp, l, 64), o_vel = __shfl(vel, l, 64), o_att = __shfl(att, l, 64), o_dec = __shfl(dec, l, 64);
    const float o_sus = __shfl(sus, l, 64), o_rel = __shfl(rel, l, 64);
    const uint64_t o_on = ((uint64_t)(uint32_t)__shfl((int)(t_start >> 32), l, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)t_start, l, 64);
    const uint64_t o_off = ((uint64_t)(uint32_t)__shfl((int)(t_release >> 32), l, 64) << 32) | (uint32_t)__shfl((int)(uint32_t)t_release, l, 64);
    const float o_attdec = o_att + o_dec;                // synth.c:410: decay_start + decay_time
    const float o_oms = 1.0f - o_sus;                    // synth.c:413
    const bool released = o_off != 0;                    // synth.c:417
    bool o_act = __shfl((int)act, l, 64) != 0;
    float *const row = a.env_gain + (size_t)__shfl(slot, l, 64) * (size_t)a.env_gain_stride;
    for (int f0 = 0; f0 < a.num_frames; f0 += 64) {
      const int j = f0 + lane;
      const uint64_t now = a.count0 + (uint64_t)j + 1;   // synth.c:521
      const float tf = (float)(now - o_on), trf = (float)(now - o_off);
      // amp_envelope_step as the reference writes it (synth.c:401-430)
      float lvl = 0.0f;
      bool runs_out = false;
      if (tf < o_att) {
        lvl = tf / o_att;
      } else if (tf < o_attdec) {
        const float prog = (tf - o_att) / o_dec;
        lvl = 1.0f - prog * o_oms;
      } else if (!released) {
        lvl = o_sus;
      } else if (trf < o_rel) {
        const float prog = trf / o_rel;
        lvl = o_sus * (1.0f - prog);
      } else {
        runs_out = true;                                 // is_active = 0 from this frame on (synth.c:429)
      }
      const uint64_t hit = __ballot(o_act && runs_out && j < a.num_frames);
      const bool active_here = o_act && (hit & (((uint64_t)1 << lane) - 1)) == 0;
      if (!active_here) lvl = 0.0f;                      // synth.c:399-400
      if (j < a.num_frames) row[j] = o_amp * (lvl * o_vel);   // synth.c:582
      if (hit != 0) o_act = false;
    }
    if (lane == l) act = o_act;
  }
  if (listed && !dead && !fake && was_act && !act)       // (a voice without envelope keeps its flag: its note is a fiction)
    reinterpret_cast<uint32_t *>(&a.rw[SKS_FILT][v])[3] = rw & ~SKR_ENV_ACTIVE;
  // who is still in motion when the next block starts
  bool keep = false;
  if (listed) {
    const sk_motion_t mo = sk_env_motion(a.count0 + (uint64_t)a.num_frames + 1, dead, act, t_start, t_release, att, att + dec, rel, sus, amp, vel, k, sgain);
    keep = mo.moving;
  }
  const uint64_t kb = __ballot(keep);
  if (lane == 0) a.mask_next[word] = kb;
}

extern "C" int sk_launch_gain(const sk_render_args_t *args, hipStream_t stream) {
  hipLaunchKernelGGL(sk_gain_kernel, dim3((unsigned)args->n_groups), dim3(256), 0, stream, *args);
  return (int)hipGetLastError();
}
