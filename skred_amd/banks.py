"""Synthetic, seeded voice banks for the BASELINE configs (SURVEY §8d recipes).

Control-path helpers only: they produce the INPUTS of the render loop (phase increments,
envelope times, biquad coefficients, pan gains) the way the reference's setters do
(SURVEY §8a row a14), vectorised over N voices.  sample_rate is a parameter here (SURVEY D1:
the reference hard-codes 44100, skred.h:6; throughput runs use 48000).

Randomness: the reference's own 64-bit LCG and float mapping (synth.c:110-123), seed 0x5EED,
so that any harness can regenerate the identical bank.
"""
from __future__ import annotations

import json
import os
from typing import Dict, Tuple

import numpy as np

from .bank import GlobalsC, VoiceBank

LCG_A = np.uint64(6364136223846793005)
LCG_C = np.uint64(1442695040888963407)
SEED = 0x5EED
_HERE = os.path.dirname(os.path.abspath(__file__))
LUT_FILE = os.path.join(_HERE, "data", "notamy_luts.npz")      # package data (skred_amd/data/extract_notamy_luts.py)
PCM_LENGTH = 1176036        # amysamples.h:7
PCM_RATE = 22050.0          # amysamples.h:6


def lcg_stream(n: int, seed: int) -> np.ndarray:
    """First n states after `seed` of s' = s*A + C (mod 2^64), vectorised by affine doubling."""
    out = np.empty(n, np.uint64)
    with np.errstate(over="ignore"):
        # state_k = A^k * s0 + C * (A^(k-1) + ... + 1)
        mul = np.empty(n, np.uint64)
        add = np.empty(n, np.uint64)
        m, a = LCG_A, LCG_C
        mul[0], add[0] = m, a
        filled = 1
        while filled < n:
            take = min(filled, n - filled)
            # compose: (x -> mul[j] x + add[j]) after (x -> m_blk x + a_blk) where blk = 'filled' steps
            mb, ab = mul[filled - 1], add[filled - 1]
            mul[filled:filled + take] = mul[:take] * mb
            add[filled:filled + take] = mul[:take] * ab + add[:take]
            filled += take
        out[:] = mul * np.uint64(seed) + add
    return out


def lcg_uniform(n: int, seed: int) -> np.ndarray:
    """n float32 in [0,1): the reference's noise mapping (int32(state>>32)/2^31, synth.c:117-123) rescaled."""
    s = lcg_stream(n, seed)
    hi = (s >> np.uint64(32)).astype(np.uint32).view(np.int32)
    f = hi.astype(np.float32) / np.float32(2147483648.0)
    return ((f + np.float32(1.0)) * np.float32(0.5)).astype(np.float32)


# ------------------------------------------------------------------ control-path helpers

def phase_inc(freq, table_size, table_rate, sample_rate, one_shot=None, offset_hz=None):
    """== osc_get_phase_inc, synth.c:125-132, with MAIN_SAMPLE_RATE -> sample_rate."""
    g = np.asarray(freq, np.float32)
    if one_shot is not None:
        g = np.where(one_shot != 0, g / np.asarray(offset_hz, np.float32), g).astype(np.float32)
    size = np.asarray(table_size).astype(np.float32)
    rate = np.asarray(table_rate, np.float32)
    return ((g * size) / rate * (rate / np.float32(sample_rate))).astype(np.float32)


def pan_gains(pan):
    """== pan_set, synth.c:838-847."""
    p = np.asarray(pan, np.float32)
    return ((np.float32(1) - p) / np.float32(2)).astype(np.float32), ((np.float32(1) + p) / np.float32(2)).astype(np.float32)


def set_envelope(bank: VoiceBank, a, d, s, r, sample_rate, velocity, sample_start):
    """== envelope_init + amp_envelope_trigger + envelope_velocity, synth.c:367-388,1146-1159."""
    e = bank["voice_amp_envelope"]
    e["a"], e["d"], e["s"], e["r"] = a, d, s, r
    e["attack_time"] = np.float32(a) * np.float32(sample_rate)
    e["decay_time"] = np.float32(d) * np.float32(sample_rate)
    e["sustain_level"] = np.clip(np.float32(s), 0, 1)
    e["release_time"] = np.float32(r) * np.float32(sample_rate)
    e["sample_start"] = sample_start
    e["sample_release"] = 0
    e["is_active"] = 1
    e["velocity"] = velocity
    bank["voice_use_amp_envelope"] = 1


def biquad_coeffs(mode, freq, q, sample_rate) -> Dict[str, np.ndarray]:
    """RBJ cookbook coefficients, == mmf_set_params synth.c:929-1008 (float32 throughout)."""
    f32 = np.float32
    mode = np.asarray(mode)
    omega = (f32(2.0) * f32(np.pi) * np.asarray(freq, f32) / f32(sample_rate)).astype(f32)
    sn, cs = np.sin(omega).astype(f32), np.cos(omega).astype(f32)
    alpha = (sn / (f32(2.0) * np.asarray(q, f32))).astype(f32)
    one = f32(1.0)
    a0 = one + alpha
    a1 = f32(-2.0) * cs
    a2 = one - alpha
    lp = ((one - cs) / f32(2.0), one - cs, (one - cs) / f32(2.0))
    hp = ((one + cs) / f32(2.0), -(one + cs), (one + cs) / f32(2.0))
    bp = (alpha, np.zeros_like(alpha), -alpha)
    nt = (np.ones_like(alpha), f32(-2.0) * cs, np.ones_like(alpha))
    ap = (one - alpha, f32(-2.0) * cs, one + alpha)
    b = [np.select([mode == 2, mode == 3, mode == 4, mode == 5], [hp[i], bp[i], nt[i], ap[i]], lp[i]) for i in range(3)]
    return {"b0": (b[0] / a0).astype(f32), "b1": (b[1] / a0).astype(f32), "b2": (b[2] / a0).astype(f32),
            "a1": (a1 / a0).astype(f32), "a2": (a2 / a0).astype(f32)}


def sine_table(size: int = 4096) -> np.ndarray:
    """The reference's run-time sine table: sinf(2*pi*phase), phase += 1/size in f32 (synth.c:1231-1248)."""
    ph = np.cumsum(np.full(size, np.float32(1.0) / np.float32(size), np.float32), dtype=np.float32) - np.float32(1.0) / np.float32(size)
    return np.sin(np.float32(2.0) * np.float32(np.pi) * ph.astype(np.float32)).astype(np.float32)


def load_luts():
    z = np.load(LUT_FILE)
    names = json.loads(str(z["names"]))
    meta = json.loads(str(z["meta"]))
    return z, names, meta


# ------------------------------------------------------------------ the recipes

def _common(bank: VoiceBank, u_freq, u_pan, u_start, sample_rate, count0):
    n = bank.n
    freq = (np.float32(27.5) * np.exp2(np.float32(7.0) * u_freq)).astype(np.float32)
    bank["voice_amp"] = 1.0
    pl, pr = pan_gains(u_pan * np.float32(2.0) - np.float32(1.0))
    bank["voice_pan_left"], bank["voice_pan_right"] = pl, pr
    stagger = np.minimum((u_start * np.float32(sample_rate)).astype(np.int64), sample_rate - 1)
    set_envelope(bank, 0.01, 0.1, 0.7, 0.2, sample_rate, 1.0, (count0 - stagger).astype(np.uint64))
    return freq


def make_globals(sample_rate: int) -> GlobalsC:
    g = GlobalsC.defaults()
    g.synth_sample_count = 2 * sample_rate          # voices started up to 1 s ago
    return g


def bank_c1(n: int = 4096, sample_rate: int = 48000, seed: int = SEED) -> Tuple[VoiceBank, np.ndarray, GlobalsC]:
    """C1: n voices on the 4096-entry sine table, ADSR + default amp smoother, no filter."""
    g = make_globals(sample_rate)
    u = lcg_uniform(3 * n, seed).reshape(3, n)
    b = VoiceBank(n)
    freq = _common(b, u[0], u[1], u[2], sample_rate, g.synth_sample_count)
    table = sine_table(4096)
    table = np.concatenate([table, table[:1]])     # guard sample: the table's first value once more behind it (SKF_GUARD)
    b["voice_table_offset"] = 0
    b["voice_table_size"] = 4096
    b["voice_loop_start_f"], b["voice_loop_end_f"] = 0.0, 4095.0   # wave_loop_end = size-1 (synth.c:1229), unused: loop off
    b["voice_loop_valid"] = 1
    b["voice_phase_inc"] = phase_inc(freq, 4096, sample_rate, sample_rate)
    b["voice_phase"] = (u[2] * np.float32(4095.0)).astype(np.float32)
    return b, table, g


def bank_c2(n: int = 65536, sample_rate: int = 48000, seed: int = SEED) -> Tuple[VoiceBank, np.ndarray, GlobalsC]:
    """C2/C3: v mod 3 -> notamy sine(256) / triangle pyramid / impulse pyramid (level by frequency so that
    highest_harmonic * f stays below Nyquist); biquad mode 1 + v mod 4, K ~ logU[100,8000], Q ~ U[0.5,4]."""
    g = make_globals(sample_rate)
    z, names, meta = load_luts()
    u = lcg_uniform(5 * n, seed).reshape(5, n)
    b = VoiceBank(n)
    freq = _common(b, u[0], u[1], u[2], sample_rate, g.synth_sample_count)
    # pool = all notamy float LUTs, in file order, each followed by a GUARD sample (its first value once more): a voice that
    # loops over a whole table then finds the second tap of the linear lookup in the next float, whatever the index
    # (include/skred_amd.h: skred_bank_set_tables_f32; skred_device_layout.h: SKF_GUARD) -- the samples are the same either way
    offs, pos, pool = {}, 0, []
    for nm in names:
        t = z["f32_" + nm]
        offs[nm] = pos
        pool.append(t)
        pool.append(t[:1])
        pos += len(t) + 1
    pool = np.concatenate(pool).astype(np.float32)
    fam = np.arange(n) % 3
    t_off = np.zeros(n, np.int64)
    t_size = np.zeros(n, np.int32)
    nyq = sample_rate / 2.0
    for f_id, family in enumerate(("sine", "triangle", "impulse")):
        levels = [nm for nm in names if nm.startswith(family + "_")]
        hh = np.array([meta[nm]["highest_harmonic"] for nm in levels], np.float64)
        sel = np.where(fam == f_id)[0]
        ok = hh[None, :] * freq[sel, None].astype(np.float64) <= nyq
        lvl = np.where(ok.any(1), ok.argmax(1), len(levels) - 1)
        t_off[sel] = np.array([offs[levels[k]] for k in lvl], np.int64)
        t_size[sel] = np.array([meta[levels[k]]["table_size"] for k in lvl], np.int32)
    b["voice_table_offset"], b["voice_table_size"] = t_off, t_size
    b["voice_loop_start_f"] = 0.0
    b["voice_loop_end_f"] = (t_size - 1).astype(np.float32)
    b["voice_loop_valid"] = (t_size > 1).astype(np.int32)
    b["voice_wave_table_index"] = 200 + fam
    b["voice_phase_inc"] = phase_inc(freq, t_size, sample_rate, sample_rate)
    b["voice_phase"] = (u[2] * (t_size - 1).astype(np.float32)).astype(np.float32)
    mode = (1 + np.arange(n) % 4).astype(np.int32)
    cutoff = (np.float32(100.0) * np.power(np.float32(80.0), u[3])).astype(np.float32)
    q = (np.float32(0.5) + np.float32(3.5) * u[4]).astype(np.float32)
    co = biquad_coeffs(mode, cutoff, q, sample_rate)
    flt = b["voice_filter"]
    for k, v in co.items():
        flt[k] = v
    flt["last_freq"], flt["last_resonance"], flt["last_mode"] = cutoff, q, mode
    b["voice_filter_mode"] = mode
    return b, pool, g


def synthetic_pcm_blob(seed: int = SEED) -> Tuple[np.ndarray, np.ndarray]:
    """C4 sample ROM stand-in: the real blob is missing from the reference mount, so
    the content is seeded noise, low-passed per region, int16; the GEOMETRY is the real pcm_map.
    Returns (int16 blob [PCM_LENGTH], pcm_map [67][5])."""
    z, _, _ = load_luts()
    pm = z["pcm_map"]
    raw = lcg_uniform(PCM_LENGTH, seed ^ 0xC4) * np.float32(2.0) - np.float32(1.0)
    k = np.array([1, 4, 6, 4, 1], np.float32) / np.float32(16.0)
    sm = np.convolve(raw, k, mode="same").astype(np.float32)
    blob = np.zeros(PCM_LENGTH, np.int16)
    for off, ln, _, _, _ in pm:
        seg = sm[off:off + ln]
        peak = np.abs(seg).max()
        blob[off:off + ln] = np.round(seg / peak * 32767.0).astype(np.int16)
    return blob, pm


def pcm_float_tables(blob: np.ndarray, pm: np.ndarray) -> np.ndarray:
    """int16 ROM -> float tables the way wave_table_init loads AMY samples: /32767 then per-region
    peak normalisation (synth.c:1278-1282,1175-1197)."""
    t = blob.astype(np.float32) / np.float32(32767.0)
    for off, ln, _, _, _ in pm:
        seg = t[off:off + ln]
        peak = np.abs(seg).max()
        if peak > 0:
            t[off:off + ln] = seg * (np.float32(1.0) / peak)
    return t


def bank_c4(n: int = 262144, sample_rate: int = 48000, seed: int = SEED) -> Tuple[VoiceBank, np.ndarray, GlobalsC]:
    """C4: PCM playback, region = pcm_map[v mod 67], looped between the map's loop points,
    phase_inc ~ U[0.25, 2] table samples per frame.  Meant for interp=linear."""
    g = make_globals(sample_rate)
    blob, pm = synthetic_pcm_blob(seed)
    pool = pcm_float_tables(blob, pm)
    u = lcg_uniform(4 * n, seed).reshape(4, n)
    b = VoiceBank(n)
    _common(b, u[0], u[1], u[2], sample_rate, g.synth_sample_count)
    reg = np.arange(n) % 67
    b["voice_table_offset"] = pm[reg, 0]
    b["voice_table_size"] = pm[reg, 1].astype(np.int32)
    b["voice_one_shot"] = 1
    b["voice_loop_enabled"] = 1
    b["voice_loop_valid"] = (pm[reg, 3] > pm[reg, 2]).astype(np.int32)
    b["voice_loop_start_f"] = pm[reg, 2].astype(np.float32)
    b["voice_loop_end_f"] = pm[reg, 3].astype(np.float32)
    b["voice_wave_table_index"] = (100 + reg).astype(np.int32)
    b["voice_phase_inc"] = (np.float32(0.25) + np.float32(1.75) * u[3]).astype(np.float32)
    b["voice_phase"] = (u[2] * pm[reg, 2].astype(np.float32)).astype(np.float32)
    return b, pool, g


PATCH_DIR = os.path.join(_HERE, "data", "patches")    # package data (generated by tests/golden/gen_patch_voices.py where the reference is mounted)


def bank_patch(patch: str, n: int, sample_rate: int = 44100) -> Tuple[VoiceBank, np.ndarray, GlobalsC]:
    """A reference PATCH tiled over a large bank: the voice state the unmodified reference holds after loading `patch`
    (skred_amd/data/patches/patch_<patch>.npz, written by tests/golden/gen_patch_voices.py from the reference's own wire()) -- its first K
    voices, K the smallest power of two that holds every voice in use -- repeated n / K times, every copy with its modulator
    indices moved along (a copy never straddles an aligned 64-voice group) and its start phases spread by the LCG.  The routings
    the shipped patches use: a modulator shared by three carriers (3sk), frequency + pan modulation from two voices and sample &
    hold (37sk), chains (7sk), a modulator BELOW its carrier (18sk)."""
    z = np.load(os.path.join(PATCH_DIR, f"patch_{patch}.npz"))
    from .bank import FIELDS
    src = {name: z["in_" + name] for name, _, _ in FIELDS}
    used = np.where((src["voice_amp"] != 0) & (src["voice_table_size"] > 0))[0]
    K = 1
    while K <= int(used.max()):
        K *= 2
    assert K <= 64 and n % K == 0
    b = VoiceBank(n)
    copies = n // K
    for name, _, _ in FIELDS:
        b.a[name] = np.ascontiguousarray(np.tile(src[name][:K], copies))
    base = np.repeat(np.arange(copies, dtype=np.int32) * K, K)
    for f in ("voice_freq_mod_osc", "voice_amp_mod_osc", "voice_pan_mod_osc", "voice_cz_mod_osc"):
        m = b.a[f]
        b.a[f] = np.where(m >= 0, m + base, m).astype(np.int32)
    dead = np.ones(K, bool)
    dead[used[used < K]] = False
    b.a["voice_amp"][np.tile(dead, copies)] = 0.0           # slots the patch does not use stay silent (skipped)
    u = lcg_uniform(n, SEED)
    span = np.maximum(b.a["voice_table_size"].astype(np.float32) - np.float32(1.0), np.float32(0.0))
    b.a["voice_phase"] = (u * span).astype(np.float32)
    g = GlobalsC.defaults()
    g.synth_sample_count = 2 * sample_rate
    return b, z["tables"].astype(np.float32), g


RECIPES = {"c1": bank_c1, "c2": bank_c2, "c3": bank_c2, "c4": bank_c4}
DEFAULT_N = {"c1": 4096, "c2": 65536, "c3": 1048576, "c4": 262144}
