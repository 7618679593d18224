"""Drive the eight per-sample functions of synth.h (audio_rng_*, cz_phasor, osc_next, quantize_bits_int,
mmf_process, amp_envelope_step) of ONE library -- the compiled reference (oracle/_ref/libskred_ref.so)
or our drop-in (skred_amd/libskred_synth.so) -- on the voice state of one golden fixture, in a fresh
process, and print a digest of every returned value and of the state they leave behind.
tests/test_dropin.py runs it twice and compares.  Also prints voice_format() of every voice (plain and
verbose, minus the wall-clock latency figure)."""
import ctypes as C
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import golden_io as gio  # noqa: E402
from skred_amd.bank import FIELDS  # noqa: E402

NV = 64
STEPS = 700


def install(L, seg, tables):
    """fixture state -> the library's global arrays (every hot field + table pointers)"""
    keep = []
    bank = seg.bank_in
    for name, dt, _ in FIELDS:
        if name == "voice_table_offset":
            continue
        raw = np.ascontiguousarray(bank.a[name][:NV])
        dst = (C.c_char * raw.nbytes).in_dll(L, name)
        C.memmove(dst, raw.ctypes.data, raw.nbytes)
    pool = np.ascontiguousarray(tables, np.float32)
    keep.append(pool)
    ptrs = (C.c_void_p * NV).in_dll(L, "voice_table")
    for v in range(min(NV, bank.n)):
        ptrs[v] = pool.ctypes.data + 4 * int(bank.a["voice_table_offset"][v])
    C.c_uint64.in_dll(L, "synth_sample_count").value = seg.g_in.synth_sample_count
    return keep


def main():
    mode, case = sys.argv[1], sys.argv[2]
    so = os.path.join(ROOT, "oracle", "_ref", "libskred_ref.so") if mode == "ref" else os.path.join(ROOT, "skred_amd", "libskred_synth.so")
    L = C.CDLL(so)
    f32 = C.c_float
    L.audio_rng_next.restype = C.c_uint64
    L.audio_rng_float.restype = f32
    L.cz_phasor.restype = f32; L.cz_phasor.argtypes = [C.c_int, f32, f32, C.c_int]
    L.osc_next.restype = f32; L.osc_next.argtypes = [C.c_int, f32]
    L.quantize_bits_int.restype = f32; L.quantize_bits_int.argtypes = [f32, C.c_int]
    L.mmf_process.restype = f32; L.mmf_process.argtypes = [C.c_int, f32]
    L.amp_envelope_step.restype = f32; L.amp_envelope_step.argtypes = [C.c_int]
    L.voice_format.restype = C.c_char_p; L.voice_format.argtypes = [C.c_int, C.c_char_p, C.c_int]

    g = gio.load(case)
    out = {}
    h = hashlib.sha256()
    texts = []

    # the libraries' own tables (built-in cycles, Korg waves of slots 32-62) and setters first: every voice
    # on another slot with a patch's worth of parameters, stepped by hand
    if mode == "ref":
        L.ref_boot()
    else:
        L.wave_table_init()
        L.voice_init()
    for name in ("freq_set", "amp_set", "pan_set"):
        getattr(L, name).argtypes = [C.c_int, f32]
    L.cz_set.argtypes = [C.c_int, C.c_int, f32]
    L.envelope_set.argtypes = [C.c_int, f32, f32, f32, f32]
    L.envelope_velocity.argtypes = [C.c_int, f32]
    L.mmf_set_freq.argtypes = [C.c_int, f32]
    L.mmf_set_res.argtypes = [C.c_int, f32]
    for name in ("freq_mod_set", "amp_mod_set", "pan_mod_set", "cmod_set"):
        getattr(L, name).argtypes = [C.c_int, C.c_int, f32]
    slots = [0, 1, 2, 3, 4] + list(range(32, 63))
    fmode = (C.c_int * NV).in_dll(L, "voice_filter_mode")
    for v in range(NV):
        L.wave_set(v, slots[v % len(slots)])
        L.freq_set(v, 55.0 * (1.0 + v * 0.37))
        L.amp_set(v, 0.1 + v * 0.01)
        L.pan_set(v, (v % 9 - 4) / 4.0)
        if v % 3 == 0:
            fmode[v] = 1 + v % 5
            L.mmf_set_freq(v, 300.0 + 40.0 * v)
            L.mmf_set_res(v, 0.6 + 0.05 * v)
        if v % 4 == 1:
            L.cz_set(v, 1 + v % 7, 0.1 + 0.01 * v)
        if v % 8 == 5:
            L.cmod_set(v, (v + 1) % NV, 0.3)
        if v % 5 == 2:
            L.envelope_set(v, 0.001 * (v + 1), 0.002 * v, 0.6, 0.003 * v)
            L.envelope_velocity(v, 0.8)
        if v % 7 == 3:
            L.wave_dir(v, 1)
        if v % 6 == 4:
            L.freq_mod_set(v, (v + 2) % NV, 2.0)
            L.amp_mod_set(v, (v + 3) % NV, 0.5)
            L.pan_mod_set(v, (v + 5) % NV, 0.25)
    count = C.c_uint64.in_dll(L, "synth_sample_count")
    incs = np.ctypeslib.as_array((C.c_float * NV).in_dll(L, "voice_phase_inc")).copy()
    h.update(incs.tobytes())
    vals = np.zeros((300, NV, 4), np.float32)
    for i in range(300):
        count.value += 1
        if i == 150:
            for v in range(2, NV, 5):
                L.envelope_velocity(v, 0.0)          # note-off
        for v in range(NV):
            s = L.osc_next(v, float(incs[v]))
            vals[i, v] = (s, L.quantize_bits_int(s, 1 + (v + i) % 12), L.mmf_process(v, s), L.amp_envelope_step(v))
            (C.c_float * NV).in_dll(L, "voice_sample")[v] = s
    h.update(vals.tobytes())
    buf = C.create_string_buffer(4096)
    for v in range(NV):
        for verbose in (0, 1):
            t = L.voice_format(v, buf, verbose).decode()
            texts.append(t[:t.rfind(" latency:")] if verbose else t)
    for seg in g.segments[:2]:
        keep = install(L, seg, g.tables)
        nv = min(NV, seg.bank_in.n)
        count = C.c_uint64.in_dll(L, "synth_sample_count")
        inc = seg.bank_in.a["voice_phase_inc"]
        vals = np.zeros((STEPS, nv, 4), np.float32)
        for i in range(STEPS):
            count.value += 1
            for v in range(nv):
                s = L.osc_next(v, float(inc[v]))
                q = L.quantize_bits_int(s, 1 + (v + i) % 12)
                y = L.mmf_process(v, s)
                e = L.amp_envelope_step(v)
                vals[i, v] = (s, q, y, e)
                (C.c_float * NV).in_dll(L, "voice_sample")[v] = s       # what a CZ modulator reads
        h.update(vals.tobytes())
        for name, dt, rw in FIELDS:
            if rw:
                h.update(bytes((C.c_char * (dt.itemsize * NV)).in_dll(L, name)))
        buf = C.create_string_buffer(4096)
        for v in range(nv):
            plain = L.voice_format(v, buf, 0).decode()
            verbose = L.voice_format(v, buf, 1).decode()
            texts.append(plain)
            texts.append(verbose[:verbose.rfind(" latency:")])
        del keep
    out["state_and_values"] = h.hexdigest()
    out["text"] = hashlib.sha256("\n".join(texts).encode()).hexdigest()
    out["text_sample"] = texts[:4]

    # the stateless ones on grids that cover every branch
    h = hashlib.sha256()
    rng = C.c_uint64(0)
    for seed in (0, 1, 0x5EED, 2 ** 64 - 1):
        L.audio_rng_init(C.byref(rng), C.c_uint64(seed))
        h.update(np.uint64(rng.value).tobytes())
        for _ in range(50):
            h.update(np.uint64(L.audio_rng_next(C.byref(rng))).tobytes())
            h.update(np.float32(L.audio_rng_float(C.byref(rng))).tobytes())
    ph = np.concatenate([np.linspace(-10, 4200, 331, dtype=np.float32), np.float32([0, 2048, 4095.5, np.nan, np.inf])])
    dd = np.float32([-0.5, 0.0, 0.1, 0.33, 0.5, 0.9, 0.999, 1.0, 3.0])
    for n in range(0, 9):
        for size in (4096, 707):
            r = np.float32([[L.cz_phasor(n, float(p), float(d), size) for p in ph] for d in dd])
            h.update(r.tobytes())
    xs = np.concatenate([np.linspace(-1.2, 1.2, 241, dtype=np.float32), np.float32([1e-8, -1e-8, 0.5, -0.5])])
    for bits in range(1, 17):
        h.update(np.float32([L.quantize_bits_int(float(x), bits) for x in xs]).tobytes())
    out["stateless"] = h.hexdigest()
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    main()
