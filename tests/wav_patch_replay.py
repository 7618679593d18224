"""Feed the wav_samples case's lines to OUR patch reader (libskred_synth.so) in the current directory
(which holds 1.wav .. 7.wav) and compare the resulting voice state with the reference's fixture.
Used by tests/test_wav.py; prints one JSON line."""
import ctypes as C
import importlib.util
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


def main():
    spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(HERE, "golden", "gen_golden.py"))
    gg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gg)
    gold = gio.load("wav_samples")
    seg = gold.segments[0]
    L = C.CDLL(os.path.join(ROOT, "skred_amd", "libskred_synth.so"))
    L.synth_init()
    L.wave_table_init()
    L.voice_init()

    class Patch(C.Structure):
        _fields_ = [("voice", C.c_int), ("unsupported", C.c_int), ("errors", C.c_int)]
    p = Patch()
    L.skred_patch_init(C.byref(p))
    loads, lines = gg.wav_case_lines()
    for ln in loads + lines:
        L.skred_patch_line(C.byref(p), ln.encode())
    res = {"unsupported": p.unsupported, "errors": p.errors, "field_mismatches": {}}
    V = 64
    for name, dt, _ in FIELDS:
        if name == "voice_table_offset":
            continue
        want = seg.bank_in.a[name]
        raw = (C.c_char * (want.dtype.itemsize * V)).in_dll(L, name)
        got = np.frombuffer(raw, dtype=want.dtype, count=V)
        if got.tobytes() != np.ascontiguousarray(want).tobytes():
            res["field_mismatches"][name] = int((got.view("u1") != np.ascontiguousarray(want).view("u1")).sum())
    # tables, voice by voice, through the pointers the voices hold
    tptr = np.frombuffer((C.c_char * (8 * V)).in_dll(L, "voice_table"), dtype="<u8", count=V)
    ok = True
    for v in range(V):
        n = int(seg.bank_in.a["voice_table_size"][v])
        off = int(seg.bank_in.a["voice_table_offset"][v])
        if n <= 0 or tptr[v] == 0:
            continue
        mine = np.ctypeslib.as_array(C.cast(int(tptr[v]), C.POINTER(C.c_float)), shape=(n,))
        ok = ok and gio.bits_equal(mine, gold.tables[off:off + n])
    res["tables_equal"] = bool(ok)
    W = 1200
    slots = [int(w[1]) for w in gold.extras["wav_loads"]]
    sl_ok = True
    for name, dt in gg.WAVE_SLOT_FIELDS:
        a = np.frombuffer((C.c_char * (4 * W)).in_dll(L, name), dtype=dt, count=W)[slots]
        sl_ok = sl_ok and a.tobytes() == gold.extras["slot_" + name].tobytes()
    res["slots_equal"] = bool(sl_ok)
    res["bad_slot_rc"] = int(L.skred_wave_load(1, 199, 0))
    res["missing_file_rc"] = int(L.skred_wave_load(99, 300, 0))
    print("RESULT " + json.dumps(res))


if __name__ == "__main__":
    main()
