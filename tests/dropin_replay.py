"""Replay one golden case through the DROP-IN build and compare with the reference's fixture.

The process image is oracle/_ref/libskred_dropin_check.so: the reference's own wire.o / seq.o /
skred.o / ... linked against OUR libskred_synth.so in place of synth.o (oracle/Makefile:
dropin_check).  The very same case script that generated the fixture from the reference
(tests/golden/gen_golden.py) is replayed: every control line goes through the reference's wire()
into our setters; with --render the audio callback runs too (reference synth_callback -> our
synth() -> GPU).  Prints one JSON line.  Run in a fresh process per case.
"""
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
from skred_amd.bank import FIELD_NAMES  # noqa: E402

LIB = os.path.join(ROOT, "oracle", "_ref", "libskred_dropin_check.so")


class Stop(Exception):
    pass


def field_mismatches(a, b):
    bad = {}
    for k in FIELD_NAMES:
        x, y = a.a[k], b.a[k]
        if x.dtype.names:
            m = sum(int((x[n].view("u%d" % x[n].dtype.itemsize) != y[n].view("u%d" % y[n].dtype.itemsize)).sum())
                    for n in x.dtype.names)
        else:
            m = int((x.view("u%d" % x.dtype.itemsize) != y.view("u%d" % y.dtype.itemsize)).sum())
        if m:
            bad[k] = m
    return bad


def main():
    case, render = sys.argv[1], "--render" in sys.argv
    spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(HERE, "golden", "gen_golden.py"))
    gg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gg)
    golden = gio.load(case)
    res = {"case": case, "segments": [], "extras": {}}

    class CompareSink:
        def __init__(self, name, desc):
            self.k = 0

        def segment(self, ref, frames, block=512, keep_stems=(), note=""):
            seg = golden.segments[self.k]
            bank, tables = ref.snapshot()
            r = {"k": self.k, "state_in": field_mismatches(bank, seg.bank_in),
                 "tables_equal": bool(tables.shape == golden.tables.shape and (tables.view(np.uint32) == golden.tables.view(np.uint32)).all())}
            g = ref.globals()
            r["globals_in_equal"] = bool(g.synth_sample_count == seg.g_in.synth_sample_count and
                                     np.float32(g.volume_final) == np.float32(seg.g_in.volume_final) and
                                     np.float32(g.volume_smoother_gain) == np.float32(seg.g_in.volume_smoother_gain))
            res["segments"].append(r)
            if not render:
                raise Stop()
            mix, stems = ref.render(frames, block)
            r["rc"] = int(ref.L.skred_synth_last_rc())
            if r["rc"]:
                ref.L.skred_synth_last_error.restype = __import__("ctypes").c_char_p
                r["error"] = ref.L.skred_synth_last_error().decode(errors="replace")
            r["stems_sha_equal"] = bool(gio.sha256(stems) == seg.stems_sha256)
            d = mix.astype(np.float64) - seg.mix.astype(np.float64)
            r["mix_rms_err"] = float(np.sqrt(np.mean(d ** 2)))
            bank_out, _ = ref.snapshot()
            r["state_out"] = bank_out.rw_equal(gio.expected_out_bank(seg))
            r["count_out_equal"] = bool(ref.globals().synth_sample_count == seg.g_out.synth_sample_count)
            self.k += 1
            return mix, stems

        def extra(self, name, value):
            # case-specific data: what the replay produced (decoded tables, slot fields, the recorder's
            # WAV file) must equal what the reference produced when the fixture was made
            want = golden.extras[name]
            got = np.ascontiguousarray(value)
            res.setdefault("extras", {})[name] = bool(got.shape == want.shape and got.tobytes() == want.tobytes())

        def save(self):
            pass

    ref = gg.Ref(LIB)
    try:
        gg.CASES[case](ref, CompareSink)
    except Stop:
        pass
    print("RESULT " + json.dumps(res))


if __name__ == "__main__":
    main()
