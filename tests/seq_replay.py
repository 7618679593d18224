"""Drive the reference's seq() (oracle/_ref/libskred_ref.so) through a scripted session of pattern edits and audio
blocks and print, per block, which (pattern, step) fired and every pattern's pointer / counter afterwards.
tests/test_seq_clock.py replays the same script on skred_seq_t (libskred_amd.so) and compares.  Fresh process:
seq() keeps function-static state (the clock)."""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
from seq_script import script  # noqa: E402


def main():
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libskred_ref.so"))
    L.ref_boot()
    L.tempo_set.argtypes = [C.c_float]
    L.seq_step_set.argtypes = [C.c_int, C.c_int, C.c_char_p]
    freq = (C.c_float * 64).in_dll(L, "voice_freq")
    ptr = (C.c_int * 16).in_dll(L, "seq_pointer")
    cnt = (C.c_int * 16).in_dll(L, "seq_counter")
    out = []
    for op in script(int(sys.argv[1])):
        if op[0] == "tempo":
            L.tempo_set(op[1])
        elif op[0] == "step":          # the step's text sets voice <pattern>'s frequency to 1000 + step: a visible store
            _, p, s, occupied = op
            L.seq_step_set(p, s, (b"v%d f%d" % (p, 1000 + s)) if occupied else b"")
        elif op[0] == "mute":
            L.seq_mute_set(op[1], op[2], op[3])
        elif op[0] == "modulo":
            L.seq_modulo_set(op[1], op[2])
        elif op[0] == "state":
            L.seq_state_set(op[1], op[2])
        elif op[0] == "reset":
            L.pattern_reset(op[1])
        elif op[0] == "block":
            for v in range(16):
                freq[v] = 0.0
            L.seq(op[1])
            fired = [(v, int(freq[v]) - 1000) for v in range(16) if freq[v] != 0.0]
            out.append({"fired": fired, "pointer": list(ptr), "counter": list(cnt)})
    print("RESULT " + json.dumps(out))


def write_golden(path):
    """tests/golden/seq_clock.npz: the four scripted sessions' results from the compiled reference (one fresh process per
    session), so that the comparison also runs where the reference tree is absent."""
    import subprocess
    arrays = {}
    for seed in range(4):
        o = subprocess.run([sys.executable, os.path.abspath(__file__), str(seed)], capture_output=True, text=True, check=True)
        res = json.loads([l for l in o.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
        arrays[f"s{seed}_fired"] = np.array([(k, p, s) for k, r in enumerate(res) for p, s in r["fired"]], np.int32).reshape(-1, 3)
        arrays[f"s{seed}_pointer"] = np.array([r["pointer"] for r in res], np.int32)
        arrays[f"s{seed}_counter"] = np.array([r["counter"] for r in res], np.int32)
    np.savez_compressed(path, **arrays)
    print("wrote", path, {k: v.shape for k, v in arrays.items()})


if __name__ == "__main__":
    if sys.argv[1] == "--write-golden":
        write_golden(os.path.join(HERE, "golden", "seq_clock.npz"))
    else:
        main()
