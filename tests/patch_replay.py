"""Load one reference patch N.sk either through the reference (wire(), libskred_ref.so) or through OUR
loader (skred_patch_load, libskred_synth.so) in a fresh process and print a digest of every per-voice
array of the synth.h ABI (pointers and timing marks excluded).  Used by tests/test_patch_loader.py."""
import ctypes as C
import hashlib
import json
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("SKRED_REFERENCE", "/root/reference")


def voice_arrays():
    text = open(os.path.join(ROOT, "include", "skred_synth_abi.h")).read()
    out = []
    for ctype, name in re.findall(r"^extern\s+(float|int|skred_mmf_t|skred_envelope_t)\s+(voice_\w+)\[SKRED_VOICE_MAX\];", text, flags=re.M):
        out.append((name, {"float": 4, "int": 4, "skred_mmf_t": 48, "skred_envelope_t": 56}[ctype]))
    return out


def digest(L):
    h = {}
    for name, size in voice_arrays():
        raw = bytes((C.c_char * (size * 64)).in_dll(L, name))
        if name == "voice_filter":          # cache keys of an unused filter may differ in padding only: keep all
            pass
        h[name] = hashlib.sha256(raw).hexdigest()[:16]
    for name in ("volume_user", "volume_final"):
        h[name] = C.c_float.in_dll(L, name).value
    return h


def main():
    n, mode = int(sys.argv[1]), sys.argv[2]
    path = os.path.join(REF, f"{n}.sk")
    res = {"patch": n, "mode": mode}
    if mode == "ref":
        L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libskred_ref.so"))
        L.ref_boot()
        res["rc"] = L.ref_load_patch(REF.encode(), n)
    else:
        L = C.CDLL(os.path.join(ROOT, "skred_amd", "libskred_synth.so"))
        L.wave_table_init()
        L.voice_init()

        class Patch(C.Structure):
            _fields_ = [("voice", C.c_int), ("unsupported", C.c_int), ("errors", C.c_int)]
        p = Patch()
        L.skred_patch_init(C.byref(p))
        os.chdir(REF)                      # `:wN` reads N.wav from the current directory (wire.c:409), as ref_load_patch does
        res["rc"] = L.skred_patch_load(path.encode(), C.byref(p))
        res["unsupported"], res["errors"] = p.unsupported, p.errors
    res["digest"] = digest(L)
    print("RESULT " + json.dumps(res))


if __name__ == "__main__":
    main()
