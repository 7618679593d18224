#!/usr/bin/env python3
"""The voice state a reference patch leaves behind, as data: skred_amd/data/patches/patch_<n>sk.npz (package data: the banks
skred_amd.banks.bank_patch() builds are used by tools/, bench-side measurements and tests alike).

Feeds the lines of a few reference patches (the modulation routings VERDICT r2 names: 3.sk, 37.sk, 7.sk, 18.sk, 1.sk) to the
UNMODIFIED reference's wire() (oracle/_ref/libskred_ref.so) and stores the 64 voices' fields plus the tables they reference.
skred_amd.banks.bank_patch() tiles such a patch over a large bank (tools/measure_banks.py, tests).  Runs only where
/root/reference exists.  Voices on AMY sample slots (w100-w199: the sample ROM is absent from the mount) are moved to
built-in waves of the same kind of use (w108 -> w1, w105 -> w4, w110 -> w2) -- the ROUTING is what these fixtures are about; the
substitutions are listed in the file's meta."""
import json
import os
import re
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
OUT_DIR = os.path.join(ROOT, "skred_amd", "data", "patches")
PATCHES = [3, 37, 7, 18, 1]
AMY_SUBST = {108: 1, 105: 4, 110: 2}


def one(n):
    import gen_golden as gg
    ref = gg.Ref()
    subst = []
    for raw in open(os.path.join(gg.REF_DIR, f"{n}.sk")):
        line = raw.split("#")[0].strip()
        if not line or line.startswith(";") or line.startswith(":") or line.startswith("{") or line[0] in "ZMxyz":
            continue                                  # sequencer / tempo / sample-file lines: not voice state
        def sub(m):
            slot = int(m.group(1))
            if 100 <= slot <= 199:
                subst.append((slot, AMY_SUBST.get(slot, 1)))
                return "w%d" % AMY_SUBST.get(slot, 1)
            return m.group(0)
        line = re.sub(r"w(\d+)", sub, line)
        if re.match(r"^v\d+w\d+/", line):             # (3.sk: `v4w110/a10T` -- a one-shot trigger line)
            continue
        if ref.L.ref_wire(line.encode()) != 0:
            print(f"  patch {n}: line skipped: {line!r}")
    bank, tables = ref.snapshot()
    out = bank.to_arrays("in_")
    out["tables"] = tables
    out["meta"] = np.array(json.dumps({"patch": f"{n}.sk", "amy_slots_moved": sorted(set(subst)), "sample_rate": 44100,
                                       "generator": "tests/golden/gen_patch_voices.py"}))
    os.makedirs(OUT_DIR, exist_ok=True)
    np.savez_compressed(os.path.join(OUT_DIR, f"patch_{n}sk.npz"), **out)
    used = np.where((bank.a["voice_amp"] != 0) & (bank.a["voice_table_size"] > 0))[0]
    print(f"patch {n}.sk: voices in use {used.tolist()}, tables {tables.size} floats, AMY slots moved {sorted(set(subst))}")


if __name__ == "__main__":
    if len(sys.argv) > 1:
        one(int(sys.argv[1]))
    else:
        for n in PATCHES:                             # a fresh process per patch: synth() keeps static state
            subprocess.run([sys.executable, os.path.abspath(__file__), str(n)], check=True)
