#!/usr/bin/env python3
"""Extract the numbers of the reference's Korg DW-8000 single-cycle waves into package data.

  /root/reference/retro/korg.h        which wave file feeds which kwave[] entry (include order)
  /root/reference/retro/*.w0-3        comma-separated int16 samples (data files, no code)

-> skred_amd/data/korg_waves.bin      (numbers only; no text of the reference is kept)

The reference's wave_table_init() (synth.c:1251-1268) turns kwave[0..30] into the float tables of
wave slots 32..62: 2048 entries each (kwave_size, retro/korg.h:219-222), value int16/32767,
rate MAIN_SAMPLE_RATE, not one-shot, loop 0..size-1.  The drop-in library does the same from this
blob (skred_amd/csrc/skred_synth_dropin.c: wave_table_init).

File layout (little endian):  char magic[8] = "SKKORG1\\0";  uint32 n_waves;  uint32 size[n_waves];
int16 samples[sum(size)].

When oracle/_ref/libskred_ref.so (the compiled, unmodified reference) is present, the script checks
that int16/32767 reproduces the reference's own float tables bit for bit.
"""
import ctypes as C
import os
import re
import struct
import sys

import numpy as np

REF = os.environ.get("SKRED_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
FIRST_SLOT, END_SLOT = 32, 63        # WAVE_TABLE_KRG1, WAVE_TABLE_KRG32 (skred.h:33,64); the loop stops before END_SLOT
WAVE_LEN = 2048                      # kwave_size[], retro/korg.h:219-222


def main():
    text = open(os.path.join(REF, "retro", "korg.h")).read()
    order = re.findall(r"int16_t\s+kw(\d+)\[\]\s*=\s*\{\s*#include\s+\"([^\"]+)\"", text)
    files = {int(k): f for k, f in order}
    n = END_SLOT - FIRST_SLOT
    waves = []
    for k in range(n):
        body = open(os.path.join(REF, "retro", files[k])).read()
        body = re.sub(r"//[^\n]*", "", body)
        vals = np.array([int(x) for x in re.split(r"[,\s]+", body.strip()) if x], np.int64)
        assert len(vals) >= WAVE_LEN, (files[k], len(vals))
        assert vals.min() >= -32768 and vals.max() <= 32767
        waves.append(vals[:WAVE_LEN].astype("<i2"))
    path = os.path.join(HERE, "korg_waves.bin")
    with open(path, "wb") as f:
        f.write(b"SKKORG1\0")
        f.write(struct.pack("<I", n))
        f.write(struct.pack("<%dI" % n, *[len(w) for w in waves]))
        for w in waves:
            f.write(w.tobytes())
    print(f"wrote {path}: {n} waves x {WAVE_LEN}, {os.path.getsize(path)} bytes")

    so = os.path.join(ROOT, "oracle", "_ref", "libskred_ref.so")
    if os.path.exists(so):
        L = C.CDLL(so)
        L.wave_table_init()
        data = (C.POINTER(C.c_float) * 1200).in_dll(L, "wave_table_data")
        size = (C.c_int * 1200).in_dll(L, "wave_size")
        for k, w in enumerate(waves):
            slot = FIRST_SLOT + k
            assert size[slot] == len(w), (slot, size[slot])
            ref = np.ctypeslib.as_array(data[slot], shape=(len(w),))
            mine = (w.astype(np.float32) / np.float32(32767)).astype(np.float32)
            assert (ref.view(np.uint32) == mine.view(np.uint32)).all(), f"slot {slot} differs from the reference"
        print(f"checked against the compiled reference: slots {FIRST_SLOT}..{END_SLOT - 1} bit-identical")


if __name__ == "__main__":
    sys.exit(main())
