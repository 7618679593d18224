#!/usr/bin/env python3
"""Extract the numeric contents of the reference's data-only LUT headers into a fixture.

  /root/reference/notamy/{sine,triangle,impulse}_lutset.h        float pyramids
  /root/reference/notamy/{sine,triangle,impulse}_lutset_fxpt.h   int16 pyramids + log2 size + scale

-> skred_amd/data/notamy_luts.npz  (arrays only; no header text is kept)

These band-limited single-cycle tables (from shorepine/AMY) are INPUT DATA of the hot path
(north_star: "wavetable lookup into the notamy/ sine/triangle/impulse LUTs"); no reference C file
includes them, so there is no reference behaviour attached to them beyond the
numbers themselves.  Keys:  f32_<name>, i16_<name>, JSON `names`, `meta` (table_size,
highest_harmonic, log2_size, scale_factor per table), and `pcm_map` [67][5] = (offset, length,
loopstart, loopend, midinote) of the AMY PCM regions.
"""
import json
import os
import re
import sys

import numpy as np

REF = os.environ.get("SKRED_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def arrays(text, ctype):
    out = {}
    for m in re.finditer(r"const\s+%s\s+(\w+)\[(\d+)\][^=]*=\s*\{([^}]*)\}" % ctype, text):
        name, n, body = m.group(1), int(m.group(2)), m.group(3)
        vals = [x for x in re.split(r"[,\s]+", body.strip()) if x]
        assert len(vals) == n, (name, n, len(vals))
        out[name] = vals
    return out


def entries(text, set_name):
    m = re.search(r"%s\[\d+\]\s*=\s*\{(.*?)\n\};" % set_name, text, re.S)
    rows = re.findall(r"\{([^{}]*)\}", m.group(1))
    return [[c.strip() for c in r.split(",")] for r in rows if not r.strip().startswith("NULL")]


def main():
    out, names, meta = {}, [], {}
    for fam in ("sine", "triangle", "impulse"):
        ftxt = open(os.path.join(REF, "notamy", f"{fam}_lutset.h")).read()
        xtxt = open(os.path.join(REF, "notamy", f"{fam}_lutset_fxpt.h")).read()
        fa, xa = arrays(ftxt, "float"), arrays(xtxt, "int16_t")
        fe, xe = entries(ftxt, f"{fam}_lutset"), entries(xtxt, f"{fam}_fxpt_lutset")
        assert len(fe) == len(xe), fam
        for i, (f, x) in enumerate(zip(fe, xe)):
            nm = f"{fam}_{i}"
            tf = np.array([float(v) for v in fa[f[0]]], np.float32)
            ti = np.array([int(v) for v in xa[x[0]]], np.int16)
            assert len(tf) == int(f[1]) == len(ti) == int(x[1]) == 1 << int(x[2]), nm
            out["f32_" + nm], out["i16_" + nm] = tf, ti
            names.append(nm)
            meta[nm] = {"table_size": int(f[1]), "highest_harmonic": int(f[2]),
                        "log2_size": int(x[2]), "scale_factor": float(x[4])}
    # pcm_map geometry (offset, length, loopstart, loopend, midinote), notamy/pcm_large.h:10-78; the sample
    # blob itself (notamy/pcm_samples_large.h) is absent from the mount, so only the geometry exists.
    ptxt = open(os.path.join(REF, "notamy", "pcm_large.h")).read()
    rows = re.findall(r"\{\s*(\d+)\s*,\s*(\d+)\s*,\s*(\d+)\s*,\s*(\d+)\s*,\s*(?:/\*.*?\*/)?\s*(\d+)\s*\}", ptxt)
    pm = np.array([[int(x) for x in r] for r in rows], np.int64)
    assert pm.shape == (67, 5), pm.shape
    assert int(pm[-1, 0] + pm[-1, 1]) == 1176036, "PCM_LENGTH mismatch"
    out["pcm_map"] = pm
    out["names"] = np.array(json.dumps(names))
    out["meta"] = np.array(json.dumps(meta))
    path = os.path.join(HERE, "notamy_luts.npz")
    np.savez_compressed(path, **out)
    total = sum(meta[n]["table_size"] for n in names)
    print(f"wrote {path}: {len(names)} tables, {total} entries, {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    sys.exit(main())
