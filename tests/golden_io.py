"""Reader for the golden fixtures written by tests/golden/gen_golden.py."""
from __future__ import annotations

import hashlib
import json
import os
from dataclasses import dataclass
from typing import List, Optional

import numpy as np

from skred_amd.bank import RW_FIELDS, GlobalsC, VoiceBank, globals_from_json

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = ["c0_0sk", "c1_sine_adsr64", "c2_mixed_filter64", "c2_notamy64", "c4_pcm_oneshot",
         "edge_basic", "edge_mod", "wav_samples", "bank256_sum", "korg_waves"]
# cases whose voices are independent (no FM/AM/pan/CZ modulators)
MOD_FREE_CASES = ["c1_sine_adsr64", "c2_mixed_filter64", "c2_notamy64", "c4_pcm_oneshot", "edge_basic",
                  "wav_samples", "bank256_sum"]


@dataclass
class Segment:
    index: int
    frames: int
    block: int
    bank_in: VoiceBank
    bank_out_rw: dict            # read-write fields after the segment
    g_in: GlobalsC
    g_out: GlobalsC
    mix: np.ndarray              # [F][2] post-master
    stems_sha256: str            # of float32 [F][64][2]
    stems: Optional[np.ndarray]  # [F][len(stems_voices)][2]
    stems_voices: Optional[np.ndarray]


@dataclass
class Golden:
    name: str
    meta: dict
    tables: np.ndarray
    segments: List[Segment]
    extras: dict                 # case-specific arrays (Case.extra in gen_golden.py), e.g. input / expected files


def load(name: str) -> Golden:
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    segs = []
    for k, sm in enumerate(meta["segments"]):
        p = f"s{k}_"
        segs.append(Segment(
            index=k, frames=sm["frames"], block=sm["block"],
            bank_in=VoiceBank.from_arrays(z, p + "in_"),
            bank_out_rw={f: z[p + "out_" + f] for f in RW_FIELDS},
            g_in=globals_from_json(str(z[p + "globals_in"])),
            g_out=globals_from_json(str(z[p + "globals_out"])),
            mix=z[p + "mix"], stems_sha256=str(z[p + "stems_sha256"]),
            stems=z[p + "stems"] if (p + "stems") in z else None,
            stems_voices=z[p + "stems_voices"] if (p + "stems_voices") in z else None))
    extras = {k[2:]: z[k] for k in z.files if k.startswith("x_")}
    return Golden(name, meta, np.ascontiguousarray(z["tables"], np.float32), segs, extras)


def expected_out_bank(seg: Segment) -> VoiceBank:
    b = seg.bank_in.copy()
    for f, v in seg.bank_out_rw.items():
        b.a[f] = np.ascontiguousarray(v).astype(b.a[f].dtype, copy=True)
    return b


def sha256(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def bits_equal(a: np.ndarray, b: np.ndarray) -> bool:
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and bool((a.view(np.uint32) == b.view(np.uint32)).all())


def bank256_from_parts(g: "Golden"):
    """The 256-voice bank of the bank256_sum case: its four 64-voice snapshots side by side on ONE time base (the
    reference rendered them one after the other, so part k's envelope clocks are shifted back by count_k - count_0).
    Returns (bank, globals at count_0)."""
    segs = g.segments
    n = sum(s.bank_in.n for s in segs)
    b = VoiceBank(n)
    for name in segs[0].bank_in.a:
        b.a[name] = np.ascontiguousarray(np.concatenate([s.bank_in.a[name] for s in segs]))
    c0 = segs[0].g_in.synth_sample_count
    e = b.a["voice_amp_envelope"]
    p = 0
    for s in segs:
        shift = s.g_in.synth_sample_count - c0
        sl = slice(p, p + s.bank_in.n)
        for f in ("sample_start", "sample_release"):
            v = e[f][sl]
            e[f][sl] = np.where(v != 0, v - np.uint64(shift), v)
        p += s.bank_in.n
    gl = segs[0].g_in
    return b, GlobalsC(gl.synth_sample_count, gl.noise_rng, gl.volume_final, gl.volume_smoother_gain,
                       gl.volume_smoother_smoothing, 0.0)
