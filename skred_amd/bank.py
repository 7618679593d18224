"""Host-side voice bank: numpy structure-of-arrays mirror of ``skred_voice_bank_t``.

Field names, order and types follow ``include/skred_amd.h`` (which in turn keeps the
names of the reference's per-voice arrays, synth.def:12-89, and the struct layouts of
synth-types.h:13-38).  The ctypes structures built here are what crosses the C ABI;
numpy is only the owner of the memory.
"""
from __future__ import annotations

import ctypes as C
import json
from typing import Dict, Iterable, Optional

import numpy as np

# == skred_mmf_t (synth-types.h:13-23), 48 bytes
MMF_DTYPE = np.dtype([
    ("x1", "<f4"), ("x2", "<f4"), ("y1", "<f4"), ("y2", "<f4"),
    ("b0", "<f4"), ("b1", "<f4"), ("b2", "<f4"), ("a1", "<f4"), ("a2", "<f4"),
    ("last_freq", "<f4"), ("last_resonance", "<f4"), ("last_mode", "<i4"),
])
# == skred_envelope_t (synth-types.h:25-38), 56 bytes
ENV_DTYPE = np.dtype([
    ("a", "<f4"), ("d", "<f4"), ("s", "<f4"), ("r", "<f4"),
    ("attack_time", "<f4"), ("decay_time", "<f4"), ("sustain_level", "<f4"), ("release_time", "<f4"),
    ("sample_start", "<u8"), ("sample_release", "<u8"),
    ("is_active", "<i4"), ("velocity", "<f4"),
])
assert MMF_DTYPE.itemsize == 48 and ENV_DTYPE.itemsize == 56

F32, I32, I64 = np.dtype("<f4"), np.dtype("<i4"), np.dtype("<i8")

# (name, dtype, read-write?) in the exact member order of skred_voice_bank_t
FIELDS = [
    ("voice_phase", F32, True),
    ("voice_phase_inc", F32, False),
    ("voice_table_offset", I64, False),
    ("voice_table_size", I32, False),
    ("voice_one_shot", I32, False),
    ("voice_finished", I32, True),
    ("voice_loop_enabled", I32, False),
    ("voice_loop_valid", I32, False),
    ("voice_loop_start_f", F32, False),
    ("voice_loop_end_f", F32, False),
    ("voice_direction", I32, False),
    ("voice_wave_table_index", I32, False),
    ("voice_sample", F32, True),
    ("voice_sample_hold", F32, True),
    ("voice_sample_hold_count", I32, True),
    ("voice_sample_hold_max", I32, False),
    ("voice_quantize", I32, False),
    ("voice_amp", F32, False),
    ("voice_use_amp_envelope", I32, False),
    ("voice_smoother_enable", I32, False),
    ("voice_smoother_gain", F32, True),
    ("voice_smoother_smoothing", F32, False),
    ("voice_filter_mode", I32, False),
    ("voice_filter", MMF_DTYPE, True),
    ("voice_amp_envelope", ENV_DTYPE, True),
    ("voice_pan_left", F32, True),
    ("voice_pan_right", F32, True),
    ("voice_disconnect", I32, False),
    ("voice_freq_mod_osc", I32, False),
    ("voice_freq_mod_depth", F32, False),
    ("voice_freq_scale", F32, False),
    ("voice_amp_mod_osc", I32, False),
    ("voice_amp_mod_depth", F32, False),
    ("voice_pan_mod_osc", I32, False),
    ("voice_pan_mod_depth", F32, False),
    ("voice_cz_mod_osc", I32, False),
    ("voice_cz_mod_depth", F32, False),
    ("voice_cz_mode", I32, False),
    ("voice_cz_distortion", F32, False),
]
FIELD_NAMES = [f[0] for f in FIELDS]
RW_FIELDS = [f[0] for f in FIELDS if f[2]]


class VoiceBankC(C.Structure):
    """ctypes image of ``skred_voice_bank_t``."""
    _fields_ = [("n_voices", C.c_int32)] + [(name, C.c_void_p) for name, _, _ in FIELDS]


class GlobalsC(C.Structure):
    """ctypes image of ``skred_globals_t``."""
    _fields_ = [
        ("synth_sample_count", C.c_uint64),
        ("noise_rng", C.c_uint64),
        ("volume_final", C.c_float),
        ("volume_smoother_gain", C.c_float),
        ("volume_smoother_smoothing", C.c_float),
        ("reserved", C.c_float),
    ]

    @classmethod
    def defaults(cls) -> "GlobalsC":
        # synth.c:85-92 (volume_user 1.0 * AMY_FACTOR 0.025, skred.h:11), LCG seeded 1 (synth.c:508)
        return cls(0, 1, np.float32(0.025), np.float32(0.0), np.float32(0.002), 0.0)

    def copy(self) -> "GlobalsC":
        return GlobalsC(self.synth_sample_count, self.noise_rng, self.volume_final,
                        self.volume_smoother_gain, self.volume_smoother_smoothing, 0.0)

    def to_dict(self) -> dict:
        return {n: getattr(self, n) for n, _ in self._fields_ if n != "reserved"}


class VoiceBank:
    """N voices, one numpy array per reference field.  Defaults follow voice_reset (synth.c:1090-1132)
    for everything that does not need a table: silent (amp 0), centre pan, smoother on (k=0.02),
    flat envelope, no modulators."""

    def __init__(self, n_voices: int):
        self.n = int(n_voices)
        self.a: Dict[str, np.ndarray] = {name: np.zeros(self.n, dtype=dt) for name, dt, _ in FIELDS}
        a = self.a
        a["voice_pan_left"][:] = 0.5
        a["voice_pan_right"][:] = 0.5
        for k in ("voice_freq_mod_osc", "voice_amp_mod_osc", "voice_pan_mod_osc"):
            a[k][:] = -1
        a["voice_freq_scale"][:] = 1.0
        a["voice_smoother_enable"][:] = 1
        a["voice_smoother_smoothing"][:] = 0.02
        a["voice_amp_envelope"]["s"][:] = 1.0
        a["voice_amp_envelope"]["sustain_level"][:] = 1.0

    def __getitem__(self, name: str) -> np.ndarray:
        return self.a[name]

    def __setitem__(self, name: str, value) -> None:
        self.a[name][...] = value

    def as_c(self) -> VoiceBankC:
        """ctypes view; valid while this VoiceBank is alive and its arrays are not re-bound."""
        c = VoiceBankC()
        c.n_voices = self.n
        for name, dt, _ in FIELDS:
            arr = self.a[name]
            assert arr.dtype == dt and arr.flags["C_CONTIGUOUS"] and arr.shape == (self.n,), name
            setattr(c, name, arr.ctypes.data)
        return c

    def copy(self) -> "VoiceBank":
        o = VoiceBank.__new__(VoiceBank)
        o.n = self.n
        o.a = {k: v.copy() for k, v in self.a.items()}
        return o

    def take(self, index) -> "VoiceBank":
        """New bank holding the voices selected by `index` (slice or index array)."""
        idx = np.arange(self.n)[index]
        o = VoiceBank.__new__(VoiceBank)
        o.n = int(idx.size)
        o.a = {k: np.ascontiguousarray(v[idx]) for k, v in self.a.items()}
        return o

    def modulation_free(self) -> bool:
        a = self.a
        return bool((a["voice_freq_mod_osc"] < 0).all() and (a["voice_amp_mod_osc"] < 0).all()
                    and (a["voice_pan_mod_osc"] < 0).all()
                    and ((a["voice_cz_mode"] == 0) | (a["voice_cz_mod_osc"] < 0)).all())

    # ---- (de)serialisation used by the golden fixtures -------------------
    def to_arrays(self, prefix: str, only: Optional[Iterable[str]] = None) -> Dict[str, np.ndarray]:
        names = FIELD_NAMES if only is None else list(only)
        return {prefix + k: self.a[k] for k in names}

    @classmethod
    def from_arrays(cls, arrays, prefix: str) -> "VoiceBank":
        n = int(arrays[prefix + "voice_phase"].shape[0])
        o = cls(n)
        for name, dt, _ in FIELDS:
            o.a[name] = np.ascontiguousarray(arrays[prefix + name]).astype(dt, copy=True)
        return o

    def rw_equal(self, other: "VoiceBank") -> Dict[str, int]:
        """Bitwise comparison of the read-write fields; returns {field: mismatching voices}."""
        bad = {}
        for k in RW_FIELDS:
            x, y = self.a[k], other.a[k]
            if k == "voice_filter":
                m = 0
                for sub in ("x1", "x2", "y1", "y2"):
                    m += int((x[sub].view("<u4") != y[sub].view("<u4")).sum())
            elif k == "voice_amp_envelope":
                m = int((x["is_active"] != y["is_active"]).sum())
            elif x.dtype == F32:
                m = int((x.view("<u4") != y.view("<u4")).sum())
            else:
                m = int((x != y).sum())
            if m:
                bad[k] = m
        return bad


def globals_from_json(s: str) -> GlobalsC:
    d = json.loads(s) if isinstance(s, str) else dict(s)
    return GlobalsC(int(d["synth_sample_count"]), int(d["noise_rng"]),
                    float(d["volume_final"]), float(d["volume_smoother_gain"]),
                    float(d["volume_smoother_smoothing"]), 0.0)
