"""Fixed-point voice bank: numpy mirror + ctypes binding of include/skred_amd_fxpt.h.

The fixed-point path is defined by this project (oracle/cpu_ref_fxpt.c), not by the reference,
which has none."""
from __future__ import annotations

import ctypes as C
import json
from typing import Dict, Optional, Tuple

import numpy as np

from . import banks
from .device import SkredAmdError, _check, load

U32, I32, U64 = np.dtype("<u4"), np.dtype("<i4"), np.dtype("<u8")
FX_FIELDS = [
    ("phase", U32, True), ("phase_inc", U32, False), ("table_offset", I32, False), ("log2_size", I32, False),
    ("amp_q15", I32, False), ("pan_left_q15", I32, False), ("pan_right_q15", I32, False),
    ("disconnect", I32, False), ("use_envelope", I32, False),
    ("attack_frames", U32, False), ("decay_frames", U32, False), ("release_frames", U32, False),
    ("sustain_q15", I32, False), ("velocity_q15", I32, False),
    ("sample_start", U64, False), ("sample_release", U64, False),
    ("is_active", I32, True), ("smoother_enable", I32, False), ("smoother_k_q15", I32, False),
    ("smoother_gain_q15", I32, True), ("voice_sample", I32, True),
    ("one_shot", I32, False), ("finished", I32, True), ("filter_mode", I32, False),
    ("b0_q30", I32, False), ("b1_q30", I32, False), ("b2_q30", I32, False), ("a1_q30", I32, False), ("a2_q30", I32, False),
    ("x1", I32, True), ("x2", I32, True), ("y1", I32, True), ("y2", I32, True),
]
FX_RW = [f[0] for f in FX_FIELDS if f[2]]
FX_ABI_SYMBOLS = ["skred_fxbank_create", "skred_fxbank_destroy", "skred_fxbank_set_tables_i16",
                  "skred_fxbank_upload", "skred_fxbank_download", "skred_fxbank_set_sample_count",
                  "skred_fxbank_get_sample_count", "skred_fxbank_render", "skred_fxbank_render_host",
                  "skred_fxbank_last_render_ms", "skred_fxbank_render_mix", "skred_fxbank_master", "skred_fxbank_set_master",
                  "skred_fxbank_get_master_gain", "skred_fxbank_stamp",
                  "skred_fxshard_create", "skred_fxshard_bank", "skred_fxshard_upload"]
FX_STAMP_TRIGGER, FX_STAMP_RELEASE = 1, 2
MASTER_TARGET_Q31 = int(0.025 * 2147483648.0)     # the library's default: volume_user 1 x AMY_FACTOR
MASTER_K_Q15 = 66                                 # 0.002


class FxBankC(C.Structure):
    _fields_ = [("n_voices", C.c_int32)] + [(n, C.c_void_p) for n, _, _ in FX_FIELDS]


class FxVoiceBank:
    def __init__(self, n: int):
        self.n = int(n)
        self.a: Dict[str, np.ndarray] = {name: np.zeros(self.n, dt) for name, dt, _ in FX_FIELDS}
        self.a["log2_size"][:] = 3
        self.a["pan_left_q15"][:] = 16384
        self.a["pan_right_q15"][:] = 16384
        self.a["velocity_q15"][:] = 32768
        self.a["sustain_q15"][:] = 32768

    def __getitem__(self, k):
        return self.a[k]

    def __setitem__(self, k, v):
        self.a[k][...] = v

    def as_c(self) -> FxBankC:
        c = FxBankC()
        c.n_voices = self.n
        for name, dt, _ in FX_FIELDS:
            arr = self.a[name]
            assert arr.dtype == dt and arr.flags["C_CONTIGUOUS"] and arr.shape == (self.n,), name
            setattr(c, name, arr.ctypes.data)
        return c

    def copy(self) -> "FxVoiceBank":
        o = FxVoiceBank.__new__(FxVoiceBank)
        o.n, o.a = self.n, {k: v.copy() for k, v in self.a.items()}
        return o

    def take(self, index) -> "FxVoiceBank":
        idx = np.arange(self.n)[index]
        o = FxVoiceBank.__new__(FxVoiceBank)
        o.n, o.a = int(idx.size), {k: np.ascontiguousarray(v[idx]) for k, v in self.a.items()}
        return o

    def rw_mismatch(self, other: "FxVoiceBank") -> Dict[str, int]:
        return {k: int((self.a[k] != other.a[k]).sum()) for k in FX_RW if (self.a[k] != other.a[k]).any()}


def _bind(L):
    vp, i32 = C.c_void_p, C.c_int
    L.skred_fxbank_create.argtypes = [i32, i32, C.POINTER(vp)]
    L.skred_fxbank_destroy.argtypes = [vp]
    L.skred_fxbank_destroy.restype = None
    L.skred_fxbank_set_tables_i16.argtypes = [vp, vp, C.c_size_t]
    L.skred_fxbank_upload.argtypes = [vp, C.POINTER(FxBankC), i32, i32, i32]
    L.skred_fxbank_download.argtypes = [vp, C.POINTER(FxBankC), i32, i32, i32]
    L.skred_fxbank_set_sample_count.argtypes = [vp, C.c_uint64]
    L.skred_fxbank_get_sample_count.argtypes = [vp]
    L.skred_fxbank_get_sample_count.restype = C.c_uint64
    L.skred_fxbank_render.argtypes = [vp, i32, i32, vp, vp, vp]
    L.skred_fxbank_render_host.argtypes = [vp, i32, i32, vp, vp]
    L.skred_fxbank_last_render_ms.argtypes = [vp]
    L.skred_fxbank_last_render_ms.restype = C.c_float
    L.skred_fxbank_render_mix.argtypes = [vp, i32, i32, vp, vp, vp]
    L.skred_fxbank_master.argtypes = [vp, vp, i32, vp, vp]
    L.skred_fxbank_set_master.argtypes = [vp, C.c_int64, C.c_int32, C.c_int64]
    L.skred_fxbank_get_master_gain.argtypes = [vp]
    L.skred_fxbank_get_master_gain.restype = C.c_int64
    L.skred_fxbank_stamp.argtypes = [vp, vp, i32, i32, vp]
    L.skred_fxshard_create.argtypes = [i32, i32, i32, i32, i32, C.POINTER(vp)]
    L.skred_fxshard_bank.argtypes = [vp]
    L.skred_fxshard_bank.restype = vp
    L.skred_fxshard_upload.argtypes = [vp, C.POINTER(FxBankC)]
    return L


class DeviceFxBank:
    def __init__(self, n_voices: int, device: int = 0):
        self.L = _bind(load())
        self.n = int(n_voices)
        h = C.c_void_p()
        _check(self.L.skred_fxbank_create(device, self.n, C.byref(h)), "skred_fxbank_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.skred_fxbank_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_tables(self, pool: np.ndarray):
        pool = np.ascontiguousarray(pool, np.int16)
        _check(self.L.skred_fxbank_set_tables_i16(self.h, pool.ctypes.data, pool.size), "skred_fxbank_set_tables_i16")

    @classmethod
    def borrowed(cls, handle, n_voices: int) -> "DeviceFxBank":
        """A view of a bank a shard owns (skred_fxshard_bank): close() does not destroy it."""
        self = cls.__new__(cls)
        self.L = _bind(load())
        self.n = int(n_voices)
        self.h = C.c_void_p(handle)
        self.close = lambda: None
        return self

    def upload(self, bank: FxVoiceBank):
        cb = bank.as_c()
        _check(self.L.skred_fxbank_upload(self.h, C.byref(cb), 0, 0, bank.n), "skred_fxbank_upload")

    def download(self, bank: FxVoiceBank):
        cb = bank.as_c()
        _check(self.L.skred_fxbank_download(self.h, C.byref(cb), 0, 0, bank.n), "skred_fxbank_download")

    def set_sample_count(self, c: int):
        _check(self.L.skred_fxbank_set_sample_count(self.h, c), "skred_fxbank_set_sample_count")

    def sample_count(self) -> int:
        return int(self.L.skred_fxbank_get_sample_count(self.h))

    def render(self, frames: int, d_mix: int, interp: int = 0, d_stems: int = 0, stream: int = 0):
        _check(self.L.skred_fxbank_render(self.h, frames, interp, d_mix, d_stems or None, stream or None), "skred_fxbank_render")

    def render_host(self, frames: int, interp: int = 0, want_stems: bool = False):
        mix = np.zeros((frames, 2), np.int64)
        stems = np.zeros((frames, self.n, 2), np.int32) if want_stems else None
        _check(self.L.skred_fxbank_render_host(self.h, frames, interp, mix.ctypes.data,
                                               stems.ctypes.data if want_stems else None), "skred_fxbank_render_host")
        return mix, stems

    def last_render_ms(self) -> float:
        return float(self.L.skred_fxbank_last_render_ms(self.h))

    def render_mix(self, frames: int, d_out: int, interp: int = 0, d_stems: int = 0, stream: int = 0):
        """Render + mix-down + master stage in one launch; d_out: device int64 [frames][2]."""
        _check(self.L.skred_fxbank_render_mix(self.h, frames, interp, d_out, d_stems or None, stream or None), "skred_fxbank_render_mix")

    def master(self, d_sum: int, frames: int, d_out: int, stream: int = 0):
        _check(self.L.skred_fxbank_master(self.h, d_sum, frames, d_out, stream or None), "skred_fxbank_master")

    def set_master(self, target_q31: int = MASTER_TARGET_Q31, k_q15: int = MASTER_K_Q15, gain_q31: int = 0):
        _check(self.L.skred_fxbank_set_master(self.h, target_q31, k_q15, gain_q31), "skred_fxbank_set_master")

    def master_gain(self) -> int:
        return int(self.L.skred_fxbank_get_master_gain(self.h))

    def stamp(self, voices, which: int, stream: int = 0):
        v = np.ascontiguousarray(voices, np.int32)
        _check(self.L.skred_fxbank_stamp(self.h, v.ctypes.data, len(v), which, stream or None), "skred_fxbank_stamp")


# ------------------------------------------------------------------ synthetic fixed-point bank

def fx_lut_pool() -> Tuple[np.ndarray, dict]:
    """All notamy int16 LUTs concatenated; returns (pool, {name: (offset, log2_size, highest_harmonic)})."""
    z, names, meta = banks.load_luts()
    info, pos, parts = {}, 0, []
    for nm in names:
        t = z["i16_" + nm]
        info[nm] = (pos, meta[nm]["log2_size"], meta[nm]["highest_harmonic"])
        parts.append(t)
        pos += len(t)
    return np.concatenate(parts).astype(np.int16), info


# what bench.py's `fixed_point` leg says about itself
DTYPE_NOTE = "q15 gains / u32 phase / q2.30 x q12 biquad / i64 mix"
WORKLOAD_NOTE = ("fixed-point analogue of the C2 recipe (fxbank.bank_fx): int16 LUT pyramids, linear interpolation, "
                 "per-voice biquad (modes 1-4), ADSR, amp smoother")


def q30_coeffs(mode, freq, q, sample_rate) -> Dict[str, np.ndarray]:
    """RBJ coefficients (banks.biquad_coeffs == mmf_set_params, synth.c:929-1008) rounded to Q2.30."""
    co = banks.biquad_coeffs(mode, freq, q, sample_rate)
    return {k + "_q30": np.clip(np.round(co[k].astype(np.float64) * (1 << 30)), -(1 << 31), (1 << 31) - 1).astype(np.int32)
            for k in ("b0", "b1", "b2", "a1", "a2")}


def bank_fx(n: int = 65536, sample_rate: int = 48000, seed: int = banks.SEED, with_filter: bool = True):
    """Fixed-point analogue of the C2 recipe: v mod 3 -> sine / triangle / impulse int16 pyramids (level by frequency),
    biquad mode 1 + v mod 4 with K ~ logU[100, 8000] Hz and Q ~ U[0.5, 4] (the float recipe's draws), ADSR
    0.01/0.1/0.7/0.2 s, staggered note-ons, smoother k=0.02."""
    pool, info = fx_lut_pool()
    u5 = banks.lcg_uniform(5 * n, seed).reshape(5, n)
    u = u5[:3]
    freq = (np.float32(27.5) * np.exp2(np.float32(7.0) * u[0])).astype(np.float64)
    b = FxVoiceBank(n)
    fam = np.arange(n) % 3
    off = np.zeros(n, np.int32)
    lg = np.zeros(n, np.int32)
    for f_id, family in enumerate(("sine", "triangle", "impulse")):
        levels = [k for k in info if k.startswith(family + "_")]
        hh = np.array([info[k][2] for k in levels], np.float64)
        sel = np.where(fam == f_id)[0]
        ok = hh[None, :] * freq[sel, None] <= sample_rate / 2.0
        lvl = np.where(ok.any(1), ok.argmax(1), len(levels) - 1)
        off[sel] = np.array([info[levels[k]][0] for k in lvl], np.int32)
        lg[sel] = np.array([info[levels[k]][1] for k in lvl], np.int32)
    b["table_offset"], b["log2_size"] = off, lg
    b["phase_inc"] = np.round(freq / sample_rate * 4294967296.0).astype(np.uint64).astype(np.uint32)
    b["phase"] = (u[2].astype(np.float64) * 4294967295.0).astype(np.uint64).astype(np.uint32)
    b["amp_q15"] = 32768
    pan = u[1].astype(np.float64) * 2.0 - 1.0
    b["pan_left_q15"] = np.round((1.0 - pan) / 2.0 * 32768).astype(np.int32)
    b["pan_right_q15"] = np.round((1.0 + pan) / 2.0 * 32768).astype(np.int32)
    b["use_envelope"] = 1
    b["attack_frames"] = int(0.01 * sample_rate)
    b["decay_frames"] = int(0.1 * sample_rate)
    b["release_frames"] = int(0.2 * sample_rate)
    b["sustain_q15"] = int(0.7 * 32768)
    count0 = 2 * sample_rate
    stagger = np.minimum((u[2] * np.float32(sample_rate)).astype(np.int64), sample_rate - 1)
    b["sample_start"] = (count0 - stagger).astype(np.uint64)
    b["is_active"] = 1
    b["smoother_enable"] = 1
    b["smoother_k_q15"] = int(round(0.02 * 32768))
    if with_filter:
        mode = (1 + np.arange(n) % 4).astype(np.int32)
        cutoff = (np.float32(100.0) * np.power(np.float32(80.0), u5[3])).astype(np.float32)
        q = (np.float32(0.5) + np.float32(3.5) * u5[4]).astype(np.float32)
        for k, v in q30_coeffs(mode, cutoff, q, sample_rate).items():
            b[k] = v
        b["filter_mode"] = mode
    return b, pool, count0
