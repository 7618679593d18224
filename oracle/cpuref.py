"""TEST INFRASTRUCTURE: ctypes loader for oracle/cpu_ref.c (our CPU restatement).

Only tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

from skred_amd.bank import GlobalsC, VoiceBank, VoiceBankC

HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build(fast: bool = False) -> str:
    """Compile cpu_ref.c with the parity flags (or, fast=True, the reference's own -O3 -march=native)."""
    name = "libskred_cpuref_fast.so" if fast else "libskred_cpuref.so"
    # the -march=native build is only valid on the machine that made it: always remake it
    cmd = ["make", "-s", "-C", HERE] + (["-B"] if fast else []) + ["_build/" + name]
    subprocess.run(cmd, check=True)
    return os.path.join(HERE, "_build", name)


def lib(fast: bool = False) -> C.CDLL:
    if fast not in _LIBS:
        L = C.CDLL(build(fast))
        f32p, f64p, vp = C.c_void_p, C.c_void_p, C.c_void_p
        L.skred_cpuref_render.argtypes = [C.POINTER(VoiceBankC), C.POINTER(GlobalsC), vp, C.c_int, C.c_int, f32p, f64p, f32p]
        L.skred_cpuref_render.restype = C.c_int
        L.skred_cpuref_master.argtypes = [C.POINTER(GlobalsC), f32p, C.c_int, C.c_int, f32p]
        L.skred_cpuref_master.restype = C.c_int
        L.skred_cpuref_synth.argtypes = [C.POINTER(VoiceBankC), C.POINTER(GlobalsC), vp, f32p, C.c_int, C.c_int, C.c_int, f32p]
        L.skred_cpuref_synth.restype = C.c_int
        L.skred_cpuref_render_mt.argtypes = [C.POINTER(VoiceBankC), C.POINTER(GlobalsC), vp, C.c_int, C.c_int, C.c_int, f64p]
        L.skred_cpuref_render_mt.restype = C.c_int
        L.skred_cpuref_is_modulation_free.argtypes = [C.POINTER(VoiceBankC)]
        L.skred_cpuref_lcg_next.argtypes = [C.c_uint64]
        L.skred_cpuref_lcg_next.restype = C.c_uint64
        _LIBS[fast] = L
    return _LIBS[fast]


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data


def render(bank: VoiceBank, g: GlobalsC, tables: np.ndarray, frames: int, interp: int = 0,
           want_stems: bool = False, fast: bool = False):
    """Render in place (bank and g advance).  Returns dict(sum32, sum64, stems)."""
    assert tables.dtype == np.float32 and tables.flags["C_CONTIGUOUS"]
    sum32 = np.zeros((frames, 2), np.float32)
    sum64 = np.zeros((frames, 2), np.float64)
    stems = np.zeros((frames, bank.n, 2), np.float32) if want_stems else None
    cb = bank.as_c()
    rc = lib(fast).skred_cpuref_render(C.byref(cb), C.byref(g), tables.ctypes.data, frames, interp,
                                       _ptr(sum32), _ptr(sum64), _ptr(stems))
    if rc != 0:
        raise RuntimeError(f"skred_cpuref_render rc={rc}")
    return {"sum32": sum32, "sum64": sum64, "stems": stems}


def master(g: GlobalsC, total: np.ndarray, channels: int = 2) -> np.ndarray:
    total = np.ascontiguousarray(total, np.float32)
    out = np.zeros((total.shape[0], channels), np.float32)
    rc = lib().skred_cpuref_master(C.byref(g), total.ctypes.data, total.shape[0], channels, out.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"skred_cpuref_master rc={rc}")
    return out


def synth(bank: VoiceBank, g: GlobalsC, tables: np.ndarray, frames: int, channels: int = 2,
          interp: int = 0, want_stems: bool = False):
    """The whole synth() contract (render + master).  Returns (buffer, stems)."""
    buf = np.zeros((frames, channels), np.float32)
    stems = np.zeros((frames, bank.n, 2), np.float32) if want_stems else None
    cb = bank.as_c()
    rc = lib().skred_cpuref_synth(C.byref(cb), C.byref(g), tables.ctypes.data, buf.ctypes.data,
                                  frames, channels, interp, _ptr(stems))
    if rc != 0:
        raise RuntimeError(f"skred_cpuref_synth rc={rc}")
    return buf, stems


def render_mt(bank: VoiceBank, g: GlobalsC, tables: np.ndarray, frames: int, threads: int,
              interp: int = 0, fast: bool = True) -> np.ndarray:
    sum64 = np.zeros((frames, 2), np.float64)
    cb = bank.as_c()
    rc = lib(fast).skred_cpuref_render_mt(C.byref(cb), C.byref(g), tables.ctypes.data, frames, interp,
                                          threads, sum64.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"skred_cpuref_render_mt rc={rc}")
    return sum64


def lcg_advance(state: int, steps: int) -> int:
    L = lib()
    for _ in range(steps):
        state = L.skred_cpuref_lcg_next(state)
    return state


def fx_render(bank, pool: np.ndarray, count: int, frames: int, interp: int = 0, want_stems: bool = False, fast: bool = False):
    """Fixed-point definition (oracle/cpu_ref_fxpt.c).  Returns (mix int64 [F][2], stems int32|None, new count)."""
    from skred_amd.fxbank import FxBankC
    L = lib(fast)
    L.skred_cpuref_fx_render.argtypes = [C.POINTER(FxBankC), C.c_void_p, C.POINTER(C.c_uint64), C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    L.skred_cpuref_fx_render.restype = C.c_int
    pool = np.ascontiguousarray(pool, np.int16)
    mix = np.zeros((frames, 2), np.int64)
    stems = np.zeros((frames, bank.n, 2), np.int32) if want_stems else None
    cnt = C.c_uint64(count)
    cb = bank.as_c()
    rc = L.skred_cpuref_fx_render(C.byref(cb), pool.ctypes.data, C.byref(cnt), frames, interp, mix.ctypes.data, _ptr(stems))
    if rc != 0:
        raise RuntimeError(f"skred_cpuref_fx_render rc={rc}")
    return mix, stems, int(cnt.value)


def fx_master(target_q31: int, k_q15: int, gain_q31: int, mix: np.ndarray):
    """The fixed-point master stage (oracle/cpu_ref_fxpt.c: skred_cpuref_fx_master).  Returns (out int64 [F][2], new gain)."""
    L = lib()
    L.skred_cpuref_fx_master.argtypes = [C.c_int64, C.c_int32, C.POINTER(C.c_int64), C.c_void_p, C.c_int, C.c_void_p]
    L.skred_cpuref_fx_master.restype = C.c_int
    mix = np.ascontiguousarray(mix, np.int64)
    out = np.zeros_like(mix)
    g = C.c_int64(gain_q31)
    rc = L.skred_cpuref_fx_master(target_q31, k_q15, C.byref(g), mix.ctypes.data, mix.shape[0], out.ctypes.data)
    if rc != 0:
        raise RuntimeError(f"skred_cpuref_fx_master rc={rc}")
    return out, int(g.value)
