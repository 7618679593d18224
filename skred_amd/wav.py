"""ctypes binding of include/skred_wav.h: WAV files in (sample tables), stem recorder out.

Plumbing for tests and tools; the code lives in libskred_amd.so (skred_wav.c, skred_recorder.c,
skred_rec_kernels.hip).  The recorder keeps its buffer in HBM and has no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np

from . import device

WAV_SYMBOLS = [
    "skred_wav_get", "skred_wav_get_mem", "skred_wav_free",
    "skred_recorder_create", "skred_recorder_destroy", "skred_recorder_start", "skred_recorder_stop",
    "skred_recorder_recording", "skred_recorder_frames", "skred_recorder_append",
    "skred_recorder_save_wav", "skred_recorder_convert",
]


class WavInfo(C.Structure):
    _fields_ = [("channels", C.c_uint16), ("sample_rate", C.c_uint32), ("bits_per_sample", C.c_uint16),
                ("format_tag", C.c_uint16), ("frames", C.c_uint32)]


_bound = False


def _lib() -> C.CDLL:
    global _bound
    L = device.load()
    if not _bound:
        vp, i32 = C.c_void_p, C.c_int
        fp = C.POINTER(C.c_float)
        L.skred_wav_get.restype = fp
        L.skred_wav_get.argtypes = [C.c_char_p, C.POINTER(i32), C.POINTER(WavInfo), i32]
        L.skred_wav_get_mem.restype = fp
        L.skred_wav_get_mem.argtypes = [vp, C.c_size_t, C.POINTER(i32), C.POINTER(WavInfo), i32]
        L.skred_wav_free.restype = fp
        L.skred_wav_free.argtypes = [fp]
        L.skred_recorder_create.argtypes = [C.POINTER(vp), i32, i32, C.c_long]
        L.skred_recorder_destroy.argtypes = [vp]
        L.skred_recorder_destroy.restype = None
        L.skred_recorder_start.argtypes = [vp, C.c_long]
        L.skred_recorder_stop.argtypes = [vp]
        L.skred_recorder_stop.restype = None
        L.skred_recorder_recording.argtypes = [vp]
        L.skred_recorder_frames.argtypes = [vp]
        L.skred_recorder_frames.restype = C.c_long
        L.skred_recorder_append.argtypes = [vp, vp, i32, vp]
        L.skred_recorder_save_wav.argtypes = [vp, C.c_char_p, C.POINTER(i32), i32]
        L.skred_recorder_convert.argtypes = [vp, C.POINTER(i32), vp, C.c_long]
        L.skred_recorder_convert.restype = C.c_long
        _bound = True
    return L


def wav_get(source, ch: int = -1) -> Optional[Tuple[np.ndarray, WavInfo]]:
    """One channel of a WAV file (path or bytes) as the float table the reference's `:w` would install
    (miniwav.c:103-147); None when it cannot be decoded."""
    L = _lib()
    n, info = C.c_int(0), WavInfo()
    if isinstance(source, (bytes, bytearray, memoryview)):
        b = bytes(source)
        p = L.skred_wav_get_mem(b, len(b), C.byref(n), C.byref(info), ch)
    else:
        p = L.skred_wav_get(str(source).encode(), C.byref(n), C.byref(info), ch)
    if not p:
        return None
    table = np.ctypeslib.as_array(p, shape=(n.value,)).copy()
    L.skred_wav_free(p)
    return table, info


class Recorder:
    """Device-resident stem recorder (skred.c:84-131 + wire.c:94-185 save_wav)."""

    def __init__(self, n_voices: int, capacity_frames: int, device_index: int = 0):
        self.L = _lib()
        self.h = C.c_void_p()
        self.n_voices = n_voices
        device._check(self.L.skred_recorder_create(C.byref(self.h), device_index, n_voices, capacity_frames),
                      "skred_recorder_create")

    def close(self):
        if self.h:
            self.L.skred_recorder_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def start(self, max_frames: int = 0):
        device._check(self.L.skred_recorder_start(self.h, max_frames), "skred_recorder_start")

    def stop(self):
        self.L.skred_recorder_stop(self.h)

    @property
    def recording(self) -> bool:
        return bool(self.L.skred_recorder_recording(self.h))

    @property
    def frames(self) -> int:
        return int(self.L.skred_recorder_frames(self.h))

    def append(self, d_stems: int, frames: int, stream: int = 0):
        device._check(self.L.skred_recorder_append(self.h, d_stems, frames, stream), "skred_recorder_append")

    def _mask(self, record: Sequence[int]):
        m = np.zeros(self.n_voices, np.int32)
        m[:len(record)] = np.asarray(record, np.int32)
        return m

    def save_wav(self, path: str, record: Sequence[int], sample_rate: int = 44100):
        m = self._mask(record)
        device._check(self.L.skred_recorder_save_wav(self.h, str(path).encode(),
                                                     m.ctypes.data_as(C.POINTER(C.c_int)), sample_rate),
                      "skred_recorder_save_wav")

    def convert(self, record: Sequence[int]) -> np.ndarray:
        m = self._mask(record)
        out = np.zeros(self.frames * int((m != 0).sum()) * 2, np.int16)
        n = self.L.skred_recorder_convert(self.h, m.ctypes.data_as(C.POINTER(C.c_int)), out.ctypes.data, out.size)
        if n < 0:
            device._check(int(n), "skred_recorder_convert")
        return out[:n]
