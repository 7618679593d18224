"""ctypes binding of libskred_amd.so (the C ABI in include/skred_amd.h).

This is plumbing for tests and bench.py; the product is the shared library.  There is no
fallback of any kind: a missing library or a machine without a usable GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import numpy as np

from .bank import GlobalsC, VoiceBank, VoiceBankC

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SKRED_AMD_LIB", os.path.join(_HERE, "libskred_amd.so"))   # override: A/B builds

# every symbol include/skred_amd.h declares
ABI_SYMBOLS = [
    "skred_amd_abi_version", "skred_amd_device_count", "skred_amd_last_error",
    "skred_bank_create", "skred_bank_destroy", "skred_bank_n_voices",
    "skred_bank_set_tables_f32", "skred_bank_upload", "skred_bank_download",
    "skred_bank_set_globals", "skred_bank_get_globals",
    "skred_bank_render", "skred_bank_master", "skred_bank_render_mix", "skred_bank_render_host",
    "skred_bank_last_render_ms", "skred_bank_timing_reset", "skred_bank_timing_summary",
    "skred_bank_set_option", "skred_bank_last_kernel", "skred_bank_last_in_place", "skred_bank_last_split", "skred_bank_last_pack", "skred_bank_list_violations", "skred_bank_set_probe",
    "skred_bank_update", "skred_bank_defer", "skred_bank_run_queue", "skred_bank_queue_pending",
    "skred_shard_partition", "skred_shard_cut_ok", "skred_shard_create", "skred_shard_create_custom", "skred_shard_destroy",
    "skred_shard_bank", "skred_shard_range", "skred_shard_upload", "skred_shard_set_ops", "skred_shard_rccl_unique_id",
    "skred_shard_init_rccl", "skred_shard_render_mix", "skred_shard_render_mix_pipelined", "skred_shard_flush",
    "skred_seq_create", "skred_seq_destroy", "skred_seq_tempo_set", "skred_seq_time_per_step", "skred_seq_step_set",
    "skred_seq_mute_set", "skred_seq_modulo_set", "skred_seq_state_set", "skred_seq_pattern_reset", "skred_seq_pointer",
    "skred_seq_counter", "skred_seq_tick",
    "skred_bank_seq", "skred_bank_set_sample_rate", "skred_bank_pattern_step_set", "skred_bank_pattern_step_clear",
]

# SKRED_DIRTY_* / SKRED_STAMP_* of include/skred_amd.h
DIRTY_PARAMS, DIRTY_PHASE, DIRTY_ENV_STATE, DIRTY_PAN = 1, 2, 4, 8
DIRTY_FILTER_STATE, DIRTY_SMOOTHER, DIRTY_HOLD, DIRTY_SAMPLE = 16, 32, 64, 128
STAMP_TRIGGER, STAMP_RELEASE, DIRTY_ENV_CLOCK = 256, 512, 1024

_lib: Optional[C.CDLL] = None


class SkredAmdError(RuntimeError):
    pass


def load() -> C.CDLL:
    """dlopen the HIP library; raise loudly when it has not been built (python __graft_entry__.py)."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process.  PyTorch bundles its own libamdhip64 and asks for it by the unversioned name,
    # which the loader does not match against an already loaded /opt/rocm copy (same SONAME, different request):
    # if this library came first, torch would bring a second runtime that finds no GPU.  Loading torch first makes
    # this library resolve its libamdhip64.so.7 to the copy torch loaded.  (C hosts link one runtime and never see this.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        raise SkredAmdError(f"{LIB_PATH} is missing: build it with `make -C skred_amd/csrc` "
                            "(there is no CPU fallback for the render path)")
    L = C.CDLL(LIB_PATH)
    vp, i32 = C.c_void_p, C.c_int
    L.skred_amd_abi_version.restype = i32
    L.skred_amd_device_count.restype = i32
    L.skred_amd_last_error.restype = C.c_char_p
    L.skred_bank_create.argtypes = [i32, i32, C.POINTER(vp)]
    L.skred_bank_destroy.argtypes = [vp]
    L.skred_bank_destroy.restype = None
    L.skred_bank_n_voices.argtypes = [vp]
    L.skred_bank_set_tables_f32.argtypes = [vp, vp, C.c_size_t]
    L.skred_bank_upload.argtypes = [vp, C.POINTER(VoiceBankC), i32, i32, i32]
    L.skred_bank_download.argtypes = [vp, C.POINTER(VoiceBankC), i32, i32, i32]
    L.skred_bank_set_globals.argtypes = [vp, C.POINTER(GlobalsC)]
    L.skred_bank_get_globals.argtypes = [vp, C.POINTER(GlobalsC)]
    L.skred_bank_render.argtypes = [vp, i32, i32, vp, vp, vp]
    L.skred_bank_master.argtypes = [vp, vp, i32, i32, vp, vp]
    L.skred_bank_render_mix.argtypes = [vp, i32, i32, vp, i32, vp, vp]
    L.skred_bank_render_host.argtypes = [vp, vp, i32, i32, i32, vp]
    L.skred_bank_last_render_ms.argtypes = [vp]
    L.skred_bank_last_render_ms.restype = C.c_float
    L.skred_bank_set_option.argtypes = [vp, i32, i32]
    L.skred_bank_last_kernel.argtypes = [vp]
    L.skred_bank_last_in_place.argtypes = [vp]
    L.skred_bank_last_split.argtypes = [vp]
    L.skred_bank_last_pack.argtypes = [vp]
    L.skred_bank_set_probe.argtypes = [vp, vp, i32, vp]
    L.skred_bank_list_violations.argtypes = [vp]
    L.skred_bank_list_violations.restype = C.c_uint
    L.skred_bank_timing_reset.argtypes = [vp]
    L.skred_bank_timing_reset.restype = None
    L.skred_bank_timing_summary.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(i32)]
    L.skred_bank_update.argtypes = [vp, C.POINTER(VoiceBankC), vp, i32, C.c_uint32, vp]
    L.skred_bank_defer.argtypes = [vp, C.c_uint64, C.POINTER(VoiceBankC), vp, i32, C.c_uint32]
    L.skred_bank_run_queue.argtypes = [vp, i32, vp]
    L.skred_bank_queue_pending.argtypes = [vp]
    L.skred_seq_create.argtypes = [C.POINTER(vp)]
    L.skred_seq_destroy.argtypes = [vp]
    L.skred_seq_destroy.restype = None
    L.skred_seq_tempo_set.argtypes = [vp, C.c_float]
    L.skred_seq_time_per_step.argtypes = [vp]
    L.skred_seq_time_per_step.restype = C.c_float
    for name in ("skred_seq_step_set", "skred_seq_mute_set"):
        getattr(L, name).argtypes = [vp, i32, i32, i32]
    for name in ("skred_seq_modulo_set", "skred_seq_state_set"):
        getattr(L, name).argtypes = [vp, i32, i32]
    for name in ("skred_seq_pattern_reset", "skred_seq_pointer", "skred_seq_counter"):
        getattr(L, name).argtypes = [vp, i32]
    L.skred_seq_tick.argtypes = [vp, i32, C.c_float, vp, i32]
    L.skred_bank_seq.argtypes = [vp]
    L.skred_bank_seq.restype = vp
    L.skred_bank_set_sample_rate.argtypes = [vp, C.c_float]
    L.skred_bank_pattern_step_set.argtypes = [vp, i32, i32, C.POINTER(VoiceBankC), vp, i32, C.c_uint32]
    L.skred_bank_pattern_step_clear.argtypes = [vp, i32, i32]
    _lib = L
    return L


def _check(rc: int, what: str):
    if rc != 0:
        msg = load().skred_amd_last_error().decode(errors="replace")
        raise SkredAmdError(f"{what} failed (rc={rc}): {msg}")


class DeviceBank:
    """A voice bank resident in one GPU's HBM."""

    def __init__(self, n_voices: int, device: int = 0):
        self.L = load()
        self.n = int(n_voices)
        self.device = device
        h = C.c_void_p()
        _check(self.L.skred_bank_create(device, self.n, C.byref(h)), "skred_bank_create")
        self.h = h

    @classmethod
    def borrowed(cls, handle, n_voices: int, device: int = 0) -> "DeviceBank":
        """A view of a bank somebody else owns (skred_shard_bank): close() does not destroy it."""
        self = cls.__new__(cls)
        self.L = load()
        self.n = int(n_voices)
        self.device = device
        self.h = C.c_void_p(handle)
        self._borrowed = True
        return self

    def close(self):
        if getattr(self, "h", None):
            if not getattr(self, "_borrowed", False):
                self.L.skred_bank_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_tables(self, pool: np.ndarray):
        pool = np.ascontiguousarray(pool, np.float32)
        _check(self.L.skred_bank_set_tables_f32(self.h, pool.ctypes.data, pool.size), "skred_bank_set_tables_f32")

    def upload(self, bank: VoiceBank, src_first: int = 0, dst_first: int = 0, count: Optional[int] = None):
        count = bank.n - src_first if count is None else count
        cb = bank.as_c()
        _check(self.L.skred_bank_upload(self.h, C.byref(cb), src_first, dst_first, count), "skred_bank_upload")

    def download(self, bank: VoiceBank, src_first: int = 0, dst_first: int = 0, count: Optional[int] = None):
        count = bank.n - dst_first if count is None else count
        cb = bank.as_c()
        _check(self.L.skred_bank_download(self.h, C.byref(cb), src_first, dst_first, count), "skred_bank_download")

    def set_globals(self, g: GlobalsC):
        _check(self.L.skred_bank_set_globals(self.h, C.byref(g)), "skred_bank_set_globals")

    def get_globals(self) -> GlobalsC:
        g = GlobalsC()
        _check(self.L.skred_bank_get_globals(self.h, C.byref(g)), "skred_bank_get_globals")
        return g

    def render(self, frames: int, d_partial: int, d_stems: int = 0, interp: int = 0, stream: int = 0):
        """Asynchronous render into device pointers (ints), e.g. torch tensor .data_ptr()."""
        _check(self.L.skred_bank_render(self.h, frames, interp, d_partial, d_stems or None, stream or None),
               "skred_bank_render")

    def render_mix(self, frames: int, d_out: int, channels: int = 2, d_stems: int = 0, interp: int = 0, stream: int = 0):
        """Single-GPU render + mix-down + master volume: one launch."""
        _check(self.L.skred_bank_render_mix(self.h, frames, interp, d_out, channels, d_stems, stream),
               "skred_bank_render_mix")

    def kernel_timing(self, every: int = 1):
        """SKRED_OPT_KERNEL_TIMING: event pair around the render kernels of every n-th launch (0: never)."""
        _check(self.L.skred_bank_set_option(self.h, 4, int(every)), "skred_bank_set_option")

    def master(self, d_sum: int, frames: int, d_out: int, channels: int = 2, stream: int = 0):
        _check(self.L.skred_bank_master(self.h, d_sum, frames, channels, d_out, stream or None), "skred_bank_master")

    def render_host(self, frames: int, channels: int = 2, interp: int = 0, want_stems: bool = False):
        """The synth() contract on host buffers.  Returns (buffer [F][ch], stems [F][N][2] | None)."""
        buf = np.zeros((frames, channels), np.float32)
        stems = np.zeros((frames, self.n, 2), np.float32) if want_stems else None
        _check(self.L.skred_bank_render_host(self.h, buf.ctypes.data, frames, channels, interp,
                                             stems.ctypes.data if want_stems else None), "skred_bank_render_host")
        return buf, stems

    # ---- block-granular updates (include/skred_amd.h: skred_bank_update / _defer / _run_queue) ----
    def update(self, bank: VoiceBank, voices, dirty: int, stream: int = 0):
        """Push the `dirty` parts (DIRTY_* | STAMP_*) of the listed voices from the host view."""
        v = np.ascontiguousarray(voices, np.int32)
        _check(self.L.skred_bank_update(self.h, C.byref(bank.as_c()), v.ctypes.data, len(v), dirty, stream),
               "skred_bank_update")

    def defer(self, when: int, bank: VoiceBank, voices, dirty: int):
        v = np.ascontiguousarray(voices, np.int32)
        _check(self.L.skred_bank_defer(self.h, when, C.byref(bank.as_c()), v.ctypes.data, len(v), dirty),
               "skred_bank_defer")

    def run_queue(self, frame_count: int, stream: int = 0) -> int:
        n = self.L.skred_bank_run_queue(self.h, frame_count, stream)
        if n < 0:
            _check(n, "skred_bank_run_queue")
        return n

    def queue_pending(self) -> int:
        return self.L.skred_bank_queue_pending(self.h)

    # ---- pattern steps on the bank's own step clock (include/skred_amd.h: skred_bank_pattern_* / skred_seq_*) ----
    def seq(self) -> "SeqClock":
        return SeqClock(self.L.skred_bank_seq(self.h), owner=False)

    def set_sample_rate(self, rate: float):
        _check(self.L.skred_bank_set_sample_rate(self.h, rate), "skred_bank_set_sample_rate")

    def pattern_step_set(self, pattern: int, step: int, bank: Optional[VoiceBank] = None, voices=(), dirty: int = 0):
        v = np.ascontiguousarray(voices, np.int32)
        cb = bank.as_c() if bank is not None else None
        _check(self.L.skred_bank_pattern_step_set(self.h, pattern, step, C.byref(cb) if cb is not None else None,
                                                  v.ctypes.data if len(v) else None, len(v), dirty), "skred_bank_pattern_step_set")

    def pattern_step_clear(self, pattern: int, step: int):
        _check(self.L.skred_bank_pattern_step_clear(self.h, pattern, step), "skred_bank_pattern_step_clear")

    def force_generic(self, on: bool = True):
        _check(self.L.skred_bank_set_option(self.h, 1, int(on)), "skred_bank_set_option")

    def fast2_min_voices(self, n: int):
        """Bank size from which the two-voices-per-lane kernel is chosen (0 = always when eligible)."""
        _check(self.L.skred_bank_set_option(self.h, 2, int(n)), "skred_bank_set_option")

    def fm2_min_voices(self, n: int):
        """Bank size from which a two-operator FM bank keeps carrier and modulator in one lane (SKRED_OPT_FM2_MIN_VOICES)."""
        _check(self.L.skred_bank_set_option(self.h, 5, int(n)), "skred_bank_set_option")

    def in_place(self, mode=1):
        """SKRED_OPT_IN_PLACE: 1 short motion lists rendered in the steady kernel's lanes where that is faster (default), 0 never
        (always the envelope kernel beside it), 2 whenever the gain rows provably suffice."""
        _check(self.L.skred_bank_set_option(self.h, 6, int(mode)), "skred_bank_set_option")

    def set_split(self, mode: int) -> None:
        """SKRED_OPT_SPLIT: 1 the oscillator-wave / post-wave form of the one-voice kernel where it is faster (default), 0 never,
        2 whenever the bank qualifies."""
        _check(self.L.skred_bank_set_option(self.h, 7, int(mode)), "skred_bank_set_option")

    def set_probe(self, voices, d_probe: int) -> None:
        """skred_bank_set_probe: (L, R) of every frame of the listed voices into d_probe[frame][i][2] (device memory), written from
        inside the kernels' fast paths; an empty list ends it."""
        v = np.ascontiguousarray(voices, np.int32)
        _check(self.L.skred_bank_set_probe(self.h, v.ctypes.data if len(v) else None, len(v), d_probe or None), "skred_bank_set_probe")

    def set_pack(self, mode: int) -> None:
        """SKRED_OPT_PACK: 1 sparse banks rendered with packed lanes where it pays (default), 0 never, 2 whenever a wavefront disappears."""
        _check(self.L.skred_bank_set_option(self.h, 9, int(mode)), "skred_bank_set_option")

    def set_fm_skew(self, on: int) -> None:
        """SKRED_OPT_FM_SKEW: 1 (default) modulator lanes of a frequency-modulated wavefront run a block ahead of their carriers, 0 per-frame exchange."""
        _check(self.L.skred_bank_set_option(self.h, 10, int(on)), "skred_bank_set_option")

    def last_pack(self) -> int:
        """Lanes per 64-voice group in the latest block, 0: not packed."""
        return int(self.L.skred_bank_last_pack(self.h))

    def set_split_pairs(self, pairs: int) -> None:
        """SKRED_OPT_SPLIT_PAIRS (tests): 0 the library's choice, 2 / 4 pairs per workgroup forced."""
        _check(self.L.skred_bank_set_option(self.h, 8, int(pairs)), "skred_bank_set_option")

    def last_split(self) -> bool:
        return bool(self.L.skred_bank_last_split(self.h))

    def last_in_place(self) -> bool:
        return bool(self.L.skred_bank_last_in_place(self.h))

    def last_kernel(self) -> int:
        """0 = generic kernel, 1 = specialised fast kernel (SKRED_KERNEL_*)."""
        return int(self.L.skred_bank_last_kernel(self.h))

    def list_violations(self) -> int:
        """Voices the steady two-per-lane kernel found in motion without being on the motion list (cross-check; 0 by construction)."""
        return int(self.L.skred_bank_list_violations(self.h))

    def last_render_ms(self) -> float:
        return float(self.L.skred_bank_last_render_ms(self.h))

    def timing_reset(self):
        self.L.skred_bank_timing_reset(self.h)

    def timing_summary(self):
        """(mean_ms, min_ms, count) of the render kernel over the calls since timing_reset()."""
        mean, mn, cnt = C.c_float(), C.c_float(), C.c_int()
        _check(self.L.skred_bank_timing_summary(self.h, C.byref(mean), C.byref(mn), C.byref(cnt)),
               "skred_bank_timing_summary")
        return float(mean.value), float(mn.value), int(cnt.value)


class SeqClock:
    """skred_seq_t: the reference's pattern step clock (seq.c:179-213) on the host."""

    def __init__(self, handle=None, owner: bool = True):
        self.L = load()
        self.owner = owner
        if handle is None:
            h = C.c_void_p()
            _check(self.L.skred_seq_create(C.byref(h)), "skred_seq_create")
            handle = h
        self.h = C.c_void_p(handle) if isinstance(handle, int) else handle

    def close(self):
        if self.owner and self.h:
            self.L.skred_seq_destroy(self.h)
        self.h = None

    def tempo(self, bpm: float):
        _check(self.L.skred_seq_tempo_set(self.h, bpm), "skred_seq_tempo_set")

    def time_per_step(self) -> float:
        return float(self.L.skred_seq_time_per_step(self.h))

    def step(self, pattern: int, step: int, occupied: bool = True):
        _check(self.L.skred_seq_step_set(self.h, pattern, step, int(occupied)), "skred_seq_step_set")

    def mute(self, pattern: int, step: int, on: bool = True):
        _check(self.L.skred_seq_mute_set(self.h, pattern, step, int(on)), "skred_seq_mute_set")

    def modulo(self, pattern: int, m: int):
        _check(self.L.skred_seq_modulo_set(self.h, pattern, m), "skred_seq_modulo_set")

    def state(self, pattern: int, state: int):
        _check(self.L.skred_seq_state_set(self.h, pattern, state), "skred_seq_state_set")

    def reset(self, pattern: int):
        _check(self.L.skred_seq_pattern_reset(self.h, pattern), "skred_seq_pattern_reset")

    def pointer(self, pattern: int) -> int:
        return int(self.L.skred_seq_pointer(self.h, pattern))

    def counter(self, pattern: int) -> int:
        return int(self.L.skred_seq_counter(self.h, pattern))

    def tick(self, frame_count: int, sample_rate: float = 44100.0):
        """[(pattern, step)] of the steps that fire in this block."""
        out = (C.c_int32 * 16)()
        n = self.L.skred_seq_tick(self.h, frame_count, sample_rate, out, 16)
        if n < 0:
            _check(n, "skred_seq_tick")
        return [(out[i] >> 16, out[i] & 0xFFFF) for i in range(n)]
