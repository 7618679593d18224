"""Voices sharded over the GPUs of one node: ctypes image of skred_amd/csrc/skred_shard.c (include/skred_amd.h:
skred_shard_*).  One process per GPU; rank r renders the contiguous block [lo, hi) of the bank into a PRE-master
partial mix float[F][2]; the only exchange step is ONE reduce(sum) of 8*F bytes per block onto the root (RCCL over
xGMI, owned by the C library), after which the root applies the master-volume stage (synth.c:616-624) once.

No logic lives here: the partition rule, the check that no modulation crosses a cut and the per-block sequence
render -> reduce -> master are the C library's.  `Shard` is the GPU form bench.py drives; `ShardedRender` plugs Python
callables into the same C sequencing so that the world_size-2/3 gloo tests rehearse it on CPUs with the oracle as the
renderer (tests/test_sharded_gloo.py).
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Optional, Tuple

import numpy as np

from . import device
from .bank import VoiceBank

RENDER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p)
MASTER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p)
REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)


class ShardOps(C.Structure):
    """== skred_shard_ops_t"""
    _fields_ = [("ctx", C.c_void_p), ("render", RENDER_FN), ("master", MASTER_FN),
                ("reduce_ctx", C.c_void_p), ("reduce", REDUCE_FN)]


def _lib():
    L = device.load()
    if not getattr(L, "_shard_ready", False):
        vp, i32 = C.c_void_p, C.c_int
        L.skred_shard_partition.argtypes = [i32, i32, i32, C.POINTER(i32), C.POINTER(i32)]
        L.skred_shard_cut_ok.argtypes = [vp, i32, i32]
        L.skred_shard_create.argtypes = [i32, i32, i32, i32, i32, C.POINTER(vp)]
        L.skred_shard_create_custom.argtypes = [i32, i32, i32, i32, C.POINTER(ShardOps), C.POINTER(vp)]
        L.skred_shard_destroy.argtypes = [vp]
        L.skred_shard_destroy.restype = None
        L.skred_shard_bank.argtypes = [vp]
        L.skred_shard_bank.restype = vp
        L.skred_shard_range.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
        L.skred_shard_upload.argtypes = [vp, vp]
        L.skred_shard_set_ops.argtypes = [vp, C.POINTER(ShardOps), i32]
        L.skred_shard_rccl_unique_id.argtypes = [vp]
        L.skred_shard_init_rccl.argtypes = [vp, vp]
        L.skred_shard_render_mix.argtypes = [vp, i32, i32, vp, vp, i32, vp]
        L.skred_shard_render_mix_pipelined.argtypes = [vp, i32, i32, vp, i32, vp]
        L.skred_shard_flush.argtypes = [vp, vp]
        L._shard_ready = True
    return L


def _check(rc: int, what: str):
    if rc != 0:
        raise device.SkredAmdError(f"{what} failed (rc={rc}): {_lib().skred_amd_last_error().decode(errors='replace')}")


def partition(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block of voices for `rank`: [lo, hi) (skred_shard_partition)."""
    lo, hi = C.c_int(), C.c_int()
    _check(_lib().skred_shard_partition(total, world, rank, C.byref(lo), C.byref(hi)), "skred_shard_partition")
    return lo.value, hi.value


def modulation_components_ok(bank: VoiceBank, lo: int, hi: int) -> bool:
    """A cut is only legal where no voice inside [lo,hi) is modulated by one outside (skred_shard_cut_ok)."""
    cb = bank.as_c()
    return bool(_lib().skred_shard_cut_ok(C.byref(cb), lo, hi))


class Shard:
    """The GPU form: this rank's block of the bank on `device_index`, the library's own RCCL reduce between the ranks."""

    def __init__(self, total_voices: int, rank: int = 0, world: int = 1, device_index: int = 0, root: int = 0):
        self.L = _lib()
        self.total, self.rank, self.world, self.root = total_voices, rank, world, root
        h = C.c_void_p()
        _check(self.L.skred_shard_create(device_index, rank, world, root, total_voices, C.byref(h)), "skred_shard_create")
        self.h = h
        lo, hi = C.c_int(), C.c_int()
        self.L.skred_shard_range(self.h, C.byref(lo), C.byref(hi))
        self.lo, self.hi = lo.value, hi.value
        self.bank = device.DeviceBank.borrowed(self.L.skred_shard_bank(self.h), self.hi - self.lo, device_index)
        self._keep = None

    @property
    def n_local(self) -> int:
        return self.hi - self.lo

    def upload(self, whole: VoiceBank):
        cb = whole.as_c()
        _check(self.L.skred_shard_upload(self.h, C.byref(cb)), "skred_shard_upload")

    @staticmethod
    def rccl_unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        _check(_lib().skred_shard_rccl_unique_id(buf), "skred_shard_rccl_unique_id")
        return buf.raw

    def init_rccl(self, unique_id: bytes):
        assert len(unique_id) == 128
        _check(self.L.skred_shard_init_rccl(self.h, C.create_string_buffer(unique_id, 128)), "skred_shard_init_rccl")

    def set_reduce(self, reduce: Optional[Callable[[int, int, int], None]], always_reduce: bool = False):
        """Replace the collective: reduce(ptr, n_floats, root) sums the ranks' buffers in place on the root (rehearsals)."""
        ops = ShardOps()
        if reduce is not None:
            def _red(_ctx, ptr, n, root, _stream):
                reduce(ptr, n, root)
                return 0
            ops.reduce = REDUCE_FN(_red)
        self._keep = ops
        _check(self.L.skred_shard_set_ops(self.h, C.byref(ops), int(always_reduce)), "skred_shard_set_ops")

    def render_mix(self, frames: int, d_out: int, channels: int = 2, interp: int = 0, stream: int = 0, d_partial: int = 0):
        _check(self.L.skred_shard_render_mix(self.h, frames, interp, d_partial or None, d_out or None, channels, stream or None),
               "skred_shard_render_mix")

    def render_mix_pipelined(self, frames: int, d_out: int, channels: int = 2, interp: int = 0, stream: int = 0):
        """The collective of this block runs beside the render of the next one.  Host-paced: `d_out` of call k is complete once
        call k + 2 has returned (alternate between two output buffers), or on `stream` after flush(stream)."""
        _check(self.L.skred_shard_render_mix_pipelined(self.h, frames, interp, d_out or None, channels, stream or None),
               "skred_shard_render_mix_pipelined")

    def flush(self, stream: int = 0):
        _check(self.L.skred_shard_flush(self.h, stream or None), "skred_shard_flush")

    def step_pipelined(self, out: np.ndarray, interp: int = 0) -> None:
        """The same block through skred_shard_render_mix_pipelined (the library's own rotating partial buffers)."""
        assert out.dtype == np.float32 and out.flags.c_contiguous
        _check(self.L.skred_shard_render_mix_pipelined(self.h, out.shape[0], interp, out.ctypes.data, out.shape[1], None),
               "skred_shard_render_mix_pipelined")

    def close(self):
        if getattr(self, "h", None):
            self.L.skred_shard_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedRender:
    """The C sequencing with caller-supplied steps (host memory): render(frames, interp, partial[F,2]),
    reduce(partial[F,2], root), master(sum[F,2], out[F,ch]).  Used by the CPU rehearsal with the oracle."""

    def __init__(self, total_voices: int, rank: int, world: int, render, master, reduce=None, root: int = 0):
        self.L = _lib()
        self.total, self.rank, self.world, self.root = total_voices, rank, world, root
        self.lo, self.hi = partition(total_voices, world, rank)

        def arr(ptr, rows, cols):
            return np.ctypeslib.as_array((C.c_float * (rows * cols)).from_address(ptr)).reshape(rows, cols)

        def _render(_ctx, frames, interp, partial, _stream):
            render(frames, interp, arr(partial, frames, 2))
            return 0

        def _master(_ctx, total, frames, channels, out, _stream):
            master(arr(total, frames, 2), arr(out, frames, channels))
            return 0

        def _reduce(_ctx, partial, n, root_, _stream):
            reduce(arr(partial, n // 2, 2), root_)
            return 0

        self._ops = ShardOps()
        self._ops.render = RENDER_FN(_render)
        self._ops.master = MASTER_FN(_master)
        if reduce is not None:
            self._ops.reduce = REDUCE_FN(_reduce)
        h = C.c_void_p()
        _check(self.L.skred_shard_create_custom(rank, world, root, total_voices, C.byref(self._ops), C.byref(h)),
               "skred_shard_create_custom")
        self.h = h

    @property
    def n_local(self) -> int:
        return self.hi - self.lo

    def step(self, partial: np.ndarray, out: np.ndarray, interp: int = 0) -> None:
        """One block: render -> reduce -> master on the root, sequenced by skred_shard_render_mix."""
        assert partial.dtype == np.float32 and out.dtype == np.float32 and partial.flags.c_contiguous and out.flags.c_contiguous
        _check(self.L.skred_shard_render_mix(self.h, partial.shape[0], interp, partial.ctypes.data, out.ctypes.data,
                                             out.shape[1], None), "skred_shard_render_mix")

    def step_pipelined(self, out: np.ndarray, interp: int = 0) -> None:
        """The same block through skred_shard_render_mix_pipelined (the library's own rotating partial buffers)."""
        assert out.dtype == np.float32 and out.flags.c_contiguous
        _check(self.L.skred_shard_render_mix_pipelined(self.h, out.shape[0], interp, out.ctypes.data, out.shape[1], None),
               "skred_shard_render_mix_pipelined")

    def close(self):
        if getattr(self, "h", None):
            self.L.skred_shard_destroy(self.h)
            self.h = None
