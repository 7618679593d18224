"""CPU-side checks of the C ABI: the library loads and exports every symbol include/skred_amd.h
declares (no compute without a GPU), and fails loudly -- never silently falls back -- when no
device is usable."""
import ctypes as C
import os
import re

import pytest

from skred_amd import device

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "skred_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(skred_(?:amd|bank|shard|seq)_\w+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(device.ABI_SYMBOLS)


def test_fxpt_header_and_binding_agree():
    from skred_amd import fxbank
    text = open(os.path.join(ROOT, "include", "skred_amd_fxpt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    assert sorted(set(re.findall(r"\b(skred_fx(?:bank|shard)_\w+)\s*\(", text))) == sorted(fxbank.FX_ABI_SYMBOLS)
    L = device.load()
    for sym in fxbank.FX_ABI_SYMBOLS:
        assert hasattr(L, sym), f"libskred_amd.so does not export {sym}"


def test_library_exports_every_declared_symbol():
    L = device.load()
    for s in declared_symbols():
        assert hasattr(L, s), f"libskred_amd.so does not export {s}"
    assert L.skred_amd_abi_version() == 1


def test_no_gpu_means_loud_failure():
    L = device.load()
    if L.skred_amd_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(device.SkredAmdError) as e:
        device.DeviceBank(64)
    assert "no HIP device" in str(e.value) or "hip" in str(e.value).lower()


def test_struct_sizes_match_header():
    from skred_amd.bank import ENV_DTYPE, MMF_DTYPE, GlobalsC, VoiceBankC, FIELDS
    assert MMF_DTYPE.itemsize == 48 and ENV_DTYPE.itemsize == 56      # synth-types.h:13-38
    assert C.sizeof(GlobalsC) == 32
    assert C.sizeof(VoiceBankC) == 8 + 8 * len(FIELDS)


def test_bank_size_ceiling_is_enforced_before_the_device():
    """SKRED_MAX_VOICES (include/skred_amd.h): voice and list indices inside the kernels are 32-bit; a bank above the
    documented ceiling is refused with SKRED_E_RANGE, GPU or not (the GPU suite renders a bank of exactly the ceiling)."""
    text = open(os.path.join(ROOT, "include", "skred_amd.h")).read()
    m = re.search(r"#define\s+SKRED_MAX_VOICES\s+\(1\s*<<\s*(\d+)\)", text)
    assert m, "SKRED_MAX_VOICES missing from the header"
    ceiling = 1 << int(m.group(1))
    L = device.load()
    h = C.c_void_p()
    for n in (ceiling + 1, 2**31 - 1):
        assert L.skred_bank_create(0, n, C.byref(h)) == -4            # SKRED_E_RANGE
        assert b"SKRED_MAX_VOICES" in L.skred_amd_last_error()
        assert not h.value
    assert L.skred_bank_create(0, 0, C.byref(h)) == -2                # SKRED_E_BAD_ARG


def test_shard_reduce_step_must_exist_when_it_is_required():
    """A one-rank shard made without a reduce step and then told to always reduce must fail, not call a NULL pointer."""
    L = device.load()
    vp, i32 = C.c_void_p, C.c_int

    class Ops(C.Structure):
        _fields_ = [("ctx", vp), ("render", vp), ("master", vp), ("reduce_ctx", vp), ("reduce", vp)]

    RENDER = C.CFUNCTYPE(i32, vp, i32, i32, vp, vp)
    MASTER = C.CFUNCTYPE(i32, vp, vp, i32, i32, vp, vp)
    render = RENDER(lambda ctx, f, interp, partial, stream: 0)
    master = MASTER(lambda ctx, s, f, ch, out, stream: 0)
    ops = Ops(None, C.cast(render, vp), C.cast(master, vp), None, None)
    L.skred_shard_create_custom.argtypes = [i32, i32, i32, i32, C.POINTER(Ops), C.POINTER(vp)]
    L.skred_shard_set_ops.argtypes = [vp, C.POINTER(Ops), i32]
    L.skred_shard_render_mix.argtypes = [vp, i32, i32, vp, vp, i32, vp]
    L.skred_shard_destroy.argtypes = [vp]
    L.skred_shard_destroy.restype = None
    sh = vp()
    assert L.skred_shard_create_custom(0, 1, 0, 64, C.byref(ops), C.byref(sh)) == 0
    assert L.skred_shard_set_ops(sh, None, 1) == 0
    buf = (C.c_float * 128)()
    assert L.skred_shard_render_mix(sh, 64, 0, buf, buf, 2, None) == -2   # SKRED_E_BAD_ARG
    assert L.skred_shard_set_ops(sh, None, 0) == 0
    assert L.skred_shard_render_mix(sh, 64, 0, buf, buf, 2, None) == 0
    L.skred_shard_destroy(sh)
