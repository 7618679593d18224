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
    assert sorted(set(re.findall(r"\b(skred_fxbank_\w+)\s*\(", text))) == sorted(fxbank.FX_ABI_SYMBOLS)


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
