"""GPU parity of the SHIPPED PATCHES' modulation routings at bank scale.

`skred_amd.banks.bank_patch(patch, n)` tiles the voice state the unmodified reference holds after loading one of its own
patches (skred_amd/data/patches/patch_<patch>.npz, written by tests/golden/gen_patch_voices.py from the reference's wire()) over
n voices: a modulator shared by three carriers (3.sk), frequency + pan modulation from two voices with sample & hold (37.sk),
chains (7.sk), amplitude modulation (1.sk), a modulator BELOW its carrier -- same-frame dependency (18.sk).  These are the
banks tools/measure_banks.py `patches` times; here they are rendered against the oracle (oracle/cpu_ref.c, pinned to the
compiled reference), semantics synth.c:548-558,584-587,597-602:

  * 2^16 voices with per-voice stems: every voice-sample bit for bit, on the kernel the library picks AND on the
    full-featured kernels (SKRED_OPT_FORCE_GENERIC), plus all read-write state;
  * 2^20 voices, 512-frame blocks (the timed workload): all read-write state bit for bit after several blocks, mix within 1e-5.
"""
import numpy as np
import pytest

import golden_io as gio
from oracle import cpuref
from skred_amd import banks

pytestmark = pytest.mark.gpu

PATCHES = ["3sk", "37sk", "7sk", "1sk", "18sk"]


def rms(x):
    return float(np.sqrt(np.mean(np.asarray(x, np.float64) ** 2)))


def rel_rms(a, b):
    return rms(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / max(rms(b), 1e-30)


@pytest.fixture(scope="module")
def dev():
    from skred_amd import device
    L = device.load()
    assert L.skred_amd_device_count() > 0, "no GPU visible"
    return device


@pytest.mark.parametrize("patch", PATCHES)
def test_patch_bank_stems_at_65536_voices(dev, patch):
    n = 1 << 16
    bank, tables, g = banks.bank_patch(patch, n)
    ref_bank, ref_g = bank.copy(), g.copy()
    frames = [96, 33]                                  # two calls: state carried over, a ragged block
    refs = []
    for f in frames:
        r = cpuref.render(ref_bank, ref_g, tables, f, 0, want_stems=True)
        refs.append((r["stems"], cpuref.master(ref_g, r["sum64"].astype(np.float32))))
    kernels = {}
    for generic in (False, True):
        db = dev.DeviceBank(n)
        db.set_tables(tables)
        db.upload(bank)
        db.set_globals(g)
        db.force_generic(generic)
        for f, (ref_stems, ref_mix) in zip(frames, refs):
            mix, stems = db.render_host(f, 2, 0, want_stems=True)
            assert gio.bits_equal(stems, ref_stems), f"{patch}: stems differ from the oracle (generic={generic}, {f} frames)"
            assert rel_rms(mix, ref_mix) <= 1e-5
        kernels[generic] = db.last_kernel()
        got = bank.copy()
        db.download(got)
        db.close()
        bad = got.rw_equal(ref_bank)
        assert not bad, (patch, generic, bad)
    assert kernels[True] in (0, 2), kernels            # the full-featured kernels: generic / modulated


@pytest.mark.parametrize("patch", PATCHES)
def test_patch_bank_full_size_blocks(dev, patch):
    """The workload `measure_banks patches` times: 2^20 voices, 512-frame blocks, no stems (the block paths)."""
    import torch
    n = 1 << 20
    bank, tables, g = banks.bank_patch(patch, n)
    ref_bank, ref_g = bank.copy(), g.copy()
    db = dev.DeviceBank(n)
    db.set_tables(tables)
    db.upload(bank)
    db.set_globals(g)
    out = torch.zeros(512, 2, device="cuda")
    for k in range(3):
        db.render_mix(512, out.data_ptr(), 2, 0, 0)
        torch.cuda.synchronize()
        r = cpuref.render(ref_bank, ref_g, tables, 512, 0)
        ref_mix = cpuref.master(ref_g, r["sum64"].astype(np.float32))
        assert rel_rms(out.cpu().numpy(), ref_mix) <= 1e-5, f"{patch}: block {k}"
    got = bank.copy()
    db.download(got)
    db.close()
    bad = got.rw_equal(ref_bank)
    assert not bad, (patch, bad)
