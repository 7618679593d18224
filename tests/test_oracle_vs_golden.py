"""Pins oracle/cpu_ref.c (our CPU restatement) to the compiled reference.

Every fixture under tests/golden/ was rendered by the unmodified reference synth.c
(tests/golden/gen_golden.py).  cpu_ref must reproduce, BIT FOR BIT: the output frames
(post master volume), every per-voice stem, the read-write voice state and the globals.
"""
import numpy as np
import pytest

import golden_io as gio
from oracle import cpuref


@pytest.mark.parametrize("case", gio.CASES)
def test_cpuref_bit_exact(case):
    g = gio.load(case)
    for seg in g.segments:
        bank = seg.bank_in.copy()
        gl = seg.g_in.copy()
        # replay with the same callback block structure the reference used
        mix = np.zeros((seg.frames, 2), np.float32)
        stems = np.zeros((seg.frames, bank.n, 2), np.float32)
        p = 0
        while p < seg.frames:
            n = min(seg.block, seg.frames - p)
            buf, st = cpuref.synth(bank, gl, g.tables, n, 2, 0, want_stems=True)
            mix[p:p + n], stems[p:p + n] = buf, st
            p += n
        assert gio.bits_equal(mix, seg.mix), f"{case} seg{seg.index}: mix differs"
        assert gio.sha256(stems) == seg.stems_sha256, f"{case} seg{seg.index}: stems differ"
        if seg.stems is not None:
            assert gio.bits_equal(stems[:, seg.stems_voices, :], seg.stems)
        bad = bank.rw_equal(gio.expected_out_bank(seg))
        assert not bad, f"{case} seg{seg.index}: state differs {bad}"
        assert gl.synth_sample_count == seg.g_out.synth_sample_count
        assert gl.noise_rng == seg.g_out.noise_rng
        assert np.float32(gl.volume_smoother_gain).tobytes() == np.float32(seg.g_out.volume_smoother_gain).tobytes()


@pytest.mark.parametrize("case", gio.CASES)
def test_cpuref_block_size_independent(case):
    """The reference renders the same bytes whatever the callback size; so must the oracle."""
    g = gio.load(case)
    seg = g.segments[0]
    frames = min(seg.frames, 2048)
    bank, gl = seg.bank_in.copy(), seg.g_in.copy()
    whole, _ = cpuref.synth(bank, gl, g.tables, frames)
    assert gio.bits_equal(whole, seg.mix[:frames])


def test_c0_anchor_hash():
    """SURVEY §8c / BASELINE.md anchor for 0.sk: FNV-1a-32 of the raw stereo f32 bytes."""
    g = gio.load("c0_0sk")
    assert g.meta["segments"][0]["mix_fnv1a32"] == "4160cd81"
    assert abs(g.meta["segments"][0]["mix_rms"] - 0.0350477384) < 1e-9


def test_f64_sum_close_to_f32_sum():
    g = gio.load("c2_mixed_filter64")
    seg = g.segments[0]
    r = cpuref.render(seg.bank_in.copy(), seg.g_in.copy(), g.tables, 1024)
    err = np.sqrt(np.mean((r["sum32"].astype(np.float64) - r["sum64"]) ** 2))
    assert err < 1e-5


def test_oracle_at_256_voices_against_four_reference_runs():
    """N > 64 pinned to the reference: the oracle renders the 256-voice bank in one go; per voice it must
    reproduce the reference's stems of the four 64-voice runs bit for bit, hence (f64) their sum."""
    import hashlib
    g = gio.load("bank256_sum")
    bank, gl = gio.bank256_from_parts(g)
    r = cpuref.render(bank, gl, g.tables, g.segments[0].frames, 0, want_stems=True)
    sha = np.frombuffer(hashlib.sha256(np.ascontiguousarray(r["stems"]).tobytes()).digest(), np.uint8)
    assert (sha == g.extras["stems256_sha256"]).all()
    want = g.extras["sum64"]
    assert np.abs(r["sum64"] - want).max() <= 1e-12 * max(1.0, np.abs(want).max())
