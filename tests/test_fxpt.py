"""Fixed-point render path (include/skred_amd_fxpt.h).  Definition = oracle/cpu_ref_fxpt.c (the
reference has no fixed-point path: unpinned upstream).  Bar: GPU == definition
sample-for-sample, per voice AND in the integer mix."""
import ctypes

import numpy as np
import pytest

from oracle import cpuref
from skred_amd import device, fxbank


def test_fx_abi_symbols_exported():
    L = device.load()
    for s in fxbank.FX_ABI_SYMBOLS:
        assert hasattr(L, s), s
    assert ctypes.sizeof(fxbank.FxBankC) == 8 + 8 * len(fxbank.FX_FIELDS)


def test_fx_definition_basic_properties():
    """CPU definition sanity: silence for amp 0, envelope reaches sustain, mix = sum of stems."""
    b, pool, c0 = fxbank.bank_fx(300)
    b["amp_q15"][::5] = 0
    mix, stems, c1 = cpuref.fx_render(b, pool, c0, 600, 1, want_stems=True)
    assert c1 == c0 + 600
    assert (stems[:, ::5, :] == 0).all()
    assert (mix == stems.astype(np.int64).sum(1)).all()
    assert (b["voice_sample"][::5] == 0).all()


def _release_odd(bank, now):
    bank["sample_release"][1::2] = now


@pytest.mark.gpu
@pytest.mark.parametrize("n,interp", [(4096, 0), (4096, 1), (65536, 1), (1000, 1)])
def test_fx_gpu_bit_exact(n, interp):
    b, pool, c0 = fxbank.bank_fx(n)
    b["disconnect"][::7] = 1
    b["amp_q15"][::11] = 0
    b["smoother_enable"][::13] = 0
    want_stems = n <= 4096
    segs = [(301, None), (333, _release_odd), (10001, None), (64, None)]
    # definition
    rb, rc_count = b.copy(), c0
    ref_mix, ref_stems = [], []
    for frames, ev in segs:
        if ev:
            ev(rb, rc_count)
        m, s, rc_count = cpuref.fx_render(rb, pool, rc_count, frames, interp, want_stems=want_stems)
        ref_mix.append(m)
        ref_stems.append(s)
    # GPU
    db = fxbank.DeviceFxBank(n)
    db.set_tables(pool)
    host = b.copy()
    db.upload(host)
    db.set_sample_count(c0)
    got_mix, got_stems = [], []
    for frames, ev in segs:
        if ev:
            db.download(host)
            ev(host, db.sample_count())
            db.upload(host)
        m, s = db.render_host(frames, interp, want_stems=want_stems)
        got_mix.append(m)
        got_stems.append(s)
    db.download(host)
    assert db.sample_count() == rc_count
    db.close()
    assert not host.rw_mismatch(rb), host.rw_mismatch(rb)
    for k in range(len(segs)):
        assert (got_mix[k] == ref_mix[k]).all(), f"integer mix differs in segment {k}"
        if want_stems:
            assert (got_stems[k] == ref_stems[k]).all(), f"stems differ in segment {k}"


@pytest.mark.gpu
@pytest.mark.parametrize("interp", [0, 1])
def test_fx_wide_parameters_bit_exact(interp):
    """Parameters outside the Q15 habit: velocity 60000 (amp * e and s * gain then need the definition's 64-bit
    products, the gain exceeds 16 bits), a smoother state of 2^20, a pan gain beyond 24 bits.  Such waves take the
    block code with full-width multiplies instead of the 24-bit ones; held notes, so the steady blocks run."""
    n = 1000
    b, pool, c0 = fxbank.bank_fx(n)
    x = slice(256, 320)
    b["velocity_q15"][x] = 60000
    b["amp_q15"][x] = 65535
    b["pan_left_q15"][x] = 8192
    b["pan_right_q15"][x] = 4096
    b["smoother_gain_q15"][x] = 1 << 20
    y = slice(512, 576)
    b["amp_q15"][y] = 100
    b["pan_left_q15"][y] = 9_000_000
    b["smoother_k_q15"][700:764] = 40000          # k above 1.0: not the narrow case either
    segs = [(6000, None), (512, None), (100, None), (7, None)]
    rb, count = b.copy(), c0
    ref = []
    for frames, _ in segs:
        m, st, count = cpuref.fx_render(rb, pool, count, frames, interp, want_stems=True)
        ref.append((m, st))
    db = fxbank.DeviceFxBank(n)
    db.set_tables(pool)
    db.upload(b)
    db.set_sample_count(c0)
    for k, (frames, _) in enumerate(segs):
        m, st = db.render_host(frames, interp, want_stems=True)
        assert (m == ref[k][0]).all(), f"integer mix differs in segment {k}"
        assert (st == ref[k][1]).all(), f"stems differ in segment {k}"
    host = b.copy()
    db.download(host)
    db.close()
    assert not host.rw_mismatch(rb), host.rw_mismatch(rb)


@pytest.mark.gpu
def test_fx_mix_is_exactly_additive():
    """Integer path: mix(A u B) == mix(A) + mix(B) EXACTLY -> a multi-GPU sum of partial mixes is bit-exact."""
    n, frames = 8192, 256
    b, pool, c0 = fxbank.bank_fx(n)

    def render(sub):
        db = fxbank.DeviceFxBank(sub.n)
        db.set_tables(pool)
        db.upload(sub)
        db.set_sample_count(c0)
        m, _ = db.render_host(frames, 1)
        db.close()
        return m

    whole = render(b)
    assert (render(b.take(slice(0, 3000))) + render(b.take(slice(3000, n))) == whole).all()


@pytest.mark.gpu
def test_fx_bad_arguments():
    db = fxbank.DeviceFxBank(4)
    with pytest.raises(device.SkredAmdError):
        db.render_host(8)                                  # no tables
    db.set_tables(np.zeros(16, np.int16))
    b = fxbank.FxVoiceBank(4)
    b["log2_size"] = 5                                     # 32 entries > 16-entry pool
    with pytest.raises(device.SkredAmdError):
        db.upload(b)
    b["log2_size"] = 4
    b["amp_q15"] = 70000
    with pytest.raises(device.SkredAmdError):
        db.upload(b)
    db.close()
