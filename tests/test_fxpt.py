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


def test_fx_definition_biquad_and_one_shots():
    """CPU definition: the Q2.30 biquad tracks the float biquad of the same coefficients; a one-shot plays one cycle,
    sounds on its last frame from the table's last entry and is skipped afterwards with its state frozen."""
    from skred_amd import banks
    n = 64
    b, pool, c0 = fxbank.bank_fx(n)
    b["use_envelope"][:] = 0
    b["smoother_enable"][:] = 0
    b["amp_q15"][:] = 32768
    b["pan_left_q15"][:] = 32768
    b["disconnect"][:] = 0
    mix, stems, _ = cpuref.fx_render(b.copy(), pool, c0, 2000, 1, want_stems=True)
    nb = b.copy()
    nb["filter_mode"][:] = 0
    _, dry, _ = cpuref.fx_render(nb, pool, c0, 2000, 1, want_stems=True)
    co = {k: b[k + "_q30"].astype(np.float64) / (1 << 30) for k in ("b0", "b1", "b2", "a1", "a2")}
    x = dry[:, :, 0].astype(np.float64)
    y = np.zeros_like(x)
    for i in range(2000):
        y[i] = co["b0"] * x[i] + (co["b1"] * x[i - 1] if i > 0 else 0) + (co["b2"] * x[i - 2] if i > 1 else 0) \
            - (co["a1"] * y[i - 1] if i > 0 else 0) - (co["a2"] * y[i - 2] if i > 1 else 0)
    unclipped = np.abs(y).max(0) < 32000
    err = np.abs(stems[:, :, 0] - y)[:, unclipped].max()
    assert unclipped.sum() > n // 2 and err < 4.0, (int(unclipped.sum()), float(err))   # a few LSBs of a 16-bit sample
    # one-shots
    o = b.copy()
    o["filter_mode"][:] = 0
    o["one_shot"][::2] = 1
    o["phase"][:] = 0
    o["phase_inc"][:] = np.uint32((1 << 32) // 100 + 12345)          # ~100 frames per cycle
    ob = o.copy()
    _, st, _ = cpuref.fx_render(ob, pool, c0, 300, 1, want_stems=True)
    assert (ob["finished"][::2] == 1).all() and (ob["finished"][1::2] == 0).all()
    assert (ob["phase"][::2] == 0xFFFFFFFF).all() and (ob["voice_sample"][::2] == 0).all()
    last = int(np.ceil((1 << 32) / float(o["phase_inc"][0]))) - 1   # 0-based frame whose add carries
    assert (st[last + 1:, ::2, :] == 0).all() and (np.abs(st[:last + 1, ::2, :]).sum(0) > 0).all()
    lut_last = np.array([pool[o["table_offset"][v] + (1 << o["log2_size"][v]) - 1] for v in range(0, n, 2)], np.int64)
    assert (st[last, ::2, 0] == lut_last).all()                    # amp 1.0, pan_left 1.0, no interpolation across the end


@pytest.mark.gpu
@pytest.mark.parametrize("n,interp", [(4096, 0), (4096, 1), (65536, 1), (1000, 1)])
def test_fx_gpu_bit_exact(n, interp):
    b, pool, c0 = fxbank.bank_fx(n)
    b["disconnect"][::7] = 1
    b["amp_q15"][::11] = 0
    b["smoother_enable"][::13] = 0
    b["filter_mode"][::5] = 0                              # some voices unfiltered: per-lane select in the blocks
    one = np.arange(3, n, 17)                              # stopping one-shots, periods of 50 .. 1500 frames
    b["one_shot"][one] = 1
    b["phase_inc"][one] = ((1 << 32) // (50 + (one * 37) % 1450)).astype(np.uint32)
    b["phase"][one] = (one.astype(np.uint64) * 2654435761 % (1 << 32)).astype(np.uint32)
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
def test_fx_gpu_bit_exact_at_bench_size():
    """2^20 voices -- the size bench.py's `fixed_point` leg times: integer mix and read-write state, no stems."""
    n = 1 << 20
    b, pool, c0 = fxbank.bank_fx(n)
    b["amp_q15"][::11] = 0
    one = np.arange(5, n, 29)
    b["one_shot"][one] = 1
    b["phase_inc"][one] = ((1 << 32) // (40 + (one * 13) % 900)).astype(np.uint32)
    segs = [(300, None), (212, _release_odd)]
    rb, count = b.copy(), c0
    db = fxbank.DeviceFxBank(n)
    db.set_tables(pool)
    host = b.copy()
    db.upload(host)
    db.set_sample_count(c0)
    for frames, ev in segs:
        if ev:
            ev(rb, count)
            db.download(host)
            ev(host, db.sample_count())
            db.upload(host)
        ref, _, count = cpuref.fx_render(rb, pool, count, frames, 1, fast=True)
        got, _ = db.render_host(frames, 1)
        assert (got == ref).all()
    db.download(host)
    db.close()
    assert not host.rw_mismatch(rb), host.rw_mismatch(rb)
    assert int(rb["finished"].sum()) > 10000


@pytest.mark.gpu
def test_fx_filter_state_saturates_like_the_definition():
    """Resonant filters driven hard: the Q12 delay line hits +-2^29 and the sample clamp to int16 acts -- the kernel must
    saturate exactly where the definition does (block and frame-by-frame paths)."""
    n = 2048
    b, pool, c0 = fxbank.bank_fx(n)
    co = fxbank.q30_coeffs(np.full(n, 1), 40.0 + 3.0 * np.arange(n, dtype=np.float32), np.full(n, 60.0, np.float32), 48000)
    for k, v in co.items():
        b[k] = v
    b["filter_mode"][:] = 1
    b["x1"][:] = (1 << 29) - 1                               # start at the rail
    b["y1"][::2] = -(1 << 29)
    b["use_envelope"][n // 2:] = 0                           # half the bank on the steady blocks from the first frame
    rb, count = b.copy(), c0
    db = fxbank.DeviceFxBank(n)
    db.set_tables(pool)
    db.upload(b)
    db.set_sample_count(c0)
    for frames in (700, 64, 9):
        ref, rst, count = cpuref.fx_render(rb, pool, count, frames, 1, want_stems=True)
        got, gst = db.render_host(frames, 1, want_stems=True)
        assert (got == ref).all() and (gst == rst).all()
    host = b.copy()
    db.download(host)
    db.close()
    assert not host.rw_mismatch(rb), host.rw_mismatch(rb)
    assert (np.abs(rb["y1"]) >= (1 << 29) - 1).any() or (np.abs(rst).max() > 20000)
    bad = fxbank.FxVoiceBank(4)
    bad["y2"][1] = 1 << 30                                   # outside what the definition can produce: refused
    d2 = fxbank.DeviceFxBank(4)
    d2.set_tables(np.zeros(16, np.int16))
    with pytest.raises(device.SkredAmdError):
        d2.upload(bad)
    d2.close()


@pytest.mark.gpu
@pytest.mark.parametrize("interp", [0, 1])
def test_fx_pool_beyond_lds_is_gathered_from_memory(interp):
    """A pool of more than 48 KB does not fit the workgroup's LDS: the kernel gathers the tables from L2 / HBM (its blocks with
    the definition's arithmetic spelled out, not the lean LDS blocks).  Same voices, the tables moved behind 30 000 entries of
    padding, some voices on tables that straddle the old LDS limit."""
    n = 3000
    b, pool, c0 = fxbank.bank_fx(n)
    pad = 30000
    rng = np.random.default_rng(5)
    big = np.concatenate([rng.integers(-32768, 32767, pad).astype(np.int16), pool])
    assert big.nbytes > 49152
    b["table_offset"] = (b["table_offset"] + pad).astype(np.int32)
    far = np.arange(7, n, 9)                                 # random-noise tables inside the padding, 2^10 entries each
    b["table_offset"][far] = (far * 13) % (pad - 1024)
    b["log2_size"][far] = 10
    b["amp_q15"][::11] = 0
    b["filter_mode"][::5] = 0
    segs = [(700, None), (333, _release_odd), (64, None)]
    rb, count = b.copy(), c0
    db = fxbank.DeviceFxBank(n)
    db.set_tables(big)
    host = b.copy()
    db.upload(host)
    db.set_sample_count(c0)
    for frames, ev in segs:
        if ev:
            ev(rb, count)
            db.download(host)
            ev(host, db.sample_count())
            db.upload(host)
        ref, rst, count = cpuref.fx_render(rb, big, count, frames, interp, want_stems=True)
        got, gst = db.render_host(frames, interp, want_stems=True)
        assert (got == ref).all() and (gst == rst).all()
    db.download(host)
    db.close()
    assert not host.rw_mismatch(rb), host.rw_mismatch(rb)


@pytest.mark.gpu
def test_fx_one_lane_on_the_rail_among_steady_lanes():
    """The lean steady blocks CHECK the delay line's clamp instead of applying it and hand a block in which some lane leaves
    +-2^29 back to the spelled-out block; the lanes around it and the blocks after it must not notice.  One resonant voice per
    wave is driven to the rail for a while (its coefficients make it ring, then the ringing decays), every other voice is the
    ordinary recipe; no stems at 2^16 voices, so the timed blocks themselves run."""
    n = 1 << 16
    b, pool, c0 = fxbank.bank_fx(n)
    hot = np.arange(5, n, 64)
    co = fxbank.q30_coeffs(np.full(hot.size, 1), np.full(hot.size, 300.0, np.float32), np.full(hot.size, 80.0, np.float32), 48000)
    for k, v in co.items():
        b[k][hot] = v
    b["filter_mode"][hot] = 1
    b["y1"][hot] = (1 << 29) - 1
    b["y2"][hot] = -(1 << 29)
    b["use_envelope"][:] = 0                                  # steady from the first frame
    rb, count = b.copy(), c0
    db = fxbank.DeviceFxBank(n)
    db.set_tables(pool)
    db.upload(b)
    db.set_sample_count(c0)
    for frames in (512, 512, 100):
        ref, _, count = cpuref.fx_render(rb, pool, count, frames, 1, fast=True)
        got, _ = db.render_host(frames, 1)
        assert (got == ref).all()
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


def test_fx_master_definition_properties():
    """CPU definition of the integer master stage: the gain climbs monotonically from 0 to its target and rests there exactly
    (Q31 state: the smoother stalls within 2^-22 of the target); the output is the mix scaled by the Q15 gain, floor-shifted."""
    mix = np.tile(np.array([[1 << 30, -(1 << 30)]], np.int64), (20000, 1))
    out, g = cpuref.fx_master(fxbank.MASTER_TARGET_Q31, fxbank.MASTER_K_Q15, 0, mix)
    gains = out[:, 0] * 32768 // (1 << 30)
    assert (np.diff(gains) >= 0).all() and gains[0] < gains[-1]
    assert abs(g - fxbank.MASTER_TARGET_Q31) < 512 and gains[-1] == (g >> 16)
    assert (out[:, 1] == ((-(1 << 30)) * (out[:, 0] * 32768 // (1 << 30))) >> 15).all()


@pytest.mark.gpu
@pytest.mark.parametrize("n,frames", [(3000, 700), (70000, 512), (1 << 20, 512)])
def test_fx_render_mix_is_one_launch_with_the_master_stage(n, frames):
    """skred_fxbank_render_mix: render + in-kernel mix-down + integer master stage in one launch, over several blocks (the
    gain carries), == the definition bit for bit; and skred_fxbank_render + skred_fxbank_master (the multi-GPU split) gives
    the same bytes.  Flat (<= 64 rows) and two-level mix-downs, up to the bench's 2^20 voices."""
    import torch
    b, pool, c0 = fxbank.bank_fx(n)
    ref, cnt, g = b.copy(), c0, 0
    want = []
    blocks = 3 if n <= 70000 else 2
    for _ in range(blocks):
        mix, _, cnt = cpuref.fx_render(ref, pool, cnt, frames, 1, fast=(n > 70000))
        out, g = cpuref.fx_master(fxbank.MASTER_TARGET_Q31, fxbank.MASTER_K_Q15, g, mix)
        want.append(out)
    for split in (False, True):
        db = fxbank.DeviceFxBank(n)
        db.set_tables(pool)
        db.upload(b)
        db.set_sample_count(c0)
        d_out = torch.zeros(frames, 2, dtype=torch.int64, device="cuda")
        d_sum = torch.zeros(frames, 2, dtype=torch.int64, device="cuda")
        for k in range(blocks):
            if split:
                db.render(frames, d_sum.data_ptr(), 1)
                db.master(d_sum.data_ptr(), frames, d_out.data_ptr())
            else:
                db.render_mix(frames, d_out.data_ptr(), 1)
            torch.cuda.synchronize()
            assert (d_out.cpu().numpy() == want[k]).all(), (split, k)
        assert db.master_gain() == g
        got = b.copy()
        db.download(got)
        db.close()
        assert not got.rw_mismatch(ref), got.rw_mismatch(ref)


@pytest.mark.gpu
def test_fx_stamps_on_resident_voices():
    """skred_fxbank_stamp: note-offs and note-ons on device-resident voices, stamped with the bank's clock when they run,
    against the definition given the same stores at the same blocks (release only takes if the envelope is active)."""
    n, frames = 6000, 256
    b, pool, c0 = fxbank.bank_fx(n)
    b["is_active"][::9] = 0                                # some voices are not sounding: a release must not stamp them
    ref, cnt = b.copy(), c0
    db = fxbank.DeviceFxBank(n)
    db.set_tables(pool)
    db.upload(b)
    db.set_sample_count(c0)
    rng = np.random.default_rng(3)
    for k in range(12):
        off = rng.choice(n, 40, replace=False).astype(np.int32)
        on = rng.choice(n, 40, replace=False).astype(np.int32)
        db.stamp(off, fxbank.FX_STAMP_RELEASE)
        db.stamp(on, fxbank.FX_STAMP_TRIGGER)
        act = ref["is_active"][off] != 0
        ref["sample_release"][off[act]] = cnt
        ref["sample_start"][on] = cnt
        ref["sample_release"][on] = 0
        ref["is_active"][on] = 1
        mix, _ = db.render_host(frames, 1)
        want, _, cnt = cpuref.fx_render(ref, pool, cnt, frames, 1)
        assert (mix == want).all(), k
    got = b.copy()
    db.download(got)
    db.close()
    assert not got.rw_mismatch(ref), got.rw_mismatch(ref)


@pytest.mark.gpu
def test_fx_shard_with_one_rank_rccl_equals_the_unsharded_render():
    """skred_fxshard_*: the fixed-point bank through the N > 1 sequence (sum-only render -> ncclReduce(sum, int64) -> integer
    master stage on the root) with the library's own one-rank RCCL communicator: the bytes of skred_fxbank_render_mix."""
    import ctypes as C
    import torch
    from skred_amd.sharded import Shard, _lib
    n, frames = 20000, 300
    b, pool, c0 = fxbank.bank_fx(n)
    ref, cnt, g = b.copy(), c0, 0
    L = fxbank._bind(_lib())
    h = C.c_void_p()
    assert L.skred_fxshard_create(0, 0, 1, 0, n, C.byref(h)) == 0
    fx = fxbank.DeviceFxBank.borrowed(L.skred_fxshard_bank(h), n)
    fx.set_tables(pool)
    cb = b.as_c()
    assert L.skred_fxshard_upload(h, C.byref(cb)) == 0
    fx.set_sample_count(c0)
    idb = C.create_string_buffer(Shard.rccl_unique_id(), 128)
    assert L.skred_shard_init_rccl(h, idb) == 0
    assert L.skred_shard_set_ops(h, None, 1) == 0
    d_out = torch.zeros(frames, 2, dtype=torch.int64, device="cuda")
    for k in range(3):
        assert L.skred_shard_render_mix(h, frames, 1, None, d_out.data_ptr(), 2, None) == 0, L.skred_amd_last_error()
        torch.cuda.synchronize()
        mix, _, cnt = cpuref.fx_render(ref, pool, cnt, frames, 1)
        want, g = cpuref.fx_master(fxbank.MASTER_TARGET_Q31, fxbank.MASTER_K_Q15, g, mix)
        assert (d_out.cpu().numpy() == want).all(), k
    L.skred_shard_destroy(h)
