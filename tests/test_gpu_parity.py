"""GPU parity tests (run with -m gpu on an MI355X).  Everything goes through the C ABI
(libskred_amd.so via skred_amd.device); the oracle (oracle/cpu_ref.c, itself pinned bit-exact to
the compiled reference by test_oracle_vs_golden.py) and the golden fixtures are the checkers.

Bars:
  * per-voice stems and read-write voice state: BIT-EXACT (truncating lookup);
  * float mix: RMS error <= 1e-5 against the reference / the f64-accumulated truth
    (the wavefront tree sum orders additions differently from the reference's voice loop).
"""
import os
import numpy as np
import pytest

import golden_io as gio
from oracle import cpuref
from skred_amd import banks
from skred_amd.bank import VoiceBank

pytestmark = pytest.mark.gpu

MIX_RMS_TOL = 1e-5


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


@pytest.mark.parametrize("case", gio.CASES)
def test_golden_case(dev, case):
    """The reference's own output, segment by segment, callback by callback."""
    g = gio.load(case)
    for seg in g.segments:
        db = dev.DeviceBank(seg.bank_in.n)
        db.set_tables(g.tables)
        db.upload(seg.bank_in)
        db.set_globals(seg.g_in)
        mix = np.zeros((seg.frames, 2), np.float32)
        stems = np.zeros((seg.frames, seg.bank_in.n, 2), np.float32)
        p = 0
        while p < seg.frames:
            n = min(seg.block, seg.frames - p)
            buf, st = db.render_host(n, 2, 0, want_stems=True)
            mix[p:p + n], stems[p:p + n] = buf, st
            p += n
        got = seg.bank_in.copy()
        db.download(got)
        gl = db.get_globals()
        db.close()
        # bit-exact: stems, state, globals
        assert gio.sha256(stems) == seg.stems_sha256, f"{case} seg{seg.index}: per-voice stems differ from the reference"
        if seg.stems is not None:
            assert gio.bits_equal(stems[:, seg.stems_voices, :], seg.stems)
        bad = got.rw_equal(gio.expected_out_bank(seg))
        assert not bad, f"{case} seg{seg.index}: voice state differs {bad}"
        assert gl.synth_sample_count == seg.g_out.synth_sample_count
        assert gl.noise_rng == seg.g_out.noise_rng
        assert np.float32(gl.volume_smoother_gain).tobytes() == np.float32(seg.g_out.volume_smoother_gain).tobytes()
        # tolerance: the mix
        err = rms(mix.astype(np.float64) - seg.mix.astype(np.float64))
        assert err <= MIX_RMS_TOL, f"{case} seg{seg.index}: mix rms error {err}"
        if rms(seg.mix) > 1e-12:
            assert rel_rms(mix, seg.mix) <= 1e-5


@pytest.mark.parametrize("recipe,n,frames,interp", [
    ("c1", 4096, 1024, 0),        # BASELINE config 1, full size
    ("c2", 65536, 512, 0),        # BASELINE config 2, full size
    ("c2", 1000, 700, 0),         # ragged: not a multiple of the 256-voice group
    ("c4", 20000, 512, 0),        # PCM pool too large for LDS -> L2/HBM gather path
    ("c4", 20000, 512, 1),        # linear interpolation (defined by cpu_ref, unpinned upstream)
    ("c2", 4096, 512, 1),
    ("c4", 262144, 512, 1),       # BASELINE config 4 at the size bench.py times it (linear: unpinned upstream)
    ("c4", 262144, 512, 0),       # ... and with the reference's truncating lookup
])
def test_synthetic_bank_vs_oracle(dev, recipe, n, frames, interp):
    """Seeded BASELINE banks at (or near) full size against the oracle on identical inputs."""
    bank, tables, g = banks.RECIPES[recipe](n)
    ref_bank, ref_g = bank.copy(), g.copy()
    r = cpuref.render(ref_bank, ref_g, tables, frames, interp, want_stems=(n <= 4096))
    ref_mix = cpuref.master(ref_g, r["sum64"].astype(np.float32))  # gain sequence is exact; sum is f64 truth
    db = dev.DeviceBank(n)
    db.set_tables(tables)
    db.upload(bank)
    db.set_globals(g)
    mix, stems = db.render_host(frames, 2, interp, want_stems=(n <= 4096))
    got = bank.copy()
    db.download(got)
    db.close()
    assert not got.rw_equal(ref_bank), got.rw_equal(ref_bank)
    if stems is not None:
        assert gio.bits_equal(stems, r["stems"])
    assert rel_rms(mix, ref_mix) <= 1e-5


def test_block_size_independence(dev):
    """Rendering F frames in one launch or in ragged pieces leaves identical state and stems."""
    bank, tables, g = banks.bank_c2(2048)
    outs = []
    for blocks in ([600], [64, 1, 199, 336], [100] * 6):
        db = dev.DeviceBank(bank.n)
        db.set_tables(tables)
        db.upload(bank)
        db.set_globals(g)
        st = [db.render_host(n, 2, 0, want_stems=True) for n in blocks]
        got = bank.copy()
        db.download(got)
        db.close()
        outs.append((np.concatenate([s[1] for s in st]), np.concatenate([s[0] for s in st]), got))
    for stems, mix, got in outs[1:]:
        assert gio.bits_equal(stems, outs[0][0])
        assert gio.bits_equal(mix, outs[0][1])          # same reduction tree -> identical mix too
        assert not got.rw_equal(outs[0][2])


def test_mix_is_linear_in_voices(dev):
    """mix(A u B) == mix(A) + mix(B) within tolerance (voices are independent): the
    property the multi-GPU sharding rests on, checked at BASELINE config-2 size."""
    n, frames = 65536, 256
    bank, tables, g = banks.bank_c2(n)

    def render(sub: VoiceBank):
        db = dev.DeviceBank(sub.n)
        db.set_tables(tables)
        db.upload(sub)
        gg = g.copy()
        gg.volume_smoother_gain = gg.volume_final      # constant master gain -> the stage is linear
        db.set_globals(gg)
        m, _ = db.render_host(frames)
        db.close()
        return m.astype(np.float64)

    whole = render(bank)
    parts = render(bank.take(slice(0, n // 2))) + render(bank.take(slice(n // 2, n)))
    assert rel_rms(parts, whole) <= 1e-5


def test_256_voices_against_four_reference_runs(dev):
    """N > 64 pinned to the reference itself: the 256-voice bank in one launch against the reference's four 64-voice
    runs -- per-voice stems bit for bit (generic kernel), and the pre-master sum of the specialised kernel against
    the f64 sum of the reference's stems."""
    import hashlib
    import torch
    g = gio.load("bank256_sum")
    bank, gl = gio.bank256_from_parts(g)
    frames = g.segments[0].frames
    db = dev.DeviceBank(bank.n)
    db.set_tables(g.tables)
    db.upload(bank)
    db.set_globals(gl)
    _, stems = db.render_host(frames, 2, 0, want_stems=True)
    sha = np.frombuffer(hashlib.sha256(np.ascontiguousarray(stems).tobytes()).digest(), np.uint8)
    assert (sha == g.extras["stems256_sha256"]).all()
    for fast2 in (False, True):
        db.upload(bank)
        db.set_globals(gl)
        db.fast2_min_voices(0 if fast2 else 1 << 30)
        part = torch.zeros(frames, 2, device="cuda")
        db.render(frames, part.data_ptr())
        torch.cuda.synchronize()
        assert db.last_kernel() == (3 if fast2 else 1)
        assert rel_rms(part.cpu().numpy(), g.extras["sum64"]) <= 1e-5
    db.close()


def test_fused_mix_equals_render_plus_master(dev):
    """skred_bank_render_mix (mix-down and master stage inside the render kernel) against skred_bank_render +
    skred_bank_master on the same bank: same samples, bit for bit, for small and large workgroup counts."""
    import torch
    for n, frames in ((300, 77), (40000, 512), (300000, 1500)):
        bank, tables, g = banks.bank_c2(n)
        outs = []
        for fused in (False, True):
            db = dev.DeviceBank(n)
            db.set_tables(tables)
            db.upload(bank)
            db.set_globals(g)
            out = torch.zeros(frames, 2, device="cuda")
            part = torch.zeros(frames, 2, device="cuda")
            for _ in range(2):
                if fused:
                    db.render_mix(frames, out.data_ptr(), 2)
                else:
                    db.render(frames, part.data_ptr())
                    db.master(part.data_ptr(), frames, out.data_ptr(), 2)
            torch.cuda.synchronize()
            outs.append(out.cpu().numpy())
            db.close()
        assert gio.bits_equal(outs[0], outs[1]), (n, frames)


def test_async_blocks_into_separate_buffers(dev):
    """Six consecutive asynchronous blocks, each into its own buffer, nothing synchronised in between, against the same
    six blocks rendered one by one through the synchronous host form: same bytes, same state (a block's mix-down happens
    inside its launch; the next launch re-uses the rows, the tickets and the gain slot)."""
    import torch
    n, frames, blocks = 70000, 300, 6
    bank, tables, g = banks.bank_c2(n)
    res = []
    for asynchronous in (False, True):
        db = dev.DeviceBank(n)
        db.set_tables(tables)
        db.upload(bank)
        db.set_globals(g)
        if asynchronous:
            outs = [torch.zeros(frames, 2, device="cuda") for _ in range(blocks)]
            for o in outs:
                db.render_mix(frames, o.data_ptr(), 2)
            torch.cuda.synchronize()
            res.append(np.concatenate([o.cpu().numpy() for o in outs]))
        else:
            res.append(np.concatenate([db.render_host(frames)[0] for _ in range(blocks)]))
        got = bank.copy()
        db.download(got)
        res.append(got)
        db.close()
    assert gio.bits_equal(res[0], res[2])
    assert not res[1].rw_equal(res[3])


def test_determinism(dev):
    """No atomics, fixed reduction order: two runs give identical bytes."""
    bank, tables, g = banks.bank_c2(10000)
    res = []
    for _ in range(2):
        db = dev.DeviceBank(bank.n)
        db.set_tables(tables)
        db.upload(bank)
        db.set_globals(g)
        res.append(db.render_host(512)[0])
        db.close()
    assert gio.bits_equal(res[0], res[1])


def test_modulator_outside_its_group_fails_loudly(dev):
    """Modulator and carrier must share an aligned 64-voice group (one wavefront exchanges
    voice_sample through LDS); anything else is refused, never rendered approximately."""
    bank, tables, g = banks.bank_c2(256)
    bank["voice_freq_mod_osc"][70] = 3          # carrier in group 1, modulator in group 0
    bank["voice_freq_mod_depth"][70] = 2.0
    db = dev.DeviceBank(bank.n)
    db.set_tables(tables)
    db.upload(bank)
    db.set_globals(g)
    with pytest.raises(dev.SkredAmdError):
        db.render_host(64)
    db.close()


def test_modulated_bank_of_many_groups_vs_oracle(dev):
    """The 64-voice modulated patch of the golden case, replicated into a 640-voice bank (10 groups)."""
    gold = gio.load("edge_mod")
    seg = gold.segments[0]
    reps = 10
    big = VoiceBank(64 * reps)
    for k in range(reps):
        sl = slice(64 * k, 64 * (k + 1))
        for name in big.a:
            big.a[name][sl] = seg.bank_in.a[name]
        for key in ("voice_freq_mod_osc", "voice_amp_mod_osc", "voice_pan_mod_osc", "voice_cz_mod_osc"):
            m = big.a[key][sl]
            big.a[key][sl] = np.where(m >= 0, m + 64 * k, m)
        big.a["voice_phase"][sl] += np.float32(k)          # de-correlate the copies
    ref_bank, ref_g = big.copy(), seg.g_in.copy()
    r = cpuref.render(ref_bank, ref_g, gold.tables, 700, 0, want_stems=True)
    ref_mix = cpuref.master(ref_g, r["sum64"].astype(np.float32))
    db = dev.DeviceBank(big.n)
    db.set_tables(gold.tables)
    db.upload(big)
    db.set_globals(seg.g_in)
    mix, stems = db.render_host(700, 2, 0, want_stems=True)
    assert db.last_kernel() == 2
    got = big.copy()
    db.download(got)
    db.close()
    assert gio.bits_equal(stems, r["stems"])
    assert not got.rw_equal(ref_bank), got.rw_equal(ref_bank)
    assert rel_rms(mix, ref_mix) <= 1e-5


def test_empty_and_bad_arguments(dev):
    db = dev.DeviceBank(3)
    with pytest.raises(dev.SkredAmdError):
        db.render_host(16)                       # no tables yet
    db.set_tables(np.zeros(8, np.float32))
    mix, _ = db.render_host(16)                  # all voices silent (amp 0): zeros
    assert not mix.any()
    b = VoiceBank(3)
    b["voice_amp"] = 1.0
    b["voice_table_size"] = 64                   # outside the 8-float pool
    with pytest.raises(dev.SkredAmdError):
        db.upload(b)
    db.close()


# ---------------------------------------------------------------- the specialised ("fast") kernel

def _run_scenario(dev, bank, tables, g, interp, segments, force_generic, fast2=False, fm2=None):
    """Render `segments` = [(frames, event)] where event(bank_host, now) mutates the host bank
    between launches (host arrays are the source of truth: download -> edit -> upload)."""
    db = dev.DeviceBank(bank.n)
    db.set_tables(tables)
    host = bank.copy()
    db.upload(host)
    db.set_globals(g)
    db.force_generic(force_generic)
    if fast2:
        db.fast2_min_voices(0)
    if fm2 is not None:
        db.fm2_min_voices(fm2)
    mixes, kernels = [], []
    for frames, event in segments:
        if event is not None:
            db.download(host)
            event(host, db.get_globals().synth_sample_count)
            db.upload(host)
        m, _ = db.render_host(frames, 2, interp)
        mixes.append(m)
        kernels.append(db.last_kernel())
    db.download(host)
    db.close()
    return np.concatenate(mixes), host, kernels


def _oracle_scenario(bank, tables, g, interp, segments):
    host, gl = bank.copy(), g.copy()
    mixes = []
    for frames, event in segments:
        if event is not None:
            event(host, gl.synth_sample_count)
        r = cpuref.render(host, gl, tables, frames, interp)
        mixes.append(cpuref.master(gl, r["sum64"].astype(np.float32)))
    return np.concatenate(mixes), host


def _release_odd_voices(host, now):
    e = host["voice_amp_envelope"]
    e["sample_release"][1::2] = now          # == amp_envelope_release, synth.c:391-395


@pytest.mark.parametrize("recipe,interp", [("c1", 0), ("c2", 0), ("c2", 1), ("c4", 0), ("c4", 1)])
def test_fast_kernel_bit_identical_to_generic_and_oracle(dev, recipe, interp):
    """Attack/decay in flight -> note-off on half the voices -> release runs out (is_active -> 0) ->
    smoother tail; ragged voice count, frame counts that are not multiples of the 64-frame chunk."""
    n = 5000
    bank, tables, g = banks.RECIPES[recipe](n)
    bank["voice_disconnect"][::7] = 1            # muted voices stay on the fast path
    bank["voice_amp"][::11] = 0.0                # skipped voices too (state frozen, sample = 0)
    segs = [(301, None), (333, _release_odd_voices), (12001, None), (64, None)]
    fast_mix, fast_state, k_fast = _run_scenario(dev, bank, tables, g, interp, segs, force_generic=False)
    gen_mix, gen_state, k_gen = _run_scenario(dev, bank, tables, g, interp, segs, force_generic=True)
    ref_mix, ref_state = _oracle_scenario(bank, tables, g, interp, segs)
    assert k_fast == [1] * len(segs), "the specialised kernel did not run"
    assert k_gen == [0] * len(segs)
    assert not fast_state.rw_equal(ref_state), fast_state.rw_equal(ref_state)
    assert not gen_state.rw_equal(ref_state), gen_state.rw_equal(ref_state)
    assert gio.bits_equal(fast_mix, gen_mix), "same per-voice samples + same reduction tree => same bytes"
    assert rel_rms(fast_mix, ref_mix) <= 1e-5


@pytest.mark.parametrize("recipe,interp", [("c1", 0), ("c2", 0), ("c2", 1)])
def test_sustained_bank_on_the_tile_blocks(dev, recipe, interp):
    """Every voice in its sustain stage (the recipe's last note-on is 5280 frames old after the first segment): the
    8-frame LDS-tile blocks of the one-voice kernel, with block counts that leave tail frames (777 = 12 chunks + 9
    frames; 100 = 1 chunk + 36), muted and skipped voices among them; then the smoother stalls (its own block variant).
    Per-voice state bit for bit against the oracle, mixes within 1e-5."""
    n = 3000
    bank, tables, g = banks.RECIPES[recipe](n)
    bank["voice_disconnect"][:1000:7] = 1        # waves with a muted live voice keep the frame pairs; the others take the blocks
    bank["voice_amp"][::11] = 0.0
    segs = [(5700, None), (777, None), (100, None), (8, None), (3, None), (4096, None), (515, None)]
    mix, state, k = _run_scenario(dev, bank, tables, g, interp, segs, force_generic=False)
    ref_mix, ref_state = _oracle_scenario(bank, tables, g, interp, segs)
    assert k == [1] * len(segs)
    assert not state.rw_equal(ref_state), state.rw_equal(ref_state)
    assert rel_rms(mix, ref_mix) <= 1e-5


def test_fast_kernel_envelope_clock_beyond_2p24(dev):
    """Notes held for more than 2^24 frames (5.8 min at 48 kHz): (float)(uint64) stops being an
    exact +1 ramp; the kernel must fall back to the integer clock.  Crosses 2^24 inside a launch."""
    n = 1024
    bank, tables, g = banks.bank_c2(n)
    e = bank["voice_amp_envelope"]
    e["sample_start"] = np.arange(n, dtype=np.uint64) * 3
    g.synth_sample_count = (1 << 24) - 200
    segs = [(512, None), (300, _release_odd_voices), (700, None)]
    fast_mix, fast_state, k = _run_scenario(dev, bank, tables, g, 0, segs, force_generic=False)
    ref_mix, ref_state = _oracle_scenario(bank, tables, g, 0, segs)
    assert k == [1, 1, 1]
    assert not fast_state.rw_equal(ref_state), fast_state.rw_equal(ref_state)
    assert rel_rms(fast_mix, ref_mix) <= 1e-5


def test_fast_kernel_big_increments(dev):
    """Phase increments beyond one loop length (general fmodf wrap, synth.c:247) on the fast path."""
    n = 2048
    bank, tables, g = banks.bank_c2(n)
    size = bank["voice_table_size"].astype(np.float32)
    bank["voice_phase_inc"][::3] = size[::3] * np.float32(2.37)
    bank["voice_phase_inc"][1::3] = size[1::3] * np.float32(0.999)
    fast_mix, fast_state, k = _run_scenario(dev, bank, tables, g, 0, [(700, None)], force_generic=False)
    ref_mix, ref_state = _oracle_scenario(bank, tables, g, 0, [(700, None)])
    assert k == [1]
    assert not fast_state.rw_equal(ref_state), fast_state.rw_equal(ref_state)
    assert rel_rms(fast_mix, ref_mix) <= 1e-5


@pytest.mark.parametrize("recipe,interp", [("c4", 0), ("c4", 1), ("c2", 0)])
def test_stopping_one_shots_on_the_specialised_kernel(dev, recipe, interp):
    """Forward one-shots without loop (a sampler's drums): each plays to its table end, is audible on the frame it
    finishes (clamped index, second tap clamped, not folded), then freezes with voice_sample = 0.  The one-per-lane
    kernel keeps such banks: finishing is checked frame by frame, the voice's planes are stored at that moment (delay
    line in reference order, whichever half of a frame pair it happens in) and the lane goes inert.  Voices finish in
    the middle of launches, on the last frame of a launch, before the first launch; note-offs and a re-trigger of
    finished voices come in between."""
    n = 3000
    bank, tables, g = banks.RECIPES[recipe](n)
    stop = np.arange(n) % 3 != 0
    bank["voice_one_shot"][stop] = 1
    bank["voice_loop_enabled"][stop] = 0
    size = bank["voice_table_size"].astype(np.float32)
    # spread the remaining play time: some finish within tens of frames, some never in this test
    left = (np.float32(5.0) + np.float32(40.0) * (np.arange(n) % 211)).astype(np.float32)
    inc = np.maximum(bank["voice_phase_inc"], np.float32(0.25))
    bank["voice_phase_inc"][stop] = inc[stop]
    bank["voice_phase"][stop] = np.maximum(size[stop] - left[stop] * inc[stop], np.float32(0.0))
    sel = np.where(stop)[0]
    bank["voice_phase"][sel[0]] = size[sel[0]] - np.float32(200.0) * bank["voice_phase_inc"][sel[0]] + np.float32(0.5)  # frame 200 = last frame of launch 1
    bank["voice_finished"][sel[1]] = 1                                       # finished before the first launch

    def retrigger(host, now):                                                # osc_trigger on every 5th stopping voice
        vs = sel[::5]
        host["voice_finished"][vs] = 0
        host["voice_phase"][vs] = 0.0

    segs = [(200, None), (333, _release_odd_voices), (512, None), (64, retrigger), (1500, None)]
    mix, state, k = _run_scenario(dev, bank, tables, g, interp, segs, force_generic=False)
    gen_mix, gen_state, kg = _run_scenario(dev, bank, tables, g, interp, segs, force_generic=True)
    ref_mix, ref_state = _oracle_scenario(bank, tables, g, interp, segs)
    assert k == [1] * len(segs) and kg == [0] * len(segs), (k, kg)
    assert int(ref_state["voice_finished"].sum()) > 500                      # the scenario does finish voices
    assert not gen_state.rw_equal(ref_state), gen_state.rw_equal(ref_state)
    assert not state.rw_equal(ref_state), state.rw_equal(ref_state)
    assert rel_rms(mix, ref_mix) <= 1e-5


@pytest.mark.parametrize("recipe,interp,mixed", [("c1", 0, False), ("c2", 0, False), ("c2", 1, False), ("c2", 0, True)])
def test_two_operator_fm_pairs_share_a_lane(dev, recipe, interp, mixed):
    """A two-operator FM bank (`v0 ... F1,depth` / `v1 ... m1` repeated: every carrier an even voice modulated by the voice
    after it) on the two-voices-per-lane kernel with carrier and modulator in ONE lane (SKM_FM_PAIR): against the oracle and
    against the one-per-lane kernel's exchange.  Deep modulation (negative and longer-than-a-loop increments), unmodulated
    pairs in between, muted and unmuted modulators, carriers and modulators switched off between launches, note-offs, and
    -- c2 -- envelopes in motion from the first frame (the pair is handed to the envelope kernel as a pair)."""
    n = 6144 + 250
    bank, tables, g = banks.RECIPES[recipe](n)
    car = np.arange(0, n - 1, 2)
    car = car[(car // 2) % 5 != 4]                      # every fifth pair stays unmodulated
    mod = car + 1
    bank["voice_freq_mod_osc"][car] = mod
    bank["voice_freq_mod_depth"][car] = (np.float32(0.05) * (1 + (car % 97))).astype(np.float32)     # up to ~5: wild
    bank["voice_freq_scale"][car] = (np.float32(0.5) + np.float32(0.01) * (car % 50)).astype(np.float32)
    bank["voice_disconnect"][mod[::2]] = 1                                                          # `m1` on half of them
    if mixed:                                           # filter / envelope on some voices only: per-lane flags
        bank["voice_filter_mode"][mod[::3]] = 0
        bank["voice_use_amp_envelope"][car[::4]] = 0
    # amplitude and pan modulation of the same shape (`A1`, `P1`: by the voice after the carrier; `A0`, `P0`: by the carrier
    # itself), on carriers with and without FM, one of them muted (a muted voice's pan is not modulated, synth.c:596)
    am = car[3::11]; bank["voice_amp_mod_osc"][am] = am + 1; bank["voice_amp_mod_depth"][am] = np.float32(3.0)
    pm = car[7::13]; bank["voice_pan_mod_osc"][pm] = pm + 1; bank["voice_pan_mod_depth"][pm] = np.float32(8.0)
    sa = car[9::29]; bank["voice_amp_mod_osc"][sa] = sa; bank["voice_amp_mod_depth"][sa] = np.float32(2.5)
    sp = car[11::31]; bank["voice_pan_mod_osc"][sp] = sp; bank["voice_pan_mod_depth"][sp] = np.float32(6.0)
    free = np.arange(8, n - 1, 10)                      # the unmodulated pairs' even voices: pan modulation only
    bank["voice_pan_mod_osc"][free[::2]] = free[::2] + 1; bank["voice_pan_mod_depth"][free[::2]] = np.float32(4.0)
    bank["voice_disconnect"][pm[::4]] = 1

    def kill_some(host, now):
        host["voice_amp"][mod[::7]] = 0.0
        host["voice_amp"][car[3::11]] = 0.0

    def retrigger_all_and_kill_more(host, now):
        # every envelope in motion again: whole passes of the steady kernel have nothing left to render, and the modulators
        # switched off HERE still owe their last sample to their carriers' first frame (voice_sample is an input of the pair)
        e = host["voice_amp_envelope"]
        e["sample_start"][:] = now; e["sample_release"][:] = 0; e["is_active"][:] = 1
        host["voice_amp"][mod[2::7]] = 0.0

    segs = [(300, None), (257, _release_odd_voices), (512, kill_some), (700, None), (192, retrigger_all_and_kill_more), (64, None)]
    mix, state, k = _run_scenario(dev, bank, tables, g, interp, segs, force_generic=False, fm2=0)
    omix, ostate, ko = _run_scenario(dev, bank, tables, g, interp, segs, force_generic=False, fm2=1 << 30)
    ref_mix, ref_state = _oracle_scenario(bank, tables, g, interp, segs)
    assert k == [3] * len(segs) and ko == [1] * len(segs), (k, ko)
    assert not ostate.rw_equal(ref_state), ostate.rw_equal(ref_state)
    assert not state.rw_equal(ref_state), state.rw_equal(ref_state)
    assert rel_rms(mix, ref_mix) <= 1e-5 and rel_rms(omix, ref_mix) <= 1e-5
    # a bank in which ONE carrier looks further up is not of that shape: the one-per-lane kernel keeps it
    bank["voice_freq_mod_osc"][car[0]] = car[0] + 2
    _, _, k2 = _run_scenario(dev, bank, tables, g, interp, segs[:1], force_generic=False, fm2=0)
    assert k2 == [1]


@pytest.mark.parametrize("recipe,interp", [("c2", 0), ("c4", 1)])
def test_previous_frame_fm_on_the_specialised_kernel(dev, recipe, interp):
    """Two-operator FM the way the reference's patches write it (`v0 ... F1,depth` / `v1 ... m1`), plus `A` and `P`
    modulation of the same kind: the modulator has the higher index, so the carrier reads its voice_sample of the
    previous frame (synth.c:548-555,584-587,597-602 in index order).
    The one-per-lane kernel serves such banks (ds_bpermute exchange inside the 64-voice group) -- against the oracle
    and against the modulated kernel; deep modulation drives increments negative and beyond a loop length; some
    carriers are stopping one-shots; a modulator is switched off (amp 0) between launches."""
    n = 3072
    bank, tables, g = banks.RECIPES[recipe](n)
    car = np.arange(0, n, 2)
    mod = car + 1
    bank["voice_freq_mod_osc"][car] = mod
    bank["voice_freq_mod_depth"][car] = (np.float32(0.05) * (1 + (car % 97))).astype(np.float32)     # up to ~5: wild
    bank["voice_freq_scale"][car] = (np.float32(0.5) + np.float32(0.01) * (car % 50)).astype(np.float32)
    bank["voice_disconnect"][mod] = 1                                                               # `m1`
    far = car[car % 64 < 32][::9]                       # some modulators further up in the same group
    bank["voice_freq_mod_osc"][far] = far + 31
    stops = car[5::40]
    bank["voice_one_shot"][stops] = 1
    bank["voice_loop_enabled"][stops] = 0
    # amplitude and pan modulation with the same previous-frame semantics (`A`, `P` with a higher-indexed voice),
    # and by the voice itself (same-frame, but no other lane involved)
    am = car[3::11]
    bank["voice_amp_mod_osc"][am] = am + 1
    bank["voice_amp_mod_depth"][am] = np.float32(3.0)
    pm = car[7::13]
    bank["voice_pan_mod_osc"][pm] = np.minimum(pm + 5, (pm // 64) * 64 + 63)
    bank["voice_pan_mod_depth"][pm] = np.float32(8.0)
    self_am = car[9::29]
    bank["voice_amp_mod_osc"][self_am] = self_am
    bank["voice_amp_mod_depth"][self_am] = np.float32(2.5)
    self_pm = car[11::31]
    bank["voice_pan_mod_osc"][self_pm] = self_pm
    bank["voice_pan_mod_depth"][self_pm] = np.float32(6.0)
    bank["voice_disconnect"][pm[::4]] = 1               # a muted voice's pan is not modulated (synth.c:596)

    def kill_some_modulators(host, now):
        host["voice_amp"][mod[::7]] = 0.0

    segs = [(300, None), (257, _release_odd_voices), (700, kill_some_modulators), (64, None)]
    mix, state, k = _run_scenario(dev, bank, tables, g, interp, segs, force_generic=False)
    mmix, mstate, km = _run_scenario(dev, bank, tables, g, interp, segs, force_generic=True)
    ref_mix, ref_state = _oracle_scenario(bank, tables, g, interp, segs)
    assert k == [1] * len(segs) and km == [2] * len(segs), (k, km)
    assert not mstate.rw_equal(ref_state), mstate.rw_equal(ref_state)
    assert not state.rw_equal(ref_state), state.rw_equal(ref_state)
    assert rel_rms(mix, ref_mix) <= 1e-5
    # a modulator BELOW its carrier is a same-frame dependency: the whole bank goes to the modulated kernel
    bank["voice_freq_mod_osc"][10] = 3
    _, _, k2 = _run_scenario(dev, bank, tables, g, interp, [(64, None)], force_generic=False)
    assert k2 == [2]


def _clean_fuzz_bank(rng):
    """A random bank of the specialised kernels' family: LDS or L2 tables, filter for all or none, envelope for all
    or none, optionally forward one-shots that stop, tame and wild increments, mutes, zero amps, odd sizes."""
    recipe = ["c2", "c4"][int(rng.integers(0, 2))]
    n = int(rng.integers(130, 2200))
    bank, tables, g = banks.RECIPES[recipe](n)
    mixed = rng.random() < 0.35                       # filter / envelope on some voices only (extended instantiation)
    if rng.random() < 0.5 and not mixed:
        bank["voice_filter_mode"][:] = 0
    elif recipe == "c4":
        bank["voice_filter_mode"][:] = 1 + (np.arange(n) % 4)
        c = banks.biquad_coeffs(bank["voice_filter_mode"], 200.0 + 37.0 * (np.arange(n) % 150), np.full(n, 0.9, np.float32), 48000)
        for k in ("b0", "b1", "b2", "a1", "a2"):
            bank["voice_filter"][k] = c[k]
    if rng.random() < 0.4 and not mixed:
        bank["voice_use_amp_envelope"][:] = 0
    stops = rng.random() < 0.5
    if mixed:                                         # (both specialised kernel families carry per-lane flags for this)
        bank["voice_filter_mode"][rng.random(n) < 0.4] = 0
        bank["voice_use_amp_envelope"][rng.random(n) < 0.4] = 0
    if stops:
        sel = rng.random(n) < 0.3
        bank["voice_one_shot"][sel] = 1
        bank["voice_loop_enabled"][sel] = 0
        size = bank["voice_table_size"].astype(np.float32)
        inc = np.maximum(bank["voice_phase_inc"], np.float32(0.125))
        bank["voice_phase_inc"][sel] = inc[sel]
        left = rng.integers(1, 3000, n).astype(np.float32)
        bank["voice_phase"][sel] = np.maximum(size[sel] - left[sel] * inc[sel], np.float32(0.0))
    wild = rng.random(n) < 0.05                       # increments of several loop lengths: the general wrap
    span = np.where(bank["voice_loop_enabled"] & bank["voice_loop_valid"], bank["voice_loop_end_f"] - bank["voice_loop_start_f"],
                    bank["voice_table_size"].astype(np.float32)).astype(np.float32)
    if not stops:
        bank["voice_phase_inc"][wild] = (span[wild] * np.float32(1.7)).astype(np.float32)
    bank["voice_disconnect"][rng.random(n) < 0.03] = 1
    bank["voice_amp"][rng.random(n) < 0.03] = 0.0
    if rng.random() < 0.4:                            # reverse playback on a random subset (extended instantiation)
        bank["voice_direction"][rng.random(n) < 0.25] = 1
        stops = True
    if rng.random() < 0.4:                            # sample & hold, crush, smoother off
        bank["voice_sample_hold_max"][rng.random(n) < 0.1] = int(rng.integers(1, 9))
        bank["voice_quantize"][rng.random(n) < 0.1] = int(rng.integers(1, 14))
        bank["voice_smoother_enable"][rng.random(n) < 0.1] = 0
        stops = True
    fm = rng.random() < 0.4                           # previous-frame FM: modulator above its carrier, same 64-voice group
    if fm:
        v = np.arange(n)
        car = v[(rng.random(n) < 0.35) & (v % 64 < 63)]
        up = rng.integers(1, 64, len(car))
        m = np.minimum(car + up, (car // 64) * 64 + 63)
        m = np.minimum(m, n - 1)
        ok = m > car
        bank["voice_freq_mod_osc"][car[ok]] = m[ok]
        bank["voice_freq_mod_depth"][car[ok]] = (rng.random(int(ok.sum())) * 3.0).astype(np.float32)
        bank["voice_freq_scale"][car[ok]] = (0.25 + rng.random(int(ok.sum()))).astype(np.float32)
        for osc, depth, top in (("voice_amp_mod_osc", "voice_amp_mod_depth", 3.0), ("voice_pan_mod_osc", "voice_pan_mod_depth", 9.0)):
            c2 = v[(rng.random(n) < 0.15) & (v % 64 < 63)]
            tgt = np.minimum(c2 + rng.integers(0, 20, len(c2)), (c2 // 64) * 64 + 63)     # +0: the voice itself
            tgt = np.minimum(tgt, n - 1)
            bank[osc][c2] = tgt
            bank[depth][c2] = (rng.random(len(c2)) * top).astype(np.float32)
    if rng.random() < 0.4:                            # w6 voices: the frame's shared LCG draw (extended instantiation)
        bank["voice_wave_table_index"][rng.random(n) < 0.06] = 6
        g.noise_rng = int(rng.integers(1, 1 << 62))
        stops = True
    return recipe, bank, tables, g, stops or fm


@pytest.mark.parametrize("seed", list(range(int(os.environ.get("SKRED_FUZZ_SEEDS", "6")))))
def test_clean_family_fuzz_vs_oracle(dev, seed):
    """Fuzz of the specialised kernels (one / two voices per lane, LDS tables / table windows, stopping one-shots, FM, noise):
    random clean banks, random block lengths, note-offs and re-triggers in between; per-voice state bit-exact."""
    rng = np.random.default_rng(1000 + seed)
    recipe, bank, tables, g, stops = _clean_fuzz_bank(rng)
    interp = int(rng.integers(0, 2))

    def retrigger(host, now):
        vs = np.arange(3, host.n, 17)
        host["voice_finished"][vs] = 0
        host["voice_phase"][vs] = np.where(host["voice_loop_enabled"][vs] != 0, host["voice_loop_start_f"][vs], np.float32(0.0))
        e = host["voice_amp_envelope"]
        e["sample_start"][vs] = now
        e["sample_release"][vs] = 0
        e["is_active"][vs] = 1

    events = [None, _release_odd_voices, None, retrigger, None]
    segs = [(int(rng.integers(1, 1400)), ev) for ev in events]
    ref_mix, ref_state = _oracle_scenario(bank, tables, g, interp, segs)
    for fast2 in ((False,) if stops else (False, True)):
        mix, state, k = _run_scenario(dev, bank, tables, g, interp, segs, force_generic=False, fast2=fast2)
        assert set(k) == {3 if fast2 else 1}, (k, recipe, stops)
        assert not state.rw_equal(ref_state), (state.rw_equal(ref_state), recipe, stops, fast2, interp)
        assert rel_rms(mix, ref_mix) <= 1e-5


@pytest.mark.parametrize("case", ["c4_pcm_oneshot", "c1_sine_adsr64", "c2_mixed_filter64"])
def test_golden_case_on_the_specialised_kernel(dev, case):
    """The reference's fixtures without stems, so that the bank takes the one-per-lane kernel: c4_pcm_oneshot is
    forward / reverse / looped / reverse-looped one-shots, several finishing mid-block (the extended instantiation).
    State after every segment bit-exact against the reference, mix within tolerance."""
    g = gio.load(case)
    for seg in g.segments:
        db = dev.DeviceBank(seg.bank_in.n)
        db.set_tables(g.tables)
        db.upload(seg.bank_in)
        db.set_globals(seg.g_in)
        mix = np.zeros((seg.frames, 2), np.float32)
        p = 0
        while p < seg.frames:
            n = min(seg.block, seg.frames - p)
            mix[p:p + n], _ = db.render_host(n, 2, 0)
            assert db.last_kernel() == 1, case
            p += n
        got = seg.bank_in.copy()
        db.download(got)
        db.close()
        bad = got.rw_equal(gio.expected_out_bank(seg))
        assert not bad, f"{case} seg{seg.index}: voice state differs {bad}"
        assert rms(mix.astype(np.float64) - seg.mix.astype(np.float64)) <= MIX_RMS_TOL


def test_reverse_playback_on_the_specialised_kernel(dev):
    """`b1`: reversed loops and reversed one-shots (which finish at the loop start), with and without FM."""
    n = 2048
    bank, tables, g = banks.bank_c4(n)
    rev = np.arange(n) % 3 == 1
    bank["voice_direction"][rev] = 1
    stop = np.arange(n) % 6 == 1                          # a subset of the reversed voices stops
    bank["voice_one_shot"][stop] = 1
    bank["voice_loop_enabled"][stop] = 0
    size = bank["voice_table_size"].astype(np.float32)
    bank["voice_phase"][stop] = np.minimum(size[stop] - 1.0, np.float32(3.0) + np.float32(9.0) * (np.arange(n)[stop] % 97))
    car = np.arange(0, n, 8)
    bank["voice_freq_mod_osc"][car] = car + 2
    bank["voice_freq_mod_depth"][car] = np.float32(1.5)
    bank["voice_freq_scale"][car] = np.float32(1.0)
    segs = [(500, None), (77, _release_odd_voices), (900, None)]
    for interp in (0, 1):
        mix, state, k = _run_scenario(dev, bank, tables, g, interp, segs, force_generic=False)
        ref_mix, ref_state = _oracle_scenario(bank, tables, g, interp, segs)
        assert k == [1] * len(segs), k
        assert int(ref_state["voice_finished"].sum()) > 100
        assert not state.rw_equal(ref_state), state.rw_equal(ref_state)
        assert rel_rms(mix, ref_mix) <= 1e-5


@pytest.mark.parametrize("recipe,interp", [("c2", 0), ("c4", 1)])
def test_partly_filtered_partly_enveloped_bank_on_the_specialised_kernel(dev, recipe, interp):
    """A bank in which only some voices run the biquad and only some use the envelope (the usual case outside
    benchmarks): per-lane flags in the one-per-lane kernel's extended instantiation.  An unfiltered voice's delay line
    and an un-enveloped voice's is_active must come back untouched."""
    n = 3000
    bank, tables, g = banks.RECIPES[recipe](n)
    v = np.arange(n)
    if recipe == "c4":
        bank["voice_filter_mode"][:] = 1 + (v % 4)
        c = banks.biquad_coeffs(bank["voice_filter_mode"], 300.0 + 29.0 * (v % 200), np.full(n, 1.1, np.float32), 48000)
        for k in ("b0", "b1", "b2", "a1", "a2"):
            bank["voice_filter"][k] = c[k]
    bank["voice_filter_mode"][v % 3 == 0] = 0
    f = bank["voice_filter"]
    f["x1"][:] = (0.01 * (v % 17)).astype(np.float32)      # recognisable delay-line content everywhere
    f["y2"][:] = (-0.02 * (v % 13)).astype(np.float32)
    bank["voice_use_amp_envelope"][v % 4 == 1] = 0
    segs = [(301, None), (333, _release_odd_voices), (1001, None), (64, None)]
    mix, state, k = _run_scenario(dev, bank, tables, g, interp, segs, force_generic=False)
    gmix, gstate, kg = _run_scenario(dev, bank, tables, g, interp, segs, force_generic=True)
    mix2, state2, k2 = _run_scenario(dev, bank, tables, g, interp, segs, force_generic=False, fast2=True)
    ref_mix, ref_state = _oracle_scenario(bank, tables, g, interp, segs)
    assert k == [1] * len(segs) and kg == [0] * len(segs) and k2 == [3] * len(segs), (k, kg, k2)
    assert not gstate.rw_equal(ref_state), gstate.rw_equal(ref_state)
    assert not state.rw_equal(ref_state), state.rw_equal(ref_state)
    assert not state2.rw_equal(ref_state), state2.rw_equal(ref_state)      # the two-per-lane kernels (MIXED instantiation)
    assert rel_rms(mix, ref_mix) <= 1e-5 and rel_rms(mix2, ref_mix) <= 1e-5


def test_hold_crush_and_unsmoothed_voices_on_the_specialised_kernel(dev):
    """`h` (sample & hold), `q` (bit-crush: the +0.5 is a double add, synth.c:343) and `s0` (amp smoother off) on the
    one-per-lane kernel's extended instantiation, alone and combined, with a stopping voice among them."""
    n = 2048
    bank, tables, g = banks.bank_c2(n)
    v = np.arange(n)
    bank["voice_sample_hold_max"][v % 5 == 1] = 1 + (v[v % 5 == 1] % 7)
    bank["voice_quantize"][v % 7 == 2] = 1 + (v[v % 7 == 2] % 12)
    bank["voice_smoother_enable"][v % 9 == 3] = 0
    stop = v % 37 == 4
    bank["voice_one_shot"][stop] = 1
    bank["voice_loop_enabled"][stop] = 0
    segs = [(333, None), (100, _release_odd_voices), (800, None)]
    for interp in (0, 1):
        mix, state, k = _run_scenario(dev, bank, tables, g, interp, segs, force_generic=False)
        gmix, gstate, kg = _run_scenario(dev, bank, tables, g, interp, segs, force_generic=True)
        ref_mix, ref_state = _oracle_scenario(bank, tables, g, interp, segs)
        assert k == [1] * len(segs) and kg == [0] * len(segs), (k, kg)
        assert not gstate.rw_equal(ref_state), gstate.rw_equal(ref_state)
        assert not state.rw_equal(ref_state), state.rw_equal(ref_state)
        assert rel_rms(mix, ref_mix) <= 1e-5


def test_exotic_voice_forces_generic_kernel(dev):
    bank, tables, g = banks.bank_c2(512)
    bank["voice_phase_inc"][5] = np.inf             # osc_next's !isfinite() branch: only the generic kernel has it
    mix, state, k = _run_scenario(dev, bank, tables, g, 0, [(64, None)], force_generic=False)
    ref_mix, ref_state = _oracle_scenario(bank, tables, g, 0, [(64, None)])
    assert k == [0]
    assert not state.rw_equal(ref_state), state.rw_equal(ref_state)


@pytest.mark.parametrize("recipe,interp", [("c2", 0), ("c4", 1)])
def test_noise_voices_on_the_specialised_kernel(dev, recipe, interp):
    """w6 voices (synth.c:543-546) take the frame's shared LCG draw instead of running their oscillator: phase
    untouched, frequency modulation ignored, everything downstream (sample & hold, crush, biquad, envelope, smoother,
    pan) as usual.  The one-per-lane kernel's extended instantiation steps the LCG once per frame in the waves that hold
    such a voice; banks with noise voices no longer need the generic kernel.  Among the noise voices: a one-shot (never
    finishes), a reversed one, one with a wild phase, a sample-and-hold one, a modulated one, a muted modulator."""
    n = 1500
    bank, tables, g = banks.RECIPES[recipe](n)
    v = np.arange(n)
    noisy = (v % 13 == 4) & (v > 200)               # the first waves stay free of noise voices
    bank["voice_wave_table_index"][noisy] = 6
    idx = v[noisy]
    bank["voice_one_shot"][idx[0]] = 1; bank["voice_loop_enabled"][idx[0]] = 0
    bank["voice_direction"][idx[1]] = 1
    bank["voice_phase"][idx[2]] = np.float32(1e9)
    bank["voice_sample_hold_max"][idx[3:40:3]] = 5
    bank["voice_quantize"][idx[4:40:3]] = 6
    car = idx[5]                                    # frequency modulation of a noise voice is ignored
    bank["voice_freq_mod_osc"][car] = car + 1; bank["voice_freq_mod_depth"][car] = 2.0
    car2 = idx[6] - 1                               # a noise voice as (muted) modulator of the voice below it
    bank["voice_freq_mod_osc"][car2] = car2 + 1; bank["voice_freq_mod_depth"][car2] = 0.3
    bank["voice_disconnect"][car2 + 1] = 1
    g.noise_rng = 0x1234567
    segs = [(301, None), (333, _release_odd_voices), (700, None), (64, None)]
    mix, state, k = _run_scenario(dev, bank, tables, g, interp, segs, force_generic=False)
    gmix, gstate, kg = _run_scenario(dev, bank, tables, g, interp, segs, force_generic=True)   # (modulators: the modulated kernel)
    ref_mix, ref_state = _oracle_scenario(bank, tables, g, interp, segs)
    assert k == [1] * len(segs) and kg == [2] * len(segs), (k, kg)
    assert not gstate.rw_equal(ref_state), gstate.rw_equal(ref_state)
    assert not state.rw_equal(ref_state), state.rw_equal(ref_state)
    assert rel_rms(mix, ref_mix) <= 1e-5


def _without_guards(bank, tables):
    """The same pool with every guard sample (the float behind a table: banks.py) overwritten: no voice is SKF_GUARD any more, the
    linear lookup takes its general form (fold test at the loop end).  The oracle never reads those floats."""
    t = tables.copy()
    pos = np.unique(bank["voice_table_offset"].astype(np.int64) + bank["voice_table_size"].astype(np.int64))
    t[pos[pos < len(t)]] = np.float32(7.0)
    return t


@pytest.mark.parametrize("recipe,interp,one_voice", [("c1", 1, True), ("c2", 1, True), ("c2", 1, False)])
def test_linear_lookup_guarded_and_general_forms_agree(dev, recipe, interp, one_voice):
    """Linear interpolation on LUT banks has two instantiations: without the fold test when every voice loops over a whole table
    that has a guard sample behind it (INTERP == 2), and the general one.  Same bank, pool with and without guards: per-voice
    state bit for bit equal to each other and to the oracle, on the one-voice and the two-per-lane kernels."""
    n = 5000
    bank, tables, g = banks.RECIPES[recipe](n)
    ref_bank, ref_g = bank.copy(), g.copy()
    r = cpuref.render(ref_bank, ref_g, tables, 700, interp)
    ref_mix = cpuref.master(ref_g, r["sum64"].astype(np.float32))
    for pool in (tables, _without_guards(bank, tables)):
        db = dev.DeviceBank(n)
        db.set_tables(pool)
        db.upload(bank)
        db.set_globals(g)
        db.fast2_min_voices(1 << 30 if one_voice else 0)
        mix, _ = db.render_host(700, 2, interp)
        assert db.last_kernel() == (1 if one_voice else 3)
        got = bank.copy()
        db.download(got)
        db.close()
        assert not got.rw_equal(ref_bank), got.rw_equal(ref_bank)
        assert rel_rms(mix, ref_mix) <= 1e-5


@pytest.mark.parametrize("recipe,interp", [("c1", 0), ("c2", 0), ("c2", 1), ("c4", 0), ("c4", 1)])
def test_two_per_lane_kernel_matches_oracle(dev, recipe, interp):
    """sk_render_fast2_kernel (two voices per lane, packed fp32): per-voice state bit-exact against the
    oracle through attack/decay, note-off, release end and the smoother tail; mix within tolerance
    (its reduction tree adds the lane's two voices first, so mix bits differ from the other kernels)."""
    n = 5000
    bank, tables, g = banks.RECIPES[recipe](n)
    bank["voice_disconnect"][::7] = 1
    bank["voice_amp"][::11] = 0.0
    segs = [(301, None), (333, _release_odd_voices), (12001, None), (64, None)]
    db = dev.DeviceBank(bank.n)
    db.set_tables(tables)
    host = bank.copy()
    db.upload(host)
    db.set_globals(g)
    db.fast2_min_voices(0)
    mixes, kernels = [], []
    for frames, event in segs:
        if event is not None:
            db.download(host)
            event(host, db.get_globals().synth_sample_count)
            db.upload(host)
        mixes.append(db.render_host(frames, 2, interp)[0])
        kernels.append(db.last_kernel())
    db.download(host)
    db.close()
    ref_mix, ref_state = _oracle_scenario(bank, tables, g, interp, segs)
    assert kernels == [3] * len(segs), kernels
    assert not host.rw_equal(ref_state), host.rw_equal(ref_state)
    assert rel_rms(np.concatenate(mixes), ref_mix) <= 1e-5


@pytest.mark.parametrize("recipe,interp,fast2", [("c2", 0, True), ("c2", 0, False), ("c1", 1, True), ("c4", 1, True), ("c4", 1, False)])
def test_envelope_ramps_segment_by_segment(dev, recipe, interp, fast2):
    """The envelope ramps of the block paths (sk_render_env2_kernel's refined-reciprocal ramp e = C*(A + B*q) in place
    of the reference's per-sample divisions, and the one-voice kernel's clock-as-float path) checked where a slip
    would show: the read-write state of every voice -- voice_smoother_gain follows the envelope within a few frames --
    is downloaded and compared bit for bit with the oracle after EVERY one of 48 short ragged segments (7..97 frames)
    that walk through attack, decay, sustain, note-off, release and its end, with re-triggers in between."""
    n = 5200
    bank, tables, g = banks.RECIPES[recipe](n)
    e = bank["voice_amp_envelope"]
    now0 = g.synth_sample_count
    # short, different stage lengths per voice so that every segment catches voices in every stage, and stage changes
    # fall inside 8-frame blocks as well as on their edges; all notes start within the last 40 frames
    v = np.arange(n)
    e["attack_time"][:] = (3.0 + (v % 89) * 1.37).astype(np.float32)
    e["decay_time"][:] = (5.0 + (v % 61) * 2.11).astype(np.float32)
    e["release_time"][:] = (11.0 + (v % 131) * 3.3).astype(np.float32)
    e["sustain_level"][:] = (0.15 + 0.8 * ((v * 7) % 19) / 19.0).astype(np.float32)
    e["sample_start"][:] = (now0 - (v % 40)).astype(np.uint64)
    e["sample_release"][:] = 0
    e["is_active"][:] = 1
    e["attack_time"][::53] = 0.0                      # no attack at all
    e["decay_time"][3::59] = 0.0                      # ... no decay
    e["release_time"][5::67] = 0.0                    # release that ends at once
    bank["voice_disconnect"][::7] = 1
    bank["voice_amp"][::11] = 0.0
    lengths = [7 + (k * 37) % 91 for k in range(48)]
    assert min(lengths) >= 7 and max(lengths) <= 97

    def event_for(k):
        if k == 9:
            return _release_odd_voices
        if k == 20:
            def retrigger(host, now):                 # == amp_envelope_trigger (synth.c:383-388) on a third of the bank
                ev = host["voice_amp_envelope"]
                ev["sample_start"][::3] = now
                ev["sample_release"][::3] = 0
                ev["is_active"][::3] = 1
            return retrigger
        if k == 30:
            def release_all(host, now):
                ev = host["voice_amp_envelope"]
                live = ev["is_active"] != 0
                ev["sample_release"][live] = now
            return release_all
        return None

    db = dev.DeviceBank(n)
    db.set_tables(tables)
    host = bank.copy()
    db.upload(host)
    db.set_globals(g)
    db.fast2_min_voices(0 if fast2 else 1 << 30)
    ref_host, ref_g = bank.copy(), g.copy()
    mixes, ref_mixes = [], []
    for k, frames in enumerate(lengths):
        ev = event_for(k)
        if ev is not None:
            db.download(host)
            ev(host, db.get_globals().synth_sample_count)
            db.upload(host)
            ev(ref_host, ref_g.synth_sample_count)
        mixes.append(db.render_host(frames, 2, interp)[0])
        assert db.last_kernel() == (3 if fast2 else 1)
        r = cpuref.render(ref_host, ref_g, tables, frames, interp)
        ref_mixes.append(cpuref.master(ref_g, r["sum64"].astype(np.float32)))
        db.download(host)
        bad = host.rw_equal(ref_host)
        assert not bad, f"segment {k} ({frames} frames): state differs from the oracle: {bad}"
    assert db.list_violations() == 0
    db.close()
    act = ref_host["voice_amp_envelope"]["is_active"]
    assert int((act == 0).sum()) > n // 2                    # the releases did run out
    assert rel_rms(np.concatenate(mixes), np.concatenate(ref_mixes)) <= 1e-5


def _late_note_bank(n):
    """C2 recipe, everyone long in sustain, every 97th voice with its note-on scheduled 5-9 blocks AHEAD of the clock."""
    bank, tables, g = banks.bank_c2(n)
    e = bank["voice_amp_envelope"]
    now0 = g.synth_sample_count
    e["sample_start"][:] = np.uint64(now0 - 20000)
    e["sample_release"][:] = 0
    e["is_active"][:] = 1
    late = np.arange(17, n, 97)
    e["sample_start"][late] = (now0 + 2500 + (late % 7) * 300).astype(np.uint64)
    return bank, tables, g, late


@pytest.mark.parametrize("two_per_lane", [True, False])
def test_note_on_ahead_of_the_clock(dev, two_per_lane):
    """A host that schedules a note by writing sample_start AHEAD of the clock: the reference reads the wrapped clock
    difference as a huge elapsed time (sustain) until the clock catches up, then the attack starts (synth.c:401) -- the
    one way an envelope leaves a constant level without a control action.  Both specialised families render that themselves
    (round 3; the bank used to go to the generic kernel meanwhile): the one-voice kernel walks such a wave's chunks on integer
    clocks, and on the two-per-lane path such a voice stays on the device's motion list -- rendered by the envelope kernel --
    until its note has started and come to rest.  Against the oracle, block by block, through the attacks' start."""
    import torch
    n = 6000
    bank, tables, g, _ = _late_note_bank(n)
    db = dev.DeviceBank(n)
    db.set_tables(tables)
    host = bank.copy()
    db.upload(host)
    db.set_globals(g)
    db.fast2_min_voices(0 if two_per_lane else 1 << 30)
    ref_host, ref_g = bank.copy(), g.copy()
    out = torch.zeros(512, 2, device="cuda")
    kernels = []
    for k in range(14):
        db.render_mix(512, out.data_ptr(), 2, 0, 0)           # asynchronous blocks: the way the host gets its reports
        kernels.append(db.last_kernel())
        torch.cuda.synchronize()
        r = cpuref.render(ref_host, ref_g, tables, 512, 0)
        ref_mix = cpuref.master(ref_g, r["sum64"].astype(np.float32))
        assert rel_rms(out.cpu().numpy(), ref_mix) <= 1e-5, f"block {k}"
    db.download(host)
    assert db.list_violations() == 0
    db.close()
    assert not host.rw_equal(ref_host), host.rw_equal(ref_host)
    assert kernels == [3 if two_per_lane else 1] * 14, kernels


@pytest.mark.parametrize("two_per_lane", [True, False])
def test_pending_note_on_rescheduled_and_bank_reused(dev, two_per_lane):
    """Two call orders that a host-side "is any note-on pending" bookkeeping got wrong (round 2's advisor findings; the
    bookkeeping is gone, the kernels look at the clocks themselves).  (1) A pending note-on is moved to a LATER clock with
    SKRED_DIRTY_ENV_CLOCK while it is pending, and some are moved again after the first ones have started.  (2) The bank is
    re-used for a new session: a second upload with note-ons scheduled ahead of a clock that set_globals() then RESETS to 0,
    i.e. behind the device's old clock.  Oracle, block by block; one update travels on a stream of its own."""
    import torch
    n = 6000
    bank, tables, g, late = _late_note_bank(n)
    db = dev.DeviceBank(n)
    db.set_tables(tables)
    host = bank.copy()
    db.upload(host)
    db.set_globals(g)
    db.fast2_min_voices(0 if two_per_lane else 1 << 30)
    ref_host, ref_g = bank.copy(), g.copy()
    out = torch.zeros(512, 2, device="cuda")
    other = torch.cuda.Stream()

    def block(tag):
        db.render_mix(512, out.data_ptr(), 2, 0, 0)
        torch.cuda.synchronize()
        r = cpuref.render(ref_host, ref_g, tables, 512, 0)
        ref_mix = cpuref.master(ref_g, r["sum64"].astype(np.float32))
        assert rel_rms(out.cpu().numpy(), ref_mix) <= 1e-5, tag

    for k in range(3):
        block(f"a{k}")
    # (1) every second pending note-on moves 6 blocks further out -- through a stream of its own, finished before the next block
    moved = late[::2]
    for hb in (host, ref_host):
        hb["voice_amp_envelope"]["sample_start"][moved] += np.uint64(6 * 512)
    db.update(host, moved, dev.DIRTY_ENV_CLOCK, stream=other.cuda_stream)
    other.synchronize()
    for k in range(8):
        block(f"b{k}")
    again = moved[::3]                                          # ... and a few of those once more, after the others have started
    for hb in (host, ref_host):
        hb["voice_amp_envelope"]["sample_start"][again] += np.uint64(5 * 512 + 77)
    db.update(host, again, dev.DIRTY_ENV_CLOCK)
    for k in range(12):
        block(f"c{k}")
    got = host.copy()
    db.download(got)
    assert not got.rw_equal(ref_host), got.rw_equal(ref_host)
    # (2) a new session on the same bank: upload first, then the clock goes back to 0
    bank2, _, g2, _ = _late_note_bank(n)
    e2 = bank2["voice_amp_envelope"]
    e2["sample_start"][:] = 0
    late2 = np.arange(5, n, 61)
    e2["sample_start"][late2] = (700 + (late2 % 5) * 400).astype(np.uint64)
    g2.synth_sample_count = 0
    host, ref_host, ref_g = bank2.copy(), bank2.copy(), g2.copy()
    db.upload(host)
    db.set_globals(g2)
    for k in range(9):
        block(f"d{k}")
    db.download(host)
    assert db.list_violations() == 0
    db.close()
    assert not host.rw_equal(ref_host), host.rw_equal(ref_host)


@pytest.mark.parametrize("interp,fast2", [(0, True), (1, True), (0, False), (1, False)])
def test_table_window_edges(dev, interp, fast2):
    """PCM bank (pool in L2/HBM) on the specialised kernels (two voices per lane / one), which serve tame waves from
    per-voice LDS table windows refilled every 8 frames: voices that wrap in (almost) every block (loops of 9..40 samples), voices
    faster than a window can cover (> 2.1875 samples per frame), voices parked just below their loop end
    (second tap folds back to the loop start), a voice on the last table of the pool (window reads into the
    pool's padding) -- all must equal the oracle bit for bit per voice."""
    n = 4096
    bank, tables, g = banks.bank_c4(n)
    lo = bank["voice_loop_start_f"]
    hi = bank["voice_loop_end_f"]
    size = bank["voice_table_size"].astype(np.float32)
    tiny = np.arange(n) % 5 == 0                                  # short loops: every block has wrapping lanes
    span = (9 + (np.arange(n) % 32)).astype(np.float32)
    hi[tiny] = np.minimum(lo[tiny] + span[tiny], size[tiny])
    bank["voice_loop_valid"][tiny] = 1
    bank["voice_phase"][tiny] = lo[tiny]
    fast = np.arange(n) % 7 == 3                                  # faster than the window, still <= span / 2
    bank["voice_phase_inc"][fast & ~tiny] = np.float32(2.19) + np.float32(0.013) * (np.arange(n)[fast & ~tiny] % 200)
    park = np.arange(n) % 11 == 5                                 # sits in [hi - 1, hi): folded second tap
    sel = park & ~tiny
    bank["voice_phase"][sel] = hi[sel] - np.float32(0.75)
    bank["voice_phase_inc"][sel] = np.float32(0.015625)
    last = np.argmax(bank["voice_table_offset"])                  # last table of the pool, phase near its end
    bank["voice_loop_enabled"][last] = 0
    bank["voice_one_shot"][last] = 0
    bank["voice_phase"][last] = np.float32(bank["voice_table_size"][last] - 3)
    bank["voice_phase_inc"][last] = np.float32(0.01)
    segs = [(512, None), (77, None), (1024, None)]
    fast_mix, fast_state, k = _run_scenario(dev, bank, tables, g, interp, segs, force_generic=False, fast2=fast2)
    ref_mix, ref_state = _oracle_scenario(bank, tables, g, interp, segs)
    assert k == [3 if fast2 else 1] * len(segs), k
    assert not fast_state.rw_equal(ref_state), fast_state.rw_equal(ref_state)
    assert rel_rms(fast_mix, ref_mix) <= 1e-5


@pytest.mark.parametrize("recipe,interp,fast2", [("c2", 0, True), ("c4", 1, False)])
def test_stalled_smoother_is_skipped_exactly(dev, recipe, interp, fast2):
    """Held notes: after ~1000 frames of constant gain the one-pole amp smoother stops moving and the specialised
    kernels (two-per-lane LDS blocks; one-per-lane table-window blocks) stop evaluating it (per 64-frame chunk, when
    every voice of the wave has stalled).  State and samples
    must stay those of the oracle through the stall, through an amplitude change that un-stalls it, and through the
    tail of released notes decaying to zero gain."""
    n = 4096
    bank, tables, g = banks.RECIPES[recipe](n)

    def louder(host, now):
        host["voice_amp"][::3] *= np.float32(1.5)

    segs = [(6400, None), (512, None), (640, louder), (2048, None), (512, _release_odd_voices), (16000, None), (512, None)]
    mix, state, k = _run_scenario(dev, bank, tables, g, interp, segs, force_generic=False, fast2=fast2)
    ref_mix, ref_state = _oracle_scenario(bank, tables, g, interp, segs)
    assert k == [3 if fast2 else 1] * len(segs), k
    assert not state.rw_equal(ref_state), state.rw_equal(ref_state)
    assert rel_rms(mix, ref_mix) <= 1e-5


def test_two_per_lane_kernel_full_size_c3(dev):
    """BASELINE config-3 bank (2^20 voices) on the kernel bench.py uses for it, 64 frames vs the oracle."""
    n = 1 << 20
    bank, tables, g = banks.bank_c2(n)
    ref_bank, ref_g = bank.copy(), g.copy()
    r = cpuref.render(ref_bank, ref_g, tables, 64, 0)
    ref_mix = cpuref.master(ref_g, r["sum64"].astype(np.float32))
    db = dev.DeviceBank(n)
    db.set_tables(tables)
    db.upload(bank)
    db.set_globals(g)
    mix, _ = db.render_host(64)
    assert db.last_kernel() == 3
    got = bank.copy()
    db.download(got)
    db.close()
    assert not got.rw_equal(ref_bank), got.rw_equal(ref_bank)
    assert rel_rms(mix, ref_mix) <= 1e-5


def test_headline_block_at_full_size(dev):
    """The exact block bench.py's headline times: BASELINE config 3 (2^20 voices), the recipe's 5 632 warm-up frames rendered
    in 512-frame blocks (envelopes in motion: the envelope kernel beside the steady one, then the list runs empty), then ONE
    512-frame all-sustain block on the steady two-per-lane kernel alone -- per-voice state of all 2^20 voices bit for bit and
    the block's mix against the oracle (16 threads, parity flags)."""
    import torch
    n, F = 1 << 20, 512
    bank, tables, g = banks.bank_c2(n)
    ref_bank, ref_g = bank.copy(), g.copy()
    threads = min(16, os.cpu_count() or 1)
    warm = cpuref.render_mt(ref_bank, ref_g, tables, 11 * F, threads, 0, fast=False)
    cpuref.master(ref_g, warm.astype(np.float32))              # (the master gain's recurrence runs through the warm-up too)
    ref_sum = cpuref.render_mt(ref_bank, ref_g, tables, F, threads, 0, fast=False)
    ref_mix = cpuref.master(ref_g, ref_sum.astype(np.float32))
    db = dev.DeviceBank(n)
    db.set_tables(tables)
    db.upload(bank)
    db.set_globals(g)
    out = torch.zeros(F, 2, device="cuda")
    for _ in range(11):
        db.render_mix(F, out.data_ptr(), 2, 0, 0)
    torch.cuda.synchronize()
    # ... and, for 64 of its voices, EVERY FRAME of that block from inside the steady kernel's 8-frame tile blocks
    # (skred_bank_set_probe; tests/test_probe.py has the other regimes): bit for bit against the oracle's stems of those voices
    ids = np.unique(np.concatenate([[0, 63, 64, 127, n - 1], np.random.default_rng(1).choice(n, 59, replace=False)])).astype(np.int32)
    sub, sub_g = bank.take(ids), g.copy()
    cpuref.render(sub, sub_g, tables, 11 * F, 0)                        # (a clean bank: voices do not depend on their neighbours)
    want = cpuref.render(sub, sub_g, tables, F, 0, want_stems=True)["stems"]
    probe = torch.zeros(F * len(ids) * 2, device="cuda")
    db.set_probe(ids, probe.data_ptr())
    db.render_mix(F, out.data_ptr(), 2, 0, 0)
    torch.cuda.synchronize()
    assert db.last_kernel() == 3
    seen = probe.cpu().numpy().reshape(F, len(ids), 2)
    db.set_probe([], 0)
    got = bank.copy()
    db.download(got)
    assert db.list_violations() == 0
    db.close()
    assert gio.bits_equal(seen, want), "a probed voice-sample of the headline block differs from the oracle's stem"
    assert not got.rw_equal(ref_bank), got.rw_equal(ref_bank)
    assert rel_rms(out.cpu().numpy(), ref_mix) <= 1e-5


def test_bank_at_the_size_ceiling(dev):
    """SKRED_MAX_VOICES (2^24; include/skred_amd.h): a bank of exactly the documented ceiling renders, and -- voices being
    independent -- every one of its 16 copies of a 2^20-voice bank ends in the state that bank ends in alone, bit for bit, with
    16 x its pre-master sum (1e-5).  64 frames from the recipe's first frame: steady kernel, envelope kernel beside it, and the
    motion list at 2^24 voices (index widths: the class of round 2's overflow)."""
    import torch
    n1, copies, F = 1 << 20, 16, 64
    one, tables, g = banks.bank_c2(n1)
    part1 = torch.zeros(F, 2, device="cuda")
    db1 = dev.DeviceBank(n1)
    db1.set_tables(tables)
    db1.upload(one)
    db1.set_globals(g)
    db1.render(F, part1.data_ptr())
    torch.cuda.synchronize()
    end1 = one.copy()
    db1.download(end1)
    db1.close()
    db = dev.DeviceBank(n1 * copies)
    db.set_tables(tables)
    for k in range(copies):
        db.upload(one, 0, k * n1, n1)
    db.set_globals(g)
    part = torch.zeros(F, 2, device="cuda")
    db.render(F, part.data_ptr())
    torch.cuda.synchronize()
    assert db.last_kernel() == 3
    for k in (0, 7, copies - 1):
        got = one.copy()
        db.download(got, k * n1, 0, n1)
        assert not got.rw_equal(end1), (k, got.rw_equal(end1))
    assert db.list_violations() == 0
    db.close()
    assert rel_rms(part.cpu().numpy(), copies * part1.cpu().numpy().astype(np.float64)) <= 1e-5


def test_two_operator_fm_full_size(dev):
    """A 2^20-voice two-operator FM bank of C2 voices (the bank tools/measure_banks.py times) on the kernel that keeps
    carrier and modulator in one lane, from the recipe's first frame (envelopes in motion: pairs are handed to the envelope
    kernel as pairs), two launches of 64 frames (the modulator's last sample crosses the launch boundary in voice_sample)."""
    n = 1 << 20
    bank, tables, g = banks.bank_c2(n)
    car = np.arange(0, n, 2)
    bank["voice_freq_mod_osc"][car] = car + 1
    bank["voice_freq_mod_depth"][car] = 0.2
    bank["voice_disconnect"][car[::2] + 1] = 1
    ref_bank, ref_g = bank.copy(), g.copy()
    refs = []
    for _ in range(2):
        r = cpuref.render(ref_bank, ref_g, tables, 64, 0)
        refs.append(cpuref.master(ref_g, r["sum64"].astype(np.float32)))
    db = dev.DeviceBank(n)
    db.set_tables(tables)
    db.upload(bank)
    db.set_globals(g)
    mixes = [db.render_host(64)[0] for _ in range(2)]
    assert db.last_kernel() == 3
    got = bank.copy()
    db.download(got)
    db.close()
    assert not got.rw_equal(ref_bank), got.rw_equal(ref_bank)
    assert rel_rms(np.concatenate(mixes), np.concatenate(refs)) <= 1e-5


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_feature_mix_vs_oracle(dev, seed):
    """Fuzz: banks whose voices switch features on and off at random (tables, one-shot / loop / reverse,
    S&H, crush, noise, filter modes, envelopes in every stage, smoother on/off, mutes, zero amps, FM / AM /
    pan / CZ modulators inside 64-voice groups) -- generic and modulated kernels against the oracle,
    per-voice stems and state bit-exact."""
    rng = np.random.default_rng(seed)
    n = 64 * int(rng.integers(1, 5))
    gold = gio.load("c4_pcm_oneshot")            # supplies sine + five one-shot tables in one pool
    tables = gold.tables
    seg = gold.segments[0]
    # table catalogue from the fixture: (offset, size) pairs actually used there
    cat = sorted({(int(o), int(s)) for o, s in zip(seg.bank_in["voice_table_offset"], seg.bank_in["voice_table_size"]) if s > 0})
    b = VoiceBank(n)
    pick = rng.integers(0, len(cat), n)
    off = np.array([cat[i][0] for i in pick]); size = np.array([cat[i][1] for i in pick])
    b["voice_table_offset"], b["voice_table_size"] = off, size.astype(np.int32)
    b["voice_one_shot"] = (rng.random(n) < 0.3).astype(np.int32)
    b["voice_loop_enabled"] = (rng.random(n) < 0.4).astype(np.int32)
    ls = (rng.random(n) * 0.4 * size).astype(np.int32)
    le = (ls + 2 + rng.random(n) * 0.5 * size).astype(np.int32)
    b["voice_loop_start_f"], b["voice_loop_end_f"] = ls.astype(np.float32), np.minimum(le, size).astype(np.float32)
    b["voice_loop_valid"] = (b["voice_loop_end_f"] > b["voice_loop_start_f"]).astype(np.int32)
    b["voice_direction"] = (rng.random(n) < 0.2).astype(np.int32)
    b["voice_phase"] = (rng.random(n) * (size - 1)).astype(np.float32)
    b["voice_phase_inc"] = (rng.random(n) ** 3 * 40.0).astype(np.float32)
    b["voice_amp"] = np.where(rng.random(n) < 0.15, 0.0, rng.random(n) * 2).astype(np.float32)
    pan = (rng.random(n) * 2 - 1).astype(np.float32)
    b["voice_pan_left"], b["voice_pan_right"] = banks.pan_gains(pan)
    b["voice_disconnect"] = (rng.random(n) < 0.15).astype(np.int32)
    b["voice_wave_table_index"] = np.where(rng.random(n) < 0.08, 6, 200).astype(np.int32)
    b["voice_sample_hold_max"] = np.where(rng.random(n) < 0.15, rng.integers(1, 9, n), 0).astype(np.int32)
    b["voice_quantize"] = np.where(rng.random(n) < 0.15, rng.integers(1, 12, n), 0).astype(np.int32)
    b["voice_smoother_enable"] = (rng.random(n) < 0.8).astype(np.int32)
    b["voice_smoother_smoothing"] = (0.001 + rng.random(n) * 0.5).astype(np.float32)
    mode = np.where(rng.random(n) < 0.5, rng.integers(1, 6, n), 0).astype(np.int32)
    co = banks.biquad_coeffs(np.maximum(mode, 1), 100 + rng.random(n) * 8000, 0.5 + rng.random(n) * 3, 44100)
    for k, v in co.items():
        b["voice_filter"][k] = v
    b["voice_filter_mode"] = mode
    g = gold.segments[0].g_in.copy()
    g.synth_sample_count = 50000
    use_env = rng.random(n) < 0.6
    e = b["voice_amp_envelope"]
    e["attack_time"] = (rng.random(n) * 300).astype(np.float32)
    e["decay_time"] = (rng.random(n) * 300).astype(np.float32)
    e["sustain_level"] = rng.random(n).astype(np.float32)
    e["release_time"] = (rng.random(n) * 400).astype(np.float32)
    e["sample_start"] = (50000 - rng.integers(0, 500, n)).astype(np.uint64)
    e["sample_release"] = np.where(rng.random(n) < 0.4, 50000 - rng.integers(0, 200, n), 0).astype(np.uint64)
    e["is_active"] = (rng.random(n) < 0.9).astype(np.int32)
    e["velocity"] = (0.2 + rng.random(n)).astype(np.float32)
    b["voice_use_amp_envelope"] = use_env.astype(np.int32)
    modulated = seed % 2 == 0
    if modulated:
        base = (np.arange(n) // 64) * 64
        for key, depth, p in (("voice_freq_mod_osc", "voice_freq_mod_depth", 0.2), ("voice_amp_mod_osc", "voice_amp_mod_depth", 0.2),
                              ("voice_pan_mod_osc", "voice_pan_mod_depth", 0.15), ("voice_cz_mod_osc", "voice_cz_mod_depth", 0.3)):
            b[key] = np.where(rng.random(n) < p, base + rng.integers(0, 64, n), -1).astype(np.int32)
            b[depth] = (rng.random(n) * 2).astype(np.float32)
        b["voice_freq_scale"] = (0.5 + rng.random(n)).astype(np.float32)
        b["voice_cz_mode"] = np.where(rng.random(n) < 0.3, rng.integers(1, 8, n), 0).astype(np.int32)
        b["voice_cz_distortion"] = rng.random(n).astype(np.float32)
    frames = 400
    ref_bank, ref_g = b.copy(), g.copy()
    r = cpuref.render(ref_bank, ref_g, tables, frames, 0, want_stems=True)
    ref_mix = cpuref.master(ref_g, r["sum64"].astype(np.float32))
    # unmodulated banks: the one-voice kernel's extended instantiation (it writes stems too), then the generic kernel
    for force in ((False,) if modulated else (False, True)):
        db = dev.DeviceBank(n)
        db.set_tables(tables)
        db.upload(b)
        db.set_globals(g)
        db.force_generic(force)
        mix, stems = db.render_host(frames, 2, 0, want_stems=True)
        assert db.last_kernel() == (2 if modulated else 0 if force else 1)
        got = b.copy()
        db.download(got)
        gl = db.get_globals()
        db.close()
        finite = np.isfinite(r["stems"])
        assert gio.bits_equal(np.where(finite, stems, 0), np.where(finite, r["stems"], 0)), "stems differ"
        assert (np.isfinite(stems) == finite).all()
        assert not got.rw_equal(ref_bank), got.rw_equal(ref_bank)
        assert gl.noise_rng == ref_g.noise_rng
        if np.isfinite(ref_mix).all():
            assert rel_rms(mix, ref_mix) <= 1e-5


def test_non_finite_phase_increments(dev):
    """osc_next's !isfinite() branch (synth.c:228-232): phase resets to 0, a one-shot voice finishes, the frame
    outputs table-independent 0 -- NaN / +-inf increments and a NaN phase, on cyclic and one-shot voices."""
    gold = gio.load("c4_pcm_oneshot")
    seg = gold.segments[0]
    b = seg.bank_in.copy()
    b["voice_amp"] = 1.0
    b["voice_finished"] = 0
    b["voice_phase_inc"][0] = np.inf
    b["voice_phase_inc"][1] = -np.inf
    b["voice_phase_inc"][2] = np.nan
    b["voice_phase"][3] = np.nan
    b["voice_phase_inc"][20] = np.inf          # voices 20+ sit on the cyclic sine table (voice_reset default)
    b["voice_phase_inc"][21] = np.nan
    b["voice_phase_inc"][22] = 3.0e38          # finite but overflows to inf after a few adds
    ref_bank, ref_g = b.copy(), seg.g_in.copy()
    r = cpuref.render(ref_bank, ref_g, gold.tables, 200, 0, want_stems=True)
    db = dev.DeviceBank(b.n)
    db.set_tables(gold.tables)
    db.upload(b)
    db.set_globals(seg.g_in)
    mix, stems = db.render_host(200, 2, 0, want_stems=True)
    got = b.copy()
    db.download(got)
    db.close()
    fin = np.isfinite(r["stems"])
    assert (np.isfinite(stems) == fin).all()
    assert gio.bits_equal(np.where(fin, stems, 0), np.where(fin, r["stems"], 0))
    for k in ("voice_finished", "voice_sample_hold_count"):
        assert (got[k] == ref_bank[k]).all(), k
    ph_g, ph_r = got["voice_phase"], ref_bank["voice_phase"]
    assert ((ph_g.view(np.uint32) == ph_r.view(np.uint32)) | (np.isnan(ph_g) & np.isnan(ph_r))).all()


@pytest.mark.parametrize("all_in_attack", [False, True])
def test_four_million_voices(dev, all_in_attack):
    """Maximum-size edge: 2^22 voices (0.8 GB of device state), 16 frames, against the oracle; as the recipe has them, and
    with every note starting on the block's first frame (every voice is handed to the envelope kernel: 2^22 list entries,
    8192 passes)."""
    n, frames = 1 << 22, 16
    bank, tables, g = banks.bank_c2(n)
    if all_in_attack:
        e = bank["voice_amp_envelope"]
        e["sample_start"][:] = g.synth_sample_count; e["sample_release"][:] = 0; e["is_active"][:] = 1
    ref_bank, ref_g = bank.copy(), g.copy()
    r = cpuref.render(ref_bank, ref_g, tables, frames, 0)     # parity build (-ffp-contract=off), not the timing build
    ref_mix = cpuref.master(ref_g, r["sum64"].astype(np.float32))
    db = dev.DeviceBank(n)
    db.set_tables(tables)
    db.upload(bank)
    db.set_globals(g)
    mix, _ = db.render_host(frames)
    got = bank.copy()
    db.download(got)
    db.close()
    assert not got.rw_equal(ref_bank), got.rw_equal(ref_bank)
    assert rel_rms(mix, ref_mix) <= 1e-5
