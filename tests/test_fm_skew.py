"""Skewed blocks of frequency-modulated wavefronts (SKRED_OPT_FM_SKEW, skred_render_fast.hip: SK_FAST_SKEW_STEP).

The shipped patches write frequency modulation as `v0 ... F3,1` with the modulator ABOVE its carriers (3.sk, 0.sk): the carrier
reads voice_sample[m] of the previous frame (synth.c:548-555).  On the one-voice-per-lane kernel the modulator lanes of such a
wavefront run one 8-frame block ahead of their carriers and hand their samples over through an LDS ring, instead of one
ds_bpermute exchange per frame.  Checked here, bit for bit:

  * state after every launch and per-voice samples of every frame (probe rows written from inside the blocks) against the oracle,
    with ragged block lengths (the skew must be taken back before the frames behind the last whole block), launches too short
    for it, envelopes that only become steady in the middle of a launch (the skew starts there), wild modulation depths (a
    step whose tameness vote fails takes the general frames with the ring's samples), modulators switched off between launches,
    linear lookup;
  * chains (7.sk: v2 -> v1 -> v0, the head two blocks ahead), amplitude and pan modulation from other voices and from the voice
    itself, sample & hold and bit-crush (the RICH form: fast_frame with the ring's samples handed in);
  * wavefronts that must NOT be skewed (an audible modulator, a source read by lanes of different leads) next to ones that are;
  * the mix of the skewed form equals the mix of the per-frame exchange to the last bit (same products, same tile sums).
"""
import numpy as np
import pytest

import golden_io as gio
from oracle import cpuref
from skred_amd import banks

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from skred_amd import device
    assert device.load().skred_amd_device_count() > 0, "no GPU visible"
    return device


def rel_rms(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / max(np.sqrt(np.mean(b ** 2)), 1e-30))


def fm_bank(recipe, n, wild_every=0, hold=False):
    """3.sk's shape over a recipe bank: voices 4k, 4k+1, 4k+2 are carriers of the muted voice 4k+3."""
    bank, tables, g = banks.RECIPES[recipe](n)
    v = np.arange(n)
    car = v[v % 4 != 3]
    mod = (car // 4) * 4 + 3
    bank["voice_freq_mod_osc"][car] = mod
    bank["voice_freq_mod_depth"][car] = (np.float32(0.02) * (1 + (car % 13))).astype(np.float32)      # tame
    bank["voice_freq_scale"][car] = (np.float32(0.5) + np.float32(0.01) * (car % 50)).astype(np.float32)
    bank["voice_disconnect"][v[v % 4 == 3]] = 1                                                      # `m1`
    if wild_every:                         # whole 64-voice groups with deep modulation: increments negative / beyond half a loop
        w = car[(car // 64) % wild_every == 0]
        bank["voice_freq_mod_depth"][w] = (np.float32(1.5) * (1 + (w % 7))).astype(np.float32)
    if hold:                               # sample & hold / bit-crush on some carriers and some modulators of every third wavefront
        h = (v // 64) % 3 == 1             # (such a wavefront keeps the per-frame exchange: the skewed steps carry no feature tests)
        bank["voice_sample_hold_max"][v[h & (v % 24 == 1)]] = 5
        bank["voice_sample_hold_max"][v[h & (v % 40 == 3)]] = 3
        bank["voice_quantize"][v[h & (v % 36 == 2)]] = 6
    return bank, tables, g


def run(dev, bank, tables, g, interp, segments, skew, probe_ids=None):
    import torch
    db = dev.DeviceBank(bank.n)
    db.set_tables(tables)
    host = bank.copy()
    db.upload(host)
    db.set_globals(g)
    db.set_fm_skew(skew)
    fmax = max(f for f, _ in segments)
    buf = None
    if probe_ids is not None:
        buf = torch.zeros(fmax * len(probe_ids) * 2, device="cuda")
        db.set_probe(probe_ids, buf.data_ptr())
    mixes, probes, kernels = [], [], []
    for frames, event in segments:
        if event is not None:
            db.download(host)
            event(host, db.get_globals().synth_sample_count)
            db.upload(host)
        if buf is not None:
            buf.zero_()
        # (other kernels' leftovers in LDS between our launches: the sample ring must not depend on what the previous launch of
        # the same bank left in it -- an uninitialised row once happened to hold exactly the right numbers that way)
        torch.sort(torch.rand(1 << 22, device="cuda"))
        out = torch.zeros(frames, 2, device="cuda")
        db.render_mix(frames, out.data_ptr(), 2, 0, interp)
        torch.cuda.synchronize()
        mixes.append(out.cpu().numpy())
        kernels.append(db.last_kernel())
        if buf is not None:
            probes.append(buf[:frames * len(probe_ids) * 2].cpu().numpy().reshape(frames, len(probe_ids), 2).copy())
    db.download(host)
    if buf is not None:
        db.set_probe([], 0)
    db.close()
    return np.concatenate(mixes), host, probes, kernels


def oracle(bank, tables, g, interp, segments, probe_ids=None):
    host, gl = bank.copy(), g.copy()
    mixes, stems = [], []
    for frames, event in segments:
        if event is not None:
            event(host, gl.synth_sample_count)
        r = cpuref.render(host, gl, tables, frames, interp, want_stems=probe_ids is not None)
        mixes.append(cpuref.master(gl, r["sum64"].astype(np.float32)))
        if probe_ids is not None:
            stems.append(r["stems"][:, probe_ids, :].copy())
    return np.concatenate(mixes), host, stems


def _release_some(host, now):
    host["voice_amp_envelope"]["sample_release"][2::5] = now          # amp_envelope_release, synth.c:391-395


def _kill_some_modulators(host, now):
    host["voice_amp"][3::28] = 0.0                                      # skipped from now on: voice_sample = 0 (synth.c:537-541)


def _probe_ids(n):
    rng = np.random.default_rng(5)
    return np.unique(np.concatenate([[0, 1, 2, 4, 62, 64, 65, n - 4, n - 3, n - 2], rng.choice(n, 60, replace=False)]))[:64].astype(np.int32)


SEGS = [(300, None), (257, _release_some), (31, None), (40, _kill_some_modulators), (520, None), (64, None), (36, None)]


@pytest.mark.parametrize("recipe,interp,wild,hold", [("c2", 0, 0, False), ("c2", 0, 3, True), ("c2", 1, 4, False), ("c1", 0, 2, True)])
def test_skewed_blocks_state_probes_and_mix(dev, recipe, interp, wild, hold):
    n = 4096
    bank, tables, g = fm_bank(recipe, n, wild, hold)
    bank["voice_amp"][8:12] = 0.0                      # a copy that is off altogether, a copy whose modulator alone is off
    bank["voice_amp"][19] = 0.0
    ids = _probe_ids(n)
    ref_mix, ref_state, ref_stems = oracle(bank, tables, g, interp, SEGS, ids)
    res = {}
    for skew in (1, 0):
        mix, state, probes, kernels = run(dev, bank, tables, g, interp, SEGS, skew, ids)
        assert kernels == [1] * len(SEGS), kernels
        bad = state.rw_equal(ref_state)
        assert not bad, (skew, bad)
        for k, (got, want) in enumerate(zip(probes, ref_stems)):
            d = np.argwhere(got.view(np.uint32) != want.view(np.uint32))
            assert len(d) == 0, f"skew={skew} launch {k}: {len(d)} probed values differ; first (frame, probe, ch) {d[0]}, voice {ids[d[0][1]]}: {got[tuple(d[0])]} vs {want[tuple(d[0])]}"
        assert rel_rms(mix, ref_mix) <= 1e-5
        res[skew] = mix
    assert gio.bits_equal(res[1], res[0]), "skewed blocks and per-frame exchange: same products, same sums"


def test_waves_that_must_not_be_skewed(dev):
    """An audible modulator (skewed all the same: its (L, R) are formed a step late from its own ring column -- probed here, every
    frame), a modulator read by lanes of different leads (by three carriers AND by another modulator: it cannot be one block ahead
    of all of them: that wavefront keeps the exchange) and a carrier reading a modulator 40 lanes away, in three wavefronts of a
    bank whose other wavefronts are skewed: every voice's state and samples against the oracle."""
    n = 1024
    bank, tables, g = fm_bank("c2", n)
    bank["voice_disconnect"][64 + 7] = 0               # wave 1: one modulator is heard
    bank["voice_freq_mod_osc"][128 + 11] = 128 + 15    # wave 2: modulator 11 takes modulator 15's sample
    bank["voice_freq_mod_depth"][128 + 11] = np.float32(0.3)
    bank["voice_freq_scale"][128 + 11] = np.float32(1.0)
    bank["voice_freq_mod_osc"][192 + 1] = 192 + 43     # wave 3: still skewed, a far modulator
    ids = np.unique(np.concatenate([np.arange(64, 80), np.arange(128 + 8, 128 + 16), [192, 193, 194, 195, 192 + 43], np.arange(0, 8)])).astype(np.int32)
    segs = [(512, None), (100, None)]
    ref_mix, ref_state, ref_stems = oracle(bank, tables, g, 0, segs, ids)
    mix, state, probes, kernels = run(dev, bank, tables, g, 0, segs, 1, ids)
    assert kernels == [1, 1]
    assert not state.rw_equal(ref_state), state.rw_equal(ref_state)
    for got, want in zip(probes, ref_stems):
        assert gio.bits_equal(got, want)
    assert rel_rms(mix, ref_mix) <= 1e-5


def mod_bank(recipe, n, wild_every=0):
    """The shipped patches' richer routings over a recipe bank, in copies of four voices.  Even copies, 7.sk's shape: v0 F1 P3,
    v1 F2 m1, v2 m1, v3 m1 -- a chain (v2 two blocks ahead, v1 and v3 one).  Odd copies, 1.sk's / 37.sk's shape: v0 F3 A2 P1
    (frequency, amplitude and pan modulation from three muted voices), some of them with sample & hold and bit-crush, some
    modulating their own amplitude or pan (synth.c:584-587,597-602: the voice's own sample of THIS frame)."""
    bank, tables, g = banks.RECIPES[recipe](n)
    v = np.arange(n)
    c = v[v % 4 == 0]
    even, odd = c[(c // 4) % 2 == 0], c[(c // 4) % 2 == 1]
    bank["voice_disconnect"][v[v % 4 != 0]] = 1
    bank["voice_freq_scale"][v] = (np.float32(0.5) + np.float32(0.01) * (v % 50)).astype(np.float32)
    bank["voice_freq_mod_osc"][even] = even + 1
    bank["voice_freq_mod_depth"][even] = (np.float32(0.03) * (1 + (even % 11))).astype(np.float32)
    bank["voice_pan_mod_osc"][even] = even + 3
    bank["voice_pan_mod_depth"][even] = np.float32(0.7)
    bank["voice_freq_mod_osc"][even + 1] = even + 2
    bank["voice_freq_mod_depth"][even + 1] = np.float32(0.2)
    bank["voice_freq_mod_osc"][odd] = odd + 3
    bank["voice_freq_mod_depth"][odd] = (np.float32(0.02) * (1 + (odd % 9))).astype(np.float32)
    bank["voice_amp_mod_osc"][odd] = odd + 2
    bank["voice_amp_mod_depth"][odd] = np.float32(1.5)
    bank["voice_pan_mod_osc"][odd] = odd + 1
    bank["voice_pan_mod_depth"][odd] = np.float32(0.9)
    bank["voice_sample_hold_max"][odd[::3]] = 5
    bank["voice_quantize"][odd[1::5]] = 5
    bank["voice_sample_hold_max"][odd[::7] + 3] = 4                    # a modulator that holds
    self_am = odd[2::9]
    bank["voice_amp_mod_osc"][self_am] = self_am
    self_pm = even[3::8]
    bank["voice_pan_mod_osc"][self_pm] = self_pm
    if wild_every:
        w = c[(c // 64) % wild_every == 0]
        bank["voice_freq_mod_depth"][w] = (np.float32(2.0) * (1 + (w % 5))).astype(np.float32)
    return bank, tables, g


@pytest.mark.parametrize("recipe,interp,wild", [("c2", 0, 0), ("c2", 1, 3), ("c1", 0, 2)])
def test_rich_skewed_blocks_chains_amp_pan_hold(dev, recipe, interp, wild):
    n = 4096
    bank, tables, g = mod_bank(recipe, n, wild)
    bank["voice_amp"][16:20] = 0.0
    bank["voice_amp"][27] = 0.0                        # one source of a 1.sk-shaped copy is off
    bank["voice_amp"][34] = 0.0                        # the head of a chain is off
    ids = np.unique(np.concatenate([np.arange(0, n, 4)[:: n // 4 // 56], [0, 4, 8, 12, 24, 32, n - 4, n - 8]])).astype(np.int32)[:64]
    ref_mix, ref_state, ref_stems = oracle(bank, tables, g, interp, SEGS, ids)
    res = {}
    for skew in (1, 0):
        mix, state, probes, kernels = run(dev, bank, tables, g, interp, SEGS, skew, ids)
        assert kernels == [1] * len(SEGS), kernels
        bad = state.rw_equal(ref_state)
        assert not bad, (skew, bad)
        for k, (got, want) in enumerate(zip(probes, ref_stems)):
            d = np.argwhere(got.view(np.uint32) != want.view(np.uint32))
            assert len(d) == 0, f"skew={skew} launch {k}: {len(d)} probed values differ; first (frame, probe, ch) {d[0]}, voice {ids[d[0][1]]}: {got[tuple(d[0])]} vs {want[tuple(d[0])]}"
        assert rel_rms(mix, ref_mix) <= 1e-5
        res[skew] = mix
    assert gio.bits_equal(res[1], res[0])


def test_audible_sources(dev):
    """37.sk's shape: a modulator WITHOUT `m1` -- heard, and read by a carrier.  Such a lane runs a block ahead like any source;
    what it adds to the mix of a frame is formed a step later from the sample it left in its own ring column (same product).
    Every copy of a mod_bank gets one audible source (some pan-modulated themselves: those wavefronts keep the exchange); probes
    on the audible sources and their carriers, ragged lengths, state, mix."""
    n = 2048
    bank, tables, g = mod_bank("c2", n, 4)
    v = np.arange(n)
    heard = v[(v % 4 == 3) & ((v // 4) % 2 == 0)]      # v3 of the 7.sk-shaped copies: the pan source of v0
    bank["voice_disconnect"][heard] = 0
    heard2 = v[(v % 4 == 2) & ((v // 4) % 2 == 1)]     # v2 of the 1.sk-shaped copies: the amplitude source of v0
    bank["voice_disconnect"][heard2[::2]] = 0
    pm = heard[(heard // 64) % 5 == 4]                 # in every fifth wavefront the audible source is pan-modulated itself
    bank["voice_pan_mod_osc"][pm] = pm
    bank["voice_pan_mod_depth"][pm] = np.float32(0.5)
    ids = np.unique(np.concatenate([heard[::9], heard2[::14], np.arange(0, n, 4)[::23], [3, 7, 259, 263]])).astype(np.int32)[:64]
    ref_mix, ref_state, ref_stems = oracle(bank, tables, g, 0, SEGS, ids)
    res = {}
    for skew in (1, 0):
        mix, state, probes, kernels = run(dev, bank, tables, g, 0, SEGS, skew, ids)
        assert kernels == [1] * len(SEGS), kernels
        assert not state.rw_equal(ref_state), (skew, state.rw_equal(ref_state))
        for k, (got, want) in enumerate(zip(probes, ref_stems)):
            d = np.argwhere(got.view(np.uint32) != want.view(np.uint32))
            assert len(d) == 0, f"skew={skew} launch {k}: {len(d)} probed values differ; first (frame, probe, ch) {d[0]}, voice {ids[d[0][1]]}: {got[tuple(d[0])]} vs {want[tuple(d[0])]}"
        assert rel_rms(mix, ref_mix) <= 1e-5
        res[skew] = mix
    assert gio.bits_equal(res[1], res[0])


@pytest.mark.parametrize("patch", ["3sk", "1sk", "7sk", "37sk"])
def test_patch_bank_skewed_equals_exchange(dev, patch):
    """banks.bank_patch(...) at 2^16 voices, 512-frame blocks (tools/measure_banks `patches` at a sixteenth of its size): skewed
    blocks against the per-frame exchange and the oracle."""
    n = 1 << 16
    bank, tables, g = banks.bank_patch(patch, n)
    segs = [(512, None), (512, None)]
    ref_mix, ref_state, _ = oracle(bank, tables, g, 0, segs)
    res = {}
    for skew in (1, 0):
        mix, state, _, kernels = run(dev, bank, tables, g, 0, segs, skew)
        assert kernels == [1, 1]
        assert not state.rw_equal(ref_state), (skew, state.rw_equal(ref_state))
        assert rel_rms(mix, ref_mix) <= 1e-5
        res[skew] = mix
    assert gio.bits_equal(res[1], res[0])


def _copies(bank, copy_ids, K):
    """The K-voice copies `copy_ids` of a tiled patch bank as a bank of their own (modulator indices moved along): what the oracle
    renders beside the device for the probed voices -- a copy never reads a voice outside itself."""
    idx = (np.asarray(copy_ids)[:, None] * K + np.arange(K)[None, :]).reshape(-1)
    sub = bank.take(idx)
    shift = np.repeat((np.arange(len(copy_ids)) - np.asarray(copy_ids)) * K, K).astype(np.int32)
    for f in ("voice_freq_mod_osc", "voice_amp_mod_osc", "voice_pan_mod_osc", "voice_cz_mod_osc"):
        m = sub.a[f]
        sub.a[f] = np.where(m >= 0, m + shift, m).astype(np.int32)
    return sub, idx


@pytest.mark.parametrize("patch,K", [("3sk", 4), ("1sk", 4), ("7sk", 4), ("37sk", 16)])
def test_full_size_patch_bank_probed_inside_the_skewed_blocks(dev, patch, K):
    """The timed workload itself (`shipped_patches`: 2^20 voices, 512-frame blocks) with 64 probed voices in 16 copies spread over
    the whole bank: (L, R) of every frame, written from inside the skewed steps, bit for bit against the oracle rendering those
    copies as a bank of their own."""
    import torch
    n, F = 1 << 20, 512
    bank, tables, g = banks.bank_patch(patch, n)
    assert bank.n % K == 0 and not np.any(np.asarray(bank["voice_freq_mod_osc"])[K:2 * K] >= 2 * K)
    rng = np.random.default_rng(11)
    copy_ids = np.unique(np.concatenate([[0, n // K - 1], rng.choice(n // K, 14, replace=False)]))
    sub, idx = _copies(bank, copy_ids, K)
    used = np.where(np.asarray(sub["voice_amp"]) != 0)[0][:64]
    ids = idx[used].astype(np.int32)
    db = dev.DeviceBank(n)
    db.set_tables(tables)
    db.upload(bank)
    db.set_globals(g)
    buf = torch.zeros(F * len(ids) * 2, device="cuda")
    db.set_probe(ids, buf.data_ptr())
    sub_g = g.copy()
    out = torch.zeros(F, 2, device="cuda")
    for k in range(3):
        buf.zero_()
        db.render_mix(F, out.data_ptr(), 2, 0, 0)
        torch.cuda.synchronize()
        got = buf.cpu().numpy().reshape(F, len(ids), 2)
        want = cpuref.render(sub, sub_g, tables, F, 0, want_stems=True)["stems"][:, used, :]
        d = np.argwhere(got.view(np.uint32) != want.view(np.uint32))
        assert len(d) == 0, f"{patch} block {k}: {len(d)} probed values differ; first (frame, probe, ch) {d[0]}, voice {ids[d[0][1]]}: {got[tuple(d[0])]} vs {want[tuple(d[0])]}"
    db.set_probe([], 0)
    db.close()


def below_bank(recipe, n):
    """Modulators BELOW their carriers (same-frame dependencies: the modulated kernel), in copies of eight voices.
    Copies 0 mod 4, 18.sk's shape: v0 F1 (previous frame, from above), v1, v2 plain, v6 F0 (SAME frame, from below) -- one level:
    the frame-lag form.  Copies 1 mod 4: v5 A2 P1 (amplitude and pan from below), v7 is a noise voice modulated in frequency... no:
    noise ignores it; v7 A3 (a level-1 noise voice: it takes the draw of ITS frame).  Copies 2 mod 4: a two-level chain v1 F0, v2 F1
    (that wavefront keeps the level loop).  Copies 3 mod 4: v4 F2 from below, and v2 F4 from above -- a previous-frame edge ACROSS
    levels: not lag-able either."""
    bank, tables, g = banks.RECIPES[recipe](n)
    v = np.arange(n)
    base = v[v % 8 == 0]
    kind = (base // 8) % 4
    bank["voice_freq_scale"][v] = (np.float32(0.5) + np.float32(0.01) * (v % 40)).astype(np.float32)
    def fm(dst, src, depth):
        bank["voice_freq_mod_osc"][dst] = src
        bank["voice_freq_mod_depth"][dst] = np.float32(depth)
    b0 = base[kind == 0]
    fm(b0, b0 + 1, 0.1); fm(b0 + 6, b0, 0.4)
    bank["voice_disconnect"][b0 + 1] = 1
    b1 = base[kind == 1]
    bank["voice_amp_mod_osc"][b1 + 5] = b1 + 2; bank["voice_amp_mod_depth"][b1 + 5] = np.float32(1.5)
    bank["voice_pan_mod_osc"][b1 + 5] = b1 + 1; bank["voice_pan_mod_depth"][b1 + 5] = np.float32(0.8)
    bank["voice_wave_table_index"][b1 + 7] = 6                       # WAVE_TABLE_NOISE_ALT (synth.c:543)
    bank["voice_amp_mod_osc"][b1 + 7] = b1 + 3; bank["voice_amp_mod_depth"][b1 + 7] = np.float32(0.9)
    b2 = base[kind == 2]
    fm(b2 + 1, b2, 0.3); fm(b2 + 2, b2 + 1, 0.3)
    b3 = base[kind == 3]
    fm(b3 + 4, b3 + 2, 0.2); fm(b3 + 2, b3 + 4, 0.2)
    return bank, tables, g


@pytest.mark.parametrize("recipe,interp", [("c2", 0), ("c1", 1)])
def test_frame_lag_form_of_the_modulated_kernel(dev, recipe, interp):
    """sk_render_mod_kernel with one level of same-frame dependencies: the dependent lanes one frame behind (SKRED_OPT_FM_SKEW 1)
    against the level loop (0) and the oracle -- state bit for bit after every launch, mix to the last bit between the two forms;
    launches of 1, 2, 65 and 300 frames, envelopes in motion, a release, a source switched off, a level-1 noise voice."""
    n = 2048
    bank, tables, g = below_bank(recipe, n)
    bank["voice_amp"][0] = 0.0                          # a level-0 source that is off: its level-1 reader takes exact zeros
    bank["voice_amp"][38] = 0.0

    def kill(host, now):
        host["voice_amp"][64::96] = 0.0

    segs = [(1, None), (2, None), (65, None), (300, _release_some), (130, kill), (64, None)]
    ref_mix, ref_state, _ = oracle(bank, tables, g, interp, segs)
    res = {}
    for lag in (1, 0):
        mix, state, _, kernels = run(dev, bank, tables, g, interp, segs, lag)
        assert kernels == [2] * len(segs), kernels
        bad = state.rw_equal(ref_state)
        assert not bad, (lag, bad)
        assert rel_rms(mix, ref_mix) <= 1e-5
        res[lag] = mix
    assert gio.bits_equal(res[1], res[0])
