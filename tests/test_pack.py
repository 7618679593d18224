"""Packed lanes (SKRED_OPT_PACK): sparse banks on the one-voice kernel with a wavefront holding the voices that CAN sound of
several aligned 64-voice groups.

The reference skips a voice whose amp is 0 (synth.c:537-542: voice_sample = 0, nothing else touched); a bank in use is mostly
such voices (every shipped patch uses 3 to 6 of the reference's 64).  The packed form gives lanes only to the voices that can
sound and to the modulators they name.  Checked here:

  * per-voice STATE of the whole bank bit for bit against the oracle after several blocks, packed and not packed, with the
    mix within 1e-5 (the mix's summation order is the only thing that changes);
  * per-frame samples of probed voices (skred_bank_set_probe: the full stem buffer switches packing off) bit for bit against
    the oracle's stems, modulated voices included;
  * modulators that cannot sound themselves (amp 0: they keep a lane and read as 0 after their first frame), self-modulation,
    sample & hold / crush / reverse / one-shots that finish inside a block;
  * control actions that change who can sound: a voice switched on grows the slots, a voice switched off loses its lane and
    gets the voice_sample = 0 the reference's skip rule gives it, DIRTY_SAMPLE onto a skipped voice.
"""
import numpy as np
import pytest

import golden_io as gio
from oracle import cpuref
from skred_amd import banks
from skred_amd.bank import VoiceBank

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    from skred_amd import device
    assert device.load().skred_amd_device_count() > 0, "no GPU visible"
    return device


def rms(x):
    return float(np.sqrt(np.mean(np.asarray(x, np.float64) ** 2)))


def rel_rms(a, b):
    return rms(np.asarray(a, np.float64) - np.asarray(b, np.float64)) / max(rms(b), 1e-30)


def sparse_bank(seed, n, live_per_group, with_env=True):
    """A clean (fast-kernel) bank: in every 64-voice group `live_per_group` voices at random places can sound; FM from a voice above
    (sometimes one that cannot sound itself), AM / pan modulation from above or from the voice itself, sample & hold, crush,
    reverse, one-shots near their end, smoother off, mutes, filters on some, envelopes in every stage on some."""
    rng = np.random.default_rng(seed)
    gold = gio.load("c4_pcm_oneshot")
    tables = gold.tables
    seg = gold.segments[0]
    cat = sorted({(int(o), int(s)) for o, s in zip(seg.bank_in["voice_table_offset"], seg.bank_in["voice_table_size"]) if s > 0})
    b = VoiceBank(n)
    pick = rng.integers(0, len(cat), n)
    off = np.array([cat[i][0] for i in pick]); size = np.array([cat[i][1] for i in pick])
    b["voice_table_offset"], b["voice_table_size"] = off, size.astype(np.int32)
    b["voice_one_shot"] = (rng.random(n) < 0.25).astype(np.int32)
    b["voice_loop_enabled"] = (rng.random(n) < 0.5).astype(np.int32)
    ls = (rng.random(n) * 0.4 * size).astype(np.int32)
    le = (ls + 2 + rng.random(n) * 0.5 * size).astype(np.int32)
    b["voice_loop_start_f"], b["voice_loop_end_f"] = ls.astype(np.float32), np.minimum(le, size).astype(np.float32)
    b["voice_loop_valid"] = (b["voice_loop_end_f"] > b["voice_loop_start_f"]).astype(np.int32)
    b["voice_direction"] = (rng.random(n) < 0.15).astype(np.int32)
    b["voice_phase"] = (rng.random(n) * (size - 1)).astype(np.float32)
    b["voice_phase_inc"] = (rng.random(n) ** 3 * 20.0).astype(np.float32)
    amp = np.zeros(n, np.float32)
    for g0 in range(0, n, 64):
        k = int(min(live_per_group, 64)) if live_per_group >= 1 else 0
        if k:
            k = int(rng.integers(max(1, k // 2), k + 1))
            amp[g0 + rng.choice(64, k, replace=False)] = (0.1 + rng.random(k) * 2).astype(np.float32)
    b["voice_amp"] = amp
    pan = (rng.random(n) * 2 - 1).astype(np.float32)
    b["voice_pan_left"], b["voice_pan_right"] = banks.pan_gains(pan)
    b["voice_disconnect"] = (rng.random(n) < 0.15).astype(np.int32)
    b["voice_wave_table_index"] = 200
    b["voice_sample_hold_max"] = np.where(rng.random(n) < 0.15, rng.integers(1, 9, n), 0).astype(np.int32)
    b["voice_quantize"] = np.where(rng.random(n) < 0.15, rng.integers(1, 12, n), 0).astype(np.int32)
    b["voice_smoother_enable"] = (rng.random(n) < 0.8).astype(np.int32)
    b["voice_smoother_smoothing"] = (0.001 + rng.random(n) * 0.5).astype(np.float32)
    b["voice_sample"] = (rng.random(n) - 0.5).astype(np.float32)       # stale samples everywhere: the skip rule has to clear them
    mode = np.where(rng.random(n) < 0.5, rng.integers(1, 6, n), 0).astype(np.int32)
    co = banks.biquad_coeffs(np.maximum(mode, 1), 100 + rng.random(n) * 8000, 0.5 + rng.random(n) * 3, 44100)
    for k, v in co.items():
        b["voice_filter"][k] = v
    b["voice_filter_mode"] = mode
    g = seg.g_in.copy()
    g.synth_sample_count = 50000
    e = b["voice_amp_envelope"]
    e["attack_time"] = (rng.random(n) * 300).astype(np.float32)
    e["decay_time"] = (rng.random(n) * 300).astype(np.float32)
    e["sustain_level"] = rng.random(n).astype(np.float32)
    e["release_time"] = (rng.random(n) * 400).astype(np.float32)
    e["sample_start"] = (50000 - rng.integers(0, 500, n)).astype(np.uint64)
    e["sample_release"] = np.where(rng.random(n) < 0.4, 50000 - rng.integers(0, 200, n), 0).astype(np.uint64)
    e["is_active"] = (rng.random(n) < 0.9).astype(np.int32)
    e["velocity"] = (0.2 + rng.random(n)).astype(np.float32)
    b["voice_use_amp_envelope"] = (rng.random(n) < 0.5).astype(np.int32) if with_env else 0
    # modulation the one-voice kernel serves: sources ABOVE the carrier inside its 64-voice group (any voice: most cannot sound
    # themselves), amplitude / pan also from the carrier itself
    lane = np.arange(n) % 64
    above = np.arange(n) + 1 + (rng.random(n) * (63 - lane)).astype(np.int64)
    ok = lane < 63
    b["voice_freq_mod_osc"] = np.where(ok & (rng.random(n) < 0.3), above, -1).astype(np.int32)
    above2 = np.arange(n) + 1 + (rng.random(n) * (63 - lane)).astype(np.int64)
    b["voice_amp_mod_osc"] = np.where(rng.random(n) < 0.1, np.arange(n), np.where(ok & (rng.random(n) < 0.15), above2, -1)).astype(np.int32)
    above3 = np.arange(n) + 1 + (rng.random(n) * (63 - lane)).astype(np.int64)
    b["voice_pan_mod_osc"] = np.where(rng.random(n) < 0.1, np.arange(n), np.where(ok & (rng.random(n) < 0.15), above3, -1)).astype(np.int32)
    b["voice_freq_mod_depth"] = (rng.random(n) * 2).astype(np.float32)
    b["voice_amp_mod_depth"] = (rng.random(n) * 2).astype(np.float32)
    b["voice_pan_mod_depth"] = (rng.random(n) * 2).astype(np.float32)
    b["voice_freq_scale"] = (0.5 + rng.random(n)).astype(np.float32)
    # half of the modulators of voices that can sound do sound themselves
    for key in ("voice_freq_mod_osc", "voice_amp_mod_osc", "voice_pan_mod_osc"):
        src = b[key][(amp != 0) & (b[key] >= 0)]
        src = src[rng.random(len(src)) < 0.5]
        b["voice_amp"][src] = np.where(b["voice_amp"][src] == 0, 0.7, b["voice_amp"][src]).astype(np.float32)
    return b, tables, g


def expected_slots(bank, limit=32):
    """Lanes per 64-voice group the library must pick: per group the voices that can sound plus the modulators they name, the largest
    such count rounded up to a power of two (0: more than `limit`, not packed -- 32 by the default rule for one-voice banks)."""
    n = bank.n
    amp = np.asarray(bank["voice_amp"])
    need = amp != 0
    live = np.flatnonzero(need)
    for key in ("voice_freq_mod_osc", "voice_amp_mod_osc", "voice_pan_mod_osc"):
        src = np.asarray(bank[key])[live]
        need[src[src >= 0]] = True
    most = int(need.reshape(n // 64, 64).sum(1).max())
    s = 1
    while s < most: s *= 2
    return s if s <= limit else 0


def expected_slots_mod(bank, limit=32):
    """expected_slots with the CZ source of voices whose CZ mode is on, and without a frequency modulator that is the voice itself."""
    b2 = bank.copy()
    fm = np.asarray(b2["voice_freq_mod_osc"]).copy()
    fm[fm == np.arange(bank.n)] = -1
    b2["voice_freq_mod_osc"] = fm
    cz = np.where(np.asarray(b2["voice_cz_mode"]) != 0, np.asarray(b2["voice_cz_mod_osc"]), -1)
    n = bank.n
    need = np.asarray(b2["voice_amp"]) != 0
    live = np.flatnonzero(need)
    for src in (fm[live], np.asarray(b2["voice_amp_mod_osc"])[live], np.asarray(b2["voice_pan_mod_osc"])[live], cz[live]):
        need[src[src >= 0]] = True
    most = int(need.reshape(n // 64, 64).sum(1).max())
    s = 1
    while s < most: s *= 2
    return s if s <= limit else 0


def run_blocks(dev, bank, tables, g, blocks, pack, probe_ids=None, actions=None):
    """Render `blocks` (frame counts) on a fresh device bank; actions[k](db, host_bank) runs before block k.  Returns the device
    bank's downloaded state, the mixes, the probe rows, (last_kernel, last_pack) per block, violations."""
    import torch
    db = dev.DeviceBank(bank.n)
    db.set_tables(tables); db.upload(bank); db.set_globals(g); db.set_pack(pack)
    host = bank.copy()
    buf = None
    if probe_ids is not None:
        buf = torch.zeros(max(blocks) * len(probe_ids) * 2, device="cuda")
        db.set_probe(probe_ids, buf.data_ptr())
    mixes, probes, kinds = [], [], []
    for k, f in enumerate(blocks):
        if actions and k in actions:
            actions[k](db, host)
        out = torch.zeros(f, 2, device="cuda")
        db.render_mix(f, out.data_ptr(), 2, 0, 0)
        torch.cuda.synchronize()
        mixes.append(out.cpu().numpy())
        if buf is not None:
            probes.append(buf[:f * len(probe_ids) * 2].cpu().numpy().reshape(f, len(probe_ids), 2).copy())
        kinds.append((db.last_kernel(), db.last_pack(), expected_slots(host)))
    got = bank.copy()
    db.download(got)
    viol = db.list_violations()
    if probe_ids is not None:
        db.set_probe([], 0)
    db.close()
    return got, mixes, probes, kinds, viol


def oracle_blocks(bank, tables, g, blocks, actions=None, want_stems=False):
    ref, ref_g = bank.copy(), g.copy()
    mixes, stems = [], []
    for k, f in enumerate(blocks):
        if actions and k in actions:
            actions[k](None, ref)
        r = cpuref.render(ref, ref_g, tables, f, 0, want_stems=want_stems)
        mixes.append(cpuref.master(ref_g, r["sum64"].astype(np.float32)))
        if want_stems:
            stems.append(r["stems"])
    return ref, mixes, stems


@pytest.mark.parametrize("seed,n,live", [(1, 4096, 4), (2, 8192, 7), (3, 2048, 14), (4, 16384, 2), (5, 4096, 16)])
def test_sparse_banks_packed_against_the_oracle(dev, seed, n, live):
    bank, tables, g = sparse_bank(seed, n, live)
    blocks = [512, 96, 33, 512]
    ref, ref_mixes, ref_stems = oracle_blocks(bank, tables, g, blocks, want_stems=True)
    rng = np.random.default_rng(seed)
    cand = np.flatnonzero(bank["voice_amp"] != 0)
    ids = np.unique(np.concatenate([rng.choice(cand, 48, replace=False), rng.choice(n, 12, replace=False)])).astype(np.int32)
    for pack in (2, 0):
        got, mixes, probes, kinds, viol = run_blocks(dev, bank, tables, g, blocks, pack, probe_ids=ids)
        assert viol == 0
        assert all(k[0] == 1 for k in kinds), kinds
        if pack:
            assert all(0 < k[1] <= 32 and k[1] == k[2] for k in kinds), kinds
        else:
            assert all(k[1] == 0 for k in kinds), kinds
        bad = got.rw_equal(ref)
        assert not bad, (pack, bad)
        for k, f in enumerate(blocks):
            want = ref_stems[k][:, ids, :]
            finite = np.isfinite(want)
            diff = np.argwhere(np.where(finite, probes[k], 0).view(np.uint32) != np.where(finite, want, 0).view(np.uint32))
            assert len(diff) == 0, f"pack={pack} block {k}: {len(diff)} probed values differ; first {diff[0]} voice {ids[diff[0][1]]}"
            if np.isfinite(ref_mixes[k]).all():
                assert rel_rms(mixes[k], ref_mixes[k]) <= 1e-5, (pack, k)


@pytest.mark.parametrize("patch", ["37sk", "3sk"])
def test_tiled_patches(dev, patch):
    """The shipped patches tiled over a bank (banks.bank_patch: what tools/ab.py patch times).  banks.bank_patch tiles the voices a
    patch USES, so most of these banks are dense; 37.sk's unit (voices 0..4 and 10) leaves every other lane empty and runs packed
    (forced here: the library's own rule wants a bank that fills the machine -- tests/test_patch_banks.py renders these banks at
    2^20 voices), 3.sk has nothing to pack.  Voices of a few groups probed, bit for bit; state of the whole bank."""
    n = 1 << 14
    bank, tables, g = banks.bank_patch(patch, n)
    blocks = [512, 512, 100]
    ref, ref_mixes, ref_stems = oracle_blocks(bank, tables, g, blocks, want_stems=True)
    ids = np.concatenate([np.arange(12), 64 * 100 + np.arange(12), n - 64 + np.arange(12)]).astype(np.int32)
    got, mixes, probes, kinds, viol = run_blocks(dev, bank, tables, g, blocks, 2, probe_ids=ids)
    assert viol == 0
    assert all(k[0] == 1 and k[1] == k[2] for k in kinds), kinds
    assert (kinds[0][1] > 0) == (patch == "37sk"), kinds
    assert not got.rw_equal(ref), got.rw_equal(ref)
    for k in range(len(blocks)):
        want = ref_stems[k][:, ids, :]
        assert gio.bits_equal(probes[k], want), f"{patch}: probed stems of block {k} differ"
        assert rel_rms(mixes[k], ref_mixes[k]) <= 1e-5


def test_sparse_global_table_bank_and_a_large_plain_one(dev):
    """(i) Tables too large for the LDS (the one-voice kernel's window instantiations, packed); (ii) a plain bank the two-per-lane
    kernel would take (2^18 voices, no modulation, nothing exotic) with one voice in twenty sounding: packed on the one-voice
    kernel by the rule for such banks (<= 16 lanes per group).  State bit for bit, mix within 1e-5."""
    rng = np.random.default_rng(9)
    bank, tables, g = banks.bank_c4(8192)
    amp = np.asarray(bank["voice_amp"]).copy()
    amp[rng.random(8192) < 0.9] = 0.0
    bank["voice_amp"] = amp
    blocks = [512, 64]
    for interp in (0,):
        ref, ref_mixes, _ = oracle_blocks(bank, tables, g, blocks)
        got, mixes, _, kinds, viol = run_blocks(dev, bank, tables, g, blocks, 2)
        assert viol == 0 and all(k[0] == 1 and k[1] == k[2] and k[1] > 0 for k in kinds), kinds
        assert not got.rw_equal(ref), got.rw_equal(ref)
        for k in range(len(blocks)):
            assert rel_rms(mixes[k], ref_mixes[k]) <= 1e-5
    n = 1 << 18
    bank, tables, g = banks.bank_c2(n)
    amp = np.asarray(bank["voice_amp"]).copy()
    amp[rng.random(n) < 0.95] = 0.0
    bank["voice_amp"] = amp
    ref, ref_mixes, _ = oracle_blocks(bank, tables, g, blocks)
    got, mixes, _, kinds, viol = run_blocks(dev, bank, tables, g, blocks, 1)
    exp = expected_slots(bank, 16)
    assert exp == 16, exp
    assert viol == 0 and all(k[0] == 1 and k[1] == 16 for k in kinds), kinds
    assert not got.rw_equal(ref), got.rw_equal(ref)
    for k in range(len(blocks)):
        assert rel_rms(mixes[k], ref_mixes[k]) <= 1e-5


@pytest.mark.parametrize("seed,n,live", [(31, 4096, 4), (32, 2048, 10), (33, 8192, 2)])
def test_sparse_modulated_banks_packed(dev, seed, n, live):
    """The modulated kernel (modulators anywhere in the group -- below the carrier: same-frame dependencies, dependency levels --
    and CZ phase distortion) packs the same way: whole-bank state bit for bit after several blocks, packed and not; 18.sk tiled
    (a modulator below its carrier; 16 voices of 64 in use)."""
    bank, tables, g = sparse_bank(seed, n, live)
    rng = np.random.default_rng(seed)
    base = (np.arange(n) // 64) * 64
    for key, p in (("voice_freq_mod_osc", 0.3), ("voice_amp_mod_osc", 0.2), ("voice_pan_mod_osc", 0.15), ("voice_cz_mod_osc", 0.3)):
        bank[key] = np.where(rng.random(n) < p, base + rng.integers(0, 64, n), -1).astype(np.int32)
    bank["voice_cz_mod_depth"] = (rng.random(n) * 2).astype(np.float32)
    bank["voice_cz_mode"] = np.where(rng.random(n) < 0.3, rng.integers(1, 8, n), 0).astype(np.int32)
    bank["voice_cz_distortion"] = rng.random(n).astype(np.float32)
    blocks = [400, 64, 33]
    ref, ref_mixes, _ = oracle_blocks(bank, tables, g, blocks)
    for pack in (2, 0):
        got, mixes, _, kinds, viol = run_blocks(dev, bank, tables, g, blocks, pack)
        assert viol == 0
        assert all(k[0] == 2 for k in kinds), kinds
        if pack:
            exp = expected_slots_mod(bank)
            assert exp > 0 and all(k[1] == exp for k in kinds), (exp, kinds)
        bad = got.rw_equal(ref)
        assert not bad, (pack, bad)
        for k in range(len(blocks)):
            if np.isfinite(ref_mixes[k]).all():
                assert rel_rms(mixes[k], ref_mixes[k]) <= 1e-5, (pack, k)


def test_tiled_18sk_runs_packed_on_the_modulated_kernel(dev):
    n = 1 << 14
    bank, tables, g = banks.bank_patch("18sk", n)
    blocks = [512, 100]
    ref, ref_mixes, _ = oracle_blocks(bank, tables, g, blocks)
    got, mixes, _, kinds, viol = run_blocks(dev, bank, tables, g, blocks, 2)
    assert viol == 0 and all(k[0] == 2 and k[1] == 16 for k in kinds), kinds
    assert not got.rw_equal(ref), got.rw_equal(ref)
    for k in range(len(blocks)):
        assert rel_rms(mixes[k], ref_mixes[k]) <= 1e-5


def test_voices_switched_on_and_off_between_blocks(dev):
    """Control actions that change who can sound: amp 0 -> x on voices of groups that had none / few (the slots grow from 4 to 8
    lanes), amp x -> 0 (the voice loses its lane; its voice_sample is cleared as the reference's skip rule clears it -- unless a
    sounding carrier still names it, then it keeps a lane and is skipped at run time), a voice_sample written onto a skipped
    voice.  State bit for bit after every stage; packed throughout."""
    from skred_amd import device
    n = 4096
    bank, tables, g = sparse_bank(11, n, 4, with_env=False)
    rng = np.random.default_rng(5)

    def on(db, h):
        vs = np.concatenate([64 * 3 + np.arange(1, 40, 3), [64 * 9 + 0, 64 * 40 + 63]]).astype(np.int32)
        h["voice_amp"][vs] = 0.9
        if db: db.update(h, vs, device.DIRTY_PARAMS)

    def off(db, h):
        live = np.flatnonzero(h["voice_amp"] != 0)
        vs = rng.choice(live, 200, replace=False).astype(np.int32) if db else off.vs
        off.vs = vs
        h["voice_amp"][vs] = 0.0
        if db: db.update(h, vs, device.DIRTY_PARAMS)

    def poke(db, h):
        vs = np.flatnonzero(h["voice_amp"] == 0)[:50].astype(np.int32)
        h["voice_sample"][vs] = 0.25
        if db: db.update(h, vs, device.DIRTY_SAMPLE)

    blocks = [256, 256, 256, 256]
    actions = {1: on, 2: off, 3: poke}
    got, mixes, _, kinds, viol = run_blocks(dev, bank, tables, g, blocks, 2, actions=actions)
    ref, ref_mixes, _ = oracle_blocks(bank, tables, g, blocks, actions=actions)
    assert viol == 0
    assert all(k[1] > 0 and k[1] == k[2] for k in kinds), kinds
    assert kinds[1][1] > kinds[0][1], kinds             # thirteen more voices in group 3: the slots grow
    assert not got.rw_equal(ref), got.rw_equal(ref)
    for k in range(len(blocks)):
        assert rel_rms(mixes[k], ref_mixes[k]) <= 1e-5, k


def test_packing_is_left_alone_where_it_does_not_pay(dev):
    """A full bank, a sparse bank that does not fill the machine (the default rule), a launch with the stem buffer: not packed;
    SKRED_OPT_PACK = 2 packs whatever has a wavefront to lose; 0 never packs."""
    import torch
    bank, tables, g = banks.bank_c2(4096)
    _, _, _, kinds, _ = run_blocks(dev, bank, tables, g, [128], 2)
    assert kinds[0][1] == 0
    bank, tables, g = sparse_bank(21, 4096, 40)
    _, _, _, kinds, _ = run_blocks(dev, bank, tables, g, [128], 1)
    assert kinds[0][1] == 0, kinds
    bank, tables, g = sparse_bank(22, 4096, 3)
    for mode, want in ((0, 0), (1, 0), (2, expected_slots(bank))):
        _, _, _, kinds, _ = run_blocks(dev, bank, tables, g, [128], mode)
        assert kinds[0][1] == want, (mode, kinds)
    db = dev.DeviceBank(bank.n)
    db.set_tables(tables); db.upload(bank); db.set_globals(g); db.set_pack(2)
    mix, stems = db.render_host(64, 2, 0, want_stems=True)
    assert db.last_pack() == 0
    out = torch.zeros(64, 2, device="cuda")
    db.render_mix(64, out.data_ptr(), 2, 0, 0)
    torch.cuda.synchronize()
    assert db.last_pack() == expected_slots(bank)
    db.close()
