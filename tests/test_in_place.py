"""The motion list of the two-voices-per-lane path rendered IN PLACE (round 3; DESIGN "The motion list"): sk_gain_kernel
walks the envelopes of the listed voices one block ahead into per-frame gain rows, the steady kernel's in-place
instantiation keeps those voices in their lanes and feeds their smoothers from the rows.  Same per-voice arithmetic as the
envelope kernel beside the steady one (SKRED_OPT_IN_PLACE 0) and as the reference: per-voice state bit-exact against the
oracle, the mix within the float tolerance of a different summation order.  Covered: sparse lists (staged 8-frame blocks),
more listed voices in one 128-voice wave than it stages and more in one 64-voice word than it owns rows for (the overflow
rows), ragged block lengths (the frame-by-frame tail), a note-on ahead of the clock, releases that run out mid-block, a
burst that outgrows the proven bound (the envelope kernel takes that block), and un-enveloped voices in a mixed bank."""
import numpy as np
import pytest

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


def stamp(truth, vs, dirty, now, D):
    """amp_envelope_trigger / amp_envelope_release (synth.c:383-395) as plain stores, on the oracle's bank."""
    e = truth["voice_amp_envelope"]
    if dirty & D.DIRTY_PHASE:
        truth["voice_phase"][vs] = 0.0
        truth["voice_finished"][vs] = 0
    if dirty & D.STAMP_TRIGGER:
        e["sample_start"][vs] = now
        e["sample_release"][vs] = 0
        e["is_active"][vs] = 1
    if dirty & D.STAMP_RELEASE:
        act = e["is_active"][vs] != 0
        e["sample_release"][vs[act]] = now


def run(dev, n, frames, blocks, plan, mode, bank_fn=banks.bank_c2, prepare=None, interp=0):
    """Render `blocks` blocks with the control actions of `plan` ({block: [(voices, dirty), ...]}); returns the mixes, the
    oracle's mixes, the downloaded state, the oracle's state and which blocks took the in-place path."""
    import torch
    D = dev
    bank, tables, g = bank_fn(n)
    if prepare:
        prepare(bank, g)
    db = dev.DeviceBank(n)
    db.set_tables(tables)
    db.upload(bank)
    db.set_globals(g)
    db.fast2_min_voices(0)
    db.in_place(mode)
    mirror, truth, gl = bank.copy(), bank.copy(), g.copy()
    out = torch.zeros(frames, 2, device="cuda")
    mixes, refs, taken = [], [], []
    for k in range(blocks):
        for vs, dirty in plan.get(k, []):
            vs = np.asarray(vs, np.int32)
            if dirty & D.DIRTY_PHASE:
                mirror["voice_phase"][vs] = 0.0
                mirror["voice_finished"][vs] = 0
            db.update(mirror, vs, dirty, 0)
            stamp(truth, vs, dirty, gl.synth_sample_count, D)
        db.render_mix(frames, out.data_ptr(), 2, 0, interp)     # asynchronous blocks: the way the host gets its reports
        assert db.last_kernel() == 3
        taken.append(db.last_in_place())
        torch.cuda.synchronize()
        mixes.append(out.cpu().numpy().copy())
        r = cpuref.render(truth, gl, tables, frames, interp)
        refs.append(cpuref.master(gl, r["sum64"].astype(np.float32)))
    got = bank.copy()
    db.download(got)
    viol = db.list_violations()
    db.close()
    assert viol == 0
    return mixes, refs, got, truth, taken


def traffic(n, D, dense_at=None, thin=1):
    plan = {}
    for k in range(3, 40, 2):
        plan.setdefault(k, []).append((np.arange(5 + k, n, 401 * thin), D.STAMP_RELEASE))
        plan.setdefault(k + 1, []).append((np.arange(9 + 3 * k, n, 331 * thin), D.STAMP_TRIGGER | D.DIRTY_PHASE))
    if dense_at is not None:
        # 44 consecutive voices of one 128-voice wave (more than it stages; one of its 64-voice words holds more than its own
        # rows), and all 64 of another word
        plan.setdefault(dense_at, []).append((np.arange(1300, 1344), D.STAMP_TRIGGER | D.DIRTY_PHASE))
        plan.setdefault(dense_at, []).append((np.arange(2560, 2624), D.STAMP_RELEASE))
        plan.setdefault(dense_at + 3, []).append((np.arange(1290, 1350), D.STAMP_RELEASE))
    return plan


@pytest.mark.parametrize("frames", [512, 300, 140])
def test_listed_voices_in_place_against_the_oracle(dev, frames):
    n = 6000
    blocks = 36 if frames >= 300 else 48
    plan = traffic(n, dev, dense_at=17, thin=1 if frames >= 300 else 3)
    mixes, refs, got, truth, taken = run(dev, n, frames, blocks, plan, 2)
    assert not got.rw_equal(truth), got.rw_equal(truth)          # every state field of every voice, bit for bit
    for k, (m, r) in enumerate(zip(mixes, refs)):
        assert rel_rms(m, r) <= 1e-5, f"block {k}"
    assert sum(taken) >= blocks // 2, taken                       # the path under test did run


def _c2_without_guards(n):
    """bank_c2 with the sample behind every table changed: no voice carries SKF_GUARD, the linear lookup keeps its fold test."""
    bank, tables, g = banks.bank_c2(n)
    t = tables.copy()
    pos = np.unique(bank["voice_table_offset"].astype(np.int64) + bank["voice_table_size"].astype(np.int64))
    t[pos[pos < len(t)]] = 7.0
    return bank, t, g


@pytest.mark.parametrize("bank_fn", [banks.bank_c2, _c2_without_guards], ids=["guarded_pool", "general_form"])
def test_listed_voices_in_place_with_linear_lookup(dev, bank_fn):
    """The in-place instantiations of both linear forms (INTERP 2: every voice loops over its whole table with a guard sample
    behind it; INTERP 1: the general form with the fold test)."""
    n = 6000
    plan = traffic(n, dev, dense_at=13)
    mixes, refs, got, truth, taken = run(dev, n, 512, 26, plan, 2, bank_fn=bank_fn, interp=1)
    assert not got.rw_equal(truth), got.rw_equal(truth)
    for k, (m, r) in enumerate(zip(mixes, refs)):
        assert rel_rms(m, r) <= 1e-5, f"block {k}"
    assert any(taken), taken


def test_in_place_and_envelope_kernel_agree(dev):
    """The same traffic through both ways of rendering the list: per-voice state identical bit for bit."""
    n = 6000
    plan = traffic(n, dev, dense_at=11)
    a = run(dev, n, 512, 30, plan, 2)
    b = run(dev, n, 512, 30, plan, 0)
    assert any(a[4]) and not any(b[4])
    assert not a[2].rw_equal(b[2]), a[2].rw_equal(b[2])
    assert rel_rms(np.concatenate(a[0]), np.concatenate(b[0])) <= 1e-5


def test_burst_beyond_the_proven_bound_goes_to_the_envelope_kernel(dev):
    """The gain rows are a buffer of fixed size; the in-place path is taken only while the length the device last reported plus
    the voices touched since stays below it.  A burst that touches most of the bank must be rendered by the envelope kernel
    beside the steady one in that block -- and correctly."""
    n = 6000
    plan = traffic(n, dev)
    plan.setdefault(20, []).append((np.arange(0, n, 2), dev.STAMP_TRIGGER | dev.DIRTY_PHASE))
    mixes, refs, got, truth, taken = run(dev, n, 512, 34, plan, 2)
    assert taken[19] and not taken[20], taken
    assert not got.rw_equal(truth), got.rw_equal(truth)
    for k, (m, r) in enumerate(zip(mixes, refs)):
        assert rel_rms(m, r) <= 1e-5, f"block {k}"


def test_note_on_ahead_of_the_clock_in_place(dev):
    """sample_start ahead of the clock (synth.c:401: the wrapped difference reads as sustain until the clock gets there):
    sk_gain_kernel evaluates every frame on integer clocks, as the reference does."""
    n = 6000

    def prepare(bank, g):
        e = bank["voice_amp_envelope"]
        now0 = int(g.synth_sample_count)
        e["sample_start"][:] = np.uint64(max(now0 - 40000, 0))
        e["sample_release"][:] = 0
        e["is_active"][:] = 1
        late = np.arange(17, n, 97)
        e["sample_start"][late] = (now0 + 2500 + (late % 7) * 300).astype(np.uint64)

    plan = {2: [(np.arange(3, n, 500), dev.STAMP_RELEASE)]}       # (a control action: the list's length gets reported)
    mixes, refs, got, truth, taken = run(dev, n, 512, 16, plan, 2, prepare=prepare)
    assert not got.rw_equal(truth), got.rw_equal(truth)
    for k, (m, r) in enumerate(zip(mixes, refs)):
        assert rel_rms(m, r) <= 1e-5, f"block {k}"
    assert any(taken), taken


def test_mixed_bank_in_place(dev):
    """A bank in which a fifth of the voices has no envelope and a third no filter (the per-lane flags of the MIXED
    instantiations): an un-enveloped voice that a control action puts on the list gets a constant row (amp * 1.0f)."""
    n = 6000

    def prepare(bank, g):
        bank["voice_use_amp_envelope"][::5] = 0
        bank["voice_filter_mode"][::3] = 0

    plan = traffic(n, dev, dense_at=9)
    mixes, refs, got, truth, taken = run(dev, n, 512, 26, plan, 2, prepare=prepare)
    assert not got.rw_equal(truth), got.rw_equal(truth)
    for k, (m, r) in enumerate(zip(mixes, refs)):
        assert rel_rms(m, r) <= 1e-5, f"block {k}"
    assert any(taken), taken


def test_block_length_changes_from_block_to_block(dev):
    """The gain rows are as long as the block: a host that changes its callback size (520, 64, 200, 512 ... frames) makes the
    buffer's row stride change between blocks.  Oracle, block by block; one update batch travels on a stream of its own."""
    import torch
    D = dev
    n = 6000
    bank, tables, g = banks.bank_c2(n)
    db = dev.DeviceBank(n)
    db.set_tables(tables)
    db.upload(bank)
    db.set_globals(g)
    db.fast2_min_voices(0)
    db.in_place(2)
    mirror, truth, gl = bank.copy(), bank.copy(), g.copy()
    rng = np.random.default_rng(11)
    other = torch.cuda.Stream()
    sizes = [512, 520, 64, 200, 512, 136, 512, 8, 300, 512, 72, 512, 512, 264, 512, 512]
    taken = []
    for k, frames in enumerate(sizes * 2):
        if k >= 2:
            vs = rng.choice(n, 24, replace=False).astype(np.int32)
            for part, dirty in ((vs[:12], D.STAMP_RELEASE), (vs[12:], D.STAMP_TRIGGER | D.DIRTY_PHASE)):
                if dirty & D.DIRTY_PHASE:
                    mirror["voice_phase"][part] = 0.0
                    mirror["voice_finished"][part] = 0
                if k % 5 == 0:                                 # (ordered by hand: the bank's streams are the caller's business)
                    torch.cuda.synchronize()
                    db.update(mirror, part, dirty, other.cuda_stream)
                    other.synchronize()
                else:
                    db.update(mirror, part, dirty, 0)
                stamp(truth, part, dirty, gl.synth_sample_count, D)
        out = torch.zeros(frames, 2, device="cuda")
        db.render_mix(frames, out.data_ptr(), 2, 0, 0)
        taken.append(db.last_in_place())
        torch.cuda.synchronize()
        r = cpuref.render(truth, gl, tables, frames, 0)
        ref = cpuref.master(gl, r["sum64"].astype(np.float32))
        assert rel_rms(out.cpu().numpy(), ref) <= 1e-5, f"block {k} ({frames} frames)"
    got = bank.copy()
    db.download(got)
    assert db.list_violations() == 0
    db.close()
    assert not got.rw_equal(truth), got.rw_equal(truth)
    assert sum(taken) >= len(taken) // 2, taken


def test_sum_only_form_in_place(dev):
    """The multi-GPU render's two halves (skred_bank_render: the pre-master sum; skred_bank_master behind it) with the list in
    place: the same bytes as the one-launch form of a twin bank under the same traffic."""
    import torch
    D = dev
    n, F = 6000, 512
    bank, tables, g = banks.bank_c2(n)
    plan = traffic(n, dev, dense_at=7)
    outs = []
    for two_halves in (False, True):
        db = dev.DeviceBank(n)
        db.set_tables(tables)
        db.upload(bank)
        db.set_globals(g)
        db.fast2_min_voices(0)
        db.in_place(2)
        mirror = bank.copy()
        out = torch.zeros(F, 2, device="cuda")
        pre = torch.zeros(F, 2, device="cuda")
        got, taken = [], []
        for k in range(20):
            for vs, dirty in plan.get(k, []):
                vs = np.asarray(vs, np.int32)
                if dirty & D.DIRTY_PHASE:
                    mirror["voice_phase"][vs] = 0.0
                    mirror["voice_finished"][vs] = 0
                db.update(mirror, vs, dirty, 0)
            if two_halves:
                db.render(F, pre.data_ptr(), 0, 0, 0)
                db.master(pre.data_ptr(), F, out.data_ptr(), 2, 0)
            else:
                db.render_mix(F, out.data_ptr(), 2, 0, 0)
            taken.append(db.last_in_place())
            torch.cuda.synchronize()
            got.append(out.cpu().numpy().copy())
        assert db.list_violations() == 0
        db.close()
        assert any(taken), taken
        outs.append(np.concatenate(got))
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))


@pytest.mark.parametrize("mode", [2, 1, 0, -1])
def test_blocks_queued_far_ahead_of_the_device(dev, mode):
    """A host that never waits: 120 blocks with notes starting and ending in every one of them are queued back to back, so the
    device's reports lag the host by many blocks (the proven bound on the list's length has to live with stale reports), the
    update path's staging ring wraps many times (its slots come back through the kernels' done words) and the report ring is
    re-used.  Afterwards every state field of every voice must equal the oracle's after the same 120 blocks."""
    import torch
    D = dev
    n, F, blocks = 6000, 512, 120
    bank, tables, g = banks.bank_c2(n)
    db = dev.DeviceBank(n)
    db.set_tables(tables)
    db.upload(bank)
    db.set_globals(g)
    db.fast2_min_voices(0 if mode >= 0 else 1 << 30)          # (-1: the same traffic on the one-voice family)
    if mode >= 0:
        db.in_place(mode)
    mirror, truth, gl = bank.copy(), bank.copy(), g.copy()
    out = torch.zeros(F, 2, device="cuda")
    rng = np.random.default_rng(3)
    actions, taken = [], []
    for k in range(blocks):
        vs = rng.choice(n, 10, replace=False).astype(np.int32)
        acts = [(vs[:5], D.STAMP_RELEASE), (vs[5:], D.STAMP_TRIGGER | D.DIRTY_PHASE)]
        if k == 60:
            acts.append((np.arange(0, n, 3, dtype=np.int32), D.STAMP_RELEASE))          # a burst: a third of the bank
        for part, dirty in acts:
            if dirty & D.DIRTY_PHASE:
                mirror["voice_phase"][part] = 0.0
                mirror["voice_finished"][part] = 0
            db.update(mirror, part, dirty, 0)
        actions.append(acts)
        db.render_mix(F, out.data_ptr(), 2, 0, 0)
        taken.append(db.last_in_place())
    torch.cuda.synchronize()
    last = out.cpu().numpy().copy()
    got = bank.copy()
    db.download(got)
    assert db.list_violations() == 0
    db.close()
    for k in range(blocks):
        for part, dirty in actions[k]:
            stamp(truth, part, dirty, gl.synth_sample_count, D)
        r = cpuref.render(truth, gl, tables, F, 0)
        ref = cpuref.master(gl, r["sum64"].astype(np.float32))
    assert not got.rw_equal(truth), got.rw_equal(truth)
    assert rel_rms(last, ref) <= 1e-5
    if mode == 2:
        assert sum(taken) > blocks // 3, taken
    if mode <= 0:
        assert not any(taken)


def test_default_rule_takes_the_path_on_a_full_machine_only_when_sparse(dev):
    """SKRED_OPT_IN_PLACE 1 (default) on a 2^19-voice bank (its 512 workgroup passes fill a 256-CU device exactly once): sparse
    traffic is rendered in place, the mix stays within tolerance of the envelope-kernel form of the same blocks."""
    import torch
    D = dev
    n, F = 1 << 19, 512
    bank, tables, g = banks.RECIPES["c3"](n)
    outs = {}
    for mode in (1, 0):
        db = dev.DeviceBank(n)
        db.set_tables(tables)
        db.upload(bank)
        db.set_globals(g)
        db.in_place(mode)
        out = torch.zeros(F, 2, device="cuda")
        rng = np.random.default_rng(5)
        taken, mix = [], []
        for k in range(40):
            if k >= 16:
                vs = rng.choice(n, 40, replace=False).astype(np.int32)
                db.update(bank, vs[:20], D.STAMP_RELEASE, 0)
                db.update(bank, vs[20:], D.STAMP_TRIGGER | D.DIRTY_PHASE | D.DIRTY_PARAMS, 0)
            db.render_mix(F, out.data_ptr(), 2, 0, 0)
            taken.append(db.last_in_place())
            if k >= 30:
                torch.cuda.synchronize()
                mix.append(out.cpu().numpy().copy())
        assert db.list_violations() == 0
        got = bank.copy()
        db.download(got)
        db.close()
        outs[mode] = (taken, np.concatenate(mix), got)
    if device_cus(dev) == 256:
        assert any(outs[1][0][20:]), outs[1][0]
    assert not any(outs[0][0])
    assert not outs[1][2].rw_equal(outs[0][2]), outs[1][2].rw_equal(outs[0][2])
    assert rel_rms(outs[1][1], outs[0][1]) <= 1e-5


def device_cus(dev):
    import torch
    return torch.cuda.get_device_properties(0).multi_processor_count
