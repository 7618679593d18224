"""GPU tests of sk_render_split_kernel (skred_render_split.hip): the one-voice-per-lane family with every frame split between
an oscillator wave and a post wave -- an OPTION (SKRED_OPT_SPLIT; off by default) for small clean banks
while nothing moves.

The checker is the oracle (oracle/cpu_ref.c, pinned bit for bit to the compiled reference by test_oracle_vs_golden.py); every
call goes through the C ABI.  Bars as everywhere: per-voice read-write state BIT-EXACT, float mix within 1e-5 (tree sum against
the reference's voice-order sum).  What these tests add to test_gpu_parity.py, which reaches the same kernel through the
library's own choice: the kernel FORCED on (option value 2) at sizes and block lengths the default rule would not send to it,
its in-kernel general path (waves that are not steady or not tame after all), two passes per workgroup, and a proof per test
that the split form actually ran.
"""
import numpy as np
import pytest

from oracle import cpuref
from skred_amd import banks

pytestmark = pytest.mark.gpu


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


def _sustained(recipe, n):
    """The recipe with every note long in its sustain stage and every amp smoother free to settle: nothing moves."""
    bank, tables, g = banks.RECIPES[recipe](n)
    e = bank["voice_amp_envelope"]
    e["sample_start"][:] = np.uint64(g.synth_sample_count - 30000)
    e["sample_release"][:] = 0
    e["is_active"][:] = 1
    return bank, tables, g


def _render_blocks(dev, bank, tables, g, interp, blocks, split, setup=None):
    """Asynchronous blocks (the way the host's reports arrive), the oracle beside them block by block."""
    import torch
    db = dev.DeviceBank(bank.n)
    db.set_tables(tables)
    host = bank.copy()
    db.upload(host)
    db.set_globals(g)
    db.fast2_min_voices(1 << 30)                      # the one-voice family at every size
    if split is not None:
        db.set_split(split)
    if setup:
        setup(db)
    ref_host, ref_g = bank.copy(), g.copy()
    took = []
    for k, frames in enumerate(blocks):
        out = torch.zeros(frames, 2, device="cuda")
        db.render_mix(frames, out.data_ptr(), 2, 0, interp)
        took.append(db.last_split())
        assert db.last_kernel() == 1
        torch.cuda.synchronize()
        r = cpuref.render(ref_host, ref_g, tables, frames, interp)
        ref_mix = cpuref.master(ref_g, r["sum64"].astype(np.float32))
        assert rel_rms(out.cpu().numpy(), ref_mix) <= 1e-5, f"block {k} ({frames} frames, split {took[-1]})"
    db.download(host)
    db.close()
    bad = host.rw_equal(ref_host)
    assert not bad, bad
    return took


@pytest.mark.parametrize("recipe,interp,n,pairs", [("c1", 0, 4096, 2), ("c2", 0, 5000, 2), ("c2", 1, 3000, 4), ("c1", 1, 700, 2), ("c2", 0, 65536, 0),
                                                    ("c1", 0, 4096, 4), ("c2", 1, 20000, 2)])
def test_split_steady_bank_vs_oracle(dev, recipe, interp, n, pairs):
    """Sustained banks, block lengths that leave every kind of tail: one short block (7), no block at all (1, 3), whole blocks
    and a ragged chunk (100 = 64 + 36 = 12 blocks + 4), many chunks (777 = 12 chunks + 9 frames, 4800), and the callback size.
    Both workgroup shapes: four pairs (`pairs` 0: the default shape) and two."""
    bank, tables, g = _sustained(recipe, n)
    bank["voice_amp"][::11] = 0.0                     # skipped voices (state frozen, sample = 0)
    blocks = [512, 512, 512, 7, 1, 3, 8, 9, 63, 64, 65, 100, 777, 512, 4800, 16, 512]
    took = _render_blocks(dev, bank, tables, g, interp, blocks, split=2, setup=lambda db: db.set_split_pairs(pairs))
    # an enveloped bank waits for a launch to report that nothing moved; from then on every block is the split form
    assert all(took[3:]), took
    assert not took[0], "the first block after an upload must not assume anything about envelopes"


def test_split_is_off_by_default_and_value_one_has_a_size_rule(dev):
    """SKRED_OPT_SPLIT: 0 is the default (the library never picks the split form by itself: it measured at best 1.4 % faster);
    value 1 takes filtered banks of half a 256-voice group to one group per CU once they have reported quiet, and nothing else."""
    bank, tables, g = _sustained("c2", 40000)
    assert not any(_render_blocks(dev, bank, tables, g, 0, [256, 256, 256, 256], split=None))
    assert all(_render_blocks(dev, bank, tables, g, 0, [256, 256, 256, 256, 256], split=1)[3:])
    small, tables, g = _sustained("c2", 4096)
    assert not any(_render_blocks(dev, small, tables, g, 0, [256, 256, 256, 256], split=1))
    big, tables, g = _sustained("c2", 150000)
    assert not any(_render_blocks(dev, big, tables, g, 0, [128, 128, 128, 128], split=1))


def test_split_muted_voices_and_untame_waves(dev):
    """Muted (`m1`) live voices: rendered, kept out of the mix by a select (their wave stays on the split path).  Increments
    beyond half a loop: such a wave is not tame, its owner wave renders it alone on the kernel's general path while its
    neighbours in the same workgroup run split -- state bit for bit either way."""
    n = 4096
    bank, tables, g = _sustained("c2", n)
    bank["voice_disconnect"][5::7] = 1
    size = bank["voice_table_size"].astype(np.float32)
    wild = np.arange(640, 900)                        # four waves and a bit, in two workgroups
    bank["voice_phase_inc"][wild[::3]] = size[wild[::3]] * np.float32(2.37)
    bank["voice_phase_inc"][wild[1::3]] = size[wild[1::3]] * np.float32(0.999)
    took = _render_blocks(dev, bank, tables, g, 0, [512, 512, 300, 512, 65, 512], split=2)
    assert all(took[3:]), took


def test_split_unfiltered_bank_keeps_a_stale_delay_line(dev):
    """A bank without any filter (the C1 recipe) whose voices still carry delay-line values from an earlier life: mmf_process is
    skipped (synth.c:577), so x1 x2 y1 y2 must come back exactly as uploaded -- also after an ODD number of frames (the
    frame pairs swap the roles of the two delay-line slots; only a filtered voice may be swapped back)."""
    n = 2048
    bank, tables, g = _sustained("c1", n)
    rng = np.random.default_rng(7)
    f = bank["voice_filter"]
    for k in ("x1", "x2", "y1", "y2"):
        f[k][:] = rng.standard_normal(n).astype(np.float32)
    for split in (0, 2):
        _render_blocks(dev, bank, tables, g, 0, [33, 33, 33, 5, 511, 1, 33], split=split)


def test_split_two_passes_per_workgroup(dev):
    """A bank of more 256-voice groups than the grid has workgroups (2 048): every workgroup renders two or three groups one
    after the other, its row of the partial mix accumulated across the passes and published in the last one."""
    n = 2048 * 256 + 77 * 256 + 100
    bank, tables, g = _sustained("c2", n)
    took = _render_blocks(dev, bank, tables, g, 0, [128, 128, 128, 200, 64], split=2)
    assert all(took[3:]), took


def test_split_general_path_note_on_ahead_of_the_clock(dev):
    """The one way a constant level ends without a control action: a note-on written AHEAD of the clock (synth.c:401).  The
    library never launches the split form while such a note is pending (sk_render_fast_kernel's envelope instantiation does
    not report "quiet" before it has started and come to rest), so the kernel is forced on regardless of the hint (option value
    3): the waves holding such voices find out for themselves and their owner waves walk the launch on integer clocks on the
    kernel's general path, beside split waves in the same workgroups.  Against the oracle block by block."""
    n = 6000
    bank, tables, g = _sustained("c2", n)
    e = bank["voice_amp_envelope"]
    late = np.arange(17, n, 97)
    e["sample_start"][late] = (g.synth_sample_count + 2500 + (late % 7) * 300).astype(np.uint64)
    took = _render_blocks(dev, bank, tables, g, 0, [512] * 14, split=3)
    assert all(took), took


@pytest.mark.parametrize("recipe,interp", [("c1", 0), ("c2", 0), ("c2", 1)])
def test_split_general_path_envelopes_in_motion(dev, recipe, interp):
    """The recipe from its first frame (attack / decay in flight on a ninth of the voices), a note-off on every other voice,
    releases running out, smoother tails -- on the split kernel whatever the host believes (option value 3): every wave that
    holds a moving envelope renders itself on the general path and comes back to the split form when its voices rest."""
    import torch
    n = 3000
    bank, tables, g = banks.RECIPES[recipe](n)
    bank["voice_disconnect"][::7] = 1
    bank["voice_amp"][::11] = 0.0
    db = dev.DeviceBank(n)
    db.set_tables(tables)
    host = bank.copy()
    db.upload(host)
    db.set_globals(g)
    db.set_split(3)
    ref_host, ref_g = bank.copy(), g.copy()
    for k, frames in enumerate([301, 333, 4000, 2000, 64, 512, 9000, 512]):
        if k == 2:                                    # == amp_envelope_release on the odd voices, synth.c:391-395
            db.download(host)
            now = db.get_globals().synth_sample_count
            host["voice_amp_envelope"]["sample_release"][1::2] = now
            ref_host["voice_amp_envelope"]["sample_release"][1::2] = now
            db.upload(host)
        out = torch.zeros(frames, 2, device="cuda")
        db.render_mix(frames, out.data_ptr(), 2, 0, interp)
        assert db.last_split() and db.last_kernel() == 1
        torch.cuda.synchronize()
        r = cpuref.render(ref_host, ref_g, tables, frames, interp)
        ref_mix = cpuref.master(ref_g, r["sum64"].astype(np.float32))
        assert rel_rms(out.cpu().numpy(), ref_mix) <= 1e-5, f"block {k}"
    db.download(host)
    db.close()
    bad = host.rw_equal(ref_host)
    assert not bad, bad


def test_split_equals_the_unsplit_kernel_bit_for_bit(dev):
    """Same per-voice samples, same order of the wave / workgroup / block sums: with four pairs per workgroup (the shape of
    sk_render_fast_kernel's 256-voice passes; forced here -- a bank this small would get the two-pair shape, whose 128-voice
    rows add up in another order) the mix of the split form equals the mix of sk_render_fast_kernel BYTE for byte."""
    import torch
    n = 20000
    bank, tables, g = _sustained("c2", n)
    outs = []
    for split in (0, 2):
        db = dev.DeviceBank(n)
        db.set_tables(tables)
        db.upload(bank.copy())
        db.set_globals(g)
        db.fast2_min_voices(1 << 30)
        db.set_split(split)
        db.set_split_pairs(4)
        got = []
        for frames in (512, 512, 512, 512, 100, 777):
            out = torch.zeros(frames, 2, device="cuda")
            db.render_mix(frames, out.data_ptr(), 2, 0, 0)
            torch.cuda.synchronize()
            got.append((out.cpu().numpy(), db.last_split()))
        db.close()
        outs.append(got)
    assert not any(s for _, s in outs[0]) and all(s for _, s in outs[1][3:])
    for (a, _), (b, _) in zip(outs[0], outs[1]):
        assert (a.view(np.uint32) == b.view(np.uint32)).all()
