"""Per-frame, per-voice evidence from INSIDE the block paths the benchmarks time (skred_bank_set_probe).

A launch that writes the full stem buffer takes the kernels' frame-by-frame paths, so the 8-frame tile blocks -- the code
bench.py times -- were only ever checked through end-of-block state and a 1e-5 mix (missing #5).  A probe names up
to 64 voices; the probe instantiations of the same kernels (same source, -DSK_PROBE_TU, the hook sits in the one function every
frame of every path goes through) write what the reference stores into its stem buffer for them (synth.c:607-611) while the
block renders on its fast paths.  Compared BIT FOR BIT with the oracle's stems of exactly those voices (clean banks: a voice's
samples do not depend on its neighbours, so the oracle renders the probed voices as a bank of their own, in lockstep).
"""
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


def _pick(n, k=64, seed=3):
    rng = np.random.default_rng(seed)
    ids = np.unique(np.concatenate([[0, 1, 63, 64, 127, 128, n - 1, n - 64, n - 65], rng.choice(n, k, replace=False)]))[:k]
    return np.sort(ids).astype(np.int32)


class Probed:
    """A device bank with a probe, and the oracle rendering the probed voices beside it."""

    def __init__(self, dev, bank, tables, g, ids, frames_max):
        import torch
        self.torch, self.ids, self.tables = torch, ids, tables
        self.db = dev.DeviceBank(bank.n)
        self.db.set_tables(tables)
        self.db.upload(bank)
        self.db.set_globals(g)
        self.buf = torch.zeros(frames_max * len(ids) * 2, device="cuda")
        self.db.set_probe(ids, self.buf.data_ptr())
        self.ref, self.ref_g = bank.take(ids), g.copy()

    def block(self, frames, interp=0, tag=""):
        out = self.torch.zeros(frames, 2, device="cuda")
        self.db.render_mix(frames, out.data_ptr(), 2, 0, interp)
        self.torch.cuda.synchronize()
        got = self.buf[:frames * len(self.ids) * 2].cpu().numpy().reshape(frames, len(self.ids), 2)
        want = cpuref.render(self.ref, self.ref_g, self.tables, frames, interp, want_stems=True)["stems"]
        bad = np.argwhere(got.view(np.uint32) != want.view(np.uint32))
        assert len(bad) == 0, f"{tag}: {len(bad)} probed values differ from the oracle's stems; first (frame, probe, ch) {bad[0]}, voice {self.ids[bad[0][1]]}: {got[tuple(bad[0])]} vs {want[tuple(bad[0])]}"
        return self.db.last_kernel()

    def update_ref(self, fn):
        fn(self.ref)

    def close(self):
        self.db.set_probe([], 0)
        self.db.close()


def test_headline_block_probed_inside_the_tile_blocks(dev):
    """BASELINE config 3 at full size (2^20 voices), the recipe from its first frame: eleven 512-frame blocks with envelopes in
    motion (the steady two-per-lane kernel with the envelope kernel beside it: a probed voice is written by whichever of the two
    renders it), then all-sustain blocks on the steady kernel alone -- the very block bench.py times -- with the truncating lookup
    and with the linear one; 64 probed voices, every frame, bit for bit."""
    n, F = 1 << 20, 512
    bank, tables, g = banks.bank_c2(n)
    bank["voice_disconnect"][5::1000] = 1
    bank["voice_amp"][7::1000] = 0.0
    ids = _pick(n)
    ids[10], ids[11] = 5005, 7007                     # a muted and a skipped voice among the probes: exact zeros
    ids = np.unique(ids).astype(np.int32)
    p = Probed(dev, bank, tables, g, ids, F)
    kernels = [p.block(F, 0, f"block {k}") for k in range(13)]
    kernels.append(p.block(F, 1, "linear"))
    kernels.append(p.block(F, 1, "linear, second block"))
    assert kernels == [3] * 15, kernels
    assert p.db.list_violations() == 0
    p.close()


@pytest.mark.parametrize("recipe,n,interp", [("c1", 4096, 0), ("c2", 65536, 0), ("c2", 131072, 0), ("c2", 65536, 1), ("c4", 20000, 1)])
def test_one_voice_kernel_probed_inside_the_tile_blocks(dev, recipe, n, interp):
    """The one-voice-per-lane kernel at the sizes bench.py times it (C1, C2, the 2^17-voice shard; a PCM bank through the table
    windows): the envelope instantiation's ramp blocks first, then the steady tile blocks (lo == 0 wrap, one-swap pan fold);
    ragged block lengths leave frame pairs and single frames behind the blocks."""
    bank, tables, g = banks.RECIPES[recipe](n)
    ids = _pick(n, 48)
    p = Probed(dev, bank, tables, g, ids, 777)
    p.db.fast2_min_voices(1 << 30)
    for k, frames in enumerate([512, 512, 512, 777, 512, 512, 512, 512, 512, 512, 512, 512, 100, 512, 64, 9, 512]):
        assert p.block(frames, interp, f"block {k} ({frames} frames)") == 1
    p.close()


def test_short_motion_list_in_place_probed(dev):
    """A 2^19-voice bank under sparse note traffic: the listed voices stay in their lanes (sk_gain_kernel + the steady kernel's
    in-place instantiation).  Probes on re-triggered voices and on their neighbours, through attack, decay and into sustain."""
    import torch
    n, F = 1 << 19, 512
    bank, tables, g = banks.bank_c2(n)
    e = bank["voice_amp_envelope"]
    e["sample_start"][:] = np.uint64(g.synth_sample_count - 30000)     # everyone in sustain: the list starts empty
    e["sample_release"][:] = 0
    e["is_active"][:] = 1
    rng = np.random.default_rng(9)
    notes = np.sort(rng.choice(n, 40, replace=False)).astype(np.int32)
    ids = np.unique(np.concatenate([notes[:24], notes[:24] + 1, _pick(n, 16)]))[:64].astype(np.int32)
    p = Probed(dev, bank, tables, g, ids, F)
    p.db.in_place(2)
    in_place = []
    for k in range(14):
        if k in (2, 5):
            host = bank.copy()
            now = p.db.get_globals().synth_sample_count
            p.db.update(host, notes, dev.STAMP_TRIGGER)                 # note-ons stamped on the device with the bank's clock

            def trig(ref, now=now):
                hit = np.isin(ids, notes)
                re = ref["voice_amp_envelope"]
                re["sample_start"][hit] = now
                re["sample_release"][hit] = 0
                re["is_active"][hit] = 1
            p.update_ref(trig)
        assert p.block(F, 0, f"block {k}") == 3
        in_place.append(p.db.last_in_place())
    assert any(in_place), in_place
    assert p.db.list_violations() == 0
    p.close()


def test_probe_refuses_kernels_without_probe_instantiations(dev):
    n = 4096
    bank, tables, g = banks.bank_c2(n)
    bank["voice_cz_mode"][10] = 1                      # CZ: the modulated kernel
    import torch
    p = Probed(dev, bank, tables, g, _pick(n, 8), 64)
    out = torch.zeros(64, 2, device="cuda")
    with pytest.raises(dev.SkredAmdError):
        p.db.render_mix(64, out.data_ptr(), 2, 0, 0)
    p.close()
