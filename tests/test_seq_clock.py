"""The pattern step clock (skred_seq_t, skred_amd/csrc/skred_seq.c == seq.c:179-213): on the CPU against the compiled
reference's seq(), block by block; on the GPU a 16-step pattern driving device-resident voices for 64 blocks against the
oracle applying the same stores."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from seq_script import script
from skred_amd import banks, device

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_step_times_equal_the_reference_seq(seed):
    """Same edits, same blocks (512-frame callbacks and ragged ones, tempo changes, pause / resume / stop, mutes, modulo
    0..7, holes, a pattern reset mid-run): the steps that fire and every pointer / counter, call by call."""
    z = np.load(os.path.join(HERE, "golden", "seq_clock.npz"))      # written by `seq_replay.py --write-golden`
    nblk = len(z[f"s{seed}_pointer"])
    fired_at = [[] for _ in range(nblk)]
    for k, p, s in z[f"s{seed}_fired"].tolist():
        fired_at[k].append([p, s])
    ref = [{"fired": fired_at[k], "pointer": z[f"s{seed}_pointer"][k].tolist(), "counter": z[f"s{seed}_counter"][k].tolist()}
           for k in range(nblk)]
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libskred_ref.so")):   # and the live reference, when built
        out = subprocess.run([sys.executable, os.path.join(HERE, "seq_replay.py"), str(seed)], capture_output=True, text=True, timeout=120)
        assert out.returncode == 0, out.stderr[-1500:]
        assert json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1][7:]) == ref, "fixture is stale"
    sq = device.SeqClock()
    k = 0
    for op in script(seed):
        if op[0] == "tempo":
            sq.tempo(op[1])
        elif op[0] == "step":
            sq.step(op[1], op[2], bool(op[3]))
        elif op[0] == "mute":
            sq.mute(op[1], op[2], bool(op[3]))
        elif op[0] == "modulo":
            sq.modulo(op[1], op[2])
        elif op[0] == "state":
            sq.state(op[1], op[2])
        elif op[0] == "reset":
            sq.reset(op[1])
        elif op[0] == "block":
            fired = sq.tick(op[1], 44100.0)
            assert [list(f) for f in fired] == ref[k]["fired"], f"block {k}"
            assert [sq.pointer(p) for p in range(16)] == ref[k]["pointer"], f"block {k}"
            assert [sq.counter(p) for p in range(16)] == ref[k]["counter"], f"block {k}"
            k += 1
    assert k == len(ref) and sum(len(r["fired"]) for r in ref) > 100
    sq.close()


def test_default_tempo_and_bounds():
    sq = device.SeqClock()
    assert sq.time_per_step() == 60.0                       # skred.c:47 until a tempo is set
    sq.tempo(120.0)
    assert sq.time_per_step() == np.float32(1.0) / np.float32(2.0) / np.float32(4.0)
    with pytest.raises(device.SkredAmdError):
        sq.step(16, 0)
    with pytest.raises(device.SkredAmdError):
        sq.mute(0, 256)
    sq.close()


@pytest.mark.gpu
def test_pattern_drives_device_resident_voices():
    """A 16-step pattern (note-ons with new pitches, note-offs, a rest, a muted step) and a second, slower one play for 64
    blocks of 512 frames on a bank of 5000 voices with no download and no re-upload in between; the oracle applies the
    same stores when the same clock fires.  State bit-exact at the end, every block's mix within tolerance."""
    import torch
    from oracle import cpuref
    D = device
    n, F, rate = 5000, 512, 48000.0
    bank, tables, g = banks.bank_c2(n)
    host = bank.copy()
    db = D.DeviceBank(n)
    db.set_tables(tables)
    db.upload(host)
    db.set_globals(g)
    db.set_sample_rate(rate)
    sq = db.seq()
    sq.tempo(1400.0)                                         # a step every 10.7 ms = one per 512-frame block, near enough
    ref_sq = D.SeqClock()
    ref_sq.tempo(1400.0)
    rng = np.random.default_rng(5)
    steps = {}                                               # (pattern, step) -> (voices, kind, new phase_inc)
    for p, length, mod in ((0, 16, 1), (1, 5, 3)):
        sq.modulo(p, mod); ref_sq.modulo(p, mod)
        for s in range(length):
            vs = rng.choice(n, 40, replace=False).astype(np.int32)
            if p == 0 and s == 6:
                db.pattern_step_set(p, s)                    # a rest
                steps[(p, s)] = None
            elif s % 3 == 2:
                db.pattern_step_set(p, s, host, vs, D.STAMP_RELEASE)
                steps[(p, s)] = (vs, "off", None)
            else:
                inc = host["voice_phase_inc"].copy()
                inc[vs] = (inc[vs] * np.float32(1.0 + 0.05 * (s + 1))).astype(np.float32)
                edited = host.copy()
                edited["voice_phase_inc"][:] = inc
                db.pattern_step_set(p, s, edited, vs, D.STAMP_TRIGGER | D.DIRTY_PARAMS)
                steps[(p, s)] = (vs, "on", inc[vs].copy())
            ref_sq.step(p, s, True)
        sq.state(p, 1); ref_sq.state(p, 1)
    sq.mute(0, 9, True); ref_sq.mute(0, 9, True)
    ref_host, ref_g = bank.copy(), g.copy()
    out = torch.zeros(F, 2, device="cuda")
    fired_total = 0
    for k in range(64):
        db.render_mix(F, out.data_ptr(), 2, 0, 0)
        torch.cuda.synchronize()
        r = cpuref.render(ref_host, ref_g, tables, F, 0)
        ref_mix = cpuref.master(ref_g, r["sum64"].astype(np.float32))
        d = out.cpu().numpy().astype(np.float64) - ref_mix
        assert np.sqrt((d ** 2).mean()) / max(np.sqrt((ref_mix.astype(np.float64) ** 2).mean()), 1e-30) <= 1e-5, f"block {k}"
        applied = db.run_queue(F)                            # seq() after synth(), skred.c:119
        fired = ref_sq.tick(F, rate)
        assert applied == len(fired)
        fired_total += len(fired)
        now = ref_g.synth_sample_count
        e = ref_host["voice_amp_envelope"]
        for key in fired:
            st = steps[key]
            if st is None:
                continue
            vs, kind, inc = st
            if kind == "off":                                # amp_envelope_release, synth.c:391-395
                live = e["is_active"][vs] != 0
                e["sample_release"][vs[live]] = now
            else:                                            # new pitch + amp_envelope_trigger, synth.c:383-388
                ref_host["voice_phase_inc"][vs] = inc
                e["sample_start"][vs] = now
                e["sample_release"][vs] = 0
                e["is_active"][vs] = 1
    assert fired_total >= 60 and sq.pointer(0) == ref_sq.pointer(0) and sq.counter(1) == ref_sq.counter(1)
    got = bank.copy()
    db.download(got)
    db.close()
    ref_sq.close()
    bad = got.rw_equal(ref_host)
    assert not bad, bad
