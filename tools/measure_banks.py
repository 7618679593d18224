#!/usr/bin/env python3
"""tools/measure_banks.py SCENARIO... -- the one-off measurements quoted in DESIGN.md, reproducible (run on the GPU box).

  kernels    the users of the one-voice kernel at BASELINE sizes (C1/C2, 2^18 / 2^20 banks forced onto it, FM, one-shots)
  crossover  one voice per lane vs two per lane, 32 768 .. 262 144 voices  (SK_FAST2_MIN_VOICES)
  overhead   what the event pairs of the sampled kernel timing cost a small bank per block
  frames     kernel time vs frames per launch on a 4096-voice bank (per-frame cost of a lone wavefront + fixed cost)
  fm         2^20-voice two-operator FM banks (carrier v, modulator v+1)
  noise      2^20-voice banks with w6 voices, specialised vs generic kernel
  live       notes starting / ending every block on a 2^20-voice bank (cost of control)
  patches    2^20-voice banks made by tiling a reference patch (banks.bank_patch: the routings of 3.sk, 37.sk, 1.sk, 7.sk, 18.sk)
  linear     C1 / C2 / C3 with linear interpolation, pools with and without guard samples
  mid        mid-size enveloped banks: which kernel family renders them, steady

Each line: ms per block over the timed blocks (wall clock), voice-samples/s, and the render kernel's duration from the
library's own event pair around the latest bracketed launch (a bracketed launch runs alone).  kernels / fm / noise print
every bank once (round 1 printed two forms, with and without SKRED_OPT_OVERLAP_TAIL; a block is one launch now).
"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from skred_amd import banks, device  # noqa: E402


def run(name, bank, tables, g, interp=0, F=512, steps=60, min2=None, generic=False, overlap=None, timing=4):
    n = bank.n
    out = torch.zeros(F, 2, device="cuda")
    db = device.DeviceBank(n)
    db.set_tables(tables)
    db.upload(bank)
    db.set_globals(g)
    if min2 is not None:
        db.fast2_min_voices(min2)
    db.force_generic(generic)
    db.kernel_timing(timing)
    for _ in range(25):
        db.render_mix(F, out.data_ptr(), 2, 0, interp)
    torch.cuda.synchronize()
    # spin-up: a GPU taken from idle needs tens of milliseconds of work before its clocks settle (bench.py: --spinup-ms);
    # then the best of three repetitions (short banks finish a repetition in a few milliseconds)
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.04:
        for _ in range(8):
            db.render_mix(F, out.data_ptr(), 2, 0, interp)
        torch.cuda.synchronize()
    best = None
    for _rep in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            db.render_mix(F, out.data_ptr(), 2, 0, interp)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        d_ = (time.perf_counter() - t0) / steps
        if best is None or d_ < best[0]:
            best = (d_, t0, t1)
    dt, t0, t1 = best
    k = f"{db.last_render_ms():.4f}" if timing else "-"
    print(f"{name:66s} kernel={db.last_kernel()} {dt * 1e3:.4f} ms/block {n * F / dt:.3e} voice-samples/s  "
          f"render kernel {k} ms  host issue {(t1 - t0) / steps * 1e6:.1f} us")
    db.close()


def kernels():
    b, t, g = banks.bank_c1(4096); run("c1 4096", b, t, g)
    b, t, g = banks.bank_c1(65536); run("c1 65536", b, t, g)
    b, t, g = banks.bank_c2(65536); run("c2 65536", b, t, g)
    b, t, g = banks.bank_c2(1 << 18); run("c2 2^18 two per lane", b, t, g, min2=1)
    b, t, g = banks.bank_c2(1 << 18); run("c2 2^18 one per lane", b, t, g, min2=1 << 30)
    b, t, g = banks.bank_c2(1 << 20); run("c2 2^20 one per lane", b, t, g, min2=1 << 30)
    b, t, g = banks.bank_c2(1 << 20)
    car = np.arange(0, 1 << 20, 8); b["voice_freq_mod_osc"][car] = car + 3; b["voice_freq_mod_depth"][car] = 0.2
    run("c2 2^20 FM (1/8 carriers)", b, t, g)
    b, t, g = banks.bank_c2(1 << 20)
    b["voice_one_shot"][::3] = 1; b["voice_loop_enabled"][::3] = 0
    run("c2 2^20 one-shots (1/3, finished after warm-up)", b, t, g)
    b, t, g = banks.bank_c2(1 << 20); run("c3 2^20 two per lane", b, t, g)
    b, t, g = banks.bank_c2(1 << 17)
    run("c3 131072 one per lane: the 8-GPU strong-scaling shard, event pairs on every 4th launch", b, t, g, min2=1 << 30, steps=200)
    run("c3 131072 one per lane: the 8-GPU strong-scaling shard, no event pairs", b, t, g, min2=1 << 30, steps=200, timing=0)
    b, t, g = banks.bank_c4(262144); run("c4 262144 linear", b, t, g, interp=1)
    b, t, g = banks.bank_c4(262144)
    b["voice_one_shot"][::3] = 1; b["voice_loop_enabled"][::3] = 0
    run("c4 262144 linear, one-shots (1/3, finished)", b, t, g, interp=1)
    b, t, g = banks.bank_c4(262144)
    b["voice_sample_hold_max"][::16] = 4
    run("c4 262144 linear, sample & hold on 1/16", b, t, g, interp=1)


def crossover():
    for rec in ("c1", "c2"):
        for n in (32768, 65536, 131072, 196608, 262144):
            b, t, g = banks.RECIPES[rec](n)
            run(f"{rec} {n} one per lane", b, t, g, min2=1 << 30, overlap=True)
            run(f"{rec} {n} two per lane", b, t, g, min2=1, overlap=True)


def overhead():
    for n in (4096, 65536):
        b, t, g = banks.bank_c1(n)
        run(f"c1 {n} an event pair on every 4th launch", b, t, g, steps=200)
        run(f"c1 {n} no event pairs", b, t, g, steps=200, timing=0)


def frames():
    for rec in ("c1", "c2"):
        for F in (64, 256, 512, 2048):
            b, t, g = banks.RECIPES[rec](4096)
            run(f"{rec} 4096 F={F}", b, t, g, F=F, steps=100, timing=1, overlap=True)


def fm():
    for rec in ("c1", "c2"):
        for mute in (False, True):
            b, t, g = banks.RECIPES[rec](1 << 20)
            car = np.arange(0, 1 << 20, 2)
            b["voice_freq_mod_osc"][car] = car + 1
            b["voice_freq_mod_depth"][car] = 0.2
            if mute:
                b["voice_disconnect"][car + 1] = 1
            run(f"{rec} 2^20 two-operator FM" + (", modulators muted (m1)" if mute else ""), b, t, g, steps=40)
    b, t, g = banks.bank_c2(1 << 20)
    car = np.arange(0, 1 << 20, 2)
    b["voice_pan_mod_osc"][car] = car + 1
    b["voice_pan_mod_depth"][car] = 0.5
    b["voice_disconnect"][car + 1] = 1
    run("c2 2^20 pan modulated by the next voice (P1 / m1)", b, t, g, steps=40)
    b["voice_freq_mod_osc"][car] = car + 1
    b["voice_freq_mod_depth"][car] = 0.2
    run("c2 2^20 two-operator FM + pan modulation (F1 P1 / m1)", b, t, g, steps=40)


def noise():
    n = 1 << 20
    for frac, label in ((0.05, "5% noise voices, scattered"), (0.0, "noise voices in the last 1/16 of the bank")):
        b, t, g = banks.bank_c2(n)
        v = np.arange(n)
        if frac:
            b["voice_wave_table_index"][(v * 2654435761 % 1000) < frac * 1000] = 6
        else:
            b["voice_wave_table_index"][n - n // 16:] = 6
        run("c2 2^20 " + label, b, t, g, steps=30)
        run("c2 2^20 " + label + " (generic kernel)", b, t, g, steps=30, generic=True)


def patches():
    for p in ("3sk", "37sk", "1sk", "7sk", "18sk"):
        b, t, g = banks.bank_patch(p, 1 << 20)
        run(f"patch {p} tiled over 2^20 voices", b, t, g, steps=20)


def linear():
    for rec, n in (("c1", 4096), ("c2", 65536), ("c2", 1 << 20)):
        b, t, g = banks.RECIPES[rec](n)
        run(f"{rec} {n} truncating lookup", b, t, g, interp=0)
        run(f"{rec} {n} linear, guarded pool (INTERP 2)", b, t, g, interp=1)
        tn = t.copy()
        pos = np.unique(b["voice_table_offset"].astype(np.int64) + b["voice_table_size"].astype(np.int64))
        tn[pos[pos < len(tn)]] = 7.0
        run(f"{rec} {n} linear, pool without guard samples (general form)", b, tn, g, interp=1)


def mid():
    for n in (196608, 229376, 262144, 294912, 327680, 360448, 393216, 458752, 524288):
        b, t, g = banks.bank_c2(n)
        run(f"c2 {n} library's choice", b, t, g, steps=100)
        run(f"c2 {n} one per lane", b, t, g, min2=1 << 30, steps=100)
        run(f"c2 {n} two per lane", b, t, g, min2=1, steps=100)


def live():
    D = device
    n, F = 1 << 20, 512
    out = torch.zeros(F, 2, device="cuda")
    bank, tables, g = banks.bank_c2(n)
    db = device.DeviceBank(n)
    db.set_tables(tables)
    db.upload(bank)
    db.set_globals(g)
    db.kernel_timing(0)
    for _ in range(30):
        db.render_mix(F, out.data_ptr(), 2)
    rng = np.random.default_rng(1)
    for frac in (0.0, 0.0001, 0.0005, 0.005, 0.02):
        k = int(n * frac)
        res = {}
        for mode in ("stream", "sync"):          # blocks queued back to back (what bench.py's live_control times) / a host that waits for every block
            for _rep in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(60):
                    if k:
                        vs = rng.choice(n, k, replace=False).astype(np.int32)
                        db.update(bank, vs[:k // 2], D.STAMP_RELEASE)
                        db.update(bank, vs[k // 2:], D.STAMP_TRIGGER | D.DIRTY_PHASE | D.DIRTY_PARAMS)
                    db.render_mix(F, out.data_ptr(), 2)
                    if mode == "sync":
                        torch.cuda.synchronize()
                torch.cuda.synchronize()
                res[mode] = (time.perf_counter() - t0) / 60
        print(f"{frac * 100:.2f} % of the voices get a note-off / note-on per block: {res['stream'] * 1e3:.3f} ms/block queued back to back "
              f"({n * F / res['stream']:.3e} voice-samples/s), {res['sync'] * 1e3:.3f} ms/block when the host waits for every block"
              f"   [motion list {'in place' if db.last_in_place() else 'by the envelope kernel' if k else '-'}]")
    db.close()


SCENARIOS = {"kernels": kernels, "crossover": crossover, "overhead": overhead, "frames": frames, "fm": fm,
             "noise": noise, "live": live, "patches": patches, "linear": linear, "mid": mid}

if __name__ == "__main__":
    names = sys.argv[1:] or list(SCENARIOS)
    for nm in names:
        if nm not in SCENARIOS:
            sys.exit(f"unknown scenario {nm!r}; one of: {', '.join(SCENARIOS)}")
    for nm in names:
        print(f"== {nm}")
        SCENARIOS[nm]()
