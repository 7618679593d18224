#!/usr/bin/env python3
"""tools/ab.py SCENARIO [options] -- one parametrised A/B measuring script (run on the GPU box; `SKRED_AMD_LIB` picks the library
build, `tools/ab_libs.sh ab.py ...` runs it once per build under _ab/).  It replaces the one-off tools/ab_*.py scripts of rounds
1-3 (their results live in profiles/ and DESIGN_HISTORY.md).  Wall clock per block over queued blocks, best and median of 5.

  steady   sustained banks of one recipe at several sizes:            --sizes c1:4096,c2:65536,...  [--split N] [--one-voice] [--interp I] [--f F]
  frames   one bank, several block lengths (per-frame + fixed cost):  --bank c2:65536 --lengths 64,256,512,2048
  live     a 2^20-voice C3 bank from its first frame, then under note traffic:  [--notes 104,524,5242] [--in-place M] [--voices N]
  patch    reference patches tiled over a bank (banks.bank_patch):    [--patches 3sk,37sk,7sk,1sk,18sk] [--voices N] [--fm-skew 0|1]
  fm       the C2 recipe with every eighth voice frequency-modulated (modulators muted / heard):                 [--voices N] [--fm-skew 0|1]
  fx       the fixed-point bank (fxbank.bank_fx) at several sizes, both lookups, with and without the biquad:  [--fx-sizes 65536,1048576]
  stamps   `steady` on a -DSKS_STAMPS build of the split kernel: what its waves recorded (cycles, waits, in-kernel clock)
"""
import argparse, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from skred_amd import banks, device

LIB = os.path.basename(os.path.dirname(os.environ.get("SKRED_AMD_LIB", "default/x")))


def timed(block, steps, reps=5):
    res = []
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(steps):
            block()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / steps * 1e6)
    return min(res), sorted(res)[len(res) // 2]


def open_bank(rec, n, a, bank=None):
    b, t, g = bank if bank else banks.RECIPES[rec](n)
    if a.sparse > 0:   # this fraction of the voices cannot sound (voice_amp 0): SKRED_OPT_PACK's case
        import numpy as np
        amp = np.asarray(b["voice_amp"]).copy()
        amp[np.random.default_rng(1).random(b.n) < a.sparse] = 0.0
        b["voice_amp"] = amp
    if a.mixed:        # half of the voices filtered: SKM_MIXED, the extended instantiation of the family
        import numpy as np
        mode = np.asarray(b["voice_filter_mode"]).copy()
        if mode.any(): mode[1::2] = 0
        else:
            mode[0::2] = 1
            co = banks.biquad_coeffs(mode, np.full(b.n, 1000.0, np.float32), np.full(b.n, 1.0, np.float32), 48000)
            flt = b["voice_filter"]
            for k, v in co.items(): flt[k] = v
            flt["last_freq"], flt["last_resonance"], flt["last_mode"] = 1000.0, 1.0, mode
        b["voice_filter_mode"] = mode
    db = device.DeviceBank(b.n); db.set_tables(t); db.upload(b); db.set_globals(g); db.kernel_timing(0)
    if a.one_voice: db.fast2_min_voices(1 << 30)
    if a.split is not None: db.set_split(a.split)
    if a.in_place is not None: db.in_place(a.in_place)
    if a.pack is not None: db.set_pack(a.pack)
    if a.fm_skew is not None: db.set_fm_skew(a.fm_skew)
    return db, b


def settle(db, out, F, interp, blocks=40):
    for _ in range(blocks): db.render_mix(F, out.data_ptr(), 2, 0, interp)          # the recipe's warm-up: every note into its sustain stage
    torch.cuda.synchronize()
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.05:                                        # clocks settle
        for _ in range(8): db.render_mix(F, out.data_ptr(), 2, 0, interp)
        torch.cuda.synchronize()


def steady(a):
    for item in a.sizes.split(","):
        rec, n = item.split(":"); n = int(n)
        out = torch.zeros(a.f, 2, device="cuda")
        db, _ = open_bank(rec, n, a)
        settle(db, out, a.f, a.interp)
        best, med = timed(lambda: db.render_mix(a.f, out.data_ptr(), 2, 0, a.interp), a.steps)
        print(f"steady {rec} {n:8d} voices F={a.f} interp={a.interp} kernel={db.last_kernel()} split={int(db.last_split())} pack={db.last_pack()}: {best:8.2f} us (med {med:8.2f})  lib={LIB}", flush=True)
        if a.scenario == "stamps" and db.last_split():
            import ctypes as C
            n_wg = (n + 1023) // 1024 * 4
            buf = np.zeros(n_wg * 64, np.int32)
            db.L.sk_debug_env_list.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
            db.L.sk_debug_env_list(db.h, buf.ctypes.data, buf.size)
            w = buf.reshape(n_wg, 8, 8).astype(np.float64)
            for name, sl in (("post", slice(0, 4)), ("osc ", slice(4, 8))):
                x = w[:, sl, :].reshape(-1, 8); x = x[x[:, 0] > 0]
                if len(x):
                    print(f"   {name}: {len(x)} waves, loop {x[:, 0].mean() / a.f:6.1f} cycles per frame, waiting {100 * x[:, 1].sum() / x[:, 0].sum():4.1f} % in {x[:, 2].mean():5.1f} waits, "
                          f"chunk ends {100 * x[:, 3].sum() / x[:, 0].sum():4.1f} %, in-kernel clock {x[:, 0].sum() / max(x[:, 4].sum(), 1) * 100:5.0f} MHz", flush=True)
        db.close()


def frames(a):
    rec, n = a.bank.split(":"); n = int(n)
    db, _ = open_bank(rec, n, a)
    pts = []
    for F in [int(x) for x in a.lengths.split(",")]:
        out = torch.zeros(F, 2, device="cuda")
        settle(db, out, F, a.interp, blocks=max(8, 20000 // F))
        best, med = timed(lambda: db.render_mix(F, out.data_ptr(), 2, 0, a.interp), max(20, a.steps * 512 // max(F, 64)))
        pts.append((F, best))
        print(f"frames {rec} {n} voices F={F:5d}: {best:8.2f} us (med {med:8.2f})  lib={LIB}", flush=True)
    if len(pts) >= 2:
        (f0, t0), (f1, t1) = pts[0], pts[-1]
        per = (t1 - t0) / (f1 - f0)
        print(f"   -> {per * 1e3:.1f} ns per frame + {t0 - per * f0:.1f} us fixed", flush=True)
    db.close()


def live(a):
    n, F = a.voices, 512
    db, bank = open_bank("c3", n, a)
    out = torch.zeros(F, 2, device="cuda")
    bank0, tables, g = banks.RECIPES["c3"](n)
    reps = []
    for _ in range(4):
        db.upload(bank0); db.set_globals(g); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(11): db.render_mix(F, out.data_ptr(), 2, 0, 0)
        torch.cuda.synchronize()
        reps.append((time.perf_counter() - t0) / 11 * 1e6)
    print(f"live envelopes_in_motion: {min(reps):8.2f} us per block (med {sorted(reps)[2]:8.2f})  lib={LIB}", flush=True)
    for _ in range(12): db.render_mix(F, out.data_ptr(), 2, 0, 0)
    rng = np.random.default_rng(1)
    for k_ev in [int(x) for x in a.notes.split(",")]:
        def blk():
            vs = rng.choice(n, k_ev, replace=False).astype(np.int32)
            db.update(bank, vs[:k_ev // 2], device.STAMP_RELEASE, 0)
            db.update(bank, vs[k_ev // 2:], device.STAMP_TRIGGER | device.DIRTY_PHASE | device.DIRTY_PARAMS, 0)
            db.render_mix(F, out.data_ptr(), 2, 0, 0)
        for _ in range(20): blk()
        torch.cuda.synchronize()
        best, med = timed(blk, 60, 3)
        print(f"live {k_ev:5d} notes per block: {best:8.2f} us (med {med:8.2f}) in_place={int(db.last_in_place())} violations={db.list_violations()}  lib={LIB}", flush=True)
    db.close()


def patch(a):
    for p in a.patches.split(","):
        bank = banks.bank_patch(p, a.voices)
        db, _ = open_bank(None, 0, a, bank=bank)
        out = torch.zeros(512, 2, device="cuda")
        settle(db, out, 512, 0, blocks=12)
        best, med = timed(lambda: db.render_mix(512, out.data_ptr(), 2, 0, 0), 40, 3)
        print(f"patch {p:5s} tiled over {a.voices} voices: {best / 1e3:7.3f} ms per block (med {med / 1e3:7.3f}) kernel={db.last_kernel()} pack={db.last_pack()} fm_skew={a.fm_skew}  lib={LIB}", flush=True)
        db.close()


def fm(a):
    """The headline recipe (mixed LUTs + biquad + ADSR) with every eighth voice a carrier of the voice three above it -- muted
    (`m1`) or heard --: the extended instantiation WITH biquad and envelope on the skewed blocks (--fm-skew 1) or the exchange (0)."""
    import numpy as np
    for muted in (1, 0):
        b, t, g = banks.bank_c2(a.voices)
        car = np.arange(0, a.voices, 8); b["voice_freq_mod_osc"][car] = car + 3; b["voice_freq_mod_depth"][car] = 0.2
        if muted: b["voice_disconnect"][car + 3] = 1
        db, _ = open_bank(None, 0, a, bank=(b, t, g))
        out = torch.zeros(512, 2, device="cuda")
        settle(db, out, 512, 0, blocks=25)
        best, med = timed(lambda: db.render_mix(512, out.data_ptr(), 2, 0, 0), 40, 3)
        print(f"fm c2 recipe, 1/8 carriers, modulators {'muted' if muted else 'heard'}, {a.voices} voices: {best / 1e3:7.3f} ms per block (med {med / 1e3:7.3f}) kernel={db.last_kernel()} fm_skew={a.fm_skew}  lib={LIB}", flush=True)
        db.close()


def fx(a):
    from skred_amd import fxbank
    for n in [int(x) for x in a.fx_sizes.split(",")]:
        for with_filter in ((True, False) if a.fx_filter < 0 else (bool(a.fx_filter),)):
            b, pool, c0 = fxbank.bank_fx(n, with_filter=with_filter)
            db = fxbank.DeviceFxBank(n); db.set_tables(pool); db.upload(b); db.set_sample_count(c0)
            out = torch.zeros(a.f, 2, device="cuda", dtype=torch.int64)
            for interp in ((1, 0) if a.fx_interp < 0 else (a.fx_interp,)):
                for _ in range(20): db.render_mix(a.f, out.data_ptr(), interp, 0, 0)
                torch.cuda.synchronize()
                best, med = timed(lambda: db.render_mix(a.f, out.data_ptr(), interp, 0, 0), 40, 3)
                print(f"fx {n:8d} voices filter={int(with_filter)} interp={interp} F={a.f}: {best / 1e3:7.4f} ms per block (med {med / 1e3:7.4f})  "
                      f"kernel {db.last_render_ms():.4f} ms  lib={LIB}", flush=True)
            db.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scenario", choices=["steady", "frames", "live", "patch", "stamps", "fx", "fm"])
    ap.add_argument("--fx-sizes", default="65536,1048576")
    ap.add_argument("--fx-filter", type=int, default=-1)
    ap.add_argument("--fx-interp", type=int, default=-1)
    ap.add_argument("--sizes", default="c1:4096,c2:65536,c2:131072,c2:196608")
    ap.add_argument("--bank", default="c2:65536")
    ap.add_argument("--lengths", default="64,256,512,2048")
    ap.add_argument("--f", type=int, default=512)
    ap.add_argument("--interp", type=int, default=0)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--split", type=int, default=None)
    ap.add_argument("--in-place", type=int, default=None)
    ap.add_argument("--one-voice", action="store_true")
    ap.add_argument("--voices", type=int, default=1 << 20)
    ap.add_argument("--notes", default="104,524,5242")
    ap.add_argument("--mixed", action="store_true")
    ap.add_argument("--sparse", type=float, default=0.0)
    ap.add_argument("--pack", type=int, default=None)
    ap.add_argument("--fm-skew", type=int, default=None)
    ap.add_argument("--patches", default="3sk,37sk,7sk,1sk,18sk")
    a = ap.parse_args()
    {"steady": steady, "stamps": steady, "frames": frames, "live": live, "patch": patch, "fx": fx, "fm": fm}[a.scenario](a)


if __name__ == "__main__":
    main()
