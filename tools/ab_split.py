#!/usr/bin/env python3
"""tools/ab_split.py [--f F] [--sizes c1:4096,c2:65536,...] -- the split form of the one-voice kernel (SKRED_OPT_SPLIT 2) against
sk_render_fast_kernel (0) on the same sustained banks, wall clock per block over queued blocks (run on the GPU box)."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from skred_amd import banks, device


def run(rec, n, split, interp, F, steps):
    b, t, g = banks.RECIPES[rec](n)
    out = torch.zeros(F, 2, device="cuda")
    db = device.DeviceBank(n); db.set_tables(t); db.upload(b); db.set_globals(g)
    db.fast2_min_voices(1 << 30); db.set_split(split); db.kernel_timing(0)
    for _ in range(40): db.render_mix(F, out.data_ptr(), 2, 0, interp)     # the recipe's warm-up: every note into its sustain stage
    torch.cuda.synchronize()
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.05:
        for _ in range(8): db.render_mix(F, out.data_ptr(), 2, 0, interp)
        torch.cuda.synchronize()
    res = []
    for _rep in range(5):
        t0 = time.perf_counter()
        for _ in range(steps): db.render_mix(F, out.data_ptr(), 2, 0, interp)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / steps * 1e6)
    took = db.last_split()
    if STAMPS and split:
        import ctypes as C, numpy as np
        n_wg = (db.n + 1023) // 1024 * 4
        buf = np.zeros(n_wg * 64, np.int32)
        db.L.sk_debug_env_list.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        rc = db.L.sk_debug_env_list(db.h, buf.ctypes.data, buf.size)
        w = buf.reshape(n_wg, 8, 8).astype(np.float64)
        for name, sl in (("post", slice(0, 4)), ("osc ", slice(4, 8))):
            x = w[:, sl, :].reshape(-1, 8)
            x = x[x[:, 0] > 0]
            if len(x):
                clk = x[:, 0].sum() / max(x[:, 4].sum(), 1) * 100.0
                print(f"   stamps {name}: waves {len(x)}  loop cycles {x[:,0].mean():9.0f} ({x[:,0].mean()/F:6.1f} per frame)  waiting {x[:,1].mean():9.0f} ({100*x[:,1].sum()/x[:,0].sum():4.1f} %) in {x[:,2].mean():6.1f} waits"
                      f"  chunk ends {x[:,3].mean():8.0f} ({100*x[:,3].sum()/x[:,0].sum():4.1f} %)  in-kernel clock {clk:6.0f} MHz", flush=True)
    db.close()
    return min(res), sorted(res)[2], took


STAMPS = False


def main():
    global STAMPS
    ap = argparse.ArgumentParser()
    ap.add_argument("--stamps", action="store_true", help="the library is a -DSKS_STAMPS build: print what its waves recorded")
    ap.add_argument("--f", type=int, default=512)
    ap.add_argument("--interp", type=int, default=0)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--sizes", default="c1:4096,c2:16384,c2:65536,c2:98304,c2:131072,c2:196608,c2:262144")
    a = ap.parse_args()
    STAMPS = a.stamps
    for item in a.sizes.split(","):
        rec, n = item.split(":"); n = int(n)
        base = run(rec, n, 0, a.interp, a.f, a.steps)
        spl = run(rec, n, 2, a.interp, a.f, a.steps)
        print(f"{rec} {n:8d} voices F={a.f} interp={a.interp}: unsplit {base[0]:7.2f} us (med {base[1]:7.2f})   split {spl[0]:7.2f} us (med {spl[1]:7.2f}) took={spl[2]}   ratio {spl[0] / base[0]:.3f}", flush=True)


if __name__ == "__main__":
    main()
