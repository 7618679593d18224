"""One voice per lane against two voices per lane at mid sizes (where the crossover SK_FAST2_MIN_VOICES sits)."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from skred_amd import banks, device
def run(rec, n, min2, F=512, steps=100):
    b, t, g = banks.RECIPES[rec](n)
    out = torch.zeros(F, 2, device="cuda")
    db = device.DeviceBank(n); db.set_tables(t); db.upload(b); db.set_globals(g)
    db.fast2_min_voices(min2); db.kernel_timing(0)
    for _ in range(30): db.render_mix(F, out.data_ptr(), 2, 0, 0)
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(steps): db.render_mix(F, out.data_ptr(), 2, 0, 0)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / steps * 1e3)
    k = db.last_kernel(); db.close()
    return min(res), k
for rec in ("c1", "c3"):
    for n in (32768, 65536, 98304, 131072, 163840, 196608, 262144, 524288):
        a, ka = run(rec, n, 1 << 30)
        b, kb = run(rec, n, 1)
        print(f"{rec} {n:7d}  one per lane (kernel {ka}) {a*1e3:7.1f} us   two per lane (kernel {kb}) {b*1e3:7.1f} us", flush=True)
