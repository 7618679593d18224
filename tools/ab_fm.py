"""Two-operator FM banks: carrier and modulator in one lane (two-per-lane kernel, SKM_FM_PAIR) against the one-per-lane
kernel's exchange, by bank size."""
import sys, time
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from skred_amd import banks, device
def run(rec, n, fm2, F=512, steps=60):
    b, t, g = banks.RECIPES[rec](n)
    car = np.arange(0, n, 2)
    b["voice_freq_mod_osc"][car] = car + 1
    b["voice_freq_mod_depth"][car] = 0.2
    out = torch.zeros(F, 2, device="cuda")
    db = device.DeviceBank(n); db.set_tables(t); db.upload(b); db.set_globals(g)
    db.fm2_min_voices(fm2); db.kernel_timing(0)
    for _ in range(20): db.render_mix(F, out.data_ptr(), 2, 0, 0)
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(steps): db.render_mix(F, out.data_ptr(), 2, 0, 0)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / steps * 1e3)
    k = db.last_kernel(); db.close()
    return min(res), k
for rec in ("c1", "c2"):
    for n in (4096, 16384, 32768, 65536, 131072, 262144, 1 << 20):
        a, ka = run(rec, n, 1 << 30)
        b, kb = run(rec, n, 0)
        print(f"{rec} {n:8d}  exchange (kernel {ka}) {a*1e3:8.1f} us   one lane (kernel {kb}) {b*1e3:8.1f} us", flush=True)
