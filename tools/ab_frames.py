"""Fixed cost vs per-frame cost of a block: the same bank at several block lengths (wall clock per block, asynchronous
launches back to back).  SKRED_AMD_LIB selects the library build."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from skred_amd import banks, device
def run(name, rec, n, F, min2=None, steps=200):
    b, t, g = banks.RECIPES[rec](n)
    out = torch.zeros(F, 2, device="cuda")
    db = device.DeviceBank(n); db.set_tables(t); db.upload(b); db.set_globals(g)
    if min2 is not None: db.fast2_min_voices(min2)
    db.kernel_timing(0)
    for _ in range(max(30, 6000 // F)): db.render_mix(F, out.data_ptr(), 2, 0, 0)
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(steps): db.render_mix(F, out.data_ptr(), 2, 0, 0)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / steps * 1e3)
    db.close()
    return min(res)
for name, rec, n, min2 in (("c3 131072 one (shard)", "c3", 131072, 1 << 30), ("c2 65536", "c2", 65536, None), ("c1 4096", "c1", 4096, None),
                           ("c3 1048576 two", "c3", 1 << 20, None)):
    ts = {F: run(name, rec, n, F, min2, steps=200 if F <= 512 else 60) for F in (64, 128, 256, 512, 1024, 2048)}
    per = (ts[2048] - ts[512]) / 1536
    print(f"{name:26s} " + " ".join(f"F={F}:{ts[F]*1e3:7.1f}us" for F in ts) + f"  per frame {per*1e3:.4f} us, fixed at F=512 {1e3*(ts[512]-512*per):.1f} us", flush=True)
