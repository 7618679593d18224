"""The two-voices-per-lane kernel at the bank sizes a 2- / 4-GPU strong-scaling shard has (and at full size)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from skred_amd import banks, device
F = 512
for rec, n in (("c3", 262144), ("c3", 524288), ("c3", 1 << 20), ("c1", 262144)):
    b, t, g = banks.RECIPES[rec](n)
    out = torch.zeros(F, 2, device="cuda")
    db = device.DeviceBank(n); db.set_tables(t); db.upload(b); db.set_globals(g); db.kernel_timing(0); db.fast2_min_voices(0)
    for _ in range(30): db.render_mix(F, out.data_ptr(), 2, 0, 0)
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(100): db.render_mix(F, out.data_ptr(), 2, 0, 0)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 100 * 1e3)
    print(f"{rec} {n:8d} two per lane kernel={db.last_kernel()} ms/block min {min(res):.4f} med {sorted(res)[1]:.4f}", flush=True)
    db.close()
