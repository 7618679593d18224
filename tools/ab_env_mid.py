"""Envelopes in motion on banks below the two-per-lane threshold (one-voice kernel): the recipe's first 11 blocks against
its steady blocks, and the same bank forced onto the two-per-lane + envelope kernels."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from skred_amd import banks, device
F = 512
SIZES = [int(x) for x in sys.argv[1:]] or [65536, 131072, 262144, 393216, 524288, 786432]
for rec, n in [("c2", x) for x in SIZES]:
    for min2, label in ((1 << 30, "one per lane"), (0, "two per lane + envelope kernel")):
        bank, t, g = banks.RECIPES[rec](n)
        out = torch.zeros(F, 2, device="cuda")
        db = device.DeviceBank(n); db.set_tables(t); db.upload(bank); db.set_globals(g); db.kernel_timing(0); db.fast2_min_voices(min2)
        for _ in range(20): db.render_mix(F, out.data_ptr(), 2, 0, 0)
        torch.cuda.synchronize()
        reps = []
        for _ in range(4):
            db.upload(bank); db.set_globals(g); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(11): db.render_mix(F, out.data_ptr(), 2, 0, 0)
            torch.cuda.synchronize()
            reps.append((time.perf_counter() - t0) / 11 * 1e3)
        for _ in range(30): db.render_mix(F, out.data_ptr(), 2, 0, 0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100): db.render_mix(F, out.data_ptr(), 2, 0, 0)
        torch.cuda.synchronize()
        st = (time.perf_counter() - t0) / 100 * 1e3
        print(f"{rec} {n:7d} {label:32s} kernel {db.last_kernel()}  in motion {min(reps)*1e3:7.1f} us/block   steady {st*1e3:7.1f} us/block", flush=True)
        db.close()
