"""After note-offs: a quarter of a bank's voices released and their releases run out -- do the waves come back to the steady
blocks?  Wall clock per 512-frame block before the note-offs, and 40 blocks after them."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from skred_amd import banks, device
D = device
F = 512
def t100(db, out):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): db.render_mix(F, out.data_ptr(), 2, 0, 0)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / 100 * 1e3
for rec, n in (("c2", 65536), ("c1", 4096), ("c2", 1 << 20)):
    bank, t, g = banks.RECIPES[rec](n)
    out = torch.zeros(F, 2, device="cuda")
    db = D.DeviceBank(n); db.set_tables(t); db.upload(bank); db.set_globals(g); db.kernel_timing(0)
    for _ in range(40): db.render_mix(F, out.data_ptr(), 2, 0, 0)
    before = t100(db, out)
    vs = np.arange(0, n, 4, dtype=np.int32)
    db.update(bank, vs, D.STAMP_RELEASE, 0)
    for _ in range(40): db.render_mix(F, out.data_ptr(), 2, 0, 0)        # 0.2 s release = 19 blocks, then the smoother tail
    after = t100(db, out)
    print(f"{rec} {n:8d}  all held {before*1e3:7.1f} us/block   a quarter released and run out {after*1e3:7.1f} us/block  (kernel {db.last_kernel()})", flush=True)
    db.close()
