"""Note traffic on a mid-size bank (one-voice kernel): every block 0.1 % of the voices get a note-off or a note-on."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from skred_amd import banks, device
D = device
F = 512
for rec, n in (("c2", 65536), ("c2", 131072), ("c1", 4096)):
    bank, t, g = banks.RECIPES[rec](n)
    out = torch.zeros(F, 2, device="cuda")
    db = D.DeviceBank(n); db.set_tables(t); db.upload(bank); db.set_globals(g); db.kernel_timing(0)
    for _ in range(40): db.render_mix(F, out.data_ptr(), 2, 0, 0)
    torch.cuda.synchronize()
    rng = np.random.default_rng(1)
    k_ev = max(2, n // 1000)
    def blk():
        vs = rng.choice(n, k_ev, replace=False).astype(np.int32)
        db.update(bank, vs[:k_ev // 2], D.STAMP_RELEASE, 0)
        db.update(bank, vs[k_ev // 2:], D.STAMP_TRIGGER | D.DIRTY_PHASE | D.DIRTY_PARAMS, 0)
        db.render_mix(F, out.data_ptr(), 2, 0, 0)
    for _ in range(40): blk()
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(100): blk()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 100 * 1e3)
    print(f"{rec} {n:7d}  {k_ev} note events per block: {min(res)*1e3:7.1f} us/block (kernel {db.last_kernel()})", flush=True)
    db.close()
