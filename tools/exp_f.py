import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from skred_amd import banks, device
def run(rec, n, F, steps=100):
    bank, tables, g = banks.RECIPES[rec](n)
    out = torch.zeros(F, 2, device='cuda')
    db = device.DeviceBank(n); db.set_tables(tables); db.upload(bank); db.set_globals(g)
    db.overlap_tail(True); db.kernel_timing(1)
    ks = []
    for i in range(steps):
        db.render_mix(F, out.data_ptr(), 2, 0, 0)
        if i >= 20: db.wait_mix(0); ks.append(db.last_render_ms())
    print(f"{rec} {n} F={F}: kernel mean {np.mean(ks)*1e3:.1f} us  min {np.min(ks)*1e3:.1f} us  -> {np.min(ks)*1e6/F:.1f} ns/frame")
    del db
for rec in ("c1", "c2"):
    for F in (64, 256, 512, 2048):
        run(rec, 4096, F)
