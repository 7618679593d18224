import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from skred_amd import banks, device
def run(name, n, overlap, timing, F=512, steps=200):
    bank, tables, g = banks.bank_c1(n)
    out = torch.zeros(F, 2, device='cuda')
    db = device.DeviceBank(n); db.set_tables(tables); db.upload(bank); db.set_globals(g)
    db.overlap_tail(overlap); db.kernel_timing(timing)
    for _ in range(25): db.render_mix(F, out.data_ptr(), 2, 0, 0)
    db.wait_mix(0); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): db.render_mix(F, out.data_ptr(), 2, 0, 0)
    t1 = time.perf_counter()
    db.wait_mix(0); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f"{name:40s} {dt*1e6:.1f} us/block (host issue {(t1-t0)/steps*1e6:.1f} us)")
    del db
for n in (4096, 65536):
    run(f"c1 {n} overlap timing/4", n, True, 4)
    run(f"c1 {n} overlap no timing", n, True, 0)
    run(f"c1 {n} no-overlap no timing", n, False, 0)
