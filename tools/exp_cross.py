import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from skred_amd import banks, device
def run(name, bank, tables, g, interp=0, min2=None, F=512, steps=60):
    n = bank.n
    out = torch.zeros(F, 2, device='cuda')
    db = device.DeviceBank(n); db.set_tables(tables); db.upload(bank); db.set_globals(g)
    if min2 is not None: db.fast2_min_voices(min2)
    db.overlap_tail(True); db.kernel_timing(4)
    for _ in range(25): db.render_mix(F, out.data_ptr(), 2, 0, interp)
    db.wait_mix(0); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): db.render_mix(F, out.data_ptr(), 2, 0, interp)
    db.wait_mix(0); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f"{name:34s} kernel={db.last_kernel()} {dt*1e3:.4f} ms/block {n*F/dt:.3e} vs/s  last_kernel_ms={db.last_render_ms():.4f}")
    del db
for rec in ("c1", "c2"):
    for n in (32768, 65536, 131072, 196608, 262144):
        b, t, g = banks.RECIPES[rec](n)
        run(f"{rec} {n} one-voice", b, t, g, min2=1 << 30)
        run(f"{rec} {n} two-per-lane", b, t, g, min2=1)
