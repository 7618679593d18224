import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from skred_amd import banks, device
def run(name, bank, tables, g, interp=0, F=512, steps=30, generic=False):
    n = bank.n
    out = torch.zeros(F, 2, device='cuda')
    db = device.DeviceBank(n); db.set_tables(tables); db.upload(bank); db.set_globals(g)
    db.force_generic(generic)
    db.overlap_tail(True); db.kernel_timing(4)
    for _ in range(25): db.render_mix(F, out.data_ptr(), 2, 0, interp)
    db.wait_mix(0); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): db.render_mix(F, out.data_ptr(), 2, 0, interp)
    db.wait_mix(0); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f"{name:44s} kernel={db.last_kernel()} {dt*1e3:.4f} ms/block {n*F/dt:.3e} vs/s  last_kernel_ms={db.last_render_ms():.4f}")
    del db
n = 1 << 20
for frac, label in ((0.05, "5% noise voices, scattered"), (0.0, "noise voices in the last 1/16 of the bank")):
    b, t, g = banks.bank_c2(n)
    v = np.arange(n)
    if frac: b["voice_wave_table_index"][(v * 2654435761 % 1000) < frac * 1000] = 6
    else: b["voice_wave_table_index"][n - n // 16:] = 6
    run("c2 2^20 " + label + " (specialised)", b, t, g)
    run("c2 2^20 " + label + " (generic)", b, t, g, generic=True)
