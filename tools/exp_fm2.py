import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from skred_amd import banks, device
def run(name, bank, tables, g, interp=0, F=512, steps=40):
    n = bank.n
    out = torch.zeros(F, 2, device='cuda')
    db = device.DeviceBank(n); db.set_tables(tables); db.upload(bank); db.set_globals(g)
    db.overlap_tail(True); db.kernel_timing(4)
    for _ in range(25): db.render_mix(F, out.data_ptr(), 2, 0, interp)
    db.wait_mix(0); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): db.render_mix(F, out.data_ptr(), 2, 0, interp)
    db.wait_mix(0); torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print(f"{name:44s} kernel={db.last_kernel()} {dt*1e3:.4f} ms/block {n*F/dt:.3e} vs/s  last_kernel_ms={db.last_render_ms():.4f}")
    del db
for rec in ("c1", "c2"):
    b, t, g = banks.RECIPES[rec](1 << 20)
    car = np.arange(0, 1 << 20, 2); b["voice_freq_mod_osc"][car] = car + 1; b["voice_freq_mod_depth"][car] = 0.2
    run(f"{rec} 2^20 two-operator FM (carrier v, modulator v+1)", b, t, g)
    b, t, g = banks.RECIPES[rec](1 << 20)
    car = np.arange(0, 1 << 20, 2); b["voice_freq_mod_osc"][car] = car + 1; b["voice_freq_mod_depth"][car] = 0.2
    b["voice_disconnect"][car + 1] = 1
    run(f"{rec} same, modulators muted (m1)", b, t, g)
