import sys, time, os
import numpy as np, torch
sys.path.insert(0, ".")
from skred_amd import banks, device
def run(name, b, t, g, interp=0, F=512, steps=60):
    n = b.n
    out = torch.zeros(F, 2, device="cuda")
    db = device.DeviceBank(n); db.set_tables(t); db.upload(b); db.set_globals(g)
    db.kernel_timing(0)
    for _ in range(30): db.render_mix(F, out.data_ptr(), 2, 0, interp)
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(steps): db.render_mix(F, out.data_ptr(), 2, 0, interp)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / steps * 1e3)
    print(f"{name:52s} kernel={db.last_kernel()} ms/block min {min(res):.4f}  {n*F/min(res)*1e3:.3e} vs/s  lib={os.path.basename(os.environ.get('SKRED_AMD_LIB','default'))}", flush=True)
    db.close()
b, t, g = banks.bank_c2(1 << 20)
car = np.arange(0, 1 << 20, 8); b["voice_freq_mod_osc"][car] = car + 3; b["voice_freq_mod_depth"][car] = 0.2
run("c2 2^20 FM (1/8 carriers)", b, t, g)
for rec in ("c1", "c2"):
    b, t, g = banks.RECIPES[rec](1 << 20)
    car = np.arange(0, 1 << 20, 2)
    b["voice_freq_mod_osc"][car] = car + 1; b["voice_freq_mod_depth"][car] = 0.2; b["voice_disconnect"][car + 1] = 1
    run(f"{rec} 2^20 two-operator FM, modulators muted", b, t, g)
b, t, g = banks.bank_c2(1 << 20)
b["voice_sample_hold_max"][::16] = 4
run("c2 2^20 sample & hold on 1/16", b, t, g)
b, t, g = banks.bank_c4(262144)
run("c4 262144 linear", b, t, g, interp=1)
b["voice_sample_hold_max"][::16] = 4
run("c4 262144 linear, sample & hold on 1/16", b, t, g, interp=1)
b, t, g = banks.bank_c4(262144)
b["voice_one_shot"][::3] = 1; b["voice_loop_enabled"][::3] = 0
run("c4 262144 linear, one-shots (1/3, finished)", b, t, g, interp=1)
