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
b, t, g = banks.bank_c1(4096); run("c1 4096", b, t, g)
b, t, g = banks.bank_c1(65536); run("c1 65536", b, t, g)
b, t, g = banks.bank_c2(65536); run("c2 65536", b, t, g)
b, t, g = banks.bank_c2(1 << 18); run("c2 2^18 fast2", b, t, g)
b, t, g = banks.bank_c2(1 << 18); run("c2 2^18 one-voice", b, t, g, min2=1 << 30)
b, t, g = banks.bank_c2(1 << 20); run("c2 2^20 one-voice", b, t, g, min2=1 << 30)
b, t, g = banks.bank_c2(1 << 20)
car = np.arange(0, 1 << 20, 8); b["voice_freq_mod_osc"][car] = car + 3; b["voice_freq_mod_depth"][car] = 0.2
run("c2 2^20 FM (1/8 carriers)", b, t, g)
b, t, g = banks.bank_c2(1 << 20)
b["voice_one_shot"][::3] = 1; b["voice_loop_enabled"][::3] = 0
run("c2 2^20 one-shots (1/3)", b, t, g)
b, t, g = banks.bank_c2(1 << 20); run("c3 2^20 fast2", b, t, g)
b, t, g = banks.bank_c4(262144); run("c4 262144 linear", b, t, g, interp=1)
