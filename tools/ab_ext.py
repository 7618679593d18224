"""Extended one-voice banks at 2^20 voices, steady: FM from further up (exchange), one-shots, sample & hold, noise."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from skred_amd import banks, device
F = 512
def run(name, edit, n=1 << 20):
    b, t, g = banks.bank_c2(n)
    edit(b)
    out = torch.zeros(F, 2, device="cuda")
    db = device.DeviceBank(n); db.set_tables(t); db.upload(b); db.set_globals(g); db.kernel_timing(0)
    for _ in range(40): db.render_mix(F, out.data_ptr(), 2, 0, 0)
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(40): db.render_mix(F, out.data_ptr(), 2, 0, 0)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 40 * 1e3)
    print(f"{name:44s} kernel={db.last_kernel()} ms/block min {min(res):.4f} med {sorted(res)[1]:.4f}", flush=True)
    db.close()
def fm8(b):
    car = np.arange(0, b.n, 8); b["voice_freq_mod_osc"][car] = car + 3; b["voice_freq_mod_depth"][car] = 0.2
def shots(b): b["voice_one_shot"][::3] = 1; b["voice_loop_enabled"][::3] = 0
def sh(b): b["voice_sample_hold_max"][::16] = 4
def noise(b): b["voice_wave_table_index"][::20] = 6
def mixed(b): b["voice_filter_mode"][::3] = 0; b["voice_use_amp_envelope"][::5] = 0
run("c2 2^20 FM (1/8 carriers, modulator +3)", fm8)
run("c2 2^20 one-shots (1/3)", shots)
run("c2 2^20 sample & hold on 1/16", sh)
run("c2 2^20 5 % noise", noise)
run("c2 131072 mixed filter / envelope use", mixed, n=131072)
