"""Envelopes in motion on EXTENDED one-voice banks (a third of the voices one-shots / every 16th with sample & hold / some
filters off): the recipe's first 11 blocks against its steady blocks."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from skred_amd import banks, device
F = 512
def run(name, n, edit):
    bank, t, g = banks.RECIPES["c2"](n)
    edit(bank)
    out = torch.zeros(F, 2, device="cuda")
    db = device.DeviceBank(n); db.set_tables(t); db.upload(bank); db.set_globals(g); db.kernel_timing(0)
    for _ in range(20): db.render_mix(F, out.data_ptr(), 2, 0, 0)
    torch.cuda.synchronize()
    reps = []
    for _ in range(4):
        db.upload(bank); db.set_globals(g); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(11): db.render_mix(F, out.data_ptr(), 2, 0, 0)
        torch.cuda.synchronize()
        reps.append((time.perf_counter() - t0) / 11 * 1e3)
    for _ in range(30): db.render_mix(F, out.data_ptr(), 2, 0, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100): db.render_mix(F, out.data_ptr(), 2, 0, 0)
    torch.cuda.synchronize()
    st = (time.perf_counter() - t0) / 100 * 1e3
    print(f"{name:52s} kernel {db.last_kernel()}  in motion {min(reps)*1e3:7.1f} us/block   steady {st*1e3:7.1f} us/block", flush=True)
    db.close()
def sh(b): b["voice_sample_hold_max"][::16] = 4
def mixed(b): b["voice_filter_mode"][::3] = 0; b["voice_use_amp_envelope"][::5] = 0
def noise(b): b["voice_wave_table_index"][::20] = 6
for n in (65536, 131072):
    run(f"c2 {n} sample & hold on 1/16", n, sh)
    run(f"c2 {n} a third unfiltered, a fifth without envelope", n, mixed)
    run(f"c2 {n} 5 % noise voices", n, noise)
