"""PCM banks (table windows): the C4 recipe and its extended variants at the bench size, wall clock per 512-frame block."""
import sys, time
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from skred_amd import banks, device
def run(name, edit=None, n=262144, interp=1, F=512, steps=100):
    b, t, g = banks.RECIPES["c4"](n)
    if edit: edit(b)
    out = torch.zeros(F, 2, device="cuda")
    db = device.DeviceBank(n); db.set_tables(t); db.upload(b); db.set_globals(g); db.kernel_timing(0)
    for _ in range(30): db.render_mix(F, out.data_ptr(), 2, 0, interp)
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(steps): db.render_mix(F, out.data_ptr(), 2, 0, interp)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / steps * 1e3)
    print(f"{name:44s} kernel={db.last_kernel()} ms/block min {min(res):.4f} med {sorted(res)[1]:.4f}", flush=True)
    db.close()
run("c4 262144 linear")
run("c4 262144 truncate", interp=0)
def sh(b):
    v = np.arange(b.n); b["voice_sample_hold_max"][v % 16 == 0] = 4
run("c4 262144 linear, sample & hold on 1/16", sh)
run("c4 1048576 linear", n=1 << 20, steps=40)
