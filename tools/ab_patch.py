"""Banks built by tiling a reference patch (banks.bank_patch: 3.sk, 37.sk, 7.sk, 18.sk, 1.sk): wall clock per 512-frame block
at 2^20 voices, which kernel renders them, voice-samples per second."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from skred_amd import banks, device
n, F = 1 << 20, 512
for patch in (sys.argv[1:] or ["3sk", "37sk", "1sk", "7sk", "18sk"]):
    b, t, g = banks.bank_patch(patch, n)
    out = torch.zeros(F, 2, device="cuda")
    db = device.DeviceBank(n); db.set_tables(t); db.upload(b); db.set_globals(g); db.kernel_timing(0)
    for _ in range(10): db.render_mix(F, out.data_ptr(), 2, 0, 0)
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(30): db.render_mix(F, out.data_ptr(), 2, 0, 0)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 30 * 1e3)
    print(f"patch {patch:5s} x {n} voices  kernel={db.last_kernel()}  ms/block min {min(res):.4f}  {n * F / (min(res) * 1e-3):.3e} voice-samples/s  finite={bool(torch.isfinite(out).all())}", flush=True)
    db.close()
