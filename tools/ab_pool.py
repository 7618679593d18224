"""Two-per-lane banks whose table pool is larger than the C3 recipe's (unused tables appended: 25 -> 41 -> 48 KB in LDS): steady
blocks.  SKRED_AMD_LIB selects the library build."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from skred_amd import banks, device
n, F = 1 << 20, 512
bank, t, g = banks.RECIPES["c3"](n)
out = torch.zeros(F, 2, device="cuda")
for pad in (0, 1600, 4000, 5900):
    t2 = np.concatenate([t, np.zeros(pad, np.float32)])
    db = device.DeviceBank(n); db.set_tables(t2); db.upload(bank); db.set_globals(g); db.kernel_timing(0)
    for _ in range(60): db.render_mix(F, out.data_ptr(), 2, 0, 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100): db.render_mix(F, out.data_ptr(), 2, 0, 0)
    torch.cuda.synchronize()
    print(f"pool {len(t2) * 4 / 1024:5.1f} KB  kernel {db.last_kernel()}  {(time.perf_counter() - t0) / 100 * 1e3:.4f} ms/block", flush=True)
    db.close()
