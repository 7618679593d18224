"""Live note traffic on the C3 bank (524 voices touched per block): a short run for a rocprofv3 kernel trace of sk_gain_kernel
and the in-place steady kernel.  SKRED_AMD_LIB selects the library build."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from skred_amd import banks, device
D = device
n, F = 1 << 20, 512
k_ev = int(sys.argv[1]) if len(sys.argv) > 1 else 524
bank, t, g = banks.RECIPES["c3"](n)
out = torch.zeros(F, 2, device="cuda")
db = D.DeviceBank(n); db.set_tables(t); db.upload(bank); db.set_globals(g); db.kernel_timing(0)
for _ in range(40): db.render_mix(F, out.data_ptr(), 2, 0, 0)
rng = np.random.default_rng(1)
for _ in range(80):
    vs = rng.choice(n, k_ev, replace=False).astype(np.int32)
    db.update(bank, vs[:k_ev // 2], D.STAMP_RELEASE, 0)
    db.update(bank, vs[k_ev // 2:], D.STAMP_TRIGGER | D.DIRTY_PHASE | D.DIRTY_PARAMS, 0)
    db.render_mix(F, out.data_ptr(), 2, 0, 0)
torch.cuda.synchronize()
print("in place:", db.last_in_place(), "violations", db.list_violations())
db.close()
