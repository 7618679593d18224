"""The two regimes in which sk_render_env2_kernel carries the bank: the C3 recipe from its first frame (attack / decay ramps
in flight) and the bank under note traffic.  Wall clock per 512-frame block; SKRED_AMD_LIB selects the library build."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from skred_amd import banks, device
D = device
n, F = 1 << 20, 512
bank, t, g = banks.RECIPES["c3"](n)
out = torch.zeros(F, 2, device="cuda")
db = D.DeviceBank(n); db.set_tables(t); db.upload(bank); db.set_globals(g); db.kernel_timing(0)
if os.environ.get("AB_IN_PLACE"): db.in_place(int(os.environ["AB_IN_PLACE"]))   # SKRED_OPT_IN_PLACE: 0 never, 2 whenever the rows suffice
for _ in range(20): db.render_mix(F, out.data_ptr(), 2, 0, 0)
torch.cuda.synchronize()
reps = []
for _ in range(4):
    db.upload(bank); db.set_globals(g); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(11): db.render_mix(F, out.data_ptr(), 2, 0, 0)
    torch.cuda.synchronize()
    reps.append((time.perf_counter() - t0) / 11 * 1e3)
print(f"envelopes_in_motion  ms/block min {min(reps):.4f} med {sorted(reps)[len(reps)//2]:.4f}", flush=True)
for _ in range(12): db.render_mix(F, out.data_ptr(), 2, 0, 0)
rng = np.random.default_rng(1)
for k_ev in (104, 524, 5242):
    def blk():
        vs = rng.choice(n, k_ev, replace=False).astype(np.int32)
        db.update(bank, vs[:k_ev // 2], D.STAMP_RELEASE, 0)
        db.update(bank, vs[k_ev // 2:], D.STAMP_TRIGGER | D.DIRTY_PHASE | D.DIRTY_PARAMS, 0)
        db.render_mix(F, out.data_ptr(), 2, 0, 0)
    for _ in range(20): blk()
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(60): blk()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 60 * 1e3)
    print(f"live_control {k_ev:5d} voices/block  ms/block min {min(res):.4f} med {sorted(res)[1]:.4f}", flush=True)
for _ in range(30): db.render_mix(F, out.data_ptr(), 2, 0, 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100): db.render_mix(F, out.data_ptr(), 2, 0, 0)
torch.cuda.synchronize()
print(f"steady               ms/block {(time.perf_counter() - t0) / 100 * 1e3:.4f}  list violations {db.list_violations()}")
db.close()
