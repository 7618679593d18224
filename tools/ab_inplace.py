"""Note traffic on C3-recipe banks of several sizes: the motion list rendered in place (sk_gain_kernel + the steady kernel's
in-place instantiation) against the envelope kernel beside the steady one (SKRED_OPT_IN_PLACE 1 / 0).  Wall clock per block."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from skred_amd import banks, device
D = device
F = 512
sizes = [int(x) for x in sys.argv[1:] if not x.startswith("d=")] or [1 << 19, 786432, 1 << 20]
DENS = [int(y) for x in sys.argv[1:] if x.startswith("d=") for y in x[2:].split(",")] or [100, 250, 500]
for n in sizes:
    bank, t, g = banks.RECIPES["c3"](n)
    out = torch.zeros(F, 2, device="cuda")
    db = D.DeviceBank(n); db.set_tables(t); db.upload(bank); db.set_globals(g); db.kernel_timing(0)
    for _ in range(60): db.render_mix(F, out.data_ptr(), 2, 0, 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): db.render_mix(F, out.data_ptr(), 2, 0, 0)
    torch.cuda.synchronize(); steady = (time.perf_counter() - t0) / 50 * 1e6
    db.close()
    print(f"{n:8d} voices ({n // 1024} passes): steady {steady:7.1f} us", flush=True)
    for per_m in DENS:                      # voices touched per block and 2^20 voices
        k_ev = max(2, int(per_m * n / (1 << 20)))
        res = {}
        for inpl in (1, 0):
            db = D.DeviceBank(n); db.set_tables(t); db.upload(bank); db.set_globals(g); db.kernel_timing(0); db.in_place(bool(inpl))
            for _ in range(30): db.render_mix(F, out.data_ptr(), 2, 0, 0)
            rng = np.random.default_rng(1)
            def blk():
                vs = rng.choice(n, k_ev, replace=False).astype(np.int32)
                db.update(bank, vs[:k_ev // 2], D.STAMP_RELEASE, 0)
                db.update(bank, vs[k_ev // 2:], D.STAMP_TRIGGER | D.DIRTY_PHASE | D.DIRTY_PARAMS, 0)
                db.render_mix(F, out.data_ptr(), 2, 0, 0)
            for _ in range(30): blk()
            torch.cuda.synchronize()
            best = 1e9
            for rep in range(3):
                t0 = time.perf_counter()
                for _ in range(50): blk()
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / 50 * 1e3)
            res[inpl] = (best, db.last_in_place(), db.list_violations())
            db.close()
        print(f"{n:8d} voices, {k_ev:4d} touched per block: in place {res[1][0]*1e3:7.1f} us (taken: {res[1][1]})   envelope kernel beside {res[0][0]*1e3:7.1f} us   violations {res[1][2]} {res[0][2]}", flush=True)
