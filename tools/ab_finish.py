# cost of the in-kernel mix-down: run once with the in-tree library and once with a library built with
#   make -C skred_amd/csrc OBJ=_obj_nofin OUT=../../_ab/nofinish/libskred_amd.so EXTRA=-DSK_ABLATE_FINISH
# (tools/ab_libs.sh ab_finish.py nofinish ...): that build skips sk_finish_block, its output is garbage
import sys, time, os
import numpy as np, torch
sys.path.insert(0, ".")
from skred_amd import banks, device
def run(name, rec, n, interp=0, F=512, steps=100, min2=None):
    b, t, g = banks.RECIPES[rec](n)
    out = torch.zeros(F, 2, device="cuda")
    db = device.DeviceBank(n); db.set_tables(t); db.upload(b); db.set_globals(g)
    if min2 is not None: db.fast2_min_voices(min2)
    db.kernel_timing(0)
    for _ in range(30): db.render_mix(F, out.data_ptr(), 2, 0, interp)
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(steps): db.render_mix(F, out.data_ptr(), 2, 0, interp)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / steps * 1e3)
    print(f"{name:40s} kernel={db.last_kernel()} ms/block min {min(res):.4f} med {sorted(res)[1]:.4f}  lib={os.environ.get('SKRED_AMD_LIB','in-tree')}", flush=True)
    db.close()
run("c1 4096", "c1", 4096)
run("c2 65536", "c2", 65536)
run("c2 131072 one", "c2", 131072, min2=1<<30)
run("c2 131072 two", "c2", 131072, min2=1)
run("c3 2^20", "c2", 1<<20)
run("c4 262144", "c4", 262144, interp=1)
