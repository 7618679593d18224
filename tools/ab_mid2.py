#!/usr/bin/env python3
"""tools/ab_mid2.py -- wall clock per 512-frame block of the small and mid-size banks the one-voice family renders (run on the GPU box;
SKRED_AMD_LIB picks the library build: tools/ab_libs.sh)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from skred_amd import banks, device
def run(rec, n, interp=0, F=512, steps=200, split=None):
    b, t, g = banks.RECIPES[rec](n)
    out = torch.zeros(F, 2, device="cuda")
    db = device.DeviceBank(n); db.set_tables(t); db.upload(b); db.set_globals(g); db.fast2_min_voices(1 << 30); db.kernel_timing(0)
    if split is not None: db.set_split(split)
    for _ in range(40): db.render_mix(F, out.data_ptr(), 2, 0, interp)
    torch.cuda.synchronize()
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.05:
        for _ in range(8): db.render_mix(F, out.data_ptr(), 2, 0, interp)
        torch.cuda.synchronize()
    res = []
    for _rep in range(5):
        t0 = time.perf_counter()
        for _ in range(steps): db.render_mix(F, out.data_ptr(), 2, 0, interp)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / steps * 1e6)
    sp = db.last_split(); db.close()
    print(f"{rec} {n:8d} interp={interp} split={sp}: {min(res):7.2f} us (med {sorted(res)[2]:7.2f})   lib={os.path.basename(os.path.dirname(os.environ.get('SKRED_AMD_LIB', 'default/x')))}", flush=True)
for rec, n in (("c1", 4096), ("c2", 65536), ("c2", 131072), ("c2", 196608)):
    run(rec, n, split=0)
run("c2", 65536, split=2)
run("c2", 65536, interp=1, split=0)
