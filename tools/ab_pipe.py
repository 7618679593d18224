"""The shard block of strong scaling (2^17 voices) through the serial and the pipelined N > 1 sequences on one GPU, with and
without the one-rank RCCL reduce in the stream: what the cross-stream event chain itself costs."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from skred_amd import banks
from skred_amd.sharded import Shard
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 17
whole, tables, g = banks.RECIPES["c2"](n)
stream = torch.cuda.current_stream().cuda_stream
for form in ("fused", "serial", "pipelined", "serial+rccl", "pipelined+rccl"):
    sh = Shard(n, 0, 1, 0)
    sh.bank.set_tables(tables); sh.upload(whole); sh.bank.set_globals(g); sh.bank.kernel_timing(0)
    if form.endswith("rccl"):
        sh.init_rccl(Shard.rccl_unique_id()); sh.set_reduce(None, always_reduce=True)
    o = [torch.zeros(512, 2, device="cuda") for _ in range(2)]
    k = [0]
    def blk():
        if form == "fused": sh.bank.render_mix(512, o[0].data_ptr(), 2, 0, 0, stream)
        elif form.startswith("serial"): sh.render_mix(512, o[0].data_ptr(), 2, 0, stream)
        else:
            sh.render_mix_pipelined(512, o[k[0] & 1].data_ptr(), 2, 0, stream); k[0] += 1
    for _ in range(60): blk()
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(200): blk()
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 200 * 1e6)
    print(f"{n} voices  {form:16s} us/block min {min(res):7.1f} med {sorted(res)[1]:7.1f}", flush=True)
    sh.close()
