"""Worker for tests/test_sharded_gloo.py: one rank of a world_size-N gloo job on CPU.
The renderer here is the ORACLE (allowed in tests); the sharding/reduce/master sequencing is the
product's skred_amd.sharded.ShardedRender, the same object bench.py drives on GPUs."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import cpuref  # noqa: E402
from skred_amd import banks  # noqa: E402
from skred_amd.sharded import ShardedRender, modulation_components_ok  # noqa: E402


def main():
    out_path, n, frames, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    overlapped = len(sys.argv) > 5 and sys.argv[5] == "overlapped"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full, tables, g = banks.bank_c2(n)
    sh = ShardedRender(n, rank, world)
    assert modulation_components_ok(full, sh.lo, sh.hi)
    bank = full.take(slice(sh.lo, sh.hi))
    g_local = g.copy()
    g_master = g.copy()
    partial = torch.zeros(frames, 2, dtype=torch.float32)
    out = torch.zeros(frames, 2, dtype=torch.float32)
    outs = []

    def render_partial(p):
        r = cpuref.render(bank, g_local, tables, frames)
        p.copy_(torch.from_numpy(r["sum32"]))

    def master(p, o):
        o.copy_(torch.from_numpy(cpuref.master(g_master, p.numpy())))

    if overlapped:
        sh.begin([partial, torch.zeros_like(partial)])
        for _ in range(steps):
            if sh.step_overlapped(render_partial, master, out) and rank == 0:
                outs.append(out.numpy().copy())
        if sh.drain(master, out) and rank == 0:
            outs.append(out.numpy().copy())
    else:
        for _ in range(steps):
            sh.step(render_partial, master, partial, out)
            if rank == 0:
                outs.append(out.numpy().copy())
    if rank == 0:
        np.save(out_path, np.concatenate(outs))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
