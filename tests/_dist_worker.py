"""Worker for tests/test_sharded_gloo.py: one rank of a world_size-N gloo job on CPU.
The renderer here is the ORACLE (allowed in tests) and the collective is gloo; the partition, the cut check and the
per-block sequence render -> reduce -> master are the product's C code (skred_amd/csrc/skred_shard.c through
skred_amd.sharded.ShardedRender), the same sequencing bench.py runs on GPUs with RCCL."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import cpuref  # noqa: E402
from skred_amd import banks  # noqa: E402
from skred_amd.sharded import ShardedRender, modulation_components_ok, partition  # noqa: E402


def main_fx():
    """The fixed-point bank through the same C sequencing: int64 partials behind the float pointers, gloo's int64 sum."""
    import ctypes as C
    from skred_amd import fxbank
    from skred_amd.sharded import RENDER_FN, MASTER_FN, REDUCE_FN, ShardOps, _lib
    out_path, n, frames, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full, pool, c0 = fxbank.bank_fx(n)
    lo, hi = partition(n, world, rank)
    bank = full.take(slice(lo, hi))
    st = {"cnt": c0, "g": 0}

    def i64(ptr, rows):
        return np.ctypeslib.as_array((C.c_int64 * (rows * 2)).from_address(ptr)).reshape(rows, 2)

    def _render(_ctx, frames_, interp, partial, _stream):
        mix, _, st["cnt"] = cpuref.fx_render(bank, pool, st["cnt"], frames_, interp)
        i64(partial, frames_)[:] = mix
        return 0

    def _reduce(_ctx, partial, n_el, root, _stream):
        t = torch.from_numpy(i64(partial, n_el // 2))
        dist.reduce(t, dst=root, op=dist.ReduceOp.SUM)
        return 0

    def _master(_ctx, total, frames_, _ch, out, _stream):
        o, st["g"] = cpuref.fx_master(fxbank.MASTER_TARGET_Q31, fxbank.MASTER_K_Q15, st["g"], i64(total, frames_).copy())
        i64(out, frames_)[:] = o
        return 0

    ops = ShardOps()
    ops.render, ops.master, ops.reduce = RENDER_FN(_render), MASTER_FN(_master), REDUCE_FN(_reduce)
    L = _lib()
    h = C.c_void_p()
    assert L.skred_shard_create_custom(rank, world, 0, n, C.byref(ops), C.byref(h)) == 0
    partial = np.zeros((frames, 2), np.int64)
    out = np.zeros((frames, 2), np.int64)
    outs = []
    for _ in range(steps):
        assert L.skred_shard_render_mix(h, frames, 1, partial.ctypes.data, out.ctypes.data, 2, None) == 0
        if rank == 0:
            outs.append(out.copy())
    L.skred_shard_destroy(h)
    if rank == 0:
        np.save(out_path, np.concatenate(outs))
    dist.barrier()
    dist.destroy_process_group()


def main():
    if len(sys.argv) > 5 and sys.argv[5] == "fx":
        return main_fx()
    out_path, n, frames, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    pipelined = len(sys.argv) > 5 and sys.argv[5] == "pipelined"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    full, tables, g = banks.bank_c2(n)
    lo, hi = partition(n, world, rank)
    assert modulation_components_ok(full, lo, hi)
    bank = full.take(slice(lo, hi))
    g_local = g.copy()
    g_master = g.copy()

    def render(frames_, interp, partial):
        partial[:] = cpuref.render(bank, g_local, tables, frames_, interp)["sum32"]

    def reduce(partial, root):
        t = torch.from_numpy(partial)              # shares the C buffer
        dist.reduce(t, dst=root, op=dist.ReduceOp.SUM)

    def master(total, out):
        out[:] = cpuref.master(g_master, total.copy())

    sh = ShardedRender(n, rank, world, render, master, reduce)
    assert (sh.lo, sh.hi) == (lo, hi)
    partial = np.zeros((frames, 2), np.float32)
    out = np.zeros((frames, 2), np.float32)
    outs = []
    for _ in range(steps):
        if pipelined:                 # skred_shard_render_mix_pipelined: the library's own rotating partial buffers
            sh.step_pipelined(out)
        else:
            sh.step(partial, out)
        if rank == 0:
            outs.append(out.copy())
    sh.close()
    if rank == 0:
        np.save(out_path, np.concatenate(outs))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
