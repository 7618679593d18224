"""N>1 path on CPU: world_size-2 (and 3, ragged) gloo jobs driving the C sharding code (skred_shard_*: partition,
cut check, render -> reduce -> master sequencing) with the oracle as renderer and gloo as the collective; the result
must match the single-process render of the whole bank within the float-mix tolerance."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from oracle import cpuref
from skred_amd import banks
from skred_amd.sharded import partition

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_partition_covers_bank():
    for total in (1, 7, 64, 1000, 1 << 20):
        for world in (1, 2, 3, 8):
            cuts = [partition(total, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == total
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_cut_check_refuses_modulation_across_ranks():
    from skred_amd.sharded import modulation_components_ok
    bank, _, _ = banks.bank_c2(512)
    assert modulation_components_ok(bank, 0, 256) and modulation_components_ok(bank, 256, 512)
    bank["voice_freq_mod_osc"][10] = 300                       # carrier on rank 0, modulator on rank 1
    assert not modulation_components_ok(bank, 0, 256)
    assert modulation_components_ok(bank, 256, 512)            # rank 1's block itself is self-contained
    bank["voice_freq_mod_osc"][10] = -1
    bank["voice_cz_mod_osc"][300] = 5                           # a CZ source only counts with CZ on (synth.c:262)
    assert modulation_components_ok(bank, 256, 512)
    bank["voice_cz_mode"][300] = 2
    assert not modulation_components_ok(bank, 256, 512)


@pytest.mark.parametrize("world,n,form", [(2, 2048, "serial"), (3, 1000, "serial"), (2, 2048, "pipelined")])
def test_sharded_matches_single_process(tmp_path, world, n, form):
    frames, steps = 256, 3
    out = str(tmp_path / "mix.npy")
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), out,
                                       str(n), str(frames), str(steps), form], env=env))
    for p in procs:
        assert p.wait(timeout=240) == 0
    got = np.load(out)
    assert got.shape == (frames * steps, 2)
    bank, tables, g = banks.bank_c2(n)
    ref, _ = cpuref.synth(bank, g, tables, frames * steps)
    err = np.sqrt(np.mean((got.astype(np.float64) - ref) ** 2)) / np.sqrt(np.mean(ref.astype(np.float64) ** 2))
    assert err <= 1e-5, err


@pytest.mark.parametrize("world", [2, 3])
def test_fixed_point_shards_sum_exactly(tmp_path, world):
    """The fixed-point path through the same C sequencing (int64 partials, gloo's integer sum): the root's output equals the
    single-process definition BIT FOR BIT, whatever the number of ranks -- integer sums do not depend on the order."""
    from skred_amd import fxbank
    n, frames, steps = 1500, 200, 3
    out = str(tmp_path / "fxmix.npy")
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), out, str(n), str(frames), str(steps), "fx"], env=env))
    for p in procs:
        assert p.wait(timeout=240) == 0
    got = np.load(out)
    b, pool, cnt = fxbank.bank_fx(n)
    g, want = 0, []
    for _ in range(steps):
        mix, _, cnt = cpuref.fx_render(b, pool, cnt, frames, 1)
        o, g = cpuref.fx_master(fxbank.MASTER_TARGET_Q31, fxbank.MASTER_K_Q15, g, mix)
        want.append(o)
    assert got.dtype == np.int64 and (got == np.concatenate(want)).all()
