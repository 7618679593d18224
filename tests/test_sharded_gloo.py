"""N>1 path on CPU: world_size-2 (and 3, ragged) gloo jobs.  Each rank renders its block of voices,
the partial mixes are summed with torch.distributed.reduce, rank 0 applies the master stage; the
result must match the single-process render of the whole bank within the float-mix tolerance."""
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


@pytest.mark.parametrize("world,n,mode", [(2, 2048, "sync"), (3, 1000, "sync"), (2, 2048, "overlapped"),
                                          (3, 1000, "overlapped")])
def test_sharded_matches_single_process(tmp_path, world, n, mode):
    frames, steps = 256, 4 if mode == "overlapped" else 3
    out = str(tmp_path / "mix.npy")
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "_dist_worker.py"), out,
                                       str(n), str(frames), str(steps), mode], env=env))
    for p in procs:
        assert p.wait(timeout=240) == 0
    got = np.load(out)
    assert got.shape == (frames * steps, 2)
    bank, tables, g = banks.bank_c2(n)
    ref, _ = cpuref.synth(bank, g, tables, frames * steps)
    err = np.sqrt(np.mean((got.astype(np.float64) - ref) ** 2)) / np.sqrt(np.mean(ref.astype(np.float64) ** 2))
    assert err <= 1e-5, err
