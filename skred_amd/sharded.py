"""Voices sharded over the GPUs of one node: one process per GPU, one RCCL reduce per launch.

The render loop shards naturally: voices are independent unless they name each
other as modulators, so rank r renders the contiguous block [lo, hi) of the bank and produces a
PRE-master partial mix float[F][2].  The only exchange step of the path is the sum of those
partials -- `torch.distributed.reduce(SUM, dst=0)` (backend "nccl" == RCCL over xGMI; 8*F bytes,
latency-bound) -- after which rank 0 runs the serial master-volume stage (synth.c:616-624) once.

The same class drives the GPU path (bench.py: callables wrap the C ABI) and the world_size-2
gloo tests on CPU (tests/test_sharded_gloo.py: callables wrap the oracle).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def partition(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block of voices for `rank`: [lo, hi).  Blocks differ by at most one voice."""
    return total * rank // world, total * (rank + 1) // world


def modulation_components_ok(bank, lo: int, hi: int) -> bool:
    """A cut is only legal where no voice inside [lo,hi) is modulated by one outside (and vice versa)."""
    a = bank.a
    for key in ("voice_freq_mod_osc", "voice_amp_mod_osc", "voice_pan_mod_osc", "voice_cz_mod_osc"):
        m = a[key][lo:hi]
        used = m >= 0
        if key == "voice_cz_mod_osc":
            used &= a["voice_cz_mode"][lo:hi] != 0
        if (used & ((m < lo) | (m >= hi))).any():
            return False
    return True


class ShardedRender:
    def __init__(self, total_voices: int, rank: int = 0, world: int = 1, root: int = 0, always_reduce: bool = False):
        self.total, self.rank, self.world, self.root = total_voices, rank, world, root
        self.reduce = world > 1 or always_reduce        # always_reduce: rehearse the collective with one rank
        self.lo, self.hi = partition(total_voices, world, rank)
        self._bufs: Optional[Sequence[torch.Tensor]] = None
        self._work: List[Optional[object]] = [None, None]
        self._k = 0

    @property
    def n_local(self) -> int:
        return self.hi - self.lo

    def step(self, render_partial: Callable[[torch.Tensor], None],
             master: Callable[[torch.Tensor, torch.Tensor], None],
             partial: torch.Tensor, out: torch.Tensor) -> None:
        """One pass of the hot path: local render -> (sum over ranks) -> master on the root."""
        render_partial(partial)
        if self.reduce:
            dist.reduce(partial, dst=self.root, op=dist.ReduceOp.SUM)
        if self.rank == self.root:
            master(partial, out)

    # ---- overlapped form ("launch k+1 renders while launch k reduces") ----
    #
    # Two partial buffers alternate.  Block k is rendered into buffer k&1 and its reduce is started
    # asynchronously; the root runs the master stage for block k-1 meanwhile, so a block's mix is
    # delivered one call later (and the last one by drain()).  With RCCL, Work.wait() only makes the
    # current HIP stream wait for the collective -- the host never blocks -- so render k+1 and
    # reduce k run concurrently on the device.  Output samples are identical to step()'s.

    def begin(self, partials: Sequence[torch.Tensor]) -> None:
        assert len(partials) == 2 and self._k == 0
        self._bufs = partials

    def _deliver(self, j: int, master, out) -> None:
        if self._work[j] is not None:
            self._work[j].wait()
            self._work[j] = None
        if self.rank == self.root:
            master(self._bufs[j], out)

    def step_overlapped(self, render_partial: Callable[[torch.Tensor], None],
                        master: Callable[[torch.Tensor, torch.Tensor], None], out: torch.Tensor) -> bool:
        """Render block k, start its reduce, finish block k-1.  True when `out` now holds block k-1."""
        i = self._k & 1
        if self._work[i] is not None:          # the reduce of block k-2 still owns this buffer
            self._work[i].wait()
            self._work[i] = None
        render_partial(self._bufs[i])
        if self.reduce:
            self._work[i] = dist.reduce(self._bufs[i], dst=self.root, op=dist.ReduceOp.SUM, async_op=True)
        delivered = self._k > 0
        if delivered:
            self._deliver(1 - i, master, out)
        self._k += 1
        return delivered

    def drain(self, master: Callable[[torch.Tensor, torch.Tensor], None], out: torch.Tensor) -> bool:
        """Finish the block still in flight.  True when `out` now holds it."""
        if self._k == 0:
            return False
        self._deliver((self._k - 1) & 1, master, out)
        self._k = 0
        return True
