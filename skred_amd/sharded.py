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

from typing import Callable, Tuple

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
    def __init__(self, total_voices: int, rank: int = 0, world: int = 1, root: int = 0):
        self.total, self.rank, self.world, self.root = total_voices, rank, world, root
        self.lo, self.hi = partition(total_voices, world, rank)

    @property
    def n_local(self) -> int:
        return self.hi - self.lo

    def step(self, render_partial: Callable[[torch.Tensor], None],
             master: Callable[[torch.Tensor, torch.Tensor], None],
             partial: torch.Tensor, out: torch.Tensor) -> None:
        """One pass of the hot path: local render -> (sum over ranks) -> master on the root."""
        render_partial(partial)
        if self.world > 1:
            dist.reduce(partial, dst=self.root, op=dist.ReduceOp.SUM)
        if self.rank == self.root:
            master(partial, out)
