"""Scripted sequencer sessions shared by tests/seq_replay.py (the compiled reference) and tests/test_seq_clock.py (ours)."""
import numpy as np


def script(seed: int):
    rng = np.random.default_rng(seed)
    ops = []
    bpm = [120.0, 97.5, 240.0, 33.0][seed % 4]
    ops.append(("tempo", bpm))
    for p in range(16):
        length = int(rng.integers(1, 20))
        for s in range(length):
            ops.append(("step", p, s, 1))
        if p % 5 == 4:                                   # a hole in the middle: the pattern wraps there, the tail never plays
            ops.append(("step", p, max(1, length // 2), 0))
        if p % 3 == 1:
            ops.append(("mute", p, int(rng.integers(0, length)), 1))
        ops.append(("modulo", p, [4, 1, 2, 3, 0, 7][p % 6]))
        if p != 13:
            ops.append(("state", p, 1))
    frames = [512, 512, 128, 100, 33, 2048, 512, 7]
    for k in range(700):
        ops.append(("block", frames[int(rng.integers(0, len(frames)))] if seed % 2 else 512))
        if k == 200:
            ops += [("state", 2, 2), ("state", 3, 0), ("tempo", bpm * 1.5), ("mute", 0, 0, 1), ("step", 1, 0, 0)]
        if k == 300:
            ops += [("state", 2, 3), ("state", 3, 1), ("state", 13, 1), ("reset", 7), ("step", 7, 0, 1), ("step", 7, 1, 1), ("state", 7, 1),
                    ("modulo", 5, 1), ("mute", 0, 0, 0)]
    return ops
