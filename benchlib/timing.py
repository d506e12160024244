"""Timing of bench.py: the host channel between ranks, blocks of exactly K launches between barrier + device sync pairs."""
import contextlib
import ctypes
import os
import sys
import time

import numpy as np


class Ranks(object):
    """Host channel between the ranks: griduniverse_amd.rendezvous (one socket per rank to rank 0; torchrun-style environment,
    no PyTorch).  A no-op for one process."""

    def __init__(self, rank, world):
        from griduniverse_amd.rendezvous import Rendezvous
        self.rank, self.world = rank, world
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        self.rdzv = Rendezvous(rank, world, join_timeout=float(os.environ.get('GU_RDZV_JOIN_TIMEOUT', '600')))
        self.rdzv.barrier()

    def barrier(self):
        self.rdzv.barrier()

    def reduce(self, values, op):
        """Element-wise MAX / MIN over ranks of a list of floats."""
        return self.rdzv.reduce(values, op)

    def gather(self, values):
        """[world][len] of every rank's list of floats."""
        return self.rdzv.gather(values)

    def gather_bytes(self, payload):
        return self.rdzv.gather_bytes(payload)

    def broadcast_bytes(self, payload, src=0):
        return self.rdzv.broadcast_bytes(payload, src)

    def close(self):
        self.rdzv.close()


def timed_block(eng, ranks, T, K):
    """EXACTLY K launches between barrier + device sync pairs.  Returns (wall seconds, HIP-event ms) of this rank."""
    eng.sync()
    ranks.barrier()
    t0 = time.perf_counter()
    eng.timer_begin()
    for _ in range(K):
        eng.rollout(T, 'uniform', auto_reset=True, trajectory=True)
    kernel_ms = eng.timer_end()  # HIP events on the engine's stream; also drains it
    eng.sync()
    elapsed = time.perf_counter() - t0
    ranks.barrier()
    return elapsed, kernel_ms


def timed_region(eng, ranks, T, K, min_seconds, max_blocks=4000):
    """One untimed probe block sizes the region (identically on every rank: its time is max-reduced), then B timed blocks.
    Returns per-block wall seconds and HIP-event ms (each MAX over ranks), this rank's own per-block wall seconds, and the
    number of launches issued."""
    probe = ranks.reduce([timed_block(eng, ranks, T, K)[0]], 'MAX')[0]
    blocks = int(min(max_blocks, max(3, np.ceil(min_seconds / max(probe, 1e-6)))))
    wall, kern = [], []
    for _ in range(blocks):
        e, k = timed_block(eng, ranks, T, K)
        wall.append(e)
        kern.append(k)
    both = ranks.reduce(wall + kern, 'MAX')
    return both[:blocks], both[blocks:], wall, (blocks + 1) * K


@contextlib.contextmanager
def native_stdout_to_stderr():
    """RCCL prints a banner (ROCm version, hostname, library path) to the C-level stdout when a communicator comes up, and C
    stdio flushes it whenever it likes -- after the JSON line, when stdout is a pipe.  The driver reads ONE JSON line from
    stdout, so everything native code prints inside this block goes to stderr instead."""
    libc = ctypes.CDLL(None)
    sys.stdout.flush()
    libc.fflush(None)
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        libc.fflush(None)
        os.dup2(saved, 1)
        os.close(saved)


def spread(values):
    v = np.sort(np.asarray(values, dtype=np.float64))
    return float(v[0]), float(np.median(v)), float(v[-1])



def settle_launches(eng, T, policy, **kw):
    """Untimed launches in front of a timed block of a launch kind: 3, or 200 for a kind whose rows keep a store schedule
    (the closed loop of the store pacing runs inside the launches themselves; 200 launches = 20 ms at the headline size)."""
    eng.rollout(T, policy, auto_reset=True, **kw)
    paced = hasattr(eng, 'rollout_pacing') and eng.rollout_pacing(policy, True, packed=kw.get('trajectory') == 'packed') is not None
    return 200 if paced else 3


def launch_ms(eng, T, K, policy='uniform', **kw):
    """HIP-event ms per launch of one launch kind after its settle launches (beside `value`, never as it)."""
    for _ in range(settle_launches(eng, T, policy, **kw)):
        eng.rollout(T, policy, auto_reset=True, **kw)
    eng.sync()
    eng.timer_begin()
    for _ in range(K):
        eng.rollout(T, policy, auto_reset=True, **kw)
    return eng.timer_end() / K
