#!/usr/bin/env python3
"""env-steps/s of the fused rollout as a function of the batch size (config-3 grid), three output modes."""
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import griduniverse_amd as gua  # noqa: E402

random.seed(123)
np.random.seed(123)
env = gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True)
spec = gua.GridSpec.from_env(env)
print('%9s %6s | %12s %8s | %12s %8s | %12s' % ('envs', 'T', 'int32 rows', 'TB/s', 'packed', 'TB/s', 'stats only'))
for N in (1024, 4096, 16384, 65536, 262144, 1 << 20, 1 << 22):
    T = 1000 if N <= 262144 else (250 if N <= (1 << 20) else 64)
    eng = gua.Engine(N, spec, seed=1)
    eng.reset()
    eng.reserve_trajectory(T)
    res = []
    for mode in (True, 'packed', False):
        for _ in range(2):
            eng.rollout(T, 'uniform', True, mode, mode is False)
        eng.sync()
        eng.timer_begin()
        reps = 10
        for _ in range(reps):
            eng.rollout(T, 'uniform', True, mode, mode is False)
        res.append(N * T * reps / (eng.timer_end() / 1e3))
    print('%9d %6d | %12.3e %8.2f | %12.3e %8.2f | %12.3e' % (N, T, res[0], res[0] * 12 / 1e12, res[1], res[1] * 4 / 1e12, res[2]))
    eng.close()
