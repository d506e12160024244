#!/usr/bin/env python3
"""Statistics-only uniform rollout: K-step kernel (gu_rollout_multi.hip) against the row-table kernel, launch time at several T
(slope = time per 1000 steps, intercept = table staging + first step), table copies 1 / 2, three batch sizes, 32x32 and 8x8 grids.
Usage: python tools/multi_ab.py"""
import os
import random
import statistics
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import griduniverse_amd as gua  # noqa: E402


def launch_us(eng, T, reps=5):
    ts = []
    for _ in range(reps):
        for _ in range(2):
            eng.rollout(T, 'uniform', True, False, True)
        eng.sync()
        eng.timer_begin()
        for _ in range(10):
            eng.rollout(T, 'uniform', True, False, True)
        ts.append(eng.timer_end() / 10 * 1e3)
    return statistics.median(ts)


random.seed(123)
np.random.seed(123)
grids = {'32x32 maze': gua.GridSpec.from_env(gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True)),
         '8x8 open': gua.GridSpec(8, 8, [0], [63], [], [])}
for gname, spec in grids.items():
    for N in (4096, 65536, 262144):
        eng = gua.Engine(N, spec, seed=1)
        eng.reset()
        for label, env in (('row table     ', {'GU_ROLLOUT_MULTI': '0'}), ('K-step          ', {'GU_ROLLOUT_MULTI': '1'}),
                           ('K-step, 2 copies', {'GU_ROLLOUT_MULTI': '1', 'GU_ROLLOUT_MULTI_COPIES': '2'}),
                           ('K = 2           ', {'GU_ROLLOUT_MULTI': '1', 'GU_ROLLOUT_MULTI_K': '2'})):
            for k in ('GU_ROLLOUT_MULTI', 'GU_ROLLOUT_MULTI_COPIES', 'GU_ROLLOUT_MULTI_K'):
                os.environ.pop(k, None)
            os.environ.update(env)
            t = {T: launch_us(eng, T) for T in (64, 1000, 2000)}
            slope = (t[2000] - t[1000])
            print('%-10s N %6d  %s  T=64 %.1f us  T=1000 %.1f us  T=2000 %.1f us  -> %.1f us per 1000 steps + %.1f us fixed = %.2e env-steps/s at T=1000'
                  % (gname, N, label, t[64], t[1000], t[2000], slope, t[1000] - slope, N * 1000 / t[1000] * 1e6), flush=True)
        eng.close()
