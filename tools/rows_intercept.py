#!/usr/bin/env python3
"""Fixed cost and slope of an int32-row launch: wall us per launch (back to back) against the number of steps, a config-4 shard
(32 768 envs, lava grid), config 2 (4096 envs) and config 3 (65 536 envs), store limiter off and on.
    python tools/rows_intercept.py"""
import random
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit('/', 2)[0])
import griduniverse_amd as gua  # noqa: E402


def workload(name):
    random.seed(123)
    np.random.seed(123)
    if name == 'c2':
        return gua.GridUniverseEnv(grid_shape=(8, 8)), 4096
    if name == 'c4':
        return gua.GridUniverseEnv(grid_shape=(32, 32), lava_states=[16 + 32 * r for r in range(24)], goal_states=[1023]), 32768
    return gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True), 65536


TS = (16, 64, 256, 512, 1000, 2000, 4000)
for wname in ('c4', 'c2', 'c3'):
    env, N = workload(wname)
    spec = gua.GridSpec.from_env(env)
    for pace in (0, None):
        eng = gua.Engine(N, spec, seed=5)
        eng.set_option('rollout_pace', pace)
        eng.reset()
        eng.reserve_trajectory(max(TS))
        row = []
        for T in TS:
            for _ in range(150 if pace is None else 10):
                eng.rollout(T, 'uniform', True, True)
            best = []
            for _ in range(3):
                eng.sync()
                eng.timer_begin()
                for _ in range(40):
                    eng.rollout(T, 'uniform', True, True)
                best.append(eng.timer_end() / 40 * 1e3)
            row.append(min(best))
        a, b = np.polyfit(np.array(TS[3:], float), np.array(row[3:]), 1)
        print('%-3s %6d envs limiter %-4s: ' % (wname, N, 'off' if pace == 0 else 'loop') + '  '.join('T=%d %.2f' % (t, r) for t, r in zip(TS, row)) +
              '   slope %.2f ns/step, intercept %.2f us (fit over T >= 512)' % (a * 1e3, b), flush=True)
        eng.close()
