#!/usr/bin/env python3
"""General kernel against transition-row kernel (options rollout_rows 0 / 1) by batch size and policy kind, int32 rows and packed
rows, 1000 steps per launch: us per launch over 12 launches, schedule limiter calibrated where it applies.  The default dispatch
(gu_rollout_rows.hip: gu_rollout_rows) is drawn from tables like this one.
    python tools/rows_crossover.py [batch sizes ...]"""
import random
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit('/', 2)[0])
import griduniverse_amd as gua  # noqa: E402

sizes = [int(x) for x in sys.argv[1:]] or [8192, 16384, 32768, 49152, 65536]
random.seed(123)
np.random.seed(123)
env = gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True)
spec = gua.GridSpec.from_env(env)
T = 1000
rs = np.random.RandomState(1)
pi = rs.dirichlet(np.ones(4), 1024)
print('%-7s %-8s %-6s %-7s %10s %10s' % ('envs', 'policy', 'auto', 'rows', 'general', 'row-table'))
for N in sizes:
    acts = rs.randint(0, 4, size=(T, N)).astype(np.int32)
    res = {}
    for rows in (0, 1):
        eng = gua.Engine(N, spec, seed=123)
        eng.reset()
        eng.reserve_trajectory(T)
        eng.vi_set(np.zeros(1024), pi)
        eng.upload_actions(acts)
        eng.set_option('rollout_rows', rows)
        for policy in ('uniform', 'stream', 'greedy', 'sample'):
            for auto in (True, False):
                for traj in (True, 'packed'):
                    for _ in range(3):
                        eng.rollout(T, policy, auto, traj)
                    eng.sync()
                    eng.timer_begin()
                    for _ in range(12):
                        eng.rollout(T, policy, auto, traj)
                    res[(policy, auto, traj, rows)] = eng.timer_end() / 12 * 1e3
        eng.close()
    for policy in ('uniform', 'stream', 'greedy', 'sample'):
        for auto in (True, False):
            for traj in (True, 'packed'):
                g, r = res[(policy, auto, traj, 0)], res[(policy, auto, traj, 1)]
                print('%-7d %-8s %-6s %-7s %10.1f %10.1f %s' % (N, policy, auto, 'int32' if traj is True else 'packed', g, r, '<- row-table' if r < 0.97 * g else ('<- general' if g < 0.97 * r else '')), flush=True)
