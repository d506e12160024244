#!/usr/bin/env python3
"""Caller-supplied action stream with trajectory rows: row-table kernel against the general kernel (packed words staged in
LDS) by batch size.  Usage: python tools/stream_crossover.py"""
import os
import random
import statistics
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import griduniverse_amd as gua  # noqa: E402

random.seed(123)
np.random.seed(123)
env = gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True)
spec = gua.GridSpec.from_env(env)
T = 1000
for N in (1024, 4096, 8192, 16384, 32768, 65536):
    eng = gua.Engine(N, spec, seed=1)
    eng.reset()
    eng.reserve_trajectory(T)
    eng.upload_actions(np.random.RandomState(2).randint(0, 4, (T, N)).astype(np.int32))
    for policy in ('stream', 'uniform'):
        for mode, kw in (('int32 rows', dict(trajectory=True)), ('packed rows', dict(trajectory='packed'))):
            res = {}
            for rows in ('0', '1'):
                os.environ['GU_ROLLOUT_ROWS'] = rows
                ts = []
                for _ in range(5):
                    for _ in range(2):
                        eng.rollout(T, policy, True, **kw)
                    eng.sync()
                    eng.timer_begin()
                    for _ in range(10):
                        eng.rollout(T, policy, True, **kw)
                    ts.append(eng.timer_end() / 10 * 1e3)
                res[rows] = statistics.median(ts)
            del os.environ['GU_ROLLOUT_ROWS']
            print('N %6d %-8s %-11s general %.1f us  row table %.1f us  -> %s' % (N, policy, mode, res['0'], res['1'], 'rows' if res['1'] < res['0'] else 'general'), flush=True)
    eng.close()
