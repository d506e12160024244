#!/usr/bin/env python3
"""Rollout launch time per policy kind (uniform / stream / greedy / sample), int32 rows and stats only, at two batch sizes on the
config-3 grid.  Usage: python tools/policy_bench.py"""
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
S, T = env.world.size, 1000
for N in (65536, 262144, 4096):
    eng = gua.Engine(N, gua.GridSpec.from_env(env), seed=1)
    eng.reset()
    eng.reserve_trajectory(T)
    eng.vi_set(np.zeros(S), np.random.RandomState(1).dirichlet(np.ones(4), S))
    eng.upload_actions(np.random.RandomState(2).randint(0, 4, (T, N)).astype(np.int32))
    for policy in ('uniform', 'stream', 'greedy', 'sample'):
        for mode, kw in (('int32 rows', (True, False)), ('stats only', (False, True))):
            for auto in (True, False):
                ts = []
                for rnd in range(5):
                    for _ in range(2):
                        eng.rollout(T, policy, auto, *kw)
                    eng.sync()
                    eng.timer_begin()
                    for _ in range(10):
                        eng.rollout(T, policy, auto, *kw)
                    ts.append(eng.timer_end() / 10 * 1e3)
                us = statistics.median(ts)
                print('N %6d %-8s %-10s auto_reset %d  %.1f us per launch  %.3e env-steps/s' % (N, policy, mode, auto, us, N * T / us * 1e6), flush=True)
    eng.close()
