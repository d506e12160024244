#!/usr/bin/env python3
"""The sampled-policy rollout with int32 rows at config 3 (65 536 envs x 1000 steps): us per launch under the closed loop of the store
pacing, without a limiter and at fixed periods, on the transition-row kernel (the default) and on the general kernel
(rollout_rows = 0), next to the uniform-policy headline of the same engine.  Usage (GPU box): python tools/sample_rows_sweep.py"""
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import griduniverse_amd as gua  # noqa: E402

N, T, K = 65536, 1000, 20
random.seed(123)
np.random.seed(123)
env = gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True)
S = env.world.size


def timed(eng, policy, settle, **kw):
    for _ in range(settle):
        eng.rollout(T, policy, auto_reset=True, **kw)
    eng.sync()
    eng.timer_begin()
    for _ in range(K):
        eng.rollout(T, policy, auto_reset=True, **kw)
    return eng.timer_end() / K * 1e3


for rows_opt in (None, 0):
    eng = gua.Engine(N, gua.GridSpec.from_env(env), seed=123)
    eng.set_option('rollout_rows', rows_opt)
    eng.vi_set(np.zeros(S), np.random.RandomState(1).dirichlet(np.ones(4), S))
    eng.reset()
    eng.reserve_trajectory(T)
    for policy in ('uniform', 'sample'):
        row = {}
        eng.set_option('rollout_pace', None)
        row['closed loop'] = timed(eng, policy, 300, trajectory=True)
        info = eng.rollout_pacing(policy, True)
        row['(its period %s)' % (None if info is None else info['period'])] = 0.0
        eng.set_option('rollout_pace', 0)
        row['no limiter'] = timed(eng, policy, 3, trajectory=True)
        for period in (150, 155, 160, 165, 170, 175, 180, 190):
            eng.set_option('rollout_pace', period)
            row['period %d' % period] = timed(eng, policy, 20, trajectory=True)
        eng.set_option('rollout_pace', None)
        row['statistics only'] = timed(eng, policy, 3, trajectory=False, stats=True)
        print('rollout_rows %-4s %-8s ' % (rows_opt, policy) + '  '.join('%s %.1f' % kv for kv in row.items()), flush=True)
    eng.close()
