#!/usr/bin/env python3
"""Timing of the int32-row launches of the transition-row kernel (config 2, a config-4 shard, the sampled policy at config 3),
for an A/B of two builds of the library (GU_LIB_PATH): us per launch of 1000 steps after the store limiter has settled.
    python tools/rows_timing.py [--settle 300] [--reps 5] [--launches 20]"""
import argparse
import random
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit('/', 2)[0])
import griduniverse_amd as gua  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--settle', type=int, default=300)
ap.add_argument('--reps', type=int, default=5)
ap.add_argument('--launches', type=int, default=20)
args = ap.parse_args()
T = 1000


def workload(name):
    random.seed(123)
    np.random.seed(123)
    if name == 'c2':
        return gua.GridUniverseEnv(grid_shape=(8, 8)), 4096
    if name == 'c4':
        return gua.GridUniverseEnv(grid_shape=(32, 32), lava_states=[16 + 32 * r for r in range(24)], goal_states=[1023]), 32768
    return gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True), 65536


for wname, policy, n_over, opts in (('c2', 'uniform', None, dict(traj_layout=0)), ('c2', 'uniform', 16384, dict(traj_layout=0)), ('c4', 'uniform', None, {}),
                                    ('c3', 'uniform', 32768, {}), ('c3', 'sample', None, {}), ('c3', 'greedy', None, {}), ('c3', 'sample', 32768, {})):
    env, N = workload(wname)
    N = n_over or N
    spec = gua.GridSpec.from_env(env)
    S = spec.W * spec.H
    eng = gua.Engine(N, spec, seed=5)
    for k, v in opts.items():
        eng.set_option(k, v)
    eng.reset()
    eng.reserve_trajectory(T)
    if policy in ('sample', 'greedy'):
        eng.vi_set(np.zeros(S), np.random.RandomState(1).dirichlet(np.ones(4), S))
    for _ in range(args.settle):
        eng.rollout(T, policy, True, True)
    times = []
    for _ in range(args.reps):
        eng.sync()
        eng.timer_begin()
        for _ in range(args.launches):
            eng.rollout(T, policy, True, True)
        times.append(eng.timer_end() / args.launches * 1e3)
    pacing = eng.rollout_pacing(policy, True)
    print('%-3s %-8s %6d envs: median %.2f us  min %.2f  (%s)  pacing %s' % (wname, policy, N, float(np.median(times)), min(times), ' '.join('%.1f' % t for t in times),
                                                                           None if pacing is None else pacing.get('period')), flush=True)
    eng.close()
