#!/usr/bin/env python3
"""Trajectory buffer with the default memory type against MTYPE_UC (GU_TRAJ_UNCACHED=0/1), engines of both kinds alive in one
process: rollout launch time (int32 rows, packed rows; 65 536 and 262 144 envs), what the probe of the chosen allocation said,
the device-to-host copy of a trajectory, and the Monte-Carlo evaluation that READS the trajectory on the device.
Usage: python tools/uncached_ab.py"""
import os
import random
import statistics
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import griduniverse_amd as gua  # noqa: E402
from griduniverse_amd.algorithms.monte_carlo import discount_table  # noqa: E402


def launch_us(eng, T, **kw):
    ts = []
    for _ in range(3):
        for _ in range(3):
            eng.rollout(T, 'uniform', True, **kw)
        eng.sync()
        eng.timer_begin()
        for _ in range(20):
            eng.rollout(T, 'uniform', True, **kw)
        ts.append(eng.timer_end() / 20 * 1e3)
    return statistics.median(ts)


random.seed(123)
np.random.seed(123)
env = gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True)
spec = gua.GridSpec.from_env(env)
S = env.world.size
for rnd in range(3):
    for N, T in ((65536, 1000), (262144, 500)):
        row = []
        for unc in ('0', '1'):
            os.environ['GU_TRAJ_UNCACHED'] = unc
            eng = gua.Engine(N, spec, seed=1)
            eng.reset()
            eng.reserve_trajectory(T)
            n, kept, worst = eng.trajectory_placement()
            t_rows = launch_us(eng, T, trajectory=True)
            t_packed = launch_us(eng, T, trajectory='packed')
            eng.rollout(T, 'uniform', True, trajectory=True)
            eng.sync()
            t0 = time.perf_counter()
            eng.read_trajectory(0, min(T, 200))
            t_read = (time.perf_counter() - t0) * 1e3
            row.append('uncached=%s: probe kept %.1f / worst %.1f us (%d candidates)  int32 rows %.1f us  packed %.1f us  D2H of 200 rows %.1f ms'
                       % (unc, kept * 1e3, worst * 1e3, n, t_rows, t_packed, t_read))
            eng.close()
        print('round %d N %6d T %4d\n   %s\n   %s' % (rnd, N, T, row[0], row[1]), flush=True)
# Monte-Carlo evaluation reads the trajectory on the device
for unc in ('0', '1', '0', '1'):
    os.environ['GU_TRAJ_UNCACHED'] = unc
    N, T = 4096, 1000
    eng = gua.Engine(N, spec, seed=3)
    eng.vi_set(np.zeros(S), np.ones((S, 4)) / 4)
    first = eng.reset()
    eng.reserve_trajectory(T)
    eng.rollout(T, 'sample', False, True)
    eng.sync()
    pw, keep = discount_table(0.99, 1e-4, T)
    eng.mc_evaluate(T, first, pw, keep, True, True, True, 0.001)
    t0 = time.perf_counter()
    for _ in range(20):
        v, visits = eng.mc_evaluate(T, first, pw, keep, True, True, True, 0.001)
    dt = (time.perf_counter() - t0) / 20
    t0 = time.perf_counter()
    for _ in range(20):
        eng.rollout(T, 'sample', False, True)
    eng.sync()
    dr = (time.perf_counter() - t0) / 20
    print('uncached=%s: mc_evaluate of 4096 episodes %.3f ms, sampled rollout %.3f ms, |v| %.6f' % (unc, dt * 1e3, dr * 1e3, float(np.abs(v).sum())), flush=True)
    eng.close()
