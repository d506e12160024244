#!/usr/bin/env python3
"""The sampled-policy rollout at config 3 with 4 / 2 / 1 copies of the transition rows in LDS (option rows_copies), statistics only and with
int32 rows: do LDS bank conflicts matter to it?  (They do not: 55.6 .. 56.7 us either way, profiles/archive/r05zz_sample_copies.txt.)
    python tools/sample_copies.py"""
import sys, random, numpy as np
sys.path.insert(0, '.')
import griduniverse_amd as gua
random.seed(123); np.random.seed(123)
env = gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True)
spec = gua.GridSpec.from_env(env); S = spec.W * spec.H
for rep in range(2):
  for copies in (4, 2, 1):
    for traj in (False, True):
        eng = gua.Engine(65536, spec, seed=5); eng.set_option('rows_copies', copies); eng.reset(); eng.reserve_trajectory(1000)
        eng.vi_set(np.zeros(S), np.random.RandomState(1).dirichlet(np.ones(4), S))
        for _ in range(250 if traj else 20): eng.rollout(1000, 'sample', True, traj)
        ts = []
        for _ in range(5):
            eng.sync(); eng.timer_begin()
            for _ in range(20): eng.rollout(1000, 'sample', True, traj)
            ts.append(eng.timer_end() / 20 * 1e3)
        print('sample 65536 envs copies %d traj=%s: median %.2f us min %.2f' % (copies, traj, float(np.median(ts)), min(ts)), flush=True)
        eng.close()
