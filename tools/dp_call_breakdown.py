#!/usr/bin/env python3
"""Where the 0.21 ms of dp.value_iteration (32x32 maze, 111 rounds) go: vi_set, vi_run, vi_get, the Python around them."""
import random
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit('/', 2)[0])
import griduniverse_amd as gua  # noqa: E402
from griduniverse_amd.algorithms import dynamic_programming as dp  # noqa: E402
from griduniverse_amd.algorithms.utils import engine_of  # noqa: E402

random.seed(1)
np.random.seed(1)
env = gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True)
S = env.world.size
uniform = np.ones((S, 4)) / 4
eng = engine_of(env)


def best(f, n=200):
    f()
    t = []
    for _ in range(n):
        t0 = time.perf_counter()
        f()
        t.append(time.perf_counter() - t0)
    return min(t) * 1e6, float(np.median(t)) * 1e6


v0 = np.zeros(S)
print('vi_set            us (min, median):', best(lambda: eng.vi_set(v0, uniform)))
eng.vi_set(v0, uniform)
print('vi_run 111 rounds us:', best(lambda: (eng.vi_set(v0, uniform), eng.vi_run(0.9, 1e-5, 1000))[1]))
print('vi_run 1 round    us:', best(lambda: eng.vi_run(0.9, 1e9, 1000)))
print('vi_sweep 1 round  us:', best(lambda: eng.vi_sweep(0.9, 1, True)))
print('vi_get            us:', best(lambda: eng.vi_get()))
print('sync              us:', best(lambda: eng.sync()))
print('dp.value_iteration us:', best(lambda: dp.value_iteration(uniform.copy(), env, discount_factor=0.9)))
print('dp.policy_iteration us:', best(lambda: dp.policy_iteration(uniform.copy(), env, discount_factor=0.9), 50))
