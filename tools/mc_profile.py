#!/usr/bin/env python3
"""Monte-Carlo evaluation timing (run under rocprofv3 --kernel-trace --stats for the per-kernel split).
Usage: python tools/mc_profile.py [reps]"""
import json
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import griduniverse_amd as gua  # noqa: E402
from griduniverse_amd.algorithms.monte_carlo import discount_table  # noqa: E402


def maze(w, h, k):
    random.seed(k)
    np.random.seed(k)
    return gua.GridUniverseEnv(grid_shape=(w, h), random_maze=True)


def run(w, N, T, reps, every_visit, incremental, stationary):
    env = maze(w, w, 1)
    S = env.world.size
    eng = gua.Engine(N, gua.GridSpec.from_env(env), seed=3)
    eng.vi_set(np.zeros(S), np.ones((S, 4)) / 4)
    first = eng.reset()
    eng.reserve_trajectory(T)
    eng.rollout(T, 'sample', False, True)
    eng.sync()
    pw, keep = discount_table(0.99, 1e-4, T)
    eng.mc_evaluate(T, first, pw, keep, every_visit, incremental, stationary, 0.001)
    t0 = time.perf_counter()
    for _ in range(reps):
        v, visits = eng.mc_evaluate(T, first, pw, keep, every_visit, incremental, stationary, 0.001)
    dt = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.rollout(T, 'sample', False, True)
    eng.sync()
    dr = (time.perf_counter() - t0) / reps
    eng.close()
    return {'grid': '%dx%d maze' % (w, w), 'episodes': N, 'max_steps': T, 'every_visit': every_visit,
            'incremental_mean': incremental, 'stationary_env': stationary, 'evaluate_ms': dt * 1e3,
            'rollout_ms': dr * 1e3, 'episodes_per_s': N / (dt + dr), 'value_digest': float(np.abs(v).sum())}


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    res = [run(8, 4096, 1000, reps, True, True, True), run(8, 4096, 1000, reps, False, False, True),
           run(8, 4096, 1000, reps, True, True, False), run(32, 4096, 1000, reps, True, True, True),
           run(32, 65536, 1000, max(1, reps // 5), True, True, True)]
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
