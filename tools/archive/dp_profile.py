#!/usr/bin/env python3
"""Tabular DP timing on one MI355X: rounds/s of the fused V1+V2 round and end-to-end value_iteration /
policy_iteration of the reference-named functions.  Usage: python tools/dp_profile.py"""
import json
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import griduniverse_amd as gua  # noqa: E402
from griduniverse_amd.algorithms import dynamic_programming as dp  # noqa: E402


def maze(w, h, k):
    random.seed(k)
    np.random.seed(k)
    return gua.GridUniverseEnv(grid_shape=(w, h), random_maze=True)


def main():
    res = []
    # grids of 4097 .. 65 536 states run twice: GU_VI_CLUSTER=0 (one launch per round) and the default cluster kernel
    for w, cluster in ((8, None), (32, None), (64, None), (101, '0'), (101, '1'), (128, '0'), (128, '1'), (256, '0'), (256, '1'), (400, None)):
        if cluster is None:
            os.environ.pop('GU_VI_CLUSTER', None)
        else:
            os.environ['GU_VI_CLUSTER'] = cluster
        env = maze(w, w, 5)
        S = env.world.size
        pol = np.ones((S, 4)) / 4
        dp.value_iteration(pol.copy(), env, discount_factor=0.9)  # warm-up (engine creation, first launches)
        reps = []
        for _ in range(7):  # best of 7: the call also moves 40 B per state through pageable host arrays
            t0 = time.perf_counter()
            p, v = dp.value_iteration(pol.copy(), env, discount_factor=0.9)
            reps.append(time.perf_counter() - t0)
        dt = min(reps)
        eng = gua.Engine(64, gua.GridSpec.from_env(env), seed=1)
        eng.vi_set(np.zeros(S), pol)
        steps, deltas = eng.vi_run(0.9, 1e-5, 1000)
        eng.vi_set(np.zeros(S), pol)
        t0 = time.perf_counter()
        eng.vi_run(1.0, 1e-5, 6400)  # gamma = 1 never meets the threshold on a maze with walls: 6400 full rounds
        dr = time.perf_counter() - t0
        eng.vi_set(np.zeros(S), pol)
        t0 = time.perf_counter()
        eng.vi_sweep(1.0, 1000, True)
        eng.sync()
        ds = time.perf_counter() - t0
        t0 = time.perf_counter()
        eng.vi_sweep(1.0, 1000, False)
        eng.sync()
        de = time.perf_counter() - t0
        eng.close()
        res.append({'grid': '%dx%d maze' % (w, w), 'states': S, 'GU_VI_CLUSTER': cluster, 'value_iteration_rounds': int(steps), 'value_iteration_ms': dt * 1e3,
                    'vi_run_rounds_per_s': 6400 / dr, 'round_V1_V2_per_s': 1000 / ds, 'sweep_V1_only_per_s': 1000 / de, 'v_digest': float(np.abs(np.asarray(v)).sum()), 'p_digest': float(np.abs(np.asarray(p)).sum())})
        env.close()
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
