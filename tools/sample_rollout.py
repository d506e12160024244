#!/usr/bin/env python3
"""The sampled-policy rollout (actions ~ pi[s] by inverse CDF on RNG stream 2) on the headline workload, beside the uniform
policy on the same bytes: us per 65 536 x 1000-step launch with int32 rows, packed rows and statistics only.
Usage: python tools/sample_rollout.py [out.json]"""
import json
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import griduniverse_amd as gua  # noqa: E402


def main():
    random.seed(123)
    np.random.seed(123)
    env = gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True)
    N, T, S = 65536, 1000, 1024
    out = {'workload': 'c3: %d envs x %d steps, 32x32 generator maze seed 123, auto-reset' % (N, T)}
    eng = gua.Engine(N, gua.GridSpec.from_env(env), seed=123)
    eng.reset()
    eng.reserve_trajectory(T)
    eng.vi_set(np.zeros(S), np.random.RandomState(1).dirichlet(np.ones(4), S))
    for policy in ('uniform', 'sample', 'greedy'):
        for name, kw in (('int32_rows', dict(trajectory=True)), ('packed_rows', dict(trajectory='packed')), ('stats_only', dict(trajectory=False, stats=True))):
            eng.calibrate_rollout(T, policy, True, **kw)  # (the store-pacing search up front, as bench.py asks for it)
            for _ in range(3):
                eng.rollout(T, policy, True, **kw)
            reps = []
            for _ in range(5):
                eng.sync()
                eng.timer_begin()
                for _ in range(20):
                    eng.rollout(T, policy, True, **kw)
                reps.append(eng.timer_end() / 20 * 1e3)
            us = float(np.median(reps))
            bytes_per = {'int32_rows': 12, 'packed_rows': 4, 'stats_only': 0}[name]
            out['%s_%s' % (policy, name)] = {'us_per_launch': us, 'env_steps_per_s': N * T / us * 1e6,
                                              'GBps': bytes_per * N * T / us / 1e3, 'frac_of_8TBps': bytes_per * N * T / us / 1e3 / 8000}
    eng.close()
    text = json.dumps(out, indent=1)
    print(text)
    if len(sys.argv) > 1:
        open(sys.argv[1], 'w').write(text + '\n')


if __name__ == '__main__':
    main()
