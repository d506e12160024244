#!/usr/bin/env python3
"""Rollout throughput on grids beyond the two-plane LDS limit (32 767 cells): flags plane alone in LDS up to ~160 000
cells, L2 records beyond.  Usage: python tools/big_grid_bench.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import griduniverse_amd as gua  # noqa: E402


def main():
    res = []
    N, T = 65536, 500
    for W in (128, 181, 256, 400, 512):
        rs = np.random.RandomState(W)
        S = W * W
        spec = gua.GridSpec(W, W, [0], [S - 1], [], [int(x) for x in rs.choice(np.arange(1, S - 1), S // 5, replace=False)])
        eng = gua.Engine(N, spec, seed=1)
        eng.reset()
        eng.reserve_trajectory(T)
        row = {'grid': '%dx%d' % (W, W), 'cells': S}
        for name, traj in (('traj', True), ('stats_only', False)):
            eng.rollout(T, 'uniform', True, traj, not traj)
            eng.sync()
            eng.timer_begin()
            for _ in range(5):
                eng.rollout(T, 'uniform', True, traj, not traj)
            ms = eng.timer_end()
            row['uniform_%s_steps_per_s' % name] = N * T * 5 / (ms / 1e3)
        eng.close()
        res.append(row)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
