#!/usr/bin/env python3
"""One maze per env (rollout MAP 5, gu_rollout.hpp) at 65 536 envs on 32x32: us per 1000-step launch with int32 rows (closed-loop
pacing / no limiter / fixed periods), packed rows and statistics only (no row stores: the length of the step chain itself), next to
the shared-maze launch of the same size.  Usage (GPU box): python tools/multigrid_ab.py [reps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import griduniverse_amd as gua  # noqa: E402

N, T, K = 65536, 1000, 20
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3


def timed(eng, settle=200, **kw):
    for _ in range(settle):
        eng.rollout(T, 'uniform', auto_reset=True, **kw)
    eng.sync()
    eng.timer_begin()
    for _ in range(K):
        eng.rollout(T, 'uniform', auto_reset=True, **kw)
    return eng.timer_end() / K * 1e3


for G in (65536, 1024, 1):
    eng = gua.Engine(N, gua.GridSpec(32, 32, [0], [1023], [], []), seed=123)
    eng.generate_mazes(G, 32, 32, 2026)
    eng.reset()
    eng.reserve_trajectory(T)
    for rep in range(reps):
        row = {}
        eng.set_option('rollout_pace', None)
        row['rows, closed loop'] = timed(eng, trajectory=True)
        eng.set_option('rollout_pace', 0)
        row['rows, no limiter'] = timed(eng, 3, trajectory=True)
        info = eng.rollout_pacing('uniform', True)
        row['(loop period %s)' % (None if info is None else info['period'])] = 0.0
        for period in (150, 160, 170, 180, 190, 200):
            eng.set_option('rollout_pace', period)
            row['rows, period %d' % period] = timed(eng, 3, trajectory=True)
        eng.set_option('rollout_pace', None)
        row['packed rows'] = timed(eng, 3, trajectory='packed')
        row['statistics only'] = timed(eng, 3, trajectory=False, stats=True)
        print('G = %5d  ' % G + '  '.join('%s %.1f' % kv for kv in row.items()), flush=True)
    eng.close()
