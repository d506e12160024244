#!/usr/bin/env python3
"""In-process A/B of a GU_TRAJ_SKEW knob (skew between the obs/reward/done trajectory planes, in int32 elements)
for the bench launch: interleaved rounds, median and min per setting (cdna_hip_programming.md 5.4 rule 24).
Result (profiles/archive/r01c_traj_skew_ab.txt): no setting helps -- every skew sits at 139-145 us with occasional 120 us
samples right after an idle gap (clock boost), so the knob was removed from the library again; this script is kept
as the record of the experiment and no longer has an effect."""
import os
import random
import statistics
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import griduniverse_amd as gua  # noqa: E402

random.seed(123)
np.random.seed(123)
env = gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True)
N, T = 65536, 1000
eng = gua.Engine(N, gua.GridSpec.from_env(env), seed=123)
eng.reset()
skews = [0, 32, 64, 256, 1024, 2048, 4096, 16384, 65536, 262144, 1048576]
res = {s: [] for s in skews}
for rnd in range(6):
    order = skews[:] if rnd % 2 == 0 else skews[::-1]
    for s in order:
        os.environ['GU_TRAJ_SKEW'] = str(s)
        eng.reserve_trajectory(1)
        eng.reserve_trajectory(T)
        for _ in range(3):
            eng.rollout(T, 'uniform', True, True)
        eng.sync()
        eng.timer_begin()
        for _ in range(30):
            eng.rollout(T, 'uniform', True, True)
        res[s].append(eng.timer_end() / 30 * 1e3)
for s in skews:
    v = res[s]
    print('skew %8d  median %.1f us  min %.1f  max %.1f   %s' % (s, statistics.median(v), min(v), max(v), ' '.join('%.1f' % x for x in v)))
