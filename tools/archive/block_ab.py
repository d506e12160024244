#!/usr/bin/env python3
"""In-process A/B of the rollout workgroup size on the bench workload (config 3: 65 536 envs, 32x32 maze, 1000 steps per
launch, trajectory written): GU_ROLLOUT_BLOCK is read per launch.  Usage: python tools/block_ab.py"""
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import griduniverse_amd as gua  # noqa: E402

random.seed(123)
np.random.seed(123)
env = gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True)
N, T = 65536, 1000
eng = gua.Engine(N, gua.GridSpec.from_env(env), seed=0)
eng.reset()
eng.reserve_trajectory(T)
for rnd in range(4):
    row = []
    for bs in (256, 128, 64):
        os.environ['GU_ROLLOUT_BLOCK'] = str(bs)
        for _ in range(3):
            eng.rollout(T, 'uniform', True, True)
        eng.sync()
        eng.timer_begin()
        for _ in range(50):
            eng.rollout(T, 'uniform', True, True)
        ms = eng.timer_end()
        row.append('block %3d: %6.1f us %.2f TB/s' % (bs, ms / 50 * 1e3, 12 * N * T * 50 / (ms / 1e3) / 1e12))
    print(' | '.join(row))
