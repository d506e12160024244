#!/usr/bin/env python3
"""The latency-bound launches of config 3 (65 536 envs, 32x32 generator maze): stats-only and packed-row rollouts, 20 launches
each.  Meant to run under rocprofv3 (kernel trace / PMC) with GU_ROLLOUT_ROWS=0 and =1 to compare the general kernel with the
transition-row kernel.  Usage: python3 tools/latency_modes.py"""
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import griduniverse_amd as gua  # noqa: E402

random.seed(123)
np.random.seed(123)
env = gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True)
eng = gua.Engine(65536, gua.GridSpec.from_env(env), seed=123)
eng.reset()
eng.reserve_trajectory(1000)
for traj, stats in ((False, True), ('packed', False)):
    for _ in range(20):
        eng.rollout(1000, 'uniform', True, traj, stats)
    eng.sync()
eng.close()
