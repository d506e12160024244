#!/usr/bin/env python3
"""monte_carlo_evaluation under the reference's own random draws (rng='numpy') against the device-RNG mode: seconds per call on the
32x32 generator maze of BASELINE.md section 5 (seed 3, uniform policy), 10 / 100 / 1000 episodes."""
import random
import sys
import time

import numpy as np

sys.path.insert(0, __file__.rsplit('/', 2)[0])
import griduniverse_amd as gua  # noqa: E402
from griduniverse_amd.algorithms import monte_carlo as mc  # noqa: E402

random.seed(3)
np.random.seed(3)
env = gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True)
policy = np.ones((env.world.size, 4)) / 4
for episodes in (10, 100, 1000):
    for rng in ('numpy', 'device'):
        mc.monte_carlo_evaluation(policy, env, num_episodes=episodes, rng=rng)  # warm (engines, calibration)
        random.seed(5)
        np.random.seed(5)
        t0 = time.perf_counter()
        v = mc.monte_carlo_evaluation(policy, env, num_episodes=episodes, rng=rng)
        dt = time.perf_counter() - t0
        print('%5d episodes  rng=%-6s  %9.2f ms   (%.1f us per episode)   mean V %.3f' % (episodes, rng, dt * 1e3, dt / episodes * 1e6, float(v.mean())))
