#!/usr/bin/env python3
"""Time the real reference step() loop and the oracle's per-instance port side by side (build
container only; fills the calibration row of BASELINE.md)."""
import contextlib
import io
import os
import random
import sys
import time

os.environ.setdefault('MPLBACKEND', 'Agg')
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [os.path.join(HERE, 'gym_stub'), os.environ.get('GU_REFERENCE', '/root/reference'), REPO]
import numpy as np  # noqa: E402
from core.envs.griduniverse_env import GridUniverseEnv  # noqa: E402
from oracle.ref_env import OracleGridUniverseEnv  # noqa: E402


def loop(env, actions):
    env.reset()
    t0 = time.perf_counter()
    for a in actions:
        if env.step(a)[2]:
            env.reset()
    return len(actions) / (time.perf_counter() - t0)


def main():
    random.seed(123)
    np.random.seed(123)
    with contextlib.redirect_stdout(io.StringIO()):
        ref = GridUniverseEnv(grid_shape=(32, 32), random_maze=True)
    port = OracleGridUniverseEnv(grid_shape=(32, 32), initial_state=list(ref.starting_states), goal_states=list(ref.goal_states),
                                 walls=list(ref.wall_indices))
    actions = [int(a) for a in np.random.RandomState(0).randint(0, 4, 200000)]
    r = [(loop(ref, actions), loop(port, actions)) for _ in range(3)]
    ref_rate, port_rate = max(x[0] for x in r), max(x[1] for x in r)
    print('reference step(): %.0f steps/s/core   oracle port: %.0f steps/s/core   port/reference = %.2f' % (ref_rate, port_rate, port_rate / ref_rate))


if __name__ == '__main__':
    main()
