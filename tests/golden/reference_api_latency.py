#!/usr/bin/env python3
"""End-to-end latency of the REAL reference's calls (build container only: imports /root/reference), the CPU column
next to tools/api_latency.py's numbers in BASELINE.md.  One core, best of a few repetitions."""
import contextlib
import io
import json
import os
import random
import sys
import time
import warnings

os.environ.setdefault('MPLBACKEND', 'Agg')
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(HERE, 'gym_stub'), os.environ.get('GU_REFERENCE', '/root/reference')]
import numpy as np  # noqa: E402
if not hasattr(np, 'float'):
    np.float = float  # the reference predates NumPy 1.24
from core.envs.griduniverse_env import GridUniverseEnv  # noqa: E402
from core.algorithms import utils  # noqa: E402
from core.algorithms.monte_carlo import run_episode, monte_carlo_evaluation  # noqa: E402
import core.algorithms.dynamic_programming as dp  # noqa: E402


def best(fn, reps=3):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3


def main():
    warnings.simplefilter('ignore')
    out = {}
    random.seed(3)
    np.random.seed(3)
    with contextlib.redirect_stdout(io.StringIO()):
        env = GridUniverseEnv(grid_shape=(32, 32), random_maze=True)
        out['GridUniverseEnv(32x32 maze)_ms'] = best(lambda: GridUniverseEnv(grid_shape=(32, 32), random_maze=True))
    env.reset()
    out['env.reset()_us'] = best(lambda: [env.reset() for _ in range(1000)]) * 1e3 / 1000

    def steps():
        env.reset()
        for _ in range(2000):
            env.step(env.action_space.sample())
    out['env.step()_us'] = best(steps) * 1e3 / 2000
    out['env.look_step_ahead()_us'] = best(lambda: [env.look_step_ahead(5, 1) for _ in range(2000)]) * 1e3 / 2000
    S = env.world.size
    uniform = np.ones((S, 4)) / 4
    out['single_step_policy_evaluation_ms'] = best(lambda: utils.single_step_policy_evaluation(uniform, env, discount_factor=0.9))
    t0 = time.perf_counter()
    pi, v = dp.value_iteration(uniform.copy(), env, discount_factor=0.9)
    out['value_iteration(32x32, gamma 0.9)_ms'] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    dp.policy_iteration(uniform.copy(), env, discount_factor=0.9)
    out['policy_iteration(32x32, gamma 0.9)_ms'] = (time.perf_counter() - t0) * 1e3
    env.reset()
    out['run_episode(optimal policy)_ms'] = best(lambda: run_episode(pi if pi.ndim == 2 else v, env))
    t0 = time.perf_counter()
    monte_carlo_evaluation(uniform, env, num_episodes=10)
    out['monte_carlo_evaluation(num_episodes=10)_ms'] = (time.perf_counter() - t0) * 1e3
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
