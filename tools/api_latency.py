#!/usr/bin/env python3
"""End-to-end latency of the reference-named calls (what a user switching from the reference sees), on one MI355X.
Usage: python tools/api_latency.py"""
import json
import os
import random
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import griduniverse_amd as gua  # noqa: E402
from griduniverse_amd.algorithms import dynamic_programming as dp  # noqa: E402
from griduniverse_amd.algorithms import monte_carlo as mc  # noqa: E402
from griduniverse_amd.algorithms import utils  # noqa: E402


def best(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3


def main():
    warnings.simplefilter('ignore')
    out = {}
    t0 = time.perf_counter()
    random.seed(3)
    np.random.seed(3)
    env = gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True)
    env.reset()
    out['first_env_ms (library load, context, maze, engine)'] = (time.perf_counter() - t0) * 1e3
    out['GridUniverseEnv(32x32 maze)_ms'] = best(lambda: gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True).close())
    out['env.reset()_us'] = best(env.reset, 20) * 1e3

    def steps():
        env.reset()
        for _ in range(200):
            env.step(env.action_space.sample())
    out['env.step()_us'] = best(steps) * 1e3 / 200
    def plain_steps():
        env.reset()
        for k in range(200):
            env.step(k & 3)
    out['env.step()_us (fixed actions, no action_space.sample())'] = best(plain_steps) * 1e3 / 200

    def device_steps():
        env.reset()
        for _ in range(200):
            env.step_on_device(env.action_space.sample())
    out['env.step_on_device()_us (kernel launch per scalar step)'] = best(device_steps) * 1e3 / 200
    out['env.look_step_ahead()_us'] = best(lambda: [env.look_step_ahead(5, 1) for _ in range(200)]) * 1e3 / 200
    S = env.world.size
    uniform = np.ones((S, 4)) / 4
    out['single_step_policy_evaluation_ms'] = best(lambda: utils.single_step_policy_evaluation(uniform, env, 0.9))
    out['value_iteration(32x32, gamma 0.9)_ms'] = best(lambda: dp.value_iteration(uniform.copy(), env, discount_factor=0.9))
    out['policy_iteration(32x32, gamma 0.9)_ms'] = best(lambda: dp.policy_iteration(uniform.copy(), env, discount_factor=0.9))
    v, pi = dp.value_iteration(uniform.copy(), env, discount_factor=0.9)
    out['run_episode(optimal policy)_ms'] = best(lambda: mc.run_episode(pi, env))
    for n in (100, 4096):
        out['monte_carlo_evaluation(num_episodes=%d)_ms' % n] = best(lambda: mc.monte_carlo_evaluation(uniform, env, num_episodes=n))
    vec = gua.VecGridUniverse(65536, template=env, seed=1, auto_reset=True)
    vec.reset()
    acts = np.random.randint(0, 4, 65536).astype(np.int32)
    out['VecGridUniverse(65536).step()_us'] = best(lambda: vec.step(acts), 20) * 1e3
    out['VecGridUniverse(65536).rollout(1000, trajectory=False)_ms'] = best(lambda: (vec.rollout(1000, trajectory=False), vec.engine.sync()))
    out['VecGridUniverse(65536).step(zero_copy=True)_us'] = best(lambda: vec.step(acts, zero_copy=True), 20) * 1e3
    eng = vec.engine
    eng.pinned_actions[:] = acts
    out['Engine(65536).step_pinned()_us (gu_step, page-locked I/O, no numpy allocation)'] = best(lambda: eng.step_pinned(True), 50) * 1e3
    out['Engine(65536).done_indices()_us (one compaction launch)'] = best(eng.done_indices, 50) * 1e3
    small = gua.VecGridUniverse(4096, template=env, seed=1, auto_reset=True)
    small.reset()
    small.engine.pinned_actions[:] = acts[:4096]
    out['Engine(4096).step_pinned()_us'] = best(lambda: small.engine.step_pinned(True), 50) * 1e3
    out['VecGridUniverse(4096).step()_us'] = best(lambda: small.step(acts[:4096]), 50) * 1e3
    out['Engine(4096).done_indices()_us'] = best(small.engine.done_indices, 50) * 1e3
    small.close()
    vec.close()
    env.close()
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
