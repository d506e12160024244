"""A driver written against the REFERENCE's import paths and call sequence (core.envs..., core.algorithms...; the
flow of examples/griduniverse_alg_examples.py:29-129) runs unmodified on the engine through compat/."""
import importlib
import os
import random
import sys
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_style_driver_through_the_import_shim(capsys):
    sys.path.insert(0, os.path.join(ROOT, 'compat'))
    try:
        GridUniverseEnv = importlib.import_module('core.envs.griduniverse_env').GridUniverseEnv
        mc = importlib.import_module('core.algorithms.monte_carlo')
        utils = importlib.import_module('core.algorithms.utils')
        dp = importlib.import_module('core.algorithms.dynamic_programming')
        random.seed(2)
        np.random.seed(2)
        world_shape = (11, 11)
        env = GridUniverseEnv(grid_shape=world_shape, random_maze=True)
        policy0 = np.ones([env.world.size, len(env.action_state_to_next_state)]) / len(env.action_state_to_next_state)
        v0 = np.zeros(env.world.size)
        val_fun = v0
        for _ in range(50):
            val_fun = utils.single_step_policy_evaluation(policy0, env, value_function=val_fun)
        policy1 = utils.greedy_policy_from_value_function(policy0, env, val_fun)
        utils.get_policy_map(policy1, world_shape)
        policy0 = np.ones([env.world.size, 4]) / 4
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            optimal_value, optimal_policy = dp.policy_iteration(policy0, env, v0, threshold=0.001, max_steps=1000)
            policy0 = np.ones([env.world.size, 4]) / 4
            optimal_value, optimal_policy = dp.value_iteration(policy0, env, v0, threshold=0.001, max_steps=100)
            curr_state = env.reset()
            env.render_policy_arrows(optimal_policy)
            for t in range(100):
                env.render(mode='graphic')
                action = np.argmax(optimal_policy[curr_state])
                curr_state, reward, done, info = env.step(action)
                if done:
                    env.render(mode='graphic')
                    env.render(close=True)
                    break
        assert done and reward == 10 and curr_state in env.goal_states
        # Monte-Carlo part (examples/griduniverse_alg_examples.py:88-129)
        env = GridUniverseEnv((8, 8), random_maze=True)
        policy0 = np.ones([env.world.size, env.action_space.n]) / env.action_space.n
        st_history, rw_history, done = mc.run_episode(policy0, env)
        assert len(st_history) == len(rw_history) + 1
        value0 = mc.monte_carlo_evaluation(policy0, env, every_visit=True, num_episodes=30)
        policy1 = utils.greedy_policy_from_value_function(policy0, env, value0)
        assert value0.shape == (64,) and policy1.shape == (64, 4) and np.isfinite(value0).all()
        out = capsys.readouterr().out
        assert '→' in out or '↓' in out  # the arrow maps were printed
    finally:
        sys.path.pop(0)
        for k in [k for k in sys.modules if k == 'core' or k.startswith('core.')]:
            del sys.modules[k]


def test_whole_driver_through_the_import_shim_equals_the_reference_run():
    """tests/golden/driver_flow.py -- policy-evaluation sweeps, greedy improvement, policy iteration, value iteration, a greedy
    run, run_episode and monte_carlo_evaluation on two seeded generator mazes, the flow of examples/griduniverse_alg_examples.py
    -- was run on the REAL reference (make_golden.py: capture_driver).  The same function on the engine, through the reference's
    own import paths (compat/): every array it produces is byte-identical -- mazes, float64 tables, trajectories, the Monte-Carlo
    value function (the shim defaults to the reference's random draws), where the env is left, where both global streams are."""
    sys.path.insert(0, os.path.join(ROOT, 'compat'))
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    np_state, py_state = np.random.get_state(), random.getstate()
    try:
        from driver_flow import driver_flow
        GridUniverseEnv = importlib.import_module('core.envs.griduniverse_env').GridUniverseEnv
        mc = importlib.import_module('core.algorithms.monte_carlo')
        utils = importlib.import_module('core.algorithms.utils')
        dp = importlib.import_module('core.algorithms.dynamic_programming')
        got = driver_flow(GridUniverseEnv, utils, dp, mc)
        want = np.load(os.path.join(ROOT, 'tests', 'golden', 'driver_alg_examples.npz'))
        assert sorted(got) == sorted(want.files)
        for k in want.files:
            assert got[k].dtype == want[k].dtype and got[k].tobytes() == want[k].tobytes(), k
        assert want['greedy_run'][-1] == 1 and want['episode_states'].size > 10  # (the run reached the goal; a real episode)
    finally:
        np.random.set_state(np_state)
        random.setstate(py_state)
        sys.path.pop(0)
        sys.path.pop(0)
        for k in [k for k in sys.modules if k == 'core' or k.startswith('core.') or k == 'driver_flow']:
            del sys.modules[k]
