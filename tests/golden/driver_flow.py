"""The driver whose results tests/golden/driver_alg_examples.npz holds for the REAL reference (make_golden.py: capture_driver)
and which tests/test_gpu_compat_drivers.py runs on the engine through the compat/ import paths.  Imports nothing but numpy."""
import random
import warnings

import numpy as np


def driver_flow(GridUniverseEnv, utils, dp, mc):
    """The call sequence of examples/griduniverse_alg_examples.py:29-129 (policy evaluation sweeps, greedy improvement, policy
    iteration, value iteration, a greedy run to the goal, then run_episode and monte_carlo_evaluation on a second maze), without
    the window calls, on whatever implementation the four arguments name.  Used twice: here on the REAL reference, and by
    tests/test_gpu_compat_drivers.py on the engine through the compat/ import paths -- every array must be identical."""
    out = {}
    random.seed(2)
    np.random.seed(2)
    world_shape = (11, 11)
    env = GridUniverseEnv(grid_shape=world_shape, random_maze=True)
    out['maze1_walls'] = np.array(sorted(env.wall_indices), dtype=np.int64)
    out['maze1_start_goal'] = np.array([env.initial_state] + list(env.goal_states), dtype=np.int64)
    n_act = len(env.action_state_to_next_state)
    policy0 = np.ones([env.world.size, n_act]) / n_act
    v0 = np.zeros(env.world.size)
    val_fun = v0
    for _ in range(50):
        val_fun = utils.single_step_policy_evaluation(policy0, env, value_function=val_fun)
    out['v_after_50_sweeps'] = np.array(val_fun)
    policy1 = utils.greedy_policy_from_value_function(policy0, env, val_fun)
    out['greedy_policy'] = np.array(policy1)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        policy0 = np.ones([env.world.size, 4]) / 4
        pi_value, pi_policy = dp.policy_iteration(policy0, env, v0, threshold=0.001, max_steps=1000)
        out['pi_value'], out['pi_policy'] = np.array(pi_value), np.array(pi_policy)
        policy0 = np.ones([env.world.size, 4]) / 4
        vi_value, vi_policy = dp.value_iteration(policy0, env, v0, threshold=0.001, max_steps=100)
        out['vi_value'], out['vi_policy'] = np.array(vi_value), np.array(vi_policy)
    curr_state = env.reset()
    run = [curr_state]
    for _ in range(100):
        action = np.argmax(vi_policy[curr_state])
        curr_state, reward, done, info = env.step(action)
        run += [int(action), int(curr_state), int(reward), int(done)]
        if done:
            break
    out['greedy_run'] = np.array(run, dtype=np.int64)
    env = GridUniverseEnv((8, 8), random_maze=True)
    out['maze2_walls'] = np.array(sorted(env.wall_indices), dtype=np.int64)
    policy0 = np.ones([env.world.size, env.action_space.n]) / env.action_space.n
    st_history, rw_history, done = mc.run_episode(policy0, env)
    out['episode_states'] = np.array(st_history, dtype=np.int64)
    out['episode_rewards'] = np.array(rw_history, dtype=np.int64)
    out['episode_done'] = np.array([done], dtype=np.int64)
    value0 = mc.monte_carlo_evaluation(policy0, env, every_visit=True, num_episodes=30)
    out['mc_value'] = np.array(value0)
    out['mc_greedy_policy'] = np.array(utils.greedy_policy_from_value_function(policy0, env, value0))
    out['env_after_mc'] = np.array([env.current_state, env.previous_state, int(env.done)], dtype=np.int64)
    out['streams_after'] = np.array([np.random.random_sample(), random.random()])
    return out
