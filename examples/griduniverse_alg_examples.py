"""Headless counterpart of the reference's examples/griduniverse_alg_examples.py:run_policy_and_value_iteration
on the MI355X engine: evaluate the uniform policy, improve it greedily, run policy iteration and value iteration
(float64 sweeps on the device, bit-identical to the reference's Python loops), then let agents act on the found
policy -- one agent through the gym-style facade like the reference, and 4096 agents in one fused launch.

    python examples/griduniverse_alg_examples.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import griduniverse_amd.algorithms.dynamic_programming as dp  # noqa: E402
from griduniverse_amd import GridUniverseEnv, VecGridUniverse  # noqa: E402
from griduniverse_amd.algorithms import utils  # noqa: E402
from griduniverse_amd.algorithms.monte_carlo import monte_carlo_evaluation, run_episode  # noqa: E402


def run_policy_and_value_iteration():
    print('\n*** value and policy iteration on an 11x11 generated maze ***\n')
    world_shape = (11, 11)
    env = GridUniverseEnv(grid_shape=world_shape, random_maze=True)
    env.render()
    n_actions = len(env.action_state_to_next_state)
    policy0 = np.ones([env.world.size, n_actions]) / n_actions
    v0 = np.zeros(env.world.size)

    val_fun = v0
    for _ in range(500):
        val_fun = utils.single_step_policy_evaluation(policy0, env, value_function=val_fun)
    np.set_printoptions(linewidth=150, precision=1, suppress=True)
    print('value of the uniform policy after 500 sweeps:\n', utils.reshape_as_griduniverse(val_fun, world_shape))
    policy1 = utils.greedy_policy_from_value_function(policy0, env, val_fun)
    print('greedy policy from it:')
    utils.get_policy_map(policy1, world_shape)

    print('policy iteration:')
    policy0 = np.ones([env.world.size, n_actions]) / n_actions
    value, policy = dp.policy_iteration(policy0, env, v0, threshold=0.001, max_steps=1000)
    utils.get_policy_map(policy, world_shape)

    print('value iteration:')
    policy0 = np.ones([env.world.size, n_actions]) / n_actions
    value, policy = dp.value_iteration(policy0, env, v0, threshold=0.001, max_steps=100)
    print(utils.reshape_as_griduniverse(value, world_shape))
    utils.get_policy_map(policy, world_shape)

    print('one agent following the value-iteration policy:')
    state = env.reset()
    for t in range(100):
        action = np.argmax(policy[state])
        state, reward, done, info = env.step(action)
        if done:
            print('terminal state reached in {} steps, reward {}'.format(t + 1, reward))
            env.render()
            break

    print('4096 agents following it in one launch:')
    envs = VecGridUniverse(4096, template=env, seed=1)
    envs.engine.vi_set(value, policy)
    envs.reset()
    out = envs.rollout(100, policy='greedy', auto_reset=False)
    print('agents at the goal after 100 steps: {} / 4096'.format(int(out['done'][-1].sum())))
    envs.close()

    print('one stochastic episode under the uniform policy (run_episode):')
    states, rewards, done = run_episode(np.ones([env.world.size, n_actions]) / n_actions, env, max_steps_per_episode=200)
    print('length {}, return {}, terminal {}'.format(len(rewards), sum(rewards), done))


def run_monte_carlo_evaluation():
    """Counterpart of the reference's run_monte_carlo_evaluation (examples/griduniverse_alg_examples.py:88-129):
    every-visit Monte-Carlo evaluation of the uniform policy on an 8x8 maze, then act greedily on it -- with
    4096 episodes in one launch instead of 30 sequential ones."""
    print('\n*** Monte-Carlo evaluation of the uniform policy and greedy policy from it ***\n')
    world_shape = (8, 8)
    env = GridUniverseEnv(world_shape, random_maze=True)
    env.render()
    policy0 = np.ones([env.world.size, env.action_space.n]) / env.action_space.n
    value0 = monte_carlo_evaluation(policy0, env, every_visit=True, num_episodes=4096)
    np.set_printoptions(linewidth=150, precision=1, suppress=True)
    print(utils.reshape_as_griduniverse(value0, world_shape))
    policy1 = utils.greedy_policy_from_value_function(policy0, env, value0)
    utils.get_policy_map(policy1, world_shape)
    state = env.reset()
    for t in range(500):
        state, reward, done, info = env.step(np.argmax(policy1[state]))
        if done:
            print('terminal state found in {} steps'.format(t + 1))
            break
    else:
        print('greedy policy from the Monte-Carlo estimate did not reach a terminal state in 500 steps')


if __name__ == '__main__':
    import random
    random.seed(4)
    np.random.seed(4)
    run_policy_and_value_iteration()
    run_monte_carlo_evaluation()
