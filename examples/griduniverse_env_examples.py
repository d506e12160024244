"""Headless counterparts of the reference's examples/griduniverse_env_examples.py, on the MI355X engine.

Same four demos (default grid, level file, random maze, lava column), random agent, ASCII rendering instead of the
pyglet window -- followed by the batched form of the same loop, which is what the engine is for.

    python examples/griduniverse_env_examples.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from griduniverse_amd import GridUniverseEnv, VecGridUniverse  # noqa: E402

# a level in the reference's text format (env:253-268): o floor, # wall, G goal, L lava, x possible start
DEMO_LEVEL = """
x o #
x o #
o o #
o o L
o o L
o o L
o o o
o G o
"""


def _random_agent(env, episodes, max_steps, render_every=0):
    for _ in range(episodes):
        env.reset()
        for t in range(max_steps):
            if render_every and t % render_every == 0:
                env.render()
            action = env.action_space.sample()
            observation, reward, done, info = env.step(action)
            if done:
                print("Episode finished after {} timesteps, final reward {}".format(t + 1, reward))
                break
        else:
            print("No terminal state within {} steps (ended in state {})".format(max_steps, observation))


def run_default_griduniverse():
    print('\n*** random agent on the default 4x4 GridUniverse ***\n')
    _random_agent(GridUniverseEnv(), 1, 100, render_every=1)


def run_griduniverse_from_text_file():
    print('\n*** random agent on a level loaded from a text file ***\n')
    import tempfile
    with tempfile.NamedTemporaryFile('w', suffix='.txt', delete=False) as f:
        f.write(DEMO_LEVEL)
    _random_agent(GridUniverseEnv(custom_world_fp=f.name), 1, 1000, render_every=50)
    os.unlink(f.name)


def run_random_maze():
    print('\n*** random agent on a generated 11x11 maze ***\n')
    env = GridUniverseEnv(grid_shape=(11, 11), random_maze=True)
    env.render()
    _random_agent(env, 1, 1000)


def run_griduniverse_with_lava():
    print('\n*** random agent on a 10x10 grid with a lava column ***\n')
    _random_agent(GridUniverseEnv(grid_shape=(10, 10), lava_states=[4, 14, 24, 34, 44, 54, 64, 74]), 5, 100)


def run_batched_random_agents(num_envs=65536, steps=1000):
    print('\n*** {} random agents x {} steps on the lava grid, one kernel launch ***\n'.format(num_envs, steps))
    envs = VecGridUniverse(num_envs, grid_shape=(10, 10), lava_states=[4, 14, 24, 34, 44, 54, 64, 74], seed=0, auto_reset=True)
    envs.reset()
    out = envs.rollout(steps, stats=True)
    finished = out['episodes'].sum()
    print('episodes finished: {}, of which in lava: {}, mean reward per step: {:.3f}'.format(
        finished, int((out['reward'] == -10).sum()), out['reward'].mean()))
    envs.close()


if __name__ == '__main__':
    np.random.seed(0)
    run_default_griduniverse()
    run_griduniverse_from_text_file()
    run_random_maze()
    run_griduniverse_with_lava()
    run_batched_random_agents()
