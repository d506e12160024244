"""Tabular-RL helpers with the reference's names and semantics (core/algorithms/utils.py),
running the sweeps on the MI355X through libgu (kernels in csrc/gu_vi.hip).

`env` is a `GridUniverseEnv` -- the algorithms read `env.world`, `env.reward_matrix` and the grid lists like the
reference's do, which the batched `VecGridUniverse` does not carry; results are bit-identical float64 to the reference's
Python loops (tests/test_gpu_dp.py).
"""
import sys
from io import StringIO

import numpy as np


def engine_of(env):
    """The libgu engine behind a `GridUniverseEnv` (or behind any object that exposes one as `.engine` AND the
    reference env's attributes the algorithm at hand reads)."""
    if hasattr(env, '_engine'):
        return env._engine()
    if hasattr(env, 'engine'):
        return env.engine
    raise TypeError('env must be a griduniverse_amd GridUniverseEnv')


def reshape_as_griduniverse(input_matrix, world_shape):
    """core/algorithms/utils.py:7-12 -- reshapes to (world_shape[0], world_shape[1]); for non-square
    grids that is (W, H), as in the reference."""
    return np.reshape(input_matrix, (world_shape[0], world_shape[1]))


def single_step_policy_evaluation(policy, env, discount_factor=1.0, value_function=None):
    """One synchronous Bellman expectation sweep (utils.py:15-27): returns the new value array."""
    eng = engine_of(env)
    v = np.zeros(env.world.size) if value_function is None else value_function
    eng.vi_set(v, policy)
    eng.vi_sweep(discount_factor, 1, greedy_update=False)
    return eng.vi_get()[0]


def greedy_policy_from_value_function(policy, env, value_function, discount_factor=1.0):
    """Tie-aware greedy policy (utils.py:55-72).  Like the reference it overwrites and returns
    the `policy` array it was given."""
    eng = engine_of(env)
    eng.vi_set(value_function, policy)
    eng.vi_greedy(discount_factor)
    policy[...] = eng.vi_get()[1]
    return policy


def get_policy_map(policy, world_shape, mode='human'):
    """Arrow view of a policy (utils.py:30-52): every action with probability > 0 contributes its
    arrow.  Returns (arrow strings reshaped like the grid, probabilities reshaped like the grid)."""
    arrows = u'↑→↓←'  # up, right, down, left
    cells = np.array([''.join(arrows[a] for a in np.flatnonzero(np.around(row, 8) > 0)) for row in policy], dtype='<U4')
    probs = np.fromiter((tuple(row) for row in policy), dtype='float64, float64, float64, float64')
    out = StringIO() if mode == 'ansi' else sys.stdout
    for row in reshape_as_griduniverse(cells, world_shape):
        out.write(''.join(cell + u'  ' for cell in row) + '\n')
    out.write('\n')
    return cells, reshape_as_griduniverse(probs, world_shape)
