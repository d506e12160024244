"""Scalar, per-instance CPU restatement of the reference env (test infrastructure).

Follows TheMTank/GridUniverse `core/envs/griduniverse_env.py` (cited below as
env:LINE).  One Python object per env instance, one Python call per env-step,
exactly like the reference -- this is deliberately the *slow* formulation: it
is (a) the semantic oracle for the HIP kernels and (b) the "port" CPU baseline
that bench.py times on the GPU box, where the reference itself cannot travel.

Pinned against golden vectors captured from the real reference
(tests/golden/make_golden.py -> tests/golden/*.json|npz) by tests/test_oracle_env.py.
"""
import random
import sys
from io import StringIO

import numpy as np

from . import maze as _maze

UP, RIGHT, DOWN, LEFT = 0, 1, 2, 3


class UnsupportedMode(Exception):
    """Stand-in for gym.error.UnsupportedMode (env:229-230 falls through to gym)."""


class _Discrete(object):
    """gym.spaces.Discrete surface the callers touch: `.n` and `.sample()`
    (core/algorithms/utils.py:62, examples/griduniverse_env_examples.py:18)."""

    def __init__(self, n):
        self.n = n

    def sample(self):
        return int(np.random.randint(self.n))


def _coordinate_table(width, height):
    # env:109-118 -- world[s] = (x, y), row-major, dtype 'int64, int64'
    table = np.empty(width * height, dtype='int64, int64')
    flat = np.arange(width * height, dtype=np.int64)
    table['f0'] = flat % width if width else flat
    table['f1'] = flat // width if width else flat
    return table


class OracleGridUniverseEnv(object):
    metadata = {'render.modes': ['human', 'ansi', 'graphic']}  # env:15

    def __init__(self, grid_shape=(4, 4), *, initial_state=0, goal_states=None, lava_states=None,
                 walls=None, custom_world_fp=None, random_maze=False):
        # argument validation, same order as env:35-43
        for name, value in (('goal_states', goal_states), ('lava_states', lava_states), ('walls', walls)):
            if value is not None and not isinstance(value, list):
                raise TypeError("{} parameter must be a list of integer indices".format(name))
        shape_ok = isinstance(grid_shape, (list, tuple)) and len(grid_shape) == 2 \
            and isinstance(grid_shape[0], int) and isinstance(grid_shape[1], int)
        if not shape_ok:
            raise TypeError("grid_shape parameter must be tuple/list of two integers")

        self.x_max, self.y_max = grid_shape[0], grid_shape[1]  # env:44-45 (columns, rows)
        self.world = _coordinate_table(self.x_max, self.y_max)  # env:46
        self.action_space = _Discrete(4)  # env:48
        self.action_state_to_next_state = [self._up, self._right, self._down, self._left]  # env:51-54
        self.action_descriptors = ['UP', 'RIGHT', 'DOWN', 'LEFT']  # env:56
        self.action_descriptor_to_int = {d: i for i, d in enumerate(self.action_descriptors)}  # env:57
        self.observation_space = _Discrete(self.world.size)  # env:59 (never refreshed: quirk 7)

        self.starting_states = [initial_state] if isinstance(initial_state, int) else initial_state  # env:61-63
        start = random.choice(self.starting_states)  # env:64 -- stdlib GLOBAL rng draw #1
        self.previous_state = self.current_state = self.initial_state = start

        self.goal_states = goal_states if goal_states else [self.world.size - 1]  # env:66-69
        self.lava_states = [] if lava_states is None else lava_states  # env:71-74

        self.wall_indices = []  # env:76-78
        self.wall_grid = np.zeros(self.world.shape)
        self._place_walls(walls)

        self.reward_matrix = np.full(self.world.shape, -1)  # env:80
        for g in self.goal_states:  # env:81-85
            try:
                self.reward_matrix[g] = 10
            except IndexError:
                raise IndexError("Terminal goal state {} is out of grid bounds or is wrong type. "
                                 "Should be an integer.".format(g))
        for l in self.lava_states:  # env:86-90 -- runs second, so lava overrides goal (quirk 4)
            try:
                self.reward_matrix[l] = -10
            except IndexError:
                raise IndexError("Lava terminal state {} is out of grid bounds or is wrong type. "
                                 "Should be an integer.".format(l))

        self.num_previous_states_to_store = 500  # env:92
        self.last_n_states = []
        self.done = False  # env:95-98
        self.info = {}
        self.screen_width, self.screen_height = 1200, 800
        self.viewer = None
        self.seed()  # env:101
        self.np_random = np.random.RandomState(55)  # env:102 (never consumed)

        if custom_world_fp:  # env:104-105
            self._load_file(custom_world_fp)
        if random_maze:  # env:106-107
            self._load_lines(_maze.create_random_maze(self.x_max, self.y_max))

    # ------------------------------------------------------------ moves env:51-54
    def _up(self, s):
        return s - self.x_max if self.world[s][1] > 0 else s

    def _right(self, s):
        return s + 1 if self.world[s][0] < (self.x_max - 1) else s

    def _down(self, s):
        return s + self.x_max if self.world[s][1] < (self.y_max - 1) else s

    def _left(self, s):
        return s - 1 if self.world[s][0] > 0 else s

    # ------------------------------------------------------------ walls env:120-134
    def _place_walls(self, walls):
        if walls is None:
            return
        for w in walls:
            if w < 0 or w > (self.world.size - 1):
                raise ValueError("Wall state {} is out of grid bounds".format(w))
            self.wall_grid[w] = 1
            self.wall_indices.append(w)

    # ------------------------------------------------------------ predicates env:157-174
    def _is_wall(self, state):
        return bool(self.wall_grid[state] == 1)

    def is_lava(self, state):
        return state in self.lava_states

    def is_terminal_goal(self, state):
        return state in self.goal_states

    def is_terminal(self, state):
        return self.is_lava(state) or self.is_terminal_goal(state)

    # ------------------------------------------------------------ transition env:136-155
    def look_step_ahead(self, state, action, care_about_terminal=True):
        if care_about_terminal and self.is_terminal(state):
            nxt = state  # absorbing (quirk 1)
        else:
            cand = self.action_state_to_next_state[action](state)
            nxt = state if self._is_wall(cand) else cand  # walls block entering only (quirk 2)
        return nxt, self.reward_matrix[nxt], self.is_terminal(nxt)

    # ------------------------------------------------------------ gym surface
    def step(self, action):  # env:176-185
        self.previous_state = self.current_state
        self.current_state, reward, self.done = self.look_step_ahead(self.current_state, action)
        self.last_n_states.append(self.world[self.current_state])
        if len(self.last_n_states) > self.num_previous_states_to_store:
            self.last_n_states.pop(0)
        return self.current_state, reward, self.done, self.info

    def reset(self):  # env:187-193
        self.done = False
        self.previous_state = self.current_state = self.initial_state = random.choice(self.starting_states)
        self.last_n_states = []
        return self.current_state

    def render(self, mode='human', close=False):  # env:195-230
        if close:
            self.viewer = None
            return None
        if mode not in self.metadata['render.modes']:
            raise UnsupportedMode('Unsupported rendering mode: {}'.format(mode))
        if mode == 'graphic':
            raise UnsupportedMode("'graphic' needs pyglet + a display; out of scope for the oracle")
        cells = ['o'] * (self.x_max * self.y_max)
        # overwrite order env:205-213: agent, goals, lava, walls.  Plain Python
        # list indexing differs from the reference's numpy 'S1' array only for
        # out-of-range entries, which the ctor has already rejected.
        cells[self.current_state] = 'x'
        for g in self.goal_states:
            cells[g] = 'G'
        for l in self.lava_states:
            cells[l] = 'L'
        for w in self.wall_indices:
            cells[w] = '#'
        out = StringIO() if mode == 'ansi' else sys.stdout
        for y in range(self.y_max):
            row = cells[y * self.x_max:(y + 1) * self.x_max]
            out.write(''.join(c + ' ' for c in row))
            out.write('\n')
        out.write('\n')
        return out

    def seed(self, seed=None):  # env:242-244
        self.np_random = np.random.RandomState(None if seed is None else int(seed) % (2 ** 32))
        return [seed]

    def close(self):  # env:239-240
        pass

    # ------------------------------------------------------------ loader env:246-316
    def _load_file(self, fp):
        with open(fp, 'r') as f:
            raw = [ln.rstrip() for ln in f.readlines()]
        self._load_lines(["".join(ln.split()) for ln in raw if ln])  # env:248-249

    def _load_lines(self, lines):
        goals, starts, lava, walls = [], [], [], []
        width = len(lines[0])  # env:276
        idx = 0
        for line in lines:
            if len(line) != width:
                raise ValueError("Input text file is not a rectangle")
            for ch in line:
                if ch == 'G':
                    goals.append(idx)
                elif ch == 'L':
                    lava.append(idx)
                elif ch == '#':
                    walls.append(idx)
                elif ch == 'x':
                    starts.append(idx)
                elif ch != 'o':
                    raise ValueError('Invalid Character "{}". Returning'.format(ch))
                idx += 1
        # the reference assigns the three lists to self before validating (env:270-272)
        self.goal_states, self.starting_states, self.lava_states = goals, starts, lava
        if not starts:
            raise ValueError("No starting states set in text file. Place \"x\" within grid. ")
        if not goals:
            raise ValueError("No terminal goal states set in text file. Place \"T\" within grid. ")
        self.reset()  # env:302 -- stdlib rng draw, BEFORE the resize
        self.y_max, self.x_max = len(lines), width  # env:304-306
        self.world = _coordinate_table(self.x_max, self.y_max)
        self.wall_grid = np.zeros(self.world.shape)  # env:308-310
        self.wall_indices = []
        self._place_walls(walls)
        self.reward_matrix = np.full(self.world.shape, -1)  # env:312-316
        for g in goals:
            self.reward_matrix[g] = 10
        for l in lava:
            self.reward_matrix[l] = -10
