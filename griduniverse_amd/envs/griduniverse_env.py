"""GridUniverseEnv -- drop-in for `core.envs.griduniverse_env.GridUniverseEnv`.

Same constructor, attributes, return types, error behaviour and quirks as the
reference class (core/envs/griduniverse_env.py:14-321), so the
tabular-RL drivers of examples/griduniverse_alg_examples.py run on it unchanged.
What differs is WHERE the transition runs: grid description, level loading, maze
generation and ASCII rendering are host Python (they run once per grid / per frame),
while the transition itself is computed on the MI355X through libgu.so: the first
`step()` / `look_step_ahead()` on a grid has the look-ahead kernel evaluate all S x 4
(state, action) pairs, and scalar calls then index that table (a launch per scalar
call would cost ~12 us for ~10 integer operations); `step_on_device()` runs the step
kernel itself.  This class is the N = 1 facade of the batched engine (see `vec_env.py`
for the N >> 1 API the engine is built for).  There is no CPU fallback: without the
library or a GPU `step` / `look_step_ahead` raise `GuError`.

Old-gym dispatch (public `step` -> `_step` ...) is flattened: the public names are the
implementation and the underscore names are aliases.
"""
import random
import sys
from io import StringIO

import numpy as np

from . import maze_generation
from .spaces import Discrete
from ..grid import GridSpec


class UnsupportedMode(Exception):
    """Raised for render modes that are not available (gym.error.UnsupportedMode's role)."""


def _grid_coordinates(n_cols, n_rows):
    """world[s] = (x, y) as a structured 'int64, int64' array, row-major (env:109-118)."""
    world = np.zeros(n_cols * n_rows, dtype='int64, int64')
    ys, xs = np.divmod(np.arange(n_cols * n_rows, dtype=np.int64), max(n_cols, 1))
    world['f0'], world['f1'] = xs, ys
    return world


class _WatchedList(list):
    """A list that tells its env when it is edited IN PLACE: the reference consults goal_states / lava_states / starting_states
    on every step (env:157-174, 187-193), so an edit takes effect at once; here the grid is compiled into the engine, and the
    edit drops the compiled form.  Compares, prints, pickles and serialises as the list it is."""

    def __init__(self, items, on_change):
        super().__init__(items)
        self._on_change = on_change

    def __reduce__(self):
        return (list, (list(self),))


def _watch(name):
    def method(self, *args, **kwargs):
        out = getattr(list, name)(self, *args, **kwargs)
        self._on_change()
        return out
    method.__name__ = name
    return method


for _name in ('append', 'extend', 'insert', 'remove', 'pop', 'clear', 'sort', 'reverse', '__setitem__', '__delitem__', '__iadd__', '__imul__'):
    setattr(_WatchedList, _name, _watch(_name))


class _WatchedArray(np.ndarray):
    """wall_grid / reward_matrix: an assignment into the array (`env.wall_grid[5] = 1`) drops the compiled grid, as above.
    Arrays derived from it (slices, comparisons, copies) are plain in behaviour: they carry no env."""

    def __new__(cls, array, on_change):
        obj = np.asarray(array).view(cls)
        obj._on_change = on_change
        return obj

    def __array_finalize__(self, obj):
        self._on_change = None

    def __setitem__(self, key, value):
        super().__setitem__(key, value)
        if self._on_change is not None:
            self._on_change()

    def __reduce__(self):
        return np.asarray(self).__reduce__()


def _require_list(value, name):
    if value is not None and not isinstance(value, list):
        raise TypeError("{} parameter must be a list of integer indices".format(name))


class GridUniverseEnv(object):
    # 'rgb_array' is an addition of this build (frames rendered on the GPU); the other three are the reference's
    metadata = {'render.modes': ['human', 'ansi', 'graphic', 'rgb_array']}
    reward_range = (-float('inf'), float('inf'))

    # The attributes the reference re-reads on every step: replacing one, or editing it in place, drops the compiled grid.  (Properties
    # on these six names only: a __setattr__ hook would tax every attribute the scalar step() sets, 0.2 us against the reference's 4.2.)
    def _watched_attribute(name, wrap):  # noqa: N805 -- (a class-body helper, not a method)
        slot = '_w_' + name

        def getter(self):
            try:
                return self.__dict__[slot]
            except KeyError:
                raise AttributeError(name) from None

        def setter(self, value):
            self.__dict__[slot] = wrap(self, value)
            self._grid_edited()

        return property(getter, setter)

    goal_states = _watched_attribute('goal_states', lambda self, v: _WatchedList(v, self._grid_edited) if isinstance(v, list) else v)
    lava_states = _watched_attribute('lava_states', lambda self, v: _WatchedList(v, self._grid_edited) if isinstance(v, list) else v)
    starting_states = _watched_attribute('starting_states', lambda self, v: _WatchedList(v, self._grid_edited) if isinstance(v, list) else v)
    wall_indices = _watched_attribute('wall_indices', lambda self, v: _WatchedList(v, self._grid_edited) if isinstance(v, list) else v)
    wall_grid = _watched_attribute('wall_grid', lambda self, v: _WatchedArray(v, self._grid_edited) if isinstance(v, np.ndarray) else v)
    reward_matrix = _watched_attribute('reward_matrix', lambda self, v: _WatchedArray(v, self._grid_edited) if isinstance(v, np.ndarray) else v)
    del _watched_attribute

    def _grid_edited(self):
        if self.__dict__.get('_engine_obj') is not None or self.__dict__.get('_tables') or self.__dict__.get('_step_tab') is not None:
            self._drop_engine()

    def __init__(self, grid_shape=(4, 4), *, initial_state=0, goal_states=None, lava_states=None, walls=None,
                 custom_world_fp=None, random_maze=False, device=0):
        _require_list(goal_states, 'goal_states')
        _require_list(lava_states, 'lava_states')
        _require_list(walls, 'walls')
        if not isinstance(grid_shape, (list, tuple)) or len(grid_shape) != 2 \
                or not all(isinstance(v, int) for v in grid_shape):
            raise TypeError("grid_shape parameter must be tuple/list of two integers")

        self._engine_obj = None
        self._tables = {}
        self._step_tab = None
        self._device = device
        self._pos_dirty = True

        self.x_max, self.y_max = grid_shape
        self.world = _grid_coordinates(self.x_max, self.y_max)
        self.action_space = Discrete(4)
        self.action_descriptors = ['UP', 'RIGHT', 'DOWN', 'LEFT']
        self.action_descriptor_to_int = {name: i for i, name in enumerate(self.action_descriptors)}
        # callers only take len() of this (examples/griduniverse_alg_examples.py:31); the moves themselves
        # are compiled into the engine's per-cell records
        self.action_state_to_next_state = [self._host_move(a) for a in range(4)]
        self.observation_space = Discrete(self.world.size)  # stale after loading, like the reference (quirk 7)

        self.starting_states = [initial_state] if isinstance(initial_state, int) else initial_state
        self.done = False
        self._state = self.previous_state = self.initial_state = random.choice(self.starting_states)

        self.goal_states = goal_states if goal_states else [self.world.size - 1]
        self.lava_states = lava_states if lava_states is not None else []
        self._install_cells(walls, from_ctor=True)

        self.num_previous_states_to_store = 500
        self.last_n_states = []
        self.info = {}
        self.screen_width, self.screen_height = 1200, 800
        self.viewer = None
        self.seed()
        self.np_random = np.random.RandomState(55)  # env:102; never consumed by any decision

        if custom_world_fp:
            self._create_custom_world_from_file(custom_world_fp)
        if random_maze:
            self._create_random_maze(self.x_max, self.y_max)

    # ------------------------------------------------------------------ grid tables (host)
    def _install_cells(self, walls, from_ctor):
        """wall_grid / wall_indices / reward_matrix for the current lists (env:76-90, 308-316)."""
        self.wall_indices = []
        self.wall_grid = np.zeros(self.world.shape)
        for w in (walls or []):
            if w < 0 or w > self.world.size - 1:
                raise ValueError("Wall state {} is out of grid bounds".format(w))
            self.wall_grid[w] = 1
            self.wall_indices.append(w)
        self.reward_matrix = np.full(self.world.shape, -1)
        for kind, states, value in (('Terminal goal', self.goal_states, 10), ('Lava terminal', self.lava_states, -10)):
            for s in states:
                try:
                    self.reward_matrix[s] = value  # numpy semantics: negatives wrap, OOB / non-int raise
                except IndexError:
                    if not from_ctor:
                        raise
                    raise IndexError("{} state {} is out of grid bounds or is wrong type. "
                                     "Should be an integer.".format(kind, s))
        self._drop_engine()

    def _host_move(self, action):
        def move(s):
            x, y = self.world[s]
            if action == 0:
                return s - self.x_max if y > 0 else s
            if action == 1:
                return s + 1 if x < self.x_max - 1 else s
            if action == 2:
                return s + self.x_max if y < self.y_max - 1 else s
            return s - 1 if x > 0 else s
        return move

    # ------------------------------------------------------------------ device plumbing
    def _drop_engine(self):
        if self._engine_obj is not None:
            self._engine_obj.close()
        self._engine_obj = None
        batch = self.__dict__.pop('_episode_batch', None)  # algorithms.monte_carlo keeps its last batch engine here
        if batch is not None:
            batch[1].close()
        self._tables = {}
        self._step_tab = None
        self._pos_dirty = True

    def invalidate(self):
        """Drops the compiled grid (engine, cached transition table).  Since round 6 the env does this itself whenever goal_states /
        lava_states / starting_states / wall_indices are replaced or edited in place and whenever wall_grid / reward_matrix are
        assigned into -- the reference re-reads them on every step (env:157-174), so an edit takes effect at the next step here
        too.  Still needed after writing through a VIEW of one of the two arrays (`v = env.wall_grid[2:]; v[0] = 1`)."""
        self._drop_engine()

    def _engine(self):
        if self._engine_obj is None:
            from ..engine import Engine  # imported late: host-only use never loads libgu
            self._engine_obj = Engine(1, GridSpec.from_env(self), device=self._device)
            self._act_buf = self._engine_obj.pinned_actions
            self._pos_dirty = True
        return self._engine_obj

    def _push_state(self, eng):
        if self._pos_dirty:
            eng.set_state(pos=[self._state], done=[1 if self.done else 0])
            self._pos_dirty = False

    @property
    def current_state(self):
        return self._state

    @current_state.setter
    def current_state(self, value):
        self._state = value
        self._pos_dirty = True

    def _transition_table(self, care_about_terminal):
        """(next, reward, done) for every (state, action), computed ONCE by the HIP kernel
        gu_look_step_ahead and cached for scalar queries (a launch per scalar query would
        cost ~20 us for ~10 integer ops)."""
        key = bool(care_about_terminal)
        if key not in self._tables:
            S = self.world.size
            states = np.repeat(np.arange(S, dtype=np.int32), 4)
            actions = np.tile(np.arange(4, dtype=np.int32), S)
            nxt, rew, don = self._engine().look_step_ahead(states, actions, key)
            self._tables[key] = (nxt.reshape(S, 4), rew.reshape(S, 4).astype(np.int64), don.reshape(S, 4).astype(bool))
        return self._tables[key]

    # ------------------------------------------------------------------ transition API
    def look_step_ahead(self, state, action, care_about_terminal=True):
        nxt, rew, don = self._transition_table(care_about_terminal)
        if not -4 <= action < 4:
            raise IndexError('list index out of range')  # env:148 indexes a 4-element list
        return int(nxt[state, action]), rew[state, action], bool(don[state, action])

    def _is_wall(self, state):
        return bool(self.wall_grid[state] == 1)

    def is_lava(self, state):
        return state in self.lava_states

    def is_terminal_goal(self, state):
        return state in self.goal_states

    def is_terminal(self, state):
        return self.is_lava(state) or self.is_terminal_goal(state)

    _TABLE_STEP_MAX_CELLS = 1 << 18

    def _build_step_tab(self):
        """The N = 1 hot path: the (state, action) -> (next, reward, done) table that the HIP kernel gu_lookahead_kernel
        produced for this grid (one launch for all S x 4 pairs, `_transition_table`), unpacked into plain Python rows.
        A scalar `step()` is then two list indexings -- ~0.3 us -- instead of a kernel launch plus a PCIe round trip
        (~12 us) for ~10 integer operations; batches step on the device (`VecGridUniverse`)."""
        nxt, rew, don = self._transition_table(True)
        boxed = {int(r): np.int64(r) for r in np.unique(rew)}  # three np.int64 objects shared by every entry
        self._step_tab = (nxt.tolist(), [[boxed[r] for r in row] for row in rew.tolist()], don.tolist(),
                          [self.world[i] for i in range(self.world.size)])
        return self._step_tab

    def step(self, action):
        """One env-step (env:176-185): a lookup in the transition table computed on the device for this grid."""
        tab = self._step_tab
        if tab is None:
            if self.world.size > self._TABLE_STEP_MAX_CELLS:  # a Python table of 4 x S entries stops being sensible
                return self.step_on_device(action)
            tab = self._build_step_tab()
        state = self._state
        nxt = tab[0][state][action]  # a 4-element list, like env:148: -4..-1 wrap, anything else is IndexError (quirk 6)
        self.previous_state = state
        self._state = nxt
        self._pos_dirty = True
        done = self.done = tab[2][state][action]
        trail = self.last_n_states
        trail.append(tab[3][nxt])
        if len(trail) > self.num_previous_states_to_store:
            trail.pop(0)
        return nxt, tab[1][state][action], done, self.info

    def reset_to_trail(self, states, done):
        """Put the instance where an episode that visited `states` (reset state first) left the reference's: current /
        previous state, done flag, and the (x, y) trail of the steps taken (env:176-185 per step, capped at 500)."""
        self.done = bool(done)
        self._state = int(states[-1])
        self.previous_state = int(states[-2]) if len(states) > 1 else int(states[-1])
        self.last_n_states = [self.world[int(s)] for s in states[1:][-self.num_previous_states_to_store:]]
        self._pos_dirty = True

    def step_on_device(self, action):
        """The same step executed by gu_step_kernel on the engine's N = 1 batch (kernel launch + page-locked I/O);
        kept for parity tests of the kernel path through the facade."""
        if not -4 <= action < 4:
            raise IndexError('list index out of range')
        eng = self._engine_obj or self._engine()
        if self._pos_dirty:
            self._push_state(eng)
        self.previous_state = self._state
        self._act_buf[0] = action % 4  # negative = Python list index (quirk 6)
        obs, reward, done = eng.step_pinned()  # page-locked I/O: no bounce copies on the N = 1 path
        self._state = int(obs[0])
        self.done = bool(done[0])
        self.last_n_states.append(self.world[self._state])
        if len(self.last_n_states) > self.num_previous_states_to_store:
            self.last_n_states.pop(0)
        return self._state, np.int64(reward[0]), self.done, self.info

    def reset(self):
        self.done = False
        pick = random.choice(range(len(self.starting_states)))  # same draw as random.choice(starting_states)
        self._state = self.previous_state = self.initial_state = self.starting_states[pick]
        self._pos_dirty = True
        self.last_n_states = []
        return self._state

    # ------------------------------------------------------------------ render / misc surface
    def render(self, mode='human', close=False):
        if close:
            self.viewer = None
            return None
        if mode not in self.metadata['render.modes']:
            raise UnsupportedMode('Unsupported rendering mode: {}'.format(mode))
        if mode == 'rgb_array':
            eng = self._engine()
            self._push_state(eng)
            return eng.render_rgb(0, 1, 16)[0]
        if mode == 'graphic':
            # The reference opens a pyglet window here (env:223-228).  This build is headless: so that drivers
            # written for the reference (examples/griduniverse_alg_examples.py renders in 'graphic' mode) keep
            # running, fall back to the text view on stdout and say so once.
            if not getattr(self, '_graphic_warned', False):
                import warnings
                warnings.warn("mode='graphic' needs pyglet and a display; rendering as text instead", UserWarning)
                self._graphic_warned = True
            mode = 'human'
        canvas = np.full(self.x_max * self.y_max, 'o', dtype='U1')
        canvas[self._state] = 'x'
        for glyph, cells in (('G', self.goal_states), ('L', self.lava_states), ('#', self.wall_indices)):
            for s in cells:
                canvas[s] = glyph
        text = ''.join(' '.join(row) + ' \n' for row in canvas.reshape(self.y_max, self.x_max)) + '\n'
        out = StringIO() if mode == 'ansi' else sys.stdout
        out.write(text)
        return out

    def render_policy_arrows(self, policy, mode='human', cell_px=52):
        """The reference draws the arrows in its pyglet viewer (env:232-237).  Headless: mode='human' prints them as a
        text map; mode='rgb_array' returns the figure (tiles + one arrow per action with probability >= 0.1, the
        geometry of core/envs/rendering.py:159-212) as uint8[H*cell_px, W*cell_px, 3], drawn on the device."""
        if mode == 'rgb_array':
            eng = self._engine()
            eng.vi_set(np.zeros(self.world.size), np.asarray(policy, dtype=np.float64))
            return eng.render_policy_rgb(cell_px)
        from ..algorithms.utils import get_policy_map
        arrows = get_policy_map(policy, (self.x_max, self.y_max), mode='ansi')[0]
        for row in np.reshape(arrows, (self.y_max, self.x_max)):
            sys.stdout.write(''.join('{:<5}'.format(cell) for cell in row) + '\n')
        sys.stdout.write('\n')

    def seed(self, seed=None):
        self.np_random = np.random.RandomState(None if seed is None else int(seed) % (2 ** 32))
        return [seed]

    def close(self):
        self._drop_engine()

    _step, _reset, _render, _seed, _close = step, reset, render, seed, close

    # ------------------------------------------------------------------ level sources
    def _create_custom_world_from_file(self, fp):
        with open(fp, 'r') as f:
            lines = [''.join(line.split()) for line in f.read().splitlines()]
        self._create_custom_world_from_text([line for line in lines if line])

    def _create_custom_world_from_text(self, text_world_lines):
        """Level text -> grid (env:253-316).  'o' floor, '#' wall, 'G' goal, 'L' lava, 'x' start."""
        width = len(text_world_lines[0])
        cells = {'G': [], 'L': [], '#': [], 'x': []}
        self.goal_states, self.lava_states, self.starting_states = [], [], []  # (what a parse error leaves behind, as in the reference)
        for y, line in enumerate(text_world_lines):
            if len(line) != width:
                raise ValueError("Input text file is not a rectangle")
            for x, ch in enumerate(line):
                if ch in cells:
                    cells[ch].append(y * width + x)
                elif ch != 'o':
                    self.goal_states, self.lava_states, self.starting_states = cells['G'], cells['L'], cells['x']
                    raise ValueError('Invalid Character "{}". Returning'.format(ch))
        self.goal_states, self.lava_states, self.starting_states = cells['G'], cells['L'], cells['x']
        if not self.starting_states:
            raise ValueError("No starting states set in text file. Place \"x\" within grid. ")
        if not self.goal_states:
            raise ValueError("No terminal goal states set in text file. Place \"T\" within grid. ")
        self.reset()  # before the resize, as in the reference (env:302) -- keeps RNG consumption identical
        self.y_max, self.x_max = len(text_world_lines), width
        self.world = _grid_coordinates(self.x_max, self.y_max)
        self._install_cells(cells['#'], from_ctor=False)

    def _create_random_maze(self, width, height):
        self._create_custom_world_from_text(maze_generation.create_random_maze(width, height))
