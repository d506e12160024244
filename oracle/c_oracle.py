"""ctypes front-end of oracle/gu_oracle.c (test infrastructure).

`Grid.from_lists` takes the grid the way the reference holds it (W, H, walls,
goal_states, lava_states, reward_matrix, starting_states) -- e.g. straight from a
golden fixture's meta -- and the functions below run the batched C restatement.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libgu_oracle.so')
_lib = None


class _CGrid(ctypes.Structure):
    _fields_ = [('W', ctypes.c_int32), ('H', ctypes.c_int32),
                ('wall', ctypes.c_void_p), ('lava', ctypes.c_void_p), ('goal', ctypes.c_void_p),
                ('reward', ctypes.c_void_p), ('starts', ctypes.c_void_p), ('n_starts', ctypes.c_int32)]


def build(force=False):
    src = os.path.join(_HERE, 'gu_oracle.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-s', '-C', _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.gu_oracle_rng_word.restype = ctypes.c_uint32
        _lib.gu_oracle_rng_word.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
        _lib.gu_oracle_rng_sample_word.restype = ctypes.c_uint32
        _lib.gu_oracle_rng_sample_word.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint64]
        _lib.gu_oracle_rng_action.restype = ctypes.c_int32
        _lib.gu_oracle_rng_action.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint64]
        _lib.gu_oracle_rng_start.restype = ctypes.c_int32
        _lib.gu_oracle_rng_start.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int32]
        _lib.gu_oracle_value_iteration_step.restype = ctypes.c_double
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class Grid(object):
    def __init__(self, W, H, wall, lava, goal, reward, starts):
        self.W, self.H, self.S = int(W), int(H), int(W) * int(H)
        self.wall = np.ascontiguousarray(wall, np.uint8)
        self.lava = np.ascontiguousarray(lava, np.uint8)
        self.goal = np.ascontiguousarray(goal, np.uint8)
        self.reward = np.ascontiguousarray(reward, np.int32)
        self.starts = np.ascontiguousarray(starts, np.int32)
        self.c = _CGrid(self.W, self.H, _p(self.wall), _p(self.lava), _p(self.goal), _p(self.reward),
                        _p(self.starts), len(self.starts))

    @classmethod
    def from_lists(cls, W, H, walls=(), goals=None, lava=(), starts=(0,), reward=None, **_ignored):
        S = W * H
        goals = [S - 1] if not goals else goals
        flags = lambda idx: np.isin(np.arange(S), [i for i in idx]).astype(np.uint8)  # noqa: E731  (`s in list`)
        if reward is None:  # env:80-90
            reward = np.full(S, -1, np.int64)
            for g_ in goals:
                reward[g_] = 10
            for l_ in lava:
                reward[l_] = -10
        return cls(W, H, flags(walls), flags(lava), flags(goals), reward, list(starts))

    @classmethod
    def from_env(cls, env):
        """From any object with the reference env's attributes."""
        S = env.world.size
        flags = lambda idx: np.isin(np.arange(S), list(idx)).astype(np.uint8)  # noqa: E731
        return cls(env.x_max, env.y_max, (np.asarray(env.wall_grid) == 1), flags(env.lava_states),
                   flags(env.goal_states), np.asarray(env.reward_matrix), list(env.starting_states))


def look_step_ahead(grid, states, actions, care_about_terminal=True):
    states = np.ascontiguousarray(states, np.int32)
    actions = np.ascontiguousarray(actions, np.int32)
    n = states.size
    nxt, rew, don = (np.empty(n, np.int32) for _ in range(3))
    lib().gu_oracle_look_step_ahead_batch(ctypes.byref(grid.c), ctypes.c_int64(n), _p(states), _p(actions),
                                          ctypes.c_int32(bool(care_about_terminal)), _p(nxt), _p(rew), _p(don))
    return nxt, rew, don


class State(object):
    """Per-env state of a batch: pos, done, episode counter, step counter."""

    def __init__(self, n, env_id0=0):
        self.n, self.env_id0 = int(n), int(env_id0)
        self.pos = np.zeros(n, np.int32)
        self.done = np.zeros(n, np.int32)
        self.episode = np.zeros(n, np.uint32)
        self.tcount = np.zeros(n, np.uint64)  # steps since seeding (64 bits: oracle/gu_rng.py, the epoch of streams 0 and 2)


def reset(grid, seed, state, mask=None):
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
    lib().gu_oracle_reset(ctypes.byref(grid.c), ctypes.c_uint64(seed), ctypes.c_int64(state.env_id0),
                          ctypes.c_int64(state.n), _p(m), _p(state.pos), _p(state.done), _p(state.episode))
    return state.pos.copy()


def rollout(grid, seed, state, T, auto_reset=True, actions=None, trajectory=True, stats=False, pi=None):
    n = state.n
    acts = None if actions is None else np.ascontiguousarray(actions, np.int32)
    if acts is not None:
        assert acts.shape == (T, n)
    obs = rew = don = None
    if trajectory:
        obs, rew, don = (np.empty((T, n), np.int32) for _ in range(3))
    pi_c = None if pi is None else np.ascontiguousarray(pi, np.float64)
    ret = np.zeros(n, np.int64) if stats else None
    eps = np.zeros(n, np.int32) if stats else None
    lib().gu_oracle_rollout(ctypes.byref(grid.c), ctypes.c_uint64(seed), ctypes.c_int64(state.env_id0),
                            ctypes.c_int64(n), ctypes.c_int64(T), ctypes.c_int32(bool(auto_reset)), _p(acts), _p(pi_c),
                            _p(state.pos), _p(state.done), _p(state.episode), _p(state.tcount),
                            _p(obs), _p(rew), _p(don), _p(ret), _p(eps))
    out = dict(obs=obs, reward=rew, done=don)
    if stats:
        out.update(ret=ret, episodes=eps)
    return out


def policy_evaluation_sweep(grid, gamma, pi, v):
    pi = np.ascontiguousarray(pi, np.float64)
    v = np.ascontiguousarray(v, np.float64)
    out = np.empty(grid.S, np.float64)
    lib().gu_oracle_policy_evaluation_sweep(ctypes.byref(grid.c), ctypes.c_double(gamma), _p(pi), _p(v), _p(out))
    return out


def greedy_policy(grid, gamma, v):
    v = np.ascontiguousarray(v, np.float64)
    pi = np.empty((grid.S, 4), np.float64)
    lib().gu_oracle_greedy_policy(ctypes.byref(grid.c), ctypes.c_double(gamma), _p(v), _p(pi))
    return pi


def value_iteration_step(grid, gamma, pi, v):
    """Returns (v_new, pi_new, delta); inputs untouched."""
    pi = np.array(pi, np.float64, order='C', copy=True)
    v = np.ascontiguousarray(v, np.float64)
    v_new = np.empty(grid.S, np.float64)
    delta = lib().gu_oracle_value_iteration_step(ctypes.byref(grid.c), ctypes.c_double(gamma), _p(pi), _p(v), _p(v_new))
    return v_new, pi, float(delta)


def generate_maze(maze_seed, grid_id, W, H):
    """Restatement of the on-device generator: returns (wall bool[S], start, goal)."""
    S = W * H
    wall = np.empty(S, np.uint8)
    stack = np.empty(S + 4, np.int32)
    start, goal = ctypes.c_int32(-1), ctypes.c_int32(-1)
    n_open = lib().gu_oracle_generate_maze(ctypes.c_uint64(maze_seed), ctypes.c_uint32(grid_id), ctypes.c_int32(W),
                                           ctypes.c_int32(H), _p(wall), ctypes.byref(start), ctypes.byref(goal), _p(stack))
    if n_open < 2:
        raise ValueError('maze has fewer than two open cells')
    return wall.astype(bool), int(start.value), int(goal.value)
