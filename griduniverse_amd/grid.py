"""GridSpec: the reference's per-instance grid attributes -> row bit-planes for libgu.

The reference keeps a grid as Python lists and numpy arrays on the env instance
(core/envs/griduniverse_env.py:61-90: starting_states, goal_states, lava_states,
wall_grid, reward_matrix).  The engine wants them as `uint32[H][ceil(W/32)]` row
bit-planes (include/gu.h: gu_set_grid), bit (x & 31) of word (x >> 5) of row y for
cell s = y*W + x.  Five planes: wall, goal membership, lava membership, and the two
reward planes R == +10 / R == -10 (the reward matrix and the terminal test can
disagree in the reference -- negative indices wrap only in the former, SURVEY.md
8(a) quirk 5 -- so the reward planes are taken from reward_matrix itself).
"""
import numpy as np


def _row_planes(flags, W, H):
    """bool[S] -> uint32[H, ceil(W/32)]"""
    wpr = (W + 31) // 32
    padded = np.zeros((H, wpr * 32), dtype=np.uint8)
    padded[:, :W] = np.asarray(flags, dtype=np.uint8).reshape(H, W)
    weights = (np.uint64(1) << np.arange(32, dtype=np.uint64))
    words = (padded.reshape(H, wpr, 32).astype(np.uint64) * weights).sum(axis=2)
    return np.ascontiguousarray(words.astype(np.uint32))


def _membership(indices, S):
    """`s in indices` for s in range(S) -- Python equality, so out-of-range or negative
    entries simply never match (env:170-174)."""
    flags = np.zeros(S, dtype=bool)
    for i in indices:
        if isinstance(i, (int, np.integer)) and 0 <= int(i) < S:
            flags[int(i)] = True
    return flags


class GridSpec(object):
    def __init__(self, W, H, starts, goals, lava, walls, reward=None):
        self.W, self.H = int(W), int(H)
        self.S = self.W * self.H
        if self.W <= 0 or self.H <= 0:
            raise ValueError('grid must have at least one cell, got {}x{}'.format(W, H))
        self.starts = [int(s) for s in starts]
        if not self.starts:
            raise ValueError('at least one starting state is required')
        for s in self.starts:
            if not 0 <= s < self.S:
                raise ValueError('starting state {} is outside the {}x{} grid'.format(s, W, H))
        self.wall = _membership(walls, self.S)
        self.goal = _membership(goals, self.S)
        self.lava = _membership(lava, self.S)
        if reward is None:  # env:80-90
            r = np.full(self.S, -1, dtype=np.int64)
            r[self.goal] = 10
            r[self.lava] = -10
        else:
            r = np.asarray(reward, dtype=np.int64).reshape(self.S)
            if not np.isin(r, (-1, 10, -10)).all():
                raise ValueError('reward_matrix may only hold -1, +10 and -10 (env:80-90)')
        self.reward = r

    @classmethod
    def from_env(cls, env):
        """From any object exposing the reference env's attributes."""
        return cls(env.x_max, env.y_max, env.starting_states, env.goal_states, env.lava_states,
                   np.flatnonzero(np.asarray(env.wall_grid) == 1).tolist(), np.asarray(env.reward_matrix))

    @property
    def words_per_row(self):
        return (self.W + 31) // 32

    def planes(self):
        W, H = self.W, self.H
        return dict(wall=_row_planes(self.wall, W, H), goal=_row_planes(self.goal, W, H),
                    lava=_row_planes(self.lava, W, H), rplus=_row_planes(self.reward == 10, W, H),
                    rminus=_row_planes(self.reward == -10, W, H))

    def key(self):
        return (self.W, self.H, tuple(self.starts), self.wall.tobytes(), self.goal.tobytes(),
                self.lava.tobytes(), self.reward.tobytes())
