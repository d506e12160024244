"""Vectorised-numpy restatement of the reference transition for a batch of instances (test infrastructure; the
"second, stronger CPU baseline" of SURVEY.md 8(d)).

Same operation structure as core/envs/griduniverse_env.py, applied to N instances at once with numpy:
  move lambdas          env:51-54   (edge clamps on x = s % W, y = s // W)
  look_step_ahead       env:136-155 (absorbing terminals :145-146, wall test on the candidate :149, reward :154)
  _step / _reset        env:176-193 (harness `if done: reset()` applied lazily before the next step)
No transition table is precomputed: every step evaluates the reference's tests on arrays.  Start choice and uniform
actions come from the build's counter RNG (oracle/gu_rng.py), like every other oracle here.
"""
import numpy as np

from . import gu_rng


class NumpyBatchEnv(object):
    def __init__(self, W, H, starts, goals, lava, walls, reward, n, seed=0, env_id0=0):
        self.W, self.H, self.S = int(W), int(H), int(W) * int(H)
        self.wall = np.zeros(self.S, bool)
        self.wall[[w for w in walls]] = True
        self.terminal = np.zeros(self.S, bool)
        for s in list(goals) + list(lava):
            if 0 <= s < self.S:  # `s in goal_states`: out-of-range entries never match (quirk 5)
                self.terminal[s] = True
        self.reward_matrix = np.asarray(reward, dtype=np.int64).reshape(self.S)
        self.starts = np.asarray(starts, dtype=np.int32)
        self.n, self.seed, self.ids = int(n), int(seed), np.arange(env_id0, env_id0 + n, dtype=np.uint64)
        self.pos = np.full(self.n, self.starts[0], np.int32)
        self.done = np.zeros(self.n, bool)
        self.episode = np.zeros(self.n, np.uint32)
        self.tcount = np.zeros(self.n, np.uint64)

    @classmethod
    def from_env(cls, env, n, seed=0, env_id0=0):
        return cls(env.x_max, env.y_max, env.starting_states, env.goal_states, env.lava_states,
                   np.flatnonzero(np.asarray(env.wall_grid) == 1).tolist(), env.reward_matrix, n, seed, env_id0)

    def reset(self, mask=None):
        m = np.ones(self.n, bool) if mask is None else np.array(mask, dtype=bool)  # a copy: `mask` may be self.done
        idx = gu_rng.start_index_v(self.seed, self.ids[m], self.episode[m], len(self.starts))
        self.pos[m] = self.starts[idx]
        self.done[m] = False
        self.episode[m] += 1
        return self.pos.copy()

    def step(self, actions, auto_reset=False):
        if auto_reset and self.done.any():
            self.reset(self.done)
        a = np.asarray(actions)
        s = self.pos
        x, y = s % self.W, s // self.W
        cand = np.where(a == 0, np.where(y > 0, s - self.W, s),
                        np.where(a == 1, np.where(x < self.W - 1, s + 1, s),
                                 np.where(a == 2, np.where(y < self.H - 1, s + self.W, s), np.where(x > 0, s - 1, s))))
        cand = np.where(self.wall[cand], s, cand)
        nxt = np.where(self.terminal[s], s, cand).astype(np.int32)
        self.pos = nxt
        self.done = self.terminal[nxt]
        self.tcount += 1
        return nxt, self.reward_matrix[nxt].astype(np.int32), self.done.astype(np.int32)

    def rollout(self, T, auto_reset=True, actions=None):
        obs, rew, don = (np.empty((T, self.n), np.int32) for _ in range(3))
        for t in range(T):
            a = gu_rng.actions_v(self.seed, self.ids, self.tcount.astype(np.uint64)) if actions is None else actions[t]
            obs[t], rew[t], don[t] = self.step(a, auto_reset)
        return dict(obs=obs, reward=rew, done=don)
