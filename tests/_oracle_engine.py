"""Oracle-backed stand-in for griduniverse_amd.engine.Engine, for CPU-only tests of the multi-process
host logic (shard plan, unique-id plumbing, gathered-view layout).  TEST CODE: it lives in tests/ and is
injected through the `engine_factory` hook; the product never constructs it."""
import os

import numpy as np

from griduniverse_amd.parallel import unpack_view
from oracle import c_oracle as C


class OracleEngine(object):
    def __init__(self, num_envs, spec, device=0, env_id0=0, seed=0):
        self.N, self.env_id0, self.spec, self.seed_value = int(num_envs), int(env_id0), spec, int(seed)
        self.grid = C.Grid(spec.W, spec.H, spec.wall, spec.lava, spec.goal, spec.reward, spec.starts)
        self.state = C.State(self.N, self.env_id0)
        self.state.pos[:] = spec.starts[0]
        self.last_reward = np.zeros(self.N, np.int32)
        self.traj = None
        self.uid = None

    def seed(self, seed):
        self.seed_value = int(seed)
        self.state.episode[:] = 0
        self.state.tcount[:] = 0

    def reset(self, mask=None, start_choice=None):
        assert start_choice is None
        return C.reset(self.grid, self.seed_value, self.state, mask)

    def step(self, actions, auto_reset=False):
        out = C.rollout(self.grid, self.seed_value, self.state, 1, auto_reset, np.asarray(actions, np.int32)[None, :])
        self.last_reward = out['reward'][0].copy()
        return out['obs'][0], out['reward'][0], out['done'][0]

    def reserve_trajectory(self, T):
        pass

    def rollout(self, T, policy='uniform', auto_reset=True, trajectory=True, stats=False):
        assert policy == 'uniform'
        self.traj = C.rollout(self.grid, self.seed_value, self.state, T, auto_reset, stats=stats)
        self.last_reward = self.traj['reward'][-1].copy()

    def read_stats(self):
        return self.traj['ret'].copy(), self.traj['episodes'].copy()

    def read_trajectory(self, t0, T):
        return {k: self.traj[k][t0:t0 + T] for k in ('obs', 'reward', 'done')}

    def read_outputs(self):
        return self.state.pos.copy(), self.last_reward.copy(), self.state.done.copy()

    def get_state(self):
        return dict(pos=self.state.pos.copy(), done=self.state.done.copy(), episode=self.state.episode.copy(),
                    tcount=self.state.tcount.copy())

    # ---- stream / timing stand-ins (host clock)
    def sync(self):
        pass

    def timer_begin(self):
        import time
        self._t0 = time.perf_counter()

    def timer_end(self):
        import time
        return (time.perf_counter() - self._t0) * 1e3

    def timer_mark(self):
        import time
        self.__dict__.setdefault('_marks', []).append(time.perf_counter())

    def timer_laps(self, max_laps=65536):
        marks, self._marks = np.asarray(self.__dict__.get('_marks', []), np.float64), []
        return np.diff(marks) * 1e3

    # ---- the gathered view over the host channel instead of RCCL (same packed layout as csrc/gu_comm.hip)
    host_channel = None  # a griduniverse_amd.rendezvous.Rendezvous; set by whoever drives several ranks (bench.py, the worker)

    @staticmethod
    def comm_unique_id():
        return os.urandom(128)

    def comm_init(self, nranks, rank, unique_id):
        if os.environ.get('GU_TEST_HANG_RCCL'):  # (tests/test_multiprocess.py: a collective that never comes back)
            import time
            time.sleep(3600)
        self.nranks, self.rank, self.uid = nranks, rank, bytes(unique_id)
        everyone = type(self).host_channel.allgather_bytes(self.uid) if nranks > 1 else [self.uid]
        assert len(everyone) == nranks and all(u == self.uid for u in everyone), 'ranks disagree on the unique id'

    def allgather_view(self):
        block = np.concatenate(self.read_outputs()).astype('<i4').tobytes()
        blocks = type(self).host_channel.allgather_bytes(block) if self.nranks > 1 else [block]
        return unpack_view(np.stack([np.frombuffer(b, dtype='<i4') for b in blocks]), self.N)

    def comm_destroy(self):
        self.uid = None

    # one process, several engines (stands in for gu_comm_init_all / gu_allgather_view_all)
    @staticmethod
    def comm_init_all(engines):
        for rank, e in enumerate(engines):
            e.nranks, e.rank = len(engines), rank

    @staticmethod
    def allgather_view_all(engines):
        blocks = [np.concatenate(e.read_outputs()).astype(np.int32) for e in engines]
        return unpack_view(np.stack(blocks), engines[0].N)

    def close(self):
        pass
