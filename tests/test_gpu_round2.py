"""Round-2 behaviours of the C ABI, against the oracle / golden fixtures:
in-kernel action validation, ballot words written by the step / rollout / reset kernels, page-locked pointer
checks, the 24-bit move limit, RNG counters beyond 2^28, and the facade's table-driven scalar step."""
import ctypes

import numpy as np
import pytest

import griduniverse_amd as gua
from griduniverse_amd import _lib
from griduniverse_amd.engine import Engine
from griduniverse_amd.grid import GridSpec
from oracle import c_oracle as C
from tests import _golden as G

pytestmark = pytest.mark.gpu


def spec_of(meta):
    return GridSpec(meta['W'], meta['H'], meta['starts'], meta['goals'], meta['lava'], meta['walls'], meta['reward'])


@pytest.mark.parametrize('N', [300, 20000])  # completion-word path (<= 8192 envs) and stream-sync path
@pytest.mark.parametrize('pinned', [False, True])
def test_invalid_action_is_caught_by_the_kernel_and_only_that_env_stays(N, pinned):
    meta, _ = G.load_traj('c4_lava32')
    grid = C.Grid.from_lists(**meta)
    st = C.State(N)
    rs = np.random.RandomState(N)
    with Engine(N, spec_of(meta), seed=5) as eng:
        assert np.array_equal(eng.reset(), C.reset(grid, 5, st))
        C.rollout(grid, 5, st, 40, True)
        eng.rollout(40, 'uniform', True, trajectory=False)
        before = eng.get_state()
        rew_before = eng.read_outputs()[1]
        acts = rs.randint(0, 4, N).astype(np.int32)
        bad = np.array([7, N // 2, N - 1])
        acts[bad] = [4, -1, 1 << 20]

        def call():
            if pinned:
                eng.pinned_actions[:] = acts
                return eng.step_pinned(auto_reset=True)
            return eng.step(acts, auto_reset=True)
        with pytest.raises(gua.GuError) as err:
            call()
        assert 'action 4 of env 7 outside 0..3' in str(err.value) and err.value.code == -1
        # envs with valid actions stepped exactly like the oracle; the offenders did not move, count or reset
        good = np.ones(N, bool)
        good[bad] = False
        want = C.rollout(grid, 5, st, 1, True, actions=np.where(good, acts, 0)[None, :])
        after = eng.get_state()
        out = eng.read_outputs()
        for k, w in (('pos', st.pos), ('done', st.done), ('episode', st.episode)):
            assert np.array_equal(after[k][good], w[good]), k
            assert np.array_equal(after[k][bad], before[k][bad]), k
        assert np.array_equal(out[1][good], want['reward'][0][good]) and np.array_equal(out[1][bad], rew_before[bad])
        assert np.all(after['tcount'][good] == 41) and np.all(after['tcount'][bad] == 40)
        # the error word is re-armed: the next valid step succeeds and equals the oracle for the good envs
        acts2 = rs.randint(0, 4, N).astype(np.int32)
        if pinned:
            eng.pinned_actions[:] = acts2
            obs = eng.step_pinned(auto_reset=True)[0].copy()
        else:
            obs = eng.step(acts2, auto_reset=True)[0]
        want2 = C.rollout(grid, 5, st, 1, True, actions=acts2[None, :])
        assert np.array_equal(obs[good], want2['obs'][0][good])
        # done ballots stay current through all of this
        assert np.array_equal(eng.done_indices(), np.flatnonzero(eng.get_state()['done']))


def test_uploaded_action_stream_is_validated_on_the_device():
    meta, _ = G.load_traj('c2_open8x8')
    N, T = 1000, 33
    acts = np.random.RandomState(0).randint(0, 4, (T, N)).astype(np.int32)
    with Engine(N, spec_of(meta), seed=1) as eng:
        eng.upload_actions(acts)
        eng.step_device(T - 1)
        acts[20, 999] = 9
        with pytest.raises(gua.GuError) as err:
            eng.upload_actions(acts)
        assert 'action 9 at flat index %d' % (20 * N + 999) in str(err.value)
        with pytest.raises(gua.GuError):  # the rejected stream is not usable
            eng.step_device(0)
        acts[20, 999] = 3
        eng.upload_actions(acts)
        eng.step_device(0)
        # look_step_ahead: out-of-grid states / bad actions raise from the kernel's error word
        with pytest.raises(gua.GuError) as err:
            eng.look_step_ahead([0, 64, 3], [1, 1, 1])
        assert 'state 64 outside the grid' in str(err.value)
        with pytest.raises(gua.GuError) as err:
            eng.look_step_ahead([0, 5, 3], [1, 1, 5])
        assert 'action 5 outside 0..3' in str(err.value)
        nxt, _, _ = eng.look_step_ahead([0, 5], [1, 2])
        assert nxt.tolist() == [1, 13]


def test_done_ballots_follow_every_kernel_that_writes_done():
    """gu_done_indices is ONE compaction launch over ballot words kept current by step / rollout / reset / sweep-step."""
    meta, _ = G.load_traj('c4_lava32')
    for N in (1, 63, 65, 1000, 70000):
        grid = C.Grid.from_lists(**meta)
        st = C.State(N)
        with Engine(N, spec_of(meta), seed=3) as eng:
            assert len(eng.done_indices()) == 0
            eng.reset()
            C.reset(grid, 3, st)
            for T, auto in ((37, False), (5, True), (64, False)):
                C.rollout(grid, 3, st, T, auto)
                eng.rollout(T, 'uniform', auto, trajectory=False)
                assert np.array_equal(eng.done_indices(), np.flatnonzero(st.done)), (N, T)
            assert N < 1000 or st.done.any()
            acts = (np.arange(N) % 4).astype(np.int32)
            C.rollout(grid, 3, st, 1, False, actions=acts[None, :])
            eng.step(acts)
            assert np.array_equal(eng.done_indices(), np.flatnonzero(st.done))
            mask = np.arange(N) % 2 == 0
            C.reset(grid, 3, st, mask=mask)
            eng.reset(mask)
            assert np.array_equal(eng.done_indices(), np.flatnonzero(st.done))
            eng.reset_done()
            assert len(eng.done_indices()) == 0
            flags = (np.arange(N) % 5 == 0).astype(np.int32)
            eng.set_state(done=flags)  # host-installed flags: the ballot pass runs once
            assert np.array_equal(eng.done_indices(), np.flatnonzero(flags))
            S = meta['W'] * meta['H']
            eng.vi_set(np.zeros(S), np.ones((S, 4)) / 4)
            eng.vi_sweep_step(0.9, auto_reset=True)
            assert np.array_equal(eng.done_indices(), np.flatnonzero(eng.get_state()['done']))


def test_pinned_io_rejects_pageable_memory_instead_of_faulting():
    meta, _ = G.load_traj('c2_open8x8')
    N = 256
    with Engine(N, spec_of(meta), seed=1) as eng:
        eng.reset()
        lib = eng.lib
        acts = np.zeros(N, np.int32)
        out = [np.empty(N, np.int32) for _ in range(3)]
        rc = lib.gu_step(eng._h, _lib.ptr(acts), _lib.F_PINNED_IO, *[_lib.ptr(o) for o in out])
        assert rc == -1 and 'not page-locked' in _lib.last_error()
        pin = _lib.PinnedArray((4, N))
        pin.array[0] = 1
        p = [pin.array[k].ctypes.data_as(ctypes.c_void_p) for k in range(4)]
        assert lib.gu_step(eng._h, p[0], _lib.F_PINNED_IO, p[1], p[2], p[3]) == 0
        assert np.all(pin.array[1] == 1)
        # actions page-locked, an output not
        rc = lib.gu_step(eng._h, p[0], _lib.F_PINNED_IO, p[1], _lib.ptr(out[1]), p[3])
        assert rc == -1 and 'reward' in _lib.last_error()
        # a range that runs past the end of the allocation
        tail = ctypes.c_void_p(pin.array.ctypes.data + 4 * N * 4 - 16)
        assert lib.gu_step(eng._h, p[0], _lib.F_PINNED_IO, tail, None, None) == -1
        # still healthy afterwards
        assert lib.gu_step(eng._h, p[0], _lib.F_PINNED_IO, p[1], p[2], p[3]) == 0
        assert np.all(pin.array[1] == 2)
        pin.free()


def test_grid_wider_than_the_24_bit_move_is_refused():
    lib = _lib.load()
    h = ctypes.c_void_p()
    _lib.check(lib.gu_create(0, 64, 0, ctypes.byref(h)))
    try:
        W, H = (1 << 23) + 1, 2
        wpr = (W + 31) // 32
        zeros = np.zeros((H, wpr), np.uint32)
        goal = zeros.copy()
        goal[1, 0] = 2
        starts = np.zeros(1, np.int32)
        rc = lib.gu_set_grid(h, W, H, wpr, _lib.ptr(zeros), _lib.ptr(goal), _lib.ptr(zeros), None, None, _lib.ptr(starts), 1)
        assert rc == -6 and '8 388 607' in _lib.last_error()
        # the widest supported grid still moves up and down correctly (arithmetic path, no LDS)
        W = (1 << 23) - 1
        wpr = (W + 31) // 32
        zeros = np.zeros((H, wpr), np.uint32)
        goal = zeros.copy()
        goal[1, 0] = 2
        starts[0] = W - 5
        _lib.check(lib.gu_set_grid(h, W, H, wpr, _lib.ptr(zeros), _lib.ptr(goal), _lib.ptr(zeros), None, None, _lib.ptr(starts), 1))
        obs = np.empty(64, np.int32)
        for a, want in ((2, 2 * W - 5), (2, 2 * W - 5), (1, 2 * W - 4), (0, W - 4), (0, W - 4), (3, W - 5)):
            acts = np.full(64, a, np.int32)
            _lib.check(lib.gu_step(h, _lib.ptr(acts), 0, _lib.ptr(obs), None, None))
            assert np.all(obs == want), (a, want, obs[0])
    finally:
        lib.gu_destroy(h)


def test_sampled_stream_does_not_repeat_beyond_2_pow_28_steps():
    meta, _ = G.load_traj('c2_open8x8')
    N, T, S = 512, 40, 64
    pi = np.random.RandomState(3).dirichlet(np.ones(4), S)
    grid = C.Grid.from_lists(**meta)
    rows = {}
    for t0 in (5, (1 << 28) - 7, (1 << 28) + 5, (3 << 28) + 5):
        st = C.State(N)
        with Engine(N, spec_of(meta), seed=9) as eng:
            eng.vi_set(np.zeros(S), pi)
            assert np.array_equal(eng.reset(), C.reset(grid, 9, st))
            st.tcount[:] = t0
            eng.set_state(tcount=st.tcount)
            eng.reserve_trajectory(T)
            for policy, kw in (('sample', dict(pi=pi)), ('uniform', {})):
                eng.rollout(T, policy, True)
                got = eng.read_trajectory(0, T)
                want = C.rollout(grid, 9, st, T, True, **kw)
                assert all(np.array_equal(got[k], want[k]) for k in got), (t0, policy)
                if policy == 'sample':
                    rows[t0] = got['obs'].copy()
    assert not np.array_equal(rows[5], rows[(1 << 28) + 5]) and not np.array_equal(rows[(1 << 28) + 5], rows[(3 << 28) + 5])


def test_get_cells_with_a_start_list_longer_than_the_grid():
    spec = GridSpec(2, 1, [0, 0, 0, 1, 0], [1], [], [])
    with Engine(8, spec) as eng:
        flags, reward, starts = eng.get_cells()
        assert starts.tolist() == [0, 0, 0, 1, 0] and reward.tolist() == [-1, 10]


@pytest.mark.parametrize('kat', G.load_json('kat.json'), ids=lambda k: k['name'])
def test_facade_table_step_equals_the_step_kernel(kat):
    """GridUniverseEnv.step() indexes the transition table the look-ahead kernel produced; step_on_device() launches the
    step kernel.  Both must give the reference's stream for its own ten KATs."""
    for pick, run in enumerate(kat['runs']):
        kw = dict(kat['kwargs'])
        if 'grid_shape' in kw:
            kw['grid_shape'] = tuple(kw['grid_shape'])
        if kat['level']:
            kw['custom_world_fp'] = G.level_path(kat['level'])
        for method in ('step', 'step_on_device'):
            env = gua.GridUniverseEnv(**kw)
            if kat['level']:
                env.current_state = env.starting_states[pick]
            got = []
            for a in kat['actions']:
                o, r, d, info = getattr(env, method)(a)
                assert type(o) is int and type(r) is np.int64 and type(d) is bool and info is env.info
                assert env.current_state == o and env.done == d
                got.append([o, int(r), d])
            assert got == run['steps'], method
            env.close()


def test_facade_sees_edits_after_invalidate():
    env = gua.GridUniverseEnv()
    assert env.step(1)[0] == 1
    env.wall_grid[2] = 1
    env.lava_states.append(5)
    env.reward_matrix[5] = -10
    assert env.step(1)[0] == 2  # compiled grid: the edit is not seen yet (documented difference, docs/API.md)
    env.invalidate()
    env.current_state = 1
    assert env.step(1)[0] == 1 and env.step(2) == (5, -10, True, env.info)
    env.close()


# ------------------------------------------------------------------------------- transition-row rollout kernel
@pytest.fixture
def force_rows(gu_option):
    gu_option('rollout_rows', 1)  # every eligible launch takes gu_rollout_rows.hip


def _oracle_and_engine(meta, N, seed, env_id0=0):
    grid = C.Grid.from_lists(**meta)
    st = C.State(N, env_id0)
    eng = Engine(N, spec_of(meta), seed=seed, env_id0=env_id0)
    assert np.array_equal(eng.reset(), C.reset(grid, seed, st))
    return grid, st, eng


@pytest.mark.parametrize('name', ['c3_maze32', 'c4_lava32', 'c2_open8x8', 'c5_maze64', 'grid1x1', 'grid9x1', 'rect25x30_busy'])
def test_rows_kernel_equals_the_oracle(force_rows, name):
    """Every launch form of the transition-row kernel against the C oracle: uniform / stream actions, auto-reset on and
    off, int32 / packed / no trajectory + stats, launches of 1, 2, 15, 16, 17, 33 and 200 steps chained (resumability:
    head / body / tail of the 16-actions-per-word schedule), ragged batch sizes."""
    meta, _ = G.load_traj(name)
    single_start = len(meta['starts']) == 1
    for N in (1, 65, 1000):
        for auto in (True, False):
            grid, st, eng = _oracle_and_engine(meta, N, 21, env_id0=4096)
            with eng:
                eng.reserve_trajectory(200)
                rs = np.random.RandomState(N)
                for T in (1, 2, 15, 16, 17, 33, 200):
                    for policy in ('uniform', 'stream'):
                        acts = rs.randint(0, 4, (T, N)).astype(np.int32) if policy == 'stream' else None
                        if acts is not None:
                            eng.upload_actions(acts)
                        for traj in (True, 'packed', False):
                            eng.rollout(T, policy, auto, traj, stats=True)
                            want = C.rollout(grid, 21, st, T, auto, actions=acts, stats=True)
                            if traj is True:
                                got = eng.read_trajectory(0, T)
                            elif traj == 'packed':
                                got = eng.read_trajectory_packed(0, T)
                            else:
                                got = {}
                            for k in got:
                                assert np.array_equal(got[k], want[k]), (name, N, auto, T, policy, traj, k)
                            ret, eps = eng.read_stats()
                            assert np.array_equal(ret, want['ret']) and np.array_equal(eps, want['episodes'])
                            s = eng.get_state()
                            for k in ('pos', 'done', 'episode', 'tcount'):
                                assert np.array_equal(s[k], getattr(st, k)), (name, N, auto, T, policy, traj, k)
                            assert np.array_equal(eng.done_indices(), np.flatnonzero(st.done))
            if not single_start:
                break  # (multi-start auto-reset falls back to the general kernel; still checked once above)


def test_rows_kernel_first_step_honours_a_stored_state_that_disagrees_with_the_cell(force_rows):
    """gu_set_state can install done = 1 on a non-terminal cell and done = 0 on a terminal one, and a reset can land on a
    terminal start: the launch's first step must use the stored flag, like the general kernel and the oracle."""
    meta, _ = G.load_traj('c4_lava32')
    N = 512
    for starts in ([0], [16]):  # 16 is a lava cell: every reset lands on a terminal start
        m = dict(meta, starts=starts)
        grid, st, eng = _oracle_and_engine(m, N, 5)
        with eng:
            rs = np.random.RandomState(2)
            free = np.setdiff1d(np.arange(1024), meta['walls'])
            st.pos[:] = rs.choice(free, N)
            st.pos[::7] = 16
            st.done[:] = rs.randint(0, 2, N)
            st.episode[:] = rs.randint(0, 9, N)
            st.tcount[:] = rs.randint(0, 50, N)  # per-env step counts: the per-lane RNG schedule
            eng.set_state(pos=st.pos, done=st.done, episode=st.episode, tcount=st.tcount)
            eng.reserve_trajectory(40)
            for auto in (True, False, True):
                eng.rollout(40, 'uniform', auto, True, stats=True)
                want = C.rollout(grid, 5, st, 40, auto, stats=True)
                got = eng.read_trajectory(0, 40)
                assert all(np.array_equal(got[k], want[k]) for k in got), (starts, auto)
                s = eng.get_state()
                assert all(np.array_equal(s[k], getattr(st, k)) for k in ('pos', 'done', 'episode', 'tcount'))


def test_rows_and_general_kernel_agree_at_config_sizes(gu_option):
    """Config 3 at full size (65 536 envs x 1000 steps) through both kernels: identical trajectory digest, stats and state;
    and the default dispatch (rows for stats-only / packed, general for int32 rows) reproduces the reference's digest."""
    import hashlib
    import random
    random.seed(123)
    np.random.seed(123)
    env = gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True)
    N, T = 65536, 1000
    out = {}
    for rows in ('0', '1', '2'):  # general kernel / row-table kernel with its pair tables (two steps per round trip) / with the one-step table
        gu_option('rollout_rows', int(rows))
        with Engine(N, GridSpec.from_env(env), seed=123) as eng:
            eng.reset()
            eng.reserve_trajectory(T)
            eng.rollout(T, 'uniform', True, True, stats=True)
            tr = eng.read_trajectory(0, T)
            h = hashlib.sha256()
            for k in ('obs', 'reward', 'done'):
                h.update(np.ascontiguousarray(tr[k], dtype='<i4').tobytes())
            st = eng.get_state()
            out[rows] = (h.hexdigest(), eng.read_stats(), st)
            del tr
    assert out['0'][0] == out['1'][0] == out['2'][0] == G.load_json('digests.json')['c3_maze32_65536x1000']['sha256']
    for other in ('1', '2'):
        assert all(np.array_equal(a, b) for a, b in zip(out['0'][1], out[other][1]))
        assert all(np.array_equal(out['0'][2][k], out[other][2][k]) for k in out['0'][2])
    gu_option('rollout_rows', None)
    with Engine(N, GridSpec.from_env(env), seed=123) as eng:  # default dispatch: stats-only launch on the row table
        eng.reset()
        eng.rollout(T, 'uniform', True, False, stats=True)
        assert all(np.array_equal(a, b) for a, b in zip(eng.read_stats(), out['0'][1]))
        st = eng.get_state()
        assert all(np.array_equal(st[k], out['0'][2][k]) for k in st)


def test_rows_kernel_on_a_single_device_generated_maze(force_rows):
    """One maze carved on the device (its start cell is only known there): the row table must be built from that start."""
    W = H = 21
    N, T = 256, 300
    with Engine(N, GridSpec(W, H, [0], [W * H - 1], [], []), seed=4) as eng:
        eng.generate_mazes(1, W, H, 11)
        wall, start, goal = C.generate_maze(11, 0, W, H)
        spec = GridSpec(W, H, [start], [goal], [], np.flatnonzero(wall).tolist())
        grid = C.Grid(spec.W, spec.H, spec.wall, spec.lava, spec.goal, spec.reward, spec.starts)
        st = C.State(N)
        assert np.array_equal(eng.reset(), C.reset(grid, 4, st))
        # park every env next to the goal, so that episodes end (and restart from the device-chosen start) within the run
        near = [goal + d for d in (-W, 1, W, -1) if 0 <= goal + d < W * H and not wall[goal + d]][0]
        st.pos[:] = near
        eng.set_state(pos=st.pos)
        eng.reserve_trajectory(T)
        for traj in (True, False):
            eng.rollout(T, 'uniform', True, traj, stats=True)
            want = C.rollout(grid, 4, st, T, True, stats=True)
            if traj:
                got = eng.read_trajectory(0, T)
                assert all(np.array_equal(got[k], want[k]) for k in got)
            ret, eps = eng.read_stats()
            assert np.array_equal(ret, want['ret']) and np.array_equal(eps, want['episodes']) and eps.sum() > 0
            assert np.array_equal(eng.get_state()['episode'], st.episode)


@pytest.mark.parametrize('rows', ['1', '0'])
@pytest.mark.parametrize('name', ['c3_maze32', 'c4_lava32', 'c2_open8x8', 'rect25x30_busy', 'grid9x1'])
def test_table_policies_on_the_row_table_equal_the_oracle(gu_option, rows, name):
    """GU_POLICY_SAMPLE / GU_POLICY_GREEDY through the policy-row kernel (option rollout_rows = 1) and through the general kernel (= 0):
    stochastic policies with zero and one entries (thresholds at both ends of the range), one-hot policies (greedy), auto-reset
    on and off, every trajectory mode, chained launches of awkward lengths, a policy that changes between launches."""
    gu_option('rollout_rows', int(rows))
    meta, _ = G.load_traj(name)
    S = meta['W'] * meta['H']
    single_start = len(meta['starts']) == 1
    rs = np.random.RandomState(len(name))
    for N in (1, 130, 1000):
        for auto in (True, False):
            grid, st, eng = _oracle_and_engine(meta, N, 33, env_id0=77)
            with eng:
                eng.reserve_trajectory(120)
                for T in (1, 7, 8, 9, 24, 120):
                    pi = rs.dirichlet(np.ones(4) * 0.5, S)
                    hot = rs.rand(S) < 0.35  # rows with exact zeros and ones
                    pi[hot] = np.eye(4)[rs.randint(0, 4, int(hot.sum()))]
                    onehot = np.eye(4)[rs.randint(0, 4, S)]
                    for policy, table in (('sample', pi), ('greedy', onehot)):
                        eng.vi_set(np.zeros(S), table)
                        for traj in (True, 'packed', False):
                            eng.rollout(T, policy, auto, traj, stats=True)
                            want = C.rollout(grid, 33, st, T, auto, stats=True, pi=table)
                            got = eng.read_trajectory(0, T) if traj is True else eng.read_trajectory_packed(0, T) if traj else {}
                            for k in got:
                                assert np.array_equal(got[k], want[k]), (name, N, auto, T, policy, traj, k)
                            ret, eps = eng.read_stats()
                            assert np.array_equal(ret, want['ret']) and np.array_equal(eps, want['episodes']), (name, N, auto, T, policy, traj)
                            s = eng.get_state()
                            for k in ('pos', 'done', 'episode', 'tcount'):
                                assert np.array_equal(s[k], getattr(st, k)), (name, N, auto, T, policy, traj, k)
            if not single_start:
                break


@pytest.mark.parametrize('order', ['gu_first', 'torch_first'])
def test_rccl_is_taken_from_the_rocm_stack_libgu_runs_on(order):
    """A process may hold two ROCm stacks (the system one and the copy a PyTorch wheel bundles).  Whichever order they are
    loaded in, the gathered view must come up: gu_comm.hip takes the librccl next to the libamdhip64 that serves its own HIP
    calls.  (A caller that imports torch brings that second stack; bench.py itself no longer does.)"""
    import subprocess
    import sys
    code = '''
import sys
sys.path.insert(0, %r)
order = %r
if order == 'torch_first':
    import torch, torch.distributed
import numpy as np
import griduniverse_amd as gua
eng = gua.Engine(4096, gua.GridSpec(8, 8, [0], [63], [], []), seed=1)
if order == 'gu_first':
    import torch, torch.distributed
eng.reset()
eng.rollout(50, 'uniform', True, False)
eng.comm_init(1, 0, gua.Engine.comm_unique_id())
view = eng.allgather_view()
own = eng.read_outputs()
assert all(np.array_equal(a, b) for a, b in zip(view, own))
eng.comm_destroy()
eng.close()
print('VIEW-OK')
''' % (__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))), order)
    out = subprocess.run([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert out.returncode == 0 and b'VIEW-OK' in out.stdout, out.stdout.decode()[-2000:]


def test_trajectory_buffer_is_chosen_among_candidates_and_kept_when_large_enough(gu_option):
    """gu_reserve_trajectory probes candidate allocations for buffers of 64 MB and more and keeps the one HBM writes fastest;
    a buffer that is already large enough is kept.  Results never depend on which allocation was taken."""
    meta, _ = G.load_traj('c3_maze32')
    N, T = 32768, 256  # 3 x 32 MB planes
    grid = C.Grid.from_lists(**meta)
    outs = []
    for cand in ('1', '5'):
        gu_option('traj_candidates', int(cand))
        st = C.State(N)
        with Engine(N, spec_of(meta), seed=2) as eng:
            assert np.array_equal(eng.reset(), C.reset(grid, 2, st))
            eng.reserve_trajectory(T)
            n, best, worst = eng.trajectory_placement()
            assert (n == 1 and best == 0.0) if cand == '1' else (1 <= n <= 5 and 0.0 < best <= worst)  # (a candidate that is fast in absolute terms ends the search)
            eng.reserve_trajectory(T // 2)  # large enough already: same buffer, same placement record
            assert eng.trajectory_placement() == (n, best, worst)
            eng.rollout(T // 2, 'uniform', True, True)
            got = eng.read_trajectory(0, T // 2)
            want = C.rollout(grid, 2, st, T // 2, True)
            assert all(np.array_equal(got[k], want[k]) for k in got)
            eng.reserve_trajectory(2 * T)   # grows: chosen again
            eng.rollout(2 * T, 'uniform', True, True)
            want = C.rollout(grid, 2, st, 2 * T, True)
            got = eng.read_trajectory(0, 2 * T)
            assert all(np.array_equal(got[k], want[k]) for k in got)
            outs.append(got['obs'][-1].copy())
    assert np.array_equal(outs[0], outs[1])
    with Engine(64, spec_of(meta)) as eng:  # small buffers are simply allocated
        eng.reserve_trajectory(16)
        assert eng.trajectory_placement()[0] == 1
    # buffers of 256 MiB and more: when the back-to-back candidates all look alike the search continues behind spacers
    gu_option('traj_candidates', 2)
    gu_option('traj_far_candidates', 3)
    gu_option('traj_stride_mib', 1536)
    gu_option('traj_probe_all', 1)  # (without it the search ends where two back-to-back candidates are alike or one is fast)
    N, T = 65536, 400
    with Engine(N, spec_of(meta), seed=2) as eng:
        eng.reset()
        eng.reserve_trajectory(T)
        n, best, worst = eng.trajectory_placement()
        assert 1 <= n <= 5 and 0.0 < best <= worst
        eng.rollout(T, 'uniform', True, True)
        got = eng.read_trajectory(T - 1, 1)
        st = C.State(2048)
        C.reset(grid, 2, st)
        want = C.rollout(grid, 2, st, T, True)
        assert all(np.array_equal(got[k][0, :2048], want[k][T - 1]) for k in got)


@pytest.mark.gpu
@pytest.mark.parametrize('N,T', [(300, 2100), (1000, 1024), (70, 1025), (5000, 130)])
def test_long_action_streams_are_staged_in_lds_in_groups(gu_option, N, T):
    """GU_POLICY_STREAM with trajectory rows reads its packed action words from LDS, refilled every <= 64 words (1024
    steps) per lane: streams longer than one group, ending on and off a group / word boundary, against the C oracle on
    the general kernel (the row-table kernel is switched off), int32 and packed rows, then resumed with a second launch."""
    gu_option('rollout_rows', 0)
    meta, _ = G.load_traj('c3_maze32')
    grid, st, eng = _oracle_and_engine(meta, N, 5)
    acts = np.random.RandomState(T).randint(0, 4, (T, N)).astype(np.int32)
    with eng:
        eng.reserve_trajectory(T)
        eng.upload_actions(acts)
        for traj in (True, 'packed', True):
            eng.rollout(T, 'stream', True, traj, stats=True)
            want = C.rollout(grid, 5, st, T, True, actions=acts, stats=True)
            got = eng.read_trajectory(0, T) if traj is True else eng.read_trajectory_packed(0, T)
            for k in got:
                assert np.array_equal(got[k], want[k]), (traj, k)
            ret, eps = eng.read_stats()
            assert np.array_equal(ret, want['ret']) and np.array_equal(eps, want['episodes'])
            s = eng.get_state()
            for k in ('pos', 'done', 'episode', 'tcount'):
                assert np.array_equal(s[k], getattr(st, k)), (traj, k)


@pytest.mark.gpu
@pytest.mark.parametrize('nranks,N', [(2, 4096), (4, 1000), (8, 32768)])
def test_gathered_view_with_several_ranks_on_one_gpu_through_a_test_double_of_rccl(tmp_path, nranks, N):
    """RCCL refuses two ranks on one device and only one device is ever at hand, so gu_comm_init with nranks > 1 and the
    rank-major -> env-major unpack of gu_allgather_view had never run on hardware.  Here GU_RCCL_LIB points libgu at a test
    double (tests/c_abi/fake_rccl.hip: ranks = threads of one process, all-gather = rendezvous + device-to-device copies);
    every rank is an engine holding the shard [rank * N, (rank + 1) * N) of one batch.  Every rank's view must equal the
    single-engine batch of nranks * N envs (8 x 32 768 = config 4).  Own process: the library is chosen once per process."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = tmp_path / 'libfake_rccl.so'
    build = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-O2', '-o', str(lib),
                            os.path.join(root, 'tests', 'c_abi', 'fake_rccl.hip')], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert build.returncode == 0, build.stdout.decode()[-2000:]
    code = '''
import sys, threading
sys.path.insert(0, %r)
import numpy as np
import griduniverse_amd as gua
nranks, N = %d, %d
lava = [16 + 32 * r for r in range(24)]
spec = gua.GridSpec(32, 32, [0], [1023], lava, [])
whole = gua.Engine(nranks * N, spec, seed=9)
whole.reset()
whole.rollout(150, 'uniform', True, False)
want = whole.read_outputs()
shards = [gua.Engine(N, spec, seed=9, env_id0=r * N) for r in range(nranks)]
for e in shards:
    e.reset()
    e.rollout(150, 'uniform', True, False)
uid = gua.Engine.comm_unique_id()
views, errors = [None] * nranks, []
def run(r):
    try:
        shards[r].comm_init(nranks, r, uid)
        views[r] = shards[r].allgather_view()
        views[r] = shards[r].allgather_view()  # (a second gather reuses the communicator)
        shards[r].comm_destroy()
    except Exception as exc:
        errors.append((r, repr(exc)))
threads = [threading.Thread(target=run, args=(r,)) for r in range(nranks)]
[t.start() for t in threads]
[t.join(120) for t in threads]
assert not errors, errors
for r in range(nranks):
    assert views[r] is not None, r
    for got, exp, name in zip(views[r], want, ('obs', 'reward', 'done')):
        assert got.shape == (nranks * N,) and np.array_equal(got, exp), (r, name)
assert want[2].sum() > 0
print('VIEW-OK')
''' % (root, nranks, N)
    env = dict(os.environ, GU_RCCL_LIB=str(lib))
    out = subprocess.run([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600, env=env)
    assert out.returncode == 0 and b'VIEW-OK' in out.stdout, out.stdout.decode()[-3000:]


# ------------------------------------------------------------------------------- K-step rollout kernel (statistics only)
@pytest.mark.gpu
@pytest.mark.parametrize('name,force_k', [('c2_open8x8', ''), ('c2_open8x8', '2'), ('c2_open8x8', '4 copies=2'), ('c3_maze32', '2 copies=2'), ('c3_maze32', ''), ('c4_lava32', ''), ('grid1x1', ''), ('grid9x1', ''),
                                          ('rect25x30_busy', ''), ('c5_maze64', '')])
def test_k_step_kernel_equals_the_oracle(gu_option, name, force_k):
    """gu_rollout_multi.hip composes K consecutive transitions (K = 4 up to 64 cells, K = 2 up to ~2000; larger grids fall
    through to the row-table kernel) for uniform-policy and caller-supplied-stream launches that keep only statistics.  Chains of launches of every
    length around K and around the 16-action RNG word -- so that launches start at every offset inside a word and a group --
    with and without auto-reset, with and without stats, ragged batch sizes, shard offsets; per-env return, episodes, state,
    done compaction against the C oracle after every launch."""
    gu_option('rollout_multi', 1)  # also launches shorter than the default threshold
    if force_k:
        gu_option('rollout_multi_k', int(force_k.split()[0]))
        if 'copies=2' in force_k:
            gu_option('rollout_multi_copies', 2)  # (diagnostic layout: the table replicated across the LDS banks)
    meta, _ = G.load_traj(name)
    single_start = len(meta['starts']) == 1
    for N in (1, 65, 1000):
        for auto in (True, False):
            grid, st, eng = _oracle_and_engine(meta, N, 33, env_id0=70000)
            with eng:
                rs = np.random.RandomState(N + auto)
                for T in (1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 33, 64, 65, 100, 257, 1000):
                    for stats, policy in ((True, 'uniform'), (False, 'uniform'), (True, 'stream')):
                        acts = None
                        if policy == 'stream':  # the caller's stream, packed on the device into the RNG word's shape
                            acts = rs.randint(0, 4, (T, N)).astype(np.int32)
                            eng.upload_actions(acts)
                        eng.rollout(T, policy, auto, False, stats=stats)
                        want = C.rollout(grid, 33, st, T, auto, actions=acts, stats=True)
                        if stats:
                            ret, eps = eng.read_stats()
                            assert np.array_equal(ret, want['ret']) and np.array_equal(eps, want['episodes']), (name, N, auto, T)
                        s = eng.get_state()
                        for k in ('pos', 'done', 'episode', 'tcount'):
                            assert np.array_equal(s[k], getattr(st, k)), (name, N, auto, T, stats, k)
                        obs, rew, don = eng.read_outputs()
                        assert np.array_equal(obs, st.pos) and np.array_equal(rew, want['reward'][-1]) and np.array_equal(don, st.done), (name, N, auto, T)
                        assert np.array_equal(eng.done_indices(), np.flatnonzero(st.done))
            if not single_start:
                break


@pytest.mark.gpu
def test_k_step_kernel_with_stored_states_and_per_env_step_counts(gu_option):
    """A stored state may disagree with its cell (done = 1 on an open cell, done = 0 on a terminal one, a terminal start), and
    gu_set_state can give every env its own step count: first step on the planes, then the per-lane schedule."""
    gu_option('rollout_multi', 1)
    meta, _ = G.load_traj('c4_lava32')
    N = 700
    for starts in ([0], [16]):
        for uniform_counts in (True, False):
            m = dict(meta, starts=starts)
            grid, st, eng = _oracle_and_engine(m, N, 5)
            with eng:
                rs = np.random.RandomState(2)
                free = np.setdiff1d(np.arange(1024), meta['walls'])
                st.pos[:] = rs.choice(free, N)
                st.pos[::7] = 16
                st.done[:] = rs.randint(0, 2, N)
                st.episode[:] = rs.randint(0, 9, N)
                st.tcount[:] = 13 if uniform_counts else rs.randint(0, 50, N)
                eng.set_state(pos=st.pos, done=st.done, episode=st.episode, tcount=st.tcount)
                for auto in (True, False, True):
                    for T in (70, 3, 129):
                        eng.rollout(T, 'uniform', auto, False, stats=True)
                        want = C.rollout(grid, 5, st, T, auto, stats=True)
                        ret, eps = eng.read_stats()
                        assert np.array_equal(ret, want['ret']) and np.array_equal(eps, want['episodes']), (starts, uniform_counts, auto, T)
                        s = eng.get_state()
                        assert all(np.array_equal(s[k], getattr(st, k)) for k in ('pos', 'done', 'episode', 'tcount'))
                        obs, rew, don = eng.read_outputs()
                        assert np.array_equal(rew, want['reward'][-1])


@pytest.mark.gpu
def test_k_step_rows_and_general_kernels_agree_at_config_size(gu_option):
    """Config 3 at full size, statistics only, through the K-step, the row-table and the general kernel: identical returns,
    episode counts and final states."""
    meta, _ = G.load_traj('c3_maze32')
    outs = []
    for multi, rows in (('1', '1'), ('0', '1'), ('0', '0')):
        gu_option('rollout_multi', int(multi))
        gu_option('rollout_rows', int(rows))
        with Engine(65536, spec_of(meta), seed=123) as eng:
            eng.reset()
            for T in (1000, 999):
                eng.rollout(T, 'uniform', True, False, stats=True)
            ret, eps = eng.read_stats()
            s = eng.get_state()
            outs.append((ret, eps, s['pos'], s['done'], s['episode']) + tuple(eng.read_outputs()))
    for other in outs[1:]:
        assert all(np.array_equal(a, b) for a, b in zip(outs[0], other))
    assert outs[0][1].sum() > 0
