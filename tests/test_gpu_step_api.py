"""The C ABI's step / state surface on the device: actions and states validated by the kernels that consume them, the done ballots every done[]-writing kernel leaves, page-locked I/O pointers, the 24-bit move limit, RNG counters beyond 2^28, get_cells."""
import ctypes

import numpy as np
import pytest

from griduniverse_amd import _lib
from griduniverse_amd.engine import Engine
from griduniverse_amd.grid import GridSpec
from oracle import c_oracle as C
from tests import _golden as G
import griduniverse_amd as gua

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
        acts[bad] = [4, -5, 1 << 20]  # (-4 .. -1 are valid: the reference indexes a Python list, see the next test)

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


def test_negative_actions_address_the_move_list_from_its_end_like_the_reference():
    """env:148 indexes a Python list of four moves, so -1 is LEFT, -2 DOWN, -3 RIGHT, -4 UP; round 5's
    batched calls rejected them.  gu_step, an uploaded stream (single steps and the rollout that reads it packed) and
    gu_look_step_ahead take them; -5 and 4 are still refused."""
    meta, _ = G.load_traj('c4_lava32')
    grid = C.Grid.from_lists(**meta)
    N, T = 777, 64
    rs = np.random.RandomState(2)
    acts = rs.randint(-4, 4, (T, N)).astype(np.int32)
    for use in ('step', 'step_device', 'stream'):
        st = C.State(N)
        with Engine(N, spec_of(meta), seed=3) as eng:
            assert np.array_equal(eng.reset(), C.reset(grid, 3, st))
            want = C.rollout(grid, 3, st, T, True, actions=acts)
            same = C.State(N)
            C.reset(grid, 3, same)
            again = C.rollout(grid, 3, same, T, True, actions=acts & 3)
            assert all(np.array_equal(want[k], again[k]) for k in ('obs', 'reward', 'done'))  # the oracle: -k is 4 - k
            if use == 'step':
                got = [eng.step(acts[t], auto_reset=True) for t in range(T)]
                assert all(np.array_equal(got[t][0], want['obs'][t]) and np.array_equal(got[t][1], want['reward'][t]) for t in range(T))
            else:
                eng.upload_actions(acts)
                if use == 'step_device':
                    for t in range(T):
                        eng.step_device(t, auto_reset=True)
                else:
                    eng.reserve_trajectory(T)
                    eng.rollout(T, 'stream', True)
                    got = eng.read_trajectory(0, T)
                    assert all(np.array_equal(got[k], want[k]) for k in got)
            s = eng.get_state()
            assert np.array_equal(s['pos'], st.pos) and np.array_equal(s['done'], st.done) and np.array_equal(s['episode'], st.episode), use
            nxt, rew, don = eng.look_step_ahead(np.arange(64), np.full(64, -1))
            nx3, rw3, dn3 = eng.look_step_ahead(np.arange(64), np.full(64, 3))
            assert np.array_equal(nxt, nx3) and np.array_equal(rew, rw3) and np.array_equal(don, dn3)
            for bad in (-5, 4):
                with pytest.raises(gua.GuError):
                    eng.step(np.full(N, bad, np.int32))


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


@pytest.mark.parametrize('name,N', [('c3_maze32', 4096), ('c2_open8x8', 512), ('multistart_test_env', 640)])
def test_step_counts_have_64_bits_a_rollout_across_2_pow_32_steps_equals_the_oracle(name, N):
    """Round 5's count had 32 bits and the action stream repeated after 2^32 steps per env (97 s at the statistics-only rate).
    Now: every env at 2^32 - 500 steps, 1000 steps rolled -- the launch in which the count passes 2^32 runs on the general kernel,
    every step asking for the prefix of its own epoch -- then launches that lie wholly in epoch 1 (the epoch folded into the seed
    prefix by the launcher: the transition-row and K-step kernels as they are), all equal to the oracle; the steps behind the
    boundary are NOT steps 0 .. 499 over again; get_state reports counts beyond 2^32."""
    meta, _ = G.load_traj(name)
    grid = C.Grid.from_lists(**meta)
    S = meta['W'] * meta['H']
    pi = np.random.RandomState(4).dirichlet(np.ones(4), S)
    T = 1000
    for policy, kw in (('uniform', {}), ('sample', dict(pi=pi))):
        with Engine(N, spec_of(meta), seed=21) as eng:
            eng.vi_set(np.zeros(S), pi)
            eng.reserve_trajectory(T)
            fresh = C.State(N)
            assert np.array_equal(eng.reset(), C.reset(grid, 21, fresh))
            eng.rollout(500, policy, True)
            from_zero = eng.read_trajectory(0, 500)  # steps 0 .. 499 of every env
            assert all(np.array_equal(from_zero[k], v) for k, v in C.rollout(grid, 21, fresh, 500, True, **kw).items() if k in from_zero)
            # the same engine put back to its reset state, 500 steps short of 2^32
            eng.seed(21)
            st = C.State(N)
            assert np.array_equal(eng.reset(), C.reset(grid, 21, st))
            st.tcount[:] = 2 ** 32 - 500
            eng.set_state(tcount=st.tcount)
            eng.rollout(T, policy, True)
            got = eng.read_trajectory(0, T)
            want = C.rollout(grid, 21, st, T, True, **kw)
            assert all(np.array_equal(got[k], want[k]) for k in got), (policy, 'across the boundary')
            assert not np.array_equal(got['obs'][500:], from_zero['obs']), policy  # a 32-bit count would replay steps 0 .. 499 here
            state = eng.get_state()
            assert state['tcount'].dtype == np.uint64 and np.all(state['tcount'] == 2 ** 32 + 500) and np.array_equal(state['pos'], st.pos)
            # wholly inside epoch 1: rows, packed rows, statistics only -- whichever kernel the launch shape selects
            for traj, stats in ((True, False), ('packed', False), (False, True), (True, True)):
                eng.rollout(T, policy, True, trajectory=traj, stats=stats)
                want = C.rollout(grid, 21, st, T, True, stats=stats, **kw)
                if traj == 'packed':
                    got = eng.read_trajectory_packed(0, T)
                    assert all(np.array_equal(got[k], want[k]) for k in got), (policy, 'packed')
                elif traj:
                    got = eng.read_trajectory(0, T)
                    assert all(np.array_equal(got[k], want[k]) for k in got), (policy, traj, stats)
                if stats:
                    ret, fin = eng.read_stats()
                    assert np.array_equal(ret, want['ret']) and np.array_equal(fin, want['episodes']), (policy, 'stats')
                s = eng.get_state()
                assert all(np.array_equal(s[k], getattr(st, k)) for k in ('pos', 'done', 'episode', 'tcount')), (policy, traj, stats)


def test_envs_that_pass_2_pow_32_steps_at_different_moments():
    """Ragged step counts around the boundary (gu_set_state), and an env held back by a rejected action: every env changes epoch at
    its own step; launches straddle the boundary for as long as some env has not passed it."""
    meta, _ = G.load_traj('c4_lava32')
    grid = C.Grid.from_lists(**meta)
    N, S = 1024, 1024
    rs = np.random.RandomState(8)
    pi = rs.dirichlet(np.ones(4), S)
    with Engine(N, spec_of(meta), seed=5) as eng:
        eng.vi_set(np.zeros(S), pi)
        eng.reserve_trajectory(300)
        st = C.State(N)
        assert np.array_equal(eng.reset(), C.reset(grid, 5, st))
        st.tcount[:] = 2 ** 32 - 400 + rs.randint(0, 700, N)   # some already beyond, most in front of the boundary
        st.tcount[:64] = 2 ** 32 - 17                           # one wave at a common count (the unrolled schedule's entry test)
        eng.set_state(tcount=st.tcount)
        acts = rs.randint(0, 4, N).astype(np.int32)
        acts[5] = 7                                            # rejected: env 5 does not step, its count stays behind
        with pytest.raises(_lib.GuError):
            eng.step(acts, auto_reset=True)
        keep = (st.pos[5], st.done[5], st.episode[5], st.tcount[5])
        acts[5] = 0
        C.rollout(grid, 5, st, 1, True, actions=acts[None, :])
        st.pos[5], st.done[5], st.episode[5], st.tcount[5] = keep
        for policy, kw, T in (('uniform', {}, 300), ('sample', dict(pi=pi), 150), ('uniform', {}, 77), ('sample', dict(pi=pi), 300), ('uniform', {}, 300)):
            eng.rollout(T, policy, True)
            got = eng.read_trajectory(0, T)
            want = C.rollout(grid, 5, st, T, True, **kw)
            assert all(np.array_equal(got[k], want[k]) for k in got), (policy, T)
        s = eng.get_state()
        assert all(np.array_equal(s[k], getattr(st, k)) for k in ('pos', 'done', 'episode', 'tcount')) and s['tcount'].min() > 2 ** 32
        with pytest.raises(_lib.GuError, match='within 2\\^31'):
            eng.set_state(tcount=np.where(np.arange(N) % 2, 5, 2 ** 33).astype(np.uint64))


def test_get_cells_with_a_start_list_longer_than_the_grid():
    spec = GridSpec(2, 1, [0, 0, 0, 1, 0], [1], [], [])
    with Engine(8, spec) as eng:
        flags, reward, starts = eng.get_cells()
        assert starts.tolist() == [0, 0, 0, 1, 0] and reward.tolist() == [-1, 10]
