"""The transition-row rollout kernel (csrc/gu_rollout_rows.hip): one-step tables, pair tables, table policies on policy-dependent rows, staged action streams -- against the oracle and the general kernel."""
import json

import numpy as np
import pytest

from griduniverse_amd import Engine, GridSpec, _lib
from griduniverse_amd.engine import Engine
from griduniverse_amd.grid import GridSpec
from oracle import c_oracle as C
from tests import _golden as G
import griduniverse_amd as gua

pytestmark = pytest.mark.gpu

def spec_of(meta):
    return GridSpec(meta['W'], meta['H'], meta['starts'], meta['goals'], meta['lava'], meta['walls'], meta['reward'])


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


@pytest.mark.parametrize('policy', ['uniform', 'stream', 'greedy', 'sample'])
def test_the_first_step_runs_on_the_table_behind_a_rollout_and_on_the_planes_behind_anything_else(force_rows, gu_option, policy):
    """A launch that follows another rollout takes its first step on the staged table (option rollout_entry, the default): the state
    a rollout leaves behind always agrees with its cell.  Behind a reset (here: onto a TERMINAL start cell), gu_set_state with flags
    that disagree with the cells, or a step call, it must not.  Every launch of such a sequence against the oracle, with the option on
    and off; both runs also equal each other in every row."""
    meta, _ = G.load_traj('c4_lava32')
    N, T = 777, 37
    S = meta['W'] * meta['H']
    rows = {}
    for entry in (1, 0):
        gu_option('rollout_entry', entry)
        m = dict(meta, starts=[16])  # a lava cell: every reset lands on a terminal start
        grid, st, eng = _oracle_and_engine(m, N, 11)
        out = []
        with eng:
            eng.reserve_trajectory(T)
            table = np.random.RandomState(3).dirichlet(np.ones(4), S)
            if policy == 'greedy':
                table = np.eye(4)[np.random.RandomState(3).randint(0, 4, S)]
            if policy in ('greedy', 'sample'):
                eng.vi_set(np.zeros(S), table)
            rs = np.random.RandomState(4)

            def launch(auto):
                acts = rs.randint(0, 4, (T, N)).astype(np.int32) if policy == 'stream' else None
                if acts is not None:
                    eng.upload_actions(acts)
                eng.rollout(T, policy, auto, True, stats=True)
                kw = dict(actions=acts) if policy == 'stream' else dict(pi=table) if policy != 'uniform' else {}
                want = C.rollout(grid, 11, st, T, auto, stats=True, **kw)
                got = eng.read_trajectory(0, T)
                assert all(np.array_equal(got[k], want[k]) for k in got), (policy, entry, len(out))
                s = eng.get_state()
                assert all(np.array_equal(s[k], getattr(st, k)) for k in ('pos', 'done', 'episode', 'tcount')), (policy, entry, len(out))
                out.append(got)

            launch(True)    # behind the reset onto a terminal start: planes
            launch(True)    # behind a rollout: table
            launch(False)   # ... without auto-reset: the absorbing rows
            launch(True)
            assert np.array_equal(eng.reset(), C.reset(grid, 11, st))
            launch(True)
            free = np.setdiff1d(np.arange(S), meta['walls'])
            st.pos[:] = rs.choice(free, N)
            st.done[:] = rs.randint(0, 2, N)  # flags that disagree with the cells
            eng.set_state(pos=st.pos, done=st.done)
            launch(True)
            launch(True)
            a = rs.randint(0, 4, N).astype(np.int32)
            obs, rew, done = eng.step(a)
            want = C.rollout(grid, 11, st, 1, False, actions=a[None, :])
            assert np.array_equal(obs, want['obs'][0]) and np.array_equal(done, want['done'][0])
            launch(True)
            launch(True)
        rows[entry] = out
    assert all(np.array_equal(x[k], y[k]) for x, y in zip(rows[0], rows[1]) for k in x)


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


def test_pair_tables_leave_the_same_rows_as_the_one_step_table_and_the_general_kernel(gu_option):
    """gu_rollout_rows.hip's pair tables (two env-steps per LDS round trip; uniform policy and caller-supplied streams, launches that
    write rows, one workgroup per CU at most): every launch shape through the pair tables (option rollout_rows = 1), the one-step
    table (= 2) and the general kernel (= 0) -- int32 and packed rows, with and without auto-reset, step counts that leave heads and
    tails around the 16-step action words, two launches in a row (the second starts inside a word), per-env statistics, final state."""
    meta, _ = G.load_traj('c4_lava32')
    rs = np.random.RandomState(5)
    cases = [(4096, 1000, 'uniform'), (4100, 777, 'uniform'), (2048, 17, 'uniform'), (3000, 33, 'stream'), (4096, 250, 'stream')]
    for N, T, policy in cases:
        acts = rs.randint(0, 4, size=(T, N)).astype(np.int32) if policy == 'stream' else None
        for auto in (True, False):
            for traj in (True, 'packed'):
                outs = {}
                for rows in (0, 1, 2):
                    gu_option('rollout_rows', rows)
                    with Engine(N, spec_of(meta), seed=21, env_id0=7) as eng:
                        eng.reset()
                        if acts is not None:
                            eng.upload_actions(acts)
                        eng.reserve_trajectory(T)
                        eng.rollout(T // 3 + 1, policy, auto, trajectory=traj, stats=True)  # the next launch starts inside an action word
                        if acts is not None:
                            eng.upload_actions(acts)
                        eng.rollout(T, policy, auto, trajectory=traj, stats=True)
                        tr = eng.read_trajectory(0, T) if traj is True else eng.read_trajectory_packed(0, T)
                        st = eng.get_state()
                        outs[rows] = [tr[k] for k in sorted(tr)] + [st[k] for k in sorted(st)] + list(eng.read_stats()) + [eng.done_indices()]
                for rows in (1, 2):
                    assert all(np.array_equal(a, b) for a, b in zip(outs[0], outs[rows])), (N, T, policy, auto, traj, rows)
    gu_option('rollout_rows', None)
    # the default dispatch at a config-4 shard (32 768 envs, int32 rows) and for packed rows at 65 536 envs is the pair path: oracle
    grid = C.Grid.from_lists(**meta)
    for N, traj in ((32768, True), (65536, 'packed')):
        T = 200
        with Engine(N, spec_of(meta), seed=4) as eng:
            eng.reset()
            eng.reserve_trajectory(T)
            eng.rollout(T, 'uniform', True, trajectory=traj)
            got = eng.read_trajectory(0, T) if traj is True else eng.read_trajectory_packed(0, T, unpack=True)
        st = C.State(2048, N - 2048)
        C.reset(grid, 4, st)
        want = C.rollout(grid, 4, st, T, True)
        for k in ('obs', 'reward', 'done'):
            assert np.array_equal(got[k][:, N - 2048:], want[k]), (N, traj, k)


@pytest.mark.gpu
@pytest.mark.parametrize('rows', ['1', '0'])
def test_sampled_policy_with_per_env_step_counts(gu_option, rows):
    """RNG stream 2 hashes one word per sixteen steps of an env; where the lanes of a wave are at ONE step count the kernels run
    an unrolled schedule, where gu_set_state gave every env its own they ask the lanes step by step.  Both against the C oracle:
    a common count that is no multiple of sixteen (head of single steps), ragged counts, counts across the 16-step groups."""
    gu_option('rollout_rows', int(rows))
    meta, _ = G.load_traj('c3_maze32')
    S = meta['W'] * meta['H']
    N = 777
    rs = np.random.RandomState(6)
    pi = rs.dirichlet(np.ones(4) * 0.6, S)
    for counts in ('common_13', 'ragged', 'common_4090'):
        grid, st, eng = _oracle_and_engine(meta, N, 21, env_id0=5)
        with eng:
            eng.reserve_trajectory(70)
            eng.vi_set(np.zeros(S), pi)
            st.tcount[:] = {'common_13': 13, 'common_4090': 4090}.get(counts, 0) if counts != 'ragged' else rs.randint(0, 100, N)
            eng.set_state(pos=st.pos, done=st.done, episode=st.episode, tcount=st.tcount)
            for T in (70, 5, 33):
                for traj in (True, False):
                    eng.rollout(T, 'sample', True, traj, stats=True)
                    want = C.rollout(grid, 21, st, T, True, stats=True, pi=pi)
                    if traj:
                        got = eng.read_trajectory(0, T)
                        for k in got:
                            assert np.array_equal(got[k], want[k]), (counts, T, k)
                    ret, eps = eng.read_stats()
                    assert np.array_equal(ret, want['ret']) and np.array_equal(eps, want['episodes']), (counts, T, traj)
                    s = eng.get_state()
                    assert all(np.array_equal(s[k], getattr(st, k)) for k in ('pos', 'done', 'episode', 'tcount')), (counts, T, traj)
