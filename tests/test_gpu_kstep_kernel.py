"""The K-step rollout kernel (csrc/gu_rollout_multi.hip): statistics-only launches on K-step transition tables, against the oracle and the other kernels."""
import numpy as np
import pytest

from griduniverse_amd.engine import Engine
from griduniverse_amd.grid import GridSpec
from oracle import c_oracle as C
from tests import _golden as G

pytestmark = pytest.mark.gpu

def spec_of(meta):
    return GridSpec(meta['W'], meta['H'], meta['starts'], meta['goals'], meta['lava'], meta['walls'], meta['reward'])


def _oracle_and_engine(meta, N, seed, env_id0=0):
    grid = C.Grid.from_lists(**meta)
    st = C.State(N, env_id0)
    eng = Engine(N, spec_of(meta), seed=seed, env_id0=env_id0)
    assert np.array_equal(eng.reset(), C.reset(grid, seed, st))
    return grid, st, eng


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
