"""Parity tests proper: the HIP kernels, called through the C ABI (libgu.so via ctypes), against
 (a) golden vectors captured from the real reference (tests/golden),
 (b) the CPU oracle on seeded inputs at BASELINE sizes,
 (c) size-independent properties (resumability, shard independence, absorbing terminals).
Bit-exact throughout: this is integer / index work."""
import hashlib
import os

import numpy as np
import pytest

import griduniverse_amd as gua
from griduniverse_amd.engine import Engine
from griduniverse_amd.grid import GridSpec
from oracle import c_oracle as C
from tests import _golden as G

pytestmark = pytest.mark.gpu


def spec_of(meta):
    return GridSpec(meta['W'], meta['H'], meta['starts'], meta['goals'], meta['lava'], meta['walls'], meta['reward'])


def digest(obs, rew, don):
    h = hashlib.sha256()
    for a in (obs, rew, don):
        h.update(np.ascontiguousarray(a, dtype='<i4').tobytes())
    return h.hexdigest()


def fresh(meta, N=None):
    eng = Engine(N or meta['N'], spec_of(meta), env_id0=meta.get('env_id0', 0), seed=meta['seed'])
    return eng


# ------------------------------------------------------------------------------- golden trajectories
@pytest.mark.parametrize('name', G.traj_names())
def test_golden_step_api(name):
    """gu_step, one launch per env-step with host action rows (env:176-185)."""
    meta, z = G.load_traj(name)
    T, N = z['actions'].shape
    T = min(T, 300)
    with fresh(meta) as eng:
        assert np.array_equal(eng.reset(), z['first_state'])
        for t in range(T):
            obs, rew, don = eng.step(z['actions'][t], auto_reset=meta['auto_reset'])
            assert np.array_equal(obs, z['obs'][t]) and np.array_equal(rew, z['reward'][t]) \
                and np.array_equal(don, z['done'][t]), (name, t)


@pytest.mark.parametrize('name', G.traj_names())
@pytest.mark.parametrize('mode', ['stream', 'uniform', 'device', 'graph'])
def test_golden_fused_and_device_paths(name, mode):
    meta, z = G.load_traj(name)
    T, N = z['actions'].shape
    if mode == 'uniform' and 'RandomState' in meta['note']:
        pytest.skip('fixture actions come from numpy, not from the build RNG')
    with fresh(meta) as eng:
        eng.reset()
        if mode in ('stream', 'uniform'):
            if mode == 'stream':
                eng.upload_actions(z['actions'])
            eng.reserve_trajectory(T)
            eng.rollout(T, mode, meta['auto_reset'], trajectory=True, stats=True)
            out = eng.read_trajectory(0, T)
            for k in ('obs', 'reward', 'done'):
                assert np.array_equal(out[k], z[k]), (name, mode, k)
            assert digest(out['obs'], out['reward'], out['done']) == meta['sha256']
            ret, eps = eng.read_stats()
            assert np.array_equal(ret, z['reward'].sum(0)) and np.array_equal(eps, z['done'].sum(0))
        else:
            eng.upload_actions(z['actions'])
            if mode == 'graph':
                eng.step_graph(0, T, meta['auto_reset'])
            else:
                for t in range(T):
                    eng.step_device(t, meta['auto_reset'])
        obs, rew, don = eng.read_outputs()
        assert np.array_equal(obs, z['obs'][-1]) and np.array_equal(rew, z['reward'][-1]) and np.array_equal(don, z['done'][-1])
        st = eng.get_state()
        assert np.all(st['tcount'] == T) and np.array_equal(st['pos'], z['obs'][-1])


@pytest.mark.parametrize('name', sorted(G.load_json('digests.json')))
def test_reference_digests(name):
    """sha256 of N envs x 1000 steps of the REAL reference: 4096 envs for C2/C3/C4 and the full 65 536 envs of C3."""
    d = G.load_json('digests.json')[name]
    with fresh(d) as eng:
        eng.reset()
        eng.reserve_trajectory(d['T'])
        if 'stream_seed' in d:  # a caller-supplied stream (numpy's RandomState), stepped by the reference: the STREAM policy
            eng.upload_actions(np.random.RandomState(d['stream_seed']).randint(0, 4, size=(d['T'], d['N'])).astype(np.int32))
            outs = []
            for rows in (1, 0):  # the row-table kernel and the general kernel (packed words staged in LDS)
                eng.set_option('rollout_rows', rows)
                eng.seed(d['seed'])
                eng.reset()
                eng.rollout(d['T'], 'stream', d['auto_reset'])
                outs.append(eng.read_trajectory(0, d['T']))
            assert all(np.array_equal(outs[0][k], outs[1][k]) for k in outs[0])
            out = outs[0]
        else:
            eng.rollout(d['T'], 'uniform', d['auto_reset'])
            out = eng.read_trajectory(0, d['T'])
    assert digest(out['obs'], out['reward'], out['done']) == d['sha256']
    assert int(out['reward'].sum()) == d['sum_reward'] and int(out['done'].sum()) == d['sum_done']


# ------------------------------------------------------------------------------- BASELINE sizes vs the oracle
FULL = [('c2_open8x8', 4096, 1000), ('c3_maze32', 65536, 1000), ('c4_lava32', 32768, 1000), ('c5_maze64', 65536, 512)]


@pytest.mark.parametrize('name,N,T', FULL)
def test_full_size_rollout_equals_oracle(name, N, T):
    meta, _ = G.load_traj(name)
    grid = C.Grid.from_lists(**meta)
    seed = 20260 + N
    st = C.State(N)
    C.reset(grid, seed, st)
    want = C.rollout(grid, seed, st, T, True, stats=True)
    with Engine(N, spec_of(meta), seed=seed) as eng:
        eng.reset()
        eng.reserve_trajectory(T)
        eng.rollout(T, 'uniform', True, trajectory=True, stats=True)
        got = eng.read_trajectory(0, T)
        ret, eps = eng.read_stats()
        state = eng.get_state()
    for k in ('obs', 'reward', 'done'):
        assert np.array_equal(got[k], want[k]), (name, k)
    assert np.array_equal(ret, want['ret']) and np.array_equal(eps, want['episodes'])
    assert np.array_equal(state['pos'], st.pos) and np.array_equal(state['done'], st.done)
    assert np.array_equal(state['episode'], st.episode) and np.array_equal(state['tcount'], st.tcount)


def test_sharded_batch_equals_single_batch():
    """C4: envs [g*N/8, (g+1)*N/8) on 'rank' g reproduce the single-engine batch byte for byte."""
    meta, _ = G.load_traj('c4_lava32')
    N, T, shards = 4096, 300, 8
    with Engine(N, spec_of(meta), seed=4) as eng:
        eng.reset()
        eng.reserve_trajectory(T)
        eng.rollout(T, 'uniform', True)
        whole = eng.read_trajectory(0, T)
    per = N // shards
    for g in (0, 3, 7):
        with Engine(per, spec_of(meta), env_id0=g * per, seed=4) as eng:
            eng.reset()
            eng.reserve_trajectory(T)
            eng.rollout(T, 'uniform', True)
            part = eng.read_trajectory(0, T)
        for k in whole:
            assert np.array_equal(part[k], whole[k][:, g * per:(g + 1) * per]), (g, k)


def test_rollout_is_resumable_and_mixes_with_steps():
    meta, z = G.load_traj('rect25x30_busy')
    T, N = z['actions'].shape
    with fresh(meta) as eng:
        eng.reset()
        got = {k: [] for k in ('obs', 'reward', 'done')}
        t = 0
        for chunk in (1, 17, 100):
            eng.reserve_trajectory(chunk)
            eng.rollout(chunk, 'uniform', True)
            out = eng.read_trajectory(0, chunk)
            for k in got:
                got[k].append(out[k])
            t += chunk
        for _ in range(20):  # single steps continue the same stream
            obs, rew, don = eng.step(z['actions'][t], auto_reset=True)
            for k, v in zip(('obs', 'reward', 'done'), (obs, rew, don)):
                got[k].append(v[None])
            t += 1
        eng.reserve_trajectory(T - t)
        eng.rollout(T - t, 'uniform', True)
        out = eng.read_trajectory(0, T - t)
        for k in got:
            got[k].append(out[k])
            assert np.array_equal(np.concatenate(got[k]), z[k]), k


def test_ragged_batch_sizes():
    """N not a multiple of the wave / block size, N = 1, and a single partial wave."""
    meta, _ = G.load_traj('c4_lava32')
    grid = C.Grid.from_lists(**meta)
    for N in (1, 2, 63, 65, 255, 257, 1000):
        st = C.State(N, 5)
        C.reset(grid, 9, st)
        want = C.rollout(grid, 9, st, 200, True)
        with Engine(N, spec_of(meta), env_id0=5, seed=9) as eng:
            eng.reset()
            eng.reserve_trajectory(200)
            eng.rollout(200, 'uniform', True)
            got = eng.read_trajectory(0, 200)
            assert sorted(eng.done_indices().tolist()) == np.flatnonzero(st.done).tolist()
        for k in ('obs', 'reward', 'done'):
            assert np.array_equal(got[k], want[k]), (N, k)


# ------------------------------------------------------------------------------- reset / state / compaction
def test_masked_reset_explicit_choice_and_done_compaction():
    meta, _ = G.load_traj('multistart_test_env')
    N = 777
    grid = C.Grid.from_lists(**meta)
    st = C.State(N)
    with Engine(N, spec_of(meta), seed=31) as eng:
        assert np.array_equal(eng.reset(), C.reset(grid, 31, st))
        C.rollout(grid, 31, st, 9, False)
        eng.rollout(9, 'uniform', False, trajectory=False)
        s = eng.get_state()
        assert np.array_equal(s['pos'], st.pos) and np.array_equal(s['done'], st.done)
        idx = eng.done_indices()
        assert idx.dtype == np.int32 and np.array_equal(idx, np.flatnonzero(st.done)) and len(idx) > 0
        # reset only the finished envs (device-side), then a masked host reset with explicit choices
        C.reset(grid, 31, st, mask=st.done.astype(bool))
        eng.reset_done()
        s = eng.get_state()
        assert np.array_equal(s['pos'], st.pos) and not s['done'].any() and np.array_equal(s['episode'], st.episode)
        assert len(eng.done_indices()) == 0
        mask = (np.arange(N) % 3 == 0)
        choice = (np.arange(N) % 2).astype(np.int32)
        obs = eng.reset(mask, choice)
        want = st.pos.copy()
        want[mask] = np.asarray(meta['starts'])[choice[mask]]
        assert np.array_equal(obs, want)
        assert np.array_equal(eng.get_state()['episode'], st.episode + mask.astype(np.uint32))
        with pytest.raises(gua.GuError):
            eng.reset(None, np.full(N, 2, np.int32))


def test_set_state_round_trip_and_absorbing_terminals():
    meta, _ = G.load_traj('c4_lava32')
    N = 512
    rs = np.random.RandomState(1)
    free = np.setdiff1d(np.arange(1024), meta['walls'])
    pos = rs.choice(free, N).astype(np.int32)
    with Engine(N, spec_of(meta), seed=8) as eng:
        eng.set_state(pos=pos, done=np.zeros(N, np.int32), episode=np.arange(N, dtype=np.uint32),
                      tcount=np.full(N, 160, np.uint64))
        s = eng.get_state()
        assert np.array_equal(s['pos'], pos) and np.all(s['tcount'] == 160) and np.array_equal(s['episode'], np.arange(N))
        lava = np.array(meta['lava'][:N // 2] * 40, np.int32)[:N]
        eng.set_state(pos=lava)
        for a in range(4):  # absorbing: state stays, reward repeats, done stays (quirk 1)
            obs, rew, don = eng.step(np.full(N, a, np.int32))
            assert np.array_equal(obs, lava) and np.all(rew == -10) and np.all(don == 1)
        with pytest.raises(gua.GuError):
            eng.set_state(pos=np.full(N, 1024, np.int32))
        with pytest.raises(gua.GuError):
            eng.step(np.full(N, 4, np.int32))
        with pytest.raises(gua.GuError):
            eng.step(np.full(N, -5, np.int32))  # (-4 .. -1 are the move list from its end, env:148)


def test_look_step_ahead_tables():
    t = G.load_json('render_quirks.json')['quirks']['lsa_table_6x5']
    spec = spec_of(t['spec'])
    s, a = np.meshgrid(np.arange(30), np.arange(4), indexing='ij')
    with Engine(4, spec) as eng:
        for care in (True, False):
            n, r, d = eng.look_step_ahead(s.ravel(), a.ravel(), care)
            want = np.array(t['table'][str(care)], dtype=np.int64).reshape(-1, 3)
            assert np.array_equal(np.stack([n, r, d], 1), want)
    for name in ('maze101', 'rect25x30_busy', 'wide40x12'):
        meta, _ = G.load_traj(name)
        grid = C.Grid.from_lists(**meta)
        S = meta['W'] * meta['H']
        s, a = np.repeat(np.arange(S), 4), np.tile(np.arange(4), S)
        with Engine(1, spec_of(meta)) as eng:
            for care in (True, False):
                got = eng.look_step_ahead(s, a, care)
                want = C.look_step_ahead(grid, s, a, care)
                for g_, w_ in zip(got, want):
                    assert np.array_equal(g_, w_), (name, care)


def test_random_grids_property():
    """Random grids (walls / lava / goals / starts anywhere, incl. overlaps), random batch sizes, every rollout
    policy that needs no table, with and without auto-reset, vs the oracle."""
    import os
    trials = int(os.environ.get('GU_FUZZ_TRIALS', '60'))  # e.g. GU_FUZZ_TRIALS=2000 for a one-off soak
    rs = np.random.RandomState(int(os.environ.get('GU_FUZZ_SEED', '2026')))
    for trial in range(trials):
        W, H = int(rs.randint(1, 70)), int(rs.randint(1, 40))
        S = W * H
        pick = lambda k: [int(x) for x in rs.choice(S, size=min(S, int(k)), replace=False)]  # noqa: E731
        walls, lava, goals, starts = pick(rs.randint(0, S // 3 + 1)), pick(rs.randint(0, 6)), pick(rs.randint(1, 5)), pick(rs.randint(1, 6) if trial % 3 else 1)
        meta = dict(W=W, H=H, walls=walls, lava=lava, goals=goals, starts=starts)
        grid = C.Grid.from_lists(**meta)
        spec = GridSpec(W, H, starts, goals, lava, walls)
        N, T, seed = int(rs.randint(1, 1500)), int(rs.randint(1, 200)), int(rs.randint(0, 2 ** 62))
        auto, stream, id0 = bool(trial % 2), bool((trial // 2) % 2), int(rs.randint(0, 2 ** 31))
        acts = rs.randint(0, 4, (T, N)).astype(np.int32) if stream else None
        st = C.State(N, id0)
        C.reset(grid, seed, st)
        want = C.rollout(grid, seed, st, T, auto, actions=acts, stats=True)
        with Engine(N, spec, env_id0=id0, seed=seed) as eng:
            eng.reset()
            if stream:
                eng.upload_actions(acts)
            eng.reserve_trajectory(T)
            # int32 rows, packed rows, stats only: the last two take the transition-row kernel when the grid has one start
            # cell (or no auto-reset); option rollout_rows = 1 sends the int32 rows there too
            mode = (True, 'packed', False)[(trial // 4) % 3]
            if mode is True:
                # int32 rows: forced store pacing (a schedule of 1 .. 600 ticks of 10 ns per 16 steps), and the general kernel where the
                # row-table kernel would take the launch -- none of it may change a byte
                eng.set_option('rollout_pace', (None, 0, int(rs.randint(1, 600)))[trial % 3])
                eng.set_option('rollout_rows', (None, 0, 1, 2)[(trial // 12) % 4])
            eng.rollout(T, 'stream' if stream else 'uniform', auto, trajectory=mode, stats=True)
            got = eng.read_trajectory(0, T) if mode is True else eng.read_trajectory_packed(0, T) if mode else {}
            ret, eps = eng.read_stats()
            state = eng.get_state()
        for k in got:
            assert np.array_equal(got[k], want[k]), (trial, W, H, N, T, k)
        assert np.array_equal(ret, want['ret']) and np.array_equal(eps, want['episodes'])
        assert np.array_equal(state['pos'], st.pos) and np.array_equal(state['episode'], st.episode) and np.array_equal(state['done'], st.done)


@pytest.mark.parametrize('W,H', [(300, 300), (190, 190), (512, 400)])
def test_grid_larger_than_lds_uses_the_global_path(W, H):
    """Beyond 32 767 cells the two record planes no longer fit 64 KiB: the uniform / stream rollouts keep the flags
    plane alone in (up to 160 KB of) LDS (190x190, 300x300); 512x400 = 204 800 cells reads the records from L2.  The
    single step and the table policies take the L2 path at all three sizes."""
    rs = np.random.RandomState(3)
    walls = [int(x) for x in rs.choice(W * H, W * H // 5, replace=False)]
    meta = dict(W=W, H=H, walls=walls, lava=[5, 777], goals=[W * H - 1, 4000], starts=[0, 301, 30000])
    grid = C.Grid.from_lists(**meta)
    st = C.State(300)
    C.reset(grid, 1, st)
    want = C.rollout(grid, 1, st, 400, True)
    with Engine(300, GridSpec(W, H, meta['starts'], meta['goals'], meta['lava'], walls), seed=1) as eng:
        eng.reset()
        eng.reserve_trajectory(400)
        eng.rollout(400, 'uniform', True)
        got = eng.read_trajectory(0, 400)
        for k in got:
            assert np.array_equal(got[k], want[k])
        obs, rew, don = eng.step(np.zeros(300, np.int32), auto_reset=True)
        w2 = C.rollout(grid, 1, st, 1, True, actions=np.zeros((1, 300), np.int32))
        assert np.array_equal(obs, w2['obs'][0]) and np.array_equal(don, w2['done'][0])
        acts = rs.randint(0, 4, (64, 300)).astype(np.int32)  # caller-supplied stream, packed rows, stats
        w3 = C.rollout(grid, 1, st, 64, True, actions=acts, stats=True)
        eng.upload_actions(acts)
        eng.rollout(64, 'stream', True, trajectory='packed' if W * H <= 65536 else True, stats=True)
        got = eng.read_trajectory_packed(0, 64) if W * H <= 65536 else eng.read_trajectory(0, 64)
        for k in ('obs', 'reward', 'done'):
            assert np.array_equal(got[k], w3[k]), k
        ret, eps = eng.read_stats()
        assert np.array_equal(ret, w3['ret']) and np.array_equal(eps, w3['episodes'])


# ------------------------------------------------------------------------------- facade (N = 1 drop-in)
def _triple(step):
    o, r, d, _ = step
    return [int(o), int(r), bool(d)]


@pytest.mark.parametrize('kat', G.load_json('kat.json'), ids=lambda k: k['name'])
def test_facade_reference_kats(kat):
    for pick, run in enumerate(kat['runs']):
        kw = dict(kat['kwargs'])
        if 'grid_shape' in kw:
            kw['grid_shape'] = tuple(kw['grid_shape'])
        if kat['level']:
            kw['custom_world_fp'] = G.level_path(kat['level'])
        env = gua.GridUniverseEnv(**kw)
        if kat['level']:
            env.current_state = env.starting_states[pick]
        assert int(env.current_state) == run['first_state']
        assert [_triple(env.step(a)) for a in kat['actions']] == run['steps']
        env.close()


def test_facade_quirks_types_and_trail():
    q = G.load_json('render_quirks.json')['quirks']
    env = gua.GridUniverseEnv()
    out = env.step(1)
    assert [type(x).__name__ for x in out] == q['types'] and out[3] is env.info
    assert env.previous_state == 0 and env.current_state == 1 and len(env.last_n_states) == 1
    env.current_state = 11
    assert [_triple(env.step(a)) for a in [2, 0, 3, 1]] == q['absorbing']
    env = gua.GridUniverseEnv(walls=[0])
    assert [_triple(env.step(a)) for a in [1, 3, 2, 0]] == q['start_on_wall']
    env = gua.GridUniverseEnv(goal_states=[5], walls=[5])
    seq = [_triple(env.step(a)) for a in [1, 2, 2, 0]]
    env.current_state = 5
    seq.append(_triple(env.step(1)))
    assert seq == q['goal_is_wall']
    env = gua.GridUniverseEnv(goal_states=[1, 15], lava_states=[1])
    assert [_triple(env.step(a)) for a in [1, 1]] == q['goal_and_lava']
    env = gua.GridUniverseEnv(goal_states=[-1])
    env.current_state = 14
    assert [_triple(env.step(a)) for a in [1, 1, 3]] == q['negative_goal']['steps']
    env = gua.GridUniverseEnv(lava_states=[1])
    got = [[int(x) if k < 2 else bool(x) for k, x in enumerate(env.look_step_ahead(s, a, c))]
           for (s, a, c) in [(1, 1, True), (1, 1, False), (15, 3, True), (15, 3, False), (1, 2, False), (0, 1, False)]]
    assert got == q['care_about_terminal_false']
    assert type(env.look_step_ahead(0, 1)[1]).__name__ == 'int64'
    for c in [c for c in G.load_json('errors.json') if 'step_action' in c]:
        env = gua.GridUniverseEnv()
        env.current_state = c['from_state']
        if c['error']:
            with pytest.raises(IndexError):
                env.step(c['step_action'])
        else:
            assert _triple(env.step(c['step_action'])) == c['result']
    env = gua.GridUniverseEnv(grid_shape=(30, 30))
    for _ in range(520):
        env.step(1)
    assert len(env.last_n_states) == 500


def test_facade_c1_default_rollout():
    """Config 1: run_default_griduniverse() shape -- 1 env, 1000 random steps, reset on done."""
    meta, z = G.load_traj('c1_default4x4')
    env = gua.GridUniverseEnv()
    env.reset()
    done = False
    for t in range(1000):
        if done:
            env.reset()
        o, r, done, _ = env.step(int(z['actions'][t, 0]))
        assert (o, r, done) == (z['obs'][t, 0], z['reward'][t, 0], bool(z['done'][t, 0]))


def test_vec_env_surface():
    env = gua.VecGridUniverse(256, grid_shape=(8, 8), seed=2, auto_reset=True)
    meta, z = G.load_traj('c2_open8x8')
    obs = env.reset()
    assert obs.shape == (256,) and obs.dtype == np.int32 and not obs.any()
    o, r, d, info = env.step(np.pad(z['actions'][0], (0, 192)))
    assert d.dtype == bool and np.array_equal(o[:64], z['obs'][0]) and info == {}
    out = env.rollout(64, stats=True)
    assert out['obs'].shape == (64, 256) and out['ret'].shape == (256,)
    out = env.rollout(8, actions=np.ones((8, 256), np.int32))
    assert out['obs'].shape == (8, 256)
    state = env.get_state()
    a = env.rollout(32, trajectory='packed')
    env.set_state(pos=state['pos'], done=state['done'], episode=state['episode'], tcount=state['tcount'])
    b = env.rollout(32)
    assert all(np.array_equal(a[k], b[k]) for k in ('obs', 'reward', 'done'))
    env.close()


# ------------------------------------------------------------------------------- RCCL gathered view
def test_rccl_gathered_view_single_rank():
    """ncclAllGather path of csrc/gu_comm.hip with a 1-rank communicator (the GPU box has one device;
    the 2-rank layout logic is covered on CPU by tests/test_multiprocess.py)."""
    from griduniverse_amd.parallel import ShardedVecGridUniverse
    env = ShardedVecGridUniverse(4096, rank=0, world_size=1, seed=5, auto_reset=True, grid_shape=(32, 32),
                                 lava_states=[16 + 32 * r for r in range(24)])
    env.reset()
    env.rollout(100, trajectory=False)
    obs, rew, don = env.gathered_view()
    own = env.local.engine.read_outputs()
    assert obs.shape == (4096,) and np.array_equal(obs, own[0]) and np.array_equal(rew, own[1])
    assert don.dtype == bool and np.array_equal(don, own[2].astype(bool))
    env.step(np.zeros(4096, np.int32))
    obs2, _, _ = env.gathered_view()  # communicator is reused
    assert np.array_equal(obs2, env.local.engine.read_outputs()[0])
    env.close()


def test_integration_md_ctypes_stub_runs():
    """The reference-side ctypes binding shown in INTEGRATION.md section 2, executed verbatim against an object
    with the reference env's attributes, equals the oracle."""
    import os
    import re
    from griduniverse_amd import _lib
    from oracle.ref_env import OracleGridUniverseEnv
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, 'INTEGRATION.md')).read()
    code = re.search(r'```python\n(# core/envs/griduniverse_gpu\.py.*?)```', text, re.S).group(1)
    ns = {}
    exec(code.replace("C.CDLL('libgu.so')", "C.CDLL(%r)" % _lib.LIB_PATH), ns)
    env = OracleGridUniverseEnv(custom_world_fp=G.level_path('test_env.txt'))
    batched = ns['BatchedGridUniverse'](env, 500, seed=9, first_global_env=100)
    grid = C.Grid.from_env(env)
    st = C.State(500, 100)
    assert np.array_equal(batched.reset(), C.reset(grid, 9, st))
    acts = (np.arange(500) % 4).astype(np.int32)
    o, r, d, _ = batched.step(acts, auto_reset=True)
    w = C.rollout(grid, 9, st, 1, True, actions=acts[None, :])
    assert np.array_equal(o, w['obs'][0]) and np.array_equal(r, w['reward'][0]) and np.array_equal(d, w['done'][0].astype(bool))
    obs, rew, done = batched.rollout(64)
    want = C.rollout(grid, 9, st, 64, True)
    assert np.array_equal(obs, want['obs']) and np.array_equal(rew, want['reward']) and np.array_equal(done, want['done'])
    batched.close()


# ------------------------------------------------------------------------------- C-ABI error behaviour
def test_c_abi_error_codes():
    """Misuse returns negative codes + a message, never crashes (include/gu.h conventions)."""
    import ctypes
    from griduniverse_amd import _lib
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.gu_create(0, 0, 0, ctypes.byref(h)) == -1 and not h.value            # GU_ERR_INVALID
    assert lib.gu_create(99, 8, 0, ctypes.byref(h)) == -2 and 'not present' in _lib.last_error()
    assert lib.gu_create(0, 8, 2 ** 32, ctypes.byref(h)) == -1
    assert lib.gu_destroy(None) == 0
    assert lib.gu_sync(None) == -1 and 'null handle' in _lib.last_error()
    assert lib.gu_create(0, 8, 0, ctypes.byref(h)) == 0 and h.value
    acts = np.zeros(8, np.int32)
    out = np.zeros(8, np.int32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    assert lib.gu_step(h, p(acts), 0, p(out), None, None) == -4 and 'gu_set_grid' in _lib.last_error()   # GU_ERR_STATE
    assert lib.gu_reset(h, None, None, None) == -4
    assert lib.gu_rollout(h, 10, 0, 0) == -4
    lib.gu_destroy(h)
    meta, _ = G.load_traj('c2_open8x8')
    with Engine(8, spec_of(meta)) as eng:
        hh = eng._h
        assert lib.gu_rollout(hh, 10, 0, 2) == -4 and 'gu_reserve_trajectory' in _lib.last_error()
        assert lib.gu_rollout(hh, 10, 7, 0) == -1
        assert lib.gu_rollout(hh, 0, 0, 0) == -1 and lib.gu_rollout(hh, 10, 0, 64) == -1
        assert lib.gu_rollout(hh, 10, 1, 0) == -4          # no action stream uploaded
        assert lib.gu_rollout(hh, 10, 2, 0) == -4          # no policy table
        assert lib.gu_step_device(hh, 0, 0) == -4
        assert lib.gu_read_stats(hh, None, None) == -4
        assert lib.gu_vi_sweep(hh, 1.0, 1, 1, None) == -4
        done = ctypes.c_int32(0)
        assert lib.gu_vi_eval_run(hh, 1.0, 1e-3, 5, ctypes.byref(done), None) == -4 and 'gu_vi_set' in _lib.last_error()
        frame = np.zeros((8 * 4, 8 * 4, 3), np.uint8)
        assert lib.gu_render_policy_rgb(hh, 4, p(frame)) == -4
        eng.vi_set(np.zeros(64), np.ones((64, 4)) / 4)
        assert lib.gu_vi_eval_run(hh, 1.0, 1e-3, 5, None, None) == -1 and lib.gu_vi_eval_run(hh, 1.0, 1e-3, -1, ctypes.byref(done), None) == -1
        assert lib.gu_render_policy_rgb(hh, 0, p(frame)) == -1 and lib.gu_render_policy_rgb(hh, 4, None) == -1
        assert lib.gu_render_policy_rgb(hh, 4, p(frame)) == 0
        assert lib.gu_allgather_view(hh, None, None, None) == -4
        assert lib.gu_step(hh, None, 0, None, None, None) == -1
        bad = np.full((2, 8), 5, np.int32)
        assert lib.gu_upload_actions(hh, p(bad), 2) == -1
        planes = eng.spec.planes()
        starts = np.array([64], np.int32)
        assert lib.gu_set_grid(hh, 8, 8, 1, p(planes['wall']), p(planes['goal']), p(planes['lava']), None, None, p(starts), 1) == -1
        assert lib.gu_set_grid(hh, 8, 8, 2, p(planes['wall']), p(planes['goal']), p(planes['lava']), None, None, p(starts), 1) == -1
        # derived reward planes (NULL, NULL) give the same engine as explicit ones
        starts = np.array([0], np.int32)
        assert lib.gu_set_grid(hh, 8, 8, 1, p(planes['wall']), p(planes['goal']), p(planes['lava']), None, None, p(starts), 1) == 0
        eng.seed(2)
        eng.reset()
        _, z = G.load_traj('c2_open8x8')
        obs, rew, don = eng.step(z['actions'][0][:8])
        assert np.array_equal(obs, z['obs'][0][:8]) and np.array_equal(rew, z['reward'][0][:8])


def test_stream_rollout_all_lengths():
    """The packed-word pipeline of the STREAM policy (16 two-bit actions per word, loaded one word ahead): every T around
    the word size, with and without trajectory; a shorter upload replaces the stream."""
    meta, z = G.load_traj('rect25x30_busy')
    grid = C.Grid.from_lists(**meta)
    N = 200
    acts = z['actions'][:40, :64].repeat(4, axis=1)[:, :N].copy()
    with Engine(N, spec_of(meta), seed=3) as eng:
        for T in (1, 7, 8, 9, 15, 16, 17, 24, 31, 32, 33, 40):
            for traj in (True, False):
                st = C.State(N)
                C.reset(grid, 3, st)
                eng.seed(3)
                eng.reset()
                want = C.rollout(grid, 3, st, T, True, actions=acts[:T])
                eng.upload_actions(acts[:T])
                eng.reserve_trajectory(T)
                eng.rollout(T, 'stream', True, trajectory=traj)
                if traj:
                    got = eng.read_trajectory(0, T)
                    for k in ('obs', 'reward', 'done'):
                        assert np.array_equal(got[k], want[k]), (T, k)
                obs, rew, don = eng.read_outputs()
                assert np.array_equal(obs, st.pos) and np.array_equal(don, st.done), T
        eng.upload_actions(acts[:40])
        eng.upload_actions(acts[:10])  # replaces the 40 rows: the stream now ends at row 10
        with pytest.raises(gua.GuError):
            eng.rollout(20, 'stream', True, trajectory=False)
        with pytest.raises(gua.GuError):
            eng.step_device(10)
        eng.step_device(9)


def test_pinned_io_paths():
    """GU_F_PINNED_IO step and pinned trajectory reads give the same bytes as the bounce-buffer paths."""
    meta, z = G.load_traj('c4_lava32')
    T, N = z['actions'].shape
    with fresh(meta) as a, fresh(meta) as b:
        a.reset()
        b.reset()
        for t in range(40):
            b.pinned_actions[:] = z['actions'][t]
            o2, r2, d2 = b.step_pinned(auto_reset=True)
            o1, r1, d1 = a.step(z['actions'][t], auto_reset=True)
            assert np.array_equal(o1, o2) and np.array_equal(r1, r2) and np.array_equal(d1, d2)
            assert np.array_equal(o2, z['obs'][t])
        for eng in (a, b):
            eng.reserve_trajectory(100)
            eng.rollout(100, 'uniform', True)
        x, y = a.read_trajectory(0, 100), b.read_trajectory(0, 100, pinned=True)
        for k in x:
            assert np.array_equal(x[k], y[k]) and np.array_equal(x[k], z[k][40:140])


@pytest.mark.parametrize('N', [1, 3, 64, 65, 1000, 8192, 8193])
def test_one_wave_step_completion_word(N):
    """gu_step finishes by publishing a sequence number in page-locked memory that the host spins on (the last block to
    arrive publishes it; a real stream sync every 1024 steps): one lane, part of a wave, a full wave, a partial block,
    several blocks.  2 500 steps against the oracle, interleaved with calls that use the stream normally, through
    the page-locked and the pageable entry."""
    meta = dict(W=9, H=7, walls=[10, 11, 12, 30, 31], lava=[40], goals=[62, 5], starts=[0, 8, 36])
    grid = C.Grid.from_lists(**meta)
    rs = np.random.RandomState(N)
    acts = rs.randint(0, 4, (2500, N)).astype(np.int32)
    st = C.State(N, 7)
    C.reset(grid, 3, st)
    want = C.rollout(grid, 3, st, 2500, True, actions=acts)
    with Engine(N, GridSpec(9, 7, meta['starts'], meta['goals'], meta['lava'], meta['walls']), env_id0=7, seed=3) as eng:
        eng.reset()
        for t in range(2500):
            eng.pinned_actions[:] = acts[t]
            o, r, d = eng.step_pinned(auto_reset=True)
            assert np.array_equal(o, want['obs'][t]) and np.array_equal(r, want['reward'][t]) and np.array_equal(d, want['done'][t]), t
            if t % 700 == 699:  # other traffic on the same stream in between
                assert np.array_equal(eng.get_state()['pos'], want['obs'][t])
                eng.done_indices()
        want2 = C.rollout(grid, 3, st, 50, True, actions=acts[:50])
        for t in range(50):  # pageable buffers: the engine's staging block stands in
            o, r, d = eng.step(acts[t], auto_reset=True)
            assert np.array_equal(o, want2['obs'][t]) and np.array_equal(d, want2['done'][t]), t


def test_long_run_counters_and_graph_reseed():
    """20 000 steps per env in uneven chunks (RNG word counters far beyond the golden horizons), then the
    hipGraph step path before and after a reseed on a multi-start level (the captured launches carry the seed)."""
    meta, _ = G.load_traj('multistart_test_env')
    grid = C.Grid.from_lists(**meta)
    N = 4096
    st = C.State(N)
    C.reset(grid, 1234567, st)
    with Engine(N, spec_of(meta), seed=1234567) as eng:
        eng.reset()
        total = 0
        for chunk in (4999, 1, 8000, 7000):
            want = C.rollout(grid, 1234567, st, chunk, True, trajectory=False, stats=True)
            eng.rollout(chunk, 'uniform', True, trajectory=False, stats=True)
            ret, eps = eng.read_stats()
            assert np.array_equal(ret, want['ret']) and np.array_equal(eps, want['episodes']), chunk
            total += chunk
        s = eng.get_state()
        assert np.array_equal(s['pos'], st.pos) and np.array_equal(s['episode'], st.episode) and np.all(s['tcount'] == total)
        acts = np.random.RandomState(3).randint(0, 4, (32, N)).astype(np.int32)
        eng.upload_actions(acts)
        for seed in (1234567, 42):
            eng.seed(seed)
            st = C.State(N)
            C.reset(grid, seed, st)
            eng.reset()
            for rep in range(2):  # second replay reuses the cached graph
                eng.step_graph(0, 32, auto_reset=True)
                C.rollout(grid, seed, st, 32, True, actions=acts, trajectory=False)
                s = eng.get_state()
                assert np.array_equal(s['pos'], st.pos) and np.array_equal(s['episode'], st.episode), (seed, rep)


def test_one_process_several_engines_equals_one_engine():
    """MultiDeviceVecGridUniverse with the device list [0, 0, 0, 0] (the GPU box has one device): four engines with
    env-index shards reproduce the single-engine batch; the random maze is drawn once and shared."""
    import random
    from griduniverse_amd.parallel import MultiDeviceVecGridUniverse
    random.seed(11)
    np.random.seed(11)
    multi = MultiDeviceVecGridUniverse(4096, [0, 0, 0, 0], seed=6, auto_reset=True, grid_shape=(16, 16), random_maze=True)
    single = gua.VecGridUniverse(4096, template=multi.shards[0].template, seed=6, auto_reset=True)
    assert np.array_equal(multi.reset(), single.reset())
    a, b = multi.rollout(200, stats=True), single.rollout(200, stats=True)
    for k in ('obs', 'reward', 'done', 'ret', 'episodes'):
        assert np.array_equal(a[k], b[k]), k
    acts = np.random.RandomState(0).randint(0, 4, 4096).astype(np.int32)
    x, y = multi.step(acts), single.step(acts)
    assert all(np.array_equal(x[i], y[i]) for i in range(3))
    assert np.array_equal(multi.view()[0], single.observations)
    multi.close()
    single.close()


def test_single_process_rccl_view():
    """gu_comm_init_all / gu_allgather_view_all with one handle (ncclCommInitAll over [device 0]); sharing a device
    between ranks is refused with a message, as RCCL requires one device per rank."""
    from griduniverse_amd.parallel import MultiDeviceVecGridUniverse
    one = MultiDeviceVecGridUniverse(2048, [0], seed=3, auto_reset=True, grid_shape=(8, 8))
    one.reset()
    one.rollout(50, trajectory=False)
    a, b = one.view(rccl=True), one.view()
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    one.step(np.ones(2048, np.int32))
    assert np.array_equal(one.view(rccl=True)[0], one.view()[0])
    one.close()
    two = MultiDeviceVecGridUniverse(2048, [0, 0], seed=3, grid_shape=(8, 8))
    with pytest.raises(gua.GuError) as ei:
        two.view(rccl=True)
    assert 'one device per rank' in str(ei.value)
    two.close()


def test_vec_env_zero_copy_step():
    meta, z = G.load_traj('c2_open8x8')
    env = gua.VecGridUniverse(64, grid_shape=(8, 8), seed=2, auto_reset=True)
    env.reset()
    for t in range(30):
        if t % 2:
            env.actions_buffer[:] = z['actions'][t]
            o, r, d, _ = env.step(None, zero_copy=True)
        else:
            o, r, d, _ = env.step(z['actions'][t], zero_copy=True)
        assert np.array_equal(o, z['obs'][t]) and np.array_equal(r, z['reward'][t]) and np.array_equal(d, z['done'][t].astype(bool))
    env.close()


def test_config4_full_batch_in_eight_shards():
    """BASELINE config 4 at full size: 262 144 envs on the 32x32 lava grid as 8 shards of 32 768 (the per-GPU
    shards of the 8-GPU run, executed one after the other on the one GPU of the box) equal ONE oracle run of the
    whole batch, compared through sha256 digests per shard and the stats of every env."""
    meta, _ = G.load_traj('c4_lava32')
    grid = C.Grid.from_lists(**meta)
    total, shards, T, seed = 262144, 8, 250, 4
    per = total // shards
    st = C.State(total)
    C.reset(grid, seed, st)
    want = C.rollout(grid, seed, st, T, True, stats=True)
    for g in range(shards):
        sl = slice(g * per, (g + 1) * per)
        with Engine(per, spec_of(meta), env_id0=g * per, seed=seed) as eng:
            eng.reset()
            eng.reserve_trajectory(T)
            eng.rollout(T, 'uniform', True, trajectory=True, stats=True)
            got = eng.read_trajectory(0, T, pinned=True)
            ret, eps = eng.read_stats()
            assert digest(got['obs'], got['reward'], got['done']) == digest(want['obs'][:, sl], want['reward'][:, sl], want['done'][:, sl]), g
            assert np.array_equal(ret, want['ret'][sl]) and np.array_equal(eps, want['episodes'][sl])


@pytest.mark.parametrize('W,H', [(40000, 1), (1, 40000), (33000, 2)])
def test_extreme_aspect_grids(W, H):
    """Grids whose width does not fit the int16 delta LUT / whose cell count does not fit LDS (L2 path, arithmetic deltas)."""
    S = W * H
    rs = np.random.RandomState(W + H)
    walls = [int(x) for x in rs.choice(S, 200, replace=False)]
    meta = dict(W=W, H=H, walls=walls, lava=[S // 3], goals=[S - 1, S // 2], starts=[0, S // 4, S - 2])
    grid = C.Grid.from_lists(**meta)
    st = C.State(500, 9)
    C.reset(grid, 77, st)
    want = C.rollout(grid, 77, st, 300, True)
    with Engine(500, GridSpec(W, H, meta['starts'], meta['goals'], meta['lava'], walls), env_id0=9, seed=77) as eng:
        eng.reset()
        eng.reserve_trajectory(300)
        eng.rollout(300, 'uniform', True)
        got = eng.read_trajectory(0, 300)
        s, a = rs.randint(0, S, 4000), rs.randint(0, 4, 4000)
        for care in (True, False):
            g_, w_ = eng.look_step_ahead(s, a, care), C.look_step_ahead(grid, s, a, care)
            assert all(np.array_equal(x, y) for x, y in zip(g_, w_))
    for k in got:
        assert np.array_equal(got[k], want[k]), k


@pytest.mark.parametrize('name', ['c3_maze32', 'c4_lava32', 'rect25x30_busy', 'multistart_test_env', 'maze101'])
def test_packed_trajectory_mode(name):
    """GU_F_PACKED: one uint32 per env-step (obs | reward << 16 | done << 24) carries exactly the golden stream."""
    meta, z = G.load_traj(name)
    T, N = z['actions'].shape
    with fresh(meta) as eng:
        eng.reset()
        eng.reserve_trajectory(T)
        eng.rollout(T, 'uniform', meta['auto_reset'], trajectory='packed', stats=True)
        got = eng.read_trajectory_packed(0, T)
        words = eng.read_trajectory_packed(0, T, unpack=False, pinned=True)
        for k in ('obs', 'reward', 'done'):
            assert np.array_equal(got[k], z[k]), (name, k)
        assert words.dtype == np.uint32 and np.array_equal(words & 0xFFFF, z['obs'])
        with pytest.raises(gua.GuError):
            eng.read_trajectory(0, T)  # the buffer holds packed rows now
        eng.upload_actions(z['actions'])
        eng.seed(meta['seed'])
        eng.reset()
        eng.rollout(T, 'stream', meta['auto_reset'], trajectory='packed')
        assert np.array_equal(eng.read_trajectory_packed(0, T)['obs'], z['obs'])
        eng.rollout(5, 'uniform', True, trajectory=True)
        assert eng.read_trajectory(0, 5)['obs'].shape == (5, N)
    with Engine(8, GridSpec(300, 300, [0], [5], [], [])) as eng:  # 90 000 cells do not fit 16 bits
        eng.reserve_trajectory(4)
        with pytest.raises(gua.GuError):
            eng.rollout(4, 'uniform', True, trajectory='packed')


def test_trajectory_planes_beyond_4_gib():
    """1 M envs x 1100 steps: each trajectory plane is 4.6 GB (13.8 GB in all), so row addresses cross the 32-bit
    byte-offset boundary; spot rows and the per-env stats are checked against the oracle for two env ranges."""
    meta, _ = G.load_traj('c4_lava32')
    grid = C.Grid.from_lists(**meta)
    N, T, seed = 1 << 20, 1100, 12
    with Engine(N, spec_of(meta), seed=seed) as eng:
        eng.reset()
        eng.reserve_trajectory(T)
        eng.rollout(T, 'uniform', True, trajectory=True, stats=True)
        rows = {t: eng.read_trajectory(t, 1) for t in (0, 1023, 1024, T - 1)}
        ret, eps = eng.read_stats()
    for lo in (0, N - 2048):
        st = C.State(2048, lo)
        C.reset(grid, seed, st)
        want = C.rollout(grid, seed, st, T, True, stats=True)
        sl = slice(lo, lo + 2048)
        for t, got in rows.items():
            for k in ('obs', 'reward', 'done'):
                assert np.array_equal(got[k][0, sl], want[k][t]), (lo, t, k)
        assert np.array_equal(ret[sl], want['ret']) and np.array_equal(eps[sl], want['episodes'])


def test_maximum_batch_of_one_engine():
    """2^25 envs, the cap of one engine (gu_create): 24 steps with the trajectory written (9.7 GB; row offsets reach the
    top of the 2^31-byte buffer-addressing window the cap is derived from), spot-checked against the oracle at both ends
    and in the middle of the batch; one env more is refused."""
    meta, _ = G.load_traj('c4_lava32')
    grid = C.Grid.from_lists(**meta)
    N, T, seed = 1 << 25, 24, 5
    with pytest.raises(gua.GuError):
        Engine(N + 1, spec_of(meta))
    with Engine(N, spec_of(meta), seed=seed) as eng:
        first = eng.reset()
        eng.reserve_trajectory(T)
        eng.rollout(T, 'uniform', True, trajectory=True, stats=True)
        rows = {t: eng.read_trajectory(t, 1) for t in (0, 7, 8, 15, 16, T - 1)}
        ret, eps = eng.read_stats()
        obs, rew, don = eng.step(np.full(N, 2, np.int32), auto_reset=True)
    for lo in (0, N // 2 - 1000, N - 2048):
        st = C.State(2048, lo)
        C.reset(grid, seed, st)
        sl = slice(lo, lo + 2048)
        assert np.array_equal(first[sl], st.pos)
        want = C.rollout(grid, seed, st, T, True, stats=True)
        for t, got in rows.items():
            for k in ('obs', 'reward', 'done'):
                assert np.array_equal(got[k][0, sl], want[k][t]), (lo, t, k)
        assert np.array_equal(ret[sl], want['ret']) and np.array_equal(eps[sl], want['episodes'])
        w2 = C.rollout(grid, seed, st, 1, True, actions=np.full((1, 2048), 2, np.int32))
        assert np.array_equal(obs[sl], w2['obs'][0]) and np.array_equal(rew[sl], w2['reward'][0]) and np.array_equal(don[sl], w2['done'][0])
