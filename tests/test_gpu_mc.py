"""Monte-Carlo evaluation on the device: stream-2 policy sampling in the fused rollout
and the csrc/gu_mc.hip reduction, bit-exact against goldens captured from the reference's own
monte_carlo_evaluation and against the oracle on larger random cases."""
import numpy as np
import pytest

import griduniverse_amd as gua
from griduniverse_amd.algorithms import monte_carlo as mc
from griduniverse_amd.engine import Engine
from griduniverse_amd.grid import GridSpec
from oracle import c_oracle as C
from oracle import mc as omc
from tests import _golden as G

pytestmark = pytest.mark.gpu


def spec_of(meta):
    return GridSpec(meta['W'], meta['H'], meta['starts'], meta['goals'], meta['lava'], meta['walls'], meta['reward'])


def env_of(meta):
    return gua.GridUniverseEnv(grid_shape=(meta['W'], meta['H']), initial_state=list(meta['starts']),
                               goal_states=list(meta['goals']), lava_states=list(meta['lava']), walls=list(meta['walls']))


@pytest.mark.parametrize('name', G.mc_names())
def test_golden_mc_evaluation(name):
    meta, z = G.load_mc(name)
    S, N, T = meta['W'] * meta['H'], meta['N'], meta['T']
    with Engine(N, spec_of(meta), seed=meta['seed']) as eng:
        eng.vi_set(np.zeros(S), z['policy'])
        first = eng.reset()
        assert np.array_equal(first, z['first_state'])
        eng.reserve_trajectory(T)
        eng.rollout(T, 'sample', auto_reset=False, trajectory=True)
        traj = eng.read_trajectory(0, T)
        for k in ('obs', 'reward', 'done'):  # the sampled episodes themselves
            assert np.array_equal(traj[k], z[k]), (name, k)
        for run in meta['runs']:
            pw, keep = mc.discount_table(run['discount_factor'], run['threshold'], T)
            v, visits = eng.mc_evaluate(T, first, pw, keep, run['every_visit'], run['incremental_mean'],
                                        run['stationary_env'], run['alpha'])
            assert v.tobytes() == z[run['key']].tobytes(), (name, run)


@pytest.mark.parametrize('name', G.mc_names())
def test_reference_named_driver(name):
    """monte_carlo_evaluation(policy, env, ...) called like examples/griduniverse_alg_examples.py:107."""
    meta, z = G.load_mc(name)
    env = env_of(meta)
    for run in meta['runs'][:4]:
        v = mc.monte_carlo_evaluation(z['policy'], env, every_visit=run['every_visit'], incremental_mean=run['incremental_mean'],
                                      stationary_env=run['stationary_env'], discount_factor=run['discount_factor'],
                                      threshold=run['threshold'], alpha=run['alpha'], num_episodes=meta['N'],
                                      max_steps_per_episode=meta['T'], seed=meta['seed'])
        assert v.tobytes() == z[run['key']].tobytes()
    env.close()


def test_larger_batch_vs_oracle_and_chunking():
    """600 episodes x 300 steps on a 32x32 lava grid (several scratch chunks would need S*N*16 B > 256 MiB;
    here one chunk) and 3000 episodes on a 64-state grid, against the Python restatement."""
    rs = np.random.RandomState(4)
    for (W, H, lava, N, T) in ((32, 32, [16 + 32 * r for r in range(24)], 600, 300), (8, 8, [9, 21], 3000, 120)):
        S = W * H
        spec = GridSpec(W, H, [0, W + 1], [S - 1], lava, [W * 2 + 3])
        pi = rs.dirichlet(np.ones(4), S)
        with Engine(N, spec, seed=99) as eng:
            eng.vi_set(np.zeros(S), pi)
            first = eng.reset()
            eng.reserve_trajectory(T)
            eng.rollout(T, 'sample', auto_reset=False, trajectory=True)
            traj = eng.read_trajectory(0, T)
            grid = C.Grid(W, H, spec.wall, spec.lava, spec.goal, spec.reward, spec.starts)
            st = C.State(N)
            C.reset(grid, 99, st)
            want_traj = C.rollout(grid, 99, st, T, auto_reset=False, pi=pi)
            for k in ('obs', 'reward', 'done'):
                assert np.array_equal(traj[k], want_traj[k])
            sub = slice(0, 64)  # the O(L^2) Python restatement on a prefix of the episodes ...
            for ev, im, stn in ((False, True, True), (True, False, True), (True, True, False)):
                pw, keep = mc.discount_table(0.97, 1e-3, T)
                v, visits = eng.mc_evaluate(T, first, pw, keep, ev, im, stn, 0.01)
                assert np.isfinite(v).all() and visits.sum() > 0
            # ... checked exactly on an engine holding only that prefix
        with Engine(64, spec, seed=99) as eng:
            eng.vi_set(np.zeros(S), pi)
            first = eng.reset()
            eng.reserve_trajectory(T)
            eng.rollout(T, 'sample', auto_reset=False, trajectory=True)
            for ev, im, stn in ((False, True, True), (True, False, True), (True, True, False)):
                pw, keep = mc.discount_table(0.97, 1e-3, T)
                v, visits = eng.mc_evaluate(T, first, pw, keep, ev, im, stn, 0.01)
                eps = omc.episodes_from_trajectory(first, want_traj['obs'][:, sub], want_traj['reward'][:, sub], want_traj['done'][:, sub])
                v_want, vis_want = omc.monte_carlo_evaluation(S, eps, ev, im, stn, 0.97, 1e-3, 0.01)
                assert v.tobytes() == v_want.tobytes() and visits.tobytes() == vis_want.tobytes(), (W, ev, im, stn)


def test_sample_policy_with_auto_reset_and_stats():
    meta, z = G.load_mc('rect6x5_multistart')
    S, N = meta['W'] * meta['H'], 512
    grid = C.Grid.from_lists(**meta)
    st = C.State(N, 40)
    C.reset(grid, 5, st)
    want = C.rollout(grid, 5, st, 200, auto_reset=True, pi=z['policy'], stats=True)
    with Engine(N, spec_of(meta), env_id0=40, seed=5) as eng:
        eng.vi_set(np.zeros(S), z['policy'])
        eng.reset()
        eng.reserve_trajectory(200)
        eng.rollout(200, 'sample', auto_reset=True, trajectory=True, stats=True)
        got = eng.read_trajectory(0, 200)
        ret, eps = eng.read_stats()
    for k in ('obs', 'reward', 'done'):
        assert np.array_equal(got[k], want[k])
    assert np.array_equal(ret, want['ret']) and np.array_equal(eps, want['episodes'])


def test_chunk_boundaries_do_not_change_the_result(gu_option):
    """The evaluation walks the episodes in chunks (scratch budget); with a 1 MiB budget 3000 episodes take dozens of
    chunks and must give the same bytes as one chunk."""
    rs = np.random.RandomState(2)
    S, N, T = 64, 3000, 100
    spec = GridSpec(8, 8, [0, 9], [63], [20, 43], [10, 11, 12])
    pi = rs.dirichlet(np.ones(4), S)
    results = []
    for mb in (256, 1):
        gu_option('mc_scratch_mb', mb)
        with Engine(N, spec, seed=5) as eng:
            eng.vi_set(np.zeros(S), pi)
            first = eng.reset()
            eng.reserve_trajectory(T)
            eng.rollout(T, 'sample', auto_reset=False, trajectory=True)
            pw, keep = mc.discount_table(0.95, 1e-3, T)
            results.append([eng.mc_evaluate(T, first, pw, keep, ev, im, stn, 0.02) for ev, im, stn in
                            ((False, True, True), (True, True, False), (True, False, True))])
    for a, b in zip(*results):
        assert a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes()


@pytest.mark.parametrize('switch', ['mc_lane_returns', 'mc_global_walk'])
def test_tiled_and_per_lane_return_kernels_agree(gu_option, switch):
    """The LDS-tiled return kernel (zero-padded columns, masked discount table) against the per-lane kernel on
    global memory, and the LDS history walk against the global-memory walk, on full-length episodes
    (2048 x 1000 steps, mean length in the hundreds), every mode."""
    rs = np.random.RandomState(8)
    N, T = 2048, 1000
    random_state = np.random.get_state()
    import random
    random.seed(1)
    np.random.seed(1)
    env = gua.GridUniverseEnv(grid_shape=(8, 8), random_maze=True)
    np.random.set_state(random_state)
    S = env.world.size
    pi = rs.dirichlet(np.ones(4), S)
    modes = ((False, True, True), (True, True, True), (True, False, True), (False, True, False))
    results = []
    for lane in (None, 1):
        gu_option(switch, lane)
        with Engine(N, GridSpec.from_env(env), seed=21) as eng:
            eng.vi_set(np.zeros(S), pi)
            first = eng.reset()
            eng.reserve_trajectory(T)
            eng.rollout(T, 'sample', auto_reset=False, trajectory=True)
            out = []
            for gamma, thr in ((0.99, 1e-4), (1.0, 0.5), (0.5, 1e-3)):
                pw, keep = mc.discount_table(gamma, thr, T)
                out += [eng.mc_evaluate(T, first, pw, keep, ev, im, stn, 0.003) for ev, im, stn in modes]
            results.append(out)
    for a, b in zip(*results):
        assert a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes()
    env.close()


def test_long_episodes_fall_back_to_the_per_lane_kernel():
    """T = 3000 rows do not fit the LDS tile (8 columns + table > 160 KB): the per-lane kernel runs; checked against
    the Python restatement on 8 short episodes."""
    rs = np.random.RandomState(3)
    W, H, N, T = 4, 4, 8, 3000
    S = W * H
    spec = GridSpec(W, H, [0], [S - 1], [], [5, 10])
    pi = rs.dirichlet(np.ones(4), S)
    with Engine(N, spec, seed=13) as eng:
        eng.vi_set(np.zeros(S), pi)
        first = eng.reset()
        eng.reserve_trajectory(T)
        eng.rollout(T, 'sample', auto_reset=False, trajectory=True)
        traj = eng.read_trajectory(0, T)
        for ev, im, stn in ((False, True, True), (True, True, False)):
            pw, keep = mc.discount_table(0.999, 1e-2, T)
            v, visits = eng.mc_evaluate(T, first, pw, keep, ev, im, stn, 0.01)
            eps = omc.episodes_from_trajectory(first, traj['obs'], traj['reward'], traj['done'])
            v_want, vis_want = omc.monte_carlo_evaluation(S, eps, ev, im, stn, 0.999, 1e-2, 0.01)
            assert v.tobytes() == v_want.tobytes() and visits.tobytes() == vis_want.tobytes()


@pytest.mark.parametrize('W,H', [(8, 8), (64, 64)])
def test_sample_policy_threshold_edge_cases(W, H):
    """The device samples with integer thresholds ceil(c * 2^32); the oracle compares u = word / 2^32 with the float64
    prefix sums.  Rows that put a prefix sum exactly on 0, on 1, above 1, below 0, on a multiple of 2^-32, next to one,
    or on NaN must give the same actions.  64x64: the threshold table does not fit LDS and is read from L2."""
    rs = np.random.RandomState(11)
    S, N, T = W * H, 512, 64
    pi = rs.dirichlet(np.ones(4), S)
    special = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1], [0.5, 0.5, 0, 0], [0, 0, 0, 0],
                        [2, 0, 0, 0], [-1, 1, 0.5, 0.5], [0.25, 0.25, 0.25, 0.25], [2.0 ** -32, 0.5, 0.25, 0.25],
                        [np.nextafter(0.5, 1), 0.25, 0.125, 0.125], [np.nextafter(0.5, 0), 0.25, 0.125, 0.125],
                        [3 * 2.0 ** -32, 2.0 ** -33, 0.5, 0.4], [np.nan, 0.5, 0.25, 0.25], [0.3, np.nan, 0.3, 0.4],
                        [1e-300, 1e-300, 1e-300, 1], [1 - 2.0 ** -32, 2.0 ** -33, 2.0 ** -33, 0]], dtype=np.float64)
    pi[rs.permutation(S)[:len(special) * 3]] = np.tile(special, (3, 1))
    pi[0] = special[4]
    spec = GridSpec(W, H, [0, 1, W, W + 1], [S - 1], [], [])
    grid = C.Grid(W, H, spec.wall, spec.lava, spec.goal, spec.reward, spec.starts)
    st = C.State(N)
    C.reset(grid, 77, st)
    want = C.rollout(grid, 77, st, T, auto_reset=True, pi=pi)
    with Engine(N, spec, seed=77) as eng:
        eng.vi_set(np.zeros(S), pi)
        eng.reset()
        eng.reserve_trajectory(T)
        eng.rollout(T, 'sample', auto_reset=True, trajectory=True)
        got = eng.read_trajectory(0, T)
    for k in ('obs', 'reward', 'done'):
        assert np.array_equal(got[k], want[k]), k


def test_random_grids_mc_property(gu_option):
    """Random small grids with several start cells, random stochastic policies, every update mode, random discount /
    threshold / step cap / batch size / scratch budget: sampled episodes against the C restatement and the
    evaluation against the Python restatement of monte_carlo_evaluation.  GU_FUZZ_TRIALS=N for a longer soak."""
    import os
    trials = int(os.environ.get('GU_FUZZ_TRIALS', '24'))
    rs = np.random.RandomState(int(os.environ.get('GU_FUZZ_SEED', '4242')))
    for trial in range(trials):
        W, H = int(rs.randint(2, 12)), int(rs.randint(2, 12))
        S = W * H
        pick = lambda k: [int(x) for x in rs.choice(S, size=min(S, int(k)), replace=False)]  # noqa: E731
        walls, lava, goals, starts = pick(rs.randint(0, S // 4 + 1)), pick(rs.randint(0, 3)), pick(rs.randint(1, 3)), pick(rs.randint(1, 4))
        spec = GridSpec(W, H, starts, goals, lava, walls)
        grid = C.Grid(W, H, spec.wall, spec.lava, spec.goal, spec.reward, spec.starts)
        pi = rs.dirichlet(np.ones(4) * rs.choice([0.3, 1.0, 5.0]), S)
        N, T, seed = int(rs.randint(1, 200)), int(rs.randint(1, 90)), int(rs.randint(0, 2 ** 40))
        gamma, thr = float(rs.choice([1.0, 0.99, 0.9, 0.5])), float(rs.choice([1e-4, 1e-2, 0.5]))
        ev, im, stn = bool(rs.randint(2)), bool(rs.randint(2)), bool(rs.randint(2))
        gu_option('mc_scratch_mb', int(rs.choice([1, 2048])))
        try:
            st = C.State(N)
            C.reset(grid, seed, st)
            first_want = st.pos.copy()
            want = C.rollout(grid, seed, st, T, auto_reset=False, pi=pi)
            with Engine(N, spec, seed=seed) as eng:
                eng.vi_set(np.zeros(S), pi)
                first = eng.reset()
                assert np.array_equal(first, first_want)
                eng.reserve_trajectory(T)
                eng.rollout(T, 'sample', auto_reset=False, trajectory=True)
                got = eng.read_trajectory(0, T)
                for k in ('obs', 'reward', 'done'):
                    assert np.array_equal(got[k], want[k]), (trial, k)
                pw, keep = mc.discount_table(gamma, thr, T)
                v, visits = eng.mc_evaluate(T, first, pw, keep, ev, im, stn, 0.05)
            eps = omc.episodes_from_trajectory(first, want['obs'], want['reward'], want['done'])
            v_want, vis_want = omc.monte_carlo_evaluation(S, eps, ev, im, stn, gamma, thr, 0.05)
            assert v.tobytes() == v_want.tobytes() and visits.tobytes() == vis_want.tobytes(), (trial, W, H, N, T, ev, im, stn, gamma, thr)
        finally:
            gu_option('mc_scratch_mb', None)


def test_random_grids_table_policies_property():
    """Greedy (first argmax) and sampled rollouts on random grids -- one or several start cells, with and without
    auto-reset, any number of steps (the 8-step unrolled body and its tail), int32 / packed / no trajectory, stats,
    grids whose threshold table fits LDS, does not fit, and whose records do not fit either -- against the C
    restatement.  A one-hot policy sampled by the oracle IS the greedy policy.  GU_FUZZ_TRIALS=N for a longer soak."""
    import os
    trials = int(os.environ.get('GU_FUZZ_TRIALS', '40'))
    rs = np.random.RandomState(int(os.environ.get('GU_FUZZ_SEED', '99')))
    for trial in range(trials):
        big = trial % 8
        W, H = ((int(rs.randint(1, 40)), int(rs.randint(1, 30))) if big < 6 else (int(rs.randint(64, 90)), int(rs.randint(64, 90)))
                if big == 6 else (int(rs.randint(182, 230)), int(rs.randint(182, 200))))
        S = W * H
        pick = lambda k: [int(x) for x in rs.choice(S, size=min(S, int(k)), replace=False)]  # noqa: E731
        walls, lava, goals = pick(rs.randint(0, S // 4 + 1)), pick(rs.randint(0, 4)), pick(rs.randint(1, 4))
        starts = pick(1 if trial % 2 else rs.randint(2, 5))
        spec = GridSpec(W, H, starts, goals, lava, walls)
        grid = C.Grid(W, H, spec.wall, spec.lava, spec.goal, spec.reward, spec.starts)
        greedy = bool(rs.randint(2))
        if greedy:
            pi = np.zeros((S, 4))
            pi[np.arange(S), rs.randint(0, 4, S)] = 1.0
        else:
            pi = rs.dirichlet(np.ones(4) * rs.choice([0.2, 1.0]), S)
        N, T, seed, id0 = int(rs.randint(1, 700)), int(rs.randint(1, 70)), int(rs.randint(0, 2 ** 50)), int(rs.randint(0, 2 ** 30))
        auto = bool(rs.randint(2))
        kind = [True, 'packed', False][rs.randint(3)] if S <= 65536 else [True, False][rs.randint(2)]
        st = C.State(N, id0)
        C.reset(grid, seed, st)
        want = C.rollout(grid, seed, st, T, auto_reset=auto, pi=pi, stats=True)
        with Engine(N, spec, env_id0=id0, seed=seed) as eng:
            eng.vi_set(np.zeros(S), pi)
            eng.reset()
            eng.reserve_trajectory(T)
            if kind is True:  # int32 rows: forced store pacing / kernel choice never change a byte
                eng.set_option('rollout_pace', (None, 0, int(rs.randint(1, 600)))[trial % 3])
                eng.set_option('rollout_rows', (None, 0, 1, 2)[(trial // 12) % 4])
            eng.rollout(T, 'greedy' if greedy else 'sample', auto_reset=auto, trajectory=kind, stats=True)
            if kind:
                got = eng.read_trajectory_packed(0, T) if kind == 'packed' else eng.read_trajectory(0, T)
                for k in ('obs', 'reward', 'done'):
                    assert np.array_equal(got[k], want[k]), (trial, W, H, N, T, greedy, auto, kind, k)
            ret, eps = eng.read_stats()
            state = eng.get_state()
        assert np.array_equal(ret, want['ret']) and np.array_equal(eps, want['episodes']), (trial, W, H, greedy, auto)
        assert np.array_equal(state['pos'], st.pos) and np.array_equal(state['episode'], st.episode)


def test_policy_rows_that_are_no_distribution_raise_like_np_random_choice():
    """The reference's run_episode draws with np.random.choice(4, p=policy[obs]), which raises ValueError for a row that does
    not sum to 1 or is negative -- when an episode visits that state, not otherwise."""
    from griduniverse_amd.algorithms import monte_carlo as mc
    env = gua.GridUniverseEnv(grid_shape=(4, 4), walls=[5])
    S = env.world.size
    pi = np.ones((S, 4)) / 4
    pi[5] = [0.5, 0.5, 0.5, 0.5]        # the wall cell is never entered: the reference never looks at this row
    assert np.isfinite(mc.monte_carlo_evaluation(pi, env, num_episodes=32, max_steps_per_episode=60)).all()
    pi[0] = [0.3, 0.3, 0.3, 0.3]        # the start cell: every episode draws from it
    with pytest.raises(ValueError, match='do not sum to 1'):
        mc.monte_carlo_evaluation(pi, env, num_episodes=32, max_steps_per_episode=60)
    pi[0] = [1.5, -0.5, 0.0, 0.0]
    with pytest.raises(ValueError, match='not non-negative'):
        mc.monte_carlo_evaluation(pi, env, num_episodes=32, max_steps_per_episode=60)
    env.close()


def mcnp_names():
    import glob
    import os
    return sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(G.GOLDEN, 'mcnp_*.npz')))


@pytest.mark.parametrize('name', mcnp_names())
def test_reference_rng_mode_returns_the_reference_value_function(name):
    """rng='numpy': `random.seed(k); np.random.seed(k); monte_carlo_evaluation(policy, env, ...)` of the REAL reference (its own
    run_episode drawing np.random.choice from the global stream, start cells from the stdlib's) against the same call on the
    engine: the returned array byte for byte, for 2 x 6 flag combinations per grid, and both global streams left exactly where
    the reference leaves them (the next uniform of each)."""
    import random
    meta, z = G.load_npz('mcnp', name)
    env = gua.GridUniverseEnv(grid_shape=(meta['W'], meta['H']), initial_state=meta['starts'], goal_states=meta['goals'],
                              lava_states=meta['lava'], walls=meta['walls'])
    assert mcnp_names() and len(meta['runs']) == 12
    np_state, py_state = np.random.get_state(), random.getstate()
    try:
        for run in meta['runs']:
            random.seed(meta['seed'])
            np.random.seed(meta['seed'])
            v = mc.monte_carlo_evaluation(z['policy'], env, every_visit=run['every_visit'], incremental_mean=run['incremental_mean'],
                                          stationary_env=run['stationary_env'], discount_factor=run['discount_factor'],
                                          threshold=run['threshold'], alpha=run['alpha'], num_episodes=meta['num_episodes'], rng='numpy')
            assert v.tobytes() == z[run['key']].tobytes(), (name, run)
            assert float(np.random.random_sample()) == run['next_numpy_uniform'] and random.random() == run['next_stdlib_uniform']
        # the instance is left where the last episode ended, like the reference's (run_episode steps the env itself)
        random.seed(meta['seed'])
        np.random.seed(meta['seed'])
        want_states = None
        for _ in range(meta['num_episodes']):
            want_states, _, want_done = mc.run_episode(z['policy'], env)
        want = (env.current_state, env.previous_state, env.done, list(env.last_n_states))
        random.seed(meta['seed'])
        np.random.seed(meta['seed'])
        mc.monte_carlo_evaluation(z['policy'], env, num_episodes=meta['num_episodes'], rng='numpy')
        assert (env.current_state, env.previous_state, env.done, list(env.last_n_states)) == want and want[0] == want_states[-1]
    finally:
        np.random.set_state(np_state)
        random.setstate(py_state)
        env.close()


def test_reference_rng_mode_raises_like_np_random_choice():
    """A policy row that is no distribution raises ValueError only when an episode draws from it (np.random.choice validates p
    before drawing), with numpy's message."""
    env = gua.GridUniverseEnv(grid_shape=(4, 1), goal_states=[3])
    pi = np.ones((4, 4)) / 4
    pi[3] = [2, 2, 2, 2]          # terminal state: never drawn from
    assert mc.monte_carlo_evaluation(pi, env, num_episodes=3, rng='numpy').shape == (4,)
    pi[1] = [0.5, 0.6, 0.0, 0.0]
    with pytest.raises(ValueError, match='do not sum to 1'):
        mc.monte_carlo_evaluation(pi, env, num_episodes=20, rng='numpy')
    pi[1] = [1.5, -0.5, 0.0, 0.0]
    with pytest.raises(ValueError, match='not non-negative'):
        mc.monte_carlo_evaluation(pi, env, num_episodes=20, rng='numpy')
    with pytest.raises(ValueError):
        mc.monte_carlo_evaluation(np.ones((4, 4)) / 4, env, num_episodes=2, rng='mt')
    # a table with fewer rows than the grid has cells, or rows that are not 4 wide: never handed to the device walk (which copies
    # S * 32 bytes of it); the host walk raises where the reference's policy[obs] / np.random.choice(4, p=row) would
    with pytest.raises(IndexError, match='out of bounds'):
        mc.monte_carlo_evaluation(np.ones((2, 4)) / 4, env, num_episodes=50, rng='numpy')
    with pytest.raises(ValueError, match='same size'):
        mc.monte_carlo_evaluation(np.ones((4, 3)) / 3, env, num_episodes=2, rng='numpy')
    eng = gua.Engine(8, gua.GridSpec.from_env(env))
    try:
        with pytest.raises(ValueError, match='cdf must have shape'):
            eng.mc_walk_lengths(np.zeros(64), 8, [0], 8, np.ones((2, 4)))
        with pytest.raises(ValueError, match='cdf must have shape'):
            eng.mc_walk_episodes(np.zeros(64), np.ones((4, 3)), np.zeros(8, np.int64), np.zeros(8, np.int32), 8, 8)
    finally:
        eng.close()
    env.close()
