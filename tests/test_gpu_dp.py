"""Config 5 parity: the float64 DP kernels (csrc/gu_vi.hip) against golden tables captured from the
reference's core/algorithms (bit-exact: compared as raw bytes), and the fused sweep+step launch
against the oracle."""
import warnings

import numpy as np
import pytest

import griduniverse_amd as gua
from griduniverse_amd.algorithms import dynamic_programming as dp
from griduniverse_amd.algorithms import utils
from griduniverse_amd.engine import Engine
from griduniverse_amd.grid import GridSpec
from oracle import c_oracle as C
from tests import _golden as G

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=['one_launch', 'per_xcd_everywhere', 'no_per_xcd_form', 'launch_per_round'])
def vi_path(request, gu_option):
    """Every test runs four times.  'one_launch': the default dispatch -- every grid that fits takes the per-XCD launch
    (csrc/gu_vi_xcd.hip: one XCD's workgroups, the whole iteration in one launch); fused sweep + step calls of ONE round take the
    single fused launch.  'per_xcd_everywhere' (option vi_path = 6): the per-XCD launch for those too.  'no_per_xcd_form'
    (vi_path = 4): the single-workgroup kernel up to 4096 states, the chip-wide workgroup cluster beyond.  'launch_per_round'
    (vi_path = 2): one launch per round on every grid size."""
    gu_option('vi_path', {'launch_per_round': 2, 'no_per_xcd_form': 4, 'per_xcd_everywhere': 6}.get(request.param))
    return request.param


def base_path(vi_path):
    """The option value a test that switches vi_path itself goes back to for 'the default dispatch' of this run."""
    return 6 if vi_path == 'per_xcd_everywhere' else None


def spec_of(meta):
    return GridSpec(meta['W'], meta['H'], meta['starts'], meta['goals'], meta['lava'], meta['walls'], meta['reward'])


def env_of(meta):
    return gua.GridUniverseEnv(grid_shape=(meta['W'], meta['H']), initial_state=list(meta['starts']),
                               goal_states=list(meta['goals']), lava_states=list(meta['lava']), walls=list(meta['walls']))


@pytest.mark.parametrize('name', G.dp_names())
def test_engine_sweeps_bit_exact(name):
    meta, z = G.load_dp(name)
    S, gamma = meta['W'] * meta['H'], meta['gamma']
    with Engine(8, spec_of(meta)) as eng:
        eng.vi_set(np.zeros(S), np.ones((S, 4)) / 4)
        done = 0
        for k in (1, 2, 10):  # repeated V1 under the uniform policy
            eng.vi_sweep(gamma, k - done, greedy_update=False)
            done = k
            assert eng.vi_get()[0].tobytes() == z['eval_v_%d' % k].tobytes(), (name, k)
        eng.vi_greedy(gamma)
        v, pi = eng.vi_get()
        assert pi.tobytes() == z['greedy_pi_after_10'].tobytes() and v.tobytes() == z['eval_v_10'].tobytes()
        # value-iteration rounds, one at a time and then all in one call
        eng.vi_set(np.zeros(S), np.ones((S, 4)) / 4)
        for k in range(1, meta['iters'] + 1):
            delta = eng.vi_sweep(gamma, 1, greedy_update=True)[0]
            v, pi = eng.vi_get()
            if 'vi_v_%d' % k in z:  # (grids beyond 4096 states keep rounds 1, 2 and the last one)
                assert v.tobytes() == z['vi_v_%d' % k].tobytes() and pi.tobytes() == z['vi_pi_%d' % k].tobytes(), (name, k)
            assert delta == meta['deltas'][k - 1]
        eng.vi_set(np.zeros(S), np.ones((S, 4)) / 4)
        deltas = eng.vi_sweep(gamma, meta['iters'], greedy_update=True)
        v, pi = eng.vi_get()
        assert deltas.tolist() == meta['deltas']
        assert v.tobytes() == z['vi_v_%d' % meta['iters']].tobytes() and pi.tobytes() == z['vi_pi_%d' % meta['iters']].tobytes()


@pytest.mark.parametrize('name', G.dp_names())
def test_reference_named_drivers(name):
    """utils.single_step_policy_evaluation / greedy_policy_from_value_function / dp.value_iteration /
    dp.policy_iteration called the way examples/griduniverse_alg_examples.py:29-63 calls them."""
    meta, z = G.load_dp(name)
    env = env_of(meta)
    S, gamma = env.world.size, meta['gamma']
    policy0 = np.ones([env.world.size, len(env.action_state_to_next_state)]) / len(env.action_state_to_next_state)
    v = np.zeros(S)
    for k in range(10):
        v = utils.single_step_policy_evaluation(policy0, env, discount_factor=gamma, value_function=v)
    assert v.tobytes() == z['eval_v_10'].tobytes()
    pi_in = policy0.copy()
    pi_out = utils.greedy_policy_from_value_function(pi_in, env, v, discount_factor=gamma)
    assert pi_out is pi_in and pi_in.tobytes() == z['greedy_pi_after_10'].tobytes()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        v3, pi3 = dp.value_iteration(policy0.copy(), env, np.zeros(S), threshold=meta['driver_threshold'],
                                     max_steps=meta['iters'], discount_factor=gamma)
    assert v3.tobytes() == z['vi_driver_v'].tobytes() and pi3.tobytes() == z['vi_driver_pi'].tobytes()
    assert (len(w) > 0) == meta['driver_warned']
    if 'pi_driver_v' in z:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            v4, pi4 = dp.policy_iteration(policy0.copy(), env, np.zeros(S), threshold=1e-3,
                                          max_steps=meta['pi_driver_max_steps'], discount_factor=gamma)
        assert v4.tobytes() == z['pi_driver_v'].tobytes() and pi4.tobytes() == z['pi_driver_pi'].tobytes()
        assert (len(w) > 0) == meta['pi_driver_warned']
    env.close()


def test_random_policies_and_values_bit_exact():
    """Non-uniform pi and non-trivial v exercise every rounding step of V1 / V2 (vs the C oracle,
    which is itself pinned to the reference by tests/test_oracle_c.py)."""
    rs = np.random.RandomState(5)
    for name in ('maze11_s3_g09', 'rect6x5_g1', 'maze32_s1_g099'):
        meta, _ = G.load_dp(name)
        grid, S = C.Grid.from_lists(**meta), meta['W'] * meta['H']
        with Engine(4, spec_of(meta)) as eng:
            for gamma in (1.0, 0.9, 0.37):
                pi = rs.dirichlet(np.ones(4), S)
                v = rs.standard_normal(S) * 50
                v[::7] = np.round(v[::7])  # force exact ties in q
                eng.vi_set(v, pi)
                d = eng.vi_sweep(gamma, 1, greedy_update=True)[0]
                v_want, pi_want, d_want = C.value_iteration_step(grid, gamma, pi, v)
                v_got, pi_got = eng.vi_get()
                assert v_got.tobytes() == v_want.tobytes() and pi_got.tobytes() == pi_want.tobytes() and d == d_want


@pytest.mark.parametrize('name,N', [('maze64_s5', 65536), ('maze64_s5_g097', 4096), ('rect6x5_g1', 100), ('lava4x4_g095', 3)])
def test_fused_sweep_step_launch(name, N):
    """Config 5: ONE launch = one V1+V2 round + one greedy env step on the updated policy."""
    meta, z = G.load_dp(name)
    grid, S, gamma = C.Grid.from_lists(**meta), meta['W'] * meta['H'], meta['gamma']
    st = C.State(N)
    C.reset(grid, 3, st)
    v, pi = np.zeros(S), np.ones((S, 4)) / 4
    with Engine(N, spec_of(meta), seed=3) as eng:
        assert np.array_equal(eng.reset(), st.pos)
        eng.vi_set(v, pi)
        for k in range(1, meta['iters'] + 1):
            delta = eng.vi_sweep_step(gamma, auto_reset=True)
            v, pi, d_want = C.value_iteration_step(grid, gamma, pi, v)
            assert 'vi_v_%d' % k not in z or v.tobytes() == z['vi_v_%d' % k].tobytes()  # the oracle itself is on the reference's trace
            acts = np.argmax(pi, axis=1).astype(np.int32)     # examples/griduniverse_alg_examples.py:76
            # lazy auto-reset happens before the action is looked up, exactly as in the kernel
            pre = st.pos.copy()
            if st.done.any():
                m = st.done.astype(bool)
                C.reset(grid, 3, st, mask=m)
                pre = st.pos.copy()
            want = C.rollout(grid, 3, st, 1, False, actions=acts[pre][None, :])
            obs, rew, don = eng.read_outputs()
            v_got, pi_got = eng.vi_get()
            assert v_got.tobytes() == v.tobytes() and pi_got.tobytes() == pi.tobytes() and delta == d_want, (name, k)
            assert np.array_equal(obs, want['obs'][0]) and np.array_equal(rew, want['reward'][0]) and np.array_equal(don, want['done'][0])
        s = eng.get_state()
        assert np.array_equal(s['episode'], st.episode)


def test_greedy_rollout_follows_value_iteration_policy():
    """After VI converges on an 11x11 maze, agents acting greedily reach the goal (the demo of
    examples/griduniverse_alg_examples.py:66-85, batched), identical to the oracle step by step."""
    meta, z = G.load_dp('maze11_s3_g09')
    grid, S = C.Grid.from_lists(**meta), meta['W'] * meta['H']
    N, T = 300, 60
    with Engine(N, spec_of(meta), seed=1) as eng:
        eng.vi_set(np.zeros(S), np.ones((S, 4)) / 4)
        eng.vi_sweep(0.9, 200, greedy_update=True)
        pi = eng.vi_get()[1]
        acts = np.argmax(pi, axis=1).astype(np.int32)
        eng.reset()
        eng.reserve_trajectory(T)
        eng.rollout(T, 'greedy', auto_reset=False, trajectory=True)
        got = eng.read_trajectory(0, T)
    st = C.State(N)
    C.reset(grid, 1, st)
    for t in range(T):
        want = C.rollout(grid, 1, st, 1, False, actions=acts[st.pos][None, :])
        assert np.array_equal(got['obs'][t], want['obs'][0]) and np.array_equal(got['done'][t], want['done'][0])
    assert got['done'][-1].all(), 'greedy agents did not reach the goal'


def test_breadth_first_search_paths():
    """algorithms.maze_solving: the path it returns is executable on the env, ends in a terminal state, and is as
    short as the value-iteration optimum (every step costs -1, so -return + 10 == path length on goal paths)."""
    import random
    from griduniverse_amd.algorithms import maze_solving
    for k in (1, 2, 3):
        random.seed(k)
        np.random.seed(k)
        env = gua.GridUniverseEnv(grid_shape=(15, 15), random_maze=True)  # the size the reference script uses (:20)
        path = maze_solving.breadth_first_search(env)
        env.reset()
        for i, a in enumerate(path):
            obs, reward, done, _ = env.step(a)
            assert done == (i == len(path) - 1)
        assert done and obs in env.goal_states
        dist = {env.initial_state: 0}
        todo = [env.initial_state]
        graph = maze_solving.create_graph(env)
        while todo:  # independent BFS distance
            s = todo.pop(0)
            for c in graph[s]:
                if c not in dist:
                    dist[c] = dist[s] + 1
                    todo.append(c)
        assert len(path) == dist[env.goal_states[0]]
        env.close()
    env = gua.GridUniverseEnv(grid_shape=(5, 5), lava_states=[2], walls=[6, 7, 8])  # lava is terminal too (:140)
    assert maze_solving.breadth_first_search(env) == [1, 1]
    env = gua.GridUniverseEnv(grid_shape=(3, 3), walls=[5, 7])
    assert maze_solving.breadth_first_search(env) is None


@pytest.mark.parametrize('name', ['maze8_s1', 'maze11_s3_g09', 'lava4x4_g095', 'maze32_s1_g099', 'level101_g099', 'maze128_s7'])
def test_vi_run_device_side_stopping(name):
    """gu_vi_run queues max_steps rounds and stops on the device: same tables and round count as stepping one
    round at a time from the host with the reference's `delta < threshold` rule."""
    meta, z = G.load_dp(name)
    S, gamma = meta['W'] * meta['H'], meta['gamma']
    with Engine(4, spec_of(meta)) as eng:
        big = S > 4096  # (the host-stepped comparison loop of a 101x101 level is kept short)
        for threshold, max_steps in ((1e-3, 60 if big else 400), (0.5, 60 if big else 400), (1e-3, 7), (1e9, 5), (1e-3, 0)):
            eng.vi_set(np.zeros(S), np.ones((S, 4)) / 4)
            want_steps, want_deltas = 0, []
            for k in range(max_steps):
                d = eng.vi_sweep(gamma, 1, greedy_update=True)[0]
                want_steps += 1
                want_deltas.append(d)
                if d < threshold:
                    break
            v_want, pi_want = eng.vi_get()
            eng.vi_set(np.zeros(S), np.ones((S, 4)) / 4)
            steps, deltas = eng.vi_run(gamma, threshold, max_steps)
            v, pi = eng.vi_get()
            assert steps == want_steps and deltas.tolist() == want_deltas, (name, threshold, max_steps)
            assert v.tobytes() == v_want.tobytes() and pi.tobytes() == pi_want.tobytes()
            # the tables stay usable afterwards (buffer parity restored)
            eng.vi_sweep(gamma, 1, greedy_update=True)


def test_vi_run_on_a_grid_larger_than_the_resident_block_capacity():
    """533 000 states = 2 083 blocks of 256 > the ~2 048 that can be resident: blocks of the stopping round's own
    greedy kernel start after its thread 0 has recorded the stop, and must still do their work."""
    W = H = 730
    spec = GridSpec(W, H, [0], [W * H - 1], [W * H // 2], list(range(W + 5, W + 400)))
    S = W * H
    with Engine(2, spec) as eng:
        for threshold, max_steps in ((1.5, 6), (-1.0, 3)):
            eng.vi_set(np.zeros(S), np.ones((S, 4)) / 4)
            want = 0
            for k in range(max_steps):
                want += 1
                if eng.vi_sweep(1.0, 1, greedy_update=True)[0] < threshold:
                    break
            v_want, pi_want = eng.vi_get()
            eng.vi_set(np.zeros(S), np.ones((S, 4)) / 4)
            steps, deltas = eng.vi_run(1.0, threshold, max_steps)
            v, pi = eng.vi_get()
            assert steps == want and v.tobytes() == v_want.tobytes() and pi.tobytes() == pi_want.tobytes()


@pytest.mark.parametrize('name', ['maze8_s1', 'maze11_s3_g09', 'lava4x4_g095', 'maze32_s1_g099', 'level101_g099'])
def test_vi_eval_run_is_the_host_loop_of_policy_iteration(name):
    """gu_vi_eval_run = V1 sweeps on the fixed policy until delta < threshold: same sweep count, deltas and v as one
    sweep per call with the host applying the rule (dynamic_programming.py:40-42)."""
    meta, z = G.load_dp(name)
    S, gamma = meta['W'] * meta['H'], meta['gamma']
    rs = np.random.RandomState(1)
    pi0 = rs.dirichlet(np.ones(4), S)
    with Engine(4, spec_of(meta)) as eng:
        for threshold, max_steps in ((1e-3, 300), (0.5, 300), (1e-6, 9), (1e9, 5), (1e-3, 0), (1e-3, 70)):
            eng.vi_set(np.zeros(S), pi0)
            want_deltas = []
            for k in range(max_steps):
                want_deltas.append(eng.vi_sweep(gamma, 1, greedy_update=False)[0])
                if want_deltas[-1] < threshold:
                    break
            v_want, pi_want = eng.vi_get()
            eng.vi_set(np.zeros(S), pi0)
            steps, deltas = eng.vi_eval_run(gamma, threshold, max_steps)
            v, pi = eng.vi_get()
            assert steps == len(want_deltas) and deltas.tolist() == want_deltas, (name, threshold, max_steps)
            assert v.tobytes() == v_want.tobytes() and pi.tobytes() == pi0.tobytes() == pi_want.tobytes()


def test_one_workgroup_and_launch_per_round_agree_at_64x64(vi_path, gu_option):
    """config 5's grid through both paths: 300 rounds of value iteration, a policy-evaluation run and sweeps from a
    random policy and value table (ties, negative and positive values), compared as raw bytes."""
    import random
    state = np.random.get_state()
    random.seed(5)
    np.random.seed(5)
    env = gua.GridUniverseEnv(grid_shape=(64, 64), random_maze=True)
    np.random.set_state(state)
    S = env.world.size
    rs = np.random.RandomState(9)
    v0, pi0 = rs.randn(S) * 3, rs.dirichlet(np.ones(4), S)
    out = []
    for multi in (False, True):
        gu_option('vi_path', 2 if multi else base_path(vi_path))
        with Engine(4, GridSpec.from_env(env)) as eng:
            res = []
            eng.vi_set(v0, pi0)
            res.append(eng.vi_run(0.95, 1e-4, 300))
            res.append(eng.vi_get())
            eng.vi_set(v0, pi0)
            res.append(eng.vi_eval_run(0.9, 1e-3, 200))
            res.append(eng.vi_get())
            res.append((0, eng.vi_sweep(1.0, 37, greedy_update=True)))
            res.append(eng.vi_get())
            res.append((0, eng.vi_sweep(0.99, 5, greedy_update=False)))
            res.append(eng.vi_get())
            out.append(res)
    env.close()
    for a, b in zip(*out):
        assert a[0] == b[0] if isinstance(a[0], int) else a[0].tobytes() == b[0].tobytes()
        assert np.asarray(a[1]).tobytes() == np.asarray(b[1]).tobytes()


def test_random_grids_dp_property():
    """Random grids (walls / lava / goals anywhere, incl. overlaps), random gamma, random value and policy tables
    (zeros, ties, negative entries): sweeps, greedy updates, the device-side loops and their stopping rounds against
    the C restatement, bytes for bytes.  GU_FUZZ_TRIALS=N for a longer soak."""
    import os
    trials = int(os.environ.get('GU_FUZZ_TRIALS', '40'))
    rs = np.random.RandomState(int(os.environ.get('GU_FUZZ_SEED', '77')))
    for trial in range(trials):
        W, H = int(rs.randint(1, 80)), int(rs.randint(1, 60))
        S = W * H
        pick = lambda k: [int(x) for x in rs.choice(S, size=min(S, int(k)), replace=False)]  # noqa: E731
        walls, lava, goals = pick(rs.randint(0, S // 3 + 1)), pick(rs.randint(0, 6)), pick(rs.randint(1, 5))
        meta = dict(W=W, H=H, walls=walls, lava=lava, goals=goals, starts=[0])
        grid = C.Grid.from_lists(**meta)
        spec = GridSpec(W, H, [0], goals, lava, walls)
        gamma = float(rs.choice([1.0, 0.9, 0.5, 0.99, rs.uniform(0, 1.2)]))
        v = rs.choice([0.0, 1.0, -1.0, 2.5]) * rs.randn(S) if trial % 3 else np.round(rs.randn(S), 1)  # ties on purpose
        pi = rs.dirichlet(np.ones(4), S) if trial % 2 else np.ones((S, 4)) / 4
        rounds, threshold = int(rs.randint(1, 12)), float(rs.choice([1e-3, 0.3, 5.0, -1.0]))
        # the reference loop, one round at a time
        v_w, pi_w, deltas_w = v.copy(), pi.copy(), []
        for _ in range(rounds):
            v_w, pi_w, d = C.value_iteration_step(grid, gamma, pi_w, v_w)
            deltas_w.append(d)
            if d < threshold:
                break
        e_v, e_deltas = v.copy(), []
        for _ in range(rounds):
            new = C.policy_evaluation_sweep(grid, gamma, pi, e_v)
            e_deltas.append(float(np.max(e_v - new)))
            e_v = new
            if e_deltas[-1] < threshold:
                break
        with Engine(2, spec) as eng:
            eng.vi_set(v, pi)
            steps, deltas = eng.vi_run(gamma, threshold, rounds)
            got_v, got_pi = eng.vi_get()
            assert steps == len(deltas_w) and deltas.tolist() == deltas_w, (trial, W, H, gamma)
            assert got_v.tobytes() == v_w.tobytes() and got_pi.tobytes() == pi_w.tobytes(), (trial, W, H, gamma)
            eng.vi_set(v, pi)
            steps, deltas = eng.vi_eval_run(gamma, threshold, rounds)
            got_v, got_pi = eng.vi_get()
            assert steps == len(e_deltas) and deltas.tolist() == e_deltas, (trial, W, H, gamma)
            assert got_v.tobytes() == e_v.tobytes() and got_pi.tobytes() == pi.tobytes()
            eng.vi_greedy(gamma)
            assert eng.vi_get()[1].tobytes() == C.greedy_policy(grid, gamma, e_v).tobytes()


@pytest.mark.parametrize('W,H', [(101, 101), (300, 300), (600, 600), (1024, 5)])
def test_cluster_kernel_and_launch_per_round_agree(W, H, vi_path, gu_option):
    """Grids of 4097 .. 524 288 states: the one-launch cluster kernel (grid barrier per round, K = 1 or 2 states per thread,
    up to 256 workgroups) against one launch per round (option vi_path = 1) -- value iteration with the device-side stopping
    rule, a policy-evaluation run, and sweeps from a random policy / value table with forced ties, as raw bytes."""
    if vi_path == 'launch_per_round':
        pytest.skip('vi_path = 2 disables the cluster kernel: nothing to compare')
    S = W * H
    rs = np.random.RandomState(W)
    walls = rs.choice(S, S // 5, replace=False)
    free = np.setdiff1d(np.arange(S), walls)
    spec = GridSpec(W, H, [int(free[0])], [int(x) for x in free[-3:]], [int(x) for x in free[5:9]], [int(x) for x in walls])
    out, form = {}, {}
    # '1': the default dispatch (the per-XCD launch where planes + values fit one workgroup's LDS, else the chip-wide cluster);
    # 'chip_wide': no per-XCD form; 'timeout': a grid-barrier timeout is injected into the chip-wide cluster -> tables restored,
    # launch-per-round path; 'xcd_gives_up': the per-XCD launch gives up at once -> tables restored, next form
    for cluster in ('0', '1', 'chip_wide', 'timeout', 'xcd_gives_up'):
        gu_option('vi_path', {'0': 1, '1': base_path(vi_path), 'chip_wide': 4, 'timeout': 3, 'xcd_gives_up': 5}[cluster])
        res = []
        with Engine(2, spec) as eng:
            eng.vi_set(np.zeros(S), np.ones((S, 4)) / 4)
            steps, deltas = eng.vi_run(0.9, 1e-3, 300)
            form[cluster] = [eng.vi_last_dp_form()]
            res += [steps, deltas.tobytes(), *(x.tobytes() for x in eng.vi_get())]
            res += [eng.vi_sweep(0.97, 3, greedy_update=True).tobytes(), *(x.tobytes() for x in eng.vi_get())]  # odd round count
            form[cluster].append(eng.vi_last_dp_form())
            pi = np.random.RandomState(3).dirichlet(np.ones(4), S)
            v = np.random.RandomState(4).randn(S) * 20
            v[::7] = np.round(v[::7])
            eng.vi_set(v, pi)
            steps, deltas = eng.vi_eval_run(0.9, 1e-2, 25)
            form[cluster].append(eng.vi_last_dp_form())
            res += [steps, deltas.tobytes(), *(x.tobytes() for x in eng.vi_get())]
            res += [eng.vi_sweep(1.0, 2, greedy_update=True).tobytes(), *(x.tobytes() for x in eng.vi_get())]
            res += [eng.vi_sweep(1.0, 1, greedy_update=False).tobytes(), *(x.tobytes() for x in eng.vi_get())]
            steps, deltas = eng.vi_run(1.0, 1e9, 5)  # stops after its first round
            res += [steps, deltas.tobytes(), *(x.tobytes() for x in eng.vi_get())]
        out[cluster] = res
    assert out['0'][0] > 3
    for cluster in ('1', 'chip_wide', 'timeout', 'xcd_gives_up'):
        assert out[cluster] == out['0'], cluster
    # ... and each mode really ran the form it names (vi_last_dp_form: 1 per XCD, 2 one workgroup, 3 chip-wide cluster, 4 one launch
    # per round) -- results alone cannot tell a form that silently gave up from one that ran: round 4's per-XCD kernel counted the
    # agents' action items for the tables alone and sent 1024x5 (and 125x32, 100x40 ..) through a failed launch and a restore
    per_xcd = 1 if S <= 32767 else 3
    want = {'0': 4, '1': per_xcd, 'chip_wide': 3, 'timeout': 4, 'xcd_gives_up': 3}
    for cluster, forms in form.items():
        assert forms == [want[cluster]] * 3, (cluster, forms)


@pytest.mark.parametrize('W,H', [(125, 32), (100, 40), (120, 30), (1024, 5), (64, 64), (8, 8)])
def test_the_tables_alone_take_the_per_xcd_launch_wherever_it_fits(W, H, vi_path, gu_option):
    """gu_vi_run / gu_vi_sweep / gu_vi_eval_run under the default dispatch: ONE launch of one XCD's workgroups finishes the call
    on every grid of up to 32 767 states -- among them the shapes whose halo fits the launch the plan chose but whose halo PLUS the
    agents' action items (which the tables alone do not exchange) would not."""
    if vi_path not in ('one_launch', 'per_xcd_everywhere'):
        pytest.skip('the default dispatch is what this test is about')
    S = W * H
    spec = GridSpec(W, H, [0], [S - 1], [S // 2], [])
    grid = C.Grid.from_lists(W=W, H=H, starts=[0], goals=[S - 1], lava=[S // 2], walls=[])
    with Engine(2, spec) as eng:
        eng.vi_set(np.zeros(S), np.ones((S, 4)) / 4)
        steps, deltas = eng.vi_run(0.9, 1e-3, 40)
        assert eng.vi_last_dp_form() == 1, (W, H)
        v, pi = np.zeros(S), np.ones((S, 4)) / 4
        for i in range(steps):
            v, pi, d = C.value_iteration_step(grid, 0.9, pi, v)
            assert d == deltas[i]
        got_v, got_pi = eng.vi_get()
        assert got_v.tobytes() == v.tobytes() and got_pi.tobytes() == pi.tobytes()
        eng.vi_sweep(0.9, 3, greedy_update=False)
        assert eng.vi_last_dp_form() == 1
        eng.vi_eval_run(0.9, 1e-2, 10)
        assert eng.vi_last_dp_form() == 1


@pytest.mark.parametrize('W,H', [(32, 32), (64, 64), (100, 40), (7, 3)])
def test_the_tables_a_dp_call_leaves_on_the_host_are_the_tables_on_the_device(W, H):
    """The per-XCD launch of gu_vi_run / gu_vi_sweep / gu_vi_eval_run writes its final tables to a page-locked copy on the host as well,
    and the gu_vi_get behind it is two memcpys (no launch, no wait).  That copy must BE the device's tables -- read here through a
    call that withdraws the copy without changing a table (gu_vi_run with max_steps = 0) -- after every kind of DP call, and it must be
    withdrawn by everything that writes a table: gu_vi_set, gu_vi_greedy, the fused sweep + step launches."""
    S = W * H
    spec = GridSpec(W, H, [0], [S - 1], [S // 2], [])
    rs = np.random.RandomState(W)
    with Engine(64, spec) as eng:
        eng.reset()

        def device_tables():
            eng.vi_run(0.9, 1e-3, 0)  # (no round: nothing changes, but the call withdraws the host's copy)
            return eng.vi_get()

        def same(a, b):
            return a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes()

        eng.vi_set(rs.rand(S), rs.dirichlet(np.ones(4), S))
        for call in (lambda: eng.vi_run(0.9, 1e-3, 25), lambda: eng.vi_sweep(0.9, 3, greedy_update=True), lambda: eng.vi_sweep(0.9, 2, greedy_update=False),
                     lambda: eng.vi_eval_run(0.9, 1e-2, 7), lambda: eng.vi_run(0.9, 1e-9, 1)):
            call()
            from_copy = eng.vi_get()
            assert same(from_copy, eng.vi_get())       # (twice from the copy)
            assert same(from_copy, device_tables()), (W, H)
        # ... and what writes a table behind such a call is seen by the next gu_vi_get
        eng.vi_run(0.9, 1e-3, 5)
        v2, pi2 = rs.rand(S), rs.dirichlet(np.ones(4), S)
        eng.vi_set(v2, pi2)
        assert same(eng.vi_get(), (v2, pi2))
        eng.vi_run(0.9, 1e-3, 5)
        before = eng.vi_get()
        eng.vi_greedy(0.9)
        after = eng.vi_get()
        assert after[0].tobytes() == before[0].tobytes() and same(after, device_tables())
        eng.vi_run(0.9, 1e-3, 5)
        eng.vi_sweep_step_run(0.9, 3, True)
        assert same(eng.vi_get(), device_tables())


@pytest.mark.parametrize('name,N,auto', [('maze64_s5', 65536, True), ('maze64_s5_g097', 4096, False), ('rect6x5_g1', 100, True),
                                         ('lava4x4_g095', 3, True), ('maze32_s1', 20000, True), ('maze64_s5', 262144, True)])
def test_sweep_step_run_is_iters_fused_launches_in_one(name, N, auto, vi_path, gu_option):
    """gu_vi_sweep_step_run (config 5 for many rounds in ONE launch: synchronised per XCD, or a workgroup cluster with a chip-wide
    barrier per round) against the same number of single gu_vi_sweep_step launches: value table, policy, deltas, and every env's
    position / reward / done flag / episode count, as raw bytes; and its first rounds against the reference's value-iteration
    trace.  Every form, every workgroup size of the per-XCD form, and both injected give-ups (state restored, next form)."""
    meta, z = G.load_dp(name)
    S, gamma = meta['W'] * meta['H'], meta['gamma']
    iters = 23
    out, form = {}, {}
    modes = {'single': None, 'run': None, 'run_xcd_256': None, 'run_xcd_512': None, 'run_xcd_1024': None, 'run_chip_wide': 4,
             'run_no_cluster': 1, 'run_timeout': 3, 'run_xcd_gives_up': 5}
    for mode, path in modes.items():
        if vi_path != 'launch_per_round':
            gu_option('vi_path', path if path is not None else base_path(vi_path))
        gu_option('vi_xcd_block', int(mode[8:]) if mode.startswith('run_xcd_') and mode[8:].isdigit() else None)
        with Engine(N, spec_of(meta), seed=3) as eng:
            eng.reset()
            eng.vi_set(np.zeros(S), np.ones((S, 4)) / 4)
            if mode == 'single':
                deltas = np.array([eng.vi_sweep_step(gamma, auto_reset=auto) for _ in range(iters)])
            else:
                deltas = eng.vi_sweep_step_run(gamma, iters, auto_reset=auto)
                form[mode] = eng.vi_last_form()
                # even + odd round counts; by default calls this short take one launch per round, under vi_path = 6 (and wherever
                # a mode sets a path of its own) the one-launch forms run them too
                deltas = np.concatenate([deltas, eng.vi_sweep_step_run(gamma, 2, auto_reset=auto)])
                eng.vi_sweep_step_run(gamma, 1, auto_reset=auto)
                if vi_path == 'launch_per_round' or (path is None and vi_path != 'per_xcd_everywhere'):  # (a mode without a path of its own runs the default dispatch)
                    assert eng.vi_last_form() == 3
                else:
                    assert eng.vi_last_form() == form[mode], (mode, eng.vi_last_form(), form[mode])
                if mode == 'run' and form[mode] == 1:  # the clusters as the hardware reported them: every workgroup of the launch in exactly one
                    clusters = eng.vi_last_clusters()
                    assert sum(clusters) >= max(8, -(-N // 1024)) and all(c >= 0 for c in clusters) and sum(c > 0 for c in clusters) >= 1, clusters
            if mode == 'single':
                deltas = np.concatenate([deltas, [eng.vi_sweep_step(gamma, auto_reset=auto) for _ in range(3)]])[:iters + 2]
            v, pi = eng.vi_get()
            st = eng.get_state()
            o = eng.read_outputs()
            out[mode] = [deltas.tobytes(), v.tobytes(), pi.tobytes(), st['pos'].tobytes(), st['done'].tobytes(), st['episode'].tobytes(),
                         o[1].tobytes(), eng.done_indices().tobytes()]
    gu_option('vi_xcd_block', None)
    for mode in modes:
        assert out[mode] == out['single'], mode
    if vi_path == 'launch_per_round':
        assert set(form.values()) == {3}
    else:
        # (262 144 envs need 1024-thread workgroups to stay at one per CU: the smaller sizes then decline and the chip-wide form runs)
        assert form['run'] == 1 and form['run_xcd_1024'] == 1 and form['run_xcd_256'] == (1 if N <= 65536 else 2), form
        assert form['run_chip_wide'] == 2 and form['run_xcd_gives_up'] == 2 and form['run_no_cluster'] == 3 and form['run_timeout'] == 3, form
    gu_option('vi_path', 2 if vi_path == 'launch_per_round' else base_path(vi_path))
    with Engine(N, spec_of(meta), seed=3) as eng:  # the table part is the reference's value-iteration trace
        eng.reset()
        eng.vi_set(np.zeros(S), np.ones((S, 4)) / 4)
        d = eng.vi_sweep_step_run(gamma, meta['iters'], auto_reset=auto)
        v, pi = eng.vi_get()
        assert d.tolist() == meta['deltas']
        assert v.tobytes() == z['vi_v_%d' % meta['iters']].tobytes() and pi.tobytes() == z['vi_pi_%d' % meta['iters']].tobytes()


# (128x128 and 101x101: two states per thread and two to three exchange loads per thread in the per-XCD kernel, its other instances)
@pytest.mark.parametrize('name,N', [('maze64_s5', 65536), ('maze128_s7', 20000), ('level101_g099', 5000), ('maze32_s1_g099', 3000), ('rect6x5_g1', 700),
                                    ('maze11_s3_g09', 64)])
@pytest.mark.parametrize('values', ['ties', 'huge', 'nonfinite'])
def test_sweep_step_run_forms_agree_on_random_tables(name, N, values, vi_path, gu_option):
    """Every form of gu_vi_sweep_step_run from RANDOM tables: exact ties of np.around(q, 8), values beyond 3.3e7 (where the tie
    test needs its IEEE divisions) and infinities / NaN -- the per-XCD form takes the maximum over rint(q * 1e8) instead of over q
    and drops the reference's `0.0 +` where it is provably the identity; its bytes must not notice."""
    if vi_path == 'launch_per_round':
        pytest.skip('vi_path = 2 leaves one form: nothing to compare')
    meta, _ = G.load_dp(name)
    S = meta['W'] * meta['H']
    rs = np.random.RandomState(11)
    out, form = {}, {}
    for gamma in (1.0, 0.9):
        pi = rs.dirichlet(np.ones(4), S)
        v = rs.standard_normal(S) * 50
        v[::7] = np.round(v[::7])
        if values == 'huge':
            v[::5] *= 1e7
            v[3::11] = 4e15
        if values == 'nonfinite':
            v[5::13] = np.inf
            v[6::17] = -np.inf
            v[7::19] = np.nan
        for mode, path in (('per_xcd', base_path(vi_path)), ('chip_wide', 4), ('per_launch', 1)):
            gu_option('vi_path', path)
            with Engine(N, spec_of(meta), seed=3) as eng:
                eng.reset()
                eng.vi_set(v, pi)
                with np.errstate(all='ignore'):
                    deltas = eng.vi_sweep_step_run(gamma, 7, auto_reset=True)
                    form[mode] = eng.vi_last_form()
                    deltas = np.concatenate([deltas, eng.vi_sweep_step_run(gamma, 2, auto_reset=True)])  # (by default: one launch per round)
                vv, pp = eng.vi_get()
                st = eng.get_state()
                out[mode] = [deltas.tobytes(), vv.tobytes(), pp.tobytes(), st['pos'].tobytes(), st['done'].tobytes(), st['episode'].tobytes(),
                             eng.read_outputs()[1].tobytes()]
        assert out['per_xcd'] == out['chip_wide'] == out['per_launch'], (gamma, [a == b for a, b in zip(out['per_xcd'], out['per_launch'])])
    assert form == {'per_xcd': 1, 'chip_wide': 2, 'per_launch': 3}


def test_calls_of_thousands_of_rounds_return_every_delta(vi_path, gu_option):
    """Calls longer than the 4096 deltas that come back with the control words in one copy (csrc/gu_vi_xcd.hip: gu_vi_xcd_dp_run),
    and a fused sweep + step call of thousands of rounds: round count, every delta and the tables equal those of the
    single-workgroup kernel / of one launch per round."""
    if vi_path != 'one_launch':
        pytest.skip('compares the default dispatch with the others itself')
    meta, _ = G.load_dp('maze8_s1')
    S = meta['W'] * meta['H']
    out = {}
    for mode, path in (('default', None), ('one_workgroup', 4), ('per_launch', 2)):
        gu_option('vi_path', path)
        with Engine(300, spec_of(meta), seed=2) as eng:
            eng.reset()
            eng.vi_set(np.zeros(S), np.ones((S, 4)) / 4)
            steps, deltas = eng.vi_run(0.999, -1.0, 5000)  # (a threshold no delta is below: every round runs)
            v, pi = eng.vi_get()
            res = [steps, deltas.tobytes(), v.tobytes(), pi.tobytes()]
            assert steps == 5000 and deltas.shape == (5000,)
            steps, deltas = eng.vi_eval_run(0.9, -1.0, 4100)
            res += [steps, deltas.tobytes(), eng.vi_get()[0].tobytes()]
            if mode != 'one_workgroup':
                eng.vi_set(np.zeros(S), np.ones((S, 4)) / 4)
                d = eng.vi_sweep_step_run(0.999, 4500, auto_reset=True)
                st = eng.get_state()
                res += [d.tobytes(), eng.vi_get()[0].tobytes(), st['pos'].tobytes(), st['episode'].tobytes()]
            out[mode] = res
    assert out['default'][:7] == out['one_workgroup'][:7] == out['per_launch'][:7]
    assert out['default'][7:] == out['per_launch'][7:]
