"""Pins oracle/mc.py (and the stream-2 action sampling of the C oracle) to the goldens captured from the
reference's own monte_carlo_evaluation (run_episode patched to replay the fixture's episodes)."""
import numpy as np
import pytest

from oracle import c_oracle as C
from oracle import gu_rng
from oracle import mc as omc
from tests import _golden as G


@pytest.mark.parametrize('name', G.mc_names())
def test_mc_restatement_bit_exact(name):
    meta, z = G.load_mc(name)
    S = meta['W'] * meta['H']
    for run in meta['runs'][::2] if name == 'maze8_uniform' else meta['runs']:
        eps = omc.episodes_from_trajectory(z['first_state'], z['obs'], z['reward'], z['done'])
        v, visits = omc.monte_carlo_evaluation(S, eps, every_visit=run['every_visit'], incremental_mean=run['incremental_mean'],
                                               stationary_env=run['stationary_env'], discount_factor=run['discount_factor'],
                                               threshold=run['threshold'], alpha=run['alpha'])
        assert v.tobytes() == z[run['key']].tobytes(), (name, run)


@pytest.mark.parametrize('name', G.mc_names())
def test_sampled_episodes_reproduce(name):
    """The fixture's episodes come back from the C oracle (policy sampling on RNG stream 2), and the scalar
    Python RNG restatement agrees with it action by action."""
    meta, z = G.load_mc(name)
    grid = C.Grid.from_lists(**meta)
    st = C.State(meta['N'])
    assert np.array_equal(C.reset(grid, meta['seed'], st), z['first_state'])
    traj = C.rollout(grid, meta['seed'], st, meta['T'], auto_reset=False, pi=z['policy'])
    for k in ('obs', 'reward', 'done'):
        assert np.array_equal(traj[k], z[k])
    for e in (0, 7, meta['N'] - 1):
        s = int(z['first_state'][e])
        for t in range(min(meta['T'], 40)):
            a = gu_rng.sampled_action(meta['seed'], e, t, z['policy'][s])
            nxt = C.look_step_ahead(grid, [s], [a])[0][0]
            assert nxt == z['obs'][t, e]
            s = int(nxt)


def test_reference_rng_episode_generator_on_an_oracle_table_reproduces_the_reference():
    """The HOST half of monte_carlo_evaluation(rng='numpy') -- drawing the reference's episodes from the global `random` /
    `np.random` streams (algorithms/monte_carlo.py: reference_rng_episodes) -- needs only a transition table; here the table
    comes from the oracle env instead of the look-ahead kernel, the episodes are reduced by the oracle's evaluation, and the
    result must be the value function the REAL reference returned for the same seeds (tests/golden/mcnp_*.npz), with both
    global streams left where the reference leaves them."""
    import glob
    import os
    import random

    import numpy as np

    from griduniverse_amd.algorithms.monte_carlo import reference_rng_episodes
    from oracle import mc as omc
    from oracle.ref_env import OracleGridUniverseEnv
    from tests import _golden as G

    names = sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(G.GOLDEN, 'mcnp_*.npz')))
    assert len(names) >= 3
    np_state, py_state = np.random.get_state(), random.getstate()
    try:
        for name in names:
            meta, z = G.load_npz('mcnp', name)
            env = OracleGridUniverseEnv(grid_shape=(meta['W'], meta['H']), initial_state=meta['starts'], goal_states=meta['goals'],
                                        lava_states=meta['lava'], walls=meta['walls'])
            S = meta['W'] * meta['H']
            table = [[env.look_step_ahead(s, a) for a in range(4)] for s in range(S)]
            env._transition_table = lambda care, t=table: (np.array([[c[0] for c in row] for row in t]),
                                                           np.array([[c[1] for c in row] for row in t]),
                                                           np.array([[c[2] for c in row] for row in t]))
            for run in meta['runs'][::3]:
                random.seed(meta['seed'])
                np.random.seed(meta['seed'])
                first, actions, lengths, last_states, last_done = reference_rng_episodes(z['policy'], env, meta['num_episodes'])
                assert float(np.random.random_sample()) == run['next_numpy_uniform'] and random.random() == run['next_stdlib_uniform']
                episodes = []
                for e in range(meta['num_episodes']):
                    s, states, rewards = int(first[e]), [int(first[e])], []
                    for t in range(int(lengths[e])):
                        s, r, d = table[s][int(actions[t, e])]
                        states.append(s)
                        rewards.append(r)
                    episodes.append((states, rewards, d))
                assert episodes[-1][0] == last_states and episodes[-1][2] == last_done
                v, _ = omc.monte_carlo_evaluation(S, episodes, run['every_visit'], run['incremental_mean'], run['stationary_env'],
                                                  run['discount_factor'], run['threshold'], run['alpha'])
                assert v.tobytes() == z[run['key']].tobytes(), (name, run)
    finally:
        np.random.set_state(np_state)
        random.setstate(py_state)


def test_sampling_words_one_hash_per_sixteen_steps_c_oracle_equals_the_specification():
    """RNG stream 2 (round 4): the word of step t is the hashed word of group t >> 4 advanced t & 15 times by the multiply-free
    bijection -- oracle/gu_oracle.c against oracle/gu_rng.py, across group boundaries and the 2^28 counter boundary; the
    bijection is one (no two of 2^16 consecutive inputs collide, and it has no short cycle through 0)."""
    lib = C.lib()
    for seed, env in ((0, 0), (7, 12345), (0xDEADBEEFCAFE, 0xFFFFFFFF)):
        for t in list(range(0, 70)) + [2 ** 32 - 18, 2 ** 32 - 17, 2 ** 32 - 16, 2 ** 32 - 11, 2 ** 32 - 1]:
            assert lib.gu_oracle_rng_sample_word(seed, env, t) == gu_rng.sample_word(seed, env, t), (seed, env, t)
        for t in range(0, 64, 16):  # the first word of a group IS the hash of stream 2 at counter t >> 4
            assert gu_rng.sample_word(seed, env, t) == gu_rng.word(seed, env, gu_rng.STREAM_SAMPLE, t >> 4)
            for k in range(15):
                assert gu_rng.sample_word(seed, env, t + k + 1) == gu_rng.sample_next(gu_rng.sample_word(seed, env, t + k))
    images = {gu_rng.sample_next(x) for x in range(1 << 16)}
    assert len(images) == 1 << 16
    x, seen = 0, set()
    for _ in range(1000):
        x = gu_rng.sample_next(x)
        assert x not in seen
        seen.add(x)


def test_sampled_actions_follow_the_policy_chi_square_over_two_million_draws():
    """The draw's distribution: chi-square of the action counts against pi over 2^21 draws (one env stream and many envs, so that
    both the words within a group of sixteen and the groups are covered), for a flat, a skewed and a nearly one-hot row; every
    position of the group on its own; and the draws of one group against each other at EVERY lag 1 .. 15 (the 4 x 4 table of
    the action pair, and the correlation of the words themselves)."""
    G = gu_rng.SAMPLE_GROUP
    n_env, T = 512, 4096
    words = np.empty((n_env, T), np.uint32)
    hashed = gu_rng.word_v(5, np.arange(n_env)[:, None], gu_rng.STREAM_SAMPLE, np.arange(T // G)[None, :]).astype(np.uint64)
    cur = hashed
    for j in range(G):
        words[:, j::G] = cur.astype(np.uint32)
        cur = cur ^ ((cur << np.uint64(13)) & np.uint64(0xFFFFFFFF))
        cur = cur ^ (cur >> np.uint64(17))
        cur = cur ^ ((cur << np.uint64(5)) & np.uint64(0xFFFFFFFF))
        cur = (cur + np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF)
    assert words[3, 9] == gu_rng.sample_word(5, 3, 9) and words[511, 4095] == gu_rng.sample_word(5, 511, 4095)
    assert words[17, 1000] == gu_rng.sample_word(5, 17, 1000)
    u = words.astype(np.float64) / 4294967296.0
    n = u.size
    for p in ([0.25, 0.25, 0.25, 0.25], [0.6, 0.05, 0.3, 0.05], [0.001, 0.997, 0.001, 0.001]):
        c = np.cumsum(p)[:3]
        acts = (u >= c[0]).astype(np.int64) + (u >= c[1]) + (u >= c[2])
        counts = np.bincount(acts.ravel(), minlength=4)
        chi2 = float(np.sum((counts - n * np.asarray(p)) ** 2 / (n * np.asarray(p))))
        assert chi2 < 21.1, (p, counts, chi2)  # 3 degrees of freedom: P(chi2 > 21.1) = 1e-4
        for j in range(G):  # ... and the same for each position inside the group on its own
            cj = np.bincount(acts[:, j::G].ravel(), minlength=4)
            chi2 = float(np.sum((cj - n / G * np.asarray(p)) ** 2 / (n / G * np.asarray(p))))
            assert chi2 < 27.9, (p, j, cj, chi2)  # P(chi2_3 > 27.9) = 4e-6: 48 such tests
    # the draws of one group are not tied to each other, at any distance inside the group: the 4 x 4 table of (action at position
    # j, action at position j + lag) over all groups, for a flat and a skewed row, and the correlation of the uniforms
    grouped = u.reshape(n_env, T // G, G)
    for p in ([0.25, 0.25, 0.25, 0.25], [0.6, 0.05, 0.3, 0.05]):
        c = np.cumsum(p)[:3]
        acts = (grouped >= c[0]).astype(np.int64) + (grouped >= c[1]) + (grouped >= c[2])
        expect = np.outer(p, p).ravel()
        for lag in range(1, G):
            a0, a1 = acts[:, :, :G - lag].ravel(), acts[:, :, lag:].ravel()
            pair = np.bincount(a0 * 4 + a1, minlength=16)
            m = a0.size
            chi2 = float(np.sum((pair - m * expect) ** 2 / (m * expect)))
            assert chi2 < 50.0, (p, lag, chi2)  # 15 degrees of freedom: P(chi2 > 50) = 1e-5; 30 such tests
    for lag in range(1, G):
        x, y = grouped[:, :, :G - lag].ravel(), grouped[:, :, lag:].ravel()
        r = float(np.corrcoef(x, y)[0, 1])
        assert abs(r) < 5.0 / np.sqrt(x.size), (lag, r)  # five standard errors


def test_sampled_action_at_the_edges_of_the_policy_simplex():
    """p = 0, p = 1 and one-hot rows: the action is decided whatever the word, including the words 0 and 2^32 - 1."""
    lib = C.lib()
    lib.gu_oracle_rng_sample.restype = np.ctypeslib.ctypes.c_int32
    for hot in range(4):
        row = np.zeros(4)
        row[hot] = 1.0
        for e in range(64):
            for t in range(8):
                assert gu_rng.sampled_action(3, e, t, row) == hot
    for row, allowed in (([0.5, 0.0, 0.5, 0.0], {0, 2}), ([0.0, 0.0, 0.25, 0.75], {2, 3}), ([0.0, 1.0, 0.0, 0.0], {1})):
        got = {gu_rng.sampled_action(9, e, t, row) for e in range(64) for t in range(16)}
        assert got <= allowed and (len(allowed) == 1 or got == allowed)
