"""Pins the batched C oracle (oracle/gu_oracle.c) and the DP restatements to the goldens
captured from the real reference."""
import hashlib

import numpy as np
import pytest

from oracle import c_oracle as C
from oracle import dp as odp
from oracle import gu_rng
from oracle.ref_env import OracleGridUniverseEnv
from tests import _golden as G


def digest(obs, rew, don):
    h = hashlib.sha256()
    for a in (obs, rew, don):
        h.update(np.ascontiguousarray(a, dtype='<i4').tobytes())
    return h.hexdigest()


def test_rng_c_equals_python():
    lib = C.lib()
    rs = np.random.RandomState(0)
    for _ in range(300):
        seed = int(rs.randint(0, 2 ** 63 - 1))
        env = int(rs.randint(0, 2 ** 31 - 1)) * 2 + int(rs.randint(0, 2))
        ctr = int(rs.randint(0, 2 ** 28))
        for stream in (0, 1, 5):
            assert lib.gu_oracle_rng_word(seed, env, stream, ctr) == gu_rng.word(seed, env, stream, ctr)
        t = int(rs.randint(0, 2 ** 31 - 1))
        assert lib.gu_oracle_rng_action(seed, env, t) == gu_rng.action(seed, env, t)
        n = int(rs.randint(1, 5000))
        assert lib.gu_oracle_rng_start(seed, env, ctr, n) == gu_rng.start_index(seed, env, ctr, n)
    assert lib.gu_oracle_rng_word(2 ** 64 - 1, 2 ** 32 - 1, 15, 2 ** 28 - 1) == gu_rng.word(2 ** 64 - 1, 2 ** 32 - 1, 15, 2 ** 28 - 1)


def _murmur3_x86_32(data, seed):
    """Byte-oriented MurmurHash3_x86_32 as published (Appleby, public domain), written independently of oracle/gu_rng.py."""
    M = 0xFFFFFFFF
    h, n = seed & M, len(data)
    rot = lambda x, r: ((x << r) | (x >> (32 - r))) & M  # noqa: E731
    for i in range(0, n - n % 4, 4):
        k = int.from_bytes(data[i:i + 4], 'little')
        k = rot((k * 0xCC9E2D51) & M, 15) * 0x1B873593 & M
        h = (rot(h ^ k, 13) * 5 + 0xE6546B64) & M
    if n % 4:
        k = int.from_bytes(data[n - n % 4:], 'little')
        h ^= rot((k * 0xCC9E2D51) & M, 15) * 0x1B873593 & M
    h ^= n
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & M
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & M
    return h ^ (h >> 16)


def test_rng_is_murmurhash3_incl_counters_beyond_2_pow_28():
    # published known answers of MurmurHash3_x86_32
    for data, seed, want in ((b'', 0, 0), (b'', 1, 0x514E28B7), (b'', 0xFFFFFFFF, 0x81F16F39), (b'\xff\xff\xff\xff', 0, 0x76293B50),
                             (b'\x21\x43\x65\x87', 0, 0xF55B516B), (b'\x21\x43\x65\x87', 0x5082EDEE, 0x2362F9DE),
                             (b'\x21\x43\x65', 0, 0x7E4A8634), (b'\x21\x43', 0, 0xA0F7B07A), (b'\x21', 0, 0x72661CF4),
                             (b'\0\0\0\0', 0, 0x2362F9DE), (b'\0\0\0', 0, 0x85F0B427), (b'\0\0', 0, 0x30F4C306), (b'\0', 0, 0x514E28B7),
                             (b'Hello, world!', 0x9747B28C, 0x24884CBA),
                             (b'The quick brown fox jumps over the lazy dog', 0x9747B28C, 0x2FA826CD)):
        assert _murmur3_x86_32(data, seed) == want, (data, seed)
    lib = C.lib()
    rs = np.random.RandomState(5)
    for i in range(200):
        seed = int(rs.randint(0, 2 ** 63 - 1)) * 2 + 1
        env, stream = int(rs.randint(0, 2 ** 32, dtype=np.uint64)), int(rs.randint(0, 4))
        ctr = int(rs.randint(0, 2 ** 32, dtype=np.uint64)) if i % 2 else int(rs.randint(0, 2 ** 28))
        words = [seed & 0xFFFFFFFF, seed >> 32, env, (stream << 28) | (ctr & 0x0FFFFFFF)] + ([ctr >> 28] if ctr >> 28 else [])
        want = _murmur3_x86_32(b''.join(w.to_bytes(4, 'little') for w in words), 0x9747B28C)
        assert gu_rng.word(seed, env, stream, ctr) == want == lib.gu_oracle_rng_word(seed, env, stream, ctr)
        assert int(gu_rng.word_v(seed, [env], stream, [ctr])[0]) == want
    # the sampled-action stream no longer repeats after 2^28 steps
    assert gu_rng.word(7, 3, 2, 5) != gu_rng.word(7, 3, 2, 5 + 2 ** 28) != gu_rng.word(7, 3, 2, 5 + 2 ** 29)


def test_step_counts_have_64_bits_and_the_epoch_keys_streams_0_and_2():
    """oracle/gu_rng.py: streams 0 and 2 are keyed by a 64-bit step count t -- counter = bits 4 .. 31, epoch = t >> 32 hashed
    right behind the seed words when it is not zero (length word 16).  Epoch 0 is the four-word hash of before; beyond 2^32 steps
    nothing repeats; scalar Python, vectorised Python and C agree everywhere."""
    lib = C.lib()
    rs = np.random.RandomState(11)
    for i in range(200):
        seed = int(rs.randint(0, 2 ** 63 - 1)) * 2 + 1
        env = int(rs.randint(0, 2 ** 32, dtype=np.uint64))
        t = int(rs.randint(0, 2 ** 63, dtype=np.uint64)) * 2 + (i & 1) if i % 4 else int(rs.randint(0, 2 ** 32, dtype=np.uint64))
        epoch, ctr = t >> 32, (t >> 4) & 0x0FFFFFFF
        for stream in (0, 2):
            keys = [seed & 0xFFFFFFFF, seed >> 32] + ([epoch] if epoch else []) + [env, (stream << 28) | ctr]
            h = 0x9747B28C  # written out once more, independently of gu_rng._block
            for k in keys:
                k = (k * 0xCC9E2D51) & 0xFFFFFFFF
                k = ((k << 15) | (k >> 17)) & 0xFFFFFFFF
                k = (k * 0x1B873593) & 0xFFFFFFFF
                h ^= k
                h = ((h << 13) | (h >> 19)) & 0xFFFFFFFF
                h = (h * 5 + 0xE6546B64) & 0xFFFFFFFF
            h ^= 16
            h ^= h >> 16
            h = (h * 0x85EBCA6B) & 0xFFFFFFFF
            h ^= h >> 13
            h = (h * 0xC2B2AE35) & 0xFFFFFFFF
            h ^= h >> 16
            assert gu_rng.word_at_step(seed, env, stream, t) == h
            if not epoch:
                assert h == _murmur3_x86_32(b''.join(w.to_bytes(4, 'little') for w in keys), 0x9747B28C)  # the plain hash
        assert gu_rng.action(seed, env, t) == lib.gu_oracle_rng_action(seed, env, t) == int(gu_rng.actions_v(seed, [env], np.array([t], np.uint64))[0])
        assert gu_rng.sample_word(seed, env, t) == lib.gu_oracle_rng_sample_word(seed, env, t)
    # a stream of actions across the first epoch boundary: what a 32-bit count would have replayed is not replayed
    first = gu_rng.action_stream(3, range(64), 0, 512)
    across = gu_rng.action_stream(3, range(64), 2 ** 32 - 256, 512)
    assert np.array_equal(across[:256], gu_rng.action_stream(3, range(64), 2 ** 32 - 256, 256))
    assert not np.array_equal(across[256:], first[:256])
    assert [gu_rng.action(3, 5, 2 ** 32 + k) for k in range(64)] == across[256:320, 5].tolist()
    # the C rollout counts in 64 bits
    grid = C.Grid.from_lists(8, 8)
    st = C.State(16)
    C.reset(grid, 3, st)
    st.tcount[:] = 2 ** 32 - 100
    out = C.rollout(grid, 3, st, 200, True)
    assert st.tcount.dtype == np.uint64 and np.all(st.tcount == 2 ** 32 + 100)
    ref = C.State(16)
    C.reset(grid, 3, ref)
    again = C.rollout(grid, 3, ref, 200, True, actions=gu_rng.action_stream(3, range(16), 2 ** 32 - 100, 200))
    assert all(np.array_equal(out[k], again[k]) for k in ('obs', 'reward', 'done'))


@pytest.mark.parametrize('name', G.traj_names())
def test_trajectories_c_oracle(name):
    meta, z = G.load_traj(name)
    grid = C.Grid.from_lists(**meta)
    T, N = z['actions'].shape
    for use_rng_actions in (False, True):
        if use_rng_actions and 'RandomState' in meta['note']:
            continue  # that fixture's actions come from numpy, not from the build RNG
        st = C.State(N, meta['env_id0'])
        first = C.reset(grid, meta['seed'], st)
        assert np.array_equal(first, z['first_state'])
        out = C.rollout(grid, meta['seed'], st, T, meta['auto_reset'], None if use_rng_actions else z['actions'],
                        stats=True)
        for k in ('obs', 'reward', 'done'):
            assert np.array_equal(out[k], z[k]), (name, k)
        assert digest(out['obs'], out['reward'], out['done']) == meta['sha256']
        assert np.array_equal(out['ret'], z['reward'].sum(0)) and np.array_equal(out['episodes'], z['done'].sum(0))
        assert np.array_equal(st.pos, z['obs'][-1]) and np.array_equal(st.done, z['done'][-1])
        assert np.all(st.tcount == T)


def test_rollout_is_resumable():
    meta, z = G.load_traj('rect25x30_busy')
    grid = C.Grid.from_lists(**meta)
    T, N = z['actions'].shape
    st = C.State(N, meta['env_id0'])
    C.reset(grid, meta['seed'], st)
    parts = [C.rollout(grid, meta['seed'], st, n, True) for n in (1, 100, T - 101)]
    for k in ('obs', 'reward', 'done'):
        assert np.array_equal(np.concatenate([p[k] for p in parts]), z[k])


@pytest.mark.parametrize('name', sorted(G.load_json('digests.json')))
def test_large_digests(name):
    """4096 envs x 1000 steps of the real reference, held as sha256."""
    d = G.load_json('digests.json')[name]
    grid = C.Grid.from_lists(**d)
    st = C.State(d['N'])
    C.reset(grid, d['seed'], st)
    acts = None
    if 'stream_seed' in d:  # a caller-supplied stream (numpy's RandomState) instead of the counter RNG's actions
        acts = np.random.RandomState(d['stream_seed']).randint(0, 4, size=(d['T'], d['N'])).astype(np.int32)
    out = C.rollout(grid, d['seed'], st, d['T'], d['auto_reset'], actions=acts)
    assert digest(out['obs'], out['reward'], out['done']) == d['sha256']
    assert int(out['reward'].sum()) == d['sum_reward'] and int(out['done'].sum()) == d['sum_done']


def test_look_step_ahead_table():
    t = G.load_json('render_quirks.json')['quirks']['lsa_table_6x5']
    grid = C.Grid.from_lists(**t['spec'])
    s, a = np.meshgrid(np.arange(30), np.arange(4), indexing='ij')
    for care in (True, False):
        n, r, d = C.look_step_ahead(grid, s.ravel(), a.ravel(), care)
        want = np.array(t['table'][str(care)], dtype=np.int64).reshape(-1, 3)
        assert np.array_equal(np.stack([n, r, d], 1), want)


def _oracle_env(meta):
    return OracleGridUniverseEnv(grid_shape=(meta['W'], meta['H']), initial_state=list(meta['starts']),
                                 goal_states=list(meta['goals']), lava_states=list(meta['lava']), walls=list(meta['walls']))


@pytest.mark.parametrize('name', G.dp_names())
def test_dp_c_oracle_bit_exact(name):
    meta, z = G.load_dp(name)
    grid = C.Grid.from_lists(**meta)
    S, gamma = grid.S, meta['gamma']
    pi0 = np.ones((S, 4)) / 4
    v = np.zeros(S)
    for k in range(1, 11):
        v = C.policy_evaluation_sweep(grid, gamma, pi0, v)
        if k in (1, 2, 10):
            assert v.tobytes() == z['eval_v_%d' % k].tobytes(), (name, k)
    assert C.greedy_policy(grid, gamma, v).tobytes() == z['greedy_pi_after_10'].tobytes()
    v, pi = np.zeros(S), pi0.copy()
    for k in range(1, meta['iters'] + 1):
        v, pi, delta = C.value_iteration_step(grid, gamma, pi, v)
        if 'vi_v_%d' % k in z:  # (grids beyond 4096 states keep rounds 1, 2 and the last one)
            assert v.tobytes() == z['vi_v_%d' % k].tobytes(), (name, k)
            assert pi.tobytes() == z['vi_pi_%d' % k].tobytes(), (name, k)
        assert delta == meta['deltas'][k - 1]


@pytest.mark.parametrize('name', [n for n in G.dp_names() if not any(big in n for big in ('maze64', 'maze32', 'level101', 'maze128'))])
def test_dp_python_oracle_bit_exact(name):
    meta, z = G.load_dp(name)
    env = _oracle_env(meta)
    S, gamma = env.world.size, meta['gamma']
    v = np.zeros(S)
    for k in range(1, 11):
        v = odp.single_step_policy_evaluation(np.ones((S, 4)) / 4, env, discount_factor=gamma, value_function=v)
        if k in (1, 2, 10):
            assert v.tobytes() == z['eval_v_%d' % k].tobytes()
    pi = odp.greedy_policy_from_value_function(np.ones((S, 4)) / 4, env, v, discount_factor=gamma)
    assert pi.tobytes() == z['greedy_pi_after_10'].tobytes()
    import warnings
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        v3, pi3 = odp.value_iteration(np.ones((S, 4)) / 4, env, np.zeros(S), threshold=meta['driver_threshold'],
                                      max_steps=meta['iters'], discount_factor=gamma)
    assert v3.tobytes() == z['vi_driver_v'].tobytes() and pi3.tobytes() == z['vi_driver_pi'].tobytes()
    assert (len(w) > 0) == meta['driver_warned']
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        v4, pi4 = odp.policy_iteration(np.ones((S, 4)) / 4, env, np.zeros(S), threshold=1e-3,
                                       max_steps=meta['pi_driver_max_steps'], discount_factor=gamma)
    assert v4.tobytes() == z['pi_driver_v'].tobytes() and pi4.tobytes() == z['pi_driver_pi'].tobytes()
    assert (len(w) > 0) == meta['pi_driver_warned']


@pytest.mark.parametrize('name', G.traj_names())
def test_trajectories_numpy_batch_oracle(name):
    """oracle/np_env.py (the vectorised-numpy CPU baseline of bench.py) against the reference trajectories."""
    from oracle.np_env import NumpyBatchEnv
    meta, z = G.load_traj(name)
    T, N = z['actions'].shape
    T = min(T, 300)
    env = NumpyBatchEnv(meta['W'], meta['H'], meta['starts'], meta['goals'], meta['lava'], meta['walls'], meta['reward'], N,
                        meta['seed'], meta['env_id0'])
    assert np.array_equal(env.reset(), z['first_state'])
    got = env.rollout(T, meta['auto_reset'], z['actions'])
    for k in ('obs', 'reward', 'done'):
        assert np.array_equal(got[k], z[k][:T]), k
    if 'RandomState' not in meta['note']:  # the build RNG produced the fixture's actions: the on-the-fly stream equals it
        env = NumpyBatchEnv(meta['W'], meta['H'], meta['starts'], meta['goals'], meta['lava'], meta['walls'], meta['reward'], N,
                            meta['seed'], meta['env_id0'])
        env.reset()
        assert np.array_equal(env.rollout(T, meta['auto_reset'])['obs'], z['obs'][:T])
