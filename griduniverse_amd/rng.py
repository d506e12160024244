"""Host-side view of the engine's per-env counter RNG (the device code is csrc/gu_rng.hpp).

The reference has no per-env RNG (it draws from process-global RNGs), so the engine defines
one, keyed by (seed, GLOBAL env id, stream, counter):

    word = MurmurHash3_x86_32 over the words [seed_lo, seed_hi, env, (stream << 28) | (counter & 0x0FFFFFFF)], hash seed
           0x9747B28C; counters of 2**28 and more append a fifth word, counter >> 28 (no stream repeats before 2**32 draws)
    stream 0  uniform actions : action(t) = (word(t >> 4) >> (2 * (t & 15))) & 3, t = steps the env has taken
    stream 1  start cell      : index = (word(episode) * n_starts) >> 32, episode = resets since gu_seed
    stream 2  sampled actions : u = word(t) / 2**32 against the cumulative policy row
    stream 3  maze generation : draw k of grid g = (word(k) * n) >> 32, keyed by (maze_seed, global grid id)

These helpers let a caller reproduce on the host exactly what a `policy='uniform'` rollout did on the device -- e.g.
to replay the same (grid, seed, action) sequence through the reference's own `step()`.
"""
import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def _rotl(x, r):
    return ((x << np.uint64(r)) | (x >> np.uint64(32 - r))) & _M32


def _block(h, k):
    k = (k * np.uint64(0xCC9E2D51)) & _M32
    k = _rotl(k, 15)
    k = (k * np.uint64(0x1B873593)) & _M32
    h = _rotl(h ^ k, 13)
    return (h * np.uint64(5) + np.uint64(0xE6546B64)) & _M32


def words(seed, env_ids, stream, counters):
    """uint32 RNG words for broadcastable arrays of global env ids and counters."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    env = np.asarray(env_ids, dtype=np.uint64) & _M32
    ctr = np.asarray(counters, dtype=np.uint64) & _M32
    env, ctr = np.broadcast_arrays(env, ctr)
    h = np.full(env.shape, 0x9747B28C, dtype=np.uint64)
    for k in (np.uint64(seed & 0xFFFFFFFF), np.uint64(seed >> 32), env,
              np.uint64((int(stream) & 0xF) << 28) | (ctr & np.uint64(0x0FFFFFFF))):
        h = _block(h, k)
    high = ctr >> np.uint64(28)
    h = np.where(high != 0, _block(h, high) ^ np.uint64(20), h ^ np.uint64(16))
    h = h ^ (h >> np.uint64(16))
    h = (h * np.uint64(0x85EBCA6B)) & _M32
    h = h ^ (h >> np.uint64(13))
    h = (h * np.uint64(0xC2B2AE35)) & _M32
    h = h ^ (h >> np.uint64(16))
    return h.astype(np.uint32)


def uniform_actions(seed, env_ids, t0, T):
    """int32[T, N]: the actions a `policy='uniform'` rollout takes for global env ids `env_ids`, steps t0 .. t0+T-1."""
    t = (np.arange(T, dtype=np.uint64) + np.uint64(t0))[:, None]
    w = words(seed, np.asarray(env_ids, dtype=np.uint64)[None, :], 0, t >> np.uint64(4)).astype(np.uint64)
    return ((w >> (np.uint64(2) * (t & np.uint64(15)))) & np.uint64(3)).astype(np.int32)


def start_indices(seed, env_ids, episodes, n_starts):
    """Index into starting_states chosen at reset number `episodes` (0 = first reset after gu_seed)."""
    w = words(seed, env_ids, 1, episodes).astype(np.uint64)
    return ((w * np.uint64(int(n_starts))) >> np.uint64(32)).astype(np.int32)
