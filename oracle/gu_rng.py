"""Build-defined per-env counter RNG -- CPU restatement (test infrastructure).

The reference has NO per-env RNG: its start choice
draws from the process-global stdlib Mersenne Twister
(core/envs/griduniverse_env.py:64,189) and random actions come from the caller
(examples/griduniverse_env_examples.py:18, core/algorithms/monte_carlo.py:20).
A batched engine needs a stream per env that does not depend on how the batch
is sharded, so the build defines one and this file is its specification:

    word(seed, env, stream, ctr) = MurmurHash3_x86_32 over the four 32-bit
        little-endian words [seed_lo, seed_hi, env, (stream << 28) | (ctr & 0x0FFFFFFF)]
        with hash seed 0x9747B28C   (Appleby's public-domain algorithm); when ctr >= 2**28 a
        fifth word, ctr >> 28, is appended (hashed length 20 instead of 16 bytes), so no
        stream repeats before 2**32 draws

    Streams 0 and 2 are keyed by the env's STEP COUNT t, a 64-bit number: their counter is bits 4 .. 31 of t
    ((t >> 4) & 0x0FFFFFFF), and the EPOCH t >> 32, when it is not zero, is hashed as one more word right behind the seed
    words -- [seed_lo, seed_hi, epoch, env, ...] -- with the length word left at 16 (word_at_step).  Epoch 0 is the plain
    four-word hash, so every draw of the first 2**32 steps is what it was when the count had 32 bits (and wrapped: at
    4.4e7 steps per second and env the action stream repeated after 97 s); beyond, no stream repeats before 2**64 steps.

    action(seed, env, t)      = (word_at_step(seed, env, 0, t) >> (2 * (t & 15))) & 3
    start_index(seed, env, e) = (word(seed, env, 1, e) * n_starts) >> 32          (keyed by the episode count: no epoch)
    sample_word(seed, env, t) = next^(t & 15)(word_at_step(seed, env, 2, t)):  ONE hashed word per SIXTEEN steps; the words of
                                the fifteen steps behind it by a multiply-free bijection of 32 bits,
                                next(x): x ^= x << 13; x ^= x >> 17; x ^= x << 5; x += 0x9E3779B9   (mod 2**32)
                                (Marsaglia's xorshift32 step followed by a Weyl increment).  Every word is a bijective
                                image of a MurmurHash3 output, so each draw by itself is exactly as uniform as the hash;
                                tests/test_oracle_mc.py checks the draws of a group against each other at every lag.
                                (Until round 3 every step hashed a word of its own: the five 32-bit multiplies of the
                                hash were what bound the sampled rollout on the GPU.  Round 4 went to one hash per four
                                steps, then per sixteen.)
    sampled(seed, env, t, p)  = #{k < 3 : u >= p[0] + .. + p[k]},  u = sample_word(seed, env, t) / 2**32
                                (float64 partial sums left to right; the batched form of
                                np.random.choice(4, p=policy[obs]), core/algorithms/monte_carlo.py:20)

`env` is the GLOBAL env index (so a sharded run equals the single-GPU run),
`t` the number of steps the env has taken since seeding, `e` its number of
resets since seeding.  The HIP implementation is csrc/gu_rng.hpp; the parity
tests drive the real reference with the numbers produced here.
"""
import numpy as np

H0 = 0x9747B28C
C1 = 0xCC9E2D51
C2 = 0x1B873593
M32 = 0xFFFFFFFF
STREAM_ACTION = 0
STREAM_START = 1
STREAM_SAMPLE = 2
CTR_MASK = 0x0FFFFFFF


def _rotl(x, r):
    return ((x << r) | (x >> (32 - r))) & M32


def _block(h, k):
    k = (k * C1) & M32
    k = _rotl(k, 15)
    k = (k * C2) & M32
    h ^= k
    h = _rotl(h, 13)
    return (h * 5 + 0xE6546B64) & M32


def _fmix(h):
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & M32
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & M32
    h ^= h >> 16
    return h


def word(seed, env, stream, ctr, epoch=0):
    """Scalar 32-bit output for (seed:uint64, env:uint32, stream:0..15, ctr:uint32[, epoch:uint32])."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    ctr = int(ctr) & M32
    epoch = int(epoch) & M32
    h = H0
    for k in (seed & M32, seed >> 32) + ((epoch,) if epoch else ()) + (int(env) & M32, ((int(stream) & 0xF) << 28) | (ctr & CTR_MASK)):
        h = _block(h, k)
    length = 16  # bytes hashed
    if ctr >> 28:
        h = _block(h, ctr >> 28)
        length = 20
    h ^= length
    return _fmix(h)


def word_at_step(seed, env, stream, t):
    """The word of stream 0 or 2 that covers step t (a 64-bit step count)."""
    t = int(t) & 0xFFFFFFFFFFFFFFFF
    return word(seed, env, stream, (t >> 4) & CTR_MASK, epoch=t >> 32)


def action(seed, env, t):
    return (word_at_step(seed, env, STREAM_ACTION, t) >> (2 * (int(t) & 15))) & 3


def start_index(seed, env, episode, n_starts):
    return (word(seed, env, STREAM_START, episode) * int(n_starts)) >> 32


SAMPLE_GROUP_LOG2 = 4  # 16 steps share one hashed word (csrc/gu_rng.hpp: GU_RNG_SAMPLE_LOG2)
SAMPLE_GROUP = 1 << SAMPLE_GROUP_LOG2


def sample_next(x):
    """The multiply-free bijection that turns one sampling word into the next one of its group of sixteen."""
    x ^= (x << 13) & M32
    x ^= x >> 17
    x ^= (x << 5) & M32
    return (x + 0x9E3779B9) & M32


def sample_word(seed, env, t):
    """The 32-bit word behind the sampled action of step t (stream 2): one hash per sixteen steps."""
    t = int(t) & 0xFFFFFFFFFFFFFFFF
    w = word_at_step(seed, env, STREAM_SAMPLE, t)
    for _ in range(t & (SAMPLE_GROUP - 1)):
        w = sample_next(w)
    return w


def sampled_action(seed, env, t, probs):
    u = sample_word(seed, env, t) / 4294967296.0
    c0 = float(probs[0])
    c1 = c0 + float(probs[1])
    c2 = c1 + float(probs[2])
    return int(u >= c0) + int(u >= c1) + int(u >= c2)


# ---------------------------------------------------------------- vectorised
def _vrotl(x, r):
    return ((x << np.uint64(r)) | (x >> np.uint64(32 - r))) & np.uint64(M32)


def _vblock(h, k):
    k = (k * np.uint64(C1)) & np.uint64(M32)
    k = _vrotl(k, 15)
    k = (k * np.uint64(C2)) & np.uint64(M32)
    h = h ^ k
    h = _vrotl(h, 13)
    return (h * np.uint64(5) + np.uint64(0xE6546B64)) & np.uint64(M32)


def _vfmix(h):
    h = h ^ (h >> np.uint64(16))
    h = (h * np.uint64(0x85EBCA6B)) & np.uint64(M32)
    h = h ^ (h >> np.uint64(13))
    h = (h * np.uint64(0xC2B2AE35)) & np.uint64(M32)
    h = h ^ (h >> np.uint64(16))
    return h


def word_v(seed, env, stream, ctr, epoch=None):
    """Vectorised `word`: env, ctr (and epoch) broadcast; returns uint32 array."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    env = np.asarray(env, dtype=np.uint64) & np.uint64(M32)
    ctr = np.asarray(ctr, dtype=np.uint64) & np.uint64(M32)
    epoch = np.zeros((), np.uint64) if epoch is None else np.asarray(epoch, dtype=np.uint64) & np.uint64(M32)
    env, ctr, epoch = np.broadcast_arrays(env, ctr, epoch)
    h = np.full(env.shape, H0, dtype=np.uint64)
    h = _vblock(h, np.uint64(seed & M32))
    h = _vblock(h, np.uint64(seed >> 32))
    if epoch.any():
        h = np.where(epoch != 0, _vblock(h, epoch), h)
    h = _vblock(h, env)
    h = _vblock(h, (np.uint64((int(stream) & 0xF) << 28)) | (ctr & np.uint64(CTR_MASK)))
    high = ctr >> np.uint64(28)
    h = np.where(high != 0, _vblock(h, high) ^ np.uint64(20), h ^ np.uint64(16))
    return _vfmix(h).astype(np.uint32)


def actions_v(seed, env, t):
    """Vectorised `action`; env, t broadcast.  Returns int32 array in 0..3."""
    t = np.asarray(t, dtype=np.uint64)
    w = word_v(seed, env, STREAM_ACTION, (t >> np.uint64(4)) & np.uint64(CTR_MASK), epoch=t >> np.uint64(32)).astype(np.uint64)
    return ((w >> (np.uint64(2) * (t & np.uint64(15)))) & np.uint64(3)).astype(np.int32)


def start_index_v(seed, env, episode, n_starts):
    w = word_v(seed, env, STREAM_START, episode).astype(np.uint64)
    return ((w * np.uint64(int(n_starts))) >> np.uint64(32)).astype(np.int32)


def action_stream(seed, env_ids, t0, T):
    """[T, N] int32 uniform actions for global env ids `env_ids`, steps t0..t0+T-1."""
    env_ids = np.asarray(env_ids, dtype=np.uint64)
    t = (np.arange(T, dtype=np.uint64) + np.uint64(t0))[:, None]
    return actions_v(seed, env_ids[None, :], t)
