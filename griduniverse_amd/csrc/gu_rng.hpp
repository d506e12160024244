// gu_rng.hpp -- per-env counter RNG (device + host), gfx950.
//
// Specification: oracle/gu_rng.py (MurmurHash3_x86_32 over the four words
// [seed_lo, seed_hi, global_env_id, (stream << 28) | (counter & 0x0FFFFFFF)], hash seed 0x9747B28C;
// a counter of 2^28 or more appends a fifth word, counter >> 28, so the streams do not repeat
// before 2^32 draws).  Streams 0 and 2 are keyed by the env's 64-bit STEP COUNT t: counter = (t >> 4) & 0x0FFFFFFF, and the
// EPOCH t >> 32, when it is not zero, is hashed as one more word right behind the seed (gu_rng_seed_prefix_epoch; the length
// word stays 16) -- the launcher folds it into the seed prefix it hands to the kernels, whose loops keep a 32-bit count.
// The reference has no per-env RNG (core/envs/griduniverse_env.py:64,189 use the
// process-global stdlib RNG), so this is build-defined and
// restated on the CPU by the oracle.
//
// The first three blocks depend only on (seed, env) and are hoisted out of the step
// loop (`gu_rng_prefix`); one 32-bit word then costs one block + the finaliser and
// feeds 16 two-bit actions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GU_RNG_STREAM_ACTION 0u
#define GU_RNG_STREAM_START 1u
#define GU_RNG_STREAM_SAMPLE 2u
#ifndef GU_RNG_SAMPLE_LOG2
#define GU_RNG_SAMPLE_LOG2 4u  // steps per hashed word of stream 2: 1 << this = 16 (oracle/gu_rng.py: SAMPLE_GROUP_LOG2)
#endif
#define GU_RNG_SAMPLE_MASK ((1u << GU_RNG_SAMPLE_LOG2) - 1u)

__host__ __device__ __forceinline__ uint32_t gu_rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

__host__ __device__ __forceinline__ uint32_t gu_mm3_block(uint32_t h, uint32_t k)
{
    k *= 0xCC9E2D51u;
    k = gu_rotl32(k, 15);
    k *= 0x1B873593u;
    h ^= k;
    h = gu_rotl32(h, 13);
    return h * 5u + 0xE6546B64u;
}

// state after hashing seed_lo, seed_hi (host side, once per gu_seed)
__host__ __device__ __forceinline__ uint32_t gu_rng_seed_prefix(uint64_t seed)
{
    uint32_t h = 0x9747B28Cu;
    h = gu_mm3_block(h, (uint32_t)seed);
    return gu_mm3_block(h, (uint32_t)(seed >> 32));
}

// ... and after the epoch (step count >> 32) of streams 0 and 2, where it is not zero (host side, once per launch; per lane and
// step only in a launch during which some env passes a multiple of 2^32 steps)
__host__ __device__ __forceinline__ uint32_t gu_rng_seed_prefix_epoch(uint32_t seed_prefix, uint32_t epoch)
{
    return epoch ? gu_mm3_block(seed_prefix, epoch) : seed_prefix;
}

// state after additionally hashing the global env id (once per lane per launch)
__host__ __device__ __forceinline__ uint32_t gu_rng_prefix(uint32_t seed_prefix, uint32_t env)
{
    return gu_mm3_block(seed_prefix, env);
}

// The word in two halves, so that a latency-bound loop can place each half in the shadow of a different LDS round trip
// (gu_rollout_multi.hip): gu_rng_word(p, s, c) == gu_rng_word_finish(gu_rng_word_begin(p, s, c)).
__host__ __device__ __forceinline__ uint32_t gu_rng_word_begin(uint32_t prefix, uint32_t stream, uint32_t ctr)
{
    uint32_t h = gu_mm3_block(prefix, (stream << 28) | (ctr & 0x0FFFFFFFu));
    uint32_t len = 16u;
    const uint32_t hi = ctr >> 28;  // beyond 2^28 draws of one stream: the high counter bits are a fifth key word
#if defined(__HIP_DEVICE_COMPILE__)
    // a WAVE-UNIFORM branch (one scalar test; the whole wave passes the block by in the practically universal case), instead of
    // a per-lane one that would be executed under an empty EXEC mask on every draw; marked unlikely, so that the block is laid out
    // of line and the usual path falls through (a TAKEN branch costs ~60 clocks where a SIMD has one wave)
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(hi != 0u) != 0ull, 0)) {
        const uint32_t h5 = gu_mm3_block(h, hi);
        h = hi ? h5 : h;
        len = hi ? 20u : 16u;
    }
#else
    if (hi) {
        h = gu_mm3_block(h, hi);
        len = 20u;
    }
#endif
    h ^= len;
    h ^= h >> 16;
    return h * 0x85EBCA6Bu;
}

__host__ __device__ __forceinline__ uint32_t gu_rng_word_finish(uint32_t h)
{
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

__host__ __device__ __forceinline__ uint32_t gu_rng_word(uint32_t prefix, uint32_t stream, uint32_t ctr)
{
    return gu_rng_word_finish(gu_rng_word_begin(prefix, stream, ctr));
}

// Stream 2, the words behind the inverse-CDF samples of a table policy (oracle/gu_rng.py: sample_word): ONE hashed word per
// SIXTEEN steps, the words of the fifteen steps behind it by a multiply-free bijection (xorshift32 step + Weyl increment).  A
// hash per step was what bound the sampled rollout (five quarter-rate 32-bit multiplies, ~80 clocks of issue time per step
// against the ~30 of the rest of the step); one per four steps still left it 9 of a step's 34 instructions, and the sampled
// rollout with int32 rows at 0.80 of the HBM peak; one per sixteen: 0.87 (profiles/archive/r04z_sample_rollout.json).
__host__ __device__ __forceinline__ uint32_t gu_rng_sample_next(uint32_t x)
{
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    return x + 0x9E3779B9u;
}

__host__ __device__ __forceinline__ uint32_t gu_rng_sample_word(uint32_t prefix, uint32_t t)
{
    uint32_t w = gu_rng_word(prefix, GU_RNG_STREAM_SAMPLE, t >> GU_RNG_SAMPLE_LOG2);
    for (uint32_t i = 0; i < (t & GU_RNG_SAMPLE_MASK); ++i) w = gu_rng_sample_next(w);
    return w;
}

// The word of step t + 1 from the word of step t.  Device: the hash is behind a WAVE-UNIFORM test, so a wave whose lanes are at
// one step count (every launch that did not start from a per-env gu_set_state) hashes once in sixteen steps.  The rollout kernels
// use this form only where the lanes' step counts differ or are not yet a multiple of sixteen; where they agree the schedule is
// unrolled (gu_rng_sample_advance_at): the test is a ballot and a branch per step, and with one wave per SIMD a taken branch
// costs ~60 clocks -- fifteen steps in sixteen.
__device__ __forceinline__ uint32_t gu_rng_sample_advance(uint32_t prefix, uint32_t t, uint32_t word)
{
    uint32_t next = gu_rng_sample_next(word);
    const bool fresh = ((t + 1u) & GU_RNG_SAMPLE_MASK) == 0u;
    if (__builtin_amdgcn_ballot_w64(fresh)) {
        const uint32_t hashed = gu_rng_word(prefix, GU_RNG_STREAM_SAMPLE, (t + 1u) >> GU_RNG_SAMPLE_LOG2);
        next = fresh ? hashed : next;
    }
    return next;
}
// the same where the caller knows at compile time whether step t + 1 starts a word (t the same in every lane)
template <bool FRESH>
__device__ __forceinline__ uint32_t gu_rng_sample_advance_at(uint32_t prefix, uint32_t t, uint32_t word)
{
    return FRESH ? gu_rng_word(prefix, GU_RNG_STREAM_SAMPLE, (t + 1u) >> GU_RNG_SAMPLE_LOG2) : gu_rng_sample_next(word);
}

// index into starts[] for episode `ep` (multiply-shift range reduction)
__host__ __device__ __forceinline__ uint32_t gu_rng_start_index(uint32_t prefix, uint32_t ep, uint32_t n_starts)
{
    return (uint32_t)(((uint64_t)gu_rng_word(prefix, GU_RNG_STREAM_START, ep) * (uint64_t)n_starts) >> 32);
}
