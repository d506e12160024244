// gu_vi.hpp -- device helpers shared by the tabular-DP kernels (gu_vi.hip, gu_vi_xcd.hip): the float64 restatement of
//   core/algorithms/utils.py:15-27   single_step_policy_evaluation        (V1)
//   core/algorithms/utils.py:55-72   greedy_policy_from_value_function    (V2)
// for ONE state, in the reference's operation order (see the header of gu_vi.hip).
#pragma once
#include "gu_internal.hpp"

typedef unsigned long long vi_u64;


struct ViMap {
    const uint8_t *f;  // flags plane (OPEN bits 0..3, TERM bit 4)
    const int8_t *r;   // reward plane
};

__device__ __forceinline__ int32_t vi_delta(uint32_t a, int32_t W)
{
    const int32_t sign = (int32_t)(a & 2u) - 1;
    return (a & 1u) ? -sign : sign * W;
}

__device__ __forceinline__ double vi_reward(const ViMap &m, int32_t s) { return (double)m.r[s]; }

__device__ __forceinline__ int32_t vi_next(int32_t s, uint32_t flags, uint32_t a, int32_t W)
{
    return ((flags >> a) & 1u) ? s + vi_delta(a, W) : s;
}

// V1 for one state
__device__ __forceinline__ double vi_eval_state(const ViMap &cell, int32_t W, double gamma, const double *__restrict__ v,
                                                const double *__restrict__ pi, int32_t s)
{
    const uint32_t rec = cell.f[s];
    double acc = __dadd_rn(0.0, vi_reward(cell, s));
#pragma unroll
    for (uint32_t a = 0; a < 4; ++a) {
        const int32_t n = vi_next(s, rec, a, W);
        acc = __dadd_rn(acc, __dmul_rn(pi[4 * s + a], __dmul_rn(gamma, v[n])));
    }
    return acc;
}

// Ties of np.around(q, 8) (utils.py:66-68).  around8(x) = rint(x * 1e8) / 1e8 and the division is a function of the
// integer k = rint(x * 1e8) alone, so equal k give equal around8.  Conversely, while |k| < 2^25 * 1e8 the quotients of
// two different integers are at least 1e-8 apart and an ulp there is at most 2^-27 < 1e-8, so they round to
// different float64: equality of around8 IS equality of k, and the five IEEE divisions (a third of the round's
// arithmetic) are only needed beyond |q| = 3.3e7 or for NaN.
__device__ __forceinline__ uint32_t vi_tie_mask(const double q[4], double qmax)
{
    double k[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) k[a] = rint(__dmul_rn(q[a], 100000000.0));
    const double kmax = rint(__dmul_rn(qmax, 100000000.0));
    const double lim = 3355443200000000.0;  // 2^25 * 1e8 (exactly representable, < 2^53)
    const bool small = fabs(k[0]) < lim && fabs(k[1]) < lim && fabs(k[2]) < lim && fabs(k[3]) < lim;  // false on NaN
    uint32_t mask = 0;
    if (small) {
#pragma unroll
        for (int a = 0; a < 4; ++a) mask |= (uint32_t)(k[a] == kmax) << a;
    } else {
        const double rmax = __ddiv_rn(kmax, 100000000.0);
#pragma unroll
        for (int a = 0; a < 4; ++a) mask |= (uint32_t)(__ddiv_rn(k[a], 100000000.0) == rmax) << a;
    }
    return mask;
}

__device__ __forceinline__ double vi_share(uint32_t mask)
{
    const int ties = __popc(mask);
    return (ties == 1) ? 1.0 : (ties == 2) ? 0.5 : (ties == 3) ? (1.0 / 3.0) : 0.25;
}

// V2 for one state given a functor returning v'(n): bit a of the result = action a ties for the maximum
// (0 for a terminal state: its row is all zeros, utils.py:62-63)
template <typename VNew>
__device__ __forceinline__ uint32_t vi_greedy_mask(const ViMap &cell, int32_t W, double gamma, VNew vnew, int32_t s)
{
    const uint32_t rec = cell.f[s];
    double q[4];
#pragma unroll
    for (uint32_t a = 0; a < 4; ++a) {
        const int32_t n = vi_next(s, rec, a, W);
        const double rn = vi_reward(cell, n);
        q[a] = __dadd_rn(0.0, __dadd_rn(rn, __dmul_rn(gamma, vnew(n))));
    }
    double qmax = q[0];
#pragma unroll
    for (int a = 1; a < 4; ++a) qmax = (q[a] > qmax) ? q[a] : qmax;
    const uint32_t mask = vi_tie_mask(q, qmax);
    return (rec & GU_CELL_TERM) ? 0u : mask;
}

template <typename VNew>
__device__ __forceinline__ void vi_greedy_state(const ViMap &cell, int32_t W, double gamma, VNew vnew, int32_t s, double out[4])
{
    const uint32_t mask = vi_greedy_mask(cell, W, gamma, vnew, s);
    const double share = vi_share(mask);
#pragma unroll
    for (int a = 0; a < 4; ++a) out[a] = ((mask >> a) & 1u) ? share : 0.0;
}

// order-preserving double -> uint64 key so that max(double) is an integer atomicMax
__device__ __forceinline__ unsigned long long vi_key(double x)
{
    unsigned long long b = (unsigned long long)__double_as_longlong(x);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

__device__ __forceinline__ double vi_unkey_dev(unsigned long long k)
{
    const unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    return __longlong_as_double((long long)b);
}

__device__ __forceinline__ double vi_ld_agent(const double *p)
{
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const vi_u64 *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

__device__ __forceinline__ void vi_st_agent(double *p, double x)
{
    __hip_atomic_store(reinterpret_cast<vi_u64 *>(p), (vi_u64)__double_as_longlong(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the grid planes of a DP kernel: staged in LDS (16-byte copies, then a workgroup barrier) or read from L2
template <bool LDS>
__device__ __forceinline__ ViMap vi_stage(const uint8_t *__restrict__ g, int32_t cell_bytes, uint8_t *smem)
{
    if (LDS) {
        for (int32_t i = threadIdx.x * 16; i < 2 * cell_bytes; i += blockDim.x * 16)
            *reinterpret_cast<uint4 *>(smem + i) = *reinterpret_cast<const uint4 *>(g + i);
        __syncthreads();
        return ViMap{smem, reinterpret_cast<const int8_t *>(smem + cell_bytes)};
    }
    return ViMap{g, reinterpret_cast<const int8_t *>(g + cell_bytes)};
}

// ---- arguments of the workgroup-cluster kernels (one launch for many rounds) ----
#define VI_CL_SPIN_LIMIT (1u << 22)

struct ViClusterArgs {
    const uint8_t *cell;
    int32_t cell_bytes, W, S;
    double gamma, threshold;
    double *v0, *v1;                 // double-buffered value table; v0 holds the current values at entry
    double *pi;                      // [S][4], updated in place when GREEDY and at least one round ran
    vi_u64 *delta_key;               // [max_rounds], zeroed before the launch
    uint32_t *sync;                  // [0] arrival counter, [1] timeout word; zeroed before the launch
    int32_t *rounds_done;
    int32_t max_rounds, use_threshold;
};


struct ViStepClusterArgs {
    ViClusterArgs vi;                // max_rounds = iters, use_threshold unused
    int32_t *pos, *reward, *done;
    uint32_t *episode;
    const int32_t *starts;
    uint32_t n_starts, seed_prefix, env_id0;
    int64_t N;
    uint32_t flags;
    uint64_t *done_bits;
};


// gu_vi_sweep_step_xcd_kernel (gu_vi_xcd.hip): the same rounds synchronised per XCD.  `vi.sync` = the launch header, zeroed
// before the launch: [0] workgroups registered, [1] fallback word, [2] rounds done, [3] 1 + XCC id of workgroup 0,
// [4 .. 11] workgroups registered per XCC.
struct ViStepXcdArgs : ViStepClusterArgs {
    vi_u64 *slots;            // [8 XCC][4: round & 3][64 members][2] tagged delta-key slots, zeroed before the launch
    uint8_t *gx;              // [8 XCC][work_bytes] the clusters' private granule buffers, zeroed before the launch:
                              //   [2 parities][S] value granules of 16 bytes | [2 parities][ceil(S / 32)] action items of 16 bytes
    uint32_t work_bytes;      // bytes per XCC
    uint32_t inject_failure;  // tests: every workgroup gives up at once
    uint32_t lds_values;      // doubles of a workgroup's value window in LDS (the launcher sets it from its plan)
    // Round 5: nothing has to be ZEROED in front of a launch.  Every tag carries the launch's number above the round (tag0 = epoch
    // << 13: up to 8190 rounds per launch, 524 287 launches before the host wipes the buffers once), so a word of an earlier launch
    // can never be taken for one of this launch -- the buffers are the engine's own, nothing else ever writes them -- and the header
    // is one of a ring of sixteen: workgroup 0 clears the NEXT launch's header when it is done.  tag0 == 0: as before (the host
    // zeroed everything; launches of more than 8190 rounds, the -DGU_VI_XCD_TORN variant with its 16-bit tags).
    uint32_t tag0;
    uint32_t *hdr_next;       // the header of the launch behind this one (16 words), or nullptr
    // ... and the tables alone leave their INPUT alone: the final values go to v_out, the final policy rows (GREEDY) to pi_out --
    // the other halves of the engine's double buffers --, which the host makes current only when the launch did not give up; no
    // snapshot, no restore.  nullptr: in place, as before (config 5: the envs' state is not double-buffered, its launch keeps a snapshot)
    double *v_out, *pi_out;
    // ... and, the tables alone, to a page-locked copy on the host as well (nullptr: no copy): gu_vi_get behind the call needs no launch
    double *v_host, *pi_host;
    // ... and so does config 5's fused launch since the engine keeps a second set of env-state arrays for it: the envs' final
    // positions, rewards, done flags, episode counters and done ballots go to these (nullptr: in place)
    int32_t *pos_out, *reward_out, *done_out;
    uint32_t *episode_out;
    uint64_t *done_bits_out;
};
