// gu_vi.hip -- tabular dynamic-programming sweeps on the engine's grid (config 5).
//
// Bit-exact float64 restatement of
//   core/algorithms/utils.py:15-27   single_step_policy_evaluation        (V1)
//   core/algorithms/utils.py:55-72   greedy_policy_from_value_function    (V2)
//   core/algorithms/dynamic_programming.py:15-20  one value-iteration round (V1, delta, V2)
// Every multiply and add is an explicit round-to-nearest f64 op (__dmul_rn/__dadd_rn):
// the reference rounds after each Python-level operation, so FMA contraction is forbidden
// (the library is also built with -ffp-contract=off).  Evaluation order is the reference's:
//   v'[s]  = ((((0.0 + R[s]) + p0*(g*v[n0])) + p1*(g*v[n1])) + p2*(g*v[n2])) + p3*(g*v[n3])
//   q[s,a] = 0.0 + (R[n_a] + g*v[n_a]);   ties: rint(q*1e8)/1e8 == rint(max*1e8)/1e8  (np.around(.,8))
//   pi[s,a]= 1/|ties| on ties, else 0; all zeros if s is terminal
// Walls and terminals are swept like any other state.
//
// One lane per state (S = 4096 at 64x64: latency-, not bandwidth-bound; v and pi live in
// L2).  gu_vi_sweep_step_kernel additionally moves every agent one step greedily on the
// policy of the SAME round; it gets pi'[pos] by re-evaluating v' at the four neighbour
// states ("pull" evaluation), which needs no grid-wide barrier inside the launch.
#include "gu_internal.hpp"
#include "gu_rng.hpp"
#include "gu_vi.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define VI_BLOCK 256
#define VI_RUN_BATCH 64  // rounds queued per host look of gu_vi_run (must not exceed the 4096 delta slots)

__device__ __forceinline__ void vi_block_max_to_global(double mine, bool valid, unsigned long long *out)
{
    __shared__ unsigned long long wave_key[VI_BLOCK / 64];
    unsigned long long k = valid ? vi_key(mine) : 0ull;
    for (int off = 32; off > 0; off >>= 1) {
        unsigned long long o = __shfl_down(k, off);
        k = o > k ? o : k;
    }
    if ((threadIdx.x & 63) == 0) wave_key[threadIdx.x >> 6] = k;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) k = wave_key[w] > k ? wave_key[w] : k;
        atomicMax(out, k);
    }
}

struct ViArgs {
    const uint8_t *cell;
    int32_t cell_bytes, W, S;
    double gamma;
    const double *v, *pi;     // old
    double *v_new, *pi_new;   // new
    unsigned long long *delta_key;
    // gu_vi_run: device-side stopping rule of value_iteration (dynamic_programming.py:22-23).  ctl[0] != 0 once a
    // round has met the threshold; prev_delta_key = the finished previous round's delta (null for the first round of
    // a batch).  Null ctl: the host drives the loop.
    int32_t *ctl;
    const unsigned long long *prev_delta_key;
    double threshold;
};

template <bool LDS>
__global__ void __launch_bounds__(VI_BLOCK) gu_vi_eval_kernel(const ViArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const ViMap cell = vi_stage<LDS>(a.cell, a.cell_bytes, smem);
    const int32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = s < a.S;
    double d = 0.0;
    if (valid) {
        const double vn = vi_eval_state(cell, a.W, a.gamma, a.v, a.pi, s);
        a.v_new[s] = vn;
        d = __dsub_rn(a.v[s], vn);  // signed, dynamic_programming.py:17
    }
    vi_block_max_to_global(d, valid, a.delta_key);
}

template <bool LDS>
__global__ void __launch_bounds__(VI_BLOCK) gu_vi_greedy_kernel(const ViArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const ViMap cell = vi_stage<LDS>(a.cell, a.cell_bytes, smem);
    const int32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= a.S) return;
    double row[4];
    const double *vn = a.v_new;
    vi_greedy_state(cell, a.W, a.gamma, [vn](int32_t n) { return vn[n]; }, s, row);
    *reinterpret_cast<double4 *>(a.pi_new + 4 * s) = make_double4(row[0], row[1], row[2], row[3]);
}

// One value-iteration round (V1, delta, V2: dynamic_programming.py:15-20) in ONE launch: the greedy update needs
// v' at the four neighbours, which each lane re-evaluates itself ("pull": 5 V1 evaluations per state instead of a
// grid-wide barrier between V1 and V2 -- the round is launch-latency-bound, not arithmetic-bound).  With ctl the
// launch first applies the stopping rule to the previous round's finished delta: every block reads the same two
// words, so either all of them run or none does.
template <bool LDS>
__global__ void __launch_bounds__(VI_BLOCK) gu_vi_round_kernel(const ViArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    if (a.ctl) {
        bool stop = a.ctl[0] != 0;
        if (!stop && a.prev_delta_key) stop = vi_unkey_dev(*a.prev_delta_key) < a.threshold;
        if (stop) {
            if (threadIdx.x == 0) a.ctl[0] = 1;  // read by the launches behind this one (same value from every block)
            return;
        }
    }
    const ViMap cell = vi_stage<LDS>(a.cell, a.cell_bytes, smem);
    const int32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = s < a.S;
    const int32_t W = a.W;
    const double gamma = a.gamma;
    const double *v = a.v, *pi = a.pi;
    auto vnew = [=](int32_t n) { return vi_eval_state(cell, W, gamma, v, pi, n); };
    double d = 0.0;
    if (valid) {
        const double vn = vnew(s);
        a.v_new[s] = vn;
        d = __dsub_rn(v[s], vn);  // signed, dynamic_programming.py:17
        double row[4];
        vi_greedy_state(cell, W, gamma, vnew, s, row);
        *reinterpret_cast<double4 *>(a.pi_new + 4 * s) = make_double4(row[0], row[1], row[2], row[3]);
    }
    vi_block_max_to_global(d, valid, a.delta_key);
}

// ------------------------------------------------------------------------------------
// Grids up to 4096 states (config 5's 64x64): the whole iteration in ONE launch of ONE 1024-thread workgroup.
// v is double-buffered in LDS next to the grid planes, each thread owns K = ceil(S / 1024) states and keeps their
// policy rows in registers, a round costs two workgroup barriers instead of a kernel boundary (~5 us: launch, staging
// the planes, L2 round trips), and the stopping rule of value_iteration / of policy_iteration's evaluation loop is a
// wave-uniform branch.  GREEDY: V1 + V2 per round (dynamic_programming.py:15-23); else V1 only with the policy fixed
// (:40-42).  Same operations in the same order as the per-round kernels above, hence the same bits.
// ------------------------------------------------------------------------------------
#define VI_PB_THREADS 1024
#define VI_PB_MAX_STATES 4096

struct ViBlockArgs {
    const uint8_t *cell;
    int32_t cell_bytes, W, S;
    double gamma, threshold;
    double *v, *pi;                  // updated in place when at least one round ran
    unsigned long long *delta_key;   // [max_rounds]
    int32_t *rounds_done;
    int32_t max_rounds, use_threshold;
};

template <int K, bool GREEDY>
__global__ void __launch_bounds__(VI_PB_THREADS) gu_vi_block_kernel(const ViBlockArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ unsigned long long wave_key[VI_PB_THREADS / 64];
    __shared__ unsigned long long round_key;
    const ViMap cell = vi_stage<true>(a.cell, a.cell_bytes, smem);
    double *vbuf = reinterpret_cast<double *>(smem + 2 * a.cell_bytes);  // [2][S]
    const int32_t tid = threadIdx.x, S = a.S, W = a.W;
    const double gamma = a.gamma;
    double p[K][4];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const int32_t s = tid + j * VI_PB_THREADS;
        if (s < S) {
            vbuf[s] = a.v[s];
            const double4 row = *reinterpret_cast<const double4 *>(a.pi + 4 * (int64_t)s);
            p[j][0] = row.x, p[j][1] = row.y, p[j][2] = row.z, p[j][3] = row.w;
        }
    }
    __syncthreads();
    int cur = 0, r = 0;
    for (; r < a.max_rounds; ++r) {
        const double *vo = vbuf + cur * S;
        double *vn = vbuf + (cur ^ 1) * S;
        unsigned long long key = 0ull;
#pragma unroll
        for (int j = 0; j < K; ++j) {  // V1 (utils.py:15-27)
            const int32_t s = tid + j * VI_PB_THREADS;
            if (s < S) {
                const uint32_t rec = cell.f[s];
                double acc = __dadd_rn(0.0, vi_reward(cell, s));
#pragma unroll
                for (uint32_t act = 0; act < 4; ++act)
                    acc = __dadd_rn(acc, __dmul_rn(p[j][act], __dmul_rn(gamma, vo[vi_next(s, rec, act, W)])));
                vn[s] = acc;
                const unsigned long long k = vi_key(__dsub_rn(vo[s], acc));  // signed, dynamic_programming.py:17
                key = k > key ? k : key;
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const unsigned long long o = __shfl_down(key, off);
            key = o > key ? o : key;
        }
        if ((tid & 63) == 0) wave_key[tid >> 6] = key;
        __syncthreads();  // v' complete, wave maxima posted
        if (tid < 64) {
            unsigned long long k = tid < VI_PB_THREADS / 64 ? wave_key[tid] : 0ull;
            for (int off = 8; off > 0; off >>= 1) {
                const unsigned long long o = __shfl_down(k, off);
                k = o > k ? o : k;
            }
            if (tid == 0) {
                round_key = k;
                a.delta_key[r] = k;
            }
        }
        if (GREEDY) {
#pragma unroll
            for (int j = 0; j < K; ++j) {  // V2 (utils.py:55-72) on v'
                const int32_t s = tid + j * VI_PB_THREADS;
                if (s < S) vi_greedy_state(cell, W, gamma, [vn](int32_t n) { return vn[n]; }, s, p[j]);
            }
        }
        __syncthreads();  // round_key visible; nobody still reads the old v
        cur ^= 1;
        if (a.use_threshold && vi_unkey_dev(round_key) < a.threshold) {
            ++r;
            break;
        }
    }
    if (r > 0) {
        const double *vf = vbuf + cur * S;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int32_t s = tid + j * VI_PB_THREADS;
            if (s < S) {
                a.v[s] = vf[s];
                if (GREEDY) *reinterpret_cast<double4 *>(a.pi + 4 * (int64_t)s) = make_double4(p[j][0], p[j][1], p[j][2], p[j][3]);
            }
        }
    }
    if (tid == 0) *a.rounds_done = r;
}

// ------------------------------------------------------------------------------------
// Grids of 4097 .. 524 288 states (e.g. the shipped 101x101 level): the whole iteration in ONE launch of a CLUSTER of
// G <= 256 workgroups of 1024 threads, K = 1 or 2 states per thread, one grid-wide barrier per round.  Against one launch per
// round (gu_vi_round_kernel) a round no longer pays a kernel boundary, the staging of the planes and the five-fold "pull"
// evaluation: each thread reads its own record and its neighbours' rewards ONCE, keeps its policy row in registers, and a round is  V1 (4 loads of v) -> store v' -> barrier -> V2 (4 loads of v') -- the same float64
// operations in the same order as every other DP kernel here, hence the same bits.
//
// Inter-workgroup protocol (cdna_hip_programming.md Guideline 16, MI355X_MICROARCH.md "Valid forms", first table row):
// every byte that crosses workgroups -- v, the per-round delta keys, the arrival counter -- is written and read with
// 8-byte / 4-byte AGENT-scope atomics (write-through `sc1` stores, `sc1` loads that bypass the per-CU L1; the per-XCD L2s
// are not coherent with each other); after its store every wave drains (`s_waitcnt vmcnt(0)`), the workgroup barriers, ONE
// lane adds to the monotonic arrival counter and polls it (relaxed, `s_sleep`) until all G workgroups of the round have
// arrived, and the other waves continue behind a second workgroup barrier.  v is double-buffered, so one barrier per round
// separates every write of a buffer from every read of it.  All G workgroups must be resident together: at most one
// 1024-thread workgroup per CU of the device; every spin is bounded and raises a timeout word instead of hanging.
// ------------------------------------------------------------------------------------
#define VI_CL_THREADS 1024
#define VI_CL_MAX_WGS 256
#define VI_CL_MAX_K 2  /* states per thread: 4 would spill at the 128 registers of a 1024-thread workgroup */
#define VI_CL_MAX_STATES (VI_CL_MAX_WGS * VI_CL_THREADS * VI_CL_MAX_K)



template <int K, bool GREEDY>
__global__ void __launch_bounds__(VI_CL_THREADS) gu_vi_cluster_kernel(const ViClusterArgs a)
{
    __shared__ vi_u64 wave_key[VI_CL_THREADS / 64];
    __shared__ vi_u64 round_key;
    __shared__ uint32_t timed_out;
    const int32_t tid = threadIdx.x, S = a.S, W = a.W;
    const int32_t stride = gridDim.x * VI_CL_THREADS;  // thread owns states s0, s0 + stride, ... (K of them)
    const int32_t s0 = blockIdx.x * VI_CL_THREADS + tid;
    const double gamma = a.gamma;
    const ViMap cell{a.cell, reinterpret_cast<const int8_t *>(a.cell + a.cell_bytes)};
    // per-state constants of every round: neighbour indices (env:136-155 through the OPEN bits), rewards, terminal bit
    // (rewards are kept as packed int8 and converted where used, neighbour indices are recomputed: registers, not time, are scarce)
    uint32_t rec[K];  // the state's own record: OPEN bits (-> neighbour indices, recomputed where used) and the TERM bit
    uint32_t rn[K];   // reward_matrix of the four neighbours, one int8 each
    int32_t r_own[K];
    double p[K][4];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const int32_t s = s0 + j * stride;
        rec[j] = 0u;
        r_own[j] = 0;
        rn[j] = 0u;
#pragma unroll
        for (int act = 0; act < 4; ++act) p[j][act] = 0.0;
        if (s < S) {
            rec[j] = cell.f[s];
            r_own[j] = cell.r[s];
#pragma unroll
            for (uint32_t act = 0; act < 4; ++act) rn[j] |= (uint32_t)(uint8_t)cell.r[vi_next(s, rec[j], act, W)] << (8 * act);
            const double4 row = *reinterpret_cast<const double4 *>(a.pi + 4 * (int64_t)s);
            p[j][0] = row.x, p[j][1] = row.y, p[j][2] = row.z, p[j][3] = row.w;
        }
    }
    if (tid == 0) timed_out = 0u;
    __syncthreads();
    double *vo = a.v0, *vn = a.v1;
    const uint32_t G = gridDim.x;
    int r = 0;
    bool failed = false;
    for (; r < a.max_rounds; ++r) {
        vi_u64 key = 0ull;
#pragma unroll
        for (int j = 0; j < K; ++j) {  // V1 (utils.py:15-27), reading the old table through the L1-bypassing loads
            const int32_t s = s0 + j * stride;
            if (s < S) {
                double acc = __dadd_rn(0.0, (double)r_own[j]);
#pragma unroll
                for (uint32_t act = 0; act < 4; ++act)
                    acc = __dadd_rn(acc, __dmul_rn(p[j][act], __dmul_rn(gamma, vi_ld_agent(vo + vi_next(s, rec[j], act, W)))));
                vi_st_agent(vn + s, acc);
                const vi_u64 k = vi_key(__dsub_rn(vi_ld_agent(vo + s), acc));  // signed, dynamic_programming.py:17
                key = k > key ? k : key;
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const vi_u64 o = __shfl_down(key, off);
            key = o > key ? o : key;
        }
        if ((tid & 63) == 0) wave_key[tid >> 6] = key;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its write-through stores ...
        __syncthreads();                                   // ... before the one lane that signals for the workgroup
        if (tid == 0) {
            vi_u64 k = wave_key[0];
            for (int w = 1; w < VI_CL_THREADS / 64; ++w) k = wave_key[w] > k ? wave_key[w] : k;
            __hip_atomic_fetch_max(a.delta_key + r, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(a.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t want = G * (uint32_t)(r + 1);
            uint32_t spins = 0;
            while (__hip_atomic_load(a.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > VI_CL_SPIN_LIMIT || __hip_atomic_load(a.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(a.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // tell everyone; never hang
                    timed_out = 1u;
                    break;
                }
            }
            if (__hip_atomic_load(a.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) timed_out = 1u;  // (someone else gave up)
            round_key = __hip_atomic_load(a.delta_key + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();  // the other waves load only behind the polling lane's match
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // (compiler ordering only: every shared load is an sc1 load)
        if (timed_out) {
            failed = true;
            break;
        }
        if (GREEDY) {
#pragma unroll
            for (int j = 0; j < K; ++j) {  // V2 (utils.py:55-72) on v'
                const int32_t s = s0 + j * stride;
                if (s < S) {
                    double q[4];
#pragma unroll
                    for (uint32_t act = 0; act < 4; ++act)
                        q[act] = __dadd_rn(0.0, __dadd_rn((double)(int8_t)(rn[j] >> (8 * act)),
                                                          __dmul_rn(gamma, vi_ld_agent(vn + vi_next(s, rec[j], act, W)))));
                    double qmax = q[0];
#pragma unroll
                    for (int act = 1; act < 4; ++act) qmax = (q[act] > qmax) ? q[act] : qmax;
                    const uint32_t mask = (rec[j] & GU_CELL_TERM) ? 0u : vi_tie_mask(q, qmax);
                    const double share = vi_share(mask);
#pragma unroll
                    for (int act = 0; act < 4; ++act) p[j][act] = ((mask >> act) & 1u) ? share : 0.0;
                }
            }
        }
        double *t = vo;
        vo = vn;
        vn = t;
        if (a.use_threshold && vi_unkey_dev(round_key) < a.threshold) {
            ++r;
            break;
        }
    }
    if (GREEDY && r > 0 && !failed) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int32_t s = s0 + j * stride;
            if (s < S) *reinterpret_cast<double4 *>(a.pi + 4 * (int64_t)s) = make_double4(p[j][0], p[j][1], p[j][2], p[j][3]);
        }
    }
    if (blockIdx.x == 0 && tid == 0) *a.rounds_done = failed ? -1 : r;
}

// ------------------------------------------------------------------------------------
// Config 5 for MANY rounds in ONE launch: `iters` x { V1 + V2 sweep of the table; every agent takes one greedy step on the
// updated policy } -- the cluster kernel above with the agents riding along.  One thread owns (at most) one state AND one
// env; a round is   V1 of the own state -> store v' -> grid barrier -> V2 of the own state (policy row in registers) and,
// for the own env, lazy reset + greedy action from v' at the four neighbours of its position (the same q / tie mask /
// first-argmax as gu_vi_sweep_step_kernel, which re-evaluates v' by "pull": the stored v' are those very values) + move.
// Against one gu_vi_sweep_step launch per round a round costs the barrier (~3 us) instead of a kernel boundary plus the
// pull evaluations (7.4 us per launch at config 5).  Same inter-workgroup protocol as gu_vi_cluster_kernel.
// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(VI_CL_THREADS) gu_vi_sweep_step_cluster_kernel(const ViStepClusterArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ vi_u64 wave_key[VI_CL_THREADS / 64];
    __shared__ uint32_t timed_out;
    const ViMap cell = vi_stage<true>(a.vi.cell, a.vi.cell_bytes, smem);  // the agents gather records of arbitrary cells
    const int32_t tid = threadIdx.x, S = a.vi.S, W = a.vi.W;
    const int64_t gid = (int64_t)blockIdx.x * VI_CL_THREADS + tid;
    const bool own_state = gid < S, own_env = gid < a.N;
    const int32_t s = (int32_t)gid;
    const double gamma = a.vi.gamma;
    uint32_t rec = 0u, rn = 0u;
    int32_t r_own = 0;
    double p[4] = {0.0, 0.0, 0.0, 0.0};
    if (own_state) {
        rec = cell.f[s];
        r_own = cell.r[s];
#pragma unroll
        for (uint32_t act = 0; act < 4; ++act) rn |= (uint32_t)(uint8_t)cell.r[vi_next(s, rec, act, W)] << (8 * act);
        const double4 row = *reinterpret_cast<const double4 *>(a.vi.pi + 4 * (int64_t)s);
        p[0] = row.x, p[1] = row.y, p[2] = row.z, p[3] = row.w;
    }
    int32_t e_pos = 0, e_rew = 0, e_done = 0;
    uint32_t e_ep = 0, e_prefix = 0;
    if (own_env) {
        e_pos = a.pos[gid];
        e_rew = a.reward[gid];
        e_done = a.done[gid];
        e_ep = a.episode[gid];
        e_prefix = gu_rng_prefix(a.seed_prefix, a.env_id0 + (uint32_t)gid);
    }
    if (tid == 0) timed_out = 0u;
    __syncthreads();
    double *vo = a.vi.v0, *vn = a.vi.v1;
    const uint32_t G = gridDim.x;
    int r = 0;
    bool failed = false;
    for (; r < a.vi.max_rounds; ++r) {
        vi_u64 key = 0ull;
        if (own_state) {  // V1 (utils.py:15-27)
            double acc = __dadd_rn(0.0, (double)r_own);
#pragma unroll
            for (uint32_t act = 0; act < 4; ++act)
                acc = __dadd_rn(acc, __dmul_rn(p[act], __dmul_rn(gamma, vi_ld_agent(vo + vi_next(s, rec, act, W)))));
            vi_st_agent(vn + s, acc);
            key = vi_key(__dsub_rn(vi_ld_agent(vo + s), acc));
        }
        for (int off = 32; off > 0; off >>= 1) {
            const vi_u64 o = __shfl_down(key, off);
            key = o > key ? o : key;
        }
        if ((tid & 63) == 0) wave_key[tid >> 6] = key;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            vi_u64 k = wave_key[0];
            for (int w = 1; w < VI_CL_THREADS / 64; ++w) k = wave_key[w] > k ? wave_key[w] : k;
            if (k) __hip_atomic_fetch_max(a.vi.delta_key + r, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(a.vi.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t want = G * (uint32_t)(r + 1);
            uint32_t spins = 0;
            while (__hip_atomic_load(a.vi.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > VI_CL_SPIN_LIMIT || __hip_atomic_load(a.vi.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(a.vi.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    timed_out = 1u;
                    break;
                }
            }
            if (__hip_atomic_load(a.vi.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) timed_out = 1u;  // (someone else gave up)
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (timed_out) {
            failed = true;
            break;
        }
        if (own_state) {  // V2 (utils.py:55-72) on v'
            double q[4];
#pragma unroll
            for (uint32_t act = 0; act < 4; ++act)
                q[act] = __dadd_rn(0.0, __dadd_rn((double)(int8_t)(rn >> (8 * act)), __dmul_rn(gamma, vi_ld_agent(vn + vi_next(s, rec, act, W)))));
            double qmax = q[0];
#pragma unroll
            for (int act = 1; act < 4; ++act) qmax = (q[act] > qmax) ? q[act] : qmax;
            const uint32_t mask = (rec & GU_CELL_TERM) ? 0u : vi_tie_mask(q, qmax);
            const double share = vi_share(mask);
#pragma unroll
            for (int act = 0; act < 4; ++act) p[act] = ((mask >> act) & 1u) ? share : 0.0;
        }
        if (own_env) {  // the agent acts greedily on the policy of THIS round (np.argmax of its row: first maximum)
            if ((a.flags & GU_F_AUTO_RESET) && e_done) {  // lazy `if done: env.reset()`
                e_pos = a.starts[gu_rng_start_index(e_prefix, e_ep, a.n_starts)];
                ++e_ep;
            }
            double *vnew = vn;
            const uint32_t mask = vi_greedy_mask(cell, W, gamma, [vnew](int32_t n) { return vi_ld_agent(vnew + n); }, e_pos);
            const uint32_t act = mask ? (uint32_t)__ffs((int)mask) - 1u : 0u;  // an all-zero row (terminal state): argmax = 0
            e_pos = vi_next(e_pos, cell.f[e_pos], act, W);
            e_rew = cell.r[e_pos];
            e_done = (cell.f[e_pos] >> GU_CELL_TERM_BIT) & 1;
        }
        double *t = vo;
        vo = vn;
        vn = t;
    }
    if (own_state && r > 0 && !failed) *reinterpret_cast<double4 *>(a.vi.pi + 4 * (int64_t)s) = make_double4(p[0], p[1], p[2], p[3]);
    if (own_env) {
        a.pos[gid] = e_pos;
        a.reward[gid] = e_rew;
        a.done[gid] = e_done;
        a.episode[gid] = e_ep;
    }
    const uint64_t bits = __ballot(own_env && e_done != 0);
    if ((tid & 63) == 0 && own_env) a.done_bits[gid >> 6] = bits;
    if (blockIdx.x == 0 && tid == 0) *a.vi.rounds_done = failed ? -1 : r;
}

// first-argmax action per state (np.argmax; examples/griduniverse_alg_examples.py:76)
__global__ void __launch_bounds__(VI_BLOCK) gu_vi_argmax_kernel(const double *__restrict__ pi, int32_t S, uint8_t *__restrict__ greedy)
{
    const int32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const double4 p = *reinterpret_cast<const double4 *>(pi + 4 * s);
    uint32_t best = 0;
    double m = p.x;
    if (p.y > m) { m = p.y; best = 1; }
    if (p.z > m) { m = p.z; best = 2; }
    if (p.w > m) { m = p.w; best = 3; }
    greedy[s] = (uint8_t)best;
}

// config 5: one V1+V2 round AND one greedy env step in one launch
struct ViStepArgs {
    ViArgs vi;
    int32_t *pos, *reward, *done;
    uint32_t *episode;
    const int32_t *starts;
    uint32_t n_starts, seed_prefix, env_id0;
    int64_t N;
    uint32_t flags;
    uint64_t *done_bits;  // [ceil(N/64)] wave ballots of the new done flags (episode-done compaction)
};

template <bool LDS>
__global__ void __launch_bounds__(VI_BLOCK) gu_vi_sweep_step_kernel(const ViStepArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const ViMap cell = vi_stage<LDS>(a.vi.cell, a.vi.cell_bytes, smem);
    const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t W = a.vi.W;
    const double gamma = a.vi.gamma;
    const double *v = a.vi.v, *pi = a.vi.pi;
    auto vnew = [=](int32_t n) { return vi_eval_state(cell, W, gamma, v, pi, n); };

    // (1) table round for state gid
    const bool sweep = gid < a.vi.S;
    double d = 0.0;
    if (sweep) {
        const int32_t s = (int32_t)gid;
        const double vn = vnew(s);
        a.vi.v_new[s] = vn;
        d = __dsub_rn(v[s], vn);
        double row[4];
        vi_greedy_state(cell, W, gamma, vnew, s, row);
        *reinterpret_cast<double4 *>(a.vi.pi_new + 4 * s) = make_double4(row[0], row[1], row[2], row[3]);
    }
    if ((int64_t)blockIdx.x * blockDim.x < (int64_t)a.vi.S) vi_block_max_to_global(d, sweep, a.vi.delta_key);  // block-uniform branch

    // (2) agent gid acts greedily on the updated policy
    int32_t dn = 0;
    if (gid < a.N) {
        int32_t s = a.pos[gid];
        if ((a.flags & GU_F_AUTO_RESET) && a.done[gid]) {
            const uint32_t ep = a.episode[gid];
            s = a.starts[gu_rng_start_index(gu_rng_prefix(a.seed_prefix, a.env_id0 + (uint32_t)gid), ep, a.n_starts)];
            a.episode[gid] = ep + 1;
        }
        double row[4];
        vi_greedy_state(cell, W, gamma, vnew, s, row);
        uint32_t act = 0;
        double m = row[0];
#pragma unroll
        for (uint32_t k = 1; k < 4; ++k)
            if (row[k] > m) { m = row[k]; act = k; }
        s = vi_next(s, cell.f[s], act, W);
        dn = (cell.f[s] >> GU_CELL_TERM_BIT) & 1;
        a.pos[gid] = s;
        a.reward[gid] = cell.r[s];
        a.done[gid] = dn;
    }
    const uint64_t bits = __ballot(dn != 0);
    if ((threadIdx.x & 63) == 0 && gid < a.N) a.done_bits[gid >> 6] = bits;
}

// ------------------------------------------------------------------------------------ host side
static inline unsigned vi_blocks(int64_t n) { return (unsigned)((n + VI_BLOCK - 1) / VI_BLOCK); }

int gu_vi_alloc(gu_engine *h)
{
    if (h->d_v[0]) return GU_OK;
    const size_t S = (size_t)h->S;
    for (int k = 0; k < 2; ++k) {
        GU_HIP(hipMalloc(&h->d_v[k], S * sizeof(double)));
        GU_HIP(hipMalloc(&h->d_pi[k], 4 * S * sizeof(double)));
    }
    GU_HIP(hipMalloc(&h->d_delta, 4096 * sizeof(unsigned long long)));
    GU_HIP(hipMalloc(&h->d_pi_thr, S * sizeof(uint4)));
    return GU_OK;
}

void gu_vi_free(gu_engine *h)
{
    for (int k = 0; k < 2; ++k) {
        if (h->d_v[k]) (void)hipFree(h->d_v[k]);
        if (h->d_pi[k]) (void)hipFree(h->d_pi[k]);
        h->d_v[k] = h->d_pi[k] = nullptr;
    }
    if (h->d_pi_thr) (void)hipFree(h->d_pi_thr);
    h->d_pi_thr = nullptr;
    if (h->d_delta) (void)hipFree(h->d_delta);
    h->d_delta = nullptr;
    gu_vi_xcd_free(h);
    h->has_vi = false;
    h->greedy_valid = false;
}

int gu_launch_greedy_table(gu_engine *h)
{
    hipLaunchKernelGGL(gu_vi_argmax_kernel, dim3(vi_blocks(h->S)), dim3(VI_BLOCK), 0, h->stream, h->d_pi[h->vi_cur], h->S, h->d_greedy);
    GU_HIP(hipGetLastError());
    h->greedy_valid = true;
    return GU_OK;
}

static double vi_unkey(unsigned long long k)
{
    unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    double x;
    memcpy(&x, &b, sizeof x);
    return x;
}

static ViArgs vi_args(gu_engine *h, double gamma, unsigned long long *delta_key)
{
    ViArgs a{};
    a.cell = h->d_cell;
    a.cell_bytes = h->cell_bytes;
    a.W = h->W;
    a.S = h->S;
    a.gamma = gamma;
    a.v = h->d_v[h->vi_cur];
    a.pi = h->d_pi[h->vi_cur];
    a.v_new = h->d_v[h->vi_cur ^ 1];
    a.pi_new = h->d_pi[h->vi_cur ^ 1];
    a.delta_key = delta_key;
    a.ctl = nullptr;
    a.prev_delta_key = nullptr;
    a.threshold = 0.0;
    return a;
}

// The single-workgroup path: grids of up to VI_PB_MAX_STATES states (GU_OPT_VI_PATH = 2 forces the per-round
// launches, for the tests).  Runs up to max_rounds rounds in place on the current tables.
// The per-XCD launch (gu_vi_xcd.hip) runs a round in ~1.06 us WHATEVER the grid's size (1.02 at 8x8, 1.06 at 64x64), against 1.35
// (8x8) .. 1.9 (32x32) .. 5 us (64x64) in one workgroup and ~4 us behind the chip-wide barrier, and since its snapshot and its
// zeroed buffers are ONE launch and its results one copy back, a call of it is no dearer either: 39 us for one round, 145 for a
// hundred at 32x32, against 46 and 240 of the single-workgroup kernel (profiles/archive/r04z_dp_calls.txt).  It is taken first on every
// grid it fits; the other paths are what it falls back to (a grid that does not fit a workgroup's LDS, workgroups that cannot all
// be resident, GU_OPT_VI_PATH).
static bool vi_xcd_preferred(const gu_engine *h, int32_t rounds)
{
    (void)rounds;
    const int64_t path = gu_opt(h, GU_OPT_VI_PATH);
    return path == 0 || path == 5 || path == 6;
}

static bool vi_block_eligible(const gu_engine *h)
{
    return h->S <= VI_PB_MAX_STATES && gu_opt(h, GU_OPT_VI_PATH) != 2;
}

static int vi_block_run(gu_engine *h, double gamma, double threshold, bool use_threshold, bool greedy, int32_t max_rounds,
                        int32_t *rounds_done, double *deltas)
{
    *rounds_done = 0;
    if (max_rounds <= 0) return GU_OK;
    const size_t key_bytes = (size_t)max_rounds * sizeof(unsigned long long);
    int rc = gu_ensure_scratch(h, key_bytes + 16);
    if (rc != GU_OK) return rc;
    unsigned long long *keys_d = (unsigned long long *)h->d_scratch;
    int32_t *done_d = (int32_t *)((char *)h->d_scratch + key_bytes);
    ViBlockArgs a{h->d_cell, h->cell_bytes, h->W, h->S, gamma, threshold, h->d_v[h->vi_cur], h->d_pi[h->vi_cur], keys_d, done_d,
                  max_rounds, use_threshold ? 1 : 0};
    const size_t smem = 2 * (size_t)h->cell_bytes + 2 * (size_t)h->S * sizeof(double);
    const int K = (h->S + VI_PB_THREADS - 1) / VI_PB_THREADS;
    void (*kern)(const ViBlockArgs) = nullptr;
    switch (K * 2 + (greedy ? 1 : 0)) {
    case 2: kern = gu_vi_block_kernel<1, false>; break;
    case 3: kern = gu_vi_block_kernel<1, true>; break;
    case 4: kern = gu_vi_block_kernel<2, false>; break;
    case 5: kern = gu_vi_block_kernel<2, true>; break;
    case 6: kern = gu_vi_block_kernel<3, false>; break;
    case 7: kern = gu_vi_block_kernel<3, true>; break;
    case 8: kern = gu_vi_block_kernel<4, false>; break;
    default: kern = gu_vi_block_kernel<4, true>; break;
    }
    if (smem > 64 * 1024) GU_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL(kern, dim3(1), dim3(VI_PB_THREADS), smem, h->stream, a);
    GU_HIP(hipGetLastError());
    int32_t done = 0;
    {
        const int rb = gu_read_back(h, &done, done_d, sizeof done);
        if (rb != GU_OK) return rb;
    }
    if (deltas && done > 0) {
        std::vector<unsigned long long> keys((size_t)done);
        GU_HIP(hipMemcpy(keys.data(), keys_d, (size_t)done * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (int32_t i = 0; i < done; ++i) deltas[i] = vi_unkey(keys[(size_t)i]);
    }
    *rounds_done = done;
    h->greedy_valid = false;
    return GU_OK;
}

// The cluster path: 4097 .. 524 288 states, one launch for the whole loop (GU_OPT_VI_PATH = 1 or 2 sends these grids down the
// one-launch-per-round path instead, for the tests and A/B runs; so does a device with too few CUs for the workgroups to be
// resident together -- a CPX partition, a CU mask, a smaller part).  Runs up to max_rounds rounds; the value table ends in
// d_v[vi_cur] (buffers swapped on an odd round count), the policy is updated in place.

static bool vi_cluster_shape(const gu_engine *h, int *K_out, unsigned *G_out)
{
    if (h->S <= VI_PB_MAX_STATES || h->S > VI_CL_MAX_STATES) return false;
    const int64_t wgs_needed = ((int64_t)h->S + VI_CL_THREADS - 1) / VI_CL_THREADS;
    const int max_wgs = h->n_cu < VI_CL_MAX_WGS ? h->n_cu : VI_CL_MAX_WGS;  // one workgroup per CU: all resident together
    int K = 1;
    while (K < VI_CL_MAX_K && wgs_needed > (int64_t)K * max_wgs) K <<= 1;
    const int64_t G = (wgs_needed + K - 1) / K;
    if (G > max_wgs) return false;
    if (K_out) *K_out = K;
    if (G_out) *G_out = (unsigned)G;
    return true;
}

static bool vi_cluster_eligible(const gu_engine *h)
{
    const int64_t path = gu_opt(h, GU_OPT_VI_PATH);
    // (4 = the chip-wide cluster where the per-XCD form would apply, 5 = the per-XCD form made to give up: both end on the cluster
    // here, as in gu_vi_sweep_step_run)
    return (path == 0 || path == 3 || path == 4 || path == 5 || path == 6) && vi_cluster_shape(h, nullptr, nullptr);
}

static int vi_cluster_run(gu_engine *h, double gamma, double threshold, bool use_threshold, bool greedy, int32_t max_rounds,
                          int32_t *rounds_done, double *deltas)
{
    *rounds_done = 0;
    if (max_rounds <= 0) return GU_OK;
    // scratch: [sync counter, timeout word, rounds_done, pad] (16 B) | delta keys [max_rounds] | snapshot of v and pi
    const size_t key_bytes = (size_t)max_rounds * sizeof(unsigned long long);
    const size_t snap_off = (16 + key_bytes + 15) & ~(size_t)15;
    const size_t v_bytes = (size_t)h->S * sizeof(double);
    const size_t total = snap_off + 5 * v_bytes;
    int rc = gu_ensure_scratch(h, total);
    if (rc != GU_OK) return rc;
    uint32_t *sync_d = (uint32_t *)h->d_scratch;
    int32_t *done_d = (int32_t *)h->d_scratch + 2;
    unsigned long long *keys_d = (unsigned long long *)((char *)h->d_scratch + 16);
    char *snap = (char *)h->d_scratch + snap_off;
    GU_HIP(hipMemsetAsync(h->d_scratch, 0, snap_off, h->stream));  // every polled word is zeroed before every launch
    // A barrier that times out (the workgroups were not resident together: another process on the device, a CU mask) leaves v
    // written halfway through a round and the policy rows of some states updated: the launch works on a snapshot's ORIGINAL,
    // and a timeout restores it (40 bytes per state, device to device) and hands the loop to the one-launch-per-round path.
    if ((rc = gu_device_copy(h, snap, h->d_v[h->vi_cur], v_bytes)) != GU_OK) return rc;
    if ((rc = gu_device_copy(h, snap + v_bytes, h->d_pi[h->vi_cur], 4 * v_bytes)) != GU_OK) return rc;
    if (gu_opt(h, GU_OPT_VI_PATH) == 3) GU_HIP(hipMemsetD32Async((hipDeviceptr_t)(sync_d + 1), 1, 1, h->stream));  // tests: an injected timeout
    ViClusterArgs a{h->d_cell, h->cell_bytes, h->W, h->S, gamma, threshold, h->d_v[h->vi_cur], h->d_v[h->vi_cur ^ 1],
                    h->d_pi[h->vi_cur], keys_d, sync_d, done_d, max_rounds, use_threshold ? 1 : 0};
    int K = 1;
    unsigned G = 0;
    GU_REQUIRE(vi_cluster_shape(h, &K, &G), GU_ERR_UNSUPPORTED, "grid of %d states does not fit a workgroup cluster on %d CUs", h->S, h->n_cu);
    void (*kern)(const ViClusterArgs) = nullptr;
    switch (K * 2 + (greedy ? 1 : 0)) {
    case 2: kern = gu_vi_cluster_kernel<1, false>; break;
    case 3: kern = gu_vi_cluster_kernel<1, true>; break;
    case 4: kern = gu_vi_cluster_kernel<2, false>; break;
    default: kern = gu_vi_cluster_kernel<2, true>; break;
    }
    hipLaunchKernelGGL(kern, dim3(G), dim3(VI_CL_THREADS), 0, h->stream, a);
    GU_HIP(hipGetLastError());
    // rounds_done is written by ONE workgroup, and a timeout need not be unanimous (the workgroup that arrives last finds the
    // counter complete and goes on while the others have given up): the timeout word, raised by whoever gives up, decides.
    int32_t ctl[4] = {0, 0, 0, 0};  // [arrival counter, timeout word, rounds_done, pad]
    {
        const int rb = gu_read_back(h, ctl, h->d_scratch, sizeof ctl);
        if (rb != GU_OK) return rb;
    }
    const int32_t done = ctl[1] ? -1 : ctl[2];
    if (done < 0) {
        if ((rc = gu_device_copy(h, h->d_v[h->vi_cur], snap, v_bytes)) != GU_OK) return rc;
        if ((rc = gu_device_copy(h, h->d_pi[h->vi_cur], snap + v_bytes, 4 * v_bytes)) != GU_OK) return rc;
        GU_HIP(hipStreamSynchronize(h->stream));
        if (gu_debug()) fprintf(stderr, "[gu] DP cluster kernel: grid barrier timed out (%u workgroups not resident together); tables restored, one launch per round instead\n", G);
        return GU_VI_FALLBACK;
    }
    if (deltas && done > 0) {
        std::vector<unsigned long long> keys((size_t)done);
        GU_HIP(hipMemcpy(keys.data(), keys_d, (size_t)done * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        for (int32_t i = 0; i < done; ++i) deltas[i] = vi_unkey(keys[(size_t)i]);
    }
    if (done & 1) {  // the value table ended in the other buffer; the policy stayed where it was
        double *t = h->d_v[0];
        h->d_v[0] = h->d_v[1];
        h->d_v[1] = t;
    }
    *rounds_done = done;
    h->greedy_valid = false;
    return GU_OK;
}

extern "C" {

int gu_vi_set(gu_handle h, const double *v, const double *pi)
{
    int rc = gu_use_device(h);
    if (rc != GU_OK) return rc;
    h->h_tables_valid = false;  // (the call may write the tables: the host's copy of them, gu_vi_xcd.hip, is withdrawn; the per-XCD launch of the tables alone renews it)
    GU_REQUIRE(h->has_grid, GU_ERR_STATE, "no grid set: call gu_set_grid first");
    GU_REQUIRE(h->n_grids == 1, GU_ERR_UNSUPPORTED, "value / policy tables need a single-grid engine");
    GU_REQUIRE(v && pi, GU_ERR_INVALID, "v or pi is NULL");
    rc = gu_vi_alloc(h);
    if (rc != GU_OK) return rc;
    const size_t vb = (size_t)h->S * sizeof(double);
    if (h->h_up && 5 * vb <= GU_UP_BYTES) {
        // through the page-locked staging area: two DMAs the stream orders in front of whatever is launched next, no wait (36 -> ~8 us
        // per call; the caller's arrays are free the moment this returns).  The wait is for an upload of an earlier call that may
        // still be reading the area.
        GU_HIP(hipStreamSynchronize(h->stream));
        memcpy(h->h_up, v, vb);
        memcpy(h->h_up + vb, pi, 4 * vb);
        GuSegments up;  // ONE launch reads both tables out of the page-locked area (it is mapped into the device's address space)
        up.add(h->d_v[h->vi_cur], h->h_up, vb);
        up.add(h->d_pi[h->vi_cur], h->h_up + vb, 4 * vb);
        if ((rc = gu_device_segments(h, up)) != GU_OK) return rc;
    } else {
        GU_HIP(hipStreamSynchronize(h->stream));
        GU_HIP(hipMemcpy(h->d_v[h->vi_cur], v, vb, hipMemcpyHostToDevice));
        GU_HIP(hipMemcpy(h->d_pi[h->vi_cur], pi, 4 * vb, hipMemcpyHostToDevice));
    }
    h->has_vi = true;
    h->greedy_valid = false;
    return GU_OK;
}

int gu_vi_get(gu_handle h, double *v, double *pi)
{
    int rc = gu_use_device(h);
    if (rc != GU_OK) return rc;
    GU_REQUIRE(h->has_vi, GU_ERR_STATE, "no value/policy tables: call gu_vi_set first");
    const size_t vb = (size_t)h->S * sizeof(double);
    if (h->h_tables_valid && h->h_tables) {  // the launch that wrote the tables last left a copy on the host (and has been waited for)
        if (v) memcpy(v, h->h_tables, vb);
        if (pi) memcpy(pi, h->h_tables + h->S, 4 * vb);
        return GU_OK;
    }
    if (v && pi && h->h_ctl && 5 * vb <= GU_CTL_WORDS * sizeof(unsigned long long)) {  // both tables: two DMAs into the landing area, ONE wait
        char *land = (char *)h->h_ctl;
        GuSegments down;  // ONE launch writes both tables into the page-locked landing area, one wait
        down.add(land, h->d_v[h->vi_cur], vb);
        down.add(land + vb, h->d_pi[h->vi_cur], 4 * vb);
        if ((rc = gu_device_segments(h, down)) != GU_OK) return rc;
        GU_HIP(hipStreamSynchronize(h->stream));
        memcpy(v, land, vb);
        memcpy(pi, land + vb, 4 * vb);
        return GU_OK;
    }
    if (v && (rc = gu_read_back(h, v, h->d_v[h->vi_cur], vb)) != GU_OK) return rc;
    if (pi && (rc = gu_read_back(h, pi, h->d_pi[h->vi_cur], 4 * vb)) != GU_OK) return rc;
    if (!v && !pi) GU_HIP(hipStreamSynchronize(h->stream));
    return GU_OK;
}

int gu_vi_sweep(gu_handle h, double gamma, int32_t iters, int32_t greedy_update, double *deltas)
{
    int rc = gu_use_device(h);
    if (rc != GU_OK) return rc;
    h->h_tables_valid = false;  // (the call may write the tables: the host's copy of them, gu_vi_xcd.hip, is withdrawn; the per-XCD launch of the tables alone renews it)
    GU_REQUIRE(h->has_vi, GU_ERR_STATE, "no value/policy tables: call gu_vi_set first");
    GU_REQUIRE(iters > 0 && iters <= 4096, GU_ERR_INVALID, "iters must be in 1..4096 per call");
    if (vi_xcd_preferred(h, iters)) {  // one launch of one XCD's workgroups, nothing leaves that XCD's L2 (gu_vi_xcd.hip)
        int32_t done = 0;
        rc = gu_vi_xcd_dp_run(h, gamma, 0.0, false, greedy_update != 0, iters, &done, deltas);
        h->vi_dp_form = 1;
        if (rc != GU_VI_FALLBACK) return rc;
    }
    if (vi_block_eligible(h)) {
        int32_t done = 0;
        h->vi_dp_form = 2;
        return vi_block_run(h, gamma, 0.0, false, greedy_update != 0, iters, &done, deltas);
    }
    if (vi_cluster_eligible(h)) {
        int32_t done = 0;
        rc = vi_cluster_run(h, gamma, 0.0, false, greedy_update != 0, iters, &done, deltas);
        h->vi_dp_form = 3;
        if (rc != GU_VI_FALLBACK) return rc;
    }
    h->vi_dp_form = 4;
    GU_HIP(hipMemsetAsync(h->d_delta, 0, (size_t)iters * sizeof(unsigned long long), h->stream));
    const dim3 grid(vi_blocks(h->S)), block(VI_BLOCK);
    const bool lds = h->S <= GU_MAX_LDS_CELLS;
    const size_t smem = lds ? 2 * (size_t)h->cell_bytes : 0;
    for (int32_t i = 0; i < iters; ++i) {
        ViArgs a = vi_args(h, gamma, (unsigned long long *)h->d_delta + i);
        if (greedy_update) {
            if (lds) hipLaunchKernelGGL(gu_vi_round_kernel<true>, grid, block, smem, h->stream, a);
            else hipLaunchKernelGGL(gu_vi_round_kernel<false>, grid, block, 0, h->stream, a);
            h->vi_cur ^= 1;  // both tables advanced
        } else {
            if (lds) hipLaunchKernelGGL(gu_vi_eval_kernel<true>, grid, block, smem, h->stream, a);
            else hipLaunchKernelGGL(gu_vi_eval_kernel<false>, grid, block, 0, h->stream, a);
            // policy unchanged: only v advances; keep pi where it is by swapping v buffers only
            double *t = h->d_v[0];
            h->d_v[0] = h->d_v[1];
            h->d_v[1] = t;
        }
    }
    GU_HIP(hipGetLastError());
    h->greedy_valid = false;
    if (deltas) {
        std::vector<unsigned long long> keys((size_t)iters);
        {
            const int rb = gu_read_back(h, keys.data(), h->d_delta, (size_t)iters * sizeof(unsigned long long));
            if (rb != GU_OK) return rb;
        }
        for (int32_t i = 0; i < iters; ++i) deltas[i] = vi_unkey(keys[(size_t)i]);
    }
    return GU_OK;
}

int gu_vi_run(gu_handle h, double gamma, double threshold, int32_t max_steps, int32_t *steps_done, double *deltas)
{
    int rc = gu_use_device(h);
    if (rc != GU_OK) return rc;
    h->h_tables_valid = false;  // (the call may write the tables: the host's copy of them, gu_vi_xcd.hip, is withdrawn; the per-XCD launch of the tables alone renews it)
    GU_REQUIRE(h->has_vi, GU_ERR_STATE, "no value/policy tables: call gu_vi_set first");
    GU_REQUIRE(max_steps >= 0 && steps_done, GU_ERR_INVALID, "max_steps < 0 or steps_done is NULL");
    if (vi_xcd_preferred(h, max_steps)) {
        rc = gu_vi_xcd_dp_run(h, gamma, threshold, true, true, max_steps, steps_done, deltas);
        h->vi_dp_form = 1;
        if (rc != GU_VI_FALLBACK) return rc;
    }
    if (vi_block_eligible(h)) {
        h->vi_dp_form = 2;
        return vi_block_run(h, gamma, threshold, true, true, max_steps, steps_done, deltas);
    }
    if (vi_cluster_eligible(h)) {
        rc = vi_cluster_run(h, gamma, threshold, true, true, max_steps, steps_done, deltas);
        h->vi_dp_form = 3;
        if (rc != GU_VI_FALLBACK) return rc;
    }
    h->vi_dp_form = 4;
    const dim3 grid(vi_blocks(h->S)), block(VI_BLOCK);
    const bool lds = h->S <= GU_MAX_LDS_CELLS;
    const size_t smem = lds ? 2 * (size_t)h->cell_bytes : 0;
    int32_t *ctl = (int32_t *)(h->d_delta + 4000);  // past the delta slots a batch uses
    GU_HIP(hipMemsetAsync(ctl, 0, 8, h->stream));
    int32_t done_total = 0, stop = 0;
    const int start_cur = h->vi_cur;
    // Rounds are queued VI_RUN_BATCH at a time; the stopping rule is evaluated on the device, so a batch never runs
    // past the round that met the threshold (the launches behind it are no-ops) and the host looks only once per
    // batch.  Small batches bound the no-op tail: a no-op round still costs its launch.  (Replaying a batch as a
    // hipGraph was measured and is no faster: a round is bound by its own execution, ~5 us, not by the launch.)
    auto enqueue = [&](int32_t n) {
        GU_HIP(hipMemsetAsync(h->d_delta, 0, (size_t)n * sizeof(unsigned long long), h->stream));
        int cur = h->vi_cur;
        for (int32_t i = 0; i < n; ++i, cur ^= 1) {
            ViArgs a = vi_args(h, gamma, (unsigned long long *)h->d_delta + i);
            a.v = h->d_v[cur];
            a.pi = h->d_pi[cur];
            a.v_new = h->d_v[cur ^ 1];
            a.pi_new = h->d_pi[cur ^ 1];
            a.ctl = ctl;
            a.prev_delta_key = i ? (const unsigned long long *)h->d_delta + i - 1 : nullptr;  // the host checked the last batch
            a.threshold = threshold;
            if (lds) hipLaunchKernelGGL(gu_vi_round_kernel<true>, grid, block, smem, h->stream, a);
            else hipLaunchKernelGGL(gu_vi_round_kernel<false>, grid, block, 0, h->stream, a);
        }
        return GU_OK;
    };
    std::vector<unsigned long long> keys(VI_RUN_BATCH);
    for (int32_t base = 0; base < max_steps && !stop; base += VI_RUN_BATCH) {
        const int32_t n = max_steps - base < VI_RUN_BATCH ? max_steps - base : VI_RUN_BATCH;
        rc = enqueue(n);
        if (rc != GU_OK) return rc;
        h->vi_cur ^= n & 1;
        GU_HIP(hipGetLastError());
        {
            const int rb = gu_read_back(h, keys.data(), h->d_delta, (size_t)n * sizeof(unsigned long long));
            if (rb != GU_OK) return rb;
        }
        // rounds of this batch that ran: up to and including the first one whose delta met the threshold (:22-23)
        for (int32_t i = 0; i < n && !stop; ++i) {
            const double delta = vi_unkey(keys[(size_t)i]);
            if (deltas) deltas[base + i] = delta;
            ++done_total;
            stop = delta < threshold;
        }
    }
    h->vi_cur = start_cur ^ (done_total & 1);  // launches queued after the stop did not touch the tables
    h->greedy_valid = false;
    *steps_done = done_total;
    return GU_OK;
}

int gu_vi_eval_run(gu_handle h, double gamma, double threshold, int32_t max_steps, int32_t *steps_done, double *deltas)
{
    int rc = gu_use_device(h);
    if (rc != GU_OK) return rc;
    h->h_tables_valid = false;  // (the call may write the tables: the host's copy of them, gu_vi_xcd.hip, is withdrawn; the per-XCD launch of the tables alone renews it)
    GU_REQUIRE(h->has_vi, GU_ERR_STATE, "no value/policy tables: call gu_vi_set first");
    GU_REQUIRE(max_steps >= 0 && steps_done, GU_ERR_INVALID, "max_steps < 0 or steps_done is NULL");
    if (vi_xcd_preferred(h, max_steps)) {
        rc = gu_vi_xcd_dp_run(h, gamma, threshold, true, false, max_steps, steps_done, deltas);
        h->vi_dp_form = 1;
        if (rc != GU_VI_FALLBACK) return rc;
    }
    if (vi_block_eligible(h)) {
        h->vi_dp_form = 2;
        return vi_block_run(h, gamma, threshold, true, false, max_steps, steps_done, deltas);
    }
    if (vi_cluster_eligible(h)) {
        rc = vi_cluster_run(h, gamma, threshold, true, false, max_steps, steps_done, deltas);
        h->vi_dp_form = 3;
        if (rc != GU_VI_FALLBACK) return rc;
    }
    h->vi_dp_form = 4;
    // larger grids: one evaluation launch per sweep, the host looks at the deltas once per batch
    const dim3 grid(vi_blocks(h->S)), block(VI_BLOCK);
    const bool lds = h->S <= GU_MAX_LDS_CELLS;
    const size_t smem = lds ? 2 * (size_t)h->cell_bytes : 0;
    std::vector<unsigned long long> keys(VI_RUN_BATCH);
    int32_t done_total = 0;
    bool stop = false;
    while (done_total < max_steps && !stop) {
        // a sweep past the converged one would change v: snapshot v, queue one batch, find the first converged sweep,
        // and if the batch overshot it restore the snapshot and replay exactly that many sweeps
        const int32_t n = std::min<int32_t>(VI_RUN_BATCH, max_steps - done_total);
        rc = gu_ensure_scratch(h, (size_t)h->S * sizeof(double));
        if (rc != GU_OK) return rc;
        GU_HIP(hipMemcpyAsync(h->d_scratch, h->d_v[h->vi_cur], (size_t)h->S * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        auto sweeps = [&](int32_t count) {
            GU_HIP(hipMemsetAsync(h->d_delta, 0, (size_t)count * sizeof(unsigned long long), h->stream));
            for (int32_t i = 0; i < count; ++i) {
                ViArgs a = vi_args(h, gamma, (unsigned long long *)h->d_delta + i);
                if (lds) hipLaunchKernelGGL(gu_vi_eval_kernel<true>, grid, block, smem, h->stream, a);
                else hipLaunchKernelGGL(gu_vi_eval_kernel<false>, grid, block, 0, h->stream, a);
                double *t = h->d_v[0];  // policy unchanged: only v advances
                h->d_v[0] = h->d_v[1];
                h->d_v[1] = t;
            }
            GU_HIP(hipGetLastError());
            return GU_OK;
        };
        rc = sweeps(n);
        if (rc != GU_OK) return rc;
        {
            const int rb = gu_read_back(h, keys.data(), h->d_delta, (size_t)n * sizeof(unsigned long long));
            if (rb != GU_OK) return rb;
        }
        int32_t ran = n;
        for (int32_t i = 0; i < n; ++i) {
            const double delta = vi_unkey(keys[(size_t)i]);
            if (deltas) deltas[done_total + i] = delta;
            if (delta < threshold) {
                ran = i + 1;
                stop = true;
                break;
            }
        }
        if (ran < n) {  // overshot: back to the snapshot, replay the sweeps up to the converged one
            GU_HIP(hipMemcpyAsync(h->d_v[h->vi_cur], h->d_scratch, (size_t)h->S * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
            rc = sweeps(ran);
            if (rc != GU_OK) return rc;
        }
        done_total += ran;
    }
    h->greedy_valid = false;
    *steps_done = done_total;
    return GU_OK;
}

int gu_vi_greedy(gu_handle h, double gamma)
{
    int rc = gu_use_device(h);
    if (rc != GU_OK) return rc;
    h->h_tables_valid = false;  // (the call may write the tables: the host's copy of them, gu_vi_xcd.hip, is withdrawn; the per-XCD launch of the tables alone renews it)
    GU_REQUIRE(h->has_vi, GU_ERR_STATE, "no value/policy tables: call gu_vi_set first");
    ViArgs a = vi_args(h, gamma, nullptr);
    a.v_new = h->d_v[h->vi_cur];  // V2 reads the CURRENT value table ...
    const dim3 grid(vi_blocks(h->S)), block(VI_BLOCK);
    if (h->S <= GU_MAX_LDS_CELLS)
        hipLaunchKernelGGL(gu_vi_greedy_kernel<true>, grid, block, 2 * (size_t)h->cell_bytes, h->stream, a);
    else
        hipLaunchKernelGGL(gu_vi_greedy_kernel<false>, grid, block, 0, h->stream, a);
    GU_HIP(hipGetLastError());
    double *t = h->d_pi[0];  // ... and only the policy table advances
    h->d_pi[0] = h->d_pi[1];
    h->d_pi[1] = t;
    h->greedy_valid = false;
    return GU_OK;
}

int gu_vi_sweep_step(gu_handle h, double gamma, uint32_t flags, double *delta)
{
    int rc = gu_use_device(h);
    if (rc != GU_OK) return rc;
    h->h_tables_valid = false;  // (the call may write the tables: the host's copy of them, gu_vi_xcd.hip, is withdrawn; the per-XCD launch of the tables alone renews it)
    h->entry_table_ok = false;  // (the fused launches step the envs: their state is consistent too, but only rollouts vouch for it)
    GU_REQUIRE(h->has_vi, GU_ERR_STATE, "no value/policy tables: call gu_vi_set first");
    GU_REQUIRE((flags & ~GU_F_AUTO_RESET) == 0, GU_ERR_INVALID, "gu_vi_sweep_step accepts only GU_F_AUTO_RESET");
    GU_REQUIRE(!h->trail_cap, GU_ERR_UNSUPPORTED, "the agent trail is on (gu_trail_enable): the fused sweep + step launches do not feed it");
    GU_HIP(hipMemsetAsync(h->d_delta, 0, sizeof(unsigned long long), h->stream));
    ViStepArgs a{};
    a.vi = vi_args(h, gamma, (unsigned long long *)h->d_delta);
    a.pos = h->pos();
    a.reward = h->reward();
    a.done = h->done();
    a.episode = h->d_episode;
    a.starts = h->d_starts;
    a.n_starts = (uint32_t)h->n_starts;
    a.seed_prefix = h->seed_prefix;
    a.env_id0 = (uint32_t)h->env_id0;
    a.N = h->N;
    a.flags = flags;
    a.done_bits = h->d_done_bits;
    const int64_t threads = h->N > h->S ? h->N : h->S;
    const dim3 grid(vi_blocks(threads)), block(VI_BLOCK);
    if (h->S <= GU_MAX_LDS_CELLS)
        hipLaunchKernelGGL(gu_vi_sweep_step_kernel<true>, grid, block, 2 * (size_t)h->cell_bytes, h->stream, a);
    else
        hipLaunchKernelGGL(gu_vi_sweep_step_kernel<false>, grid, block, 0, h->stream, a);
    GU_HIP(hipGetLastError());
    h->vi_cur ^= 1;
    h->greedy_valid = false;
    h->steps_taken += 1;
    if (delta) {
        unsigned long long key = 0;
        if ((rc = gu_read_back(h, &key, h->d_delta, sizeof key)) != GU_OK) return rc;
        *delta = vi_unkey(key);
    }
    return GU_OK;
}

int gu_vi_sweep_step_run(gu_handle h, double gamma, int32_t iters, uint32_t flags, double *deltas)
{
    int rc = gu_use_device(h);
    if (rc != GU_OK) return rc;
    h->h_tables_valid = false;  // (the call may write the tables: the host's copy of them, gu_vi_xcd.hip, is withdrawn; the per-XCD launch of the tables alone renews it)
    h->entry_table_ok = false;
    GU_REQUIRE(h->has_vi, GU_ERR_STATE, "no value/policy tables: call gu_vi_set first");
    GU_REQUIRE((flags & ~GU_F_AUTO_RESET) == 0, GU_ERR_INVALID, "gu_vi_sweep_step_run accepts only GU_F_AUTO_RESET");
    GU_REQUIRE(!h->trail_cap, GU_ERR_UNSUPPORTED, "the agent trail is on (gu_trail_enable): the fused sweep + step launches do not feed it");
    GU_REQUIRE(iters > 0 && iters <= 1000000, GU_ERR_INVALID, "iters must be in 1..1000000");
    const int64_t threads = h->N > h->S ? h->N : (int64_t)h->S;
    const int64_t G = (threads + VI_CL_THREADS - 1) / VI_CL_THREADS;
    const int64_t path = gu_opt(h, GU_OPT_VI_PATH);
    // Three forms, fastest first: ONE launch synchronised per XCD (gu_vi_xcd.hip), ONE launch with a chip-wide barrier per round,
    // one launch per round.  The one-launch forms work on a snapshot's ORIGINAL: a form that gives up (every spin in them is bounded)
    // leaves half-advanced state, which is put back before the next form runs.
    GuXcdPlan xp{};
    // (a call of ONE round: the one-launch forms cost ~10 us more to start -- the registration wait, the snapshot, the zeroed
    // exchange buffers -- than they save; profiles/archive/r04z_c5_forms.json, xcd_default_us_per_call_by_rounds.  GU_OPT_VI_PATH = 6: always)
    const bool short_call = path == 0 && iters <= 1;
    const bool try_xcd = !short_call && (path == 0 || path == 5 || path == 6) && gu_vi_xcd_plan(h, true, &xp);
    const bool try_cluster = h->n_grids == 1 && h->S <= GU_MAX_LDS_CELLS && G <= (h->n_cu < VI_CL_MAX_WGS ? h->n_cu : VI_CL_MAX_WGS) &&
                             !short_call && (path == 0 || path == 3 || path == 4 || path == 5 || path == 6);
    if (try_xcd) {  // ONE launch per up to 4096 rounds, results into the other halves of the double buffers (gu_vi_xcd.hip)
        rc = gu_vi_xcd_fused_run(h, xp, gamma, iters, flags, deltas);
        if (rc == GU_OK) h->vi_run_form = 1;
        if (rc != GU_VI_FALLBACK) return rc;
    }
    if (try_cluster) {
        // scratch: header (64 B) | delta keys [iters] | delta-key slots of the per-XCD form | snapshot | the per-XCD granule buffers
        const size_t key_bytes = (size_t)iters * sizeof(unsigned long long);
        const size_t slots_off = (64 + key_bytes + 255) & ~(size_t)255;
        const size_t snap_off = slots_off + (try_xcd ? xp.slots_bytes : 0);
        const size_t v_bytes = (size_t)h->S * sizeof(double), n4 = (size_t)h->N * 4, bits_bytes = (((size_t)h->N + 63) / 64) * 8;
        const size_t snap_bytes = (5 * v_bytes + 4 * n4 + bits_bytes + 255) & ~(size_t)255;
        rc = gu_ensure_scratch(h, snap_off + snap_bytes + (try_xcd ? 8 * xp.work_bytes : 0));
        if (rc != GU_OK) return rc;
        uint32_t *hdr_d = (uint32_t *)h->d_scratch;
        int32_t *done_d = (int32_t *)h->d_scratch + 2;
        unsigned long long *keys_d = (unsigned long long *)((char *)h->d_scratch + 64);
        char *snap = (char *)h->d_scratch + snap_off;
        // what a form that gives up would leave half-advanced -- tables, positions, rewards, done flags and their ballots,
        // episode counters -- is snapshot first (device to device)
        void *live[5] = {h->d_v[h->vi_cur], h->d_pi[h->vi_cur], h->d_out3, h->d_episode, h->d_done_bits};
        const size_t size[5] = {v_bytes, 4 * v_bytes, 3 * n4, n4, bits_bytes};
        size_t off = 0;
        bool snapped = false;
        for (int form = 1; form < 2; ++form) {  // (0: per XCD -- handled above since round 5), 1: chip-wide
            if (!try_cluster) continue;
            // ONE launch: the snapshot (first form tried only), every polled word zeroed (before every launch), and for the per-XCD
            // form its exchange buffers zeroed -- every word that crosses workgroups there is tagged with its round, no tag of an
            // earlier launch may be left
            GuSegments seg;
            off = 0;
            for (int k = 0; k < 5 && !snapped; off += size[k], ++k) seg.add(snap + off, live[k], size[k]);
            snapped = true;
            seg.add(h->d_scratch, nullptr, form == 0 ? snap_off : slots_off);
            if (form == 0) seg.add(snap + snap_bytes, nullptr, 8 * xp.work_bytes);
            if ((rc = gu_device_segments(h, seg)) != GU_OK) return rc;
            ViStepXcdArgs a{};
            a.vi = ViClusterArgs{h->d_cell, h->cell_bytes, h->W, h->S, gamma, 0.0, h->d_v[h->vi_cur], h->d_v[h->vi_cur ^ 1],
                                 h->d_pi[h->vi_cur], keys_d, hdr_d, done_d, iters, 0};
            a.pos = h->pos();
            a.reward = h->reward();
            a.done = h->done();
            a.episode = h->d_episode;
            a.starts = h->d_starts;
            a.n_starts = (uint32_t)h->n_starts;
            a.seed_prefix = h->seed_prefix;
            a.env_id0 = (uint32_t)h->env_id0;
            a.N = h->N;
            a.flags = flags;
            a.done_bits = h->d_done_bits;
            if (form == 0) {
                a.slots = (vi_u64 *)((char *)h->d_scratch + slots_off);
                a.gx = (uint8_t *)(snap + snap_bytes);
                a.work_bytes = (uint32_t)xp.work_bytes;
                a.inject_failure = path == 5;  // tests: the per-XCD form gives up
                if ((rc = gu_vi_xcd_launch(h, xp, a, true, true)) != GU_OK) return rc;
            } else {
                if (path == 3) GU_HIP(hipMemsetD32Async((hipDeviceptr_t)(hdr_d + 1), 1, 1, h->stream));  // tests: an injected timeout
                hipLaunchKernelGGL(gu_vi_sweep_step_cluster_kernel, dim3((unsigned)G), dim3(VI_CL_THREADS), 2 * (size_t)h->cell_bytes, h->stream,
                                   static_cast<const ViStepClusterArgs &>(a));
                GU_HIP(hipGetLastError());
            }
            // rounds_done is written by ONE workgroup, and giving up need not be unanimous: the fallback word, raised by whoever
            // gives up, decides (see vi_cluster_run)
            // (header and delta keys lie side by side: ONE copy back, one wait)
            std::vector<unsigned long long> back(8 + (deltas ? (size_t)iters : 0));
            if ((rc = gu_read_back(h, back.data(), h->d_scratch, back.size() * sizeof(unsigned long long))) != GU_OK) return rc;
            int32_t ctl[4];  // [arrival counter (chip-wide form), fallback word, rounds_done, -]
            memcpy(ctl, back.data(), sizeof ctl);
            if (form == 0) {  // the per-XCD form's registration word: eight 7-bit counts of workgroups per HW_REG_XCC_ID (gu_vi_xcd.hip)
                for (int k = 0; k < 8; ++k) h->vi_xcd_members[k] = (int32_t)((back[2] >> (7 * k)) & 0x7Full);
                h->vi_xcd_torn += (int64_t)(back[4] & 0xFFFFFFFFull);  // (hdr[8]: counted by a -DGU_VI_XCD_TORN build only)
            }
            if (!ctl[1] && ctl[2] == iters) {
                for (int32_t i = 0; deltas && i < iters; ++i) deltas[i] = vi_unkey(back[8 + (size_t)i]);
                if (iters & 1) {
                    double *t = h->d_v[0];
                    h->d_v[0] = h->d_v[1];
                    h->d_v[1] = t;
                }
                h->greedy_valid = false;
                h->steps_taken += (uint64_t)iters;
                h->vi_run_form = form == 0 ? 1 : 2;
                return GU_OK;
            }
            off = 0;
            for (int k = 0; k < 5; off += size[k], ++k)
                if ((rc = gu_device_copy(h, live[k], snap + off, size[k])) != GU_OK) return rc;
            GU_HIP(hipStreamSynchronize(h->stream));
            if (gu_debug())
                fprintf(stderr, "[gu] sweep-step %s kernel gave up (workgroups not resident together, or clusters too uneven); state restored, next form\n",
                        form == 0 ? "per-XCD" : "chip-wide cluster");
        }
    }
    // one fused launch per round; the deltas are collected once at the end
    std::vector<double> one((size_t)iters);
    for (int32_t i = 0; i < iters; ++i) {
        rc = gu_vi_sweep_step(h, gamma, flags, deltas ? &one[(size_t)i] : nullptr);
        if (rc != GU_OK) return rc;
    }
    if (deltas) memcpy(deltas, one.data(), (size_t)iters * sizeof(double));
    h->vi_run_form = 3;
    return GU_OK;
}

int gu_vi_last_form(gu_handle h) { return h ? h->vi_run_form : 0; }
int gu_vi_last_dp_form(gu_handle h) { return h ? h->vi_dp_form : 0; }
int gu_vi_xcd_torn_words(gu_handle h, int64_t *count)
{
    if (!h) return gu_fail(GU_ERR_INVALID, "null handle");
#ifdef GU_VI_XCD_TORN
    if (count) *count = h->vi_xcd_torn;
    return GU_OK;
#else
    if (count) *count = -1;
    return gu_fail(GU_ERR_UNSUPPORTED, "this library was built without -DGU_VI_XCD_TORN (make variant VARIANT=_torn EXTRA=-DGU_VI_XCD_TORN)");
#endif
}

int gu_vi_last_clusters(gu_handle h, int32_t *members)
{
    if (!h || !members) return gu_fail(GU_ERR_INVALID, "null handle or pointer");
    memcpy(members, h->vi_xcd_members, sizeof h->vi_xcd_members);
    return GU_OK;
}

}  // extern "C"
