// gu_vi_xcd.hip -- config 5 for many rounds in ONE launch, synchronised PER XCD instead of chip-wide.
//
// `iters` x { V1 + V2 sweep of the value / policy tables (core/algorithms/utils.py:15-27, 55-72 under
// core/algorithms/dynamic_programming.py:15-20); every agent takes one greedy step on the policy of that round
// (examples/griduniverse_alg_examples.py:76 on env:136-193) }.
//
// gu_vi_sweep_step_cluster_kernel (gu_vi.hip) runs this with ONE copy of the table shared by every workgroup of the chip:
// a round there is V1 -> write-through store of v' -> barrier over all workgroups on 8 XCDs whose L2s are not coherent with
// each other (every shared byte crosses the fabric) -> V2 + agent step: 5.25 us per round at config 5, of which the
// arithmetic is < 0.5 us.  Here EVERY XCD sweeps the WHOLE table redundantly (4096 states at 64x64: nothing next to 65 536
// agents) with its own workgroups and steps its own share of the agents, so nothing a round needs ever leaves the XCD:
//
//   cluster      = the workgroups that read the same HW_REG_XCC_ID; each claims a rank in its cluster at start (one returning
//                  atomic per workgroup + ONE chip-wide arrival wait per launch, so that every cluster knows its size).
//                  Membership is what the hardware reports, not a guess from blockIdx: any placement gives correct results,
//                  the observed round-robin placement (blocks b and b + 8 share an XCD) gives equal clusters.
//   state chunk  = cluster of n workgroups, workgroup `rank` owns states [rank * chunk, (rank + 1) * chunk), chunk =
//                  ceil(S / n) rounded up to whole waves; K states per thread at most (a cluster too small for that gives up).
//   value table  = every workgroup keeps the WHOLE table in LDS (V1, V2 and the agents' four neighbour values are LDS reads);
//                  the new values travel through a per-XCD double-buffered copy in global memory that only this XCD ever
//                  touches: 8-byte stores that stay in the XCD's L2 (`sc0`), every storing wave drains (`s_waitcnt vmcnt(0)`
//                  = the L2 has them), workgroup barrier, and after the cluster barrier every workgroup reloads the table with
//                  16-byte L1-bypassing loads (`sc1`: served by that same L2).
//   cluster barrier = one 16-byte slot per workgroup and round parity in that same L2, {delta key high | round, delta key low |
//                  round}: a workgroup's first lane stores its slot, the lanes of its first wave poll the cluster's slots (L1-
//                  bypassing 8-byte loads) until every tag is this round's.  A torn slot shows a wrong tag and is polled again;
//                  a slot of parity p is rewritten only two rounds later, which its owner cannot reach before every member
//                  has passed the barrier in between.  No atomics, no fabric traffic, and the round's delta (the maximum over
//                  states of v - v', dynamic_programming.py:17) arrives with the barrier.
//
// The same float64 operations in the same order per state as every other DP kernel of the library, hence the same bits; the
// agents of all clusters see identical tables, so which cluster steps which env is invisible in the results.  Every spin is
// bounded and raises the launch's fallback word; the caller (gu_vi_sweep_step_run) restores its snapshot and takes the chip-wide
// cluster kernel, then one launch per round.  All workgroups must be resident together: at most one per CU.
#include "gu_internal.hpp"
#include "gu_rng.hpp"
#include "gu_vi.hpp"

#define VI_XCD_MAX_XCC 8
#define VI_XCD_SLOTS 64                      /* workgroups of one cluster at most */
#define VI_XCD_GETREG_XCC_ID (20 | (3 << 11)) /* s_getreg_b32 hwreg(HW_REG_XCC_ID, 0, 4) */

// cache policy of the loads that fetch the members' chunks (buffer_load aux: 2 = nt, 16 = sc1, 17 = sc0 sc1; each bypasses this CU's L1)
#ifndef VI_XCD_LOAD_AUX
#define VI_XCD_LOAD_AUX 16
#endif

typedef uint32_t vi_u32x4 __attribute__((ext_vector_type(4)));

// 8-byte store that stays in this XCD's L2 (workgroup scope: `global_store_dwordx2 ... sc0`)
__device__ __forceinline__ void vi_st_l2(vi_u64 *p, vi_u64 x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// 8-byte / 4-byte loads that bypass this CU's L1 (`sc1`)
__device__ __forceinline__ vi_u64 vi_ld_l2(const vi_u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t vi_ld_word(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Maximum over the wave, left in its LAST lane (lane 63): DPP row shifts inside each row of 16 lanes, then the two row broadcasts
// of the gfx9 family -- register moves inside the SIMD, no LDS round trips.  (A 64-bit __shfl butterfly costs ~700 clocks of
// dependent ds_bpermute traffic.  An LDS atomicMax per lane is turned by the compiler's atomic optimizer into a scalar loop over
// the active lanes at ~100 clocks per lane: 7300 clocks for a full wave, and still 3400 for the 32 lanes that hold a cluster's
// slot keys -- in ONE workgroup, for which the whole cluster then waits every round.)
__device__ __forceinline__ vi_u64 vi_wave_max_last(vi_u64 k)
{
#define VI_DPP_STEP(ctrl, rows)                                                                                          \
    {                                                                                                                    \
        const uint32_t lo_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)k, ctrl, rows, 0xF, false);         \
        const uint32_t hi_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(k >> 32), ctrl, rows, 0xF, false); \
        const vi_u64 o_ = ((vi_u64)hi_ << 32) | lo_;                                                                     \
        k = o_ > k ? o_ : k;                                                                                             \
    }
    VI_DPP_STEP(0x111, 0xF)  // row_shr:1 (a lane without a source reads 0: keys are never below 1)
    VI_DPP_STEP(0x112, 0xF)  // row_shr:2
    VI_DPP_STEP(0x114, 0xF)  // row_shr:4
    VI_DPP_STEP(0x118, 0xF)  // row_shr:8      -> lane 15 of every row holds the row's maximum
    VI_DPP_STEP(0x142, 0xA)  // row_bcast:15   -> rows 1 and 3 take in the last lane of rows 0 and 2
    VI_DPP_STEP(0x143, 0xC)  // row_bcast:31   -> rows 2 and 3 take in lane 31
#undef VI_DPP_STEP
    return k;
}

// -DGU_VI_XCD_STAMPS (a diagnostic variant library, tools/c5_stamps.py; never the product): workgroup rank 0 of the cluster that
// writes the tables sums the shader-clock cycles its first wave spends in every phase of a round and returns the sums IN PLACE
// OF the first deltas (delta_key[0 .. 11]; [10] = poll turns, [11] = 100 MHz ticks of the whole loop).
#ifdef GU_VI_XCD_STAMPS
#define VI_STAMP(i)                                             \
    do {                                                        \
        const uint64_t now_ = __builtin_amdgcn_s_memtime();     \
        stamp_acc[i] += now_ - stamp_last;                      \
        stamp_last = now_;                                      \
    } while (0)
#else
#define VI_STAMP(i) do { } while (0)
#endif

// K = states per thread at most.  Launched with 256, 512 or 1024 threads per workgroup.
template <int K>
__global__ void __launch_bounds__(1024) gu_vi_sweep_step_xcd_kernel(const ViStepXcdArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ vi_u64 wg_key[4];
    __shared__ uint32_t info[4];  // [0] XCC id, [1] rank in the cluster, [2] members, [3] bit 0: failed, bit 1: this cluster writes the tables
    const ViMap cell = vi_stage<true>(a.vi.cell, a.vi.cell_bytes, smem);  // the agents gather records of arbitrary cells
    double *vL = reinterpret_cast<double *>(smem + 2 * a.vi.cell_bytes);  // [S2 + 2] this workgroup's copy of the value table (+ a spare slot)
    const int32_t tid = threadIdx.x, B = blockDim.x, S = a.vi.S, W = a.vi.W;
    const int32_t S2 = (S + 1) & ~1, cb = a.vi.cell_bytes;
    const uint8_t *actL = smem + 2 * cb + S2 * 8 + 16;  // [cell_bytes] greedy action per state under the policy of the round before
    const double gamma = a.vi.gamma;
    uint32_t *hdr = a.vi.sync;  // [0] workgroups registered, [1] fallback word, [3] 1 + XCC id of workgroup 0, [4 .. 11] members per XCC

    // ---- registration: who shares this XCD ----
    if (tid == 0) {
        const uint32_t xcc = __builtin_amdgcn_s_getreg(VI_XCD_GETREG_XCC_ID);
        uint32_t rank = 0, bad = xcc >= VI_XCD_MAX_XCC;
        if (!bad) {
            rank = __hip_atomic_fetch_add(hdr + 4 + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (blockIdx.x == 0) __hip_atomic_store(hdr + 3, xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the claim (returned) and the leader word (written through) are out
        __hip_atomic_fetch_add(hdr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t spins = 0;
        while (!bad && vi_ld_word(hdr) < gridDim.x) {  // the one chip-wide wait of the launch
            __builtin_amdgcn_s_sleep(1);
            if (++spins > VI_CL_SPIN_LIMIT || vi_ld_word(hdr + 1)) bad = 1u;
        }
        uint32_t members = 0, writes = 0;
        if (!bad) {
            members = vi_ld_word(hdr + 4 + xcc);
            writes = vi_ld_word(hdr + 3) == xcc + 1u;
            const int64_t chunk = ((((int64_t)S + members - 1) / members) + 63) & ~(int64_t)63;
            bad = rank >= VI_XCD_SLOTS || members > VI_XCD_SLOTS || chunk > (int64_t)K * B;
        }
        if (bad || a.inject_failure) {
            bad = 1u;
            __hip_atomic_store(hdr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // tell everyone; never hang
        }
        info[0] = xcc & (VI_XCD_MAX_XCC - 1);
        info[1] = rank;
        info[2] = members ? members : 1u;
        info[3] = bad | (writes << 1);
    }
    __syncthreads();
    const uint32_t xcc = info[0], rank = info[1], members = info[2];
    const bool writes_tables = (info[3] & 2u) != 0;
    bool failed = (info[3] & 1u) != 0;
    const int32_t chunk = (int32_t)(((((int64_t)S + members - 1) / members) + 63) & ~(int64_t)63);
    vi_u64 *slots = a.slots + (size_t)xcc * 2 * VI_XCD_SLOTS * 2;                         // [parity][member][2]
    double *vx = a.vx + (size_t)xcc * 2 * S2;                                             // [parity][S2]
    uint8_t *ax = a.ax + (size_t)xcc * 2 * cb;                                            // [parity][cell_bytes]

    // ---- per-state constants, the initial table, the own env ----
    int32_t st[K];    // the thread's states (-1: none)
    uint32_t rec[K], rn[K];
    int32_t r_own[K];
    double p[K][4];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const int32_t local = tid + j * B;
        const int32_t s = (int32_t)rank * chunk + local;
        st[j] = (!failed && local < chunk && s < S) ? s : -1;
        rec[j] = rn[j] = 0u;
        r_own[j] = 0;
#pragma unroll
        for (int act = 0; act < 4; ++act) p[j][act] = 0.0;
        if (st[j] >= 0) {
            rec[j] = cell.f[s];
            r_own[j] = cell.r[s];
#pragma unroll
            for (uint32_t act = 0; act < 4; ++act) rn[j] |= (uint32_t)(uint8_t)cell.r[vi_next(s, rec[j], act, W)] << (8 * act);
            const double4 row = *reinterpret_cast<const double4 *>(a.vi.pi + 4 * (int64_t)s);
            p[j][0] = row.x, p[j][1] = row.y, p[j][2] = row.z, p[j][3] = row.w;
        }
    }
    for (int32_t i = tid; i < S2; i += B) vL[i] = i < S ? a.vi.v0[i] : 0.0;
    const int64_t gid = (int64_t)blockIdx.x * B + tid;
    const bool own_env = gid < a.N;
    int32_t e_pos = 0, e_rew = 0, e_done = 0;
    uint32_t e_ep = 0, e_prefix = 0;
    if (own_env) {
        e_pos = a.pos[gid];
        e_rew = a.reward[gid];
        e_done = a.done[gid];
        e_ep = a.episode[gid];
        e_prefix = gu_rng_prefix(a.seed_prefix, a.env_id0 + (uint32_t)gid);
    }
    __syncthreads();

    const int32_t wave = tid >> 6, lane = tid & 63;
    bool wave_has_states = false;  // (wave-uniform: the states of a chunk are dealt to whole waves)
#pragma unroll
    for (int j = 0; j < K; ++j) wave_has_states = wave_has_states || __any(st[j] >= 0);
    if (tid < 4) wg_key[tid] = 0ull;  // [0, 1] this workgroup's delta key by round parity, [2, 3] the cluster's (workgroup 0 of the writing cluster)
    __syncthreads();
#ifdef GU_VI_XCD_STAMPS
    uint64_t stamp_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t stamp_last = __builtin_amdgcn_s_memtime();
    const uint64_t stamp_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    const bool keeps_deltas = writes_tables && rank == 0;
    // What a round fetches, in 16-byte items: the new values of the own chunk and of one grid row either side of it -- every
    // state a V1 / V2 of the chunk reads (a state's successors are itself, s +- 1 and s +- W) -- and the action table.
    const int32_t lo = (int32_t)rank * chunk, hi = lo + chunk < S ? lo + chunk : S;
    const int32_t iv0 = (lo - W > 0 ? lo - W : 0) >> 1, iv1 = ((hi + W < S2 ? hi + W : S2) + 1) >> 1;
    const int32_t nv = lo < S ? iv1 - iv0 : 0, na = cb >> 4;
    const uint32_t off_v = (uint32_t)(reinterpret_cast<const char *>(vx) - reinterpret_cast<const char *>(a.vx));
    const uint32_t off_a = (uint32_t)(reinterpret_cast<const char *>(ax) - reinterpret_cast<const char *>(a.vx));
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)a.vx, 0, a.work_bytes, 0x00020000);
    const uint32_t lds_v = 2u * (uint32_t)cb, lds_a = lds_v + (uint32_t)S2 * 8u + 16u, lds_spare = lds_v + (uint32_t)S2 * 8u;
    // One cluster barrier: post this workgroup's slot for `tag`, wait until every member's slot shows it, fetch.  Polling is per
    // WAVE (the lanes of every wave read the members' slots): no workgroup barrier between the poll and the loads that depend on it.
    auto exchange = [&](uint32_t par, uint32_t tag, bool with_v, bool with_act) {
        vi_u64 *slot = slots + (size_t)par * VI_XCD_SLOTS * 2;
        if (tid == 0) {
            const vi_u64 mine = wg_key[par];
            wg_key[par] = 0ull;  // (next written two rounds on, two workgroup barriers away)
            vi_st_l2(slot + 2 * rank, (mine & 0xFFFFFFFF00000000ull) | tag);
            vi_st_l2(slot + 2 * rank + 1, (mine << 32) | tag);
        }
        const bool polls = (uint32_t)lane < members;
        vi_u64 khi = 0ull, klo = 0ull;
        uint32_t spins = 0, bad = 0u;
        for (;;) {
            if (polls) {
                khi = vi_ld_l2(slot + 2 * lane);
                klo = vi_ld_l2(slot + 2 * lane + 1);
            }
            if (__all(!polls || ((uint32_t)khi == tag && (uint32_t)klo == tag))) break;
            if (++spins > VI_CL_SPIN_LIMIT || ((spins & 255u) == 0u && vi_ld_word(hdr + 1))) {
                bad = 1u;
                break;
            }
            if (spins > 4u) __builtin_amdgcn_s_sleep(1);
        }
#ifdef GU_VI_XCD_STAMPS
        stamp_acc[10] += spins;
#endif
        VI_STAMP(4);
        if (bad) {
            if (lane == 0) {
                __hip_atomic_store(hdr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                atomicOr(&info[3], 1u);
            }
            return;
        }
        if (keeps_deltas && wave == 0) {  // the round's delta: the maximum of the members' keys
            const vi_u64 k = vi_wave_max_last(polls ? (khi & 0xFFFFFFFF00000000ull) | (klo >> 32) : 0ull);
            if (lane == 63) wg_key[2 + par] = k;
        }
        // Straight-line, unpredicated loads: a load or an LDS write inside a divergent branch makes the compiler wait for every
        // load in flight at the branch (a first version ran its eight loads ONE AFTER THE OTHER, 450 clocks each).  A lane without
        // an item loads from beyond the buffer's size (the bounds check returns zeros without a memory access) and writes to a
        // spare 16 bytes behind the table.
#ifdef VI_XCD_EXPERIMENT  // timing experiments of the diagnostic build (results are WRONG): 1 = no action table, 2 = no values fetched
        if (VI_XCD_EXPERIMENT == 1) with_act = false;
        if (VI_XCD_EXPERIMENT == 2) with_v = false;
#endif
        const int32_t n_v = with_v ? nv : 0, total = n_v + (with_act ? na : 0);
        const uint32_t src_v = off_v + par * (uint32_t)S2 * 8u + (uint32_t)iv0 * 16u, src_a = off_a + par * (uint32_t)cb;
        for (int32_t base = 0; base < total; base += 4 * B) {
            vi_u32x4 t[4];
            uint32_t dst[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int32_t x = base + m * B + tid;
                const bool is_v = x < n_v, valid = x < total;
                const uint32_t src = is_v ? src_v + (uint32_t)x * 16u : src_a + (uint32_t)(x - n_v) * 16u;
                dst[m] = !valid ? lds_spare : is_v ? lds_v + (uint32_t)(iv0 + x) * 16u : lds_a + (uint32_t)(x - n_v) * 16u;
                t[m] = __builtin_amdgcn_raw_buffer_load_b128(rs, valid ? src : 0xFFFFFFF0u, 0, VI_XCD_LOAD_AUX);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) *reinterpret_cast<vi_u32x4 *>(smem + dst[m]) = t[m];
        }
    };
    // every agent takes the step of round `r - 1`: the greedy action of its cell (np.argmax of its policy row: the first maximum;
    // examples/griduniverse_alg_examples.py:76) is in the action table the state's owner published
    auto agents = [&]() {
        if (own_env) {
            if ((a.flags & GU_F_AUTO_RESET) && e_done) {  // lazy `if done: env.reset()`
                e_pos = a.starts[gu_rng_start_index(e_prefix, e_ep, a.n_starts)];
                ++e_ep;
            }
            e_pos = vi_next(e_pos, cell.f[e_pos], actL[e_pos], W);
            e_rew = cell.r[e_pos];
            e_done = (cell.f[e_pos] >> GU_CELL_TERM_BIT) & 1;
        }
    };
    uint32_t act_prev[K];  // greedy action of the thread's states under the policy of the round before
#pragma unroll
    for (int j = 0; j < K; ++j) act_prev[j] = 0u;
    int r = 0;
    for (; r < a.vi.max_rounds && !failed; ++r) {
        const uint32_t par = (uint32_t)r & 1u;
        double *vxr = vx + (size_t)par * S2;
        uint8_t *axr = ax + (size_t)par * cb;
        if (wave_has_states) {
            vi_u64 key = 0ull;
#pragma unroll
            for (int j = 0; j < K; ++j) {  // V1 (utils.py:15-27) on the LDS copy of the old values
                const int32_t s = st[j];
                if (s >= 0) {
                    double acc = __dadd_rn(0.0, (double)r_own[j]);
#pragma unroll
                    for (uint32_t act = 0; act < 4; ++act)
                        acc = __dadd_rn(acc, __dmul_rn(p[j][act], __dmul_rn(gamma, vL[vi_next(s, rec[j], act, W)])));
                    vi_st_l2(reinterpret_cast<vi_u64 *>(vxr + s), (vi_u64)__double_as_longlong(acc));
                    axr[s] = (uint8_t)act_prev[j];  // (the action of round r - 1 travels with the values of round r)
                    const vi_u64 k = vi_key(__dsub_rn(vL[s], acc));  // signed, dynamic_programming.py:17
                    key = k > key ? k : key;
                }
            }
            VI_STAMP(0);
            // the workgroup's maximum: DPP over the wave, then one LDS atomic by its last lane; the barrier below orders them
            key = vi_wave_max_last(key);
            if (lane == 63 && key) atomicMax(&wg_key[par], key);
            VI_STAMP(1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores have reached the L2 ...
        }
        VI_STAMP(2);
        __syncthreads();  // ... and so have every other wave's of the workgroup; nobody reads vL or actL any more
        VI_STAMP(3);
        exchange(par, (uint32_t)r + 1u, true, r > 0);
        VI_STAMP(6);
        __syncthreads();
        VI_STAMP(7);
        if (info[3] & 1u) {
            failed = true;
            break;
        }
        if (keeps_deltas && tid == 0) {
#ifndef GU_VI_XCD_STAMPS
            a.vi.delta_key[r] = wg_key[2 + par];
#endif
            wg_key[2 + par] = 0ull;
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {  // V2 (utils.py:55-72) on v'
            const int32_t s = st[j];
            if (s >= 0) {
                double q[4];
#pragma unroll
                for (uint32_t act = 0; act < 4; ++act)
                    q[act] = __dadd_rn(0.0, __dadd_rn((double)(int8_t)(rn[j] >> (8 * act)), __dmul_rn(gamma, vL[vi_next(s, rec[j], act, W)])));
                double qmax = q[0];
#pragma unroll
                for (int act = 1; act < 4; ++act) qmax = (q[act] > qmax) ? q[act] : qmax;
                const uint32_t mask = (rec[j] & GU_CELL_TERM) ? 0u : vi_tie_mask(q, qmax);
                const double share = vi_share(mask);
#pragma unroll
                for (int act = 0; act < 4; ++act) p[j][act] = ((mask >> act) & 1u) ? share : 0.0;
                act_prev[j] = mask ? (uint32_t)__ffs((int)mask) - 1u : 0u;  // an all-zero row (terminal state): argmax = 0
            }
        }
        VI_STAMP(8);
        if (r > 0) agents();
#ifdef GU_VI_XCD_STAMPS
        asm volatile("" ::"v"(e_pos), "v"(e_rew), "v"(e_done), "v"(p[0][0]));  // the round ends here, not wherever its results are needed
#endif
        VI_STAMP(9);
    }
    if (!failed && r > 0) {  // the agents' step of the last round: its action table alone crosses the cluster
        const uint32_t par = (uint32_t)r & 1u;
        uint8_t *axr = ax + (size_t)par * cb;
        if (wave_has_states) {
#pragma unroll
            for (int j = 0; j < K; ++j)
                if (st[j] >= 0) axr[st[j]] = (uint8_t)act_prev[j];
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        exchange(par, (uint32_t)r + 1u, false, true);
        __syncthreads();
        if (info[3] & 1u) failed = true;
        else agents();
    }
#ifdef GU_VI_XCD_STAMPS
    if (keeps_deltas && tid == 0 && a.vi.max_rounds >= 12) {
        stamp_acc[11] = __builtin_amdgcn_s_memrealtime() - stamp_t0;
        for (int i = 0; i < 12; ++i) a.vi.delta_key[i] = stamp_acc[i];
    }
#endif
    // ---- results: one cluster writes the tables, every workgroup its envs ----
    if (writes_tables && !failed && r > 0) {
        double *vf = (r & 1) ? a.vi.v1 : a.vi.v0;  // where `r` swaps of the double-buffered table leave the current values
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int32_t s = st[j];
            if (s >= 0) {
                vf[s] = vL[s];
                *reinterpret_cast<double4 *>(a.vi.pi + 4 * (int64_t)s) = make_double4(p[j][0], p[j][1], p[j][2], p[j][3]);
            }
        }
    }
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

// ------------------------------------------------------------------------------------ host side
// The launch shape for this engine, or false when the per-XCD form does not apply: one grid, table + planes within one
// workgroup's LDS, every workgroup resident (at most one per CU), enough workgroups per XCD under the round-robin placement
// for K <= 2 states per thread.
bool gu_vi_xcd_plan(const gu_engine *h, GuXcdPlan *plan)
{
    if (h->n_grids != 1 || h->S > GU_MAX_LDS_CELLS || h->n_cu < VI_XCD_MAX_XCC) return false;
    const int64_t S2 = ((int64_t)h->S + 1) & ~(int64_t)1;
    const size_t lds = 3 * (size_t)h->cell_bytes + (size_t)S2 * sizeof(double) + 16;  // planes | values | spare slot of the fetch | actions
    if ((int64_t)lds + 1024 > h->lds_per_cu) return false;
    const int64_t forced = gu_opt(h, GU_OPT_VI_XCD_BLOCK);
    const int max_wgs = h->n_cu;
    for (int64_t B = forced ? forced : 256; B <= 1024; B <<= 1) {
        const int64_t env_wgs = (h->N + B - 1) / B;
        if (env_wgs > max_wgs) {
            if (forced) return false;
            continue;
        }
        // enough workgroups for one state per thread in every cluster, as far as the device has CUs for them
        int64_t G = VI_XCD_MAX_XCC * (((int64_t)h->S + B - 1) / B);
        if (G > (max_wgs & ~(VI_XCD_MAX_XCC - 1))) G = max_wgs & ~(VI_XCD_MAX_XCC - 1);
        if (G < env_wgs) G = env_wgs;
        const int64_t per_xcc = G / VI_XCD_MAX_XCC > 0 ? G / VI_XCD_MAX_XCC : 1;  // the smallest cluster under round-robin placement
        if ((G + VI_XCD_MAX_XCC - 1) / VI_XCD_MAX_XCC > VI_XCD_SLOTS) return false;
        const int64_t chunk = ((((int64_t)h->S + per_xcc - 1) / per_xcc) + 63) & ~(int64_t)63;
        const int K = chunk <= B ? 1 : chunk <= 2 * B ? 2 : 0;
        if (K == 0) {
            if (forced) return false;
            continue;
        }
        plan->block = (int)B;
        plan->G = (unsigned)G;
        plan->K = K;
        plan->lds = lds;
        plan->slots_bytes = (size_t)VI_XCD_MAX_XCC * 2 * VI_XCD_SLOTS * 2 * sizeof(vi_u64);
        plan->vx_bytes = (size_t)VI_XCD_MAX_XCC * 2 * (size_t)S2 * sizeof(double);
        plan->ax_bytes = (size_t)VI_XCD_MAX_XCC * 2 * (size_t)h->cell_bytes;
        return true;
    }
    return false;
}

int gu_vi_xcd_launch(gu_engine *h, const GuXcdPlan &plan, const ViStepXcdArgs &a)
{
    static std::atomic<uint64_t> lds_mask[2];
    if (plan.K == 1) {
        gu_allow_lds(gu_vi_sweep_step_xcd_kernel<1>, lds_mask[0], h->device, plan.lds, (size_t)h->lds_per_cu);
        hipLaunchKernelGGL(gu_vi_sweep_step_xcd_kernel<1>, dim3(plan.G), dim3(plan.block), plan.lds, h->stream, a);
    } else {
        gu_allow_lds(gu_vi_sweep_step_xcd_kernel<2>, lds_mask[1], h->device, plan.lds, (size_t)h->lds_per_cu);
        hipLaunchKernelGGL(gu_vi_sweep_step_xcd_kernel<2>, dim3(plan.G), dim3(plan.block), plan.lds, h->stream, a);
    }
    GU_HIP(hipGetLastError());
    return GU_OK;
}
