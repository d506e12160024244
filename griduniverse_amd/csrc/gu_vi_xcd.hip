// gu_vi_xcd.hip -- config 5 for many rounds in ONE launch, synchronised PER XCD instead of chip-wide.
//
// `iters` x { V1 + V2 sweep of the value / policy tables (core/algorithms/utils.py:15-27, 55-72 under
// core/algorithms/dynamic_programming.py:15-20); every agent takes one greedy step on the policy of that round
// (examples/griduniverse_alg_examples.py:76 on env:136-193) }.
//
// gu_vi_sweep_step_cluster_kernel (gu_vi.hip) runs this with ONE copy of the table shared by every workgroup of the chip:
// a round there is V1 -> write-through store of v' -> barrier over all workgroups on 8 XCDs whose L2s are not coherent with
// each other (every shared byte crosses the fabric) -> V2 + agent step (each agent re-deriving its action from four values):
// 5.2 us per round at config 5, of which the arithmetic is < 0.5 us.  Here EVERY XCD sweeps the WHOLE table redundantly (4096
// states at 64x64: nothing next to 65 536 agents) with its own workgroups and steps its own share of the agents, so nothing a
// round needs ever leaves the XCD, and there is NO barrier between workgroups at all -- the data carries its own round number:
//
//   cluster      = the workgroups that read the same HW_REG_XCC_ID; each claims a rank in its cluster at start (ONE returning
//                  atomic per workgroup on a word of eight packed counters + ONE chip-wide wait per launch until the counters
//                  add up to the grid, so that every cluster knows its size).
//                  Membership is what the hardware reports, not a guess from blockIdx: any placement gives correct results,
//                  the observed round-robin placement (blocks b and b + 8 share an XCD) gives equal clusters.
//   state chunk  = cluster of n workgroups, workgroup `rank` owns states [rank * chunk, (rank + 1) * chunk), chunk =
//                  ceil(S / n) rounded up to whole waves; K states per thread at most (a cluster too small for that gives up).
//   what crosses = per round a workgroup needs from the others (a) the new values of ONE GRID ROW either side of its chunk -- a
//                  state's successors are itself, s +- 1 and s +- W -- and (b) the greedy action of EVERY state, which its agents
//                  look up (2 bits per state; an agent re-deriving its action from the values would need the whole float64 table
//                  in every workgroup and ~100 more float64 instructions per wave and round).  The actions of round r are known
//                  only after V2 of round r: they are published right there, tagged for round r + 1 and fetched with its values
//                  -- every agent waits for EVERY member's actions, so they get the length of a V1 as a head start -- and the
//                  agents run ONE ROUND BEHIND the tables (they feed nothing back into them); a last exchange after the loop
//                  delivers the last actions.
//   granules     = every word that crosses carries the round: a value is two 8-byte words {high half | tag}, {low half | tag}
//                  (16 bytes per state), sixteen actions are one word {32 action bits | tag}; tag = round + 1, buffers zeroed
//                  before the launch, double-buffered by round parity.  A consumer loads what it needs (16-byte L1-bypassing
//                  loads, served by the XCD's L2 the stores went to) and looks at the tags: wrong tag = not there yet or half
//                  there, load again.  8-byte words are single-copy atomic, so a word with the right tag IS the word of that
//                  round.  No flag, no drain of the stores, no second hop -- a round costs ONE store-to-load latency.
//   overwriting  = a buffer of parity p is rewritten two rounds later.  To store round r + 2 a workgroup must have finished its
//                  fetch of round r + 1, which contains words EVERY member stored in its own round r + 1, i.e. after that
//                  member's fetch of round r was complete: nobody still wants the words of round r.
//   deltas       = the per-round maximum of v - v' (dynamic_programming.py:17): every lane leaves its key in LDS, a wave reduces them
//                  (two passes of 32-bit DPP maxima) and posts the workgroup's key in a tagged slot (four deep), and workgroup 0 of
//                  the cluster that writes the tables collects the slots TWO rounds late.  Where a workgroup has waves that own no
//                  states (config 5: two of four), those do both, behind barrier 2, ahead of their agents' step -- nothing of it
//                  is on the path store -> exchange -> V2 -> V1 -> store that bounds the round.  The tables alone (no agents): the
//                  stopping rule wants every round's delta in every workgroup.  Each takes in every member's key ONE round late
//                  (asked for ahead of the round's exchange, looked at behind it) and applies the rule before V2; a round that was
//                  one too many is taken back -- its V1 kept the values of the round before in registers, its V2 has not run.
//   waves        = a round of a wave that owns states: V1 from registers, granule stored, key to LDS | barrier 1 | own value to LDS,
//                  exchange | barrier 2 | V2, actions published.  Of a wave that owns none: | barrier 1 | exchange | barrier 2 |
//                  (deltas collected) (key reduced, posted) agents' step.  tools/c5_stamps.py times the phases per wave.
//
// The same float64 operations in the same order per state as every other DP kernel of the library, hence the same bits; the
// agents of all clusters see identical tables, so which cluster steps which env is invisible in the results.  Every spin is
// bounded and raises the launch's fallback word; the caller (gu_vi_sweep_step_run) restores its snapshot and takes the chip-wide
// cluster kernel, then one launch per round.  All workgroups must be resident together: at most one per CU.
#include "gu_internal.hpp"
#include "gu_rng.hpp"
#include "gu_vi.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#define VI_XCD_MAX_XCC 8
#define VI_XCD_SLOTS 64                      /* workgroups of one cluster at most */
#define VI_XCD_GETREG_XCC_ID (20 | (3 << 11)) /* s_getreg_b32 hwreg(HW_REG_XCC_ID, 0, 4) */

// cache policy of the loads that fetch the members' chunks (buffer_load aux: 2 = nt, 16 = sc1, 17 = sc0 sc1; each bypasses this CU's L1)
#ifndef VI_XCD_LOAD_AUX
#define VI_XCD_LOAD_AUX 16
#endif
// ... and of the stores of the value granules (1 = sc0: write-back, the line stays in this XCD's L2)
#ifndef VI_XCD_STORE_AUX
#define VI_XCD_STORE_AUX 1
#endif

typedef uint32_t vi_u32x4 __attribute__((ext_vector_type(4)));

// -DGU_VI_XCD_TORN (a diagnostic variant library, tools/xcd_stress.py; never the product): the exchange relies on each 8-byte half
// {tag | payload} of a 16-byte store landing as ONE unit -- a consumer that sees the right tag takes the payload beside it.  The
// tags catch a half that is not there yet; they cannot catch a half torn at 4 bytes (new tag, old payload).  This build can: the
// tag word carries the round in its low 16 bits and, above them, a check of the payload it was stored with; a word whose tag is
// right and whose check is not is COUNTED (hdr[8], gu_vi_xcd_torn_words) and fetched again.
#ifdef GU_VI_XCD_TORN
__device__ __forceinline__ uint32_t vi_tagw(uint32_t tag, uint32_t payload) { return (tag & 0xFFFFu) | (((payload >> 16) ^ payload ^ tag) << 16); }
__device__ __forceinline__ bool vi_tag_is(uint32_t tagw, uint32_t tag) { return (tagw & 0xFFFFu) == (tag & 0xFFFFu); }
__device__ __forceinline__ bool vi_tag_torn(uint32_t tagw, uint32_t tag, uint32_t payload) { return vi_tag_is(tagw, tag) && tagw != vi_tagw(tag, payload); }
#else
__device__ __forceinline__ uint32_t vi_tagw(uint32_t tag, uint32_t) { return tag; }
__device__ __forceinline__ bool vi_tag_is(uint32_t tagw, uint32_t tag) { return tagw == tag; }
#endif

// 8-byte store that stays in this XCD's L2 (workgroup scope: `global_store_dwordx2 ... sc0`)
__device__ __forceinline__ void vi_st_l2(vi_u64 *p, vi_u64 x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// 8-byte / 4-byte loads that bypass this CU's L1 (`sc1`)
__device__ __forceinline__ vi_u64 vi_ld_l2(const vi_u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t vi_ld_word(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Maximum of a 64-bit key over the wave, in every lane: DPP row shifts inside each row of 16 lanes, then the two row broadcasts of
// the gfx9 family -- register moves inside the SIMD, no LDS round trips -- in two passes of 32-bit halves (the keys order like
// (high word, low word)): v_max_u32 takes the DPP operand itself, so a pass is six instructions, where a 64-bit compare-and-select
// needs ~8 per step.  (A 64-bit __shfl butterfly costs ~700 clocks of dependent ds_bpermute traffic.  An LDS atomicMax per lane is
// turned by the compiler's atomic optimizer into a scalar loop over the active lanes at ~100 clocks per lane: 7300 clocks for a
// full wave; a hand-written ds_max_u64 of 64 lanes on one word holds the LDS for ~4 clocks per lane.)
__device__ __forceinline__ uint32_t vi_wave_max32(uint32_t x)
{
#define VI_DPP_STEP(ctrl, rows)                                                                  \
    {                                                                                            \
        const uint32_t o_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, ctrl, rows, 0xF, false); \
        x = o_ > x ? o_ : x;                                                                     \
    }
    VI_DPP_STEP(0x111, 0xF)  // row_shr:1 (a lane without a source reads 0)
    VI_DPP_STEP(0x112, 0xF)  // row_shr:2
    VI_DPP_STEP(0x114, 0xF)  // row_shr:4
    VI_DPP_STEP(0x118, 0xF)  // row_shr:8      -> lane 15 of every row holds the row's maximum
    VI_DPP_STEP(0x142, 0xA)  // row_bcast:15   -> rows 1 and 3 take in the last lane of rows 0 and 2
    VI_DPP_STEP(0x143, 0xC)  // row_bcast:31   -> rows 2 and 3 take in lane 31: lane 63 holds the wave's
#undef VI_DPP_STEP
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}
__device__ __forceinline__ vi_u64 vi_wave_max_all(vi_u64 k)
{
    const uint32_t hi = (uint32_t)(k >> 32), top = vi_wave_max32(hi);
    return ((vi_u64)top << 32) | vi_wave_max32(hi == top ? (uint32_t)k : 0u);
}

// -DGU_VI_XCD_STAMPS (a diagnostic variant library, tools/c5_stamps.py; never the product): ONE wave of one member of the cluster
// that writes the tables (GU_VI_STAMP_RANK, GU_VI_STAMP_WAVE in the environment at launch) sums the shader-clock cycles it spends
// in every phase of a round and returns the sums IN PLACE OF the first deltas (delta_key[0 .. 11]; [10] = poll turns, [11] =
// 100 MHz ticks of the whole loop).
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
// AGENTS: config 5, the loop described above.  !AGENTS: the tables alone -- gu_vi_sweep / gu_vi_run / gu_vi_eval_run on grids whose
// planes and values fit one workgroup's LDS: the workgroups of ONE cluster (the XCD of workgroup 0; the others leave after the
// registration) run { V1; delta; V2 if GREEDY } with the stopping rule of value_iteration / of policy_iteration's evaluation loop
// (dynamic_programming.py:14-27, 40-42) decided in the kernel: every workgroup waits for every member's delta key of the round --
// posted with the round's values, fetched beside them -- so all of them stop behind the same round.
template <int K, bool AGENTS, bool GREEDY, int NB>
__global__ void __launch_bounds__(1024) gu_vi_xcd_kernel(const ViStepXcdArgs a)
{
    static_assert(!AGENTS || GREEDY, "the agents follow the greedy policy of the round");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ vi_u64 round_key_lds;  // !AGENTS: the cluster's delta key of the round
    __shared__ uint32_t info[4];  // [0] XCC id, [1] rank in the cluster, [2] members, [3] bit 0: failed, bit 1: this cluster writes the tables
#ifdef GU_VI_XCD_STAMPS
    uint64_t life[7];  // 100 MHz ticks: entry, planes staged, registered, loop begins, loop ends, last exchange done, results written
    life[0] = __builtin_amdgcn_s_memrealtime();
#endif
    const ViMap cell = vi_stage<true>(a.vi.cell, a.vi.cell_bytes, smem);  // the agents gather records of arbitrary cells
#ifdef GU_VI_XCD_STAMPS
    life[1] = __builtin_amdgcn_s_memrealtime();
#endif
    double *vL_window = reinterpret_cast<double *>(smem + 2 * a.vi.cell_bytes);  // [a.lds_values] the values of the own chunk and of one grid row either side: all a workgroup ever reads
    const int32_t tid = threadIdx.x, B = blockDim.x, S = a.vi.S, W = a.vi.W;
    const int32_t cb = a.vi.cell_bytes;
    const double gamma = a.vi.gamma;
    uint32_t *hdr = a.vi.sync;  // [1] fallback word, [2] rounds done, [4 .. 5] the registration word (below)

    // ---- registration: who shares this XCD ----
    if (tid == 0) {
        const uint32_t xcc = __builtin_amdgcn_s_getreg(VI_XCD_GETREG_XCC_ID);
        uint32_t bad = xcc >= VI_XCD_MAX_XCC;
        // ONE returning atomic on ONE 64-bit word does the claim, the arrival and the leader's mark: eight 7-bit counters of
        // workgroups per XCC (at most one workgroup per CU: <= 32) and, in bits 56 .. 59, 1 + the XCC of workgroup 0.  The word
        // read back until its counters add up to the grid is all a workgroup needs to know -- its rank came with the claim.
        // (Separate words for claim, arrival and leader cost a second atomic behind a wait for the first, and two more loads
        // behind the poll: 8 us of a launch, all of it round trips through memory across the XCDs.)
        vi_u64 *reg = reinterpret_cast<vi_u64 *>(hdr + 4);
        const uint32_t field = 7u * (xcc & (VI_XCD_MAX_XCC - 1));
        const vi_u64 mine = ((vi_u64)1 << field) | (blockIdx.x == 0 ? (vi_u64)((xcc & 7u) + 1u) << 56 : 0ull);
        vi_u64 seen = __hip_atomic_fetch_add(reg, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t rank = (uint32_t)(seen >> field) & 0x7Fu;
        seen += mine;  // (the word as this workgroup left it: the last one to arrive needs no further look)
        auto arrived = [](vi_u64 w) {
            uint32_t n = 0;
            for (int k = 0; k < VI_XCD_MAX_XCC; ++k) n += (uint32_t)(w >> (7 * k)) & 0x7Fu;
            return n;
        };
        // (a seven-bit counter holds 127: a cluster of more than VI_XCD_SLOTS workgroups -- no MI355X partition has one -- gives the
        // launch up BEFORE a carry can reach the neighbouring counter and leave everybody waiting for a sum that never comes)
        if (rank >= VI_XCD_SLOTS) {
            bad = 1u;
            __hip_atomic_store(hdr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        uint32_t spins = 0;
        while (!bad && arrived(seen) < gridDim.x) {  // the one chip-wide wait of the launch
            __builtin_amdgcn_s_sleep(1);
            seen = vi_ld_l2(reg);
            if (++spins > VI_CL_SPIN_LIMIT || ((spins & 7u) == 0u && vi_ld_word(hdr + 1))) bad = 1u;
        }
        uint32_t members = 0, writes = 0;
        if (!bad) {
            members = (uint32_t)(seen >> field) & 0x7Fu;
            writes = ((uint32_t)(seen >> 56) & 0xFu) == xcc + 1u;
            const int64_t chunk = ((((int64_t)S + members - 1) / members) + 63) & ~(int64_t)63;
            // what one thread's fetch items must cover: the halo granules, and -- with agents -- the action items (the same sum
            // gu_vi_xcd_plan sized NB for; round 4 counted the action items for the tables alone too and sent 125x32, 100x40,
            // 1024x5 .. through a failed launch and a restore to the older forms)
            const int64_t items = 2 * (int64_t)(W < S ? W : S) + (AGENTS ? (((S + 15) >> 4) + 1) / 2 : 0);
            // the tables alone run on ONE cluster: the shape of the others (whose members leave right behind the registration) is
            // nobody's business, and must not veto the launch
            bad = (AGENTS || writes) && (rank >= VI_XCD_SLOTS || members > VI_XCD_SLOTS || chunk > (int64_t)K * B || items > (int64_t)NB * B ||
                                         chunk + 2 * (int64_t)(W < S ? W : S) > (int64_t)a.lds_values);
        }
        if (bad || (a.inject_failure & 1u)) {
            bad = 1u;
            __hip_atomic_store(hdr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // tell everyone; never hang
        }
        info[0] = xcc & (VI_XCD_MAX_XCC - 1);
        info[1] = rank;
        info[2] = members ? members : 1u;
        info[3] = bad | (writes << 1);
    }
    __syncthreads();
#ifdef GU_VI_XCD_STAMPS
    life[2] = __builtin_amdgcn_s_memrealtime();
#endif
    const uint32_t xcc = info[0], rank = info[1], members = info[2];
    const bool writes_tables = (info[3] & 2u) != 0;
    bool failed = (info[3] & 1u) != 0;
    const int32_t chunk = (int32_t)(((((int64_t)S + members - 1) / members) + 63) & ~(int64_t)63);
    vi_u64 *slots = a.slots + (size_t)xcc * 4 * VI_XCD_SLOTS * 2;   // [round & 3][member][2]
    uint8_t *gx = a.gx + (size_t)xcc * a.work_bytes;                // this cluster's granules: [2][S] values | [2][n_aw] action items

    if (!AGENTS && !writes_tables) return;  // the tables need ONE cluster (every member of the others decides the same: no one waits for them)
    const int32_t lo = (int32_t)rank * chunk, hi = lo + chunk < S ? lo + chunk : S;  // this workgroup's states [lo, hi)

    // ---- per-state constants, the initial table, the own envs ----
    int32_t st[K];    // the thread's states (-1: none)
    uint32_t rec[K];
    double rn[K][4];  // the rewards of a state's four successors (integers, as doubles: converted once)
    int32_t r_own[K];
    double p[K][4];
#pragma unroll
    for (int j = 0; j < K; ++j) {
        const int32_t local = tid + j * B;
        const int32_t s = (int32_t)rank * chunk + local;
        st[j] = (!failed && local < chunk && s < S) ? s : -1;
        rec[j] = 0u;
#pragma unroll
        for (int act = 0; act < 4; ++act) rn[j][act] = 0.0;
        r_own[j] = 0;
#pragma unroll
        for (int act = 0; act < 4; ++act) p[j][act] = 0.0;
        if (st[j] >= 0) {
            rec[j] = cell.f[s];
            r_own[j] = cell.r[s];
#pragma unroll
            for (uint32_t act = 0; act < 4; ++act) rn[j][act] = (double)cell.r[vi_next(s, rec[j], act, W)];
            const double4 row = *reinterpret_cast<const double4 *>(a.vi.pi + 4 * (int64_t)s);
            p[j][0] = row.x, p[j][1] = row.y, p[j][2] = row.z, p[j][3] = row.w;
        }
    }
    // the value window: states [vbase, vbase + a.lds_values) -- the chunk and W states either side; vL[state] as if the table were whole
    const int32_t vbase = lo - W > 0 ? lo - W : 0;
    double *vL = vL_window - vbase;
    {
        const int32_t top = lo < S ? (hi + W < S ? hi + W : S) : vbase;
        for (int32_t i = vbase + tid; i < top; i += B) vL[i] = a.vi.v0[i];
    }
    // Division of labour inside the workgroup: the waves that own states carry the round's critical path (V1, V2).  When at
    // most half of the waves do, they carry NO agents: every other wave steps two blocks of 64 envs (two independent gather
    // chains that interleave) and one of them reduces the workgroup's delta keys -- both beside the critical path, not on it.
    const int32_t waves = B >> 6, wave = tid >> 6, lane = tid & 63;
    const int32_t own_states = lo < S ? (chunk < S - lo ? chunk : S - lo) : 0;                // states of this workgroup
    const int32_t state_waves = K == 1 ? (own_states + 63) >> 6 : (own_states > 0 ? waves : 0);
    const bool split = K == 1 && 2 * state_waves <= waves;
    const bool envs_here = AGENTS && (!split || wave >= state_waves);  // (wave-uniform)
    int64_t gid[2];
    bool own_env[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int32_t block = split ? 2 * (wave - state_waves) + e : wave;  // which 64 envs of the workgroup's B
        const bool mine = AGENTS && (split ? (wave >= state_waves && block < waves) : e == 0);
        gid[e] = (int64_t)blockIdx.x * B + (int64_t)block * 64 + lane;
        own_env[e] = mine && gid[e] < a.N;
    }
    int32_t e_pos[2] = {0, 0}, e_rew[2] = {0, 0}, e_done[2] = {0, 0};
    uint32_t e_ep[2] = {0u, 0u}, e_prefix[2] = {0u, 0u};
#pragma unroll
    for (int e = 0; e < 2; ++e)
        if (own_env[e]) {
            e_pos[e] = a.pos[gid[e]];
            e_rew[e] = a.reward[gid[e]];
            e_done[e] = a.done[gid[e]];
            e_ep[e] = a.episode[gid[e]];
            e_prefix[e] = gu_rng_prefix(a.seed_prefix, a.env_id0 + (uint32_t)gid[e]);
        }
    __syncthreads();

    bool any_states = false;  // (wave-uniform: the states of a chunk are dealt to whole waves)
#pragma unroll
    for (int j = 0; j < K; ++j) any_states = any_states || __any(st[j] >= 0);
    vi_u64 *lane_key = reinterpret_cast<vi_u64 *>(smem + ((2u * (uint32_t)cb + a.lds_values * 8u + 16u + (uint32_t)((((S + 15) >> 4) + 1) >> 1) * 8u + 15u) & ~15u));  // [2][B] the lanes' delta keys, by round parity
    for (int32_t i = tid; i < 2 * B; i += B) lane_key[i] = 0ull;       // (a lane without a state never writes its entry)
    const int32_t key_wave = split ? state_waves : 0;                 // the wave that reduces them
    __syncthreads();
#ifdef GU_VI_XCD_STAMPS
    uint64_t stamp_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t stamp_last = __builtin_amdgcn_s_memtime();
    const uint64_t stamp_t0 = __builtin_amdgcn_s_memrealtime();
    life[3] = stamp_t0;
#endif
    const bool keeps_deltas = writes_tables && rank == 0;
    // The tables alone: every workgroup takes in every member's delta key of the round BEFORE every round -- the stopping rule needs
    // it, and it is what keeps the members within a round of each other (their values only tie neighbours together: without it a
    // far member could run a dozen rounds ahead and overwrite its four-deep key slots before they are collected).  With agents the
    // action words tie every workgroup to every other one anyway, and workgroup 0 collects the keys two rounds late.
    const bool sync_delta = !AGENTS;
    // What a round fetches, in 16-byte items: the value granules of one grid row either side of the own chunk (the chunk's own
    // values go from registers to LDS), then the action words, two per item.
    const int32_t below0 = lo - W > 0 ? lo - W : 0, n_below = lo < S ? lo - below0 : 0;    // states [below0, lo)
    const int32_t n_above = lo < S ? (hi + W < S ? hi + W : S) - hi : 0;                     // states [hi, hi + n_above)
    const int32_t n_words = (S + 15) >> 4, n_aw = (n_words + 1) >> 1;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void *)gx, 0, a.work_bytes, 0x00020000);
    const uint32_t lds_v = 2u * (uint32_t)cb, lds_spare = lds_v + a.lds_values * 8u, lds_a = lds_spare + 16u;
    const uint32_t gv_bytes = (uint32_t)S * 16u, aw_off = 2u * gv_bytes, aw_bytes = (uint32_t)n_aw * 16u;  // [2][S] granules | [2][n_aw] action items
    // The items this thread fetches are the same every round: source (parity 0), destination in LDS and kind are worked out ONCE
    // (computed inside the loop, the selects and bounds of this bookkeeping were three quarters of the fetch's 1900 clocks).
    // kind: 0 = none, 1 = value granule, 2 = action item with two words, 3 = action item whose second word does not exist.
    const int32_t n_items = n_below + n_above + (AGENTS ? n_aw : 0);
    uint32_t it_src[NB], it_dst[NB], it_kind[NB];
#pragma unroll
    for (int m = 0; m < NB; ++m) {
        const int32_t x = m * B + tid, n_v = n_below + n_above;
        const int32_t s = x < n_below ? below0 + x : hi + (x - n_below);  // the state of a value item
        const int32_t w = x - n_v;                                        // the index of an action item
        it_kind[m] = x >= n_items ? 0u : x < n_v ? 1u : (2 * w + 1 < n_words) ? 2u : 3u;
        it_src[m] = it_kind[m] == 1u ? (uint32_t)s * 16u : aw_off + (uint32_t)w * 16u;
        it_dst[m] = it_kind[m] == 0u ? lds_spare : it_kind[m] == 1u ? lds_v + (uint32_t)(s - vbase) * 8u : lds_a + (uint32_t)w * 8u;
    }
    // One exchange: fetch the halo granules (with_v) and the action words (with_act) tagged `tag` from the buffers of parity
    // `par`, reloading until every tag is there.  Straight-line, unpredicated loads: a load or an LDS write inside a divergent
    // branch makes the compiler wait for every load in flight at the branch (a first version ran its eight loads ONE AFTER THE
    // OTHER, 450 clocks each).  A lane without an item loads from beyond the buffer's size (the bounds check returns zeros
    // without a memory access) and writes to a spare 16 bytes behind the table.
    // NB = 16-byte loads per thread: 1 at config 5 (the host picks the instance: 1 where the items fit the workgroup, else 4)
    auto fetch = [&](uint32_t par, uint32_t tag, bool with_v, bool with_act, auto &&meanwhile) {
        uint32_t src[NB];
        bool on[NB];
#pragma unroll
        for (int m = 0; m < NB; ++m) {
            on[m] = it_kind[m] == 1u ? with_v : (it_kind[m] != 0u && with_act);
            src[m] = on[m] ? it_src[m] + par * (it_kind[m] == 1u ? gv_bytes : aw_bytes) : 0xFFFFFFF0u;
        }
        vi_u32x4 t[NB];
        uint32_t spins = 0, bad = 0u;
#pragma unroll
        for (int m = 0; m < NB; ++m) t[m] = __builtin_amdgcn_raw_buffer_load_b128(rs, src[m], 0, VI_XCD_LOAD_AUX);
        meanwhile();  // (work of this wave that does not depend on the exchange runs under the loads' round trip)
        VI_STAMP(2);
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int m = 0; m < NB; ++m) ok = ok && (!on[m] || (vi_tag_is(t[m].x, tag) && (vi_tag_is(t[m].z, tag) || it_kind[m] == 3u)));
#ifdef GU_VI_XCD_TORN
#pragma unroll
            for (int m = 0; m < NB; ++m) {  // right tag, wrong payload: a half torn at four bytes -- counted, and fetched again
                const uint32_t torn = on[m] ? (uint32_t)vi_tag_torn(t[m].x, tag, t[m].y) + (uint32_t)(it_kind[m] != 3u && vi_tag_torn(t[m].z, tag, t[m].w)) : 0u;
                if (torn) {
                    atomicAdd(hdr + 8, torn);
                    ok = false;
                }
            }
#endif
            if (__builtin_expect(__all(ok) != 0, 1)) break;
            if (++spins > VI_CL_SPIN_LIMIT || ((spins & 255u) == 0u && vi_ld_word(hdr + 1))) {
                bad = 1u;
                break;
            }
            if (spins > 8u) __builtin_amdgcn_s_sleep(1);
#pragma unroll
            for (int m = 0; m < NB; ++m) t[m] = __builtin_amdgcn_raw_buffer_load_b128(rs, src[m], 0, VI_XCD_LOAD_AUX);
        }
#ifdef GU_VI_XCD_STAMPS
        stamp_acc[10] += spins;
        asm volatile("" ::"v"(t[0].x));
#endif
        VI_STAMP(4);
        // a value granule {tag, high half} {tag, low half} becomes one double, an action item two 32-bit words of 16 actions
#pragma unroll
        for (int m = 0; m < NB; ++m) {
            const bool is_v = it_kind[m] == 1u;
            *reinterpret_cast<uint2 *>(smem + (on[m] ? it_dst[m] : lds_spare)) = make_uint2(is_v ? t[m].w : t[m].y, is_v ? t[m].y : t[m].w);
        }
        if (bad && lane == 0) {
            __hip_atomic_store(hdr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            atomicOr(&info[3], 1u);
        }
    };
    // this wave's sixteen-action words of parity `par`: lanes 0 .. 3 assemble them from two ballots and store them
    auto publish_actions = [&](uint32_t par, uint32_t tag, const uint32_t act[K]) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int32_t s_first = lo + (wave << 6) + j * B;  // the state of lane 0 (chunks and waves are whole multiples of 64)
            const uint64_t b0 = __ballot((act[j] & 1u) != 0u), b1 = __ballot((act[j] & 2u) != 0u);
            if (lane < 4 && (wave << 6) + j * B < chunk && s_first + 16 * lane < S) {
                // sixteen states' low action bits in the word's low half, their high bits in its high half
                const uint32_t e = (uint32_t)(b0 >> (16 * lane)) & 0xFFFFu, o = (uint32_t)(b1 >> (16 * lane)) & 0xFFFFu;
                const uint32_t word = (uint32_t)(s_first >> 4) + (uint32_t)lane;
                vi_st_l2(reinterpret_cast<vi_u64 *>(gx + aw_off + par * aw_bytes) + word, ((vi_u64)(e | (o << 16)) << 32) | vi_tagw(tag, e | (o << 16)));
            }
        }
    };
    // every agent takes the step of the round before: the greedy action of its cell (np.argmax of its policy row: the first
    // maximum; examples/griduniverse_alg_examples.py:76) is in the action table the state's owner published
    const uint32_t *actL = reinterpret_cast<const uint32_t *>(smem + lds_a);
    auto agents = [&]() {
#pragma unroll
        for (int e = 0; e < 2; ++e)
            if (own_env[e] && (a.flags & GU_F_AUTO_RESET) && e_done[e]) {  // lazy `if done: env.reset()`
                e_pos[e] = a.starts[gu_rng_start_index(e_prefix[e], e_ep[e], a.n_starts)];
                ++e_ep[e];
            }
        // (unpredicated: a lane without an env walks cell 0; two independent chains of three dependent LDS gathers)
        uint32_t act[2], rec_e[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const uint32_t w = actL[e_pos[e] >> 4] >> (e_pos[e] & 15);
            act[e] = (w & 1u) | ((w >> 15) & 2u), rec_e[e] = cell.f[e_pos[e]];
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) e_pos[e] = vi_next(e_pos[e], rec_e[e], act[e], W);
#pragma unroll
        for (int e = 0; e < 2; ++e) e_rew[e] = cell.r[e_pos[e]], e_done[e] = (cell.f[e_pos[e]] >> GU_CELL_TERM_BIT) & 1;
    };
    // workgroup 0 of the writing cluster: the delta of round `rr` = the maximum of the members' keys, from their slots.  The slots
    // are LOADED before the round's fetch and looked at behind it, so that the two round trips overlap.
    vi_u64 d_hi = 0ull, d_lo = 0ull;
    auto delta_load = [&](int32_t rr) {
        const vi_u64 *slot = slots + (size_t)((uint32_t)rr & 3u) * VI_XCD_SLOTS * 2;
        if ((uint32_t)lane < members) {
            d_hi = vi_ld_l2(slot + 2 * lane);
            d_lo = vi_ld_l2(slot + 2 * lane + 1);
        }
    };
    auto delta_finish = [&](int32_t rr) -> vi_u64 {
        const uint32_t tag = a.tag0 + (uint32_t)rr + 1u;
        const bool polls = (uint32_t)lane < members;
        uint32_t spins = 0;
        // (normally there at the first look: the slots of round rr were posted before anything of round rr + 1 was stored)
        while (!__all(!polls || ((uint32_t)d_hi == tag && (uint32_t)d_lo == tag))) {
            if (++spins > VI_CL_SPIN_LIMIT || ((spins & 255u) == 0u && vi_ld_word(hdr + 1))) {
                if (lane == 0) {
                    __hip_atomic_store(hdr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    atomicOr(&info[3], 1u);
                }
                return 0ull;
            }
            __builtin_amdgcn_s_sleep(1);
            delta_load(rr);
        }
        return vi_wave_max_all(polls ? (d_hi & 0xFFFFFFFF00000000ull) | (d_lo >> 32) : 0ull);
    };
    // the workgroup's delta key of round `rr`: the maximum of its lanes' keys in LDS (in every lane), and its post
    auto reduce_keys = [&](int32_t rr) -> vi_u64 {
        vi_u64 m = 0ull;
        if (K == 1 && state_waves <= 2) {  // (both reads in flight at once)
            const vi_u64 k0 = lane_key[(rr & 1) * B + lane], k1 = state_waves > 1 ? lane_key[(rr & 1) * B + 64 + lane] : 0ull;
            m = k0 > k1 ? k0 : k1;
        } else {
            for (int32_t i = lane; i < (K == 1 ? state_waves * 64 : B); i += 64) {
                const vi_u64 k = lane_key[(rr & 1) * B + i];
                m = k > m ? k : m;
            }
        }
        return vi_wave_max_all(m);
    };
    auto post = [&](int32_t rr, vi_u64 m) {
        if (lane == 0) {
            const uint32_t tag = a.tag0 + (uint32_t)rr + 1u;
            vi_u64 *slot = slots + (size_t)((uint32_t)rr & 3u) * VI_XCD_SLOTS * 2;
            vi_st_l2(slot + 2 * rank, (m & 0xFFFFFFFF00000000ull) | tag);
            vi_st_l2(slot + 2 * rank + 1, (m << 32) | tag);
        }
    };
    auto collect = [&](int32_t rr) {
        const vi_u64 k = delta_finish(rr);
#ifndef GU_VI_XCD_STAMPS
        if (lane == 0) a.vi.delta_key[rr] = k;
#else
        asm volatile("" ::"v"(k));
#endif
    };
    const bool late = split;                             // this workgroup has waves that own no states
    const int32_t collect_wave = late ? waves - 1 : 0;  // (the key wave is the first of them)
    uint32_t act_prev[K];  // greedy action of the thread's states under the policy of the round before
    double v_new[K];       // the states' current values
    double v_prev[K];      // ... and those of the round before (the tables alone: a round is taken back when the stopping rule says so)
    double gv[K][4];       // gamma * value of the four successors
#pragma unroll
    for (int j = 0; j < K; ++j) {
        act_prev[j] = 0u;
        v_prev[j] = 0.0;
        v_new[j] = st[j] >= 0 ? vL[st[j]] : 0.0;
#pragma unroll
        for (uint32_t act = 0; act < 4; ++act) gv[j][act] = st[j] >= 0 ? __dmul_rn(gamma, vL[vi_next(st[j], rec[j], act, W)]) : 0.0;
    }
    // One round.  Its code exists once per ROLE of a wave, so that a wave passes none of the other roles' work: with one wave per SIMD
    // a taken branch costs ~60 clocks (measured, tools/c5_stamps.py), and a wave that owns states skipped eleven blocks per round.
    // ROLE 1: a wave that owns states in a workgroup where other waves do everything else; 2: one of those other waves; 0: any wave
    // of a workgroup whose waves all own states.
    int r = 0;
    auto round = [&](auto role) -> bool {
        constexpr int ROLE = decltype(role)::value;
        const bool wave_has_states = ROLE == 1 ? true : ROLE == 2 ? false : any_states;
        const bool has_envs = ROLE == 1 ? false : envs_here;
        const uint32_t par = (uint32_t)r & 1u, tag = a.tag0 + (uint32_t)r + 1u;
        if (wave_has_states) {
            vi_u64 key = 0ull;
#pragma unroll
            for (int j = 0; j < K; ++j) {  // V1 (utils.py:15-27)
                const int32_t s = st[j];
                if (s >= 0) {
                    // (0.0 + R == R: an integer converted to double is never -0.  gamma * v[next] was formed by V2 of the round
                    // before -- same table, same successor -- and is still in registers: V1 reads nothing from LDS)
                    double acc = (double)r_own[j];
#pragma unroll
                    for (uint32_t act = 0; act < 4; ++act) acc = __dadd_rn(acc, __dmul_rn(p[j][act], gv[j][act]));
                    const double v_old = v_new[j];
                    v_prev[j] = v_old;
                    v_new[j] = acc;
                    const vi_u64 bits = (vi_u64)__double_as_longlong(acc);
                    vi_u32x4 g;
                    g.x = vi_tagw(tag, (uint32_t)(bits >> 32)), g.y = (uint32_t)(bits >> 32), g.z = vi_tagw(tag, (uint32_t)bits), g.w = (uint32_t)bits;
                    __builtin_amdgcn_raw_buffer_store_b128(g, rs, par * gv_bytes + (uint32_t)s * 16u, 0, VI_XCD_STORE_AUX);
                    const vi_u64 k = vi_key(__dsub_rn(v_old, acc));  // signed, dynamic_programming.py:17
                    key = k > key ? k : key;
                }
            }
            VI_STAMP(0);
            // the thread's delta key goes to LDS as it is; the workgroup's maximum is taken behind the barrier, under the exchange's
            // round trip, by a wave that owns no states where there is one
            lane_key[(r & 1) * B + tid] = key;
            VI_STAMP(1);
        }
        __syncthreads();  // nobody reads the old values or actions in LDS any more; the lanes' keys are in LDS
        VI_STAMP(3);
#pragma unroll
        for (int j = 0; j < K; ++j)
            if (st[j] >= 0) vL[st[j]] = v_new[j];
        // (two rounds late, and where there is one by a wave that owns no states, behind barrier 2: off everybody's critical path)
        const bool collects = ROLE != 1 && !sync_delta && keeps_deltas && wave == collect_wave && r > 1;
        if (collects) delta_load(r - 2);
        // The workgroup's key: a wave without states reduces and posts it behind barrier 2, where it has time to spare; a wave that
        // owns states too reduces it under the exchange's round trip and posts it BEHIND the exchange (a store ahead of it would
        // turn the wait for the loads into a wait for the store's acknowledgement as well).  With agents workgroup 0 collects the
        // keys two rounds late.  The tables alone: EVERY workgroup takes in every member's key ONE round late -- asked for ahead
        // of this round's exchange, looked at behind it -- and the round that turns out to be one too many is taken back (below).
        const bool posts = ROLE != 1 && wave == key_wave, posts_late = late;
        const bool takes_in = sync_delta && posts && r > 0;
        if (takes_in) delta_load(r - 1);
        vi_u64 mine = 0ull;
        fetch(par, tag, true, AGENTS && r > 0, [&]() {
            if (!posts || posts_late) return;
            mine = reduce_keys(r);
        });
        if (posts && !posts_late) post(r, mine);
        VI_STAMP(6);
        if (takes_in) {  // the delta of the round before, for everybody
            const vi_u64 k = delta_finish(r - 1);
            if (lane == 0) round_key_lds = k;
        }
        if (collects && !late) collect(r - 2);
        VI_STAMP(5);
        __syncthreads();
        VI_STAMP(7);
        if (info[3] & 1u) {
            failed = true;
            return false;
        }
        if (sync_delta && r > 0) {
            // The stopping rule (dynamic_programming.py:22-23 / :42) on the delta of round r - 1, which took this round to cross the
            // cluster: if it says stop, round r was one too many.  Its V1 is taken back (the values of the round before are kept
            // beside the new ones; V2 of round r has not run, so the policy is still that of round r - 1) and r rounds count.
            const vi_u64 round_key = round_key_lds;
            if (keeps_deltas && tid == 0) a.vi.delta_key[r - 1] = round_key;
            if (a.vi.use_threshold && vi_unkey_dev(round_key) < a.vi.threshold) {
#pragma unroll
                for (int j = 0; j < K; ++j) v_new[j] = v_prev[j];
                return false;
            }
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {  // V2 (utils.py:55-72) on v'
            const int32_t s = st[j];
            if (s >= 0) {
                // q[a] = 0.0 + (R[next] + gamma * v'[next]) without the `0.0 +`: the sum of a non-zero integer and anything is never
                // -0, so the addition is the identity.  Ties of np.around(q, 8) through k = rint(q * 1e8) as in vi_tie_mask; the k of
                // the largest q is the largest k (x -> rint(x * 1e8) is monotone), so the maximum is taken over the k and the chain
                // of compares and selects over q runs only where the divisions decide (|q| >= 3.3e7, or NaN).
                double q[4], k[4];
#pragma unroll
                for (uint32_t act = 0; act < 4; ++act) {
                    gv[j][act] = __dmul_rn(gamma, vL[vi_next(s, rec[j], act, W)]);
                    q[act] = __dadd_rn(rn[j][act], gv[j][act]);
                    k[act] = rint(__dmul_rn(q[act], 100000000.0));
                }
                if (GREEDY) {  // (!GREEDY: the evaluation sweeps of policy_iteration keep the policy; V1 still wants gamma * v'[next])
                    const double kmax = fmax(fmax(k[0], k[1]), fmax(k[2], k[3]));
                    uint32_t mask = 0u, top = 0u;
#pragma unroll
                    for (int act = 0; act < 4; ++act) {
                        mask |= (uint32_t)(k[act] == kmax) << act;
                        const uint32_t h = (uint32_t)((vi_u64)__double_as_longlong(k[act]) >> 32) & 0x7FFFFFFFu;
                        top = h > top ? h : top;  // (sign off, exponent and the top of the mantissa: orders like |k|, NaN and inf on top)
                    }
                    // |k| >= 2^25 * 1e8 = 3355443200000000.0 = 0x4327D784_00000000, or NaN (proof: gu_vi.hpp, vi_tie_mask), told from
                    // the high words alone -- four integer instructions instead of seven double-rate ones
                    if (__builtin_expect(top >= 0x4327D784u, 0)) {  // (unlikely: laid out of line, the usual path falls through)
                        double qmax = q[0];
#pragma unroll
                        for (int act = 1; act < 4; ++act) qmax = (q[act] > qmax) ? q[act] : qmax;
                        mask = vi_tie_mask(q, qmax);
                    }
                    if (rec[j] & GU_CELL_TERM) mask = 0u;
                    const double share = vi_share(mask);
#pragma unroll
                    for (int act = 0; act < 4; ++act) p[j][act] = ((mask >> act) & 1u) ? share : 0.0;
                    act_prev[j] = mask ? (uint32_t)__ffs((int)mask) - 1u : 0u;  // an all-zero row (terminal state): argmax = 0
                }
            }
        }
        // the actions of this round leave before the values of the next: every agent of the cluster waits for every member's
        // actions, and a word that has a V1 of head start is there when it is asked for
        if (AGENTS && wave_has_states) publish_actions(par ^ 1u, tag + 1u, act_prev);
        VI_STAMP(8);
        // (ahead of the agents' step, whose LDS round trips then cover the stores' acknowledgements -- the next barrier waits for them;
        // the lanes' keys of this round stay in LDS until V1 of the round after next)
        if (collects && late) collect(r - 2);
        if (posts && posts_late) post(r, reduce_keys(r));
        if (AGENTS && r > 0 && has_envs) agents();
#ifdef GU_VI_XCD_STAMPS
        asm volatile("" ::"v"(e_pos[0]), "v"(e_rew[0]), "v"(e_done[1]), "v"(p[0][0]));  // the round ends here, not wherever its results are needed
#endif
        VI_STAMP(9);
        return true;
    };
    if (!split) {
        for (; r < a.vi.max_rounds && !failed; ++r)
            if (!round(std::integral_constant<int, 0>{})) break;
    } else if (wave < state_waves) {
        for (; r < a.vi.max_rounds && !failed; ++r)
            if (!round(std::integral_constant<int, 1>{})) break;
    } else {
        for (; r < a.vi.max_rounds && !failed; ++r)
            if (!round(std::integral_constant<int, 2>{})) break;
    }
#ifdef GU_VI_XCD_STAMPS
    life[4] = __builtin_amdgcn_s_memrealtime();
#endif
    if (!AGENTS && !failed && r > 0 && r == a.vi.max_rounds && keeps_deltas) {  // the tables alone, not stopped: the last round's delta is still out
        __syncthreads();
        if (wave == key_wave) {
            delta_load(r - 1);
            collect(r - 1);
        }
    }
    if (AGENTS && !failed && r > 0) {  // the agents' step of the last round: its actions alone cross the cluster
        const uint32_t par = (uint32_t)r & 1u, tag = a.tag0 + (uint32_t)r + 1u;  // (published behind V2 of the last round)
        __syncthreads();
        const bool collects = keeps_deltas && wave == collect_wave;
        if (collects && r > 1) delta_load(r - 2);
        fetch(par, tag, false, true, []() {});
        if (collects) {  // the last two deltas
            if (r > 1) collect(r - 2);
            delta_load(r - 1);
            collect(r - 1);
        }
        __syncthreads();
        if (info[3] & 1u) failed = true;
        else if (envs_here) agents();
    }
#ifdef GU_VI_XCD_STAMPS
    life[5] = __builtin_amdgcn_s_memrealtime();
    // (which wave of which member is stamped: GU_VI_STAMP_WAVE / GU_VI_STAMP_RANK in the environment of the variant library)
    const bool stamps_here = writes_tables && rank == ((a.inject_failure >> 16) & 0xFFu) && tid == (int32_t)((a.inject_failure >> 8) & 0xFFu) * 64;
    if (writes_tables && rank == ((a.inject_failure >> 16) & 0xFFu) && tid == (int32_t)((a.inject_failure >> 8) & 0xFFu) * 64 && a.vi.max_rounds >= 12) {
        stamp_acc[11] = __builtin_amdgcn_s_memrealtime() - stamp_t0;
        for (int i = 0; i < 12; ++i) a.vi.delta_key[i] = stamp_acc[i];
    }
#endif
    // ---- results: one cluster writes the tables, every workgroup its envs ----
    if (writes_tables && !failed && r > 0) {
        // (in place: where `r` swaps of the double-buffered table leave the current values; else: the buffers the host hands over)
        double *vf = a.v_out ? a.v_out : (r & 1) ? a.vi.v1 : a.vi.v0;
        double *pf = a.pi_out ? a.pi_out : a.vi.pi;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int32_t s = st[j];
            if (s >= 0) {
                vf[s] = v_new[j];
                if (GREEDY) *reinterpret_cast<double4 *>(pf + 4 * (int64_t)s) = make_double4(p[j][0], p[j][1], p[j][2], p[j][3]);
                if (!AGENTS && a.v_host) {  // (the policy rows too when they did not change: the copy is of BOTH tables)
                    a.v_host[s] = v_new[j];
                    *reinterpret_cast<double4 *>(a.pi_host + 4 * (int64_t)s) = make_double4(p[j][0], p[j][1], p[j][2], p[j][3]);
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; AGENTS && e < 2; ++e) {
        if (own_env[e]) {
            (a.pos_out ? a.pos_out : a.pos)[gid[e]] = e_pos[e];
            (a.reward_out ? a.reward_out : a.reward)[gid[e]] = e_rew[e];
            (a.done_out ? a.done_out : a.done)[gid[e]] = e_done[e];
            (a.episode_out ? a.episode_out : a.episode)[gid[e]] = e_ep[e];
        }
        const uint64_t bits = __ballot(own_env[e] && e_done[e] != 0);
        if (lane == 0 && own_env[e]) (a.done_bits_out ? a.done_bits_out : a.done_bits)[gid[e] >> 6] = bits;
    }
    if (blockIdx.x == 0 && tid == 0) *a.vi.rounds_done = failed ? -1 : r;
    if (blockIdx.x == 0 && tid < 16 && a.hdr_next) a.hdr_next[tid] = 0u;  // the header of the launch behind this one (nobody of THIS launch looks at it)
#ifdef GU_VI_XCD_STAMPS
    if (stamps_here && a.vi.max_rounds >= 20) {  // (the kernel's life beside the loop: delta_key[12 .. 18], ticks since entry)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        life[6] = __builtin_amdgcn_s_memrealtime();
        for (int i = 1; i < 7; ++i) a.vi.delta_key[12 + i] = life[i] - life[0];
    }
#endif
}

// ------------------------------------------------------------------------------------ host side
// The launch shape for this engine, or false when the per-XCD form does not apply: one grid, planes + values (+ action words) within
// one workgroup's LDS, every workgroup resident (at most one per CU), enough workgroups per XCD under the round-robin placement for
// K <= 2 states per thread, four fetch items per thread at most.  `agents`: config 5 (every cluster steps a share of the envs);
// otherwise the tables alone (one cluster does the work, 8 x as many workgroups are launched so that it has enough members).
bool gu_vi_xcd_plan(const gu_engine *h, bool agents, GuXcdPlan *plan)
{
    if (h->n_grids != 1 || h->S > GU_MAX_LDS_CELLS || h->n_cu < VI_XCD_MAX_XCC) return false;
    const int64_t n_words = ((int64_t)h->S + 15) / 16, n_aw = (n_words + 1) / 2, halo = 2 * (int64_t)(h->W < h->S ? h->W : h->S);
    const int64_t forced = gu_opt(h, GU_OPT_VI_XCD_BLOCK);
    const int max_wgs = h->n_cu & ~(VI_XCD_MAX_XCC - 1);
    const int64_t items = halo + (agents ? n_aw : 0);  // halo granules (+ action items): four per thread at most
    // Two passes over the workgroup sizes: first for a shape in which half of a workgroup's waves own no states (they take the delta
    // keys, the collection and the agents off the waves that carry the round: 1.05 against 1.43 us per round at 80x80, 1.53 against
    // 1.91 at 128x128 for the tables alone), then for any shape that fits.
    for (int pass = forced ? 1 : 0; pass < 2; ++pass)
    for (int64_t B = forced ? forced : 256; B <= 1024; B <<= 1) {
        const int64_t env_wgs = agents ? (h->N + B - 1) / B : 0;
        if (env_wgs > h->n_cu) {
            if (forced) return false;
            continue;
        }
        // enough workgroups for one state per TWO threads in every cluster, as far as the device has CUs for them
        const int64_t per_wg = B / 2;
        int64_t G = VI_XCD_MAX_XCC * (((int64_t)h->S + per_wg - 1) / per_wg);
        if (G > max_wgs) G = max_wgs;
        if (G < env_wgs) G = env_wgs;
        const int64_t per_xcc = G / VI_XCD_MAX_XCC > 0 ? G / VI_XCD_MAX_XCC : 1;  // the smallest cluster under round-robin placement
        if ((G + VI_XCD_MAX_XCC - 1) / VI_XCD_MAX_XCC > VI_XCD_SLOTS) return false;
        const int64_t chunk = ((((int64_t)h->S + per_xcc - 1) / per_xcc) + 63) & ~(int64_t)63;
        const int K = items > 4 * B ? 0 : chunk <= B ? 1 : chunk <= 2 * B ? 2 : 0;
        if (K == 0) {
            if (forced) return false;
            continue;
        }
        if (pass == 0 && !(K == 1 && 2 * chunk <= B)) continue;
        // planes | value window (the chunk and one grid row either side) | spare slot of the fetch | action words | the lanes' delta keys
        const int64_t values = ((int64_t)K * B + halo + 2) & ~(int64_t)1;
        const size_t lds = ((2 * (size_t)h->cell_bytes + (size_t)values * sizeof(double) + 16 + (size_t)n_aw * 8 + 15) & ~(size_t)15) + 2 * 1024 * sizeof(vi_u64);
        if ((int64_t)lds + 1024 > h->lds_per_cu) {
            if (forced) return false;
            continue;
        }
        plan->block = (int)B;
        plan->G = (unsigned)G;
        plan->K = K;
        plan->NB = items <= B ? 1 : 4;
        plan->lds = lds;
        plan->values = (uint32_t)values;
        plan->slots_bytes = (size_t)VI_XCD_MAX_XCC * 4 * VI_XCD_SLOTS * 2 * sizeof(vi_u64);
        plan->work_bytes = ((2 * (size_t)h->S * 16 + 2 * (size_t)n_aw * 16) + 255) & ~(size_t)255;
        return true;
    }
    return false;
}

// agents: config 5.  Otherwise the tables alone, with (`greedy`) or without the policy update.
int gu_vi_xcd_launch(gu_engine *h, const GuXcdPlan &plan, const ViStepXcdArgs &args, bool agents, bool greedy)
{
    ViStepXcdArgs a = args;
    a.lds_values = plan.values;
#ifdef GU_VI_XCD_STAMPS
    if (const char *w = getenv("GU_VI_STAMP_WAVE")) a.inject_failure |= ((uint32_t)atoi(w) & 0xFFu) << 8;
    if (const char *k = getenv("GU_VI_STAMP_RANK")) a.inject_failure |= ((uint32_t)atoi(k) & 0xFFu) << 16;
#endif
    typedef void (*Kernel)(const ViStepXcdArgs);
    static std::atomic<uint64_t> lds_mask[12];
    const int which = (plan.K == 1 ? 0 : 1) + (agents ? 0 : greedy ? 2 : 4) + (plan.NB == 1 ? 0 : 6);
    static const Kernel kernels[12] = {gu_vi_xcd_kernel<1, true, true, 1>,   gu_vi_xcd_kernel<2, true, true, 1>,   gu_vi_xcd_kernel<1, false, true, 1>,
                                       gu_vi_xcd_kernel<2, false, true, 1>,  gu_vi_xcd_kernel<1, false, false, 1>, gu_vi_xcd_kernel<2, false, false, 1>,
                                       gu_vi_xcd_kernel<1, true, true, 4>,   gu_vi_xcd_kernel<2, true, true, 4>,   gu_vi_xcd_kernel<1, false, true, 4>,
                                       gu_vi_xcd_kernel<2, false, true, 4>,  gu_vi_xcd_kernel<1, false, false, 4>, gu_vi_xcd_kernel<2, false, false, 4>};
    const Kernel kern = kernels[which];
    gu_allow_lds(kern, lds_mask[which], h->device, plan.lds, (size_t)h->lds_per_cu - 1024);  // (the kernel also has a few static LDS words)
    hipLaunchKernelGGL(kern, dim3(plan.G), dim3(plan.block), plan.lds, h->stream, a);
    GU_HIP(hipGetLastError());
    return GU_OK;
}

// The per-XCD launches' own buffers (see gu_vi_xcd_dp_run): allocated on first use, again when the plan needs more, wiped whenever
// they are (re)allocated and when the launch number is about to repeat.
#define VI_XCD_CTL_KEYS 1024u                       /* bytes of the sixteen launch headers in front of the delta keys */
#define VI_XCD_CTL_SLOTS (VI_XCD_CTL_KEYS + 4096u * 8u) /* the delta-key slots behind 4096 keys */
int gu_vi_xcd_buffers(gu_engine *h, const GuXcdPlan &xp)
{
    const size_t ctl_bytes = VI_XCD_CTL_SLOTS + xp.slots_bytes, work_bytes = 8 * xp.work_bytes;
    bool wipe = (h->vi_xcd_epoch & 0x7FFFFu) == 0u;  // (the tags hold 19 bits of it)
    if (wipe) h->vi_xcd_epoch += 1u;
    if (!h->d_vi_xcd_ctl || h->vi_xcd_ctl_bytes < ctl_bytes) {
        if (h->d_vi_xcd_ctl) (void)hipFree(h->d_vi_xcd_ctl);
        h->d_vi_xcd_ctl = nullptr;
        GU_HIP(hipMalloc(&h->d_vi_xcd_ctl, ctl_bytes));
        h->vi_xcd_ctl_bytes = ctl_bytes;
        wipe = true;
    }
    if (!h->d_vi_xcd_work || h->vi_xcd_work_bytes < work_bytes) {
        if (h->d_vi_xcd_work) (void)hipFree(h->d_vi_xcd_work);
        h->d_vi_xcd_work = nullptr;
        GU_HIP(hipMalloc(&h->d_vi_xcd_work, work_bytes));
        h->vi_xcd_work_bytes = work_bytes;
        wipe = true;
    }
#ifdef GU_VI_XCD_TORN
    wipe = true;  // (16-bit tags: no room for the launch's number -- this variant zeroes everything in front of every launch, like round 4)
#endif
    if (wipe) {
        GU_HIP(hipMemsetAsync(h->d_vi_xcd_ctl, 0, h->vi_xcd_ctl_bytes, h->stream));
        GU_HIP(hipMemsetAsync(h->d_vi_xcd_work, 0, h->vi_xcd_work_bytes, h->stream));
    }
    return GU_OK;
}

uint32_t gu_vi_xcd_tag0(const gu_engine *h)
{
#ifdef GU_VI_XCD_TORN
    (void)h;
    return 0u;
#else
    return (h->vi_xcd_epoch & 0x7FFFFu) << 13;
#endif
}

void gu_vi_xcd_free(gu_engine *h)
{
    if (h->d_vi_xcd_ctl) (void)hipFree(h->d_vi_xcd_ctl);
    if (h->d_vi_xcd_work) (void)hipFree(h->d_vi_xcd_work);
    h->d_vi_xcd_ctl = h->d_vi_xcd_work = nullptr;
    h->vi_xcd_ctl_bytes = h->vi_xcd_work_bytes = 0;
}

// gu_vi_sweep / gu_vi_run / gu_vi_eval_run as ONE launch of one XCD's workgroups (see the kernel, !AGENTS).  Runs up to max_rounds
// rounds on the current tables; GU_VI_FALLBACK: not applicable here or the launch gave up -- the tables are as they were.
static int vi_xcd_dp_launch(gu_engine *h, double gamma, double threshold, bool use_threshold, bool greedy, int32_t max_rounds, int32_t *rounds_done,
                            double *deltas)
{
    *rounds_done = 0;
    if (max_rounds <= 0) return GU_OK;
    const int64_t path = gu_opt(h, GU_OPT_VI_PATH);
    GuXcdPlan xp{};
    if (!(path == 0 || path == 5 || path == 6) || !gu_vi_xcd_plan(h, false, &xp)) return GU_VI_FALLBACK;
    // The engine's OWN buffers (nothing else ever writes them -- the scratch area is shared with every other call):
    //   ctl  = a ring of sixteen 64-byte launch headers | delta keys [4096] | delta-key slots
    //   work = the clusters' granule buffers
    // Every tag in them carries this launch's number above the round (ViStepXcdArgs::tag0), the kernel clears the NEXT launch's
    // header when it is done, and the final tables go to the other halves of the double buffers: ONE launch per call, nothing
    // zeroed, nothing snapshot, nothing to restore.  (Until round 4: a launch of copies and zero fills in front of every call, ~6 us.)
    int rc = gu_vi_xcd_buffers(h, xp);
    if (rc != GU_OK) return rc;
    char *ctl = (char *)h->d_vi_xcd_ctl;
    const uint32_t slot_i = h->vi_xcd_epoch & 15u;
    uint32_t *hdr = (uint32_t *)(ctl + 64 * (size_t)slot_i);
    vi_u64 *keys = (vi_u64 *)(ctl + VI_XCD_CTL_KEYS);
    ViStepXcdArgs a{};
    a.vi = ViClusterArgs{h->d_cell, h->cell_bytes, h->W, h->S, gamma, threshold, h->d_v[h->vi_cur], h->d_v[h->vi_cur ^ 1], h->d_pi[h->vi_cur],
                         keys, hdr, (int32_t *)hdr + 2, max_rounds, use_threshold ? 1 : 0};
    a.N = 0;
    a.slots = (vi_u64 *)(ctl + VI_XCD_CTL_SLOTS);
    a.gx = (uint8_t *)h->d_vi_xcd_work;
    a.work_bytes = (uint32_t)xp.work_bytes;
    a.inject_failure = path == 5;
    a.tag0 = gu_vi_xcd_tag0(h);
    a.hdr_next = (uint32_t *)(ctl + 64 * (size_t)((slot_i + 1u) & 15u));
    a.v_out = h->d_v[h->vi_cur ^ 1];
    a.pi_out = greedy ? h->d_pi[h->vi_cur ^ 1] : nullptr;
    // the final tables also land in a page-locked copy on the host (40 KB at 32 x 32: a few us of stores at the end of the launch),
    // so that the gu_vi_get that follows a gu_vi_run / gu_vi_sweep / gu_vi_eval_run is two memcpys instead of a launch and a wait
    const size_t tables_bytes = 5 * (size_t)h->S * sizeof(double);
    if (!h->h_tables || h->h_tables_bytes < tables_bytes) {
        if (h->h_tables) (void)hipHostFree(h->h_tables);
        h->h_tables = nullptr;
        if (hipHostMalloc((void **)&h->h_tables, tables_bytes, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            h->h_tables = nullptr;
        }
        h->h_tables_bytes = h->h_tables ? tables_bytes : 0;
    }
    h->h_tables_valid = false;
    a.v_host = h->h_tables;
    a.pi_host = h->h_tables ? h->h_tables + h->S : nullptr;
    if ((rc = gu_vi_xcd_launch(h, xp, a, false, greedy)) != GU_OK) return rc;
    ++h->vi_xcd_epoch;
    // (the sixteen headers and the first delta keys lie side by side: ONE copy back and one wait)
    const size_t first_keys = deltas ? (size_t)(max_rounds < 3968 ? max_rounds : 3968) : 0;
    std::vector<unsigned long long> back(128 + first_keys);
    if ((rc = gu_read_back(h, back.data(), ctl, back.size() * sizeof(unsigned long long))) != GU_OK) return rc;
    int32_t ctlw[4];  // [workgroups registered, fallback word, rounds_done, -]
    memcpy(ctlw, back.data() + 8 * (size_t)slot_i, sizeof ctlw);
    h->vi_xcd_torn += (int64_t)(back[8 * (size_t)slot_i + 4] & 0xFFFFFFFFull);  // (hdr[8]: counted by a -DGU_VI_XCD_TORN build only)
    const int32_t done = ctlw[1] ? -1 : ctlw[2];
    if (done < 0) {  // the input tables are as they were; the ring is wiped (who knows which header this launch left how)
        GU_HIP(hipMemsetAsync(ctl, 0, VI_XCD_CTL_KEYS, h->stream));
        if (gu_debug()) fprintf(stderr, "[gu] DP per-XCD kernel gave up (workgroups not resident together, or clusters too uneven); next form\n");
        return GU_VI_FALLBACK;
    }
    if (deltas && done > 0) {
        std::vector<unsigned long long> rest;
        if ((size_t)done > first_keys) {
            rest.resize((size_t)done - first_keys);
            GU_HIP(hipMemcpy(rest.data(), keys + first_keys, rest.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        }
        for (int32_t i = 0; i < done; ++i) {
            const unsigned long long k = (size_t)i < first_keys ? back[128 + (size_t)i] : rest[(size_t)i - first_keys];
            const unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
            memcpy(&deltas[i], &b, sizeof(double));
        }
    }
    if (done > 0) {  // the results are in the other halves of the double buffers: they become the current ones
        std::swap(h->d_v[0], h->d_v[1]);
        if (greedy) std::swap(h->d_pi[0], h->d_pi[1]);
    }
    h->h_tables_valid = h->h_tables != nullptr && done > 0;  // (no round: nothing was written, anywhere)
    *rounds_done = done;
    h->greedy_valid = false;
    return GU_OK;
}

// Config 5 (gu_vi_sweep_step_run) on the per-XCD kernel, ONE launch per up to 4096 rounds and nothing else: the tables' and the
// envs' final state go to the other halves of the engine's double buffers (the env-state arrays got a second set for this), which
// become the current ones only when the launch did not give up -- no snapshot launch in front, nothing to put back behind.
// GU_VI_FALLBACK: the launch gave up (or does not apply), everything is as it was.
int gu_vi_xcd_fused_run(gu_engine *h, const GuXcdPlan &xp, double gamma, int32_t iters, uint32_t flags, double *deltas)
{
    const int64_t path = gu_opt(h, GU_OPT_VI_PATH);
    const size_t n4 = (size_t)h->N * 4, bits_bytes = (((size_t)h->N + 63) / 64) * 8;
    if (!h->d_out3_alt) {  // all three or none: a half-made set would make the launch write two of them in place
        int32_t *out3 = nullptr;
        uint32_t *episode = nullptr;
        uint64_t *bits = nullptr;
        if (hipMalloc((void **)&out3, 3 * n4) != hipSuccess || hipMalloc((void **)&episode, n4) != hipSuccess ||
            hipMalloc((void **)&bits, bits_bytes) != hipSuccess) {
            (void)hipGetLastError();
            if (out3) (void)hipFree(out3);
            if (episode) (void)hipFree(episode);
            if (bits) (void)hipFree(bits);
            return gu_fail(GU_ERR_NOMEM, "gu_vi_sweep_step_run: no device memory for the second set of env-state arrays");
        }
        h->d_out3_alt = out3;
        h->d_episode_alt = episode;
        h->d_done_bits_alt = bits;
    }
    int32_t total = 0;
    while (total < iters) {
        const int32_t n = iters - total < 4096 ? iters - total : 4096;
        int rc = gu_vi_xcd_buffers(h, xp);
        if (rc != GU_OK) return rc;
        char *ctl = (char *)h->d_vi_xcd_ctl;
        const uint32_t slot_i = h->vi_xcd_epoch & 15u;
        uint32_t *hdr = (uint32_t *)(ctl + 64 * (size_t)slot_i);
        vi_u64 *keys = (vi_u64 *)(ctl + VI_XCD_CTL_KEYS);
        ViStepXcdArgs a{};
        a.vi = ViClusterArgs{h->d_cell, h->cell_bytes, h->W, h->S, gamma, 0.0, h->d_v[h->vi_cur], h->d_v[h->vi_cur ^ 1], h->d_pi[h->vi_cur],
                             keys, hdr, (int32_t *)hdr + 2, n, 0};
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
        a.slots = (vi_u64 *)(ctl + VI_XCD_CTL_SLOTS);
        a.gx = (uint8_t *)h->d_vi_xcd_work;
        a.work_bytes = (uint32_t)xp.work_bytes;
        a.inject_failure = path == 5;
        a.tag0 = gu_vi_xcd_tag0(h);
        a.hdr_next = (uint32_t *)(ctl + 64 * (size_t)((slot_i + 1u) & 15u));
        a.v_out = h->d_v[h->vi_cur ^ 1];
        a.pi_out = h->d_pi[h->vi_cur ^ 1];
        a.pos_out = h->d_out3_alt;
        a.reward_out = h->d_out3_alt + h->N;
        a.done_out = h->d_out3_alt + 2 * h->N;
        a.episode_out = h->d_episode_alt;
        a.done_bits_out = h->d_done_bits_alt;
        if ((rc = gu_vi_xcd_launch(h, xp, a, true, true)) != GU_OK) return rc;
        ++h->vi_xcd_epoch;
        const size_t first_keys = deltas ? (size_t)(n < 3968 ? n : 3968) : 0;
        std::vector<unsigned long long> back(128 + first_keys);
        if ((rc = gu_read_back(h, back.data(), ctl, back.size() * sizeof(unsigned long long))) != GU_OK) return rc;
        int32_t ctlw[4];  // [-, fallback word, rounds_done, -]
        memcpy(ctlw, back.data() + 8 * (size_t)slot_i, sizeof ctlw);
        for (int k = 0; k < 8; ++k) h->vi_xcd_members[k] = (int32_t)((back[8 * (size_t)slot_i + 2] >> (7 * k)) & 0x7Full);  // the registration word
        h->vi_xcd_torn += (int64_t)(back[8 * (size_t)slot_i + 4] & 0xFFFFFFFFull);
        if (ctlw[1] || ctlw[2] != n) {
            GU_HIP(hipMemsetAsync(ctl, 0, VI_XCD_CTL_KEYS, h->stream));
            if (gu_debug()) fprintf(stderr, "[gu] sweep-step per-XCD kernel gave up (workgroups not resident together, or clusters too uneven); next form\n");
            if (total > 0) return gu_fail(GU_ERR_HIP, "the per-XCD sweep + step launch gave up %d rounds into a call", total);
            return GU_VI_FALLBACK;
        }
        if (deltas) {
            std::vector<unsigned long long> rest;
            if ((size_t)n > first_keys) {
                rest.resize((size_t)n - first_keys);
                GU_HIP(hipMemcpy(rest.data(), keys + first_keys, rest.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            }
            for (int32_t i = 0; i < n; ++i) {
                const unsigned long long k = (size_t)i < first_keys ? back[128 + (size_t)i] : rest[(size_t)i - first_keys];
                const unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
                memcpy(&deltas[total + i], &b, sizeof(double));
            }
        }
        // the results become the current state
        std::swap(h->d_v[0], h->d_v[1]);
        std::swap(h->d_pi[0], h->d_pi[1]);
        std::swap(h->d_out3, h->d_out3_alt);
        std::swap(h->d_episode, h->d_episode_alt);
        std::swap(h->d_done_bits, h->d_done_bits_alt);
        h->done_bits_valid = true;
        if (h->graph_exec) {  // (a captured step graph carries the old arrays in its arguments)
            (void)hipGraphExecDestroy(h->graph_exec);
            h->graph_exec = nullptr;
        }
        h->greedy_valid = false;
        h->steps_taken += (uint64_t)n;
        total += n;
    }
    return GU_OK;
}

// gu_vi_sweep / gu_vi_run / gu_vi_eval_run on the per-XCD kernel: launches of up to 4096 rounds (the tags keep 13 bits for the round,
// the engine's buffer 4096 delta keys), one after the other while the stopping rule has not fired -- a call of up to 4096 rounds
// is ONE launch.  The tables pass from launch to launch in memory, exactly as between two calls.
int gu_vi_xcd_dp_run(gu_engine *h, double gamma, double threshold, bool use_threshold, bool greedy, int32_t max_rounds, int32_t *rounds_done,
                     double *deltas)
{
    *rounds_done = 0;
    int32_t total = 0;
    while (total < max_rounds) {
        const int32_t n = max_rounds - total < 4096 ? max_rounds - total : 4096;
        int32_t done = 0;
        const int rc = vi_xcd_dp_launch(h, gamma, threshold, use_threshold, greedy, n, &done, deltas ? deltas + total : nullptr);
        if (rc == GU_VI_FALLBACK && total > 0) return gu_fail(GU_ERR_HIP, "the per-XCD DP launch gave up %d rounds into a call", total);
        if (rc != GU_OK) return rc;
        total += done;
        *rounds_done = total;
        if (done < n) break;  // the stopping rule fired
    }
    return GU_OK;
}
