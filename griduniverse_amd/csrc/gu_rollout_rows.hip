// gu_rollout_rows.hip -- the fused rollout on a TRANSITION-ROW table: the latency-bound form of the caller loop of
// core/algorithms/monte_carlo.py:7-26 for one small grid with a single start cell (or no auto-reset).
//
// The general kernel (gu_rollout.hpp) walks the per-cell records: per step  record -> TERM test + two selects (lazy reset)
// -> OPEN bit -> multiply-add -> LDS gather, i.e. ~6 dependent vector ops around one LDS round trip.  Here the whole
// transition of core/envs/griduniverse_env.py:136-155 INCLUDING the harness's `if done: env.reset()` is tabulated per
// (cell, action) once per grid by gu_build_rows_kernel:
//
//     row[s][a] = { LDS byte address of row[next] : bits 0..19 | done(next) : bit 23 | reward_matrix[next] int8 : bits 24..31 }
//     next      = look_step_ahead(base, a).next,   base = start cell if (auto-reset and s is terminal) else s
//
// so that one env-step is   addr = (rec & 0xFFFFF) | (copy * 16 + action * 4);   rec = LDS[addr]   -- ONE v_and_or_b32 and one
// ds_read_b32 on the dependent chain; cell index, reward, done flag, episode count and the trajectory stores all hang off
// `rec` beside the chain.  The 16-byte rows are replicated `copies` (<= 8) times so that the lanes of a half-wave spread
// over the LDS banks (bank = copy * 4 + action).  The invariant the table relies on -- "the env is done exactly when its
// cell is terminal" -- holds after any step but not for an arbitrary stored state (gu_set_state, a reset onto a terminal
// start), so a launch takes its FIRST step on the per-cell planes, like the general kernel -- unless it follows a rollout of
// the same engine (RolloutArgs::entry_table, round 5: the state a rollout leaves behind always agrees with its cell).
//
// Table policies use the same idea with policy-dependent rows, rebuilt by every launch (the policy table may have changed):
//   GU_POLICY_GREEDY  row[s]    = the one record reached by the greedy action of the (post-reset) cell: 4 bytes, up to 32
//                                 copies; a step is v_and_or_b32 + ds_read_b32
//   GU_POLICY_SAMPLE  row[s]    = { inverse-CDF thresholds of pi[cell], SORTED (uint4, from gu_pi_threshold_kernel) | the four next
//                                 records }: 32 bytes, up to 4 copies; a step is ONE round trip for two ds_read_b128, the three
//                                 threshold compares and a three-select pick of the next record (the general kernel: a
//                                 threshold gather, the compares, the move, and a second gather for the record)
//
// Store waves -- a second half of the workgroup that drains an LDS ring of records and issues the rows' stores, so that the stepping
// waves' in-order instruction stream carries neither the stores nor their stalls -- were built in round 4 and measured SLOWER for
// every launch whose chain is short (uniform / greedy at 4096 .. 32 768 envs: 75 against 50 .. 60 us per 1000 steps: the hand-over
// itself, a produced / consumed count pair per wave and eight records per group, costs ~180 clocks per step) and no faster where the
// chain is long (sampled policy at 65 536 envs: 130 against 128; 110 against 117 at 32 768): profiles/archive/r04s_rows_store_waves_ab.json.
// Removed again; the sampled launch with int32 rows stays bound by its one wave per SIMD issuing chain and stores in order.
//
// Results are bit-identical to the general kernel (tests/test_gpu_rows_kernel.py runs both on the same seeds); the launcher
// picks this one where it is faster (profiles/archive/r02b_map_ab.txt, profiles/archive/r02d_rows_crossover.txt).
#include "gu_rollout.hpp"

#define GU_ROW_ADDR_MASK 0xFFFFFu
#define GU_ROW_DONE_BIT 23

struct BuildRowsArgs {
    const uint8_t *cell;  // absorbing-aware planes [flags | reward]
    int32_t cell_bytes, S, W, start0, auto_reset, row_shift;
    uint32_t *rows;       // [S][4]
};

__global__ void __launch_bounds__(256) gu_build_rows_kernel(const BuildRowsArgs a)
{
    const int32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= a.S) return;
    const int32_t base = (a.auto_reset && (a.cell[s] & GU_CELL_TERM)) ? a.start0 : s;  // lazy `if done: env.reset()` (env:187-193)
    const uint32_t fb = a.cell[base];  // OPEN bits of the absorbing map: a terminal cell does not move (env:145-146)
    uint32_t out[4];
#pragma unroll
    for (uint32_t act = 0; act < 4; ++act) {
        const int32_t n = base + (((fb >> act) & 1u) ? gu_delta<false>(act, 0, a.W) : 0);
        const uint32_t d = (a.cell[n] >> GU_CELL_TERM_BIT) & 1u;
        const uint32_t r = (uint8_t)a.cell[a.cell_bytes + n];
        out[act] = ((uint32_t)n << a.row_shift) | (d << GU_ROW_DONE_BIT) | (r << 24);
    }
    *reinterpret_cast<uint4 *>(a.rows + 4 * (int64_t)s) = make_uint4(out[0], out[1], out[2], out[3]);
}

// Table-policy rows.  `thr` == nullptr: greedy (one dword per cell); else sampled (8 dwords per cell).
struct BuildPolicyRowsArgs {
    const uint8_t *cell;
    const uint8_t *greedy;  // first-argmax action per cell
    const uint4 *thr;       // inverse-CDF thresholds per cell
    int32_t cell_bytes, S, W, start0, auto_reset, row_shift;
    uint32_t *rows;
};

__device__ __forceinline__ uint32_t gu_row_record(const uint8_t *cell, int32_t cell_bytes, int32_t base, uint32_t act, int32_t W, int32_t row_shift)
{
    const uint32_t fb = cell[base];
    const int32_t n = base + (((fb >> act) & 1u) ? gu_delta<false>(act, 0, W) : 0);
    const uint32_t d = (cell[n] >> GU_CELL_TERM_BIT) & 1u;
    const uint32_t r = (uint8_t)cell[cell_bytes + n];
    return ((uint32_t)n << row_shift) | (d << GU_ROW_DONE_BIT) | (r << 24);
}

__global__ void __launch_bounds__(256) gu_build_policy_rows_kernel(const BuildPolicyRowsArgs a)
{
    const int32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= a.S) return;
    // the policy is consulted at the POST-reset cell, like the general kernel's table policies
    const int32_t base = (a.auto_reset && (a.cell[s] & GU_CELL_TERM)) ? a.start0 : s;
    if (!a.thr) {
        a.rows[s] = gu_row_record(a.cell, a.cell_bytes, base, a.greedy[base], a.W, a.row_shift);
        return;
    }
    // The step's action is  #{k : word >= thr_k} - never  (gu_sample_action; thresholds no word can reach are stored as 0 and
    // counted in `never`).  A count does not care about the order of its thresholds: with them SORTED, "word >= the c-th" implies
    // "word >= every one before it", and the record of count c can be picked by three chained selects -- 3 compares + 3 selects
    // on the step's dependent chain instead of the count, its carry adds, and a two-level select on the action's bits (21
    // instructions with their wait states).  The records are stored BY COUNT: entry c = the record of action c - never.
    uint4 *out = reinterpret_cast<uint4 *>(a.rows + 8 * (int64_t)s);
    const uint4 q = a.thr[base];
    uint32_t t0 = q.x, t1 = q.y, t2 = q.z;
    if (t0 > t1) { const uint32_t x = t0; t0 = t1; t1 = x; }
    if (t1 > t2) { const uint32_t x = t1; t1 = t2; t2 = x; }
    if (t0 > t1) { const uint32_t x = t0; t0 = t1; t1 = x; }
    out[0] = make_uint4(t0, t1, t2, q.w);
    uint32_t rec[4];
#pragma unroll
    for (uint32_t c = 0; c < 4; ++c) {
        const uint32_t act = c >= q.w ? (c - q.w > 3u ? 3u : c - q.w) : 0u;  // (a count below `never` cannot occur)
        rec[c] = gu_row_record(a.cell, a.cell_bytes, base, act, a.W, a.row_shift);
    }
    out[1] = make_uint4(rec[0], rec[1], rec[2], rec[3]);
}

// PAIR tables (uniform policy / caller-supplied stream, launches that write rows): TWO env-steps per LDS round trip.  The actions
// of these policies do not depend on the env state, so the transitions of two consecutive steps compose ahead of time like the
// K-step tables of gu_rollout_multi.hip -- but here every step leaves a trajectory row behind, so an entry keeps BOTH records:
//
//     pair[s][a1 | a2 << 2] = { record reached by a1 from s,  record reached by a2 from there }          (8 bytes, 128 per cell)
//
// each in the format above with the cell's PAIR row as its address.  One ds_read_b64 then advances an env by two steps, and the
// two records are emitted beside the chain as before.  Steps that do not fill a pair of an action word (the first step of a
// launch, the steps up to the next 16-step word, the last T mod 16) run on a one-step table of the same record format that sits
// behind the pair table in LDS (144 bytes per cell in all: grids of up to ~1100 cells, one workgroup per CU).  What it buys:
// launches bound by the dependent chain, not by the write path -- packed rows at 65 536 envs, int32 rows at a config-4 shard of
// 32 768 envs (profiles/archive/r03r_pair_rows.txt).
__device__ __forceinline__ uint32_t gu_row_record(const uint8_t *cell, int32_t cell_bytes, int32_t base, uint32_t act, int32_t W, int32_t row_shift);

struct BuildPairRowsArgs {
    const uint8_t *cell;
    int32_t cell_bytes, S, W, start0, auto_reset;
    uint32_t *rows2;  // [S][16][2] pair records, then [S][4] one-step records (row_shift 7)
};

#define GU_PAIR_SHIFT 7

__global__ void __launch_bounds__(256) gu_build_pair_rows_kernel(const BuildPairRowsArgs a)
{
    const int32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.S * 16) return;
    const int32_t s = idx >> 4;
    const uint32_t a1 = (uint32_t)idx & 3u, a2 = ((uint32_t)idx >> 2) & 3u;
    const int32_t b1 = (a.auto_reset && (a.cell[s] & GU_CELL_TERM)) ? a.start0 : s;  // lazy `if done: env.reset()` (env:187-193)
    const uint32_t r1 = gu_row_record(a.cell, a.cell_bytes, b1, a1, a.W, GU_PAIR_SHIFT);
    const int32_t n1 = (int32_t)((r1 & GU_ROW_ADDR_MASK) >> GU_PAIR_SHIFT);
    const int32_t b2 = (a.auto_reset && (a.cell[n1] & GU_CELL_TERM)) ? a.start0 : n1;
    const uint32_t r2 = gu_row_record(a.cell, a.cell_bytes, b2, a2, a.W, GU_PAIR_SHIFT);
    reinterpret_cast<uint2 *>(a.rows2)[idx] = make_uint2(r1, r2);
}

typedef __attribute__((address_space(3))) const uint32_t *lds_u32_ptr;
typedef uint32_t gu_v2u __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) const gu_v2u *lds_v2u_ptr;
typedef uint32_t gu_v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) const gu_v4u *lds_v4u_ptr;

// bytes per row (log2): 4 next records / one record / thresholds + 4 next records
template <int POLICY>
struct RowBytes {
    static constexpr int log2 = POLICY == GU_POLICY_GREEDY ? 2 : POLICY == GU_POLICY_SAMPLE ? 5 : 4;
};

// nothing but the chain between the arrival of a record and the issue of the read it addresses (see word16)
#define GU_CHAIN_FENCE() __builtin_amdgcn_sched_barrier(0)
// (workgroups of up to 512 lanes, i.e. 256 registers: the staging below keeps up to twelve 16-byte rows per lane in flight beside the
// 32 registers of the first wave's pacing slots)
#define GU_ROWS_MAX_BLOCK 512
template <int POLICY, int TRAJ, bool STATS, bool PAIR = false>
__global__ void __launch_bounds__(GU_ROWS_MAX_BLOCK) gu_rollout_rows_kernel(const RolloutArgs a, const int32_t auto_reset)
{
    static_assert(!PAIR || POLICY == GU_POLICY_UNIFORM || POLICY == GU_POLICY_STREAM, "pair tables: policies whose actions do not depend on the state");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    GuPacer pacer;
    pacer.fetch(a.pace, TRAJ != 0);  // (asked for ahead of the staging: gu_rollout.hpp)
    const int32_t shift = PAIR ? GU_PAIR_SHIFT : a.row_shift;  // log2(row bytes * copies)
    const int32_t copies_log2 = shift - RowBytes<POLICY>::log2;
    // LDS address of the staged table.  This kernel has no static LDS: its dynamic block starts at LDS address 0 and a record's
    // address bits ARE the ds_read address -- no base is added on the dependent chain.  (Until late in round 5 the block's address
    // was folded into every record; the compiler cannot know that it is 0 -- a link-time constant to it -- and spent one vector
    // instruction per emitted record on `rec - base`, a seventh of config 2's per-step vector work.  The launcher checks it:
    // rows_dispatch refuses an instantiation that reports static LDS.)
    constexpr uint32_t lds_base = 0u;
    // ---- The launch's fixed cost (round 5: 8.6 us of a 60 us config-4 shard, profiles/archive/r05r_rows_intercept.txt) is LATENCY: the table
    // came in through four dependent rounds of global loads (every copy of a row fetched separately, eight loads in flight), and
    // the first step's three dependent global reads (state -> the cell's flags -> the next cell's flags and reward) started behind
    // the staging barrier.  Now: every SOURCE row is loaded once, all of a thread's loads in flight together, and written to its
    // `copies` places from registers; the env's state is asked for right behind the table's first loads; and the first step's reads
    // are issued between the stages of that, so that their round trips pass under it.
    const uint32_t half = (uint32_t)a.half_waves;
    const uint32_t slot_in_block = half ? ((threadIdx.x >> 6) << 5) | (threadIdx.x & 31u) : threadIdx.x;
    const int64_t e64 = (int64_t)gu_env_block(a.xcd_remap) * (blockDim.x >> half) + slot_in_block;
    const bool live = e64 < a.N && !(half && (threadIdx.x & 32u));
    const uint32_t e = live ? (uint32_t)e64 : 0u;  // (a lane without an env reads the state of env 0 and stores nothing)
    const uint32_t prefix = gu_rng_prefix(a.seed_prefix, a.env_id0 + e);
    const int32_t start0 = auto_reset ? a.starts[0] : 0;
    const char *pa = (const char *)a.actions;
    const uint32_t e4 = e * 4u;
    // First step of the launch, on the per-cell planes: the stored done flag decides the lazy reset (it may disagree with the
    // cell: fresh reset onto a terminal start, gu_set_state).  In four parts, each one global round trip behind the one before:
    // the state; the (post-reset) cell's flags and what the policy wants to know there; the flags and the reward of the cell the
    // move ends on; the record.  (The `asm volatile` pins: left to itself the compiler hoists the first USE of a load to right
    // behind its issue -- a wait for the state ahead of the table's loads.)
    int32_t s = 0;
    uint32_t d = 0, ep = 0, t_lane = 0, f0 = 0, first_x = 0, f1 = 0, r1 = 0, first_word = 0;
    uint4 first_thr = make_uint4(0u, 0u, 0u, 0u);
    auto first_0 = [&]() {
        uint32_t ee = e;
        asm volatile("" : "+v"(ee));
        s = a.pos[ee];
        d = (uint32_t)a.done[ee];
        ep = a.episode[ee];
        t_lane = a.tcount[ee];
    };
    // ... unless the state was left by a rollout (a.entry_table: gu_launch_rollout vouches for it).  Then the done flag IS the TERM bit
    // of the cell, the table's rows apply to the first step like to any other, and the launch's fixed cost is ONE global round
    // trip (the state, under the staging) instead of three.
    const bool entry_table = a.entry_table != 0;
    auto first_a = [&]() {
        asm volatile("" : "+v"(t_lane), "+v"(s), "+v"(d));
        t_lane += a.steps_taken;
        if (POLICY == GU_POLICY_SAMPLE) first_word = gu_rng_sample_word(prefix, t_lane);
        if (auto_reset && d) {
            if (!entry_table) s = start0;  // (on the table the row of a terminal cell is the row of the start cell)
            ++ep;
        }
        if (!entry_table) {
            f0 = a.cell[s];
            if (POLICY == GU_POLICY_GREEDY) first_x = a.greedy[s];
            if (POLICY == GU_POLICY_SAMPLE) first_thr = a.pi_thr[s];
        }
        if (POLICY == GU_POLICY_UNIFORM) first_x = (gu_rng_word(prefix, GU_RNG_STREAM_ACTION, t_lane >> 4) >> (2u * (t_lane & 15u))) & 3u;
        if (POLICY == GU_POLICY_STREAM) first_x = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(__builtin_amdgcn_make_buffer_rsrc((void *)pa, 0, 0xFFFFFFFFu, 0x00020000), e4, 0, 0) & 3u;
    };
    auto first_b = [&]() {
        if (entry_table) return;
        asm volatile("" : "+v"(f0));
        const uint32_t act = POLICY == GU_POLICY_SAMPLE ? gu_sample_action(first_word, first_thr) : first_x;
        const int8_t *rew = reinterpret_cast<const int8_t *>(a.cell + a.cell_bytes);
        s += ((f0 >> act) & 1u) ? gu_delta<false>(act, 0, a.W) : 0;
        f1 = a.cell[s];
        r1 = (uint32_t)(uint8_t)rew[s];
    };
    {
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        const uint32_t copies = 1u << copies_log2;
        if (POLICY == GU_POLICY_GREEDY && copies_log2 < 2) {  // (fewer than four copies of a dword row: the old way, dword by dword)
            const int32_t units = (a.S << shift) >> 4;
            first_0();
            first_a();
#pragma unroll 8
            for (int32_t u = threadIdx.x; u < units; u += blockDim.x) {
                const int32_t d0 = u << 2;
                uint4 v = make_uint4(a.rows[d0 >> copies_log2], a.rows[(d0 + 1) >> copies_log2], a.rows[(d0 + 2) >> copies_log2], a.rows[(d0 + 3) >> copies_log2]);
                v.x += lds_base, v.y += lds_base, v.z += lds_base, v.w += lds_base;
                dst[u] = v;
            }
            first_b();
        } else {
            // Source rows: the LDS image itself (PAIR: pair table, then the one-step table; nothing is replicated), or one dword per
            // cell (greedy), one 16-byte unit per cell (uniform / stream), two per cell (sampled: thresholds, next records), each
            // written to `places` places: place k of a row is copy (k + cell) mod copies, so that the lanes of a store, which hold
            // consecutive cells, hit different 16-byte slots of the bank row instead of all the same one.
            // No guards: a lane beyond the end works on the LAST row again, same value to the same place, and the code is chosen once,
            // by the number of source rows per thread, among straight-line variants.  (A per-lane guard is an exec-mask region of ~25
            // clocks around every one of a thread's 30 .. 60 loads and stores; wave-uniform guards make the compiler wait for ALL loads
            // before every guarded one -- for all it knows a guarded load's guarded use was skipped --; an unrolled loop with a
            // `break` came back as a loop over an array in scratch memory.)
            const int32_t B = (int32_t)blockDim.x, log2_b = 31 - __builtin_clz((uint32_t)B);
            const int32_t n_src = PAIR ? a.S * 9 : POLICY == GU_POLICY_SAMPLE ? 2 * a.S : a.S;
            const uint4 *g4 = reinterpret_cast<const uint4 *>(PAIR ? a.rows2 : a.rows);
            const uint32_t places = PAIR ? 1u : POLICY == GU_POLICY_GREEDY ? copies >> 2 : copies;  // 16-byte places per source row
            auto put = [&](uint4 v, int32_t g) {
                if (PAIR || (POLICY != GU_POLICY_GREEDY && (POLICY != GU_POLICY_SAMPLE || (g & 1)))) v.x += lds_base, v.y += lds_base, v.z += lds_base, v.w += lds_base;
                if (PAIR) {
                    dst[g] = v;
                    return;
                }
                const uint32_t cell = POLICY == GU_POLICY_SAMPLE ? (uint32_t)g >> 1 : (uint32_t)g;
                auto at = [&](uint32_t k) {
                    const uint32_t place = (k + cell) & (places - 1u);
                    return POLICY == GU_POLICY_SAMPLE ? ((cell * places + place) << 1) | ((uint32_t)g & 1u) : cell * places + place;
                };
                if (places == 8u) {
#pragma unroll
                    for (uint32_t k = 0; k < 8u; ++k) dst[at(k)] = v;
                } else if (places == 4u) {
#pragma unroll
                    for (uint32_t k = 0; k < 4u; ++k) dst[at(k)] = v;
                } else {
                    for (uint32_t k = 0; k < places; ++k) dst[at(k)] = v;
                }
            };
            // NB rows per thread from `base` on: all loads out, `between()`, then the stores
            auto stage = [&](auto nb, int32_t base, auto between) {
                constexpr int NB = decltype(nb)::value;
                uint4 v[NB];
#pragma unroll
                for (int32_t j = 0; j < NB; ++j) {
                    const uint32_t g = (uint32_t)min(base + j * B + (int32_t)threadIdx.x, n_src - 1);  // (unsigned: base register + 32-bit offset, no address pair per load)
                    if (POLICY == GU_POLICY_GREEDY && !PAIR) {
                        const uint32_t w = a.rows[g] + lds_base;
                        v[j] = make_uint4(w, w, w, w);
                    } else {
                        v[j] = g4[g];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);  // (the state BEHIND the table's loads: an address of theirs otherwise waits for it, see first_0)
                between();
#pragma unroll
                for (int32_t j = 0; j < NB; ++j) put(v[j], min(base + j * B + (int32_t)threadIdx.x, n_src - 1));
            };
            auto first_0a = [&]() {
                first_0();
                first_a();
            };
            auto nothing = []() {};
            const int32_t rows_total = (n_src + B - 1) >> log2_b;  // source rows per thread
            if (rows_total <= 1) {
                stage(std::integral_constant<int, 1>{}, 0, first_0a);
                first_b();
            } else if (rows_total <= 2) {
                stage(std::integral_constant<int, 2>{}, 0, first_0a);
                first_b();
            } else if (rows_total <= 4) {
                stage(std::integral_constant<int, 4>{}, 0, first_0a);
                first_b();
            } else if (TRAJ != 0 && rows_total <= 8) {
                stage(std::integral_constant<int, 8>{}, 0, first_0a);
                first_b();
            } else {
                // (launches that write no rows run several workgroups per CU: they keep to four rows in flight, 16 registers)
                constexpr int WIDE = TRAJ == 0 ? 4 : 12;
                stage(std::integral_constant<int, WIDE>{}, 0, first_0a);
                stage(std::integral_constant<int, WIDE>{}, WIDE * B, first_b);
#pragma unroll 1
                for (int32_t base = 2 * WIDE * B; base < n_src; base += WIDE * B) stage(std::integral_constant<int, WIDE>{}, base, nothing);
            }
        }
        __syncthreads();
    }
    // HALF WAVES (a.half_waves, round 5): lanes 0 .. 31 of every wave carry an env, the others leave -- twice as many waves for the
    // batch.  (The idea: a wave that has its SIMD to itself is bound by its own in-order issue, and the row stores might cost by the
    // lanes they carry.  They do not -- see the launcher for what was measured and where this is used.)
    if (half) pacer.decide_early(a.pace);  // (the launch's first wave sums the launch before while it still has its 64 lanes)
    if (!live) return;
    const uint32_t lane_copy = PAIR ? 0u : (threadIdx.x & ((1u << copies_log2) - 1u)) << RowBytes<POLICY>::log2;  // this lane's copy of every row
    const uint32_t base1 = lds_base + ((uint32_t)a.S << GU_PAIR_SHIFT);  // PAIR: the one-step table behind the pair table

    int32_t ret = 0;
    uint32_t fin = 0;
    uint32_t rec;

    char *po = (char *)a.tr_obs, *pr = (char *)a.tr_reward, *pd = (char *)a.tr_done;
    const int64_t row = a.N * 4;
    // the trajectory's own row pitch and lane offset (TRAJ == 3: one row of (obs, reward, done) triples per step, gu_rollout.hpp)
    const int64_t trow = TRAJ == 3 ? a.N * 12 : row;
    const uint32_t trow32 = (uint32_t)trow;
    const uint32_t te = TRAJ == 3 ? e * 12u : e4;
    __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(po, 0, 0xFFFFFFFFu, 0x00020000);
    __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(pr, 0, 0xFFFFFFFFu, 0x00020000);
    __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(pd, 0, 0xFFFFFFFFu, 0x00020000);
    auto rebase = [&](int64_t rows) {
        po += rows * trow;
        ro = __builtin_amdgcn_make_buffer_rsrc(po, 0, 0xFFFFFFFFu, 0x00020000);
        if (TRAJ == 1) {
            pr += rows * trow;
            pd += rows * trow;
            rr = __builtin_amdgcn_make_buffer_rsrc(pr, 0, 0xFFFFFFFFu, 0x00020000);
            rd = __builtin_amdgcn_make_buffer_rsrc(pd, 0, 0xFFFFFFFFu, 0x00020000);
        }
    };
    // everything a step leaves behind hangs off the record that the step fetched -- none of it is on the chain.  It is
    // emitted one iteration late, i.e. AFTER the next step's LDS read has been issued, so that it fills that read's latency
    // instead of delaying its issue: iteration i fetches record i and emits record i - 1 into trajectory row i - 1.
    auto emit = [&](uint32_t rec, uint32_t soff) {
        const int32_t r = (int32_t)rec >> 24;
        const uint32_t dn = __builtin_amdgcn_ubfe(rec, GU_ROW_DONE_BIT, 1);
        fin += dn;
        if (STATS) ret += r;
        if (TRAJ) {
            const int32_t cell = (int32_t)__builtin_amdgcn_ubfe(rec - lds_base, (uint32_t)shift, 20u - (uint32_t)shift);
            if (TRAJ == 1) {
                __builtin_amdgcn_raw_buffer_store_b32(cell, ro, e4, soff, GU_STORE_AUX);
                __builtin_amdgcn_raw_buffer_store_b32(r, rr, e4, soff, GU_STORE_AUX);
                __builtin_amdgcn_raw_buffer_store_b32((int32_t)dn, rd, e4, soff, GU_STORE_AUX);
            } else if (TRAJ == 3) {
                const gu_v3u triple = {(uint32_t)cell, (uint32_t)r, dn};
                __builtin_amdgcn_raw_buffer_store_b96(triple, ro, te, soff, GU_STORE_AUX);
            GU_WIDE_STORE_PAD();
            } else {
                __builtin_amdgcn_raw_buffer_store_b32((int32_t)((uint32_t)cell | (((uint32_t)r & 0xFFu) << 16) | (dn << 24)), ro, e4, soff, GU_STORE_AUX_PACKED);
            }
        }
    };
    // last part of the launch's first step (its loads were issued under the staging, above): the record it ends on -- or, on the
    // table, the record the state at entry stands for (only its address counts: it is never emitted)
    rec = (((uint32_t)s << shift) + lds_base) | (((f1 >> GU_CELL_TERM_BIT) & 1u) << GU_ROW_DONE_BIT) | (r1 << 24);
    // `x`: the action (uniform / stream), nothing (greedy), the step's RNG word (sample); `between`: work that does not depend
    // on the env state (hashing the next step's RNG word), placed between the issue of the LDS reads and their first use
    auto step_e = [&](uint32_t x, uint32_t soff, auto between, auto emits) {
        const uint32_t prev = rec;
        if (POLICY == GU_POLICY_SAMPLE) {
            const uint32_t addr = (prev & GU_ROW_ADDR_MASK) | lane_copy;
            const gu_v4u q = *(lds_v4u_ptr)(uintptr_t)addr;            // thresholds of the cell ...
            const gu_v4u nx = *(lds_v4u_ptr)(uintptr_t)(addr + 16u);   // ... and its four next records, one round trip
            __builtin_amdgcn_sched_barrier(0);  // the two reads are issued BEFORE the ~17 vector ops of the hash, not behind them
            between();
            __builtin_amdgcn_sched_barrier(0);
            // sorted thresholds, records by count (gu_build_policy_rows_kernel).  The three compares FIRST, into three condition
            // registers, then the three selects: interleaved, every compare wrote VCC and every select waited out the two wait
            // states behind it -- twelve issue slots on the step's dependent chain instead of six.
            const bool c0 = x >= q.x, c1 = x >= q.y, c2 = x >= q.z;
            __builtin_amdgcn_sched_barrier(0);  // (statistics only 68.2 -> 65.0 us, packed rows 76.0 -> 73.7 at config 3; int32 rows, bound by their stores: 114 either way)
            // (A tree two selects deep -- lo = c0 ? y : x, hi = c2 ? w : z, rec = c1 ? hi : lo, which the sorted thresholds allow -- was
            // measured SLOWER than the chain of three: 55.4 against 54.7 us statistics only, 114 .. 115.5 against 113 with rows, round 5.)
            uint32_t sel = c0 ? nx.y : nx.x;
            sel = c1 ? nx.z : sel;
            rec = c2 ? nx.w : sel;
            asm volatile("" ::"v"(q.w));  // keep the unused fourth word of the read live up to here: the compiler otherwise takes its
                                          // register for the hash's temporaries and has to WAIT for the read before the hash can start
        } else if (POLICY == GU_POLICY_GREEDY) {
            rec = *(lds_u32_ptr)(uintptr_t)((prev & GU_ROW_ADDR_MASK) | lane_copy);
            between();
        } else {
            between();
            if (PAIR) {  // (a step outside the pairs: its 16-byte row of the one-step table; records carry PAIR-row addresses)
                rec = *(lds_u32_ptr)(uintptr_t)(base1 + (((prev & GU_ROW_ADDR_MASK) - lds_base) >> 3) + (x << 2));
            } else {
                uint32_t actoff = lane_copy | (x << 2);  // off the chain (the action does not depend on the env state)
                asm("" : "+v"(actoff));                  // keep it ONE value: otherwise the three-way OR is re-associated onto the chain
                const uint32_t addr = (prev & GU_ROW_ADDR_MASK) | actoff;  // v_and_or_b32: the only vector op between two LDS reads
                rec = *(lds_u32_ptr)(uintptr_t)addr;
            }
        }
        if (decltype(emits)::value) emit(prev, soff);
    };
    auto step = [&](uint32_t x, uint32_t soff, auto between) { step_e(x, soff, between, std::true_type{}); };
    auto nothing = [] {};
    // the first step on the table (a.entry_table): like any other, but the record it starts from is nobody's row
    if (entry_table) step_e(POLICY == GU_POLICY_SAMPLE ? first_word : first_x, 0, nothing, std::false_type{});
    pacer.start(a.pace, TRAJ != 0);
    const int32_t T32 = (int32_t)a.T;
    auto step1 = [&](uint32_t x) {
        step(x, 0, nothing);
        if (TRAJ) rebase(1);
    };
    // the 16 steps of one action word.  PAIR: eight round trips of two steps each; the records of a pair are emitted one round
    // trip late, behind the issue of the next read (rows base .. base + 15, the last record stays pending like everywhere here)
    auto word16 = [&](uint32_t word) {
        if (PAIR) {
            uint32_t r1[8], r2[8];
#pragma unroll
            for (uint32_t q = 0; q < 8; ++q) {
                const uint32_t prev = q ? r2[q - 1] : rec;
                uint32_t off = __builtin_amdgcn_ubfe(word, 4 * q, 4) << 3;  // off the chain
                asm("" : "+v"(off));
                // (nothing but the chain between the arrival of `prev` and the issue of the next read: left to itself the scheduler
                // put ~10 instructions of the records' emission there, ~50 clocks on every round trip of a wave that has its SIMD alone)
                GU_CHAIN_FENCE();
                const gu_v2u pr = *(lds_v2u_ptr)(uintptr_t)((prev & GU_ROW_ADDR_MASK) | off);
                GU_CHAIN_FENCE();
                r1[q] = pr.x;
                r2[q] = pr.y;
                if (q == 0) {
                    emit(rec, 0);
                } else {
                    emit(r1[q - 1], (2 * q - 1) * trow32);
                    emit(r2[q - 1], (2 * q) * trow32);
                }
            }
            emit(r1[7], 15 * trow32);
            rec = r2[7];
        } else {
#pragma unroll
            for (uint32_t j = 0; j < 16; ++j) step(__builtin_amdgcn_ubfe(word, 2 * j, 2), j * trow32, nothing);
        }
    };

    if (POLICY == GU_POLICY_UNIFORM) {
        uint32_t t = t_lane;
        uint32_t word = gu_rng_word(prefix, GU_RNG_STREAM_ACTION, t >> 4);  // (the first step is taken: first_a / first_b)
        ++t;
        int32_t i = 1;  // (32-bit counters: gu_rollout caps T at 1e8; the 64-bit ones cost the loop three scalar instructions per group)
        const uint32_t t_first = __builtin_amdgcn_readfirstlane(t);
        if (__all(t == t_first)) {  // every lane at the same step count: the 16-actions-per-word schedule is wave-uniform
            t = t_first;
            if (t & 15u) {  // head: finish the current word
                word = gu_rng_word(prefix, GU_RNG_STREAM_ACTION, t >> 4);
                for (; i < T32 && (t & 15u); ++i, ++t) step1((word >> (2u * (t & 15u))) & 3u);
                pacer.after((uint32_t)i);
            }
            auto groups = [&](auto paced) {  // (two copies: GuPacer::idle)
                for (; i + 16 <= T32; i += 16, t += 16) {
                    word = gu_rng_word(prefix, GU_RNG_STREAM_ACTION, t >> 4);
                    word16(word);
                    if (TRAJ) rebase(16);
                    if (decltype(paced)::value && i + 16 < T32) pacer.after(16);  // (gu_rollout.hpp: GuPacer)
                }
            };
            if (TRAJ == 0 || pacer.idle()) groups(std::false_type{});
            else groups(std::true_type{});
            if (i < T32) {
                word = gu_rng_word(prefix, GU_RNG_STREAM_ACTION, t >> 4);
                for (uint32_t j = 0; i < T32; ++i, ++j) step1((word >> (2u * j)) & 3u);
            }
        } else {
            if ((t & 15u) == 0u) word = gu_rng_word(prefix, GU_RNG_STREAM_ACTION, t >> 4);
            for (; i < T32; ++i) {
                step1((word >> (2u * (t & 15u))) & 3u);
                ++t;
                if ((t & 15u) == 0u) word = gu_rng_word(prefix, GU_RNG_STREAM_ACTION, t >> 4);
            }
        }
    } else if (POLICY == GU_POLICY_GREEDY || POLICY == GU_POLICY_SAMPLE) {
        // first step on the planes; the policy is consulted at the post-reset cell
        uint32_t t = t_lane;
        uint32_t word = first_word;  // (the first step is taken: first_a / first_b)
        if (POLICY == GU_POLICY_SAMPLE) word = gu_rng_sample_advance(prefix, t, word);
        ++t;
        int32_t i = 1;  // (32-bit counters: gu_rollout caps T at 1e8; the 64-bit ones cost the loop three scalar instructions per group)
        auto pstep = [&](uint32_t soff) {  // the word of the NEXT step is hashed while this step's reads are in flight
            uint32_t next_word = 0u;
            step(word, soff, [&] {
                if (POLICY == GU_POLICY_SAMPLE) next_word = gu_rng_sample_advance(prefix, t, word);
            });
            word = next_word;
            ++t;
        };
        // Sampled policy, every lane at the same step count: a head of single steps up to a multiple of four, then groups of eight
        // in which the steps that start a word (the fourth and the eighth) are known at compile time -- no ballot, no branch.
        const uint32_t t_first = __builtin_amdgcn_readfirstlane(t);
        if (POLICY == GU_POLICY_SAMPLE && __all(t == t_first)) {
            t = t_first;
            for (; i < T32 && (t & GU_RNG_SAMPLE_MASK); ++i) {
                pstep(0);
                if (TRAJ) rebase(1);
            }
            if (i > 1) pacer.after((uint32_t)i);
            constexpr uint32_t G = GU_RNG_SAMPLE_MASK + 1u < 8u ? 8u : GU_RNG_SAMPLE_MASK + 1u;  // steps per unrolled group
            auto groups = [&](auto paced) {  // (two copies: GuPacer::idle)
                for (; i + G <= T32; i += G) {
#pragma unroll
                    for (uint32_t j = 0; j < G; ++j) {
                        uint32_t next_word = 0u;
                        step(word, j * trow32, [&] {
                            next_word = (j & GU_RNG_SAMPLE_MASK) == GU_RNG_SAMPLE_MASK ? gu_rng_sample_advance_at<true>(prefix, t, word) : gu_rng_sample_advance_at<false>(prefix, t, word);
                        });
                        word = next_word;
                        ++t;
                    }
                    if (TRAJ) rebase(G);
                    if (decltype(paced)::value && i + G < T32) pacer.after(G);
                }
            };
            if (TRAJ == 0 || pacer.idle()) groups(std::false_type{});
            else groups(std::true_type{});
        } else {
            auto groups = [&](auto paced) {  // (two copies: GuPacer::idle)
                for (; i + 8 <= T32; i += 8) {
#pragma unroll
                    for (uint32_t j = 0; j < 8; ++j) pstep(j * trow32);
                    if (TRAJ) rebase(8);
                    if (decltype(paced)::value && i + 8 < T32) pacer.after(8);
                }
            };
            if (TRAJ == 0 || pacer.idle()) groups(std::false_type{});
            else groups(std::true_type{});
        }
        for (; i < T32; ++i) {
            pstep(0);
            if (TRAJ) rebase(1);
        }
    } else {  // GU_POLICY_STREAM: packed words of 16 two-bit actions (gu_pack_actions_kernel), see gu_stream_run
        gu_stream_run(
            pa, row, e4, a.T, 1,
            [&](uint32_t word) {
                word16(word);
                if (TRAJ) rebase(16);
                pacer.after(16);
            },
            step1);
    }
    emit(rec, 0);  // the last step's record (row T - 1: every loop above leaves the row base one step behind)
    pacer.finish();
    // resets performed = steps that started from a done env = (done at entry: counted in first_step) + done flags seen
    // on every step but the last
    const uint32_t d_last = __builtin_amdgcn_ubfe(rec, GU_ROW_DONE_BIT, 1);
    if (auto_reset) ep += fin - d_last;
    a.pos[e] = (int32_t)__builtin_amdgcn_ubfe(rec - lds_base, (uint32_t)shift, 20u - (uint32_t)shift);
    a.reward[e] = (int32_t)rec >> 24;
    a.done[e] = (int32_t)d_last;
    a.episode[e] = ep;
    if (STATS) {
        a.ret[e] = ret;
        a.episodes_fin[e] = (int32_t)fin;
    }
    const uint64_t bits = __ballot(d_last != 0);
    if ((threadIdx.x & 63) == 0) {
        if (half) reinterpret_cast<uint32_t *>(a.done_bits)[e >> 5] = (uint32_t)bits;  // (this wave's 32 envs: one half of a ballot word)
        else a.done_bits[e >> 6] = bits;
    }
}

// ------------------------------------------------------------------------------------ host side
// (block size, copies) for the row table, or false when it does not fit: row_bytes * copies bytes per cell and block, one table
// shared by all waves of a block; the batch must fit in (blocks per CU the LDS admits) x 256 CUs blocks.
static bool rows_shape(const gu_engine *h, int row_bytes, int max_copies, int *block, int *copies)
{
    if (h->n_grids != 1) return false;
    // the smallest workgroup that fits, with as many copies as its LDS share admits (the copy count matters little once the
    // table is staged with wide, pipelined stores; the workgroup size does: profiles/archive/r02e_rows_copies.txt)
    // (128- and 64-thread workgroups, which spread a 32 768-env launch over all CUs instead of half of them, are no faster: 70 .. 72 us
    // either way, profiles/archive/r03h_rows_block.txt)
    for (int bs = 256; bs <= GU_ROWS_MAX_BLOCK; bs <<= 1) {
        const int64_t blocks = (h->N + bs - 1) / bs, per_cu = (blocks + h->n_cu - 1) / h->n_cu;
        for (int c = max_copies; c >= 1; c >>= 1) {
            if ((int64_t)h->S * row_bytes * c * per_cu <= h->lds_per_cu - 2048) {
                *block = bs;
                *copies = c;
                return true;
            }
        }
    }
    return false;
}

// copies of every row across the LDS banks: up to 8 (uniform / stream), 16 (greedy), 4 (sampled); GU_ROWS_COPIES overrides
// (a power of two; diagnostics)
static int rows_max_copies(const gu_engine *h, int32_t policy)
{
    const int v = (int)gu_opt(h, GU_OPT_ROWS_COPIES);
    if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16 || v == 32) return v;
    return policy == GU_POLICY_GREEDY ? 16 : policy == GU_POLICY_SAMPLE ? 4 : 8;
}

static int rows_mode(const gu_engine *h) { return (int)gu_opt(h, GU_OPT_ROLLOUT_ROWS); }

template <int POLICY, bool PAIR = false>
static bool rows_dispatch(const gu_engine *h, const RolloutArgs &a, int traj, bool stats, int auto_reset, dim3 grid, dim3 block, size_t lds, hipStream_t stream)
{
#define GU_ROWS_LAUNCH(TR, ST)                                                                                           \
    do {                                                                                                                 \
        auto kern = gu_rollout_rows_kernel<POLICY, TR, ST, PAIR>;                                                        \
        static std::atomic<uint64_t> raised{0}; /* per instantiation and per device: raise the dynamic-LDS limit once */ \
        gu_allow_lds(kern, raised, h->device, lds, (size_t)h->lds_per_cu);                                               \
        static std::atomic<int> base_zero{0}; /* 1: no static LDS, the dynamic block starts at 0 (the kernel relies on it) */ \
        if (!base_zero.load(std::memory_order_relaxed)) {                                                                \
            hipFuncAttributes fa{};                                                                                      \
            const bool got = hipFuncGetAttributes(&fa, (const void *)kern) == hipSuccess;                                \
            base_zero.store(got && fa.sharedSizeBytes == 0 ? 1 : 2, std::memory_order_relaxed);                          \
        }                                                                                                                \
        if (base_zero.load(std::memory_order_relaxed) != 1) return false;                                                \
        hipLaunchKernelGGL(kern, grid, block, lds, stream, a, auto_reset);                                               \
    } while (0)
    if (traj == 1) {
        if (stats) GU_ROWS_LAUNCH(1, true); else GU_ROWS_LAUNCH(1, false);
    } else if (traj == 2) {
        if (stats) GU_ROWS_LAUNCH(2, true); else GU_ROWS_LAUNCH(2, false);
    } else if (traj == 3) {
        if (stats) GU_ROWS_LAUNCH(3, true); else GU_ROWS_LAUNCH(3, false);
    } else {
        if (stats) GU_ROWS_LAUNCH(0, true); else GU_ROWS_LAUNCH(0, false);
    }
    return true;
#undef GU_ROWS_LAUNCH
}

// the pair tables fit this engine's grid: one grid, one start cell (or no auto-reset), 144 bytes of LDS per cell in one workgroup's share
bool gu_rows_pairs_fit(const gu_engine *h)
{
    return h->n_grids == 1 && (int64_t)h->S * 144 <= h->lds_per_cu - 2048 && ((int64_t)h->S << GU_PAIR_SHIFT) <= (int64_t)GU_ROW_ADDR_MASK && rows_mode(h) != 0 &&
           rows_mode(h) != 2;
}

// Returns true when the launch was taken by the row-table kernel.
bool gu_rollout_rows(gu_engine *h, RolloutArgs a, int32_t policy, int auto_mode, int traj, bool stats, int *rc)
{
    *rc = GU_OK;
    if (policy < GU_POLICY_UNIFORM || policy > GU_POLICY_SAMPLE) return false;
    if (auto_mode == 2) return false;  // several start cells: the reset draws from the RNG, it cannot be tabulated
    const int mode = rows_mode(h);
    // Default policy (profiles/archive/r02b_map_ab.txt, profiles/archive/r02d_rows_crossover.txt, profiles/archive/r02e_policy_rows.txt; interleaved
    // A/B in one process): every launch that is bound by the dependent chain rather than by the HBM write path --
    //   stats only           : every batch size (uniform: 62 -> 40 us at 65 536 envs, 68 -> 35 us at 262 144)
    //   packed rows (4 B)    : up to one 256-env workgroup per CU (83 -> 51 us at 65 536 envs; 103 against 111 us at 131 072)
    //   int32 rows (12 B)    : uniform / stream / greedy up to 32 768 envs (80 -> 59 us at 4096..16 384 envs, 84 -> 74 us at
    //                          32 768 -- config 2, and a config-4 shard; beyond that the general kernel's store timing is the
    //                          better one: 124 against 133 us at 65 536 envs); sampled up to one workgroup per CU
    //   sampled policy       : only with auto-reset (122 against 141 us stats only, 148 against 168 us int32 rows at 65 536
    //                          envs); without it the general kernel's shorter step wins (107 against 122 us) -- a sampled step
    //                          is bound by its ~45 vector instructions, half of them the MurmurHash3 of its RNG word, not by
    //                          the LDS round trips the row table saves
    if (mode == 0) return false;
    if (mode != 1 && mode != 2 && mode != 3) {
        const unsigned blocks = gu_blocks(h->N, 256);
        // (a caller-supplied stream with int32 rows: the row-table kernel reads its action words straight from HBM, and a load
        // among streaming stores waits for all of them -- beyond 16 384 envs the general kernel, which stages the words in LDS,
        // is the quicker one: 88 against 116 us at 32 768 envs, profiles/archive/r02j_stream_crossover.txt)
        // (measured on the 256 CUs of an MI355X; stated relative to the CU count: one workgroup per CU, a quarter, half of them)
        const unsigned cus = (unsigned)h->n_cu;
        // (round 3, under the schedule limiter, 65 536 envs with int32 rows, profiles/archive/r03s_rows_vs_general.txt: greedy with auto-reset
        // 108 .. 111 us here against 119 .. 120 on the general kernel, whose step then has two dependent LDS reads; sampled without
        // auto-reset 135 against 142; uniform / stream / greedy without auto-reset: the same on both, they stay where they were)
        // (the whole table, 8192 .. 65 536 envs x four policy kinds x int32 / packed rows on both kernels: profiles/archive/r03s_rows_crossover.txt.
        // A caller-supplied stream with int32 rows used to leave this kernel at 16 384 envs -- its action words are read straight
        // from HBM among the streaming stores --; with sc1 + nt stores it is the quicker one up to 32 768 like the uniform policy:
        // 60 .. 61 against 66 .. 71 us)
        const unsigned int32_limit = (policy == GU_POLICY_SAMPLE || (policy == GU_POLICY_GREEDY && auto_mode == 1)) ? cus : cus / 2;
        if (((traj == 1 || traj == 3) && blocks > int32_limit) || (traj == 2 && blocks > cus)) return false;
        if (policy == GU_POLICY_SAMPLE && auto_mode != 1 && traj == 0) return false;
    }
    const bool table_policy = policy == GU_POLICY_GREEDY || policy == GU_POLICY_SAMPLE;
    const int row_log2 = policy == GU_POLICY_GREEDY ? 2 : policy == GU_POLICY_SAMPLE ? 5 : 4;
    int bs = 0, copies = 0;
    if (!rows_shape(h, 1 << row_log2, rows_max_copies(h, policy), &bs, &copies)) return false;
    int shift = row_log2;
    while ((1 << (shift - row_log2)) < copies) ++shift;
    const int which = auto_mode ? 1 : 0;
    if (table_policy) {
        // policy-dependent rows: rebuilt by every launch (one tiny kernel), like the threshold table itself
        if (!h->d_prow) {
            if (hipMalloc(&h->d_prow, (size_t)h->S * 32) != hipSuccess) {
                (void)hipGetLastError();
                return false;
            }
        }
        BuildPolicyRowsArgs b{h->d_cell, h->d_greedy, policy == GU_POLICY_SAMPLE ? h->d_pi_thr : nullptr, h->cell_bytes, h->S, h->W,
                              h->start0, which, shift, h->d_prow};
        hipLaunchKernelGGL(gu_build_policy_rows_kernel, dim3((unsigned)((h->S + 255) / 256)), dim3(256), 0, h->stream, b);
        a.rows = h->d_prow;
    } else {
        if (!h->d_rows[which]) {
            if (hipMalloc(&h->d_rows[which], (size_t)h->S * 4 * sizeof(uint32_t)) != hipSuccess) {
                (void)hipGetLastError();
                return false;
            }
            h->rows_shift[which] = -1;
        }
        if (h->rows_shift[which] != shift) {
            BuildRowsArgs b{h->d_cell, h->cell_bytes, h->S, h->W, h->start0, which, shift, h->d_rows[which]};
            hipLaunchKernelGGL(gu_build_rows_kernel, dim3((unsigned)((h->S + 255) / 256)), dim3(256), 0, h->stream, b);
            h->rows_shift[which] = shift;
        }
        a.rows = h->d_rows[which];
    }
    // Pair tables (two steps per LDS round trip) where they pay: state-independent actions, PACKED rows (int32 rows are as fast or
    // faster on the one-step table -- 58 .. 61 against 65 us at 32 768 envs, 110 against 115 at 65 536: six stores and two records'
    // worth of unpacking per round trip cost what the shorter chain saves, and those launches are close to the write path's rate
    // anyway -- round 4, forced for int32 rows again, the chain fenced (GU_CHAIN_FENCE): 43.8 against 44.6 ns per step at config 2's
    // 4096 envs, slower at 32 768 and 65 536.  What bounds a wave that has its SIMD alone is the ISSUE of its stores, ~25 clocks per
    // 256-byte buffer_store_dword: 107 clocks per step with three of them, 143 per packed pair with two, 210 per int32 pair with six
    // (slopes over T = 2000 .. 4000); without rows the K-step kernel of gu_rollout_multi.hip is the tool), one 256-lane workgroup per CU at most (144 bytes
    // of LDS per cell).  GU_OPT_ROLLOUT_ROWS = 2 keeps the one-step table (A/B, tests).  profiles/archive/r03r_pair_rows.txt
    // (int32 TRIPLES, round 5: two 12-byte stores per pair instead of six 4-byte ones -- 37 against 50 us at config 2's 4096 envs, 39 at
    // 8192, level with the one-step table at 16 384, slower beyond; the launcher hands triples to this kernel only where they pay,
    // GU_OPT_ROLLOUT_ROWS = 3 forces the pairs for every triples launch)
    bool pair = !table_policy && (traj == 2 || (traj == 3 && (mode == 3 || policy == GU_POLICY_UNIFORM) && (mode == 3 || (int64_t)gu_blocks(h->N, 256) * 4 <= h->n_cu))) &&
                mode != 2 && gu_blocks(h->N, 256) <= (unsigned)h->n_cu && gu_rows_pairs_fit(h);
    if (pair) {
        if (!h->d_rows2[which]) {
            if (hipMalloc(&h->d_rows2[which], (size_t)h->S * 144) != hipSuccess) {
                (void)hipGetLastError();
                pair = false;
            }
            h->rows2_built[which] = false;
        }
        if (pair && !h->rows2_built[which]) {
            BuildPairRowsArgs b2{h->d_cell, h->cell_bytes, h->S, h->W, h->start0, which, h->d_rows2[which]};
            hipLaunchKernelGGL(gu_build_pair_rows_kernel, dim3((unsigned)((h->S * 16 + 255) / 256)), dim3(256), 0, h->stream, b2);
            BuildRowsArgs b1{h->d_cell, h->cell_bytes, h->S, h->W, h->start0, which, GU_PAIR_SHIFT, h->d_rows2[which] + (size_t)h->S * 32};
            hipLaunchKernelGGL(gu_build_rows_kernel, dim3((unsigned)((h->S + 255) / 256)), dim3(256), 0, h->stream, b1);
            h->rows2_built[which] = true;
        }
        if (pair) {
            a.rows2 = h->d_rows2[which];
            bs = 256;
            shift = GU_PAIR_SHIFT;
        }
    }
    a.row_shift = shift;
    const size_t lds = pair ? (size_t)h->S * 144 : ((size_t)h->S << row_log2) << (shift - row_log2);
    // Half waves (see the kernel).  Measured (profiles/archive/r05m_half_sizes.txt, r05m_half_ab.txt): the stores of a wave do NOT get cheaper
    // with fewer lanes -- planes at 4096 .. 8192 envs: 52.3 us either way, 55.4 against 53.3 at 16 384, and 223 against 126 us
    // where two half waves share a SIMD -- so this is no cure for the issue-bound launches.  It pays in ONE place: triples with
    // the pair tables between 8192 and 16 384 envs (43.6 against 47.6 us; the planes: 53.3), where a workgroup per four CUs
    // becomes one per two.  That is the default; GU_OPT_ROLLOUT_HALF_WAVES = 1 / 0 forces / forbids it.
    const int64_t half_opt = gu_opt(h, GU_OPT_ROLLOUT_HALF_WAVES);
    const bool half = traj != 0 && h->N % 32 == 0 &&
                      (half_opt == 1 || (half_opt == -1 && traj == 3 && pair && (int64_t)gu_blocks(h->N, 256) * 8 > h->n_cu && (int64_t)gu_blocks(h->N, 256) * 4 <= h->n_cu));
    a.half_waves = half ? 1 : 0;
    const dim3 grid(gu_blocks(h->N, half ? bs / 2 : bs)), block(bs);
    a.xcd_remap = a.xcd_remap && grid.x % 8 == 0;
    bool launched = true;  // (false: an instantiation with static LDS below its table -- not this build; the launch fails loudly)
    auto launch = [&](const RolloutArgs &args) {
        switch (policy) {
        case GU_POLICY_UNIFORM:
            launched = pair ? rows_dispatch<GU_POLICY_UNIFORM, true>(h, args, traj, stats, which, grid, block, lds, h->stream)
                            : rows_dispatch<GU_POLICY_UNIFORM>(h, args, traj, stats, which, grid, block, lds, h->stream);
            break;
        case GU_POLICY_STREAM:
            launched = pair ? rows_dispatch<GU_POLICY_STREAM, true>(h, args, traj, stats, which, grid, block, lds, h->stream)
                            : rows_dispatch<GU_POLICY_STREAM>(h, args, traj, stats, which, grid, block, lds, h->stream);
            break;
        case GU_POLICY_GREEDY: launched = rows_dispatch<GU_POLICY_GREEDY>(h, args, traj, stats, which, grid, block, lds, h->stream); break;
        default: launched = rows_dispatch<GU_POLICY_SAMPLE>(h, args, traj, stats, which, grid, block, lds, h->stream); break;
        }
    };
    if (traj) {  // rows to write: the store stream is rate-limited here too (gu_rollout.hpp: GuPacer)
        *rc = gu_pace_for(h, (traj != 2 ? 12 : 24) + policy * 3 + auto_mode, a.T, grid.x, bs, traj != 2 ? 12 : 4, &a.pace);
        if (*rc != GU_OK) return true;
    }
    launch(a);
    if (!launched) *rc = gu_fail(GU_ERR_HIP, "gu_rollout_rows_kernel reports static LDS: its table must start at LDS address 0");
    return true;
}
