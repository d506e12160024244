// gu_kernels.hip -- step / reset / rollout kernels for gfx950 (CDNA4, wave64).
//
// One wavefront lane per env instance.  Every kernel first stages the grid's two per-cell
// byte planes (flags, reward; see gu_internal.hpp) from L2 into LDS with 16-byte loads.  The
// transition of core/envs/griduniverse_env.py:136-155 is then, per env-step,
//
//     open   = (flags >> a) & 1          0 = grid edge / wall at the candidate / absorbing terminal
//     s      = s + open * delta[a]       delta = {-W, +1, +W, -1} (env:51-54), one v_mad_i32_i24
//     flags  = F[s];  reward = R[s]      two LDS byte reads (ds_read_u8 / ds_read_i8)
//     done   = (flags >> 4) & 1          (env:163-168)
//
// i.e. a dependent chain of 2 VALU ops + 1 LDS read per step.  State traffic is coalesced int32
// SoA: lane e touches word e of pos[] / reward[] / done[] / actions[] and of each trajectory row
// (scalar row base + lane offset, so the row advance costs only SALU).  This is HBM-write-bound
// integer work: no MFMA, and no inter-block reuse apart from the <=64 KiB record planes that
// every XCD's L2 holds after first touch -- so there is nothing for an XCD-aware block remap to
// win here; the grid is N/256 four-wave workgroups (at N = 65 536 that is one workgroup per CU,
// one wave per SIMD; measured 3 % faster than 1024 one-wave workgroups, profiles/r01b_bench_blocksize.txt).
//
// The action -> delta LUT is four int16 lanes of one 64-bit scalar register (v_lshrrev_b64 +
// v_bfe_i32): staging a 4-entry table in LDS instead would put a second, dependent ds_read on
// every step for no gain.
#include "gu_internal.hpp"
#include "gu_rng.hpp"

#include <cstdlib>

#define GU_BLOCK 256

// ------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------
struct CellMap {
    const uint8_t *f;  // flags plane
    const int8_t *r;   // reward plane
};

// LDS variants: the whole block uses ONE grid (single-grid engines, or multi-grid engines whose group size is a
// multiple of the block size -- the launcher guarantees it), whose planes are staged into LDS.
__device__ __forceinline__ uint32_t gu_block_grid(const GridSel &gs)
{
    return gs.n_grids > 1 ? (uint32_t)(((int64_t)blockIdx.x * blockDim.x) / gs.group) : 0u;
}

template <bool LDS>
__device__ __forceinline__ CellMap gu_stage_map(const uint8_t *__restrict__ g, int32_t cell_bytes, uint8_t *smem, const GridSel &gs)
{
    if (LDS) {
        g += (int64_t)gu_block_grid(gs) * gs.grid_stride;
        for (int32_t i = threadIdx.x * 16; i < 2 * cell_bytes; i += blockDim.x * 16)
            *reinterpret_cast<uint4 *>(smem + i) = *reinterpret_cast<const uint4 *>(g + i);
        __syncthreads();
        return CellMap{smem, reinterpret_cast<const int8_t *>(smem + cell_bytes)};
    }
    return CellMap{g, reinterpret_cast<const int8_t *>(g + cell_bytes)};
}

// Which grid does lane e use?  In the LDS variants it is the block's grid (start table selected with scalar
// arithmetic); the L2 variants serve any group size: env e uses grid e / group, its planes sit g * grid_stride
// bytes into the plane buffer.
struct LaneGrid {
    const int32_t *starts;
    uint32_t n_starts;
};

template <bool LDS>
__device__ __forceinline__ LaneGrid gu_lane_grid(const GridSel &gs, const int32_t *starts, uint32_t n_starts0, uint32_t e, CellMap &m)
{
    if (gs.n_grids <= 1) return LaneGrid{starts, n_starts0};
    if (LDS) {
        const uint32_t gb = gu_block_grid(gs);
        return LaneGrid{starts + (int64_t)gb * gs.max_starts, (uint32_t)gs.n_starts[gb]};
    }
    const uint32_t g = e / (uint32_t)gs.group;
    m.f += (int64_t)g * gs.grid_stride;
    m.r += (int64_t)g * gs.grid_stride;
    return LaneGrid{starts + (int64_t)g * gs.max_starts, (uint32_t)gs.n_starts[g]};
}

// delta[a]: LUT = four int16 lanes {-W, +1, +W, -1}; ARITH = any W (grids too big for the LUT / LDS)
template <bool LUT>
__device__ __forceinline__ int32_t gu_delta(uint32_t a, uint64_t lut, int32_t W)
{
    if (LUT) return __builtin_amdgcn_sbfe((int32_t)(uint32_t)(lut >> (a << 4)), 0, 16);
    const int32_t sign = (int32_t)(a & 2u) - 1;  // UP,RIGHT -> -1 ; DOWN,LEFT -> +1
    return (a & 1u) ? -sign : sign * W;
}

__device__ __forceinline__ int32_t gu_reward_packed(uint32_t flags)
{
    return (flags & GU_CELL_RMINUS) ? -10 : ((flags & GU_CELL_RPLUS) ? 10 : -1);
}

__device__ __forceinline__ int32_t gu_move(int32_t s, uint32_t flags, uint32_t a, int32_t delta)
{
    return __mul24((int32_t)__builtin_amdgcn_ubfe(flags, a, 1), delta) + s;  // v_bfe_u32 + v_mad_i32_i24
}

// ------------------------------------------------------------------------------------
// reset: GridUniverseEnv._reset (env:187-193) for the masked / done envs
// ------------------------------------------------------------------------------------
struct ResetArgs {
    int32_t *pos, *done;
    uint32_t *episode;
    const int32_t *starts;
    const uint8_t *mask;
    const int32_t *choice;
    uint32_t n_starts, seed_prefix, env_id0;
    int64_t N;
    int32_t only_done;
    GridSel gs;
};

__global__ void __launch_bounds__(GU_BLOCK) gu_reset_kernel(const ResetArgs a)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.N) return;
    if (a.mask && !a.mask[e]) return;
    if (a.only_done && !a.done[e]) return;
    uint32_t ep = a.episode[e];
    const int32_t *starts = a.starts;
    uint32_t n_starts = a.n_starts;
    if (a.gs.n_grids > 1) {
        const uint32_t g = (uint32_t)e / (uint32_t)a.gs.group;
        starts += (int64_t)g * a.gs.max_starts;
        n_starts = (uint32_t)a.gs.n_starts[g];
    }
    uint32_t idx;
    if (a.choice) {
        idx = (uint32_t)a.choice[e];
        if (idx >= n_starts) idx = 0;  // host validates; never index out of the table
    } else {
        idx = gu_rng_start_index(gu_rng_prefix(a.seed_prefix, a.env_id0 + (uint32_t)e), ep, n_starts);
    }
    a.pos[e] = starts[idx];
    a.done[e] = 0;
    a.episode[e] = ep + 1;
}

// ------------------------------------------------------------------------------------
// single step: GridUniverseEnv._step (env:176-185), actions from a device row
//   algorithmic HBM bytes per env-step: action 4 + pos 4 in, pos 4 + reward 4 + done 4 out = 20 B
//   (+4 B done read with GU_F_AUTO_RESET)
// ------------------------------------------------------------------------------------
struct StepArgs {
    const uint8_t *cell;
    int32_t cell_bytes, W;
    uint64_t lut;
    const int32_t *actions;
    int32_t *pos, *reward, *done;
    uint32_t *episode;
    const int32_t *starts;
    uint32_t n_starts, seed_prefix, env_id0;
    int64_t N;
    uint32_t flags;
    GridSel gs;
    int32_t *host_obs, *host_reward, *host_done;  // optional page-locked host mirrors written by the kernel itself
};

template <bool LDS>
__global__ void __launch_bounds__(GU_BLOCK) gu_step_kernel(const StepArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    CellMap m = gu_stage_map<LDS>(a.cell, a.cell_bytes, smem, a.gs);
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.N) return;
    const LaneGrid lg = gu_lane_grid<LDS>(a.gs, a.starts, a.n_starts, (uint32_t)e, m);
    const uint32_t act = (uint32_t)a.actions[e] & 3u;
    int32_t s = a.pos[e];
    if ((a.flags & GU_F_AUTO_RESET) && a.done[e]) {  // lazy `if done: env.reset()`
        const uint32_t ep = a.episode[e];
        s = lg.starts[gu_rng_start_index(gu_rng_prefix(a.seed_prefix, a.env_id0 + (uint32_t)e), ep, lg.n_starts)];
        a.episode[e] = ep + 1;
    }
    s = gu_move(s, m.f[s], act, gu_delta<LDS>(act, a.lut, a.W));
    const int32_t r = m.r[s], d = (m.f[s] >> GU_CELL_TERM_BIT) & 1;
    a.pos[e] = s;
    a.reward[e] = r;
    a.done[e] = d;
    // zero-copy host path (GU_F_PINNED_IO): results also go straight to the caller's page-locked buffers over PCIe
    if (a.host_obs) a.host_obs[e] = s;
    if (a.host_reward) a.host_reward[e] = r;
    if (a.host_done) a.host_done[e] = d;
}

// ------------------------------------------------------------------------------------
// fused rollout: T env-steps per lane in one launch
//   algorithmic HBM bytes per env-step with GU_F_TRAJECTORY: 3 x 4 B row writes = 12 B
//   (+4 B action read for GU_POLICY_STREAM); state is loaded/stored once per launch.
// ------------------------------------------------------------------------------------
// GU_POLICY_SAMPLE draws a = #{k < 3 : u >= p0 + .. + pk} with u = word / 2^32 (oracle/gu_rng.py).  Both sides of
// u >= c scale exactly by 2^32, and the word is an integer, so the test is word >= ceil(c * 2^32): three uint32
// thresholds per state (x = always, y, z) and a mask of the sums no word can reach (c * 2^32 > 2^32 - 1, or NaN).
// The float64 prefix sums are formed here, once per rollout, in the oracle's order; the step then costs one 16-byte
// read and three integer compares.
__global__ void __launch_bounds__(256) gu_pi_threshold_kernel(const double *pi, int32_t S, uint4 *thr)
{
    const int32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= S) return;
    const double4 p = *reinterpret_cast<const double4 *>(pi + 4 * (int64_t)s);
    const double c[3] = {p.x, __dadd_rn(p.x, p.y), __dadd_rn(__dadd_rn(p.x, p.y), p.z)};
    uint32_t t[3], never = 0;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double x = __dmul_rn(c[k], 4294967296.0);  // exact (power of two)
        if (!(x <= 4294967295.0)) {
            never |= 1u << k;
            t[k] = 0xFFFFFFFFu;
        } else {
            t[k] = x <= 0.0 ? 0u : (uint32_t)ceil(x);
        }
    }
    thr[s] = make_uint4(t[0], t[1], t[2], never);
}

__device__ __forceinline__ uint32_t gu_sample_action(uint32_t word, const uint4 q)
{
    return (uint32_t)(word >= q.x && !(q.w & 1u)) + (uint32_t)(word >= q.y && !(q.w & 2u)) + (uint32_t)(word >= q.z && !(q.w & 4u));
}

struct RolloutArgs {
    const uint8_t *cell;
    const uint8_t *greedy;  // first-argmax action per state (GU_POLICY_GREEDY)
    const uint4 *pi_thr;    // [S] inverse-CDF thresholds of the action probabilities (GU_POLICY_SAMPLE)
    int32_t S, pi_lds;      // pi_lds: the threshold table fits in LDS behind the two grid planes
    int32_t cell_bytes, W;
    uint64_t lut;
    int32_t *pos, *reward, *done;
    uint32_t *episode;
    const uint32_t *tcount;  // per-env offsets
    const int32_t *starts;
    const int32_t *actions;  // [T][N]
    int32_t *tr_obs, *tr_reward, *tr_done;  // [T][N] each
    int32_t *ret, *episodes_fin;
    uint32_t n_starts, seed_prefix, env_id0, steps_taken;
    int64_t N, T;
    GridSel gs;
};

// AUTO: 0 = no reset; 1 = auto-reset, single start cell (branch-free selects keyed on the TERM bit of the
//       register copy of flags); 2 = auto-reset, several start cells (RNG stream 1, rare divergent branch)
// MAP : 0 = records read from L2 (any grid size, any grid-per-env assignment)
//       1 = the block's grid staged in LDS, shared by its lanes
//       2 = every lane keeps a PRIVATE copy of its own grid's flags plane in LDS (multi-grid engines whose groups
//           do not align with blocks, e.g. one maze per env; 64-lane blocks, S16 + 16 bytes per lane)
#define GU_PRIVATE_PAD 16
// TRAJ: 0 = no trajectory; 1 = int32 obs / reward / done rows (12 B per env-step);
//       2 = ONE packed uint32 row: obs | (reward & 0xFF) << 16 | done << 24 (4 B per env-step, grids up to 65 536 cells)
template <int POLICY, int AUTO, int TRAJ, bool STATS, int MAP>
__global__ void __launch_bounds__(GU_BLOCK) gu_rollout_kernel(const RolloutArgs a)
{
    constexpr bool LDS = MAP == 1;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    CellMap m = gu_stage_map<LDS>(a.cell, a.cell_bytes, smem, a.gs);
    const uint8_t *greedy = a.greedy;
    if (LDS && POLICY == GU_POLICY_GREEDY) {
        uint8_t *dst = smem + 2 * a.cell_bytes;
        for (int32_t i = threadIdx.x * 16; i < a.cell_bytes; i += blockDim.x * 16)
            *reinterpret_cast<uint4 *>(dst + i) = *reinterpret_cast<const uint4 *>(a.greedy + i);
        __syncthreads();
        greedy = dst;
    }
    // the LDS copy keeps its own pointer (never merged with the global one): a pointer that may be either becomes a
    // FLAT load, whose wait also covers every trajectory store still in flight
    const bool thr_in_lds = LDS && POLICY == GU_POLICY_SAMPLE && a.pi_lds;
    uint4 *thr_lds = reinterpret_cast<uint4 *>(smem + 2 * a.cell_bytes);
    if (thr_in_lds) {
        for (int32_t i = threadIdx.x; i < a.S; i += blockDim.x) thr_lds[i] = a.pi_thr[i];
        __syncthreads();
    }
    const int64_t e64 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e64 >= a.N) return;
    const uint32_t e = (uint32_t)e64;
    LaneGrid lg = gu_lane_grid<LDS>(a.gs, a.starts, a.n_starts, e, m);
    if (MAP == 2) {  // copy this lane's own flags plane (which also carries the reward code) into its LDS slice
        uint8_t *mine = smem + threadIdx.x * (a.cell_bytes + GU_PRIVATE_PAD);
        for (int32_t i = 0; i < a.cell_bytes; i += 16)
            *reinterpret_cast<uint4 *>(mine + i) = *reinterpret_cast<const uint4 *>(m.f + i);
        m.f = mine;
    }

    int32_t s = a.pos[e];
    int32_t r = a.reward[e];
    uint32_t d = (uint32_t)a.done[e];
    uint32_t ep = a.episode[e];
    const uint32_t t_lane = a.tcount[e] + a.steps_taken;
    const uint32_t prefix = gu_rng_prefix(a.seed_prefix, a.env_id0 + e);
    uint32_t flags = m.f[s];
    int32_t ret = 0, fin = 0;
    const int32_t W = a.W;
    const uint64_t lut = a.lut;
    const int32_t start0 = lg.starts[0];
    const uint32_t start0_flags = m.f[start0];
    // Trajectory rows are addressed as buffer resource (wave-uniform base, rebuilt per 16-step chunk)
    // + lane byte offset e4 (VGPR) + scalar row offset (SGPR): buffer_store_dword ... offen, so that
    // advancing a row costs SALU only and no per-lane 64-bit address arithmetic.
    char *po = (char *)a.tr_obs, *pr = (char *)a.tr_reward, *pd = (char *)a.tr_done;
    const char *pa = (const char *)a.actions;
    const uint32_t e4 = e * 4u;
    const int64_t row = a.N * 4;
    const uint32_t row32 = (uint32_t)row;  // gu_create caps N at 2^25, so lane offset + 15 rows < 2^31 bytes
    __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(po, 0, 0xFFFFFFFFu, 0x00020000);
    __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(pr, 0, 0xFFFFFFFFu, 0x00020000);
    __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(pd, 0, 0xFFFFFFFFu, 0x00020000);
    auto rebase = [&](int64_t rows) {
        po += rows * row;
        pr += rows * row;
        pd += rows * row;
        ro = __builtin_amdgcn_make_buffer_rsrc(po, 0, 0xFFFFFFFFu, 0x00020000);
        rr = __builtin_amdgcn_make_buffer_rsrc(pr, 0, 0xFFFFFFFFu, 0x00020000);
        rd = __builtin_amdgcn_make_buffer_rsrc(pd, 0, 0xFFFFFFFFu, 0x00020000);
    };

    // AUTO == 1 keeps the invariant "d == TERM bit of the REGISTER copy of flags", so the lazy reset needs no
    // separate test on the dependent chain; at entry the stored done flag may disagree with the cell (fresh reset
    // onto a terminal start, gu_set_state), so the register copy takes its TERM bit from the stored flag.
    if (AUTO == 1) flags = (flags & ~GU_CELL_TERM) | (d << GU_CELL_TERM_BIT);

    // `soff`: wave-uniform byte offset of this step's row from the resource base
    auto step = [&](uint32_t act, uint32_t soff) {
        const int32_t delta = gu_delta<MAP != 0>(act, lut, W);
        if (AUTO == 1) {
            // lazy `if done: env.reset()` (env:187-193) with a single start cell: two selects keyed directly on the
            // TERM bit of the record that just arrived (no separate done register on the dependent chain).  A variant
            // that precomputes the move from the start cell off the chain was measured slower at every occupancy
            // (profiles/r01e_auto_form_ab.txt).
            const bool was_done = flags & GU_CELL_TERM;
            ep += was_done;
            s = was_done ? start0 : s;
            flags = was_done ? start0_flags : flags;
            s = gu_move(s, flags, act, delta);
        } else {
            if (AUTO == 2) {
                if (d) {
                    s = lg.starts[gu_rng_start_index(prefix, ep, lg.n_starts)];
                    ++ep;
                    flags = m.f[s];
                }
            }
            s = gu_move(s, flags, act, delta);
        }
        flags = m.f[s];
        r = (MAP == 2) ? gu_reward_packed(flags) : (int32_t)m.r[s];
        d = __builtin_amdgcn_ubfe(flags, GU_CELL_TERM_BIT, 1);
        if (STATS) {
            ret += r;
            fin += (int32_t)d;
        }
        if (TRAJ == 1) {
            __builtin_amdgcn_raw_buffer_store_b32(s, ro, e4, soff, 0);
            __builtin_amdgcn_raw_buffer_store_b32(r, rr, e4, soff, 0);
            __builtin_amdgcn_raw_buffer_store_b32((int32_t)d, rd, e4, soff, 0);
        } else if (TRAJ == 2) {
            __builtin_amdgcn_raw_buffer_store_b32((int32_t)((uint32_t)s | (((uint32_t)r & 0xFFu) << 16) | (d << 24)), ro, e4, soff, 0);
        }
    };
    auto step1 = [&](uint32_t act) {  // one step, then advance the resource base by one row
        step(act, 0);
        if (TRAJ) rebase(1);
    };

    if (POLICY == GU_POLICY_UNIFORM) {
        // Fast path: every lane of the wave is at the same step count (always true unless
        // gu_set_state installed per-env counters), so the 16-actions-per-word schedule is
        // wave-uniform: constant bit-field offsets, one hash per 16 steps.
        const uint32_t t_first = __builtin_amdgcn_readfirstlane(t_lane);
        if (__all(t_lane == t_first)) {
            uint32_t t = t_first;
            int64_t i = 0;
            if (t & 15u) {  // head: finish the current word
                const uint32_t word = gu_rng_word(prefix, GU_RNG_STREAM_ACTION, t >> 4);
                for (; i < a.T && (t & 15u); ++i, ++t) step1((word >> (2u * (t & 15u))) & 3u);
            }
            for (; i + 16 <= a.T; i += 16, t += 16) {  // body: 16 steps per word, fully unrolled
                const uint32_t word = gu_rng_word(prefix, GU_RNG_STREAM_ACTION, t >> 4);
#pragma unroll
                for (uint32_t j = 0; j < 16; ++j) step(__builtin_amdgcn_ubfe(word, 2 * j, 2), j * row32);
                if (TRAJ) rebase(16);
            }
            if (i < a.T) {  // tail
                const uint32_t word = gu_rng_word(prefix, GU_RNG_STREAM_ACTION, t >> 4);
                for (uint32_t j = 0; i < a.T; ++i, ++j) step1((word >> (2u * j)) & 3u);
            }
        } else {
            uint32_t t = t_lane;
            uint32_t word = gu_rng_word(prefix, GU_RNG_STREAM_ACTION, t >> 4);
            for (int64_t i = 0; i < a.T; ++i) {
                step1((word >> (2u * (t & 15u))) & 3u);
                ++t;
                if ((t & 15u) == 0u) word = gu_rng_word(prefix, GU_RNG_STREAM_ACTION, t >> 4);
            }
        }
    } else if (POLICY == GU_POLICY_STREAM) {
        // Action rows are read 8 at a time, one chunk AHEAD of the steps that consume them: the loads are
        // independent of the env state, so with one wave per SIMD this is what hides their HBM latency.
        constexpr int CH = 8;
        int64_t i = 0;
        uint32_t cur[CH], nxt[CH];
        auto load_chunk = [&](uint32_t (&dst)[CH], const char *base) {
            const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 0xFFFFFFFFu, 0x00020000);
#pragma unroll
            for (int j = 0; j < CH; ++j) dst[j] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(ra, e4, j * row32, 0);
        };
        if (a.T >= CH) load_chunk(cur, pa);
        for (; i + CH <= a.T; i += CH) {
            pa += CH * row;
            if (i + 2 * CH <= a.T) load_chunk(nxt, pa);
#pragma unroll
            for (int j = 0; j < CH; ++j) step(cur[j] & 3u, j * row32);
            if (TRAJ) rebase(CH);
#pragma unroll
            for (int j = 0; j < CH; ++j) cur[j] = nxt[j];
        }
        for (; i < a.T; ++i) {  // tail
            const uint32_t act = (uint32_t)(*(const int32_t *)(pa + e4)) & 3u;
            pa += row;
            step1(act);
        }
    } else {
        // Table policies: greedy[] / the sampling thresholds are read at the post-reset position, so the lazy reset
        // is explicit here.  8 steps per resource rebase (scalar row offsets, as on the uniform path); the sampling
        // word of the NEXT step is hashed while this step's threshold read is in flight (it does not depend on s).
        uint32_t t = t_lane;
        uint32_t word = POLICY == GU_POLICY_SAMPLE ? gu_rng_word(prefix, GU_RNG_STREAM_SAMPLE, t) : 0u;
        auto run = [&](auto thr_at) {
            auto tstep = [&](uint32_t soff) {
                if (AUTO == 1) {
                    const bool was_done = flags & GU_CELL_TERM;
                    s = was_done ? start0 : s;
                    ep += was_done;
                    flags = was_done ? (start0_flags & ~GU_CELL_TERM) : flags;
                    d = 0;
                } else if (AUTO == 2) {
                    if (d) {
                        s = lg.starts[gu_rng_start_index(prefix, ep, lg.n_starts)];
                        ++ep;
                        flags = m.f[s];
                        d = 0;
                    }
                }
                uint32_t act;
                if (POLICY == GU_POLICY_GREEDY) {
                    act = greedy[s];
                } else {
                    // inverse CDF of pi[s] on one uniform 32-bit word (RNG stream 2, counter = step count), as integer
                    // thresholds (gu_pi_threshold_kernel)
                    const uint4 q = thr_at(s);
                    const uint32_t next_word = gu_rng_word(prefix, GU_RNG_STREAM_SAMPLE, t + 1u);
                    act = gu_sample_action(word, q);
                    word = next_word;
                }
                ++t;
                step(act, soff);
            };
            int64_t i = 0;
            for (; i + 8 <= a.T; i += 8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) tstep(j * row32);
                if (TRAJ) rebase(8);
            }
            for (; i < a.T; ++i) {
                tstep(0);
                if (TRAJ) rebase(1);
            }
        };
        if (thr_in_lds) run([thr_lds](int32_t at) { return thr_lds[at]; });
        else run([&a](int32_t at) { return a.pi_thr[at]; });
    }
    a.pos[e] = s;
    a.reward[e] = r;
    a.done[e] = (int32_t)d;
    a.episode[e] = ep;
    if (STATS) {
        a.ret[e] = ret;
        a.episodes_fin[e] = fin;
    }
}

// ------------------------------------------------------------------------------------
// look_step_ahead for n (state, action) pairs (env:136-155), both care_about_terminal modes
// ------------------------------------------------------------------------------------
struct LookArgs {
    const uint8_t *cell_move;  // OPEN bits: absorbing map (care=True) or raw map (care=False)
    int32_t cell_bytes, W, S;
    uint64_t lut;
    const int32_t *states, *actions;
    int32_t *next, *reward, *done;
    int64_t n;
};

template <bool LDS>
__global__ void __launch_bounds__(GU_BLOCK) gu_lookahead_kernel(const LookArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const CellMap m = gu_stage_map<LDS>(a.cell_move, a.cell_bytes, smem, GridSel{0, 0, nullptr, 1, 0});
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    int32_t s = a.states[i];
    const uint32_t act = (uint32_t)a.actions[i] & 3u;
    s = gu_move(s, m.f[s], act, gu_delta<LDS>(act, a.lut, a.W));
    a.next[i] = s;
    a.reward[i] = m.r[s];  // reward / terminal bits are identical in both maps
    a.done[i] = (m.f[s] >> GU_CELL_TERM_BIT) & 1;
}

// ------------------------------------------------------------------------------------
// episode-done compaction: wave ballot -> 64-bit mask per wave -> ordered index list
// ------------------------------------------------------------------------------------
__global__ void __launch_bounds__(GU_BLOCK) gu_done_ballot_kernel(const int32_t *__restrict__ done, int64_t N,
                                                                  uint64_t *__restrict__ bits)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool flag = (e < N) && done[e] != 0;
    const uint64_t m = __ballot(flag);  // 64-bit on gfx950
    if ((threadIdx.x & 63) == 0 && (e >> 6) < ((N + 63) >> 6)) bits[e >> 6] = m;
}

// One 1024-thread block: thread i owns a contiguous chunk of ballot words; exclusive scan of
// the per-thread popcounts in LDS, then each thread expands its words in ascending order.
__global__ void __launch_bounds__(1024) gu_done_compact_kernel(const uint64_t *__restrict__ bits, int64_t n_words,
                                                               int32_t *__restrict__ idx, int32_t *__restrict__ count)
{
    __shared__ int32_t part[1024];
    const int tid = threadIdx.x;
    const int64_t chunk = (n_words + 1023) / 1024;
    const int64_t w0 = tid * chunk;
    const int64_t w1 = (w0 + chunk < n_words) ? w0 + chunk : n_words;
    int32_t mine = 0;
    for (int64_t w = w0; w < w1; ++w) mine += __popcll(bits[w]);
    part[tid] = mine;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
        int32_t v = (tid >= off) ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int32_t out = part[tid] - mine;
    if (tid == 1023) *count = part[1023];
    for (int64_t w = w0; w < w1; ++w) {
        uint64_t m = bits[w];
        while (m) {
            const int b = __ffsll((long long)m) - 1;
            idx[out++] = (int32_t)(w * 64 + b);
            m &= m - 1;
        }
    }
}

// ------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------
static inline unsigned gu_blocks(int64_t n, int block) { return (unsigned)((n + block - 1) / block); }

// Largest block size <= preferred for which every block uses one grid (0 = none: use the L2 variant)
static int gu_lds_block(const gu_engine *h, int preferred, int planes)
{
    if (h->S > GU_MAX_LDS_CELLS || (size_t)planes * h->cell_bytes > 65536) return 0;
    if (h->n_grids == 1) return preferred;
    for (int bs = preferred; bs >= 64; bs >>= 1)
        if (h->group % bs == 0) return bs;
    return 0;
}

static int gu_rollout_block()
{
    static int cached = 0;
    if (!cached) {
        const char *s = std::getenv("GU_ROLLOUT_BLOCK");
        int v = s ? std::atoi(s) : 256;
        cached = (v == 64 || v == 128 || v == 256) ? v : 256;
    }
    return cached;
}

int gu_launch_reset(gu_engine *h, const uint8_t *d_mask, const int32_t *d_choice, bool only_done)
{
    ResetArgs a{h->pos(), h->done(), h->d_episode, h->d_starts, d_mask, d_choice,
                (uint32_t)h->n_starts, h->seed_prefix, (uint32_t)h->env_id0, h->N, only_done ? 1 : 0, gu_grid_sel(h)};
    hipLaunchKernelGGL(gu_reset_kernel, dim3(gu_blocks(h->N, GU_BLOCK)), dim3(GU_BLOCK), 0, h->stream, a);
    GU_HIP(hipGetLastError());
    return GU_OK;
}

int gu_launch_step(gu_engine *h, const int32_t *d_actions_row, uint32_t flags, int32_t *host_obs, int32_t *host_reward,
                   int32_t *host_done)
{
    StepArgs a{h->d_cell, h->cell_bytes, h->W, h->delta_lut, d_actions_row, h->pos(), h->reward(), h->done(),
               h->d_episode, h->d_starts, (uint32_t)h->n_starts, h->seed_prefix, (uint32_t)h->env_id0, h->N, flags,
               gu_grid_sel(h), host_obs, host_reward, host_done};
    const int lds_bs = gu_lds_block(h, GU_BLOCK, 2);
    if (lds_bs)
        hipLaunchKernelGGL(gu_step_kernel<true>, dim3(gu_blocks(h->N, lds_bs)), dim3(lds_bs), 2 * (size_t)h->cell_bytes, h->stream, a);
    else
        hipLaunchKernelGGL(gu_step_kernel<false>, dim3(gu_blocks(h->N, GU_BLOCK)), dim3(GU_BLOCK), 0, h->stream, a);
    GU_HIP(hipGetLastError());
    h->steps_taken += 1;
    return GU_OK;
}

template <int POLICY, int AUTO, int TRAJ, bool STATS>
static void gu_rollout_launch(gu_engine *h, const RolloutArgs &a, int bs)
{
    const int planes = POLICY == GU_POLICY_GREEDY ? 3 : 2;
    const int lds_bs = gu_lds_block(h, bs, planes);
    if (lds_bs) {
        size_t lds = (size_t)planes * h->cell_bytes;
        RolloutArgs b = a;
        if (POLICY == GU_POLICY_SAMPLE && lds + (size_t)h->S * sizeof(uint4) <= 65536) {
            b.pi_lds = 1;
            lds += (size_t)h->S * sizeof(uint4);
        }
        hipLaunchKernelGGL((gu_rollout_kernel<POLICY, AUTO, TRAJ, STATS, 1>), dim3(gu_blocks(h->N, lds_bs)), dim3(lds_bs), lds, h->stream, b);
        return;
    }
    if constexpr (POLICY == GU_POLICY_UNIFORM || POLICY == GU_POLICY_STREAM) {
        // misaligned multi-grid engine (e.g. one maze per env): private per-lane copies in LDS if 64 of them fit
        const size_t priv = 64 * ((size_t)h->cell_bytes + GU_PRIVATE_PAD);
        if (h->n_grids > 1 && h->W <= 32767 && priv <= 160 * 1024) {
            auto kern = gu_rollout_kernel<POLICY, AUTO, TRAJ, STATS, 2>;
            if (priv > 64 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)priv);
            hipLaunchKernelGGL(kern, dim3(gu_blocks(h->N, 64)), dim3(64), priv, h->stream, a);
            return;
        }
    }
    hipLaunchKernelGGL((gu_rollout_kernel<POLICY, AUTO, TRAJ, STATS, 0>), dim3(gu_blocks(h->N, bs)), dim3(bs), 0, h->stream, a);
}

template <int POLICY, int AUTO>
static void gu_rollout_dispatch2(gu_engine *h, const RolloutArgs &a, int traj, bool stats, int bs)
{
    if (traj == 1) {
        if (stats) gu_rollout_launch<POLICY, AUTO, 1, true>(h, a, bs);
        else gu_rollout_launch<POLICY, AUTO, 1, false>(h, a, bs);
    } else if (traj == 2) {
        if (stats) gu_rollout_launch<POLICY, AUTO, 2, true>(h, a, bs);
        else gu_rollout_launch<POLICY, AUTO, 2, false>(h, a, bs);
    } else {
        if (stats) gu_rollout_launch<POLICY, AUTO, 0, true>(h, a, bs);
        else gu_rollout_launch<POLICY, AUTO, 0, false>(h, a, bs);
    }
}

template <int POLICY>
static void gu_rollout_dispatch(gu_engine *h, const RolloutArgs &a, int auto_mode, int traj, bool stats, int bs)
{
    switch (auto_mode) {
    case 0: gu_rollout_dispatch2<POLICY, 0>(h, a, traj, stats, bs); break;
    case 1: gu_rollout_dispatch2<POLICY, 1>(h, a, traj, stats, bs); break;
    default: gu_rollout_dispatch2<POLICY, 2>(h, a, traj, stats, bs); break;
    }
}

int gu_launch_rollout(gu_engine *h, int64_t T, int32_t policy, uint32_t flags)
{
    const int traj = (flags & GU_F_PACKED) ? 2 : ((flags & GU_F_TRAJECTORY) ? 1 : 0);
    const bool stats = flags & GU_F_STATS;
    const int auto_mode = (flags & GU_F_AUTO_RESET) ? (h->all_single_start ? 1 : 2) : 0;
    const int64_t rows = traj ? h->traj_T * h->N : 0;
    RolloutArgs a{};
    a.cell = h->d_cell;
    a.greedy = h->d_greedy;
    a.pi_thr = h->d_pi_thr;
    a.S = h->S;
    a.pi_lds = 0;
    a.cell_bytes = h->cell_bytes;
    a.W = h->W;
    a.lut = h->delta_lut;
    a.pos = h->pos();
    a.reward = h->reward();
    a.done = h->done();
    a.episode = h->d_episode;
    a.tcount = h->d_tcount;
    a.starts = h->d_starts;
    a.actions = h->d_actions;
    a.tr_obs = h->d_traj;
    a.tr_reward = h->d_traj ? h->d_traj + rows : nullptr;
    a.tr_done = h->d_traj ? h->d_traj + 2 * rows : nullptr;
    a.ret = h->d_ret;
    a.episodes_fin = h->d_episodes_fin;
    a.n_starts = (uint32_t)h->n_starts;
    a.seed_prefix = h->seed_prefix;
    a.env_id0 = (uint32_t)h->env_id0;
    a.steps_taken = h->steps_taken;
    a.N = h->N;
    a.T = T;
    a.gs = gu_grid_sel(h);
    const int bs = gu_rollout_block();
    if (policy == GU_POLICY_SAMPLE)
        hipLaunchKernelGGL(gu_pi_threshold_kernel, dim3(gu_blocks(h->S, 256)), dim3(256), 0, h->stream, h->d_pi[h->vi_cur], h->S, h->d_pi_thr);
    switch (policy) {
    case GU_POLICY_UNIFORM: gu_rollout_dispatch<GU_POLICY_UNIFORM>(h, a, auto_mode, traj, stats, bs); break;
    case GU_POLICY_STREAM: gu_rollout_dispatch<GU_POLICY_STREAM>(h, a, auto_mode, traj, stats, bs); break;
    case GU_POLICY_GREEDY: gu_rollout_dispatch<GU_POLICY_GREEDY>(h, a, auto_mode, traj, stats, bs); break;
    case GU_POLICY_SAMPLE: gu_rollout_dispatch<GU_POLICY_SAMPLE>(h, a, auto_mode, traj, stats, bs); break;
    default: return gu_fail(GU_ERR_INVALID, "unknown policy kind %d", policy);
    }
    GU_HIP(hipGetLastError());
    h->steps_taken += (uint32_t)T;
    return GU_OK;
}

int gu_launch_lookahead(gu_engine *h, int64_t n, const int32_t *d_states, const int32_t *d_actions, bool care,
                        int32_t *d_next, int32_t *d_reward, int32_t *d_done)
{
    LookArgs a{care ? h->d_cell : h->d_cell_raw, h->cell_bytes, h->W, h->S, h->delta_lut, d_states, d_actions,
               d_next, d_reward, d_done, n};
    const dim3 grid(gu_blocks(n, GU_BLOCK)), block(GU_BLOCK);
    if (h->S <= GU_MAX_LDS_CELLS)
        hipLaunchKernelGGL(gu_lookahead_kernel<true>, grid, block, 2 * (size_t)h->cell_bytes, h->stream, a);
    else
        hipLaunchKernelGGL(gu_lookahead_kernel<false>, grid, block, 0, h->stream, a);
    GU_HIP(hipGetLastError());
    return GU_OK;
}

int gu_launch_done_compact(gu_engine *h)
{
    const int64_t n_words = (h->N + 63) / 64;
    hipLaunchKernelGGL(gu_done_ballot_kernel, dim3(gu_blocks(h->N, GU_BLOCK)), dim3(GU_BLOCK), 0, h->stream,
                       h->done(), h->N, h->d_done_bits);
    hipLaunchKernelGGL(gu_done_compact_kernel, dim3(1), dim3(1024), 0, h->stream, h->d_done_bits, n_words,
                       h->d_done_idx, h->d_done_count);
    GU_HIP(hipGetLastError());
    return GU_OK;
}
