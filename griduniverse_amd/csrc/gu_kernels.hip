// gu_kernels.hip -- step / reset / rollout kernels for gfx950 (CDNA4, wave64).
//
// One wavefront lane per env instance.  Every kernel first stages the grid's per-cell
// record map (one byte per cell, see gu_internal.hpp) from L2 into LDS with 16-byte
// loads; the transition of core/envs/griduniverse_env.py:136-155 is then ONE LDS byte
// read per env-step (the record of the cell the agent lands on) plus a handful of
// integer VALU ops:
//
//     blocked = (rec >> a) & 1          edge / wall-at-candidate / absorbing terminal
//     s      += blocked ? 0 : delta[a]  delta = {-W, +1, +W, -1}   (env:51-54)
//     rec     = cell[s]                 LDS
//     reward  = rec&RMINUS ? -10 : rec&RPLUS ? +10 : -1            (env:80-90)
//     done    = rec&TERM                                           (env:163-168)
//
// State traffic is coalesced int32 SoA: lane e touches word e of pos[] / reward[] /
// done[] / actions[] and of each trajectory row.  This is HBM-bound integer work:
// no MFMA, no inter-block reuse (so no XCD-aware block remap is needed -- the only
// shared data is the <=64 KiB record map, which every XCD's L2 holds after first touch).
#include "gu_internal.hpp"
#include "gu_rng.hpp"

#include <cstdlib>

#define GU_BLOCK 256

// ------------------------------------------------------------------------------------
// helpers
// ------------------------------------------------------------------------------------
// action -> state delta LUT (env:51-56): UP -W, RIGHT +1, DOWN +W, LEFT -1.  Kept in
// registers as arithmetic on the two action bits; an LDS-resident 4-entry table would
// put a second ds_read on every step for no gain.
__device__ __forceinline__ int32_t gu_delta(uint32_t a, int32_t W)
{
    const int32_t sign = (int32_t)(a & 2u) - 1;          // a=0,1 -> -1 ; a=2,3 -> +1
    return (a & 1u) ? -sign : sign * W;                  // RIGHT(1): +1, LEFT(3): -1, UP(0): -W, DOWN(2): +W
}

__device__ __forceinline__ int32_t gu_reward_of(uint32_t rec)
{
    return (rec & GU_CELL_RMINUS) ? -10 : ((rec & GU_CELL_RPLUS) ? 10 : -1);
}

// cooperative global -> LDS copy of `bytes16` (multiple of 16) bytes
__device__ __forceinline__ void gu_stage(const uint8_t *__restrict__ src, uint8_t *dst, int32_t bytes16)
{
    for (int32_t i = threadIdx.x * 16; i < bytes16; i += blockDim.x * 16)
        *reinterpret_cast<uint4 *>(dst + i) = *reinterpret_cast<const uint4 *>(src + i);
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
};

__global__ void __launch_bounds__(GU_BLOCK) gu_reset_kernel(const ResetArgs a)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.N) return;
    if (a.mask && !a.mask[e]) return;
    if (a.only_done && !a.done[e]) return;
    uint32_t ep = a.episode[e];
    uint32_t idx;
    if (a.choice) {
        idx = (uint32_t)a.choice[e];
        if (idx >= a.n_starts) idx = 0;  // host validates; never index out of the table
    } else {
        idx = gu_rng_start_index(gu_rng_prefix(a.seed_prefix, a.env_id0 + (uint32_t)e), ep, a.n_starts);
    }
    a.pos[e] = a.starts[idx];
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
    const int32_t *actions;
    int32_t *pos, *reward, *done;
    uint32_t *episode;
    const int32_t *starts;
    uint32_t n_starts, seed_prefix, env_id0;
    int64_t N;
    uint32_t flags;
};

template <bool LDS>
__global__ void __launch_bounds__(GU_BLOCK) gu_step_kernel(const StepArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint8_t *cell = a.cell;
    if (LDS) {
        gu_stage(a.cell, smem, a.cell_bytes);
        __syncthreads();
        cell = smem;
    }
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.N) return;
    const uint32_t act = (uint32_t)a.actions[e] & 3u;
    int32_t s = a.pos[e];
    if ((a.flags & GU_F_AUTO_RESET) && a.done[e]) {  // lazy `if done: env.reset()`
        const uint32_t ep = a.episode[e];
        s = a.starts[gu_rng_start_index(gu_rng_prefix(a.seed_prefix, a.env_id0 + (uint32_t)e), ep, a.n_starts)];
        a.episode[e] = ep + 1;
    }
    uint32_t rec = cell[s];
    const bool blocked = (rec >> act) & 1u;
    s = blocked ? s : s + gu_delta(act, a.W);
    rec = cell[s];
    a.pos[e] = s;
    a.reward[e] = gu_reward_of(rec);
    a.done[e] = (rec & GU_CELL_TERM) ? 1 : 0;
}

// ------------------------------------------------------------------------------------
// fused rollout: T env-steps per lane in one launch
//   algorithmic HBM bytes per env-step with GU_F_TRAJECTORY: 3 x 4 B row writes = 12 B
//   (+4 B action read for GU_POLICY_STREAM); state is loaded/stored once per launch.
// ------------------------------------------------------------------------------------
struct RolloutArgs {
    const uint8_t *cell;
    const uint8_t *greedy;  // first-argmax action per state (GU_POLICY_GREEDY)
    int32_t cell_bytes, W;
    int32_t *pos, *reward, *done;
    uint32_t *episode;
    const uint32_t *tcount;  // per-env offsets
    const int32_t *starts;
    const int32_t *actions;  // [T][N]
    int32_t *tr_obs, *tr_reward, *tr_done;  // [T][N] each
    int32_t *ret, *episodes_fin;
    uint32_t n_starts, seed_prefix, env_id0, steps_taken;
    int64_t N, T;
    uint32_t flags;
};

template <int POLICY, bool TRAJ, bool LDS>
__global__ void __launch_bounds__(GU_BLOCK) gu_rollout_kernel(const RolloutArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint8_t *cell = a.cell;
    const uint8_t *greedy = a.greedy;
    if (LDS) {
        gu_stage(a.cell, smem, a.cell_bytes);
        if (POLICY == GU_POLICY_GREEDY) gu_stage(a.greedy, smem + a.cell_bytes, a.cell_bytes);
        __syncthreads();
        cell = smem;
        greedy = smem + a.cell_bytes;
    }
    const int64_t e64 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e64 >= a.N) return;
    const uint32_t e = (uint32_t)e64;
    const bool auto_reset = a.flags & GU_F_AUTO_RESET;

    int32_t s = a.pos[e];
    int32_t r = a.reward[e];
    uint32_t d = (uint32_t)a.done[e];
    uint32_t ep = a.episode[e];
    uint32_t t = a.tcount[e] + a.steps_taken;
    const uint32_t prefix = gu_rng_prefix(a.seed_prefix, a.env_id0 + e);
    uint32_t rec = cell[s];
    uint32_t word = 0;
    if (POLICY == GU_POLICY_UNIFORM) word = gu_rng_word(prefix, GU_RNG_STREAM_ACTION, t >> 4);
    int32_t ret = 0, fin = 0;
    const int32_t W = a.W;

    for (int64_t i = 0; i < a.T; ++i) {
        if (auto_reset && d) {  // lazy `if done: env.reset()` (env:187-193)
            s = a.starts[gu_rng_start_index(prefix, ep, a.n_starts)];
            ++ep;
            rec = cell[s];
        }
        uint32_t act;
        if (POLICY == GU_POLICY_UNIFORM) {
            act = (word >> (2u * (t & 15u))) & 3u;
        } else if (POLICY == GU_POLICY_STREAM) {
            act = (uint32_t)a.actions[i * a.N + e] & 3u;
        } else {
            act = greedy[s];
        }
        const bool blocked = (rec >> act) & 1u;
        s = blocked ? s : s + gu_delta(act, W);
        rec = cell[s];
        r = gu_reward_of(rec);
        d = (rec >> 4) & 1u;
        ret += r;
        fin += (int32_t)d;
        if (TRAJ) {
            const int64_t o = i * a.N + e;
            a.tr_obs[o] = s;
            a.tr_reward[o] = r;
            a.tr_done[o] = (int32_t)d;
        }
        ++t;
        if (POLICY == GU_POLICY_UNIFORM) {
            if ((t & 15u) == 0u) word = gu_rng_word(prefix, GU_RNG_STREAM_ACTION, t >> 4);
        }
    }
    a.pos[e] = s;
    a.reward[e] = r;
    a.done[e] = (int32_t)d;
    a.episode[e] = ep;
    if (a.flags & GU_F_STATS) {
        a.ret[e] = ret;
        a.episodes_fin[e] = fin;
    }
}

// ------------------------------------------------------------------------------------
// look_step_ahead for n (state, action) pairs (env:136-155), both care_about_terminal modes
// ------------------------------------------------------------------------------------
struct LookArgs {
    const uint8_t *cell_move;  // blocked bits: absorbing map (care=True) or raw map (care=False)
    int32_t cell_bytes, W, S;
    const int32_t *states, *actions;
    int32_t *next, *reward, *done;
    int64_t n;
};

template <bool LDS>
__global__ void __launch_bounds__(GU_BLOCK) gu_lookahead_kernel(const LookArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint8_t *cell = a.cell_move;
    if (LDS) {
        gu_stage(a.cell_move, smem, a.cell_bytes);
        __syncthreads();
        cell = smem;
    }
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    int32_t s = a.states[i];
    const uint32_t act = (uint32_t)a.actions[i] & 3u;
    uint32_t rec = cell[s];
    const bool blocked = (rec >> act) & 1u;
    s = blocked ? s : s + gu_delta(act, a.W);
    rec = cell[s];  // reward / terminal bits are identical in both maps
    a.next[i] = s;
    a.reward[i] = gu_reward_of(rec);
    a.done[i] = (rec & GU_CELL_TERM) ? 1 : 0;
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

static int gu_rollout_block()
{
    static int cached = 0;
    if (!cached) {
        const char *s = std::getenv("GU_ROLLOUT_BLOCK");
        int v = s ? std::atoi(s) : 64;
        cached = (v == 64 || v == 128 || v == 256) ? v : 64;
    }
    return cached;
}

int gu_launch_reset(gu_engine *h, const uint8_t *d_mask, const int32_t *d_choice, bool only_done)
{
    ResetArgs a{h->pos(), h->done(), h->d_episode, h->d_starts, d_mask, d_choice,
                (uint32_t)h->n_starts, h->seed_prefix, (uint32_t)h->env_id0, h->N, only_done ? 1 : 0};
    hipLaunchKernelGGL(gu_reset_kernel, dim3(gu_blocks(h->N, GU_BLOCK)), dim3(GU_BLOCK), 0, h->stream, a);
    GU_HIP(hipGetLastError());
    return GU_OK;
}

int gu_launch_step(gu_engine *h, const int32_t *d_actions_row, uint32_t flags)
{
    StepArgs a{h->d_cell, h->cell_bytes, h->W, d_actions_row, h->pos(), h->reward(), h->done(), h->d_episode,
               h->d_starts, (uint32_t)h->n_starts, h->seed_prefix, (uint32_t)h->env_id0, h->N, flags};
    const dim3 grid(gu_blocks(h->N, GU_BLOCK)), block(GU_BLOCK);
    if (h->S <= GU_MAX_LDS_CELLS)
        hipLaunchKernelGGL(gu_step_kernel<true>, grid, block, (size_t)h->cell_bytes, h->stream, a);
    else
        hipLaunchKernelGGL(gu_step_kernel<false>, grid, block, 0, h->stream, a);
    GU_HIP(hipGetLastError());
    h->steps_taken += 1;
    return GU_OK;
}

template <int POLICY>
static void gu_rollout_dispatch(gu_engine *h, const RolloutArgs &a, bool traj, int bs)
{
    const dim3 grid(gu_blocks(h->N, bs)), block(bs);
    const bool lds = h->S <= GU_MAX_LDS_CELLS / (POLICY == GU_POLICY_GREEDY ? 2 : 1);
    const size_t smem = lds ? (size_t)h->cell_bytes * (POLICY == GU_POLICY_GREEDY ? 2 : 1) : 0;
    if (traj) {
        if (lds) hipLaunchKernelGGL((gu_rollout_kernel<POLICY, true, true>), grid, block, smem, h->stream, a);
        else hipLaunchKernelGGL((gu_rollout_kernel<POLICY, true, false>), grid, block, 0, h->stream, a);
    } else {
        if (lds) hipLaunchKernelGGL((gu_rollout_kernel<POLICY, false, true>), grid, block, smem, h->stream, a);
        else hipLaunchKernelGGL((gu_rollout_kernel<POLICY, false, false>), grid, block, 0, h->stream, a);
    }
}

int gu_launch_rollout(gu_engine *h, int64_t T, int32_t policy, uint32_t flags)
{
    const bool traj = flags & GU_F_TRAJECTORY;
    const int64_t rows = traj ? h->traj_T * h->N : 0;
    RolloutArgs a{};
    a.cell = h->d_cell;
    a.greedy = h->d_greedy;
    a.cell_bytes = h->cell_bytes;
    a.W = h->W;
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
    a.flags = flags;
    const int bs = gu_rollout_block();
    switch (policy) {
    case GU_POLICY_UNIFORM: gu_rollout_dispatch<GU_POLICY_UNIFORM>(h, a, traj, bs); break;
    case GU_POLICY_STREAM: gu_rollout_dispatch<GU_POLICY_STREAM>(h, a, traj, bs); break;
    case GU_POLICY_GREEDY: gu_rollout_dispatch<GU_POLICY_GREEDY>(h, a, traj, bs); break;
    default: return gu_fail(GU_ERR_INVALID, "unknown policy kind %d", policy);
    }
    GU_HIP(hipGetLastError());
    h->steps_taken += (uint32_t)T;
    return GU_OK;
}

int gu_launch_lookahead(gu_engine *h, int64_t n, const int32_t *d_states, const int32_t *d_actions, bool care,
                        int32_t *d_next, int32_t *d_reward, int32_t *d_done)
{
    LookArgs a{care ? h->d_cell : h->d_cell_raw, h->cell_bytes, h->W, h->S, d_states, d_actions, d_next, d_reward, d_done, n};
    const dim3 grid(gu_blocks(n, GU_BLOCK)), block(GU_BLOCK);
    if (h->S <= GU_MAX_LDS_CELLS)
        hipLaunchKernelGGL(gu_lookahead_kernel<true>, grid, block, (size_t)h->cell_bytes, h->stream, a);
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
