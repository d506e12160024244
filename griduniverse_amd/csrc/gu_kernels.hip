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
// one wave per SIMD; measured 3 % faster than 1024 one-wave workgroups, profiles/archive/r01b_bench_blocksize.txt).
//
// The action -> delta LUT is four int16 lanes of one 64-bit scalar register (v_lshrrev_b64 +
// v_bfe_i32): staging a 4-entry table in LDS instead would put a second, dependent ds_read on
// every step for no gain.
#include "gu_rollout.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

// ------------------------------------------------------------------------------------
// reset: GridUniverseEnv._reset (env:187-193) for the masked / done envs
// ------------------------------------------------------------------------------------
struct ResetArgs {
    int32_t *pos, *done;
    uint32_t *episode;
    const int32_t *starts;
    const uint8_t *mask;
    const int32_t *choice;
    uint64_t *done_bits;  // [ceil(N/64)] wave ballots of the done flags, kept current by every kernel that writes done[]
    uint32_t n_starts, seed_prefix, env_id0;
    int64_t N;
    int32_t only_done;
    GridSel gs;
};

__global__ void __launch_bounds__(GU_BLOCK) gu_reset_kernel(const ResetArgs a)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = e < a.N;
    int32_t still_done = 0;
    if (live) {
        bool take = !a.mask || a.mask[e];
        const int32_t was_done = a.done[e];
        if (a.only_done && !was_done) take = false;
        if (take) {
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
        } else {
            still_done = was_done;
        }
    }
    const uint64_t bits = __ballot(still_done != 0);
    if ((threadIdx.x & 63) == 0 && live) a.done_bits[e >> 6] = bits;
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
    uint32_t *tcount;  // per-env step-count offsets (touched only by a lane whose action is rejected)
    const int32_t *starts;
    uint32_t n_starts, seed_prefix, env_id0;
    int64_t N;
    uint32_t flags;
    GridSel gs;
    int32_t *host_obs, *host_reward, *host_done;  // optional page-locked host mirrors written by the kernel itself
    uint32_t *host_seq;  // optional page-locked completion word: set to `seq` after every block's mirror stores
    uint32_t seq;
    uint32_t *blocks_done;  // device counter behind the completion word (zero between launches)
    uint32_t *host_err;     // optional page-locked error word: actions come from the caller and are validated HERE
    uint64_t *done_bits;    // [ceil(N/64)] wave ballots of the new done flags (episode-done compaction, gu_done_indices)
};

template <bool LDS>
__global__ void __launch_bounds__(GU_BLOCK) gu_step_kernel(const StepArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    CellMap m = gu_stage_map<LDS>(a.cell, a.cell_bytes, smem, a.gs);
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int32_t d = 0;
    if (e < a.N) {
        const LaneGrid lg = gu_lane_grid<LDS>(a.gs, a.starts, a.n_starts, (uint32_t)e, m);
        const uint32_t raw = (uint32_t)a.actions[e];
        const uint32_t act = raw & 3u;
        int32_t s = a.pos[e], r;
        if (a.host_err && !GU_ACTION_OK(raw)) {
            // Action outside -4..3 (env:148 raises IndexError before it touches the instance): this env does not
            // step -- position, flags, pending lazy reset and step count stay as they were -- and the host is told
            // through the page-locked error word (any offender may write it; the host finds the first one itself).
            __hip_atomic_store(a.host_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            a.tcount[e] -= 1u;  // the lock-step counter advances for the whole batch
            r = a.reward[e];
            d = a.done[e];
        } else {
            if ((a.flags & GU_F_AUTO_RESET) && a.done[e]) {  // lazy `if done: env.reset()`
                const uint32_t ep = a.episode[e];
                s = lg.starts[gu_rng_start_index(gu_rng_prefix(a.seed_prefix, a.env_id0 + (uint32_t)e), ep, lg.n_starts)];
                a.episode[e] = ep + 1;
            }
            s = gu_move(s, m.f[s], act, gu_delta<LDS>(act, a.lut, a.W));
            r = m.r[s];
            d = (m.f[s] >> GU_CELL_TERM_BIT) & 1;
            a.pos[e] = s;
            a.reward[e] = r;
            a.done[e] = d;
        }
        // zero-copy host path (GU_F_PINNED_IO): results also go straight to the caller's page-locked buffers over PCIe
        if (a.host_obs) a.host_obs[e] = s;
        if (a.host_reward) a.host_reward[e] = r;
        if (a.host_done) a.host_done[e] = d;
    }
    // episode-done compaction, first half: one 64-bit ballot word per wave (every workgroup size used is a multiple of 64)
    const uint64_t bits = __ballot(d != 0);
    if ((threadIdx.x & 63) == 0 && e < a.N) a.done_bits[e >> 6] = bits;
    if (a.host_seq) {
        // Completion word: the host spins on a sequence number in page-locked memory instead of going through the
        // runtime's completion path (no interrupt, no runtime call).  Every block makes its mirror stores visible to the
        // host, then counts itself; the block that completes the count publishes the number (and re-arms the counter).
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t arrived = atomicAdd(a.blocks_done, 1u);
            if (arrived == gridDim.x - 1) {
                *a.blocks_done = 0u;
                __threadfence_system();
                __hip_atomic_store(a.host_seq, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// Validation of a caller-supplied action stream on the device (gu_upload_actions): any value outside -4..3 raises the
// page-locked error word.  Replaces a serial host loop over T x N values.
__global__ void __launch_bounds__(256) gu_validate_actions_kernel(const int32_t *__restrict__ actions, int64_t count, uint32_t *host_err)
{
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
        bad |= !GU_ACTION_OK(actions[i]);
    if (__ballot(bad) && (threadIdx.x & 63) == 0) __hip_atomic_store(host_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The uploaded stream as the rollout kernels read it: word [k][env] holds the two-bit actions of steps 16 k .. 16 k + 15 of
// that env (step t in bits 2 (t & 15) ..), the shape of the uniform policy's RNG word -- one 4-byte read per env and 16 steps
// instead of 64 bytes.  Validates while it packs (the int32 rows stay: the single-step launches read those).
__global__ void __launch_bounds__(256) gu_pack_actions_kernel(const int32_t *__restrict__ actions, int64_t N, int64_t T, uint32_t *__restrict__ packed,
                                                              uint32_t *host_err)
{
    const int64_t words = (T + 15) / 16 * N;
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = i / N, e = i - k * N;
        uint32_t word = 0;
        for (int64_t j = 0; j < 16 && 16 * k + j < T; ++j) {
            const uint32_t act = (uint32_t)actions[(16 * k + j) * N + e];
            bad |= !GU_ACTION_OK(act);
            word |= (act & 3u) << (2 * j);
        }
        packed[i] = word;
    }
    if (__ballot(bad) && (threadIdx.x & 63) == 0) __hip_atomic_store(host_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------------------------
// sampling thresholds for GU_POLICY_SAMPLE (the rollout kernel itself: gu_rollout.hpp)
// ------------------------------------------------------------------------------------
// GU_POLICY_SAMPLE draws a = #{k < 3 : u >= p0 + .. + pk} with u = word / 2^32 (oracle/gu_rng.py).  Both sides of
// u >= c scale exactly by 2^32, and the word is an integer, so the test is word >= ceil(c * 2^32): three uint32
// thresholds per state.  A sum that no word can reach (c * 2^32 > 2^32 - 1, or NaN) is stored as threshold 0 -- passed by
// every word -- and counted in the fourth component, which gu_sample_action subtracts again.
// The float64 prefix sums are formed here, once per rollout, in the oracle's order; the step then costs one 16-byte
// read, three integer compares with carry adds and one subtract.
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
            ++never;
            t[k] = 0u;
        } else {
            t[k] = x <= 0.0 ? 0u : (uint32_t)ceil(x);
        }
    }
    thr[s] = make_uint4(t[0], t[1], t[2], never);
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
    uint32_t *host_err;  // page-locked error word: a state outside the grid or an action outside 0..3
};

template <bool LDS>
__global__ void __launch_bounds__(GU_BLOCK) gu_lookahead_kernel(const LookArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const CellMap m = gu_stage_map<LDS>(a.cell_move, a.cell_bytes, smem, GridSel{0, 0, nullptr, 1, 0, 0});
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n) return;
    int32_t s = a.states[i];
    const uint32_t raw = (uint32_t)a.actions[i];
    if ((uint32_t)s >= (uint32_t)a.S || !GU_ACTION_OK(raw)) {  // validated here instead of in a serial host loop
        __hip_atomic_store(a.host_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    const uint32_t act = raw & 3u;  // (-4 .. -1: the list's negative indices)
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
// the per-thread popcounts in LDS, then each thread expands its words in ascending order.  idx / count point into the
// engine's page-locked staging block: the list lands in host memory without a copy command.
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
int gu_launch_reset(gu_engine *h, const uint8_t *d_mask, const int32_t *d_choice, bool only_done)
{
    h->entry_table_ok = false;  // (a reset may land on a terminal start cell: done = 0 on a terminal cell)
    int trail_rc = gu_trail_before_reset(h, d_mask, only_done);  // (reads the done flags the reset is about to clear)
    if (trail_rc != GU_OK) return trail_rc;
    ResetArgs a{h->pos(), h->done(), h->d_episode, h->d_starts, d_mask, d_choice, h->d_done_bits,
                (uint32_t)h->n_starts, h->seed_prefix, (uint32_t)h->env_id0, h->N, only_done ? 1 : 0, gu_grid_sel(h)};
    hipLaunchKernelGGL(gu_reset_kernel, dim3(gu_blocks(h->N, GU_BLOCK)), dim3(GU_BLOCK), 0, h->stream, a);
    GU_HIP(hipGetLastError());
    return GU_OK;
}

int gu_launch_step(gu_engine *h, const int32_t *d_actions_row, uint32_t flags, int32_t *host_obs, int32_t *host_reward,
                   int32_t *host_done, uint32_t *host_seq, uint32_t seq, uint32_t *host_err)
{
    h->entry_table_ok = false;
    StepArgs a{h->d_cell, h->cell_bytes, h->W, h->delta_lut, d_actions_row, h->pos(), h->reward(), h->done(),
               h->d_episode, h->d_tcount, h->d_starts, (uint32_t)h->n_starts, h->seed_prefix, (uint32_t)h->env_id0, h->N, flags,
               gu_grid_sel(h), host_obs, host_reward, host_done, host_seq, seq, h->d_blocks_done, host_err, h->d_done_bits};
    const int lds_bs = gu_lds_block(h, GU_BLOCK, 2);
    if (lds_bs)
        hipLaunchKernelGGL(gu_step_kernel<true>, dim3(gu_blocks(h->N, lds_bs)), dim3(lds_bs), 2 * (size_t)h->cell_bytes, h->stream, a);
    else
        hipLaunchKernelGGL(gu_step_kernel<false>, dim3(gu_blocks(h->N, GU_BLOCK)), dim3(GU_BLOCK), 0, h->stream, a);
    GU_HIP(hipGetLastError());
    h->steps_taken += 1;
    return gu_trail_after_step(h, flags);
}

static double gu_wall_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double g_last_rollout_ms[64];  // per device: when this process last launched a rollout there (is the device at its working clocks?)

__global__ void __launch_bounds__(256) gu_copy_kernel(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, size_t words)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int gu_device_copy(gu_engine *h, void *dst, const void *src, size_t bytes)
{
    if (!bytes) return GU_OK;
    GU_REQUIRE(bytes % 4 == 0 && ((uintptr_t)dst | (uintptr_t)src) % 4 == 0, GU_ERR_INVALID, "gu_device_copy: %zu bytes / unaligned", bytes);
    const size_t words = bytes / 4;
    const unsigned blocks = (unsigned)std::min<size_t>((words + 255) / 256, (size_t)h->n_cu * 8);
    hipLaunchKernelGGL(gu_copy_kernel, dim3(blocks), dim3(256), 0, h->stream, (uint32_t *)dst, (const uint32_t *)src, words);
    GU_HIP(hipGetLastError());
    return GU_OK;
}

// Up to GU_MAX_SEGMENTS device-to-device copies and fills with zero in ONE launch (a launch costs ~6 us of stream time; the
// snapshot and the zeroed exchange buffers of the one-launch DP forms were seven of them per call).  src == nullptr: zero.
__global__ void __launch_bounds__(256) gu_segments_kernel(const GuSegments s)
{
    const size_t step = (size_t)gridDim.x * blockDim.x, first = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int k = 0; k < s.n; ++k) {
        uint32_t *dst = (uint32_t *)s.dst[k];
        const uint32_t *src = (const uint32_t *)s.src[k];
        if (src) {
            for (size_t i = first; i < s.words[k]; i += step) dst[i] = src[i];
        } else {
            for (size_t i = first; i < s.words[k]; i += step) dst[i] = 0u;
        }
    }
}

int gu_device_segments(gu_engine *h, const GuSegments &s)
{
    size_t most = 0;
    for (int k = 0; k < s.n; ++k) {
        GU_REQUIRE(((uintptr_t)s.dst[k] | (uintptr_t)s.src[k]) % 4 == 0, GU_ERR_INVALID, "gu_device_segments: segment %d unaligned", k);
        most = std::max(most, s.words[k]);
    }
    if (!most) return GU_OK;
    const unsigned blocks = (unsigned)std::min<size_t>((most + 255) / 256, (size_t)h->n_cu * 8);
    hipLaunchKernelGGL(gu_segments_kernel, dim3(blocks), dim3(256), 0, h->stream, s);
    GU_HIP(hipGetLastError());
    return GU_OK;
}

int gu_read_back(gu_engine *h, void *dst, const void *src, size_t bytes)
{
    if (!bytes) return GU_OK;
    if (h->h_ctl && bytes <= GU_CTL_WORDS * sizeof(unsigned long long)) {
        GU_HIP(hipMemcpyAsync(h->h_ctl, src, bytes, hipMemcpyDeviceToHost, h->stream));
        GU_HIP(hipStreamSynchronize(h->stream));
        memcpy(dst, h->h_ctl, bytes);
        return GU_OK;
    }
    GU_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
    GU_HIP(hipStreamSynchronize(h->stream));
    return GU_OK;
}

static void gu_rollout_general(gu_engine *h, const RolloutArgs &a, int32_t policy, int auto_mode, int traj, bool stats, int bs)
{
    switch (policy) {
    case GU_POLICY_UNIFORM: gu_rollout_uniform(h, a, auto_mode, traj, stats, bs); break;
    case GU_POLICY_STREAM: gu_rollout_stream(h, a, auto_mode, traj, stats, bs); break;
    case GU_POLICY_GREEDY: gu_rollout_greedy(h, a, auto_mode, traj, stats, bs); break;
    default: gu_rollout_sample(h, a, auto_mode, traj, stats, bs); break;
    }
}

// The schedule of this launch.  `slot` = policy * 3 + auto mode (+ 12 for the transition-row kernel's int32 rows, + 24 for its packed
// rows, `row_bytes` = 12 / 4 per env-step).  (The open-loop period search of rounds 3 and 4 is gone from the library: docs/HISTORY.md.)
//   GU_OPT_ROLLOUT_PACE = 0: no limiter.  n > 0: that period, fixed.  -1 (the default): the launches of a kind choose their period
//   themselves, closed loop, from the first one on (GuPacer / gu_pace_next): no search, no dedicated launch, nothing on the host.
// Launches that cannot be bound by the HBM write path (less than 128 MB of rows, or fewer workgroups than half the CUs), launches of
// fewer than 64 steps (fewer than four groups to schedule) and batches of more than four waves per SIMD (524 288 envs and more on
// 256 CUs: a per-wave schedule found nothing to gain there, 0.96 .. 0.98 ms = 6.4 .. 6.6 TB/s with and without, and a batch that does
// not fit the device at once is not on one schedule anyway; profiles/archive/r03n_batch_sizes.txt) keep no schedule and no record --
// a fixed period applies to them all the same.  A kind's ring belongs to one launch SHAPE (trajectory buffer, workgroups, length
// within a factor of two, row bytes): another shape starts it over from the model, the rows of 16 steps at GU_OPT_PACE_TARGET GB/s
// (7200; the cliff sits at 7.4 .. 7.5 TB/s on the allocations measured in rounds 3 and 4).
static int gu_pace_ring_for(gu_engine *h, int slot, int64_t T, unsigned blocks, int block_size, int row_bytes, GuPaceArgs *pace)
{
    const int64_t waves = (h->N + 31) / 32;  // (the most a launch can have: the transition-row kernel's half waves)
    if (!h->d_pace_ring) {
        const size_t ring_bytes = sizeof(GuPaceEntry) * GU_PACE_RING * 36;
        h->pace_slot_stride = (waves + 63) & ~(int64_t)63;
        const size_t slot_bytes = sizeof(uint64_t) * 2 * (size_t)h->pace_slot_stride * 36;
        GU_HIP(hipMalloc((void **)&h->d_pace_ring, ring_bytes));
        GU_HIP(hipMalloc((void **)&h->d_pace_slots, slot_bytes));
        GU_HIP(hipMemsetAsync(h->d_pace_ring, 0, ring_bytes, h->stream));
        GU_HIP(hipMemsetAsync(h->d_pace_slots, 0, slot_bytes, h->stream));
    }
    gu_engine::PaceKind &k = h->pace[slot];
    GuPaceEntry *ring = h->d_pace_ring + (size_t)slot * GU_PACE_RING;
    uint64_t *slots = h->d_pace_slots + (size_t)slot * 2 * (size_t)h->pace_slot_stride;
    const bool same = k.active && k.buffer == (const void *)h->d_traj && k.blocks == blocks && k.row_bytes == row_bytes && !(T > 2 * k.T || 2 * T < k.T);
    if (!same) {
        if (k.seq) {
            GU_HIP(hipMemsetAsync(ring, 0, sizeof(GuPaceEntry) * GU_PACE_RING, h->stream));
            GU_HIP(hipMemsetAsync(slots, 0, sizeof(uint64_t) * 2 * (size_t)h->pace_slot_stride, h->stream));
        }
        k.active = true;
        k.buffer = h->d_traj;
        k.blocks = blocks;
        k.row_bytes = row_bytes;
        k.T = T;
        k.seq = 0;
        const double target = (double)gu_opt(h, GU_OPT_PACE_TARGET) * 1e9;  // bytes per second
        const double ticks = 16.0 * (double)h->N * (double)row_bytes / target * 1e8;
        k.model = (uint32_t)std::min<double>(std::max(1.0, ticks + 0.5), 1e6);
    }
    pace->ring = ring;
    pace->slots = slots;
    pace->seq = ++k.seq;
    pace->period = k.model;
    pace->lo = std::max<uint32_t>(1u, k.model * 3u / 4u);
    pace->hi = std::max<uint32_t>(k.model * 2u, k.model + 4u);
    pace->groups = (uint32_t)std::min<int64_t>(T / 16, 0x7FFFFFFF);
    pace->report_at = (uint32_t)std::min<int64_t>(T > 160 ? T - 96 : std::max<int64_t>(T - 32, 1), 0x7FFFFFFF);
    pace->n_waves = (uint32_t)std::min<int64_t>((int64_t)blocks * (block_size / 64), (int64_t)h->pace_slot_stride);  // the waves of THIS launch
    pace->slot_stride = (uint32_t)h->pace_slot_stride;
    k.n_waves = pace->n_waves;
    pace->bar_num = (uint16_t)gu_opt(h, GU_OPT_PACE_BAR_NUM);
    pace->gain_q = (uint32_t)gu_opt(h, GU_OPT_PACE_GAIN_Q);
    pace->dec_q = (uint32_t)gu_opt(h, GU_OPT_PACE_DEC_Q);
    pace->probe_every = (uint32_t)gu_opt(h, GU_OPT_PACE_PROBE_EVERY);
    pace->adapt = (uint32_t)gu_opt(h, GU_OPT_PACE_ADAPT);
    pace->fixed = 0;
    return GU_OK;
}

static bool gu_pace_eligible(const gu_engine *h, int64_t T, unsigned blocks, int row_bytes)
{
    return !((double)h->N * (double)T * (double)row_bytes < 128e6 || (int64_t)blocks * 2 < h->n_cu || T < 64 || h->N > (int64_t)h->n_cu * 1024);
}

int gu_pace_for(gu_engine *h, int slot, int64_t T, unsigned blocks, int block_size, int row_bytes, GuPaceArgs *pace)
{
    *pace = GuPaceArgs{};
    const int64_t opt = gu_opt(h, GU_OPT_ROLLOUT_PACE);
    if (opt == 0) return GU_OK;
    const bool eligible = gu_pace_eligible(h, T, blocks, row_bytes);
    if (eligible && !(opt > 0 && gu_opt(h, GU_OPT_PACE_RECORD) == 0)) {
        const int rc = gu_pace_ring_for(h, slot, T, blocks, block_size, row_bytes, pace);
        if (rc != GU_OK) return rc;
        if (opt > 0) pace->period = (uint32_t)opt, pace->fixed = 1;  // fixed, and recorded all the same
        if (h->device >= 0 && h->device < 64) g_last_rollout_ms[h->device] = gu_wall_ms();
        return GU_OK;
    }
    if (opt > 0) pace->period = (uint32_t)opt;
    return GU_OK;
}

// Rollout MAP 5 (gu_rollout.hpp): every env's grid at four bits per cell -- RPLUS, RMINUS, TERM, WALL from bit 0 up: the upper half of
// its cell records with the reward code first -- padded with wall cells (one column left of every row, one row above and below: cell (x, y) at (y + 1)(W + 1) + x + 1) and
// laid out per wave of 64 envs as [dword][lane], so that a wave stages its image with one contiguous copy and its gathers are
// free of bank conflicts.  Built once per grid installation, on the device, from the cell planes.
__global__ void __launch_bounds__(256) gu_nibble_planes_kernel(const uint8_t *__restrict__ cell, GridSel gs, int32_t W, int32_t H, int32_t dwords, int64_t N,
                                                               uint32_t *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (wave, dword, lane)
    const int64_t total = ((N + 63) / 64) * dwords * 64;
    if (i >= total) return;
    const int64_t lane = i & 63, j = (i >> 6) % dwords, wave = (i >> 6) / dwords, e = wave * 64 + lane;
    uint32_t word = 0x88888888u;  // (padding, cells past the image, envs past the batch: walls)
    if (e < N) {
        const uint8_t *f = cell + (e / gs.group) * gs.grid_stride;
        word = 0;
        for (int32_t k = 0; k < 8; ++k) {
            const int32_t p = (int32_t)j * 8 + k, yp = p / (W + 1), xp = p % (W + 1);
            const bool real = yp >= 1 && yp <= H && xp >= 1;
            const uint32_t b = real ? f[(yp - 1) * W + xp - 1] : GU_CELL_WALL;
            const uint32_t four = ((b >> 5) & 3u) | (((b >> GU_CELL_TERM_BIT) & 1u) << 2) | ((b >> 7) << 3);
            word |= four << (4 * k);
        }
    }
    out[i] = word;
}

int gu_nibble_planes(gu_engine *h)
{
    if (h->nib_valid) return GU_OK;
    const int32_t dwords = gu_nibble_dwords(h);
    const int64_t total = ((h->N + 63) / 64) * dwords * 64;
    if (!h->d_nib) GU_HIP(hipMalloc((void **)&h->d_nib, (size_t)total * sizeof(uint32_t)));
    hipLaunchKernelGGL(gu_nibble_planes_kernel, dim3(gu_blocks(total, 256)), dim3(256), 0, h->stream, h->d_cell, gu_grid_sel(h), h->W, h->H, dwords, h->N, h->d_nib);
    GU_HIP(hipGetLastError());
    h->nib_valid = true;
    return GU_OK;
}

int gu_launch_rollout(gu_engine *h, int64_t T, int32_t policy, uint32_t flags)
{
    // int32 rows: three planes [T][N], or one plane of (obs, reward, done) triples [T][N][3] -- the same words, one 12-byte store
    // per lane and step (gu_rollout.hpp: TRAJ == 3; the readers take them apart again: gu_read_trajectory, gu_mc_evaluate).
    // GU_OPT_TRAJ_LAYOUT: 0 = planes, 1 = triples wherever possible, -1 (default) = triples where they are faster.  Measured
    // (profiles/archive/r05b_layout_ab.txt, r05c_layout_sizes.txt, five variants interleaved in one process): a launch bound by the HBM write path is
    // SLOWER with triples -- 65 536 envs: 117 against 112 us, a config-4 shard of 32 768: 68 against 63 -- and so is every table
    // policy; a launch of a few waves, bound by the ISSUE of its stores (~25 clocks per 256-byte store of a wave that has its SIMD
    // alone), gains little from the triple alone (config 2, 4096 envs: 49.4 against 49.9 us) but 25 % together with the pair tables
    // (two steps per LDS round trip, two stores per pair instead of six: 37.3 us), up to 8192 envs = one workgroup per eight
    // CUs; at 16 384 the two are level -- unless the batch is spread over twice the waves (32 envs each: 43.6 against 53.3 us,
    // profiles/archive/r05m_half_sizes.txt) --, beyond it the planes win.  So: triples for the uniform policy on the transition-row
    // kernel with pair tables, up to n_cu / 4 workgroups of 256 (half waves for the upper half of that range).  Batches of more than 2^24 envs (lane offset + 15 rows must stay
    // below 2^32 bytes) and engines with the agent trail on always keep the planes.
    int traj = (flags & GU_F_PACKED) ? 2 : ((flags & GU_F_TRAJECTORY) ? 1 : 0);
    if (traj == 1 && h->N <= ((int64_t)1 << 24) && !h->trail_cap) {
        const int64_t layout = gu_opt(h, GU_OPT_TRAJ_LAYOUT);
        if (layout == 1 || (layout == -1 && policy == GU_POLICY_UNIFORM && gu_rows_pairs_fit(h) && (int64_t)gu_blocks(h->N, 256) * 4 <= h->n_cu)) traj = 3;
    }
    h->traj_written = traj;
    const bool stats = flags & GU_F_STATS;
    const int auto_mode = (flags & GU_F_AUTO_RESET) ? (h->all_single_start ? 1 : 2) : 0;
    const int64_t rows = traj ? h->traj_T * h->N : 0;
    GU_REQUIRE(!h->trail_cap || traj, GU_ERR_UNSUPPORTED, "the agent trail is on (gu_trail_enable): a rollout must write rows (GU_F_TRAJECTORY or GU_F_PACKED) to feed it");
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
    a.actions = h->d_actions_packed;
    a.tr_obs = h->d_traj;
    a.tr_reward = h->d_traj ? h->d_traj + rows : nullptr;
    a.tr_done = h->d_traj ? h->d_traj + 2 * rows : nullptr;
    a.ret = h->d_ret;
    a.episodes_fin = h->d_episodes_fin;
    a.done_bits = h->d_done_bits;
    a.n_starts = (uint32_t)h->n_starts;
    a.env_id0 = (uint32_t)h->env_id0;
    a.steps_taken = (uint32_t)h->steps_taken;
    a.steps_hi = (uint32_t)(h->steps_taken >> 32);
    a.seed_prefix0 = h->seed_prefix;
    {   // The epoch (step count >> 32) of RNG streams 0 and 2, csrc/gu_rng.hpp: folded into the seed prefix when every env stays in
        // ONE epoch for the whole launch (the kernels then count in 32 bits, as ever); else -- once in 2^32 steps -- the launch goes
        // to the general kernel, which asks per lane and step.  The stream and greedy policies draw nothing from those streams.
        const bool draws = policy == GU_POLICY_UNIFORM || policy == GU_POLICY_SAMPLE;
        int64_t lo = std::max<int64_t>((int64_t)h->steps_taken + h->off_lo, 0), hi = (int64_t)h->steps_taken + h->off_hi + T - 1;
        if (draws && (lo >> 32) != (hi >> 32) && !h->off_exact) {
            // the bounds are only bounds (envs were held back by rejected actions since the host last knew): read the offsets once,
            // so that launches do not take the slow path for longer than the envs really are on both sides of the boundary
            std::vector<int32_t> off((size_t)h->N);
            GU_HIP(hipMemcpyAsync(off.data(), h->d_tcount, (size_t)h->N * 4, hipMemcpyDeviceToHost, h->stream));
            GU_HIP(hipStreamSynchronize(h->stream));
            const auto mm = std::minmax_element(off.begin(), off.end());
            h->off_lo = *mm.first, h->off_hi = *mm.second, h->off_exact = true;
            lo = std::max<int64_t>((int64_t)h->steps_taken + h->off_lo, 0), hi = (int64_t)h->steps_taken + h->off_hi + T - 1;
        }
        a.straddle = draws && (lo >> 32) != (hi >> 32) ? 1 : 0;
        a.seed_prefix = a.straddle ? h->seed_prefix : gu_rng_seed_prefix_epoch(h->seed_prefix, (uint32_t)(lo >> 32));
    }
    a.N = h->N;
    a.T = T;
    a.gs = gu_grid_sel(h);
    a.rows = nullptr;
    a.rows2 = nullptr;
    a.row_shift = 0;
    a.stream_lds_off = 0;
    a.stream_lds_words = 0;
    a.half_waves = 0;
    a.entry_table = (h->entry_table_ok && gu_opt(h, GU_OPT_ROLLOUT_ENTRY) != 0) ? 1 : 0;
    const int bs = gu_rollout_block(h);
    a.nib = nullptr;
    a.nib_dwords = 0;
    if (h->n_grids > 1 && (policy == GU_POLICY_UNIFORM || policy == GU_POLICY_STREAM) && !gu_lds_block(h, bs, 2) && h->W <= 1022 &&
        gu_nibble_bytes_per_wave(h) <= (size_t)h->lds_per_cu - 512) {  // groups that do not align with blocks (one maze per env): MAP 5
        const int rc = gu_nibble_planes(h);
        if (rc != GU_OK) return rc;
        a.nib = h->d_nib;
        a.nib_dwords = gu_nibble_dwords(h);
        // (cells of the padded image times 32: W <= 1022 keeps a row's step inside an int16)
        const uint64_t wp = (uint64_t)(uint16_t)(int16_t)((h->W + 1) * 32), mwp = (uint64_t)(uint16_t)(int16_t)(-(h->W + 1) * 32);
        a.lut_p = mwp | (32ull << 16) | (wp << 32) | ((uint64_t)(uint16_t)(int16_t)-32 << 48);
    }
    a.pace = GuPaceArgs{};
    a.xcd_remap = gu_opt(h, GU_OPT_ROLLOUT_XCD) != 0 && h->n_grids == 1;  // XCD-aware env-block order (see gu_env_block; measured slower, off)
    if (policy == GU_POLICY_SAMPLE)
        hipLaunchKernelGGL(gu_pi_threshold_kernel, dim3(gu_blocks(h->S, 256)), dim3(256), 0, h->stream, h->d_pi[h->vi_cur], h->S, h->d_pi_thr);
    if (!a.straddle && gu_rollout_multi(h, a, policy, auto_mode, traj, stats)) {
        GU_HIP(hipGetLastError());
        h->steps_taken += (uint64_t)T;
        h->entry_table_ok = true;
        return gu_trail_after_rollout(h, T, traj, auto_mode != 0);
    }
    {
        int rows_rc = GU_OK;
        if (!a.straddle && gu_rollout_rows(h, a, policy, auto_mode, traj, stats, &rows_rc)) {
            if (rows_rc != GU_OK) return rows_rc;
            GU_HIP(hipGetLastError());
            h->steps_taken += (uint64_t)T;
            h->entry_table_ok = true;
            if (h->device >= 0 && h->device < 64) g_last_rollout_ms[h->device] = gu_wall_ms();
            return gu_trail_after_rollout(h, T, traj, auto_mode != 0);
        }
    }
    if (policy < GU_POLICY_UNIFORM || policy > GU_POLICY_SAMPLE) return gu_fail(GU_ERR_INVALID, "unknown policy kind %d", policy);
    if (traj == 1 || traj == 3) {  // int32 rows on the general kernel: the store stream is rate-limited (gu_rollout.hpp: GuPacer)
        int rc = gu_pace_for(h, policy * 3 + auto_mode, T, gu_blocks(h->N, bs), bs, 12, &a.pace);
        if (rc != GU_OK) return rc;
        gu_rollout_general(h, a, policy, auto_mode, traj, stats, bs);
    } else {
        gu_rollout_general(h, a, policy, auto_mode, traj, stats, bs);
    }
    if (h->device >= 0 && h->device < 64) g_last_rollout_ms[h->device] = gu_wall_ms();
    GU_HIP(hipGetLastError());
    h->steps_taken += (uint64_t)T;
    h->entry_table_ok = true;
    return gu_trail_after_rollout(h, T, traj, auto_mode != 0);
}

int gu_launch_lookahead(gu_engine *h, int64_t n, const int32_t *d_states, const int32_t *d_actions, bool care,
                        int32_t *d_next, int32_t *d_reward, int32_t *d_done)
{
    LookArgs a{care ? h->d_cell : h->d_cell_raw, h->cell_bytes, h->W, h->S, h->delta_lut, d_states, d_actions,
               d_next, d_reward, d_done, n, h->h_seq + GU_HOST_ERR_WORD};
    const dim3 grid(gu_blocks(n, GU_BLOCK)), block(GU_BLOCK);
    if (h->S <= GU_MAX_LDS_CELLS)
        hipLaunchKernelGGL(gu_lookahead_kernel<true>, grid, block, 2 * (size_t)h->cell_bytes, h->stream, a);
    else
        hipLaunchKernelGGL(gu_lookahead_kernel<false>, grid, block, 0, h->stream, a);
    GU_HIP(hipGetLastError());
    return GU_OK;
}

int gu_launch_validate_actions(gu_engine *h, const int32_t *d_actions, int64_t count)
{
    const unsigned blocks = (unsigned)std::min<int64_t>((count + 255) / 256, 4096);
    hipLaunchKernelGGL(gu_validate_actions_kernel, dim3(blocks), dim3(256), 0, h->stream, d_actions, count, h->h_seq + GU_HOST_ERR_WORD);
    GU_HIP(hipGetLastError());
    return GU_OK;
}

int gu_launch_pack_actions(gu_engine *h, int64_t T)
{
    const int64_t words = (T + 15) / 16 * h->N;
    const unsigned blocks = (unsigned)std::min<int64_t>((words + 255) / 256, 8192);
    hipLaunchKernelGGL(gu_pack_actions_kernel, dim3(blocks), dim3(256), 0, h->stream, h->d_actions, (int64_t)h->N, T, h->d_actions_packed,
                       h->h_seq + GU_HOST_ERR_WORD);
    GU_HIP(hipGetLastError());
    return GU_OK;
}

// One full write of a candidate trajectory buffer in the rollout's own store shape (three int32 rows per step, one lane per
// env, workgroups of 256), timed with events: gu_alloc_trajectory keeps the allocation that HBM takes fastest.
__global__ void __launch_bounds__(GU_BLOCK) gu_traj_probe_kernel(int32_t *__restrict__ buf, int64_t N, int64_t T)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    const int64_t plane = N * T;
    int64_t o = e;
    for (int64_t t = 0; t < T; ++t, o += N) {
        buf[o] = 0;
        buf[plane + o] = 0;
        buf[2 * plane + o] = 0;
    }
}

int gu_probe_trajectory_buffer(gu_engine *h, int32_t *buf, int64_t T, float *ms)
{
    const dim3 grid(gu_blocks(h->N, GU_BLOCK)), block(GU_BLOCK);
    hipLaunchKernelGGL(gu_traj_probe_kernel, grid, block, 0, h->stream, buf, h->N, T);  // first touch
    GU_HIP(hipEventRecord(h->ev_begin, h->stream));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(gu_traj_probe_kernel, grid, block, 0, h->stream, buf, h->N, T);
    GU_HIP(hipEventRecord(h->ev_end, h->stream));
    GU_HIP(hipEventSynchronize(h->ev_end));
    GU_HIP(hipGetLastError());
    GU_HIP(hipEventElapsedTime(ms, h->ev_begin, h->ev_end));
    *ms /= 3.0f;
    return GU_OK;
}

// int32 triples [count][3] (GU_OPT_TRAJ_LAYOUT = 1) -> three planes [3][count], for the readers that hand planes to the host
__global__ void __launch_bounds__(256) gu_deinterleave_kernel(const int32_t *__restrict__ triples, int32_t *__restrict__ planes, int64_t count)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
        planes[i] = triples[3 * i];
        planes[count + i] = triples[3 * i + 1];
        planes[2 * count + i] = triples[3 * i + 2];
    }
}

int gu_launch_deinterleave(gu_engine *h, const int32_t *triples, int32_t *planes, int64_t count)
{
    const unsigned blocks = (unsigned)std::min<int64_t>((count + 255) / 256, (int64_t)h->n_cu * 16);
    hipLaunchKernelGGL(gu_deinterleave_kernel, dim3(blocks), dim3(256), 0, h->stream, triples, planes, count);
    GU_HIP(hipGetLastError());
    return GU_OK;
}

int gu_launch_done_compact(gu_engine *h)
{
    const int64_t n_words = (h->N + 63) / 64;
    // The ballot words are written by the step / rollout / reset kernels themselves; only a done[] installed from the
    // host (gu_set_state) needs the separate ballot pass.
    if (!h->done_bits_valid) {
        hipLaunchKernelGGL(gu_done_ballot_kernel, dim3(gu_blocks(h->N, GU_BLOCK)), dim3(GU_BLOCK), 0, h->stream,
                           h->done(), h->N, h->d_done_bits);
        h->done_bits_valid = true;
    }
    hipLaunchKernelGGL(gu_done_compact_kernel, dim3(1), dim3(1024), 0, h->stream, h->d_done_bits, n_words,
                       h->h_pin, (int32_t *)(h->h_seq + GU_HOST_COUNT_WORD));
    GU_HIP(hipGetLastError());
    return GU_OK;
}
