// gu_mc_ref.hip -- the REFERENCE'S OWN episodes on the device: core/algorithms/monte_carlo.py:7-26 (`run_episode`) draws one
// `np.random.choice(4, p=policy[obs])` per step from numpy's GLOBAL stream, episodes strictly one after the other, so episode e + 1
// begins with the uniform behind the last one episode e used -- a chain that looks sequential.  Until round 4 the host walked it
// (monte_carlo_evaluation(rng='numpy'): 18.6 ms per 100 episodes, 181 per 1000, against 0.6 ms for the device's own RNG).
//
// It is sequential only in WHERE an episode begins, not in what it does from there: `np.random.choice(n, p)` consumes exactly one
// uniform per call (cdf = p.cumsum(); cdf /= cdf[-1]; cdf.searchsorted(u, side='right')), so the episode that begins at uniform i
// in start cell c is a pure function of (i, c).  Hence
//   gu_mc_walk_lengths  : for EVERY offset i of a block of pre-drawn uniforms (and every start cell the grid has) one lane walks
//                         the episode that would begin there and records its length -- speculative, massively parallel;
//   (host)              : follows the chain offset -> offset + length through that table: n_episodes table look-ups;
//   gu_mc_walk_episodes : one lane per episode walks it again from its now known offset and writes the (obs, reward, done) rows
//                         into the trajectory buffer, exactly as the STREAM rollout of the host-built action table did (rows past an
//                         episode's end: the absorbing state), ready for gu_mc_evaluate.
// The host draws the uniforms in bulk (numpy: random_sample(a) then random_sample(b) == random_sample(a + b)) and afterwards puts
// the global stream back to exactly as many draws as the episodes used; the start cells come from the stdlib's global stream, one
// draw per episode, which does not depend on the walks at all (griduniverse_amd/algorithms/monte_carlo.py).
#include "gu_map.hpp"

#include <vector>

struct McWalkArgs {
    const uint8_t *cell;   // absorbing-aware planes [flags | reward]
    int32_t cell_bytes, W, S;
    const double *u;       // [K] uniforms of numpy's global stream, in order
    int64_t K;
    const double *cdf;     // [S][4] the reference's normalised cumulative rows
    int32_t cap;           // run_episode's max_steps_per_episode
    int32_t lds;           // 1: flags and cdf rows are staged in LDS
};

// one step of run_episode: the action of uniform `x` in state s (searchsorted(cdf[s], x, side='right') = how many entries are <= x;
// the last entry is exactly 1.0 > x), then env.step (env:136-185 through the absorbing-aware cell records)
__device__ __forceinline__ uint32_t mc_action(const double *row, double x)
{
    const double4 c = *reinterpret_cast<const double4 *>(row);
    const uint32_t a = (uint32_t)(c.x <= x) + (uint32_t)(c.y <= x) + (uint32_t)(c.z <= x) + (uint32_t)(c.w <= x);
    return a > 3u ? 3u : a;
}

__global__ void __launch_bounds__(256) gu_mc_walk_lengths_kernel(const McWalkArgs a, const int32_t *__restrict__ starts, int64_t n_offsets,
                                                                 uint16_t *__restrict__ len_out)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t mc_lds[];
    const uint8_t *flags = a.cell;
    const double *cdf = a.cdf;
    if (a.lds) {
        double *c = reinterpret_cast<double *>(mc_lds);
        uint8_t *f = mc_lds + (size_t)a.S * 32;
        for (int32_t i = threadIdx.x; i < a.S * 4; i += blockDim.x) c[i] = a.cdf[i];
        for (int32_t i = threadIdx.x; i < a.S; i += blockDim.x) f[i] = a.cell[i];
        __syncthreads();
        flags = f;
        cdf = c;
    }
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_offsets) return;
    int32_t s = starts[blockIdx.y];
    uint32_t len = 0xFFFFu;  // "ran out of uniforms before the episode ended"
    const int64_t avail = a.K - i;
    double x = avail > 0 ? a.u[i] : 0.0;
    for (int32_t t = 0; t < a.cap; ++t) {
        if (t >= avail) break;
        const double next_x = t + 1 < avail ? a.u[i + t + 1] : 0.0;  // (does not depend on the walk: asked for ahead of it)
        const uint32_t act = mc_action(cdf + 4 * (int64_t)s, x);
        const uint32_t f = flags[s];
        s += ((f >> act) & 1u) ? gu_delta<false>(act, 0, a.W) : 0;
        if (flags[s] & GU_CELL_TERM) {
            len = (uint32_t)t + 1u;
            break;
        }
        x = next_x;
        if (t + 1 == a.cap) len = (uint32_t)a.cap;
    }
    len_out[(int64_t)blockIdx.y * n_offsets + i] = (uint16_t)len;
}

__global__ void __launch_bounds__(64) gu_mc_walk_episodes_kernel(const McWalkArgs a, const int64_t *__restrict__ offsets, const int32_t *__restrict__ first_state,
                                                                 int64_t N, int64_t T, int32_t *__restrict__ obs, int32_t *__restrict__ reward,
                                                                 int32_t *__restrict__ done)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t mc_lds[];
    const uint8_t *flags = a.cell;
    const int8_t *rew = reinterpret_cast<const int8_t *>(a.cell + a.cell_bytes);
    const double *cdf = a.cdf;
    if (a.lds) {
        double *c = reinterpret_cast<double *>(mc_lds);
        uint8_t *f = mc_lds + (size_t)a.S * 32;
        int8_t *r = reinterpret_cast<int8_t *>(f + a.S);
        for (int32_t i = threadIdx.x; i < a.S * 4; i += blockDim.x) c[i] = a.cdf[i];
        for (int32_t i = threadIdx.x; i < a.S; i += blockDim.x) f[i] = a.cell[i], r[i] = rew[i];
        __syncthreads();
        flags = f, rew = r, cdf = c;
    }
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    int32_t s = first_state[e];
    const int64_t i = offsets[e];
    bool over = false;
    double x = i < a.K ? a.u[i] : 0.0;
    for (int64_t t = 0; t < T; ++t) {
        const double next_x = i + t + 1 < a.K ? a.u[i + t + 1] : 0.0;  // (does not depend on the walk: asked for ahead of it)
        if (!over) {  // (an episode that is over draws nothing: its rows repeat the absorbing state, as a rollout without auto-reset leaves them)
            const uint32_t act = mc_action(cdf + 4 * (int64_t)s, x);
            const uint32_t f = flags[s];
            s += ((f >> act) & 1u) ? gu_delta<false>(act, 0, a.W) : 0;
        }
        x = next_x;
        const uint32_t d = ((uint32_t)flags[s] >> GU_CELL_TERM_BIT) & 1u;
        obs[t * N + e] = s;
        reward[t * N + e] = (int32_t)rew[s];
        done[t * N + e] = (int32_t)d;
        over = over || d != 0u || t + 1 >= a.cap;
    }
}

static int mc_walk_args(gu_engine *h, McWalkArgs *a, const double *d_u, int64_t K, const double *d_cdf, int32_t cap)
{
    a->cell = h->d_cell;
    a->cell_bytes = h->cell_bytes;
    a->W = h->W;
    a->S = h->S;
    a->u = d_u;
    a->K = K;
    a->cdf = d_cdf;
    a->cap = cap;
    a->lds = (int64_t)h->S * 33 <= 60 * 1024 ? 1 : 0;
    return GU_OK;
}

extern "C" int gu_mc_walk_lengths(gu_handle h, int64_t K, const double *u, int64_t n_offsets, int32_t n_starts, const int32_t *start_states,
                                  int32_t cap, const double *cdf, uint16_t *lengths)
{
    int rc = gu_use_device(h);
    if (rc != GU_OK) return rc;
    GU_REQUIRE(h->has_grid && h->n_grids == 1, GU_ERR_STATE, "gu_mc_walk_lengths needs a single-grid engine");
    GU_REQUIRE(K > 0 && u && n_offsets > 0 && n_offsets <= K && n_starts > 0 && n_starts <= 65535 && start_states && cdf && lengths, GU_ERR_INVALID, "bad arguments");
    GU_REQUIRE(cap > 0 && cap <= 65534, GU_ERR_UNSUPPORTED, "episodes of up to 65534 steps");
    for (int32_t k = 0; k < n_starts; ++k)
        GU_REQUIRE(start_states[k] >= 0 && start_states[k] < h->S, GU_ERR_INVALID, "start state %d outside the grid", start_states[k]);
    // scratch: u [K] | cdf [S][4] | starts | lengths [n_starts][n_offsets]
    const size_t off_cdf = (size_t)K * 8, off_starts = off_cdf + (size_t)h->S * 32, off_len = (off_starts + (size_t)n_starts * 4 + 15) & ~(size_t)15;
    const size_t len_bytes = (size_t)n_starts * (size_t)n_offsets * 2;
    rc = gu_ensure_scratch(h, off_len + len_bytes);
    if (rc != GU_OK) return rc;
    char *base = (char *)h->d_scratch;
    GU_HIP(hipMemcpyAsync(base, u, (size_t)K * 8, hipMemcpyHostToDevice, h->stream));
    GU_HIP(hipMemcpyAsync(base + off_cdf, cdf, (size_t)h->S * 32, hipMemcpyHostToDevice, h->stream));
    GU_HIP(hipMemcpyAsync(base + off_starts, start_states, (size_t)n_starts * 4, hipMemcpyHostToDevice, h->stream));
    McWalkArgs a;
    mc_walk_args(h, &a, (const double *)base, K, (const double *)(base + off_cdf), cap);
    const dim3 grid((unsigned)((n_offsets + 255) / 256), (unsigned)n_starts);
    hipLaunchKernelGGL(gu_mc_walk_lengths_kernel, grid, dim3(256), a.lds ? (size_t)h->S * 33 : 0, h->stream, a, (const int32_t *)(base + off_starts), n_offsets,
                       (uint16_t *)(base + off_len));
    GU_HIP(hipGetLastError());
    GU_HIP(hipMemcpyAsync(lengths, base + off_len, len_bytes, hipMemcpyDeviceToHost, h->stream));
    GU_HIP(hipStreamSynchronize(h->stream));
    return GU_OK;
}

extern "C" int gu_mc_walk_episodes(gu_handle h, int64_t K, const double *u, const double *cdf, const int64_t *offsets, const int32_t *first_state,
                                   int32_t cap, int64_t T)
{
    int rc = gu_use_device(h);
    if (rc != GU_OK) return rc;
    GU_REQUIRE(h->has_grid && h->n_grids == 1, GU_ERR_STATE, "gu_mc_walk_episodes needs a single-grid engine");
    GU_REQUIRE(K >= 0 && u && cdf && offsets && first_state && cap > 0 && cap <= 65534, GU_ERR_INVALID, "bad arguments");
    GU_REQUIRE(h->d_traj && T > 0 && T <= h->traj_T, GU_ERR_STATE, "trajectory buffer does not hold %lld rows: call gu_reserve_trajectory", (long long)T);
    GU_REQUIRE(!h->trail_cap, GU_ERR_UNSUPPORTED, "the agent trail is on: episodes replayed from the reference's RNG stream do not feed it");
    const int64_t N = h->N;
    for (int64_t e = 0; e < N; ++e) {
        GU_REQUIRE(first_state[e] >= 0 && first_state[e] < h->S, GU_ERR_INVALID, "first_state[%lld]=%d outside the grid", (long long)e, first_state[e]);
        GU_REQUIRE(offsets[e] >= 0 && offsets[e] <= K, GU_ERR_INVALID, "offsets[%lld] outside the uniforms", (long long)e);
    }
    const size_t off_cdf = (size_t)(K > 0 ? K : 1) * 8, off_off = off_cdf + (size_t)h->S * 32, off_first = off_off + (size_t)N * 8;
    rc = gu_ensure_scratch(h, off_first + (size_t)N * 4);
    if (rc != GU_OK) return rc;
    char *base = (char *)h->d_scratch;
    if (K > 0) GU_HIP(hipMemcpyAsync(base, u, (size_t)K * 8, hipMemcpyHostToDevice, h->stream));
    GU_HIP(hipMemcpyAsync(base + off_cdf, cdf, (size_t)h->S * 32, hipMemcpyHostToDevice, h->stream));
    GU_HIP(hipMemcpyAsync(base + off_off, offsets, (size_t)N * 8, hipMemcpyHostToDevice, h->stream));
    GU_HIP(hipMemcpyAsync(base + off_first, first_state, (size_t)N * 4, hipMemcpyHostToDevice, h->stream));
    McWalkArgs a;
    mc_walk_args(h, &a, (const double *)base, K, (const double *)(base + off_cdf), cap);
    a.lds = (int64_t)h->S * 34 <= 60 * 1024 ? 1 : 0;
    const int64_t rows = h->traj_T * N;
    hipLaunchKernelGGL(gu_mc_walk_episodes_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64), a.lds ? (size_t)h->S * 34 : 0, h->stream, a, (const int64_t *)(base + off_off),
                       (const int32_t *)(base + off_first), N, T, h->d_traj, h->d_traj + rows, h->d_traj + 2 * rows);
    GU_HIP(hipGetLastError());
    GU_HIP(hipStreamSynchronize(h->stream));  // (the host buffers of the copies above are the caller's)
    h->traj_kind = 1;
    return GU_OK;
}
