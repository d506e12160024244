// gu_trail.hip -- the agent trail of the reference's viewer, per env, on the device (rest of it).
//
// core/envs/griduniverse_env.py keeps `last_n_states`: every step appends the cell the agent is on afterwards (env:182), the
// list is cut to its newest 500 entries (env:92, 183-184) and emptied by reset (env:190; the start cell is NOT entered).
// core/envs/rendering.py:287-311 draws it every frame: newest entry first, a quad over the entry's tile with alpha
// a_i = 0.3 * 0.96^(i + 1) for the i-th newest, entries on the agent's CURRENT cell skipped (but counted: the alpha decays
// past them).  Off by default and free when off: the ring is kept by small kernels of its own that the step / reset / rollout
// launchers enqueue behind (reset: in front of) their kernel when gu_trail_enable was called -- the hot kernels are untouched.
//   step    : the lazy auto-reset of a GU_F_AUTO_RESET step empties the ring first (the env was done after the step before:
//             the harness's `if done: env.reset()`), then the new cell is appended
//   reset   : the envs the reset kernel is about to reset (mask / done-only / all) are emptied
//   rollout : the same per row of the trajectory the launch wrote (int32 or packed rows; a launch that keeps no rows cannot
//             feed the trail and is refused while it is enabled)
#include "gu_internal.hpp"

struct TrailArgs {
    int32_t *ring;       // [N][cap], slot (head - 1 - i) mod cap = the i-th newest cell
    int32_t *len, *head; // [N]
    uint8_t *was_done;   // [N] the env's done flag behind the last append
    int64_t N;
    int32_t cap;
};

__device__ __forceinline__ void gu_trail_append(const TrailArgs &t, int64_t e, int32_t &len, int32_t &head, int32_t cell)
{
    t.ring[e * t.cap + head] = cell;
    head = head + 1 == t.cap ? 0 : head + 1;
    len = len < t.cap ? len + 1 : len;
}

__global__ void __launch_bounds__(256) gu_trail_step_kernel(const TrailArgs t, const int32_t *__restrict__ pos, const int32_t *__restrict__ done, int32_t auto_reset)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= t.N) return;
    int32_t len = t.len[e], head = t.head[e];
    if (auto_reset && t.was_done[e]) len = 0;
    gu_trail_append(t, e, len, head, pos[e]);
    t.len[e] = len;
    t.head[e] = head;
    t.was_done[e] = done[e] != 0;
}

__global__ void __launch_bounds__(256) gu_trail_reset_kernel(const TrailArgs t, const uint8_t *__restrict__ mask, const int32_t *__restrict__ done, int32_t only_done)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= t.N) return;
    const bool resets = only_done ? done[e] != 0 : (mask ? mask[e] != 0 : true);
    if (resets) {
        t.len[e] = 0;
        t.was_done[e] = 0;
    }
}

// rows 0 .. T-1 of the trajectory a rollout just wrote: obs / done as int32 rows, or packed (obs | reward << 16 | done << 24)
__global__ void __launch_bounds__(256) gu_trail_rows_kernel(const TrailArgs t, const int32_t *__restrict__ obs, const int32_t *__restrict__ done,
                                                            const uint32_t *__restrict__ packed, int64_t T, int32_t auto_reset)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= t.N) return;
    int32_t len = t.len[e], head = t.head[e];
    bool was = t.was_done[e] != 0;
    for (int64_t i = 0; i < T; ++i) {
        int32_t cell, dn;
        if (packed) {
            const uint32_t w = packed[i * t.N + e];
            cell = (int32_t)(w & 0xFFFFu);
            dn = (int32_t)((w >> 24) & 1u);
        } else {
            cell = obs[i * t.N + e];
            dn = done[i * t.N + e];
        }
        if (auto_reset && was) len = 0;
        gu_trail_append(t, e, len, head, cell);
        was = dn != 0;
    }
    t.len[e] = len;
    t.head[e] = head;
    t.was_done[e] = was;
}

static TrailArgs trail_args(gu_engine *h) { return TrailArgs{h->d_trail, h->d_trail_len, h->d_trail_head, h->d_trail_done, h->N, h->trail_cap}; }
static unsigned trail_blocks(const gu_engine *h) { return (unsigned)((h->N + 255) / 256); }

int gu_trail_after_step(gu_engine *h, uint32_t flags)
{
    if (!h->trail_cap) return GU_OK;
    hipLaunchKernelGGL(gu_trail_step_kernel, dim3(trail_blocks(h)), dim3(256), 0, h->stream, trail_args(h), h->pos(), h->done(),
                       (flags & GU_F_AUTO_RESET) ? 1 : 0);
    GU_HIP(hipGetLastError());
    return GU_OK;
}

int gu_trail_before_reset(gu_engine *h, const uint8_t *d_mask, bool only_done)
{
    if (!h->trail_cap) return GU_OK;
    hipLaunchKernelGGL(gu_trail_reset_kernel, dim3(trail_blocks(h)), dim3(256), 0, h->stream, trail_args(h), d_mask, h->done(), only_done ? 1 : 0);
    GU_HIP(hipGetLastError());
    return GU_OK;
}

// gu_set_state put envs somewhere else and / or installed done flags: an env that was moved by hand has no trail to continue
// (its ring is emptied -- the reference has no counterpart: `current_state` assigned from outside leaves `last_n_states` stale), and
// the flag that decides the next lazy reset is the installed one, not the one behind the last append.
__global__ void __launch_bounds__(256) gu_trail_set_state_kernel(const TrailArgs t, const int32_t *__restrict__ done, int32_t moved, int32_t done_given)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= t.N) return;
    if (moved) t.len[e] = 0;
    if (done_given) t.was_done[e] = done[e] != 0;
}

int gu_trail_after_set_state(gu_engine *h, bool moved, bool done_given)
{
    if (!h->trail_cap || !(moved || done_given)) return GU_OK;
    hipLaunchKernelGGL(gu_trail_set_state_kernel, dim3(trail_blocks(h)), dim3(256), 0, h->stream, trail_args(h), h->done(), moved ? 1 : 0, done_given ? 1 : 0);
    GU_HIP(hipGetLastError());
    return GU_OK;
}

int gu_trail_after_rollout(gu_engine *h, int64_t T, int traj, bool auto_reset)
{
    if (!h->trail_cap) return GU_OK;
    const int64_t rows = h->traj_T * h->N;
    hipLaunchKernelGGL(gu_trail_rows_kernel, dim3(trail_blocks(h)), dim3(256), 0, h->stream, trail_args(h), traj == 1 ? h->d_traj : nullptr,
                       traj == 1 ? h->d_traj + 2 * rows : nullptr, traj == 2 ? (const uint32_t *)h->d_traj : nullptr, T, auto_reset ? 1 : 0);
    GU_HIP(hipGetLastError());
    return GU_OK;
}

void gu_trail_free(gu_engine *h)
{
    if (h->d_trail) (void)hipFree(h->d_trail);
    if (h->d_trail_len) (void)hipFree(h->d_trail_len);
    if (h->d_trail_done) (void)hipFree(h->d_trail_done);
    if (h->d_trail_alpha) (void)hipFree(h->d_trail_alpha);
    h->d_trail = h->d_trail_len = h->d_trail_head = nullptr;
    h->d_trail_done = nullptr;
    h->d_trail_alpha = nullptr;
    h->trail_cap = 0;
}

extern "C" {

int gu_trail_enable(gu_handle h, int32_t capacity)
{
    int rc = gu_use_device(h);
    if (rc != GU_OK) return rc;
    GU_REQUIRE(capacity >= 0 && capacity <= 500, GU_ERR_INVALID, "trail capacity %d outside 0 .. 500 (env:92 keeps 500 states)", capacity);
    GU_HIP(hipStreamSynchronize(h->stream));
    gu_trail_free(h);
    if (h->graph_exec) {  // a captured step graph does not carry the trail kernels (or carries them although the trail is off now)
        (void)hipGraphExecDestroy(h->graph_exec);
        h->graph_exec = nullptr;
    }
    if (!capacity) return GU_OK;
    const size_t n = (size_t)h->N;
    GU_HIP(hipMalloc(&h->d_trail, n * (size_t)capacity * sizeof(int32_t)));
    GU_HIP(hipMalloc(&h->d_trail_len, 2 * n * sizeof(int32_t)));
    h->d_trail_head = h->d_trail_len + n;
    GU_HIP(hipMalloc(&h->d_trail_done, n));
    GU_HIP(hipMalloc(&h->d_trail_alpha, (size_t)capacity * sizeof(uint32_t)));
    GU_HIP(hipMemset(h->d_trail_len, 0, 2 * n * sizeof(int32_t)));
    GU_HIP(hipMemset(h->d_trail_done, 0, n));
    // alpha of the i-th newest entry as the viewer forms it -- a = 0.3, then `a *= 0.96` before every entry (rendering.py:289-295),
    // float64 -- in 16 fractional bits, rounded to nearest: the blend below is integer arithmetic that can be restated exactly
    std::vector<uint32_t> alpha((size_t)capacity);
    double a = 0.3;
    for (int32_t i = 0; i < capacity; ++i) {
        a *= 0.96;
        alpha[(size_t)i] = (uint32_t)(a * 65536.0 + 0.5);
    }
    GU_HIP(hipMemcpy(h->d_trail_alpha, alpha.data(), alpha.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    h->trail_cap = capacity;
    return GU_OK;
}

int gu_trail_read(gu_handle h, int64_t env0, int64_t n_envs, int32_t *cells, int32_t *length)
{
    int rc = gu_use_device(h);
    if (rc != GU_OK) return rc;
    GU_REQUIRE(h->trail_cap > 0, GU_ERR_STATE, "the trail is off: call gu_trail_enable first");
    GU_REQUIRE(cells && length && env0 >= 0 && n_envs > 0 && env0 + n_envs <= h->N, GU_ERR_INVALID, "env range [%lld,%lld) outside the batch, or a NULL pointer",
               (long long)env0, (long long)(env0 + n_envs));
    GU_HIP(hipStreamSynchronize(h->stream));
    const size_t cap = (size_t)h->trail_cap, n = (size_t)n_envs;
    std::vector<int32_t> ring(n * cap), len(n), head(n);
    GU_HIP(hipMemcpy(ring.data(), h->d_trail + (size_t)env0 * cap, ring.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
    GU_HIP(hipMemcpy(len.data(), h->d_trail_len + env0, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    GU_HIP(hipMemcpy(head.data(), h->d_trail_head + env0, n * sizeof(int32_t), hipMemcpyDeviceToHost));
    for (size_t k = 0; k < n; ++k) {  // oldest first, like the reference's list
        length[k] = len[k];
        for (int32_t i = 0; i < len[k]; ++i) {
            const size_t slot = (size_t)((head[k] - len[k] + i + 2 * (int32_t)cap) % (int32_t)cap);
            cells[k * cap + (size_t)i] = ring[k * cap + slot];
        }
        for (size_t i = (size_t)len[k]; i < cap; ++i) cells[k * cap + i] = -1;
    }
    return GU_OK;
}

}  // extern "C"
