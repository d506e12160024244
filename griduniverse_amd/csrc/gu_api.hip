// gu_api.hip -- C ABI of libgu.so (see include/gu.h for the contract and the reference
// interfaces each entry point replaces).
#include "gu_internal.hpp"
#include "gu_rng.hpp"

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <chrono>
#include <mutex>
#include <vector>

// ---------------------------------------------------------------------------------- errors
static thread_local std::string g_last_error;

void gu_set_error(const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

int gu_fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

int gu_use_device(gu_engine *h)
{
    GU_REQUIRE(h != nullptr, GU_ERR_INVALID, "null handle");
    GU_HIP(hipSetDevice(h->device));
    return GU_OK;
}

int gu_ensure_scratch(gu_engine *h, size_t bytes)
{
    if (bytes <= h->scratch_bytes) return GU_OK;
    if (h->d_scratch) {
        GU_HIP(hipStreamSynchronize(h->stream));
        GU_HIP(hipFree(h->d_scratch));
        h->d_scratch = nullptr;
        h->scratch_bytes = 0;
    }
    GU_HIP(hipMalloc(&h->d_scratch, bytes));
    h->scratch_bytes = bytes;
    return GU_OK;
}

#define GU_ENTER(h)                                  \
    do {                                             \
        int _rc = gu_use_device(h);                  \
        if (_rc != GU_OK) return _rc;                \
    } while (0)

#define GU_NEED_GRID(h) GU_REQUIRE((h)->has_grid, GU_ERR_STATE, "no grid set: call gu_set_grid first")

static void gu_placement_release(gu_engine *h);  // (the registry of chosen trajectory buffers, below)

extern "C" {

int gu_version(void) { return GU_ABI_VERSION; }

int gu_last_error(char *buf, size_t len)
{
    if (buf && len) {
        size_t n = g_last_error.size() < len - 1 ? g_last_error.size() : len - 1;
        memcpy(buf, g_last_error.data(), n);
        buf[n] = 0;
    }
    return (int)g_last_error.size();
}

int gu_device_count(int *count)
{
    GU_REQUIRE(count != nullptr, GU_ERR_INVALID, "count is NULL");
    *count = 0;
    GU_HIP(hipGetDeviceCount(count));
    return GU_OK;
}

// ---------------------------------------------------------------------------------- lifetime
int gu_create(int device_id, int64_t num_envs, int64_t env_id0, gu_handle *out)
{
    GU_REQUIRE(out != nullptr, GU_ERR_INVALID, "out handle is NULL");
    *out = nullptr;
    GU_REQUIRE(num_envs > 0 && num_envs <= (1 << 25), GU_ERR_INVALID, "num_envs %lld out of range (1 .. 2^25 per device)", (long long)num_envs);
    GU_REQUIRE(env_id0 >= 0 && env_id0 + num_envs <= 0xFFFFFFFFLL, GU_ERR_INVALID, "global env ids must fit 32 bits");
    int n_dev = 0;
    GU_HIP(hipGetDeviceCount(&n_dev));
    GU_REQUIRE(device_id >= 0 && device_id < n_dev, GU_ERR_HIP, "device %d not present (%d visible)", device_id, n_dev);
    gu_engine *h = new (std::nothrow) gu_engine();
    GU_REQUIRE(h != nullptr, GU_ERR_NOMEM, "host allocation failed");
    h->device = device_id;
    h->N = num_envs;
    h->env_id0 = env_id0;
    h->seed = 0;
    h->seed_prefix = gu_rng_seed_prefix(0);
    for (auto &o : h->opt) o = GU_OPT_UNSET;
    for (auto &o : h->opt_x) o = 0;
    int rc = GU_OK;
    auto body = [&]() -> int {
        GU_HIP(hipSetDevice(device_id));
        // launch shapes are sized from what the device reports, not from MI355X constants (a CPX partition, a CU mask or a
        // smaller part has fewer CUs; the absolute write-rate rule of the placement search applies to gfx950 only)
        hipDeviceProp_t prop;
        memset(&prop, 0, sizeof prop);
        GU_HIP(hipGetDeviceProperties(&prop, device_id));
        if (prop.multiProcessorCount > 0) h->n_cu = prop.multiProcessorCount;
        if (prop.maxSharedMemoryPerMultiProcessor >= 64 * 1024) h->lds_per_cu = (int64_t)prop.maxSharedMemoryPerMultiProcessor;
        h->gfx950 = strncmp(prop.gcnArchName, "gfx950", 6) == 0;
        GU_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
        GU_HIP(hipEventCreate(&h->ev_begin));
        GU_HIP(hipEventCreate(&h->ev_end));
        GU_HIP(hipEventCreateWithFlags(&h->ev_sync, hipEventDisableTiming));
        const size_t n = (size_t)num_envs;
        GU_HIP(hipMalloc(&h->d_out3, 3 * n * sizeof(int32_t)));
        GU_HIP(hipMalloc(&h->d_episode, n * sizeof(uint32_t)));
        GU_HIP(hipMalloc(&h->d_tcount, n * sizeof(uint32_t)));
        GU_HIP(hipMalloc(&h->d_ret, n * sizeof(int32_t)));
        GU_HIP(hipMalloc(&h->d_episodes_fin, n * sizeof(int32_t)));
        GU_HIP(hipMalloc(&h->d_done_bits, ((n + 63) / 64) * sizeof(uint64_t)));
        GU_HIP(hipMemsetAsync(h->d_done_bits, 0, ((n + 63) / 64) * sizeof(uint64_t), h->stream));
        h->done_bits_valid = true;
        GU_HIP(hipHostMalloc(&h->h_pin, 4 * n * sizeof(int32_t), hipHostMallocDefault));
        GU_HIP(hipHostMalloc(&h->h_seq, 64, hipHostMallocDefault));
        GU_HIP(hipHostMalloc(&h->h_ctl, GU_CTL_WORDS * sizeof(unsigned long long), hipHostMallocDefault));
        GU_HIP(hipHostMalloc(&h->h_up, GU_UP_BYTES, hipHostMallocDefault));
        memset(h->h_seq, 0, 64);
        GU_HIP(hipMalloc(&h->d_blocks_done, sizeof(uint32_t)));
        GU_HIP(hipMemsetAsync(h->d_blocks_done, 0, sizeof(uint32_t), h->stream));
        GU_HIP(hipMemsetAsync(h->d_out3, 0, 3 * n * sizeof(int32_t), h->stream));
        GU_HIP(hipMemsetAsync(h->d_episode, 0, n * sizeof(uint32_t), h->stream));
        GU_HIP(hipMemsetAsync(h->d_tcount, 0, n * sizeof(uint32_t), h->stream));
        GU_HIP(hipStreamSynchronize(h->stream));
        return GU_OK;
    };
    rc = body();
    if (rc != GU_OK) {
        std::string keep = g_last_error;
        gu_destroy(h);
        g_last_error = keep;
        return rc;
    }
    *out = h;
    return GU_OK;
}

int gu_destroy(gu_handle h)
{
    if (!h) return GU_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    gu_comm_free(h);
    gu_vi_free(h);
    gu_trail_free(h);
    gu_placement_release(h);
    if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
    void *bufs[] = {h->d_kind, h->d_rows[0], h->d_rows[1], h->d_rows2[0], h->d_rows2[1], h->d_mrows[0], h->d_mrows[1], h->d_mrows1[0], h->d_mrows1[1], h->d_prow, h->d_cell, h->d_cell_raw, h->d_nib, h->d_starts, h->d_nstarts, h->d_out3, h->d_episode, h->d_tcount, h->d_actions, h->d_actions_packed,
                    h->d_traj, h->d_ret, h->d_episodes_fin, h->d_done_bits, h->d_scratch, h->d_greedy, h->d_pace_ring, h->d_pace_slots, h->d_out3_alt, h->d_episode_alt, h->d_done_bits_alt};
    for (void *p : bufs)
        if (p) (void)hipFree(p);
    if (h->h_pin) (void)hipHostFree(h->h_pin);
    if (h->h_seq) (void)hipHostFree(h->h_seq);
    if (h->h_ctl) (void)hipHostFree(h->h_ctl);
    if (h->h_tables) (void)hipHostFree(h->h_tables);
    if (h->h_up) (void)hipHostFree(h->h_up);
    if (h->d_blocks_done) (void)hipFree(h->d_blocks_done);
    for (hipEvent_t ev : h->ev_marks) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : h->ev_cal)
        if (ev) (void)hipEventDestroy(ev);
    if (h->ev_begin) (void)hipEventDestroy(h->ev_begin);
    if (h->ev_end) (void)hipEventDestroy(h->ev_end);
    if (h->ev_sync) (void)hipEventDestroy(h->ev_sync);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
    return GU_OK;
}

// ---------------------------------------------------------------------------------- grid
static inline bool plane_bit(const uint32_t *rows, int32_t wpr, int32_t x, int32_t y)
{
    return rows && ((rows[(size_t)y * wpr + (x >> 5)] >> (x & 31)) & 1u);
}

// Compile one grid's row bit-planes into the two byte planes (flags | reward), absorbing and raw variant,
// appended to `cell` / `raw`.
static void compile_planes(int32_t W, int32_t H, int32_t wpr, const uint32_t *wall_rows, const uint32_t *goal_rows,
                           const uint32_t *lava_rows, const uint32_t *rplus_rows, const uint32_t *rminus_rows,
                           uint8_t *cell, uint8_t *raw, uint8_t *kind, int32_t cell_bytes)
{
    auto wall = [&](int32_t x, int32_t y) { return plane_bit(wall_rows, wpr, x, y); };
    for (int32_t y = 0; y < H; ++y) {
        for (int32_t x = 0; x < W; ++x) {
            const size_t s = (size_t)y * W + x;
            const bool lava = plane_bit(lava_rows, wpr, x, y);
            const bool goal = plane_bit(goal_rows, wpr, x, y);
            const bool term = lava || goal;                                          // env:163-168
            const bool rminus = rminus_rows ? plane_bit(rminus_rows, wpr, x, y) : lava;   // env:86-88
            const bool rplus = rplus_rows ? plane_bit(rplus_rows, wpr, x, y) : (goal && !lava);
            uint8_t open = 0;
            if (y > 0 && !wall(x, y - 1)) open |= 1u;        // UP    env:51, env:149
            if (x < W - 1 && !wall(x + 1, y)) open |= 2u;    // RIGHT env:52
            if (y < H - 1 && !wall(x, y + 1)) open |= 4u;    // DOWN  env:53
            if (x > 0 && !wall(x - 1, y)) open |= 8u;        // LEFT  env:54
            const uint8_t t = (term ? GU_CELL_TERM : 0) | (rplus ? GU_CELL_RPLUS : 0) | (rminus ? GU_CELL_RMINUS : 0) |
                              (wall(x, y) ? GU_CELL_WALL : 0);
            raw[s] = open | t;
            cell[s] = (term ? 0 : open) | t;                                         // env:145-146 absorbing
            const int8_t r = rminus ? -10 : (rplus ? 10 : -1);                       // env:80-90
            raw[(size_t)cell_bytes + s] = cell[(size_t)cell_bytes + s] = (uint8_t)r;
            kind[s] = goal ? 3 : lava ? 2 : wall(x, y) ? 1 : 0;                      // core/envs/rendering.py:119-133
        }
    }
}

}  // extern "C" (helpers below have C++ signatures)

// Upload compiled planes of `n_grids` grids ([g][flags|reward]) and their start tables; resets env state.
int gu_install_grids(gu_engine *h, int32_t n_grids, int32_t W, int32_t H, const std::vector<uint8_t> &cell,
                     const std::vector<uint8_t> &raw, const std::vector<uint8_t> &kind, const std::vector<int32_t> &starts,
                     const std::vector<int32_t> &n_starts, int32_t max_starts)
{
    const int32_t S = W * H;
    const int32_t cell_bytes = (S + 15) & ~15;
    GU_HIP(hipStreamSynchronize(h->stream));
    h->entry_table_ok = false;  // (other cells, other flags)
    h->nib_valid = false;
    if (h->d_nib) GU_HIP(hipFree(h->d_nib));
    h->d_nib = nullptr;
    for (void *p : {(void *)h->d_cell, (void *)h->d_cell_raw, (void *)h->d_kind, (void *)h->d_starts, (void *)h->d_nstarts, (void *)h->d_greedy})
        if (p) GU_HIP(hipFree(p));
    h->d_cell = h->d_cell_raw = h->d_kind = h->d_greedy = nullptr;
    h->d_starts = h->d_nstarts = nullptr;
    for (int k = 0; k < 2; ++k) {  // the transition-row tables belong to the old grid
        if (h->d_rows[k]) GU_HIP(hipFree(h->d_rows[k]));
        h->d_rows[k] = nullptr;
        h->rows_shift[k] = -1;
        if (h->d_rows2[k]) GU_HIP(hipFree(h->d_rows2[k]));
        h->d_rows2[k] = nullptr;
        h->rows2_built[k] = false;
        if (h->d_mrows[k]) GU_HIP(hipFree(h->d_mrows[k]));
        if (h->d_mrows1[k]) GU_HIP(hipFree(h->d_mrows1[k]));
        h->d_mrows[k] = h->d_mrows1[k] = nullptr;
        h->mrows_K[k] = 0;
        h->mrows_shift[k] = -1;
    }
    if (h->d_prow) GU_HIP(hipFree(h->d_prow));
    h->d_prow = nullptr;
    h->has_grid = false;
    for (auto &p : h->pace) p.active = false;  // (another grid: another chain length)
    gu_vi_free(h);
    const size_t plane_bytes = 2 * (size_t)cell_bytes * n_grids;
    GU_HIP(hipMalloc(&h->d_cell, plane_bytes));
    GU_HIP(hipMalloc(&h->d_cell_raw, plane_bytes));
    GU_HIP(hipMalloc(&h->d_greedy, cell_bytes));
    GU_HIP(hipMalloc(&h->d_starts, starts.size() * sizeof(int32_t)));
    GU_HIP(hipMalloc(&h->d_nstarts, (size_t)n_grids * sizeof(int32_t)));
    if (!cell.empty()) {  // empty = the caller fills the planes on the device (gu_generate_mazes)
        GU_HIP(hipMemcpy(h->d_cell, cell.data(), plane_bytes, hipMemcpyHostToDevice));
        GU_HIP(hipMemcpy(h->d_cell_raw, raw.data(), plane_bytes, hipMemcpyHostToDevice));
    }
    if (!kind.empty()) {
        GU_HIP(hipMalloc(&h->d_kind, kind.size()));
        GU_HIP(hipMemcpy(h->d_kind, kind.data(), kind.size(), hipMemcpyHostToDevice));
    }
    GU_HIP(hipMemset(h->d_greedy, 0, cell_bytes));
    GU_HIP(hipMemcpy(h->d_starts, starts.data(), starts.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    GU_HIP(hipMemcpy(h->d_nstarts, n_starts.data(), (size_t)n_grids * sizeof(int32_t), hipMemcpyHostToDevice));
    h->W = W;
    h->H = H;
    h->S = S;
    {   // action -> state delta LUT, four int16 lanes of one 64-bit scalar: UP -W, RIGHT +1, DOWN +W, LEFT -1
        const uint64_t w16 = (uint64_t)(uint16_t)(int16_t)(W <= 32767 ? W : 0);
        const uint64_t mw16 = (uint64_t)(uint16_t)(int16_t)(W <= 32767 ? -W : 0);
        h->delta_lut = mw16 | (1ull << 16) | (w16 << 32) | (0xFFFFull << 48);
    }
    h->cell_bytes = cell_bytes;
    h->n_grids = n_grids;
    h->group = h->N / n_grids;
    h->n_starts = n_starts[0];
    h->start0 = starts[0];
    h->max_starts = max_starts;
    h->all_single_start = true;
    for (int32_t g = 0; g < n_grids; ++g) h->all_single_start = h->all_single_start && n_starts[(size_t)g] == 1;
    h->has_grid = true;
    h->greedy_valid = false;
    if (h->graph_exec) {
        (void)hipGraphExecDestroy(h->graph_exec);
        h->graph_exec = nullptr;
    }
    // every env sits on its grid's first start cell until the caller resets (pos must always be a valid cell)
    std::vector<int32_t> init((size_t)h->N);
    for (int64_t e = 0; e < h->N; ++e) init[(size_t)e] = starts[(size_t)(e / h->group) * max_starts];
    GU_HIP(hipMemcpy(h->pos(), init.data(), (size_t)h->N * sizeof(int32_t), hipMemcpyHostToDevice));
    GU_HIP(hipMemset(h->reward(), 0, 2 * (size_t)h->N * sizeof(int32_t)));
    GU_HIP(hipMemset(h->d_done_bits, 0, (((size_t)h->N + 63) / 64) * sizeof(uint64_t)));
    h->done_bits_valid = true;
    return GU_OK;
}

extern "C" {

int gu_set_grids(gu_handle h, int32_t n_grids, int32_t W, int32_t H, int32_t words_per_row, const uint32_t *wall_rows,
                 const uint32_t *goal_rows, const uint32_t *lava_rows, const uint32_t *rplus_rows,
                 const uint32_t *rminus_rows, const int32_t *starts, const int32_t *n_starts, int32_t max_starts)
{
    GU_ENTER(h);
    GU_REQUIRE(n_grids > 0 && h->N % n_grids == 0, GU_ERR_INVALID, "n_grids=%d must divide num_envs=%lld", n_grids, (long long)h->N);
    GU_REQUIRE(W > 0 && H > 0 && (int64_t)W * H <= (1 << 30), GU_ERR_INVALID, "bad grid shape %d x %d", W, H);
    // gu_move multiplies the OPEN bit by delta = +-W with v_mad_i32_i24, which sign-extends delta from 24 bits
    GU_REQUIRE(W <= 8388607, GU_ERR_UNSUPPORTED, "grids wider than 8 388 607 columns are not supported (W=%d)", W);
    GU_REQUIRE(words_per_row == (W + 31) / 32, GU_ERR_INVALID, "words_per_row must be ceil(W/32)");
    GU_REQUIRE(wall_rows && goal_rows && lava_rows, GU_ERR_INVALID, "wall/goal/lava planes are required");
    GU_REQUIRE((rplus_rows == nullptr) == (rminus_rows == nullptr), GU_ERR_INVALID, "give both reward planes or neither");
    GU_REQUIRE(starts && n_starts && max_starts > 0, GU_ERR_INVALID, "start tables are required");
    const int32_t S = W * H;
    const int32_t cell_bytes = (S + 15) & ~15;
    GU_REQUIRE(2 * (int64_t)cell_bytes * n_grids < (1ll << 31), GU_ERR_UNSUPPORTED, "%d grids of %d cells exceed 2 GiB of records", n_grids, S);
    for (int32_t g = 0; g < n_grids; ++g) {
        GU_REQUIRE(n_starts[g] > 0 && n_starts[g] <= max_starts, GU_ERR_INVALID, "grid %d: n_starts=%d outside 1..%d", g, n_starts[g], max_starts);
        for (int32_t i = 0; i < n_starts[g]; ++i) {
            const int32_t st = starts[(size_t)g * max_starts + i];
            GU_REQUIRE(st >= 0 && st < S, GU_ERR_INVALID, "grid %d: starting state %d outside the grid", g, st);
        }
    }
    std::vector<uint8_t> cell(2 * (size_t)cell_bytes * n_grids, 0), raw(2 * (size_t)cell_bytes * n_grids, 0), kind((size_t)cell_bytes * n_grids, 0);
    const size_t plane_words = (size_t)H * words_per_row;
    for (int32_t g = 0; g < n_grids; ++g)
        compile_planes(W, H, words_per_row, wall_rows + g * plane_words, goal_rows + g * plane_words, lava_rows + g * plane_words,
                       rplus_rows ? rplus_rows + g * plane_words : nullptr, rminus_rows ? rminus_rows + g * plane_words : nullptr,
                       cell.data() + 2 * (size_t)cell_bytes * g, raw.data() + 2 * (size_t)cell_bytes * g, kind.data() + (size_t)cell_bytes * g,
                       cell_bytes);
    std::vector<int32_t> st(starts, starts + (size_t)n_grids * max_starts), ns(n_starts, n_starts + n_grids);
    for (int32_t g = 0; g < n_grids; ++g)  // pad unused slots with a valid cell
        for (int32_t i = ns[(size_t)g]; i < max_starts; ++i) st[(size_t)g * max_starts + i] = st[(size_t)g * max_starts];
    return gu_install_grids(h, n_grids, W, H, cell, raw, kind, st, ns, max_starts);
}

int gu_set_grid(gu_handle h, int32_t W, int32_t H, int32_t words_per_row, const uint32_t *wall_rows,
                const uint32_t *goal_rows, const uint32_t *lava_rows, const uint32_t *rplus_rows,
                const uint32_t *rminus_rows, const int32_t *starts, int32_t n_starts)
{
    GU_REQUIRE(starts && n_starts > 0, GU_ERR_INVALID, "at least one starting state is required");
    return gu_set_grids(h, 1, W, H, words_per_row, wall_rows, goal_rows, lava_rows, rplus_rows, rminus_rows, starts, &n_starts, n_starts);
}

// Read back grid `grid_index` as the engine holds it: flags[S] and reward[S] (absorbing map), its start table.
int gu_get_cells(gu_handle h, int32_t grid_index, uint8_t *flags, int8_t *reward, int32_t *starts, int32_t *n_starts)
{
    GU_ENTER(h);
    GU_NEED_GRID(h);
    GU_REQUIRE(grid_index >= 0 && grid_index < h->n_grids, GU_ERR_INVALID, "grid index %d outside [0,%d)", grid_index, h->n_grids);
    GU_HIP(hipStreamSynchronize(h->stream));
    const uint8_t *base = h->d_cell + 2 * (size_t)h->cell_bytes * grid_index;
    if (flags) GU_HIP(hipMemcpy(flags, base, (size_t)h->S, hipMemcpyDeviceToHost));
    if (reward) GU_HIP(hipMemcpy(reward, base + h->cell_bytes, (size_t)h->S, hipMemcpyDeviceToHost));
    int32_t ns = 0;
    GU_HIP(hipMemcpy(&ns, h->d_nstarts + grid_index, sizeof ns, hipMemcpyDeviceToHost));
    if (n_starts) *n_starts = ns;
    if (starts) GU_HIP(hipMemcpy(starts, h->d_starts + (size_t)grid_index * h->max_starts, (size_t)ns * sizeof(int32_t), hipMemcpyDeviceToHost));
    return GU_OK;
}

// ---------------------------------------------------------------------------------- RNG
int gu_seed(gu_handle h, uint64_t seed)
{
    GU_ENTER(h);
    h->seed = seed;
    h->seed_prefix = gu_rng_seed_prefix(seed);
    h->steps_taken = 0;
    h->off_lo = h->off_hi = 0;
    h->off_exact = true;
    if (h->graph_exec) {  // captured step launches carry the old seed in their arguments
        (void)hipGraphExecDestroy(h->graph_exec);
        h->graph_exec = nullptr;
    }
    GU_HIP(hipMemsetAsync(h->d_episode, 0, (size_t)h->N * sizeof(uint32_t), h->stream));
    GU_HIP(hipMemsetAsync(h->d_tcount, 0, (size_t)h->N * sizeof(uint32_t), h->stream));
    GU_HIP(hipStreamSynchronize(h->stream));
    return GU_OK;
}

// ---------------------------------------------------------------------------------- reset
int gu_reset(gu_handle h, const uint8_t *mask, const int32_t *start_choice, int32_t *obs_out)
{
    GU_ENTER(h);
    GU_NEED_GRID(h);
    const size_t n = (size_t)h->N;
    uint8_t *d_mask = nullptr;
    int32_t *d_choice = nullptr;
    if (mask || start_choice) {
        int rc = gu_ensure_scratch(h, n * 5 + 16);
        if (rc != GU_OK) return rc;
        if (start_choice) {
            for (size_t i = 0; i < n; ++i)
                if (!mask || mask[i])
                    GU_REQUIRE(start_choice[i] >= 0 && start_choice[i] < (h->n_grids == 1 ? h->n_starts : h->max_starts), GU_ERR_INVALID,
                               "start_choice[%zu]=%d outside the start table", i, start_choice[i]);
            d_choice = (int32_t *)h->d_scratch;
            GU_HIP(hipMemcpyAsync(d_choice, start_choice, n * 4, hipMemcpyHostToDevice, h->stream));
        }
        if (mask) {
            d_mask = (uint8_t *)h->d_scratch + n * 4;
            GU_HIP(hipMemcpyAsync(d_mask, mask, n, hipMemcpyHostToDevice, h->stream));
        }
    }
    int rc = gu_launch_reset(h, d_mask, d_choice, false);
    if (rc != GU_OK) return rc;
    if (obs_out) GU_HIP(hipMemcpyAsync(obs_out, h->pos(), n * 4, hipMemcpyDeviceToHost, h->stream));
    GU_HIP(hipStreamSynchronize(h->stream));
    return GU_OK;
}

int gu_reset_done(gu_handle h)
{
    GU_ENTER(h);
    GU_NEED_GRID(h);
    return gu_launch_reset(h, nullptr, nullptr, true);
}

// ---------------------------------------------------------------------------------- step
// Page-locked ranges the step kernel may dereference directly (GU_F_PINNED_IO).  Only memory that gu_host_alloc handed out
// is remembered (the library sees it freed, in gu_host_free, which bumps the generation and so empties every thread's
// cache): a call on such a buffer costs a few compares.  Memory the caller pinned by other means (hipHostRegister, a
// framework's pinned tensors) can be unpinned or freed behind the library's back, so it is asked about on EVERY call.
static std::atomic<uint32_t> g_pinned_generation{1};
static std::mutex g_host_alloc_mu;
static std::vector<std::pair<uintptr_t, uintptr_t>> g_host_allocs;  // [lo, hi) of every live gu_host_alloc block

struct PinnedRanges {
    uint32_t generation = 0;
    int n = 0;
    uintptr_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
};
static thread_local PinnedRanges g_pinned;

static int gu_require_pinned(const void *p, size_t bytes, const char *what)
{
    if (!p) return GU_OK;
    const uint32_t gen = g_pinned_generation.load(std::memory_order_acquire);
    if (g_pinned.generation != gen) {
        g_pinned.generation = gen;
        g_pinned.n = 0;
    }
    const uintptr_t lo = (uintptr_t)p, hi = lo + bytes;
    for (int i = 0; i < g_pinned.n; ++i)
        if (lo >= g_pinned.lo[i] && hi <= g_pinned.hi[i]) return GU_OK;
    {   // one of the library's own blocks?  then it stays valid until gu_host_free (which bumps the generation)
        std::lock_guard<std::mutex> lock(g_host_alloc_mu);
        for (const auto &r : g_host_allocs)
            if (lo >= r.first && lo < r.second) {
                if (hi > r.second)
                    return gu_fail(GU_ERR_INVALID, "GU_F_PINNED_IO: %s (%p + %zu bytes) runs past the end of its page-locked allocation", what, p, bytes);
                const int slot = g_pinned.n < 4 ? g_pinned.n++ : 0;
                g_pinned.lo[slot] = r.first;
                g_pinned.hi[slot] = r.second;
                return GU_OK;
            }
    }
    // foreign memory: asked about on every call, never cached
    hipPointerAttribute_t attr;
    memset(&attr, 0, sizeof attr);
    if (hipPointerGetAttributes(&attr, p) != hipSuccess || attr.type != hipMemoryTypeHost) {
        (void)hipGetLastError();  // an ordinary malloc / numpy pointer is "invalid value" to the runtime: not sticky
        return gu_fail(GU_ERR_INVALID, "GU_F_PINNED_IO: %s (%p) is not page-locked host memory (use gu_host_alloc, or drop the flag)", what, p);
    }
    uintptr_t base = lo, size = bytes;
    void *start = nullptr;
    size_t range = 0;
    if (hipPointerGetAttribute(&start, HIP_POINTER_ATTRIBUTE_RANGE_START_ADDR, (hipDeviceptr_t)p) == hipSuccess &&
        hipPointerGetAttribute(&range, HIP_POINTER_ATTRIBUTE_RANGE_SIZE, (hipDeviceptr_t)p) == hipSuccess && start && range) {
        base = (uintptr_t)start;
        size = range;
        if (hi > base + size)
            return gu_fail(GU_ERR_INVALID, "GU_F_PINNED_IO: %s (%p + %zu bytes) runs past the end of its page-locked allocation", what, p, bytes);
    } else {
        (void)hipGetLastError();
    }
    return GU_OK;
}

// The step kernel raised the page-locked error word: name the first offender (the actions are host memory in every
// gu_step path) and re-arm the word.
static int gu_step_action_error(gu_engine *h, const int32_t *actions)
{
    __atomic_store_n(h->h_seq + GU_HOST_ERR_WORD, 0u, __ATOMIC_RELAXED);
    h->off_lo -= 1;  // (the rejected envs did not step: their offsets to the lock-step counter went down by one)
    h->off_exact = false;
    for (int64_t i = 0; i < h->N; ++i)
        if (!GU_ACTION_OK(actions[i]))
            return gu_fail(GU_ERR_INVALID, "action %d of env %lld outside 0..3 (that env did not step; envs with valid actions did)", actions[i], (long long)i);
    return gu_fail(GU_ERR_INVALID, "an action outside 0..3 was seen by the step kernel (the action buffer changed during the call)");
}

// One step whose actions / results live in page-locked host memory, and the wait for it.  The kernel publishes a
// sequence number in page-locked memory after the last block's result stores and the host spins on it -- a PCIe round
// trip instead of the runtime's completion path: 14.0 -> 10.4 us per call at up to 64 envs, 15.5 -> 12.7 at 4096.  Only
// for batches of up to 8192 envs: the per-block system-scope fence serialises the PCIe result stream of larger ones
// (65 536 envs: 57 us against 43 us with the ordinary synchronisation, measured in one process).  Bounded: after ~1 ms
// of spinning, and every 1024 steps anyway, the real stream synchronisation runs (GU_OPT_STEP_SYNC forces it).
// Actions are validated by the kernel itself (an error word next to the completion word), not by a host loop.
static int gu_step_and_wait(gu_engine *h, const int32_t *actions, uint32_t flags, int32_t *obs, int32_t *reward, int32_t *done)
{
    int rc;
    uint32_t *err = h->h_seq + GU_HOST_ERR_WORD;
    bool finished = false;
    if (h->N <= 8192 && h->seq_since_sync < 1024 && !gu_opt(h, GU_OPT_STEP_SYNC)) {
        const uint32_t seq = ++h->seq;
        rc = gu_launch_step(h, actions, flags, obs, reward, done, h->h_seq, seq, err);
        if (rc != GU_OK) return rc;
        ++h->seq_since_sync;
        for (int spin = 0; spin < 2000000 && !finished; ++spin) finished = __atomic_load_n(h->h_seq, __ATOMIC_ACQUIRE) == seq;
    } else {
        rc = gu_launch_step(h, actions, flags, obs, reward, done, nullptr, 0, err);
        if (rc != GU_OK) return rc;
    }
    if (!finished) {
        GU_HIP(hipStreamSynchronize(h->stream));
        h->seq_since_sync = 0;
    }
    if (__atomic_load_n(err, __ATOMIC_ACQUIRE)) return gu_step_action_error(h, actions);
    return GU_OK;
}

int gu_step(gu_handle h, const int32_t *actions, uint32_t flags, int32_t *obs, int32_t *reward, int32_t *done)
{
    GU_ENTER(h);
    GU_NEED_GRID(h);
    GU_REQUIRE(actions != nullptr, GU_ERR_INVALID, "actions is NULL");
    GU_REQUIRE((flags & ~(GU_F_AUTO_RESET | GU_F_PINNED_IO)) == 0, GU_ERR_INVALID, "gu_step accepts only GU_F_AUTO_RESET | GU_F_PINNED_IO");
    const bool direct = flags & GU_F_PINNED_IO;
    flags &= GU_F_AUTO_RESET;
    const size_t n = (size_t)h->N;
    int rc = GU_OK;
    if (direct) {
        // The caller's buffers are page-locked (gu_host_alloc), i.e. mapped into the device's address space: the
        // kernel reads the actions from them and writes the results into them itself over PCIe -- one launch and
        // one synchronisation, no copy commands at all.  (Checked: a pageable pointer here would be a GPU fault.)
        if ((rc = gu_require_pinned(actions, n * 4, "actions")) != GU_OK) return rc;
        if ((rc = gu_require_pinned(obs, n * 4, "obs")) != GU_OK) return rc;
        if ((rc = gu_require_pinned(reward, n * 4, "reward")) != GU_OK) return rc;
        if ((rc = gu_require_pinned(done, n * 4, "done")) != GU_OK) return rc;
        return gu_step_and_wait(h, actions, flags, obs, reward, done);
    }
    // Ordinary (pageable) caller buffers: the engine's own page-locked staging block plays the caller's part of the
    // zero-copy path above -- the kernel reads the actions from it and writes the results into it over PCIe, so the
    // call is two memcpys around one launch + one sync instead of two copy commands around the launch.
    memcpy(h->h_pin, actions, n * 4);
    const bool want = obs || reward || done;
    rc = gu_step_and_wait(h, h->h_pin, flags, want ? h->h_pin + n : nullptr, want ? h->h_pin + 2 * n : nullptr, want ? h->h_pin + 3 * n : nullptr);
    if (rc != GU_OK) return rc;
    if (obs) memcpy(obs, h->h_pin + n, n * 4);
    if (reward) memcpy(reward, h->h_pin + 2 * n, n * 4);
    if (done) memcpy(done, h->h_pin + 3 * n, n * 4);
    return GU_OK;
}

int gu_upload_actions(gu_handle h, const int32_t *actions, int64_t T)
{
    GU_ENTER(h);
    GU_REQUIRE(actions != nullptr && T > 0, GU_ERR_INVALID, "actions NULL or T <= 0");
    const size_t count = (size_t)T * (size_t)h->N;
    h->actions_T = 0;  // until the new stream has been accepted
    if (T > h->actions_cap) {
        GU_HIP(hipStreamSynchronize(h->stream));
        if (h->d_actions) GU_HIP(hipFree(h->d_actions));
        if (h->d_actions_packed) GU_HIP(hipFree(h->d_actions_packed));
        h->d_actions = nullptr;
        h->d_actions_packed = nullptr;
        h->actions_cap = 0;
        GU_HIP(hipMalloc(&h->d_actions, count * sizeof(int32_t)));
        GU_HIP(hipMalloc(&h->d_actions_packed, (size_t)((T + 15) / 16 + GU_STREAM_PAD_WORDS) * (size_t)h->N * sizeof(uint32_t)));  // + look-ahead rows
        h->actions_cap = T;
        if (h->graph_exec) {
            (void)hipGraphExecDestroy(h->graph_exec);
            h->graph_exec = nullptr;
        }
    }
    GU_HIP(hipMemcpyAsync(h->d_actions, actions, count * sizeof(int32_t), hipMemcpyHostToDevice, h->stream));
    // validated on the device, where the stream now is (one pass at HBM speed instead of a host loop over T x N values), and
    // packed in the same pass to the two-bit form the rollout kernels read
    int rc = gu_launch_pack_actions(h, T);
    if (rc != GU_OK) return rc;
    GU_HIP(hipStreamSynchronize(h->stream));
    if (__atomic_load_n(h->h_seq + GU_HOST_ERR_WORD, __ATOMIC_ACQUIRE)) {
        __atomic_store_n(h->h_seq + GU_HOST_ERR_WORD, 0u, __ATOMIC_RELAXED);
        // the rejected stream is not usable (the buffers themselves are kept for the next upload)
        for (size_t i = 0; i < count; ++i)
            if (!GU_ACTION_OK(actions[i])) return gu_fail(GU_ERR_INVALID, "action %d at flat index %zu outside 0..3", actions[i], i);
        return gu_fail(GU_ERR_INVALID, "an action outside 0..3 was uploaded");
    }
    h->actions_T = T;
    return GU_OK;
}

int gu_step_device(gu_handle h, int64_t t, uint32_t flags)
{
    GU_ENTER(h);
    GU_NEED_GRID(h);
    GU_REQUIRE(h->d_actions && t >= 0 && t < h->actions_T, GU_ERR_STATE, "row %lld not in the uploaded action stream", (long long)t);
    GU_REQUIRE((flags & ~GU_F_AUTO_RESET) == 0, GU_ERR_INVALID, "gu_step_device accepts only GU_F_AUTO_RESET");
    return gu_launch_step(h, h->d_actions + t * h->N, flags);
}

int gu_step_graph(gu_handle h, int64_t t0, int64_t T, uint32_t flags)
{
    GU_ENTER(h);
    h->entry_table_ok = false;
    GU_NEED_GRID(h);
    GU_REQUIRE(h->d_actions && t0 >= 0 && T > 0 && t0 + T <= h->actions_T, GU_ERR_STATE, "rows [%lld,%lld) not in the uploaded action stream",
               (long long)t0, (long long)(t0 + T));
    GU_REQUIRE((flags & ~GU_F_AUTO_RESET) == 0, GU_ERR_INVALID, "gu_step_graph accepts only GU_F_AUTO_RESET");
    if (!h->graph_exec || h->graph_t0 != t0 || h->graph_T != T || h->graph_flags != flags) {
        if (h->graph_exec) {
            (void)hipGraphExecDestroy(h->graph_exec);
            h->graph_exec = nullptr;
        }
        const uint64_t saved = h->steps_taken;
        hipGraph_t graph = nullptr;
        GU_HIP(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
        int rc = GU_OK;
        for (int64_t t = t0; t < t0 + T && rc == GU_OK; ++t) rc = gu_launch_step(h, h->d_actions + t * h->N, flags);
        hipError_t e = hipStreamEndCapture(h->stream, &graph);
        h->steps_taken = saved;  // capture launched nothing
        if (rc != GU_OK) return rc;
        GU_HIP(e);
        e = hipGraphInstantiate(&h->graph_exec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        GU_HIP(e);
        h->graph_t0 = t0;
        h->graph_T = T;
        h->graph_flags = flags;
    }
    GU_HIP(hipGraphLaunch(h->graph_exec, h->stream));
    h->steps_taken += (uint64_t)T;
    return GU_OK;
}

int gu_read_outputs(gu_handle h, int32_t *obs, int32_t *reward, int32_t *done)
{
    GU_ENTER(h);
    const size_t n = (size_t)h->N;
    GU_HIP(hipMemcpyAsync(h->h_pin, h->d_out3, 3 * n * 4, hipMemcpyDeviceToHost, h->stream));
    GU_HIP(hipStreamSynchronize(h->stream));
    if (obs) memcpy(obs, h->h_pin, n * 4);
    if (reward) memcpy(reward, h->h_pin + n, n * 4);
    if (done) memcpy(done, h->h_pin + 2 * n, n * 4);
    return GU_OK;
}

// ---------------------------------------------------------------------------------- rollout
// Where a large buffer lands in HBM decides how fast it can be written: the SAME store kernel, on the same GPU at the same
// clocks, writes twelve 786 MB allocations of one process at 5.7 .. 6.9 TB/s -- each buffer at its own, stable rate
// (tools/archive/micro/store_placement.hip, profiles/archive/r02f_store_placement.txt; this, not the device, is the "box-to-box" spread of
// the rollout kernel: 117 .. 141 us per launch).  So a large trajectory buffer is CHOSEN: a few candidate allocations are
// written once in the rollout's own store shape, timed with events, and the fastest one is kept.  A one-off cost of a few
// milliseconds at reservation; GU_OPT_TRAJ_CANDIDATES = 1 turns it off.  What the search may hold and when it gives up is
// stated in include/gu.h next to gu_trajectory_placement.
static double gu_now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Chosen trajectory buffers of this process, per device: a second engine on a device does not repeat the first one's search
// (it knows the write rate that search ended on) and holds far less while it looks.
struct PlacementRegistry {
    int owners = 0;        // engines that hold a chosen (searched) trajectory buffer on the device right now
    double rate = 0.0;     // bytes per ms of the best probe any search on the device ended on ...
    size_t bytes = 0;      // ... and the buffer size it was measured with (rates of very different sizes do not compare)
};
static std::mutex g_placement_mu;
static PlacementRegistry g_placement[64];

static void gu_placement_release(gu_engine *h)
{
    if (!h->traj_registered) return;
    std::lock_guard<std::mutex> lock(g_placement_mu);
    if (h->device >= 0 && h->device < 64 && g_placement[h->device].owners > 0) --g_placement[h->device].owners;
    h->traj_registered = false;
}

static hipError_t gu_traj_malloc(gu_engine *h, int32_t **p, size_t bytes)
{
#ifdef GU_EXPERIMENTS
    // EXPERIMENT (libgu_exp.so only): the trajectory buffer with the uncached memory type (MTYPE_UC: stores do not allocate in
    // L2).  The rollout kernel runs 8 % faster on it -- 112 against 121 .. 124 us per 65 536 x 1000 launch
    // (profiles/archive/r02i_uncached_ab.txt) -- but kernels that READ such a buffer can see stale bytes
    // (tests/test_gpu_mc.py::test_chunk_boundaries_do_not_change_the_result fails reproducibly when an earlier engine of the
    // process has used the same memory), and nothing in user space can flush that.  A library whose
    // contract is bit-exactness does not ship the mode: the product build has no code for it.
    if (gu_opt(h, GU_OPT_X_TRAJ_UNCACHED)) {
        const hipError_t e = hipExtMallocWithFlags((void **)p, bytes, hipDeviceMallocUncached);
        if (e == hipSuccess || e == hipErrorOutOfMemory) return e;
        (void)hipGetLastError();  // a runtime without the flag: the default type
    }
#else
    (void)h;
#endif
    return hipMalloc(p, bytes);
}

static int gu_alloc_trajectory(gu_engine *h, size_t bytes, int64_t T, int32_t **out)
{
    *out = nullptr;
    h->traj_probe_ms.clear();
    h->traj_probe_addr.clear();
    h->traj_kept = -1;
    h->traj_search_ms = 0.0f;
    h->traj_peak_bytes = 0;
    h->traj_candidates = 1;
    h->traj_probe_ms_best = h->traj_probe_ms_worst = 0.0f;
    const double t_start = gu_now_ms();
    int want = (int)gu_opt(h, GU_OPT_TRAJ_CANDIDATES);
    int far = (int)gu_opt(h, GU_OPT_TRAJ_FAR_CANDIDATES);
    size_t stride = (size_t)gu_opt(h, GU_OPT_TRAJ_STRIDE_MIB) << 20;
    const size_t far_cap = (size_t)gu_opt(h, GU_OPT_TRAJ_FAR_MIB) << 20;
    const bool exhaustive = gu_opt(h, GU_OPT_TRAJ_PROBE_ALL) != 0;  // measurement aid: probe every candidate
    // what other engines of this process already learned on the device
    int others = 0;
    double known_rate = 0.0;
    if (h->device >= 0 && h->device < 64) {
        std::lock_guard<std::mutex> lock(g_placement_mu);
        const PlacementRegistry &r = g_placement[h->device];
        others = r.owners;
        if (r.rate > 0.0 && bytes >= r.bytes / 2 && bytes <= r.bytes * 2) known_rate = r.rate;
    }
    size_t free_b = 0, total_b = 0;
    if (bytes < ((size_t)64 << 20) || want <= 1 || hipMemGetInfo(&free_b, &total_b) != hipSuccess) want = 1;
    // Never hold more than a TENTH of what is free, even briefly (a third until round 4: a co-tenant of the device saw that) -- a
    // sixteenth, and no spacers, once another engine of the process owns a chosen buffer on the device (several engines, or several
    // ranks of one process group, share it).  What the search is still worth under the closed-loop store pacing: the first
    // allocation an engine gets runs the headline launch at 105.9 .. 110.2 us (ten buffers of one process, median 107.5), the
    // searched one at 105.2 -- 2 % in the median, 4 % at worst; the probe time ranks them in the same order
    // (profiles/archive/r05f_placement_loop.txt).  Why buffers differ is still not known (tools/archive/micro/placement_*.hip).
    size_t budget = others ? free_b / 16 : free_b / 10;
    if (others) {
        want = want < 4 ? want : 4;
        far = 0;
    }
    if (far_cap > 0 && budget > far_cap) budget = far_cap;
    while (want > 1 && (size_t)want * bytes > budget) --want;
    if (want <= 1) {
        GU_HIP(gu_traj_malloc(h, out, bytes));
        h->traj_peak_bytes = bytes;
        h->traj_search_ms = (float)(gu_now_ms() - t_start);
        return GU_OK;
    }
    if (far <= 0 || bytes < ((size_t)256 << 20)) stride = 0;  // (spacers: buffers of 256 MiB and more)
    std::vector<int32_t *> cand;
    std::vector<void *> spacers;
    std::vector<float> &ms = h->traj_probe_ms;
    size_t best = 0, held = 0;
    float worst = 0.0f;
    double t_malloc = 0.0, t_probe = 0.0;
    auto release = [&](int32_t *keep) {
        for (int32_t *q : cand)
            if (q != keep) (void)hipFree(q);
        for (void *q : spacers) (void)hipFree(q);
    };
    // absolute rule: the fast class writes 6.6 .. 6.9 TB/s on MI355X, the slow one 5.5 .. 5.9 (gfx950 with all 256 CUs only)
    const double fast_rate = (h->gfx950 && h->n_cu == 256) ? 6.5e9 : 0.0;  // bytes per ms
    for (int i = 0; i < want + (stride ? far : 0); ++i) {
        if (i >= want) {
            // The back-to-back candidates are done.  Going further afield pays only where they showed two classes: a device
            // on which `want` allocations in a row write within 6 % of each other is alike everywhere it was ever probed
            // (BENCH_r02: 22 candidates over 40 GiB, 5 % apart), and the spacers cost a background wipe of all they held.
            if (i == want && !exhaustive && (double)(worst - ms[best]) < 0.06 * (double)worst) break;
            if (held + stride + bytes > budget) break;
            void *sp = nullptr;
            const double t0 = gu_now_ms();
            if (hipMalloc(&sp, stride) != hipSuccess) {
                (void)hipGetLastError();
                break;
            }
            t_malloc += gu_now_ms() - t0;
            spacers.push_back(sp);
            held += stride;
        }
        int32_t *p = nullptr;
        const double t1 = gu_now_ms();
        if (gu_traj_malloc(h, &p, bytes) != hipSuccess) {
            (void)hipGetLastError();
            break;
        }
        const double t2 = gu_now_ms();
        t_malloc += t2 - t1;
        cand.push_back(p);
        held += bytes;
        if (held > h->traj_peak_bytes) h->traj_peak_bytes = held;
        float t = 0.0f;
        int rc = gu_probe_trajectory_buffer(h, p, T, &t);
        t_probe += gu_now_ms() - t2;
        if (rc != GU_OK) {
            release(nullptr);
            ms.clear();
            h->traj_probe_addr.clear();
            return rc;
        }
        ms.push_back(t);
        h->traj_probe_addr.push_back((uint64_t)(uintptr_t)p);
        if (t < ms[best]) best = ms.size() - 1;
        worst = t > worst ? t : worst;
        if (exhaustive) continue;
        // (in between there are buffers 5 .. 8 % quicker than the slow class: only a candidate that is clearly in the fast class,
        // >= 14 % quicker than the slowest seen, ends the search early)
        if (ms.size() >= 2 && ms[best] <= 0.86f * worst) break;
        // ... or one that is fast in absolute terms: where every candidate so far is fast there is no slow one to compare with
        if (fast_rate > 0.0 && (double)bytes / (double)ms[best] >= fast_rate) break;
        // ... or one as fast as the buffer an earlier search of this process ended on
        if (known_rate > 0.0 && (double)bytes / (double)ms[best] >= 0.97 * known_rate) break;
    }
    GU_REQUIRE(!cand.empty(), GU_ERR_NOMEM, "hipMalloc of the %zu-byte trajectory buffer failed", bytes);
    const double t_rel0 = gu_now_ms();
    release(cand[best]);
    h->traj_search_ms = (float)(gu_now_ms() - t_start);
    if (gu_debug())
        fprintf(stderr, "[gu] trajectory placement: %zu candidates, %zu spacers, %.1f MiB held at most, malloc %.1f ms, probe %.1f ms, free %.1f ms\n",
                cand.size(), spacers.size(), (double)h->traj_peak_bytes / 1048576.0, t_malloc, t_probe, gu_now_ms() - t_rel0);
    *out = cand[best];
    h->traj_candidates = (int32_t)cand.size();
    h->traj_kept = (int32_t)best;
    h->traj_probe_ms_best = ms[best];
    h->traj_probe_ms_worst = worst;
    if (h->device >= 0 && h->device < 64) {
        std::lock_guard<std::mutex> lock(g_placement_mu);
        PlacementRegistry &r = g_placement[h->device];
        ++r.owners;
        h->traj_registered = true;
        const double rate = (double)bytes / (double)ms[best];
        if (rate > r.rate || !(bytes >= r.bytes / 2 && bytes <= r.bytes * 2)) {
            r.rate = rate;
            r.bytes = bytes;
        }
    }
    return GU_OK;
}

int gu_reserve_trajectory(gu_handle h, int64_t T)
{
    GU_ENTER(h);
    GU_REQUIRE(T > 0, GU_ERR_INVALID, "T <= 0");
    if (h->d_traj && T <= h->traj_T) return GU_OK;  // room for AT LEAST T rows: a buffer that is large enough is kept (and so is its placement)
    GU_HIP(hipStreamSynchronize(h->stream));
    if (h->d_traj) GU_HIP(hipFree(h->d_traj));
    h->d_traj = nullptr;
    h->traj_T = 0;
    gu_placement_release(h);
    int rc = gu_alloc_trajectory(h, 3 * (size_t)T * (size_t)h->N * sizeof(int32_t), T, &h->d_traj);
    if (rc != GU_OK) return rc;
#ifdef GU_EXPERIMENTS
    if (gu_opt(h, GU_OPT_X_TRAJ_POISON))  // debugging aid: nothing may depend on rows no rollout has written
        GU_HIP(hipMemsetAsync(h->d_traj, 0x5A, 3 * (size_t)T * (size_t)h->N * sizeof(int32_t), h->stream));
#endif
    h->traj_T = T;
    h->traj_kind = 0;
    return GU_OK;
}

// How the trajectory buffer was chosen: candidates tried, probe time of the kept and of the slowest one (ms per full write).
int gu_trajectory_placement(gu_handle h, int32_t *candidates, float *best_ms, float *worst_ms)
{
    GU_ENTER(h);
    GU_REQUIRE(h->d_traj != nullptr, GU_ERR_STATE, "no trajectory buffer: call gu_reserve_trajectory first");
    if (candidates) *candidates = h->traj_candidates;
    if (best_ms) *best_ms = h->traj_probe_ms_best;
    if (worst_ms) *worst_ms = h->traj_probe_ms_worst;
    return GU_OK;
}

int gu_trajectory_placement_detail(gu_handle h, int32_t capacity, float *probe_ms, uint64_t *address, int32_t *count, int32_t *kept,
                                   float *search_ms, uint64_t *peak_bytes)
{
    GU_ENTER(h);
    GU_REQUIRE(h->d_traj != nullptr, GU_ERR_STATE, "no trajectory buffer: call gu_reserve_trajectory first");
    const int32_t n = (int32_t)h->traj_probe_ms.size();
    if (count) *count = n;
    if (kept) *kept = h->traj_kept;
    if (search_ms) *search_ms = h->traj_search_ms;
    if (peak_bytes) *peak_bytes = h->traj_peak_bytes;
    if (probe_ms || address) GU_REQUIRE(capacity >= n, GU_ERR_INVALID, "room for %d candidates, %d were probed", capacity, n);
    for (int32_t i = 0; i < n; ++i) {
        if (probe_ms) probe_ms[i] = h->traj_probe_ms[(size_t)i];
        if (address) address[i] = h->traj_probe_addr[(size_t)i];
    }
    return GU_OK;
}

int gu_probe_trajectory(gu_handle h, float *milliseconds)
{
    GU_ENTER(h);
    GU_REQUIRE(h->d_traj != nullptr, GU_ERR_STATE, "no trajectory buffer: call gu_reserve_trajectory first");
    GU_REQUIRE(milliseconds != nullptr, GU_ERR_INVALID, "milliseconds is NULL");
    GU_HIP(hipStreamSynchronize(h->stream));
    h->traj_kind = 0;  // the probe overwrites the rows
    return gu_probe_trajectory_buffer(h, h->d_traj, h->traj_T, milliseconds);
}

int gu_rollout(gu_handle h, int64_t T, int32_t policy_kind, uint32_t flags)
{
    GU_ENTER(h);
    GU_NEED_GRID(h);
    GU_REQUIRE(T > 0 && T <= 100000000, GU_ERR_INVALID, "T %lld out of range", (long long)T);
    GU_REQUIRE((flags & ~(GU_F_AUTO_RESET | GU_F_TRAJECTORY | GU_F_STATS | GU_F_PACKED)) == 0, GU_ERR_INVALID, "unknown flags 0x%x", flags);
    GU_REQUIRE(!((flags & GU_F_PACKED) && (flags & GU_F_TRAJECTORY)), GU_ERR_INVALID, "GU_F_PACKED and GU_F_TRAJECTORY exclude each other");
    if (flags & GU_F_PACKED) GU_REQUIRE(h->S <= 65536, GU_ERR_UNSUPPORTED, "packed trajectories hold 16-bit states: grid has %d cells", h->S);
    if (flags & (GU_F_TRAJECTORY | GU_F_PACKED))
        GU_REQUIRE(h->d_traj && T <= h->traj_T, GU_ERR_STATE, "trajectory buffer holds %lld rows, need %lld: call gu_reserve_trajectory",
                   (long long)h->traj_T, (long long)T);
    if (policy_kind == GU_POLICY_STREAM)
        GU_REQUIRE(h->d_actions && T <= h->actions_T, GU_ERR_STATE, "action stream holds %lld rows, need %lld", (long long)h->actions_T, (long long)T);
    if (policy_kind == GU_POLICY_SAMPLE) GU_REQUIRE(h->has_vi, GU_ERR_STATE, "no policy table: call gu_vi_set first");
    if (policy_kind == GU_POLICY_GREEDY) {
        GU_REQUIRE(h->has_vi, GU_ERR_STATE, "no policy table: call gu_vi_set first");
        if (!h->greedy_valid) {
            int rc = gu_launch_greedy_table(h);
            if (rc != GU_OK) return rc;
        }
    }
    int rc = gu_launch_rollout(h, T, policy_kind, flags);
    if (rc == GU_OK) {
        h->stats_valid = (flags & GU_F_STATS) != 0;
        if (flags & (GU_F_TRAJECTORY | GU_F_PACKED)) h->traj_kind = h->traj_written;
    }
    return rc;
}

int gu_rollout_calibrate(gu_handle h, int64_t T, int32_t policy_kind, uint32_t flags)
{
    // (rounds 3 and 4: gu_rollout with the pacing search made now.  There is no search any more -- the launches of a kind choose
    // their period themselves from the first one on -- and this entry point is gu_rollout, kept for callers that still call it.)
    return gu_rollout(h, T, policy_kind, flags);
}

// the launch kind's slot: the general kernel's ring, else the transition-row kernel's (a launch kind of one engine runs on one of the two)
static int gu_pace_slot_in_use(gu_engine *h, int32_t policy_kind, uint32_t flags)
{
    const int auto_mode = (flags & GU_F_AUTO_RESET) ? (h->all_single_start ? 1 : 2) : 0;
    if (flags & GU_F_PACKED) {  // packed rows: the transition-row kernel's ring of its own
        const int slot = 24 + policy_kind * 3 + auto_mode;
        return h->pace[slot].active && h->pace[slot].seq && h->pace[slot].buffer == (const void *)h->d_traj ? slot : -1;
    }
    int found = -1;
    uint32_t most = 0;
    for (int base : {0, 12}) {
        const gu_engine::PaceKind &k = h->pace[base + policy_kind * 3 + auto_mode];
        if (k.active && k.seq > most && k.buffer == (const void *)h->d_traj) found = base + policy_kind * 3 + auto_mode, most = k.seq;
    }
    return found;
}

int gu_rollout_pacing_totals(gu_handle h, float *calibration_ms, int32_t *launches_spent, int32_t *kinds_paced, int32_t *kinds_from_cache,
                             int32_t *kinds_waiting)
{
    GU_ENTER(h);
    int32_t paced = 0;
    for (const gu_engine::PaceKind &k : h->pace) paced += (k.active && k.seq) ? 1 : 0;
    if (calibration_ms) *calibration_ms = 0.0f;  // (nothing is ever spent on a search: the loop runs inside the caller's launches)
    if (launches_spent) *launches_spent = 0;
    if (kinds_paced) *kinds_paced = paced;
    if (kinds_from_cache) *kinds_from_cache = 0;
    if (kinds_waiting) *kinds_waiting = 0;
    return GU_OK;
}

int gu_rollout_pace_log(gu_handle h, int32_t policy_kind, uint32_t flags, int32_t capacity, uint64_t *entries, int32_t *count, uint32_t *launches)
{
    GU_ENTER(h);
    GU_REQUIRE(policy_kind >= GU_POLICY_UNIFORM && policy_kind <= GU_POLICY_SAMPLE, GU_ERR_INVALID, "unknown policy kind %d", policy_kind);
    const int slot = gu_pace_slot_in_use(h, policy_kind, flags);
    GU_REQUIRE(slot >= 0, GU_ERR_STATE, "this launch kind keeps no schedule on the current trajectory buffer (not launched yet, or not a launch that is paced)");
    const gu_engine::PaceKind &k = h->pace[slot];
    GuPaceEntry ring[GU_PACE_RING];
    GU_HIP(hipStreamSynchronize(h->stream));
    GU_HIP(hipMemcpy(ring, h->d_pace_ring + (size_t)slot * GU_PACE_RING, sizeof(ring), hipMemcpyDeviceToHost));
    // oldest first: launches seq - n + 1 .. seq of the kind (a launch's reports are summed by the NEXT one: the last entry has no
    // verdict yet; the slot one ahead of the last launch holds the period of the launch to come)
    const uint32_t have = std::min<uint32_t>(k.seq, GU_PACE_RING - 3u);
    const uint32_t n = std::min<uint32_t>(have, capacity > 0 ? (uint32_t)capacity : 0u);
    for (uint32_t i = 0; i < n && entries; ++i) {
        const uint32_t seq = k.seq - n + 1u + i;
        const GuPaceEntry &e = ring[seq & (GU_PACE_RING - 1u)];
        const GuPaceEntry &next = ring[(seq + 1u) & (GU_PACE_RING - 1u)];
        uint64_t *o = entries + (size_t)i * 8;
        o[0] = e.seq;
        o[1] = e.unpaced ? 0u : e.period_q;
        o[2] = e.verdict | (e.phase << 8) | ((uint64_t)e.dec_q << 16);
        o[3] = e.waves;
        o[4] = e.elapsed;
        o[5] = e.ended_late;
        o[6] = e.max_behind;
        o[7] = (seq < k.seq && next.seq == seq + 1u && next.t_start > e.t_start) ? next.t_start - e.t_start : 0;  // start to start, 10 ns ticks
    }
    if (count) *count = (int32_t)n;
    if (launches) *launches = k.seq;
    return GU_OK;
}

int gu_rollout_pace_waves(gu_handle h, int32_t policy_kind, uint32_t flags, int32_t capacity, uint32_t *elapsed, int32_t *count)
{
    GU_ENTER(h);
    GU_REQUIRE(policy_kind >= GU_POLICY_UNIFORM && policy_kind <= GU_POLICY_SAMPLE, GU_ERR_INVALID, "unknown policy kind %d", policy_kind);
    const int slot = gu_pace_slot_in_use(h, policy_kind, flags);
    GU_REQUIRE(slot >= 0, GU_ERR_STATE, "this launch kind keeps no schedule on the current trajectory buffer");
    const gu_engine::PaceKind &k = h->pace[slot];
    const int64_t n = std::min<int64_t>(k.n_waves, capacity > 0 ? capacity : 0);  // (half-wave launches: N / 32 of them)
    std::vector<uint64_t> words((size_t)n);
    GU_HIP(hipStreamSynchronize(h->stream));
    // the set the LAST launch of the kind reported into (nobody has summed -- and cleared -- it yet)
    if (n) GU_HIP(hipMemcpy(words.data(), h->d_pace_slots + ((size_t)slot * 2 + (k.seq & 1u)) * (size_t)h->pace_slot_stride, (size_t)n * 8, hipMemcpyDeviceToHost));
    for (int64_t i = 0; i < n && elapsed; ++i) elapsed[i] = (words[(size_t)i] >> 63) ? (uint32_t)(words[(size_t)i] & 0x7FFFFFFFu) : 0u;
    if (count) *count = (int32_t)n;
    return GU_OK;
}

int gu_rollout_pacing(gu_handle h, int32_t policy_kind, uint32_t flags, int32_t *period, float *ms_unpaced, float *ms_paced,
                      int32_t *evaluated, float *calibration_ms)
{
    GU_ENTER(h);
    uint64_t last[8];
    int32_t n = 0;
    uint32_t launches = 0;
    const int rc = gu_rollout_pace_log(h, policy_kind, flags, 1, last, &n, &launches);
    if (rc != GU_OK) return rc;
    GU_REQUIRE(n == 1, GU_ERR_STATE, "no launch of this kind has been recorded yet");
    if (period) *period = (int32_t)((last[1] + 32u) >> 6);
    if (ms_unpaced) *ms_unpaced = 0.0f;  // (nothing runs without the limiter any more)
    if (ms_paced) *ms_paced = (float)(((last[1] + 32u) >> 6) * (uint64_t)(h->pace[gu_pace_slot_in_use(h, policy_kind, flags)].T / 16)) * 1e-5f;  // its schedule
    if (evaluated) *evaluated = (int32_t)launches;
    if (calibration_ms) *calibration_ms = 0.0f;
    return GU_OK;
}

int gu_read_trajectory(gu_handle h, int64_t t0, int64_t T, int32_t *obs, int32_t *reward, int32_t *done)
{
    GU_ENTER(h);
    GU_REQUIRE(h->d_traj && t0 >= 0 && T > 0 && t0 + T <= h->traj_T, GU_ERR_STATE, "rows [%lld,%lld) not in the trajectory buffer",
               (long long)t0, (long long)(t0 + T));
    GU_REQUIRE(h->traj_kind != 2, GU_ERR_STATE, "the buffer holds a PACKED trajectory: use gu_read_trajectory_packed");
    const size_t n = (size_t)h->N, rows = (size_t)h->traj_T * n, count = (size_t)T * n;
    int32_t *dst[3] = {obs, reward, done};
    if (h->traj_kind == 3) {  // triples [T][N][3]: taken apart on the device, a bounded number of rows at a time, then copied like planes
        const size_t chunk_rows = std::max<size_t>(1, std::min<size_t>((size_t)T, ((size_t)256 << 20) / (12 * n)));
        int rc = gu_ensure_scratch(h, chunk_rows * n * 12);
        if (rc != GU_OK) return rc;
        for (size_t r0 = 0; r0 < (size_t)T; r0 += chunk_rows) {
            const size_t nr = std::min(chunk_rows, (size_t)T - r0), cnt = nr * n;
            if ((rc = gu_launch_deinterleave(h, h->d_traj + ((size_t)t0 + r0) * n * 3, (int32_t *)h->d_scratch, (int64_t)cnt)) != GU_OK) return rc;
            GU_HIP(hipStreamSynchronize(h->stream));
            for (int k = 0; k < 3; ++k)
                if (dst[k]) GU_HIP(hipMemcpy(dst[k] + r0 * n, (int32_t *)h->d_scratch + k * cnt, cnt * 4, hipMemcpyDeviceToHost));
        }
        return GU_OK;
    }
    GU_HIP(hipStreamSynchronize(h->stream));
    for (int k = 0; k < 3; ++k)
        if (dst[k]) GU_HIP(hipMemcpy(dst[k], h->d_traj + k * rows + (size_t)t0 * n, count * 4, hipMemcpyDeviceToHost));
    return GU_OK;
}

int gu_read_trajectory_packed(gu_handle h, int64_t t0, int64_t T, uint32_t *packed)
{
    GU_ENTER(h);
    GU_REQUIRE(h->d_traj && t0 >= 0 && T > 0 && t0 + T <= h->traj_T && packed, GU_ERR_STATE, "rows [%lld,%lld) not in the trajectory buffer",
               (long long)t0, (long long)(t0 + T));
    GU_REQUIRE(h->traj_kind == 2, GU_ERR_STATE, "the last rollout did not run with GU_F_PACKED");
    GU_HIP(hipStreamSynchronize(h->stream));
    GU_HIP(hipMemcpy(packed, h->d_traj + (size_t)t0 * (size_t)h->N, (size_t)T * (size_t)h->N * 4, hipMemcpyDeviceToHost));
    return GU_OK;
}

int gu_read_stats(gu_handle h, int64_t *reward_sum, int32_t *episodes)
{
    GU_ENTER(h);
    GU_REQUIRE(h->stats_valid, GU_ERR_STATE, "the last rollout did not run with GU_F_STATS");
    const size_t n = (size_t)h->N;
    GU_HIP(hipStreamSynchronize(h->stream));
    if (reward_sum) {
        std::vector<int32_t> tmp(n);
        GU_HIP(hipMemcpy(tmp.data(), h->d_ret, n * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) reward_sum[i] = tmp[i];
    }
    if (episodes) GU_HIP(hipMemcpy(episodes, h->d_episodes_fin, n * 4, hipMemcpyDeviceToHost));
    return GU_OK;
}

// ---------------------------------------------------------------------------------- state
int gu_get_state(gu_handle h, int32_t *pos, int32_t *done, uint32_t *episode, uint64_t *tcount)
{
    GU_ENTER(h);
    const size_t n = (size_t)h->N;
    GU_HIP(hipStreamSynchronize(h->stream));
    if (pos) GU_HIP(hipMemcpy(pos, h->pos(), n * 4, hipMemcpyDeviceToHost));
    if (done) GU_HIP(hipMemcpy(done, h->done(), n * 4, hipMemcpyDeviceToHost));
    if (episode) GU_HIP(hipMemcpy(episode, h->d_episode, n * 4, hipMemcpyDeviceToHost));
    if (tcount) {
        std::vector<int32_t> off(n);
        GU_HIP(hipMemcpy(off.data(), h->d_tcount, n * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) tcount[i] = (uint64_t)((int64_t)h->steps_taken + (int64_t)off[i]);
    }
    return GU_OK;
}

int gu_set_state(gu_handle h, const int32_t *pos, const int32_t *done, const uint32_t *episode, const uint64_t *tcount)
{
    GU_ENTER(h);
    h->entry_table_ok = false;
    GU_NEED_GRID(h);
    const size_t n = (size_t)h->N;
    if (pos)
        for (size_t i = 0; i < n; ++i)
            GU_REQUIRE(pos[i] >= 0 && pos[i] < h->S, GU_ERR_INVALID, "pos[%zu]=%d outside the grid", i, pos[i]);
    uint64_t t_min = 0, t_max = 0;
    if (tcount && n) {
        t_min = t_max = tcount[0];
        for (size_t i = 1; i < n; ++i) {
            t_min = tcount[i] < t_min ? tcount[i] : t_min;
            t_max = tcount[i] > t_max ? tcount[i] : t_max;
        }
        GU_REQUIRE(t_max - t_min < ((uint64_t)1 << 31) && t_max < ((uint64_t)1 << 63), GU_ERR_INVALID,
                   "tcount: the step counts of one engine must lie within 2^31 of each other (%llu .. %llu)", (unsigned long long)t_min, (unsigned long long)t_max);
    }
    GU_HIP(hipStreamSynchronize(h->stream));
    if (pos) GU_HIP(hipMemcpy(h->pos(), pos, n * 4, hipMemcpyHostToDevice));
    if (done) {
        std::vector<int32_t> d(done, done + n);
        for (auto &x : d) x = x ? 1 : 0;
        GU_HIP(hipMemcpy(h->done(), d.data(), n * 4, hipMemcpyHostToDevice));
        h->done_bits_valid = false;  // gu_done_indices re-ballots a done[] that came from the host
    }
    if (episode) GU_HIP(hipMemcpy(h->d_episode, episode, n * 4, hipMemcpyHostToDevice));
    if (tcount) {  // the lock-step counter moves to the smallest count; every env keeps its (non-negative) offset to it
        std::vector<int32_t> off(n);
        for (size_t i = 0; i < n; ++i) off[i] = (int32_t)(tcount[i] - t_min);
        GU_HIP(hipMemcpy(h->d_tcount, off.data(), n * 4, hipMemcpyHostToDevice));
        h->steps_taken = t_min;
        h->off_lo = 0;
        h->off_hi = (int64_t)(t_max - t_min);
        h->off_exact = true;
        if (h->graph_exec) {  // (captured rollout-free step launches carry no count, but a stale graph is not worth the doubt)
            (void)hipGraphExecDestroy(h->graph_exec);
            h->graph_exec = nullptr;
        }
    }
    return gu_trail_after_set_state(h, pos != nullptr, done != nullptr);
}

int gu_done_indices(gu_handle h, int32_t *idx, int32_t *count)
{
    GU_ENTER(h);
    GU_REQUIRE(count != nullptr, GU_ERR_INVALID, "count is NULL");
    // one launch: the compaction kernel expands the ballot words (written by the step / rollout / reset kernels) into the
    // ascending index list, straight into the page-locked staging block
    int rc = gu_launch_done_compact(h);
    if (rc != GU_OK) return rc;
    GU_HIP(hipStreamSynchronize(h->stream));
    *count = (int32_t)__atomic_load_n(h->h_seq + GU_HOST_COUNT_WORD, __ATOMIC_ACQUIRE);
    if (idx && *count > 0) memcpy(idx, h->h_pin, (size_t)*count * 4);
    return GU_OK;
}

// ---------------------------------------------------------------------------------- look_step_ahead
int gu_look_step_ahead(gu_handle h, int64_t n, const int32_t *states, const int32_t *actions, int32_t care_about_terminal,
                       int32_t *next, int32_t *reward, int32_t *done)
{
    GU_ENTER(h);
    GU_NEED_GRID(h);
    GU_REQUIRE(n > 0 && states && actions, GU_ERR_INVALID, "n <= 0 or NULL inputs");
    const size_t bytes = (size_t)n * 4;
    int rc = gu_ensure_scratch(h, 5 * bytes);
    if (rc != GU_OK) return rc;
    int32_t *d = (int32_t *)h->d_scratch;
    GU_HIP(hipMemcpyAsync(d, states, bytes, hipMemcpyHostToDevice, h->stream));
    GU_HIP(hipMemcpyAsync(d + n, actions, bytes, hipMemcpyHostToDevice, h->stream));
    rc = gu_launch_lookahead(h, n, d, d + n, care_about_terminal != 0, d + 2 * n, d + 3 * n, d + 4 * n);
    if (rc != GU_OK) return rc;
    GU_HIP(hipStreamSynchronize(h->stream));
    if (__atomic_load_n(h->h_seq + GU_HOST_ERR_WORD, __ATOMIC_ACQUIRE)) {  // raised by the kernel; name the first offender
        __atomic_store_n(h->h_seq + GU_HOST_ERR_WORD, 0u, __ATOMIC_RELAXED);
        for (int64_t i = 0; i < n; ++i) {
            GU_REQUIRE(states[i] >= 0 && states[i] < h->S, GU_ERR_INVALID, "state %d outside the grid", states[i]);
            GU_REQUIRE(GU_ACTION_OK(actions[i]), GU_ERR_INVALID, "action %d outside 0..3", actions[i]);
        }
        return gu_fail(GU_ERR_INVALID, "a state outside the grid or an action outside 0..3 was passed");
    }
    if (next) GU_HIP(hipMemcpy(next, d + 2 * n, bytes, hipMemcpyDeviceToHost));
    if (reward) GU_HIP(hipMemcpy(reward, d + 3 * n, bytes, hipMemcpyDeviceToHost));
    if (done) GU_HIP(hipMemcpy(done, d + 4 * n, bytes, hipMemcpyDeviceToHost));
    return GU_OK;
}

// ---------------------------------------------------------------------------------- pinned host memory
int gu_host_alloc(size_t bytes, void **ptr)
{
    GU_REQUIRE(ptr != nullptr && bytes > 0, GU_ERR_INVALID, "ptr is NULL or bytes == 0");
    *ptr = nullptr;
    GU_HIP(hipHostMalloc(ptr, bytes, hipHostMallocDefault));
    std::lock_guard<std::mutex> lock(g_host_alloc_mu);
    g_host_allocs.emplace_back((uintptr_t)*ptr, (uintptr_t)*ptr + bytes);
    return GU_OK;
}

int gu_host_free(void *ptr)
{
    g_pinned_generation.fetch_add(1, std::memory_order_acq_rel);  // validated GU_F_PINNED_IO ranges are looked up again
    {
        std::lock_guard<std::mutex> lock(g_host_alloc_mu);
        for (size_t i = 0; i < g_host_allocs.size(); ++i)
            if (g_host_allocs[i].first == (uintptr_t)ptr) {
                g_host_allocs.erase(g_host_allocs.begin() + (long)i);
                break;
            }
    }
    if (ptr) GU_HIP(hipHostFree(ptr));
    return GU_OK;
}

// ---------------------------------------------------------------------------------- stream / timing
// Wait for an event that has been recorded on the engine's stream: polled for up to GU_OPT_SYNC_SPIN_US (5 ms), then the runtime's
// own interrupt-driven wait.  hipEventSynchronize learns of the end 10 .. 20 us late (1 % of a 2 ms block of launches); a poll sees it
// within a microsecond.  Bounded so that a long wait does not hold a core; 0 = no poll (more ranks or engines than host cores).
static int gu_wait_event(const gu_engine *h, hipEvent_t ev)
{
    const int64_t spin_us = gu_opt(h, GU_OPT_SYNC_SPIN_US);
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t turn = 0; spin_us > 0; ++turn) {
        const hipError_t q = hipEventQuery(ev);
        if (q == hipSuccess) return GU_OK;
        if (q != hipErrorNotReady) {
            (void)hipGetLastError();
            break;
        }
        __builtin_ia32_pause();
        if ((turn & 63u) == 63u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spin_us)) break;
    }
    GU_HIP(hipEventSynchronize(ev));
    return GU_OK;
}

int gu_sync(gu_handle h)
{
    GU_ENTER(h);
    if (h->ev_sync) {
        GU_HIP(hipEventRecord(h->ev_sync, h->stream));
        return gu_wait_event(h, h->ev_sync);
    }
    GU_HIP(hipStreamSynchronize(h->stream));
    return GU_OK;
}

int gu_timer_begin(gu_handle h)
{
    GU_ENTER(h);
    GU_HIP(hipEventRecord(h->ev_begin, h->stream));
    return GU_OK;
}

int gu_timer_end(gu_handle h, float *milliseconds)
{
    GU_ENTER(h);
    GU_REQUIRE(milliseconds != nullptr, GU_ERR_INVALID, "milliseconds is NULL");
    GU_HIP(hipEventRecord(h->ev_end, h->stream));
    int rc = gu_wait_event(h, h->ev_end);
    if (rc != GU_OK) return rc;
    GU_HIP(hipEventElapsedTime(milliseconds, h->ev_begin, h->ev_end));
    return GU_OK;
}

// Lap timing: gu_timer_mark records one event on the stream per call; gu_timer_laps waits for the last one and returns the
// n_marks - 1 intervals between consecutive marks, then forgets the marks.
int gu_timer_mark(gu_handle h)
{
    GU_ENTER(h);
    if (h->n_marks == h->ev_marks.size()) {
        hipEvent_t ev = nullptr;
        GU_HIP(hipEventCreate(&ev));
        h->ev_marks.push_back(ev);
    }
    GU_HIP(hipEventRecord(h->ev_marks[h->n_marks], h->stream));
    ++h->n_marks;
    return GU_OK;
}

int gu_timer_laps(gu_handle h, float *milliseconds, int32_t capacity, int32_t *count)
{
    GU_ENTER(h);
    GU_REQUIRE(count != nullptr, GU_ERR_INVALID, "count is NULL");
    const size_t laps = h->n_marks ? h->n_marks - 1 : 0;
    *count = (int32_t)laps;
    if (h->n_marks) GU_HIP(hipEventSynchronize(h->ev_marks[h->n_marks - 1]));
    if (milliseconds) {
        GU_REQUIRE((size_t)capacity >= laps, GU_ERR_INVALID, "room for %d laps, %zu recorded", capacity, laps);
        for (size_t i = 0; i < laps; ++i) GU_HIP(hipEventElapsedTime(milliseconds + i, h->ev_marks[i], h->ev_marks[i + 1]));
    }
    h->n_marks = 0;
    return GU_OK;
}

}  // extern "C"
