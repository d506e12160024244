// gu_maze.hip -- on-device batched maze generation: one distinct maze per env group.
//
// The reference builds ONE maze per env instance on the host (core/envs/maze_generation.py:41-149, ~4 ms per
// 32x32 grid, i.e. minutes for tens of thousands of distinct grids).  Here one lane carves one maze with the same
// algorithm -- depth-first "recursive backtracker" in strides of two from a random origin, neighbour order
// +x, -x, +y, -y (:61-68), then one start 'x' and one goal 'G' on two distinct open cells (:134-142) -- but the
// draws come from the build's counter RNG (stream 3, key = (maze_seed, global grid id), counter = draw index)
// instead of the two process-global host RNGs, so grids do not depend on how the batch is sharded.  Parity is
// therefore against the CPU restatement of THIS generator (oracle/gu_oracle.c: gu_oracle_generate_maze); against
// the reference's generator the tests check the structural invariants (room cells share the origin's parity,
// the open cells form a spanning tree of the rooms, exactly one 'x' and one 'G' on distinct open cells).
//
// A second kernel compiles the wall bytes + goal into the engine's per-cell planes on the device.
#include "gu_internal.hpp"
#include "gu_rng.hpp"

#define GU_RNG_STREAM_MAZE 3u

__host__ __device__ __forceinline__ uint32_t gu_mulhi(uint32_t w, uint32_t n) { return (uint32_t)(((uint64_t)w * n) >> 32); }

struct MazeArgs {
    uint8_t *wall;       // [G][S] out: 1 = wall
    uint16_t *stack;     // [G][rooms_max] scratch
    int32_t *start, *goal;  // [G] out
    int32_t *status;     // != 0 if some maze ended with fewer than two open cells
    int32_t W, H, n_grids, rooms_max;
    uint32_t seed_prefix, grid_id0;
};

__global__ void __launch_bounds__(64) gu_maze_carve_kernel(const MazeArgs a)
{
    const int32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.n_grids) return;
    const int32_t W = a.W, H = a.H, S = W * H;
    uint8_t *wall = a.wall + (int64_t)g * S;
    uint16_t *stack = a.stack + (int64_t)g * a.rooms_max;
    const uint32_t prefix = gu_rng_prefix(a.seed_prefix, a.grid_id0 + (uint32_t)g);
    uint32_t k = 0;
    for (int32_t s = 0; s < S; ++s) wall[s] = 1;
    int32_t x = (int32_t)gu_mulhi(gu_rng_word(prefix, GU_RNG_STREAM_MAZE, k++), (uint32_t)W);
    int32_t y = (int32_t)gu_mulhi(gu_rng_word(prefix, GU_RNG_STREAM_MAZE, k++), (uint32_t)H);
    const int32_t origin = y * W + x;
    int32_t sp = 0;
    bool moved = false;
    for (;;) {
        int32_t opt[4], n = 0;  // unvisited stride-2 neighbours; a room is visited iff carved (or the origin)
        if (x + 2 < W && wall[y * W + x + 2] && y * W + x + 2 != origin) opt[n++] = y * W + x + 2;
        if (x - 2 >= 0 && wall[y * W + x - 2] && y * W + x - 2 != origin) opt[n++] = y * W + x - 2;
        if (y + 2 < H && wall[(y + 2) * W + x] && (y + 2) * W + x != origin) opt[n++] = (y + 2) * W + x;
        if (y - 2 >= 0 && wall[(y - 2) * W + x] && (y - 2) * W + x != origin) opt[n++] = (y - 2) * W + x;
        if (n > 0) {
            const int32_t nb = opt[gu_mulhi(gu_rng_word(prefix, GU_RNG_STREAM_MAZE, k++), (uint32_t)n)];
            const int32_t cur = y * W + x;
            stack[sp++] = (uint16_t)cur;
            wall[(cur + nb) / 2] = 0;  // the cell between them (same row or same column)
            wall[nb] = 0;
            wall[cur] = 0;
            x = nb % W;
            y = nb / W;
            moved = true;
        } else if (sp > 0) {
            const int32_t cur = stack[--sp];
            x = cur % W;
            y = cur / W;
        } else {
            break;
        }
    }
    int32_t n_open = 0;
    for (int32_t s = 0; s < S; ++s) n_open += wall[s] == 0;
    if (!moved || n_open < 2) {
        atomicExch(a.status, 1);
        a.start[g] = a.goal[g] = 0;
        return;
    }
    const uint32_t i = gu_mulhi(gu_rng_word(prefix, GU_RNG_STREAM_MAZE, k++), (uint32_t)n_open);
    uint32_t j = gu_mulhi(gu_rng_word(prefix, GU_RNG_STREAM_MAZE, k++), (uint32_t)(n_open - 1));
    if (j >= i) ++j;
    uint32_t seen = 0;
    for (int32_t s = 0; s < S; ++s) {
        if (wall[s]) continue;
        if (seen == i) a.start[g] = s;
        if (seen == j) a.goal[g] = s;
        ++seen;
    }
}

struct MazeCompileArgs {
    const uint8_t *wall;
    const int32_t *start, *goal;
    uint8_t *cell, *raw;  // [G][flags | reward]
    int32_t *starts, *n_starts;  // [G][1], [G]
    int32_t W, H, S, cell_bytes, n_grids;
};

__global__ void __launch_bounds__(256) gu_maze_compile_kernel(const MazeCompileArgs a)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)a.n_grids * a.S) return;
    const int32_t g = (int32_t)(i / a.S), s = (int32_t)(i % a.S), x = s % a.W, y = s / a.W;
    const uint8_t *wall = a.wall + (int64_t)g * a.S;
    const bool term = s == a.goal[g];  // generated mazes have one goal and no lava (quirk 9)
    uint8_t open = 0;
    if (y > 0 && !wall[s - a.W]) open |= 1u;
    if (x < a.W - 1 && !wall[s + 1]) open |= 2u;
    if (y < a.H - 1 && !wall[s + a.W]) open |= 4u;
    if (x > 0 && !wall[s - 1]) open |= 8u;
    const uint8_t t = (term ? (GU_CELL_TERM | GU_CELL_RPLUS) : 0) | (wall[s] ? GU_CELL_WALL : 0);
    const int64_t base = 2 * (int64_t)a.cell_bytes * g;
    a.raw[base + s] = open | t;
    a.cell[base + s] = (term ? 0 : open) | t;
    a.raw[base + a.cell_bytes + s] = a.cell[base + a.cell_bytes + s] = (uint8_t)(int8_t)(term ? 10 : -1);
    if (s == 0) {
        a.starts[g] = a.start[g];
        a.n_starts[g] = 1;
    }
}

__global__ void __launch_bounds__(256) gu_init_pos_kernel(int32_t *pos, int32_t *reward_done, const int32_t *starts,
                                                          int64_t N, int64_t group, int32_t max_starts)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N) return;
    pos[e] = starts[(e / group) * max_starts];
    reward_done[e] = 0;
    reward_done[N + e] = 0;
}

extern "C" int gu_generate_mazes(gu_handle h, int32_t n_grids, int32_t W, int32_t H, uint64_t maze_seed)
{
    int rc = gu_use_device(h);
    if (rc != GU_OK) return rc;
    h->entry_table_ok = false;
    GU_REQUIRE(n_grids > 0 && h->N % n_grids == 0, GU_ERR_INVALID, "n_grids=%d must divide num_envs=%lld", n_grids, (long long)h->N);
    GU_REQUIRE(W > 0 && H > 0 && (W >= 4 || H >= 4), GU_ERR_INVALID, "a %d x %d grid has no room for a corridor (need max(W,H) >= 4)", W, H);
    GU_REQUIRE((int64_t)W * H <= 65535, GU_ERR_UNSUPPORTED, "generated mazes are limited to 65535 cells");
    const int64_t group = h->N / n_grids;
    GU_REQUIRE(h->env_id0 % group == 0, GU_ERR_INVALID, "env_id0 must be a multiple of the group size %lld", (long long)group);
    const int32_t S = W * H, cell_bytes = (S + 15) & ~15, rooms_max = ((W + 1) / 2) * ((H + 1) / 2) + 1;
    GU_REQUIRE(2 * (int64_t)cell_bytes * n_grids < (1ll << 31), GU_ERR_UNSUPPORTED, "%d grids of %d cells exceed 2 GiB of records", n_grids, S);
    std::vector<uint8_t> no_planes;  // allocate + set metadata through the common path; planes are filled on the device
    std::vector<int32_t> st((size_t)n_grids, 0), ns((size_t)n_grids, 1);
    rc = gu_install_grids(h, n_grids, W, H, no_planes, no_planes, no_planes, st, ns, 1);
    if (rc != GU_OK) return rc;
    h->has_grid = false;
    const size_t wall_bytes = ((size_t)n_grids * S + 15) & ~(size_t)15, stack_bytes = ((size_t)n_grids * rooms_max * 2 + 15) & ~(size_t)15;
    rc = gu_ensure_scratch(h, wall_bytes + stack_bytes + 2 * (size_t)n_grids * 4 + 16);
    if (rc != GU_OK) return rc;
    uint8_t *d_wall = (uint8_t *)h->d_scratch;
    uint16_t *d_stack = (uint16_t *)(d_wall + wall_bytes);
    int32_t *d_start = (int32_t *)((uint8_t *)d_stack + stack_bytes), *d_goal = d_start + n_grids, *d_status = d_goal + n_grids;
    GU_HIP(hipMemsetAsync(d_status, 0, 4, h->stream));
    MazeArgs ma{d_wall, d_stack, d_start, d_goal, d_status, W, H, n_grids, rooms_max, gu_rng_seed_prefix(maze_seed),
                (uint32_t)(h->env_id0 / group)};
    hipLaunchKernelGGL(gu_maze_carve_kernel, dim3((unsigned)((n_grids + 63) / 64)), dim3(64), 0, h->stream, ma);
    MazeCompileArgs ca{d_wall, d_start, d_goal, h->d_cell, h->d_cell_raw, h->d_starts, h->d_nstarts, W, H, S, cell_bytes, n_grids};
    const int64_t cells = (int64_t)n_grids * S;
    hipLaunchKernelGGL(gu_maze_compile_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, h->stream, ca);
    hipLaunchKernelGGL(gu_init_pos_kernel, dim3((unsigned)((h->N + 255) / 256)), dim3(256), 0, h->stream, h->pos(), h->reward(),
                       h->d_starts, h->N, group, 1);
    GU_HIP(hipGetLastError());
    int32_t status = 0;
    GU_HIP(hipMemcpyAsync(&status, d_status, 4, hipMemcpyDeviceToHost, h->stream));
    GU_HIP(hipMemcpyAsync(&h->start0, h->d_starts, 4, hipMemcpyDeviceToHost, h->stream));  // host copy of grid 0's start cell
    GU_HIP(hipStreamSynchronize(h->stream));
    GU_REQUIRE(status == 0, GU_ERR_INVALID, "a generated maze has fewer than two open cells");
    h->has_grid = true;
    return GU_OK;
}
