// gu_bfs.hip -- batched breadth-first shortest paths, one lane per grid.
//
// Restates the search of the reference's demo script core/algorithms/maze_solving.py for every grid of the engine:
//   graph      edge s -> s' for each action whose look_step_ahead(s, a, care_about_terminal=False) moves the agent,
//              children in action order UP, RIGHT, DOWN, LEFT                                  (:43-50)
//   search     FIFO queue from the grid's first start cell; a child is enqueued the first time it is seen;
//              the search stops when a TERMINAL state (goal or lava) is dequeued             (:123-169)
//   path       actions along the parent chain, first action first                            (:171-193)
// Many small graphs (one per grid, <= 65 535 nodes) rather than one large one, so the natural mapping is the one
// used for maze carving: one lane runs one grid's search sequentially -- which also reproduces the reference's
// tie-breaking on open grids by construction -- with its queue and parent table in a global scratch slice.
#include "gu_internal.hpp"

struct BfsArgs {
    const uint8_t *raw;       // [G][flags | reward] care_about_terminal=False records
    const int32_t *starts;    // [G][max_starts]
    uint16_t *queue, *parent; // [G][S] scratch
    int8_t *path;             // [G][max_path] out
    int32_t *path_len;        // [G] out: number of actions, -1 = no terminal reachable, -2 = longer than max_path
    int32_t *terminal;        // [G] out: the terminal state reached (or -1)
    int64_t grid_stride;
    int32_t W, S, n_grids, max_starts, max_path;
};

__global__ void __launch_bounds__(64) gu_bfs_kernel(const BfsArgs a)
{
    const int32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= a.n_grids) return;
    const uint8_t *flags = a.raw + (int64_t)g * a.grid_stride;
    uint16_t *queue = a.queue + (int64_t)g * a.S, *parent = a.parent + (int64_t)g * a.S;
    const uint16_t NONE = 0xFFFF;
    for (int32_t s = 0; s < a.S; ++s) parent[s] = NONE;
    const int32_t start = a.starts[(int64_t)g * a.max_starts];
    const int32_t delta[4] = {-a.W, 1, a.W, -1};
    int32_t head = 0, tail = 0, found = -1;
    if (flags[start] & GU_CELL_WALL) {  // wall cells are not nodes of the graph (:45); the reference fails here
        a.terminal[g] = -1;
        a.path_len[g] = -1;
        return;
    }
    queue[tail++] = (uint16_t)start;
    parent[start] = (uint16_t)start;  // the root is its own parent
    while (head < tail) {
        const int32_t s = queue[head++];
        const uint32_t f = flags[s];
        if (f & GU_CELL_TERM) { found = s; break; }
#pragma unroll
        for (int32_t act = 0; act < 4; ++act) {
            if (!((f >> act) & 1u)) continue;  // the move does not change the position: no edge
            const int32_t c = s + delta[act];
            if (parent[c] != NONE) continue;
            parent[c] = (uint16_t)s;
            queue[tail++] = (uint16_t)c;
        }
    }
    a.terminal[g] = found;
    if (found < 0) { a.path_len[g] = -1; return; }
    int32_t len = 0;
    for (int32_t s = found; s != start; s = parent[s]) ++len;
    if (len > a.max_path) { a.path_len[g] = -2; return; }
    a.path_len[g] = len;
    int8_t *out = a.path + (int64_t)g * a.max_path;
    int32_t i = len;
    for (int32_t s = found; s != start; s = parent[s]) {
        const int32_t d = s - parent[s];  // calculate_action (:113-127)
        out[--i] = (int8_t)(d == 1 ? 1 : d == -1 ? 3 : d > 1 ? 2 : 0);
    }
}

extern "C" int gu_shortest_paths(gu_handle h, int32_t max_path, int8_t *path, int32_t *path_len, int32_t *terminal)
{
    int rc = gu_use_device(h);
    if (rc != GU_OK) return rc;
    GU_REQUIRE(h->has_grid, GU_ERR_STATE, "no grid set");
    GU_REQUIRE(h->S <= 65534, GU_ERR_UNSUPPORTED, "shortest paths are limited to grids of 65534 cells");
    GU_REQUIRE(max_path > 0 && path && path_len, GU_ERR_INVALID, "max_path <= 0 or NULL output");
    const size_t G = (size_t)h->n_grids, S = (size_t)h->S;
    const size_t off_parent = (G * S * 2 + 15) & ~(size_t)15, off_path = off_parent * 2;
    const size_t off_len = off_path + ((G * (size_t)max_path + 15) & ~(size_t)15), off_term = off_len + G * 4;
    rc = gu_ensure_scratch(h, off_term + G * 4);
    if (rc != GU_OK) return rc;
    char *base = (char *)h->d_scratch;
    BfsArgs a{h->d_cell_raw, h->d_starts, (uint16_t *)base, (uint16_t *)(base + off_parent), (int8_t *)(base + off_path),
              (int32_t *)(base + off_len), (int32_t *)(base + off_term), 2 * (int64_t)h->cell_bytes, h->W, h->S, h->n_grids,
              h->max_starts, max_path};
    hipLaunchKernelGGL(gu_bfs_kernel, dim3((unsigned)((G + 63) / 64)), dim3(64), 0, h->stream, a);
    GU_HIP(hipGetLastError());
    GU_HIP(hipMemcpyAsync(path, base + off_path, G * (size_t)max_path, hipMemcpyDeviceToHost, h->stream));
    GU_HIP(hipMemcpyAsync(path_len, base + off_len, G * 4, hipMemcpyDeviceToHost, h->stream));
    if (terminal) GU_HIP(hipMemcpyAsync(terminal, base + off_term, G * 4, hipMemcpyDeviceToHost, h->stream));
    GU_HIP(hipStreamSynchronize(h->stream));
    return GU_OK;
}
