// gu_render.hip -- headless RGB frames for a range of envs (second half).
//
// The reference draws its 'graphic' mode with pyglet textures in a window (core/envs/rendering.py:236-343) -- a GUI
// that cannot exist on a headless GPU box.  This kernel produces the equivalent information as plain RGB arrays,
// one thread per pixel: ground / wall / goal / lava tiles, a thin grid line on the top and left edge of every cell, and
// the agent as an inset square on its current cell.  WHICH of the four textures a cell gets, and the geometry of the
// policy arrows, are the reference's and are pinned to it (tests/golden/arrows.json: the viewer's tile loop and its
// render_policy_arrows, lifted from the parsed module and run without a window; oracle/render.py restates them and the
// rasterisation rule below).  The four flat colours stand in for the textures (core/resources, OUT OF SCOPE) and, like
// the grid line and the agent square, are build-defined.  With gu_trail_enable the frame also carries the viewer's agent trail
// (rendering.py:287-311): which cells, which alpha, which order are the reference's (tests/golden/trail.json); the blend is the
// integer rule stated in the kernel.
#include "gu_internal.hpp"

struct RenderArgs {
    const uint8_t *cell;   // [G][flags | reward] (absorbing map: flags carry TERM / reward code / WALL)
    const uint8_t *kind;   // [G][cell_bytes] texture class of the reference's viewer, or nullptr (device mazes: from the flags)
    const int32_t *pos;    // [N]
    uint8_t *rgb;          // [n][H*px][W*px][3]
    int64_t env0, n_envs, group, grid_stride;
    int32_t W, H, px, n_grids;
    // agent trail (gu_trail.hip; nullptr / 0 when off)
    const int32_t *trail, *trail_len, *trail_head;
    const uint32_t *trail_alpha;
    int32_t trail_cap;
};

// Which texture a cell gets is the reference's rule (core/envs/rendering.py:119-133: goal, else lava, else wall, else ground;
// pinned by tests/golden/arrows.json "tiles"); the COLOURS stand in for its four textures and are build-defined.
__device__ __forceinline__ uint32_t gu_tile_kind(const uint8_t *kind, uint32_t f, int64_t index)
{
    if (kind) return kind[index];
    // device-generated mazes: one goal, no lava, the goal never on a wall -- the flags are unambiguous
    return (f & GU_CELL_TERM) ? ((f & GU_CELL_RMINUS) ? 2u : 3u) : (f & GU_CELL_WALL) ? 1u : 0u;
}

__device__ __forceinline__ void gu_tile_colour(uint32_t k, uint8_t &r, uint8_t &g, uint8_t &b)
{
    if (k == 3u) { r = 40; g = 180; b = 60; }        // goal   (wbs_texture_05_resized_green.jpg)
    else if (k == 2u) { r = 220; g = 60; b = 30; }   // lava   (lava-resized.jpg)
    else if (k == 1u) { r = 64; g = 64; b = 64; }    // wall   (wbs_texture_05_resized_wall.jpg)
    else { r = 220; g = 220; b = 220; }              // ground (wbs_texture_05_resized.jpg)
}

__global__ void __launch_bounds__(256) gu_render_kernel(const RenderArgs a)
{
    const int64_t frame_px = (int64_t)a.W * a.px * a.H * a.px;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.n_envs * frame_px) return;
    const int64_t k = i / frame_px, p = i % frame_px;
    const int32_t wpx = a.W * a.px;
    const int32_t y = (int32_t)(p / wpx), x = (int32_t)(p % wpx);
    const int32_t cy = y / a.px, cx = x / a.px, iy = y % a.px, ix = x % a.px;
    const int64_t e = a.env0 + k;
    const int64_t grid = a.n_grids > 1 ? e / a.group : 0;
    const uint8_t *flags = a.cell + grid * a.grid_stride;
    const int32_t s = cy * a.W + cx;
    uint8_t r, g, b;
    gu_tile_colour(gu_tile_kind(a.kind, flags[s], grid * (a.grid_stride / 2) + s), r, g, b);
    if (a.px >= 4 && (iy == 0 || ix == 0)) { r = r * 3 / 4; g = g * 3 / 4; b = b * 3 / 4; }  // grid line
    const int32_t lo = a.px / 4, hi = a.px - a.px / 4;
    if (s == a.pos[e] && iy >= lo && iy < hi && ix >= lo && ix < hi) { r = 40; g = 90; b = 220; }  // agent
    // The agent's trail (rendering.py:287-311), drawn last like there: newest entry first, every entry a quad over its tile with
    // alpha 0.3 * 0.96^(i + 1), entries on the agent's current cell skipped (their alpha step is still taken).  The quad's corner
    // colours are the reference's -- red, yellow, green, blue from the bottom-left corner counter-clockwise (glColor4f clamps
    // 0xFF to 1) -- interpolated bilinearly at the pixel centre; blended in integer arithmetic, 16 fractional bits of alpha,
    // rounded to nearest per entry: c = (c * (65536 - A) + colour * A + 32768) >> 16.
    if (a.trail_cap && s != a.pos[e]) {
        const int32_t len = a.trail_len[e], head = a.trail_head[e];
        const uint32_t two = 2u * (uint32_t)a.px, u = 2u * (uint32_t)ix + 1u, v = 2u * (uint32_t)(a.px - 1 - iy) + 1u;  // doubled, y up
        const uint32_t cr = (255u * (two - v) + (uint32_t)a.px) / two;                       // red + yellow on the bottom edge
        const uint32_t cg = (255u * u + (uint32_t)a.px) / two;                               // yellow + green on the right edge
        const uint32_t cb = (255u * (two - u) * v + two * two / 2u) / (two * two);           // blue in the top-left corner
        uint32_t R = r, G = g, B = b;
        for (int32_t k = 0; k < len; ++k) {
            int32_t slot = head - 1 - k;
            slot += slot < 0 ? a.trail_cap : 0;
            if (a.trail[e * a.trail_cap + slot] != s) continue;
            const uint32_t A = a.trail_alpha[k];
            R = (R * (65536u - A) + cr * A + 32768u) >> 16;
            G = (G * (65536u - A) + cg * A + 32768u) >> 16;
            B = (B * (65536u - A) + cb * A + 32768u) >> 16;
        }
        r = (uint8_t)R, g = (uint8_t)G, b = (uint8_t)B;
    }
    uint8_t *out = a.rgb + 3 * i;
    out[0] = r;
    out[1] = g;
    out[2] = b;
}

// ---- policy arrows (stands in for Viewer.render_policy_arrows, core/envs/rendering.py:159-212) ----------------------
// The reference adds, for every state that is neither terminal nor a wall and every action with probability >= 0.1,
// a line of round(p * 20) pixels from the tile centre in the action's direction and a triangular head of half-width 5
// and height 5, on tiles of 52 pixels.  Here the same figure is rasterised per pixel, scaled by cell_px / 52, in
// integer arithmetic on doubled coordinates (pixel centres relative to the tile centre), so that the rule can be
// restated exactly (tests/test_gpu_render.py): with (t, u) = (along, across) the action's direction,
//   shaft:  0 <= t,  52 t <= 2 L px,            |u| <= max(1, px / 26)
//   head :  2 L px < 52 t <= 2 (L + 5) px,      52 |u| <= 2 (L + 5) px - 52 t
struct PolicyRenderArgs {
    const uint8_t *cell;
    const uint8_t *kind;
    const double *pi;  // [S][4]
    uint8_t *rgb;      // [H*px][W*px][3]
    int32_t W, H, px;
};

__global__ void __launch_bounds__(256) gu_render_policy_kernel(const PolicyRenderArgs a)
{
    const int64_t frame_px = (int64_t)a.W * a.px * a.H * a.px;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= frame_px) return;
    const int32_t wpx = a.W * a.px;
    const int32_t y = (int32_t)(i / wpx), x = (int32_t)(i % wpx);
    const int32_t cy = y / a.px, cx = x / a.px, iy = y % a.px, ix = x % a.px;
    const int32_t s = cy * a.W + cx;
    const uint32_t f = a.cell[s];
    uint8_t r, g, b;
    gu_tile_colour(gu_tile_kind(a.kind, f, s), r, g, b);
    if (a.px >= 4 && (iy == 0 || ix == 0)) { r = r * 3 / 4; g = g * 3 / 4; b = b * 3 / 4; }  // grid line
    if (!(f & (GU_CELL_TERM | GU_CELL_WALL))) {
        const int64_t X = 2 * ix + 1 - a.px, Y = a.px - (2 * iy + 1);  // doubled, y up
        const int64_t shaft = a.px / 26 > 1 ? a.px / 26 : 1;
        bool on = false;
#pragma unroll
        for (int act = 0; act < 4; ++act) {
            const double p = a.pi[4 * (int64_t)s + act];
            if (!(p >= 0.1)) continue;  // "arrow base length too small to render" (:176-178); NaN draws nothing
            double Ld = rint(p * 20.0);
            Ld = Ld > 1000.0 ? 1000.0 : Ld;
            const int64_t L = (int64_t)Ld;
            const int64_t t = act == 0 ? Y : act == 1 ? X : act == 2 ? -Y : -X;  // UP, RIGHT, DOWN, LEFT
            int64_t u = (act & 1) ? Y : X;
            u = u < 0 ? -u : u;
            const int64_t t52 = 52 * t, base = 2 * L * a.px, tip = 2 * (L + 5) * a.px;
            on |= t >= 0 && t52 <= base && u <= shaft;
            on |= t52 > base && t52 <= tip && 52 * u <= tip - t52;
        }
        if (on) { r = 20; g = 20; b = 20; }
    }
    uint8_t *out = a.rgb + 3 * i;
    out[0] = r;
    out[1] = g;
    out[2] = b;
}

extern "C" int gu_render_policy_rgb(gu_handle h, int32_t cell_px, uint8_t *rgb)
{
    int rc = gu_use_device(h);
    if (rc != GU_OK) return rc;
    GU_REQUIRE(h->has_grid, GU_ERR_STATE, "no grid set");
    GU_REQUIRE(h->has_vi, GU_ERR_STATE, "no policy table: call gu_vi_set first");
    GU_REQUIRE(rgb && cell_px >= 1 && cell_px <= 64, GU_ERR_INVALID, "rgb is NULL or cell_px outside 1..64");
    const int64_t pixels = (int64_t)h->W * cell_px * h->H * cell_px;
    GU_REQUIRE(pixels * 3 <= (1ll << 32), GU_ERR_INVALID, "%lld pixels are too many for one call", (long long)pixels);
    rc = gu_ensure_scratch(h, (size_t)pixels * 3);
    if (rc != GU_OK) return rc;
    PolicyRenderArgs a{h->d_cell, h->d_kind, h->d_pi[h->vi_cur], (uint8_t *)h->d_scratch, h->W, h->H, cell_px};
    hipLaunchKernelGGL(gu_render_policy_kernel, dim3((unsigned)((pixels + 255) / 256)), dim3(256), 0, h->stream, a);
    GU_HIP(hipGetLastError());
    GU_HIP(hipMemcpyAsync(rgb, h->d_scratch, (size_t)pixels * 3, hipMemcpyDeviceToHost, h->stream));
    GU_HIP(hipStreamSynchronize(h->stream));
    return GU_OK;
}

extern "C" int gu_render_rgb(gu_handle h, int64_t env0, int64_t n_envs, int32_t cell_px, uint8_t *rgb)
{
    int rc = gu_use_device(h);
    if (rc != GU_OK) return rc;
    GU_REQUIRE(h->has_grid, GU_ERR_STATE, "no grid set");
    GU_REQUIRE(rgb && env0 >= 0 && n_envs > 0 && env0 + n_envs <= h->N, GU_ERR_INVALID, "env range [%lld,%lld) outside the batch",
               (long long)env0, (long long)(env0 + n_envs));
    GU_REQUIRE(cell_px >= 1 && cell_px <= 64, GU_ERR_INVALID, "cell_px must be 1..64");
    const int64_t pixels = n_envs * (int64_t)h->W * cell_px * h->H * cell_px;
    GU_REQUIRE(pixels * 3 <= (1ll << 32), GU_ERR_INVALID, "%lld pixels are too many for one call", (long long)pixels);
    rc = gu_ensure_scratch(h, (size_t)pixels * 3);
    if (rc != GU_OK) return rc;
    RenderArgs a{h->d_cell, h->d_kind, h->pos(), (uint8_t *)h->d_scratch, env0, n_envs, h->group, 2 * (int64_t)h->cell_bytes,
                 h->W, h->H, cell_px, h->n_grids, h->d_trail, h->d_trail_len, h->d_trail_head, h->d_trail_alpha, h->trail_cap};
    hipLaunchKernelGGL(gu_render_kernel, dim3((unsigned)((pixels + 255) / 256)), dim3(256), 0, h->stream, a);
    GU_HIP(hipGetLastError());
    GU_HIP(hipMemcpyAsync(rgb, h->d_scratch, (size_t)pixels * 3, hipMemcpyDeviceToHost, h->stream));
    GU_HIP(hipStreamSynchronize(h->stream));
    return GU_OK;
}
