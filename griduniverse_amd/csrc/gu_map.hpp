// gu_map.hpp -- device helpers shared by the step / rollout / look-ahead kernels: the per-cell record map (global,
// block-shared LDS or private LDS), the action -> delta LUT and the branch-free move.
#pragma once
#include "gu_internal.hpp"
#include "gu_rng.hpp"

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
    if (gs.n_grids <= 1) return 0u;
    const int64_t first = (int64_t)blockIdx.x * blockDim.x + (gs.per_wave ? (int64_t)(threadIdx.x & ~63u) : 0);  // the workgroup's / the wave's first env
    const int64_t grid = first / gs.group;
    return (uint32_t)(grid < gs.n_grids ? grid : gs.n_grids - 1);  // (a wave past the batch stages the last grid: its lanes leave right after)
}

template <bool LDS>
__device__ __forceinline__ CellMap gu_stage_map(const uint8_t *__restrict__ g, int32_t cell_bytes, uint8_t *smem, const GridSel &gs,
                                                int planes = 2)
{
    if (LDS) {
        g += (int64_t)gu_block_grid(gs) * gs.grid_stride;
        if (gs.per_wave) {  // every wave its own grid, in its own region of the workgroup's LDS
            smem += (threadIdx.x >> 6) * (uint32_t)(planes * cell_bytes);
            for (int32_t i = (threadIdx.x & 63) * 16; i < planes * cell_bytes; i += 64 * 16)
                *reinterpret_cast<uint4 *>(smem + i) = *reinterpret_cast<const uint4 *>(g + i);
        } else {
            for (int32_t i = threadIdx.x * 16; i < planes * cell_bytes; i += blockDim.x * 16)
                *reinterpret_cast<uint4 *>(smem + i) = *reinterpret_cast<const uint4 *>(g + i);
        }
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

// reward code in bits 5 (RPLUS) and 6 (RMINUS) of the record -> -1, +10, -10, -10 (lava over goal, env:86-88): one byte of a constant
__device__ __forceinline__ int32_t gu_reward_packed(uint32_t flags)
{
    return __builtin_amdgcn_sbfe((int32_t)0xF6F60AFFu, (flags >> 2) & 24u, 8);
}

__device__ __forceinline__ int32_t gu_move(int32_t s, uint32_t flags, uint32_t a, int32_t delta)
{
    return __mul24((int32_t)__builtin_amdgcn_ubfe(flags, a, 1), delta) + s;  // v_bfe_u32 + v_mad_i32_i24
}
