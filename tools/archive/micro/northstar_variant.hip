// northstar_variant.hip -- A/B of two LDS layouts for the step transition (tuning evidence, not product code).
//
//   rows  : the layout sketched in BASELINE.json's north_star -- per-row wall / goal / lava bitmasks and the
//           action -> (dx, dy) LUT staged in LDS; the lane tracks (x, y).
//   cells : the layout the product uses (csrc/gu_kernels.hip) -- one flags byte + one reward byte per cell
//           compiled from the same row bitmasks; delta LUT in a 64-bit scalar.
// Both run the same 65 536 envs x 1000 steps on the same 32x32 grid with the same action bits and must
// produce identical (obs, reward, done) checksums.  Each is timed with and without the trajectory writes,
// interleaved in one process (cdna_hip_programming.md 5.4 rule 24).
//   hipcc --offload-arch=gfx950 -O3 -o northstar_variant northstar_variant.hip && ./northstar_variant
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int W = 32, H = 32, S = W * H;

__device__ __forceinline__ uint32_t mix(uint32_t h)
{
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

struct Args {
    const uint32_t *wall_rows, *goal_rows, *lava_rows;  // [H]
    const uint8_t *cells;                               // [S flags | S reward]
    int *obs, *rew, *don;                               // [T][N] or null
    unsigned long long *checksum;
    int N, T, start;
};

template <bool TRAJ>
__global__ void __launch_bounds__(256) k_rows(const Args a)
{
    __shared__ uint32_t wr[H], gr[H], lr[H];
    __shared__ int2 lut[4];
    if (threadIdx.x < H) { wr[threadIdx.x] = a.wall_rows[threadIdx.x]; gr[threadIdx.x] = a.goal_rows[threadIdx.x]; lr[threadIdx.x] = a.lava_rows[threadIdx.x]; }
    if (threadIdx.x == 0) { lut[0] = make_int2(0, -1); lut[1] = make_int2(1, 0); lut[2] = make_int2(0, 1); lut[3] = make_int2(-1, 0); }
    __syncthreads();
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    int x = a.start % W, y = a.start / W;
    bool term = ((gr[y] | lr[y]) >> x) & 1u;
    unsigned long long sum = 0;
    uint32_t word = 0;
    for (int t = 0; t < a.T; ++t) {
        if ((t & 15) == 0) word = mix(e * 0x9E3779B9u + (uint32_t)(t >> 4));
        const uint32_t act = (word >> (2 * (t & 15))) & 3u;
        const int2 d = lut[act];                                  // LDS read 1 (independent of the env state)
        const int nx = x + d.x, ny = y + d.y;
        const bool inside = (unsigned)nx < (unsigned)W && (unsigned)ny < (unsigned)H;
        const bool wall = inside ? (wr[inside ? ny : y] >> nx) & 1u : true;   // LDS read 2 (candidate's wall row)
        if (!term && !wall) { x = nx; y = ny; }
        const uint32_t g = gr[y], l = lr[y];                      // LDS reads 3, 4 (landing row)
        const bool lava = (l >> x) & 1u, goal = (g >> x) & 1u;
        term = lava | goal;
        const int s = y * W + x, r = lava ? -10 : (goal ? 10 : -1);
        sum += (unsigned)(s * 31 + r * 7 + (int)term) * (unsigned)(t + 1);
        if (TRAJ) { const size_t o = (size_t)t * a.N + e; a.obs[o] = s; a.rew[o] = r; a.don[o] = term; }
    }
    atomicAdd(a.checksum, sum);
}

template <bool TRAJ>
__global__ void __launch_bounds__(256) k_cells(const Args a)
{
    __shared__ __attribute__((aligned(16))) uint8_t cell[2 * S];
    for (int i = threadIdx.x * 16; i < 2 * S; i += blockDim.x * 16) *(uint4 *)(cell + i) = *(const uint4 *)(a.cells + i);
    __syncthreads();
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t lut = (uint64_t)(uint16_t)(int16_t)(-W) | (1ull << 16) | ((uint64_t)(uint16_t)W << 32) | (0xFFFFull << 48);
    int s = a.start;
    uint32_t flags = cell[s];
    unsigned long long sum = 0;
    uint32_t word = 0;
    for (int t = 0; t < a.T; ++t) {
        if ((t & 15) == 0) word = mix(e * 0x9E3779B9u + (uint32_t)(t >> 4));
        const uint32_t act = (word >> (2 * (t & 15))) & 3u;
        const int delta = __builtin_amdgcn_sbfe((int)(uint32_t)(lut >> (act << 4)), 0, 16);
        s = __mul24((int)__builtin_amdgcn_ubfe(flags, act, 1), delta) + s;
        flags = cell[s];                                          // LDS read 1
        const int r = (int8_t)cell[S + s];                        // LDS read 2 (off the dependent chain)
        const int term = (flags >> 4) & 1;
        sum += (unsigned)(s * 31 + r * 7 + term) * (unsigned)(t + 1);
        if (TRAJ) { const size_t o = (size_t)t * a.N + e; a.obs[o] = s; a.rew[o] = r; a.don[o] = term; }
    }
    atomicAdd(a.checksum, sum);
}

// cells16: one interleaved 16-bit record per cell (flags | reward << 8), position kept as a byte offset (2 s):
// ONE ds_read_u16 per step instead of two byte reads.
template <bool TRAJ>
__global__ void __launch_bounds__(256) k_cells16(const Args a)
{
    __shared__ __attribute__((aligned(16))) uint16_t rec[S];
    for (int i = threadIdx.x; i < S; i += blockDim.x) rec[i] = (uint16_t)(a.cells[i] | ((uint16_t)a.cells[S + i] << 8));
    __syncthreads();
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t lut = (uint64_t)(uint16_t)(int16_t)(-2 * W) | (2ull << 16) | ((uint64_t)(uint16_t)(2 * W) << 32) | ((uint64_t)(uint16_t)(int16_t)(-2) << 48);
    int p = 2 * a.start;
    const char *base = (const char *)rec;
    uint32_t r16 = *(const uint16_t *)(base + p);
    unsigned long long sum = 0;
    uint32_t word = 0;
    for (int t = 0; t < a.T; ++t) {
        if ((t & 15) == 0) word = mix(e * 0x9E3779B9u + (uint32_t)(t >> 4));
        const uint32_t act = (word >> (2 * (t & 15))) & 3u;
        const int delta = __builtin_amdgcn_sbfe((int)(uint32_t)(lut >> (act << 4)), 0, 16);
        p = __mul24((int)__builtin_amdgcn_ubfe(r16, act, 1), delta) + p;
        r16 = *(const uint16_t *)(base + p);                       // the only LDS read of the step
        const int r = __builtin_amdgcn_sbfe((int)r16, 8, 8), term = (r16 >> 4) & 1, s = p >> 1;
        sum += (unsigned)(s * 31 + r * 7 + term) * (unsigned)(t + 1);
        if (TRAJ) { const size_t o = (size_t)t * a.N + e; a.obs[o] = s; a.rew[o] = r; a.don[o] = term; }
    }
    atomicAdd(a.checksum, sum);
}

// cells_ws: warp-specialised variant of `cells + trajectory`.  A 512-thread block = 4 compute waves (256 envs) + 4
// store waves.  The compute waves run the transition chain and drop (obs, reward, done) into an LDS ring of two halves
// of CH steps each; the store waves drain the other half to HBM.  One __syncthreads per CH steps hands the halves over.
// Purpose: keep the dependent LDS chain of the compute waves free of store-issue stalls at one env-wave per SIMD.
constexpr int CH = 16;
__global__ void __launch_bounds__(512) k_cells_ws(const Args a)
{
    __shared__ __attribute__((aligned(16))) uint8_t cell[2 * S];
    __shared__ int ring[2][CH][3][256];  // 96 KB
    for (int i = threadIdx.x * 16; i < 2 * S; i += blockDim.x * 16) *(uint4 *)(cell + i) = *(const uint4 *)(a.cells + i);
    __syncthreads();
    const bool producer = threadIdx.x < 256;
    const unsigned lane = threadIdx.x & 255;
    const unsigned e = blockIdx.x * 256 + lane;
    const uint64_t lut = (uint64_t)(uint16_t)(int16_t)(-W) | (1ull << 16) | ((uint64_t)(uint16_t)W << 32) | (0xFFFFull << 48);
    int s = a.start;
    uint32_t flags = cell[s];
    unsigned long long sum = 0;
    uint32_t word = 0;
    const int chunks = a.T / CH;  // T is a multiple of CH in this experiment
    for (int c = 0; c <= chunks; ++c) {
        if (producer) {
            if (c < chunks) {
                for (int j = 0; j < CH; ++j) {
                    const int t = c * CH + j;
                    if ((t & 15) == 0) word = mix(e * 0x9E3779B9u + (uint32_t)(t >> 4));
                    const uint32_t act = (word >> (2 * (t & 15))) & 3u;
                    const int delta = __builtin_amdgcn_sbfe((int)(uint32_t)(lut >> (act << 4)), 0, 16);
                    s = __mul24((int)__builtin_amdgcn_ubfe(flags, act, 1), delta) + s;
                    flags = cell[s];
                    const int r = (int8_t)cell[S + s];
                    const int term = (flags >> 4) & 1;
                    sum += (unsigned)(s * 31 + r * 7 + term) * (unsigned)(t + 1);
                    ring[c & 1][j][0][lane] = s;
                    ring[c & 1][j][1][lane] = r;
                    ring[c & 1][j][2][lane] = term;
                }
            }
        } else if (c > 0) {
            for (int j = 0; j < CH; ++j) {
                const size_t o = (size_t)((c - 1) * CH + j) * a.N + e;
                a.obs[o] = ring[(c - 1) & 1][j][0][lane];
                a.rew[o] = ring[(c - 1) & 1][j][1][lane];
                a.don[o] = ring[(c - 1) & 1][j][2][lane];
            }
        }
        __syncthreads();
    }
    if (producer) atomicAdd(a.checksum, sum);
}

// cells_twin: every env is computed by TWO lanes in different waves of a 512-thread block (the transition is
// deterministic and cheap), and each twin stores only the steps of its parity: the same bytes leave the chip, but from 8
// storing waves per CU instead of 4 -- more stores in flight per CU, fewer store-issue stalls per wave.
__global__ void __launch_bounds__(512) k_cells_twin(const Args a)
{
    __shared__ __attribute__((aligned(16))) uint8_t cell[2 * S];
    for (int i = threadIdx.x * 16; i < 2 * S; i += blockDim.x * 16) *(uint4 *)(cell + i) = *(const uint4 *)(a.cells + i);
    __syncthreads();
    const unsigned parity = threadIdx.x >> 8;
    const unsigned e = blockIdx.x * 256 + (threadIdx.x & 255);
    const uint64_t lut = (uint64_t)(uint16_t)(int16_t)(-W) | (1ull << 16) | ((uint64_t)(uint16_t)W << 32) | (0xFFFFull << 48);
    int s = a.start;
    uint32_t flags = cell[s];
    unsigned long long sum = 0;
    uint32_t word = 0;
    for (int t = 0; t < a.T; t += 2) {
        if ((t & 15) == 0) word = mix(e * 0x9E3779B9u + (uint32_t)(t >> 4));
        int so[2], ro[2], to[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const uint32_t act = (word >> (2 * ((t + j) & 15))) & 3u;
            const int delta = __builtin_amdgcn_sbfe((int)(uint32_t)(lut >> (act << 4)), 0, 16);
            s = __mul24((int)__builtin_amdgcn_ubfe(flags, act, 1), delta) + s;
            flags = cell[s];
            const int r = (int8_t)cell[S + s];
            const int term = (flags >> 4) & 1;
            sum += (unsigned)(s * 31 + r * 7 + term) * (unsigned)(t + j + 1);
            so[j] = s, ro[j] = r, to[j] = term;
        }
        const size_t o = (size_t)(t + parity) * a.N + e;  // this twin's step of the pair
        a.obs[o] = parity ? so[1] : so[0];
        a.rew[o] = parity ? ro[1] : ro[0];
        a.don[o] = parity ? to[1] : to[0];
    }
    if (!parity) atomicAdd(a.checksum, sum);
}

int main()
{
    const int N = 65536, T = 1008, reps = 20;  // T a multiple of 16 (cells_ws chunks)
    // grid: pseudo-random walls (25 %), a lava column, goal in the far corner; same planes feed both layouts
    std::vector<uint32_t> wall(H, 0), goal(H, 0), lava(H, 0);
    uint32_t h = 12345;
    for (int s = 1; s < S - 1; ++s) { h = h * 1664525u + 1013904223u; if ((h >> 24) < 64) wall[s / W] |= 1u << (s % W); }
    for (int r = 4; r < 28; ++r) { lava[r] |= 1u << 16; wall[r] &= ~(1u << 16); }
    goal[H - 1] |= 1u << (W - 1);
    std::vector<uint8_t> cells(2 * S);
    auto bit = [&](const std::vector<uint32_t> &p, int x, int y) { return (p[y] >> x) & 1u; };
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
        const bool term = bit(goal, x, y) | bit(lava, x, y);
        uint8_t open = 0;
        if (y > 0 && !bit(wall, x, y - 1)) open |= 1;
        if (x < W - 1 && !bit(wall, x + 1, y)) open |= 2;
        if (y < H - 1 && !bit(wall, x, y + 1)) open |= 4;
        if (x > 0 && !bit(wall, x - 1, y)) open |= 8;
        cells[y * W + x] = (term ? 0 : open) | (term ? 16 : 0);
        cells[S + y * W + x] = (uint8_t)(int8_t)(bit(lava, x, y) ? -10 : bit(goal, x, y) ? 10 : -1);
    }
    Args a{};
    uint32_t *dw, *dg, *dl; uint8_t *dc; unsigned long long *dsum;
    CK(hipMalloc(&dw, H * 4)); CK(hipMalloc(&dg, H * 4)); CK(hipMalloc(&dl, H * 4)); CK(hipMalloc(&dc, 2 * S)); CK(hipMalloc(&dsum, 8));
    CK(hipMemcpy(dw, wall.data(), H * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dg, goal.data(), H * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dl, lava.data(), H * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dc, cells.data(), 2 * S, hipMemcpyHostToDevice));
    const size_t bytes = (size_t)N * T * 4;
    CK(hipMalloc(&a.obs, bytes)); CK(hipMalloc(&a.rew, bytes)); CK(hipMalloc(&a.don, bytes));
    a.wall_rows = dw; a.goal_rows = dg; a.lava_rows = dl; a.cells = dc; a.checksum = dsum; a.N = N; a.T = T; a.start = 0;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](int which) {
        dim3 g(N / 256), b(256);
        switch (which) {
        case 0: k_rows<true><<<g, b>>>(a); break;
        case 1: k_cells<true><<<g, b>>>(a); break;
        case 2: k_rows<false><<<g, b>>>(a); break;
        case 3: k_cells<false><<<g, b>>>(a); break;
        case 4: k_cells16<true><<<g, b>>>(a); break;
        case 5: k_cells16<false><<<g, b>>>(a); break;
        case 6: k_cells_ws<<<g, dim3(512)>>>(a); break;
        default: k_cells_twin<<<g, dim3(512)>>>(a); break;
        }
    };
    const char *names[8] = {"rows  + trajectory", "cells + trajectory", "rows  , no trajectory", "cells , no trajectory",
                            "cells16 + trajectory", "cells16, no trajectory", "cells_ws + trajectory", "cells_twin + trajectory"};
    unsigned long long sums[8];
    for (int w = 0; w < 8; ++w) {
        CK(hipMemset(dsum, 0, 8)); run(w); CK(hipDeviceSynchronize());
        CK(hipMemcpy(&sums[w], dsum, 8, hipMemcpyDeviceToHost));
    }
    printf("checksums: rows %llu cells %llu -> %s\n", sums[0], sums[1], (sums[0] == sums[1] && sums[2] == sums[3] && sums[0] == sums[2] && sums[4] == sums[0] && sums[5] == sums[0] && sums[6] == sums[0] && sums[7] == sums[0]) ? "IDENTICAL" : "DIFFERENT");
    for (int round = 0; round < 3; ++round)
        for (int w = 0; w < 8; ++w) {
            run(w);
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) run(w);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("round %d  %-22s %8.1f us/launch  %.3e env-steps/s\n", round, names[w], ms / reps * 1e3, (double)N * T / (ms / reps * 1e-3));
        }
    return 0;
}
