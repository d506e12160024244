// store_pacing.hip -- does the PACE at which a wave hands its three trajectory stores to the memory system matter?  (tuning
// evidence, not product code; round 3)
//
// Round 2 found (profiles/archive/r02g_store_sleep_spacing.txt) that the bare three-store loop runs at 113..117 us per 65 536 x 1000
// launch on buffers of the SLOW write-rate class (134..136 us unpaced) once an `s_sleep 1` separates the three stores -- and that
// the same sleeps made the rollout kernel slower.  This file measures, per buffer of one process (so that both classes show up):
//   bare<G>     : the bare loop, the three stores of a step separated by G idle clocks (s_nop)
//   burst<G>    : the bare loop, three stores back to back and ONE gap of G clocks per step
//   cells       : the product-shaped rollout step (per-cell byte planes in LDS, uniform RNG actions), stores as the compiler
//                 places them (= back to back at the end of the step)
//   tail<G>     : cells + one idle gap of G clocks at the end of every step (a pure rate limiter)
//   paced / spread / pipe : the stores spread over the step (LDS waits or fixed gaps between them; the rows of step t - 1 during step t)
//   tailx       : tail + a workgroup barrier per step and / or the XCD-aware block order
//   timer, timer2 : a per-wave deadline on the shader clock (s_memtime), read on / off the critical path
// FINDINGS (profiles/archive/r03d_store_pacing_*.txt): only the average rate matters (burst 176 = bare g64 = 110 .. 116 us on every buffer
// against 115 .. 139 us unpaced); the window is narrow (burst 160: collapse on slow buffers again; burst 192: 118 .. 121 everywhere);
// LDS waits do not pace well; fixed gaps that idle the wave while its own chain waits are too expensive (pipe); a clock read per
// step costs more than it saves; a barrier per step works in this micro-kernel (122 .. 125 us everywhere) but not in the product's
// 16-way unrolled body.  What the product does with this: csrc/gu_rollout.hpp, gu_idle.
// All rollout variants must produce the same checksum.
//   hipcc --offload-arch=gfx950 -O3 -o store_pacing store_pacing.hip && ./store_pacing [buffers] [gaps|pipe|rate|timer|sync]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int W = 32, H = 32, S = W * H;

__device__ __forceinline__ uint32_t mix(uint32_t h)
{
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

// G idle clocks: `s_nop n` holds the wave for n + 1 issue slots of 4 clocks each (measured: s_nop 15 = 64 clocks = s_sleep 1)
template <int G>
__device__ __forceinline__ void gap()
{
    static_assert(G % 4 == 0 && G <= 128, "gap in clocks, a multiple of 4");
    if (G > 64) {
        asm volatile("s_nop 15" ::: "memory");
        asm volatile("s_nop %0" ::"n"((G - 64) / 4 - 1) : "memory");
    } else if (G > 0) {
        asm volatile("s_nop %0" ::"n"(G / 4 - 1) : "memory");
    } else {
        asm volatile("" ::: "memory");
    }
}

struct Args {
    const uint8_t *cells;  // [S flags | S reward]
    int *obs, *rew, *don;  // [T][N]
    unsigned long long *checksum;
    int N, T, start;
    int period, phase;  // k_timer: clocks per step and per wave, and whether waves start at staggered phases
};

template <int G>
__global__ void __launch_bounds__(256) k_bare(const Args a)
{
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    size_t o = e;
    for (int t = 0; t < a.T; ++t, o += a.N) {
        a.obs[o] = t;
        gap<G>();
        a.rew[o] = t;
        gap<G>();
        a.don[o] = t;
        gap<G>();
    }
}

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
        flags = cell[s];
        const int r = (int8_t)cell[S + s];
        const int term = (flags >> 4) & 1;
        sum += (unsigned)(s * 31 + r * 7 + term) * (unsigned)(t + 1);
        const size_t o = (size_t)t * a.N + e;
        a.obs[o] = s;
        a.rew[o] = r;
        a.don[o] = term;
    }
    atomicAdd(a.checksum, sum);
}

// the same step, stores spread over it.  MODE 0: obs | LDS wait | done | gap G | reward | gap G.
// MODE 1: the reward byte is read only after the flags have arrived (a second LDS round trip paces the third store):
//         obs | LDS wait | done | LDS wait | reward | gap G.
template <int G, int MODE>
__global__ void __launch_bounds__(256) k_paced(const Args a)
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
    size_t o = e;
    for (int t = 0; t < a.T; ++t, o += a.N) {
        if ((t & 15) == 0) word = mix(e * 0x9E3779B9u + (uint32_t)(t >> 4));
        const uint32_t act = (word >> (2 * (t & 15))) & 3u;
        const int delta = __builtin_amdgcn_sbfe((int)(uint32_t)(lut >> (act << 4)), 0, 16);
        s = __mul24((int)__builtin_amdgcn_ubfe(flags, act, 1), delta) + s;
        uint32_t f_new = cell[s];               // issue the flags read ...
        int r = 0;
        if (MODE == 0) r = (int8_t)cell[S + s]; // ... (and the reward read)
        asm volatile("" ::: "memory");          // (the LDS reads stay above, the stores below)
        a.obs[o] = s;                           // ... store obs while they are in flight
        asm volatile("; flags needed" ::"v"(f_new) : "memory");  // the LDS round trip separates obs from done
        flags = f_new;
        const int term = (flags >> 4) & 1;
        a.don[o] = term;
        if (MODE == 1) {
            r = (int8_t)cell[S + s];            // a second LDS round trip separates done from reward
            asm volatile("; reward needed" ::"v"(r) : "memory");
        } else {
            gap<G>();
        }
        a.rew[o] = r;
        gap<G>();
        sum += (unsigned)(s * 31 + r * 7 + term) * (unsigned)(t + 1);
    }
    atomicAdd(a.checksum, sum);
}

// pipe<G1, G2, G3>: the stores of step t - 1 are issued DURING step t, separated by fixed idle gaps (G1 after obs, G2 after reward,
// G3 after done), while the LDS round trip of step t is in flight: the chain's latency hides under the gaps, the step time is
// 3 x (store issue + gap) whatever the LDS latency happens to be -- the bare g64 loop with the transition riding along.
template <int G1, int G2, int G3>
__global__ void __launch_bounds__(256) k_pipe(const Args a)
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
    size_t o = e;
    int s_p = 0, r_p = 0, d_p = 0;  // step t - 1, not stored yet
    for (int t = 0; t < a.T; ++t) {
        if ((t & 15) == 0) word = mix(e * 0x9E3779B9u + (uint32_t)(t >> 4));
        const uint32_t act = (word >> (2 * (t & 15))) & 3u;
        const int delta = __builtin_amdgcn_sbfe((int)(uint32_t)(lut >> (act << 4)), 0, 16);
        s = __mul24((int)__builtin_amdgcn_ubfe(flags, act, 1), delta) + s;
        uint32_t f_new = cell[s];      // the LDS reads of step t are issued first ...
        int r = (int8_t)cell[S + s];
        asm volatile("" ::: "memory");
        if (t > 0) {                   // ... and the three rows of step t - 1 go out under their latency, evenly spaced
            a.obs[o] = s_p;
            gap<G1>();
            a.rew[o] = r_p;
            gap<G2>();
            a.don[o] = d_p;
            gap<G3>();
            o += a.N;
        }
        asm volatile("; step t needed" ::"v"(f_new), "v"(r) : "memory");
        flags = f_new;
        const int term = (flags >> 4) & 1;
        sum += (unsigned)(s * 31 + r * 7 + term) * (unsigned)(t + 1);
        s_p = s, r_p = r, d_p = term;
    }
    a.obs[o] = s_p;
    a.rew[o] = r_p;
    a.don[o] = d_p;
    atomicAdd(a.checksum, sum);
}

// burst<G>: the bare loop with its three stores back to back and ONE idle gap of G clocks per step (same average rate as three
// gaps of G / 3): is it the smoothness or only the RATE of the store stream that matters?
template <int G>
__global__ void __launch_bounds__(256) k_burst(const Args a)
{
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    size_t o = e;
    for (int t = 0; t < a.T; ++t, o += a.N) {
        a.obs[o] = t;
        a.rew[o] = t;
        a.don[o] = t;
#pragma unroll
        for (int i = 0; i < G / 64; ++i) gap<64>();
        gap<G % 64>();
    }
}

// tail<G>: the product-shaped step (stores where the compiler puts them) plus ONE idle gap of G clocks at the end of every step:
// a pure rate limiter.
template <int G>
__global__ void __launch_bounds__(256) k_tail(const Args a)
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
        flags = cell[s];
        const int r = (int8_t)cell[S + s];
        const int term = (flags >> 4) & 1;
        sum += (unsigned)(s * 31 + r * 7 + term) * (unsigned)(t + 1);
        const size_t o = (size_t)t * a.N + e;
        a.obs[o] = s;
        a.rew[o] = r;
        a.don[o] = term;
#pragma unroll
        for (int i = 0; i < G / 64; ++i) gap<64>();
        gap<G % 64>();
    }
    atomicAdd(a.checksum, sum);
}

// spread<P1, P2>: obs goes out as soon as the new position is known (the LDS round trip follows it), done when the flags are
// back, P1 idle clocks, reward, P2 idle clocks.
template <int P1, int P2>
__global__ void __launch_bounds__(256) k_spread(const Args a)
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
    size_t o = e;
    for (int t = 0; t < a.T; ++t, o += a.N) {
        if ((t & 15) == 0) word = mix(e * 0x9E3779B9u + (uint32_t)(t >> 4));
        const uint32_t act = (word >> (2 * (t & 15))) & 3u;
        const int delta = __builtin_amdgcn_sbfe((int)(uint32_t)(lut >> (act << 4)), 0, 16);
        s = __mul24((int)__builtin_amdgcn_ubfe(flags, act, 1), delta) + s;
        uint32_t f_new = cell[s];
        int r = (int8_t)cell[S + s];
        asm volatile("" ::: "memory");
        a.obs[o] = s;
        asm volatile("; flags needed" ::"v"(f_new), "v"(r) : "memory");
        flags = f_new;
        const int term = (flags >> 4) & 1;
        a.don[o] = term;
#pragma unroll
        for (int i = 0; i < P1 / 64; ++i) gap<64>();
        gap<P1 % 64>();
        a.rew[o] = r;
#pragma unroll
        for (int i = 0; i < P2 / 64; ++i) gap<64>();
        gap<P2 % 64>();
        sum += (unsigned)(s * 31 + r * 7 + term) * (unsigned)(t + 1);
    }
    atomicAdd(a.checksum, sum);
}

// timer: the product-shaped step, RATE-LIMITED by the clock: every wave keeps a deadline `next` (s_memtime clocks) that advances by
// a.period per step, and idles (s_nop loop, no memory traffic) until the deadline before it goes on.  A step that took longer than
// the period is not delayed; debt is not carried over.  a.phase staggers the waves' first deadlines over one period.
__global__ void __launch_bounds__(256) k_timer(const Args a)
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
    const int wave = __builtin_amdgcn_readfirstlane((int)(e >> 6));
    long long next = (long long)__builtin_readcyclecounter() + a.period + (a.phase ? (long long)(wave & 15) * a.period / 16 : 0);
    for (int t = 0; t < a.T; ++t) {
        if ((t & 15) == 0) word = mix(e * 0x9E3779B9u + (uint32_t)(t >> 4));
        const uint32_t act = (word >> (2 * (t & 15))) & 3u;
        const int delta = __builtin_amdgcn_sbfe((int)(uint32_t)(lut >> (act << 4)), 0, 16);
        s = __mul24((int)__builtin_amdgcn_ubfe(flags, act, 1), delta) + s;
        flags = cell[s];
        const int r = (int8_t)cell[S + s];
        const long long now = (long long)__builtin_readcyclecounter();  // (its latency rides with the LDS round trip: both are lgkmcnt)
        const int term = (flags >> 4) & 1;
        sum += (unsigned)(s * 31 + r * 7 + term) * (unsigned)(t + 1);
        const size_t o = (size_t)t * a.N + e;
        a.obs[o] = s;
        a.rew[o] = r;
        a.don[o] = term;
        int rem = (int)(next - now);
        if (rem < -a.period) next = now;  // fell behind by more than a step: no debt
        next += a.period;
        while (rem > 0) {
            asm volatile("s_nop 3" ::: "memory");
            rem -= 24;
        }
    }
    atomicAdd(a.checksum, sum);
}

// tailx<G, SYNC, XCD>: tail<G> with (SYNC) a workgroup barrier per step, so that the four waves of a CU put their 1 KB of a row down
// together, and / or (XCD) the XCD-aware env-block order: workgroup b works on env block (b % 8) * (blocks / 8) + b / 8, so that each
// XCD writes one contiguous eighth of every row.
template <int G, bool SYNC, bool XCD>
__global__ void __launch_bounds__(256) k_tailx(const Args a)
{
    __shared__ __attribute__((aligned(16))) uint8_t cell[2 * S];
    for (int i = threadIdx.x * 16; i < 2 * S; i += blockDim.x * 16) *(uint4 *)(cell + i) = *(const uint4 *)(a.cells + i);
    __syncthreads();
    const unsigned blk = XCD ? (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : blockIdx.x;
    const unsigned e = blk * blockDim.x + threadIdx.x;
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
        flags = cell[s];
        const int r = (int8_t)cell[S + s];
        const int term = (flags >> 4) & 1;
        sum += (unsigned)(s * 31 + r * 7 + term) * (unsigned)(t + 1);
        const size_t o = (size_t)t * a.N + e;
        if (SYNC) __builtin_amdgcn_s_barrier();
        a.obs[o] = s;
        a.rew[o] = r;
        a.don[o] = term;
#pragma unroll
        for (int i = 0; i < G / 64; ++i) gap<64>();
        gap<G % 64>();
    }
    atomicAdd(a.checksum, sum);
}

// timer2: rate limiter by deadline, the clock read taken OFF the critical path: s_memtime is issued right after the LDS results
// of step t have arrived (SMEM and LDS share lgkmcnt and return out of order, so a read in flight would stretch the next LDS wait
// only if it took longer than the next step's chain) and consumed one step later.  The idle loop is coarse (~50 clocks per turn);
// the deadline makes the AVERAGE period exact.  a.phase: deadlines aligned to global multiples of the period (all waves of the
// chip in one phase) instead of each wave's own start.
__global__ void __launch_bounds__(256) k_timer2(const Args a)
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
    long long now = (long long)__builtin_readcyclecounter();
    long long next = a.phase ? (now / a.period + 2) * a.period : now + a.period;
    for (int t = 0; t < a.T; ++t) {
        if ((t & 15) == 0) word = mix(e * 0x9E3779B9u + (uint32_t)(t >> 4));
        const uint32_t act = (word >> (2 * (t & 15))) & 3u;
        const int delta = __builtin_amdgcn_sbfe((int)(uint32_t)(lut >> (act << 4)), 0, 16);
        s = __mul24((int)__builtin_amdgcn_ubfe(flags, act, 1), delta) + s;
        uint32_t f_new = cell[s];
        int r = (int8_t)cell[S + s];
        asm volatile("; step needed" ::"v"(f_new), "v"(r), "s"(now) : "memory");  // one wait: LDS results of t, clock of t - 1
        const long long seen = now;
        now = (long long)__builtin_readcyclecounter();  // for step t + 1
        asm volatile("" ::: "memory");
        flags = f_new;
        const int term = (flags >> 4) & 1;
        sum += (unsigned)(s * 31 + r * 7 + term) * (unsigned)(t + 1);
        const size_t o = (size_t)t * a.N + e;
        a.obs[o] = s;
        a.rew[o] = r;
        a.don[o] = term;
        int rem = (int)(next - seen) - a.period;  // `seen` is one period old
        if (rem < -2 * a.period) next = seen + a.period;
        next += a.period;
        while (rem > 0) {
            asm volatile("s_nop 7" ::: "memory");
            rem -= 48;
        }
    }
    atomicAdd(a.checksum, sum);
}

int main(int argc, char **argv)
{
    const int N = 65536, T = 1000, reps = 5;
    const int buffers = argc > 1 ? atoi(argv[1]) : 10;
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
    uint8_t *dc; unsigned long long *dsum;
    CK(hipMalloc(&dc, 2 * S)); CK(hipMalloc(&dsum, 8));
    CK(hipMemcpy(dc, cells.data(), 2 * S, hipMemcpyHostToDevice));
    const size_t plane = (size_t)N * T;
    std::vector<int *> bufs;
    for (int b = 0; b < buffers; ++b) {
        int *p = nullptr;
        if (hipMalloc(&p, 3 * plane * 4) != hipSuccess) break;
        bufs.push_back(p);
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    Args a{};
    a.cells = dc; a.checksum = dsum; a.N = N; a.T = T; a.start = 0;
    // variant sets (the profiles/archive/r03d_store_pacing_*.txt files, in the order they were recorded); first_rollout: where the variants
    // that compute the rollout -- and must agree on its checksum -- begin
    struct Variant { const char *name; std::function<void(dim3, dim3)> launch; };
#define BARE(G) {"bare g" #G, [&](dim3 g, dim3 b) { k_bare<G><<<g, b>>>(a); }}
#define BURST(G) {"burst " #G, [&](dim3 g, dim3 b) { k_burst<G><<<g, b>>>(a); }}
#define TAIL(G) {"tail " #G, [&](dim3 g, dim3 b) { k_tail<G><<<g, b>>>(a); }}
#define PACED(G, M) {"paced" #M " g" #G, [&](dim3 g, dim3 b) { k_paced<G, M><<<g, b>>>(a); }}
#define PIPE(A, B, C) {"pipe " #A "/" #B "/" #C, [&](dim3 g, dim3 b) { k_pipe<A, B, C><<<g, b>>>(a); }}
#define SPREAD(A, B) {"spread " #A "/" #B, [&](dim3 g, dim3 b) { k_spread<A, B><<<g, b>>>(a); }}
#define TAILX(NAME, G, SY, XC) {NAME " " #G, [&](dim3 g, dim3 b) { k_tailx<G, SY, XC><<<g, b>>>(a); }}
#define TIMER(K, NAME, P, PH) {NAME " " #P, [&](dim3 g, dim3 b) { a.period = P; a.phase = PH; K<<<g, b>>>(a); }}
    const Variant v_cells{"cells", [&](dim3 g, dim3 b) { k_cells<<<g, b>>>(a); }};
    const std::string set = argc > 2 ? argv[2] : "rate";
    std::vector<Variant> v;
    int first_rollout = 0;
    if (set == "gaps") {          // 1: idle gaps between the three stores, LDS waits as gaps
        v = {BARE(0), BARE(16), BARE(32), BARE(48), BARE(64), BARE(96), v_cells, PACED(0, 0), PACED(16, 0), PACED(32, 0), PACED(48, 0), PACED(64, 0),
             PACED(0, 1), PACED(16, 1), PACED(32, 1), PACED(48, 1)};
        first_rollout = 6;
    } else if (set == "pipe") {   // 2: the stores of step t - 1 spread over step t with fixed gaps
        v = {BARE(0), BARE(64), v_cells, PACED(32, 1), PIPE(64, 64, 64), PIPE(64, 64, 32), PIPE(64, 64, 0), PIPE(48, 48, 48), PIPE(80, 80, 0),
             PIPE(56, 56, 56), PIPE(72, 72, 24)};
        first_rollout = 2;
    } else if (set == "rate") {   // 3: is it the smoothness or the rate?
        v = {BARE(0), BARE(64), BURST(192), BURST(176), BURST(160), BURST(128), BURST(224), v_cells, TAIL(8), TAIL(16), TAIL(24), TAIL(32), TAIL(40),
             TAIL(48), TAIL(64), TAIL(96), SPREAD(0, 0), SPREAD(32, 32), SPREAD(48, 48), SPREAD(64, 64), SPREAD(96, 0), SPREAD(0, 96)};
        first_rollout = 7;
    } else if (set == "timer") {  // 4, 6: a clock deadline per step (s_memtime), read on and off the critical path
        v = {BARE(0), BURST(176), v_cells, TAIL(32), TIMER(k_timer, "timer", 236, 0), TIMER(k_timer, "timer", 260, 0), TIMER(k_timer, "timer", 276, 0),
             TIMER(k_timer, "timer", 300, 0), TIMER(k_timer, "tmr ph", 276, 1), TIMER(k_timer2, "tmr2", 100, 0), TIMER(k_timer2, "tmr2", 264, 0),
             TIMER(k_timer2, "tmr2", 280, 0), TIMER(k_timer2, "tmr2", 296, 0), TIMER(k_timer2, "t2ph", 280, 1)};
        first_rollout = 2;
    } else {                      // 5 ("sync"): a workgroup barrier per step, the XCD-aware block order
        v = {BARE(0), BURST(176), v_cells, TAIL(16), TAIL(32), TAILX("sync", 0, true, false), TAILX("sync", 8, true, false), TAILX("sync", 16, true, false),
             TAILX("sync", 32, true, false), TAILX("xcd", 0, false, true), TAILX("xcd", 16, false, true), TAILX("xcd", 32, false, true),
             TAILX("sync+xcd", 0, true, true), TAILX("sync+xcd", 16, true, true), TAILX("sync+xcd", 32, true, true), TAIL(40), TAIL(48)};
        first_rollout = 2;
    }
    const int n_var = (int)v.size();
    std::vector<const char *> names;
    for (const Variant &x : v) names.push_back(x.name);
    auto run = [&](int w) { v[(size_t)w].launch(dim3(N / 256), dim3(256)); };
    // correctness: every rollout variant leaves the same checksum
    a.obs = bufs[0]; a.rew = bufs[0] + plane; a.don = bufs[0] + 2 * plane;
    unsigned long long want = 0;
    for (int w = first_rollout; w < n_var; ++w) {
        CK(hipMemset(dsum, 0, 8)); run(w); CK(hipDeviceSynchronize());
        unsigned long long got; CK(hipMemcpy(&got, dsum, 8, hipMemcpyDeviceToHost));
        if (w == first_rollout) want = got;
        if (got != want) { printf("checksum of %s differs\n", names[w]); return 1; }
    }
    printf("checksums identical (%llu); %zu buffers of %.0f MB; us per launch:\n%-4s", want, bufs.size(), 3 * plane * 4 / 1e6, "buf");
    for (int w = 0; w < n_var; ++w) printf(" %13s", names[w]);
    printf("\n");
    for (size_t b = 0; b < bufs.size(); ++b) {
        a.obs = bufs[b]; a.rew = bufs[b] + plane; a.don = bufs[b] + 2 * plane;
        printf("%-4zu", b);
        for (int w = 0; w < n_var; ++w) {
            run(w); run(w);
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) run(w);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf(" %13.1f", ms / reps * 1e3);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
