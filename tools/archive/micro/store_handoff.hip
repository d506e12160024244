// store_handoff.hip -- should the rollout's rows leave the CU through ONE wave?  (tuning evidence, not product code; round 3)
//
// write_ceiling.hip: the bare three-plane store stream of the bench launch (786 MB) runs at 6.0 .. 7.0 TB/s when 1024 waves put it
// down with dword stores (the product's shape, rate-limited), and at 7.2 .. 7.4 TB/s on every buffer when 256 waves -- one per CU --
// put it down with dwordx4 stores (1 KB per instruction), rate-limited.  A lane cannot own four environments (the step would be
// issue-bound), but the four compute waves of a workgroup can hand their rows to a fifth wave through LDS, which then is the
// CU's only store stream -- and its only rate limiter, since everything is in lockstep with it through the workgroup barrier.
// This file measures that on the product-shaped step (per-cell byte planes in LDS, uniform RNG actions; store_pacing.hip's):
//   tail        : the product's form -- every lane stores its own three dwords per step, idle turns every 4 steps
//   handoff<CH> : 4 compute waves + 1 store wave per workgroup; the compute waves write CH steps x 3 values per lane into one half
//                 of a double-buffered LDS block, ONE barrier per CH steps, the store wave reads the other half as dwordx4 and
//                 stores 3 x CH rows of 1 KB, then idles (pace per CH steps)
//   rotate<CH>  : no fifth wave: after the barrier wave (chunk mod 4) does the storing
//   bare x4     : write_ceiling's 256-wave stream, for the ceiling on the same buffers
// Every variant must leave the same trajectory bytes as `tail` (compared on the device) and the same checksum.
//   hipcc --offload-arch=gfx950 -O3 -o store_handoff store_handoff.hip && ./store_handoff [buffers]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <functional>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int W = 32, H = 32, S = W * H;
typedef int int4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix(uint32_t h)
{
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}

__device__ __forceinline__ void idle(uint32_t pace)  // busy turns | sleeping turns << 8 (csrc/gu_rollout.hpp: gu_idle)
{
    uint32_t c;
    asm volatile("s_and_b32 %0, %1, 0xff\n s_cmp_eq_u32 %0, 0\n s_cbranch_scc1 2f\n 1:\n s_sub_u32 %0, %0, 1\n s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 1b\n 2:\n"
                 "s_lshr_b32 %0, %1, 8\n s_cmp_eq_u32 %0, 0\n s_cbranch_scc1 4f\n 3:\n s_sleep 1\n s_sub_u32 %0, %0, 1\n s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 3b\n 4:"
                 : "=&s"(c) : "s"(pace) : "scc", "memory");
}

static uint32_t pace_word(int turns)
{
    if (turns <= 0) return 0u;
    if (turns <= 15) return (uint32_t)turns;
    return (uint32_t)(turns % 3) | (uint32_t)((turns / 3) << 8);
}

// LDS hand-over: wait for this wave's LDS traffic only (NOT for its global stores: __syncthreads() would wait for vmcnt(0), i.e.
// for the acknowledgement of every store in flight), then the workgroup barrier
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier" ::: "memory"); }

struct Args {
    const uint8_t *cells;  // [S flags | S reward]
    int *obs, *rew, *don;  // [T][N]
    unsigned long long *checksum;
    int N, T, start;
    uint32_t pace;
};

struct Env {
    int s;
    uint32_t flags, word;
    unsigned long long sum;
};

__device__ __forceinline__ void env_step(Env &v, const uint8_t *cell, unsigned e, int t, int &r, int &term)
{
    const uint64_t lut = (uint64_t)(uint16_t)(int16_t)(-W) | (1ull << 16) | ((uint64_t)(uint16_t)W << 32) | (0xFFFFull << 48);
    if ((t & 15) == 0) v.word = mix(e * 0x9E3779B9u + (uint32_t)(t >> 4));
    const uint32_t act = (v.word >> (2 * (t & 15))) & 3u;
    const int delta = __builtin_amdgcn_sbfe((int)(uint32_t)(lut >> (act << 4)), 0, 16);
    v.s = __mul24((int)__builtin_amdgcn_ubfe(v.flags, act, 1), delta) + v.s;
    v.flags = cell[v.s];
    r = (int8_t)cell[S + v.s];
    term = (v.flags >> 4) & 1;
    v.sum += (unsigned)(v.s * 31 + r * 7 + term) * (unsigned)(t + 1);
}

__device__ __forceinline__ void load_cells(uint8_t *cell, const uint8_t *src, int threads)
{
    for (int i = threadIdx.x * 16; i < 2 * S; i += threads * 16) *(uint4 *)(cell + i) = *(const uint4 *)(src + i);
    __syncthreads();
}

__global__ void __launch_bounds__(256) k_tail(const Args a)
{
    __shared__ __attribute__((aligned(16))) uint8_t cell[2 * S];
    load_cells(cell, a.cells, 256);
    const unsigned e = blockIdx.x * 256 + threadIdx.x;
    Env v{a.start, cell[a.start], 0u, 0ull};
    for (int t = 0; t < a.T; ++t) {
        int r, term;
        env_step(v, cell, e, t, r, term);
        const size_t o = (size_t)t * a.N + e;
        a.obs[o] = v.s;
        a.rew[o] = r;
        a.don[o] = term;
        if ((t & 3) == 3 && a.pace) idle(a.pace);
    }
    atomicAdd(a.checksum, v.sum);
}

// tailv<EPL, NT, SYNC>: the product's form with EPL adjacent environments per lane (one dword / x2 / x4 store per plane and step;
// 1024 / 512 / 256 waves), non-temporal stores, a workgroup barrier at the idle point (the waves of a CU stay on the same rows)
template <int EPL> struct VecOf;
template <> struct VecOf<1> { typedef int type; };
template <> struct VecOf<2> { typedef int type __attribute__((ext_vector_type(2))); };
template <> struct VecOf<4> { typedef int type __attribute__((ext_vector_type(4))); };

template <int EPL, bool NT, bool SYNC>
__global__ void __launch_bounds__(256) k_tailv(const Args a)
{
    typedef typename VecOf<EPL>::type V;
    __shared__ __attribute__((aligned(16))) uint8_t cell[2 * S];
    load_cells(cell, a.cells, blockDim.x);
    const unsigned e0 = (blockIdx.x * blockDim.x + threadIdx.x) * EPL;
    Env v[EPL];
#pragma unroll
    for (int k = 0; k < EPL; ++k) v[k] = Env{a.start, cell[a.start], 0u, 0ull};
    for (int t = 0; t < a.T; ++t) {
        int s[EPL], r[EPL], term[EPL];
#pragma unroll
        for (int k = 0; k < EPL; ++k) {
            env_step(v[k], cell, e0 + k, t, r[k], term[k]);
            s[k] = v[k].s;
        }
        const size_t o = (size_t)t * a.N + e0;
        V x, y, z;
        if constexpr (EPL == 1) x = s[0], y = r[0], z = term[0];
        else {
#pragma unroll
            for (int k = 0; k < EPL; ++k) x[k] = s[k], y[k] = r[k], z[k] = term[k];
        }
        if (NT) {
            __builtin_nontemporal_store(x, (V *)(a.obs + o));
            __builtin_nontemporal_store(y, (V *)(a.rew + o));
            __builtin_nontemporal_store(z, (V *)(a.don + o));
        } else {
            *(V *)(a.obs + o) = x;
            *(V *)(a.rew + o) = y;
            *(V *)(a.don + o) = z;
        }
        if ((t & 3) == 3) {
            if (SYNC) asm volatile("s_barrier" ::: "memory");
            if (a.pace) idle(a.pace);
        }
    }
    unsigned long long sum = 0;
#pragma unroll
    for (int k = 0; k < EPL; ++k) sum += v[k].sum;
    atomicAdd(a.checksum, sum);
}

// MODE 0: a fifth wave stores (320 threads); MODE 1: the compute waves take turns (256 threads)
template <int CH, int MODE, bool NT, bool DRY = false>
__global__ void __launch_bounds__(320) k_handoff(const Args a)
{
    __shared__ __attribute__((aligned(16))) uint8_t cell[2 * S];
    __shared__ __attribute__((aligned(16))) int hand[2][CH][3][256];
    load_cells(cell, a.cells, MODE == 0 ? 320 : 256);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int chunks = a.T / CH;
    auto put_down = [&](int c) {  // chunk c: CH rows x 3 planes of this workgroup's 256 columns, 1 KB per store
        const size_t col = (size_t)blockIdx.x * 256 + lane * 4;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const size_t o = (size_t)(DRY ? j : c * CH + j) * a.N + col;  // DRY: the same rows over and over (no HBM traffic to speak of)
            const int4v x = *(const int4v *)&hand[c & 1][j][0][lane * 4], y = *(const int4v *)&hand[c & 1][j][1][lane * 4],
                        z = *(const int4v *)&hand[c & 1][j][2][lane * 4];
            if (NT) {
                __builtin_nontemporal_store(x, (int4v *)(a.obs + o));
                __builtin_nontemporal_store(y, (int4v *)(a.rew + o));
                __builtin_nontemporal_store(z, (int4v *)(a.don + o));
            } else {
                *(int4v *)(a.obs + o) = x;
                *(int4v *)(a.rew + o) = y;
                *(int4v *)(a.don + o) = z;
            }
        }
    };
    if (MODE == 0 && wave == 4) {  // the store wave: one chunk behind the compute waves
        for (int c = 0; c < chunks; ++c) {
            lds_barrier();  // chunk c is in hand[c & 1]
            put_down(c);
            if (a.pace) idle(a.pace);
        }
        return;
    }
    const unsigned e = blockIdx.x * 256 + threadIdx.x;
    Env v{a.start, cell[a.start], 0u, 0ull};
    for (int c = 0; c < chunks; ++c) {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            int r, term;
            env_step(v, cell, e, c * CH + j, r, term);
            hand[c & 1][j][0][threadIdx.x] = v.s;
            hand[c & 1][j][1][threadIdx.x] = r;
            hand[c & 1][j][2][threadIdx.x] = term;
        }
        lds_barrier();
        if (MODE == 1 && wave == (c & 3)) {
            put_down(c);
            if (a.pace) idle(a.pace);
        }
    }
    atomicAdd(a.checksum, v.sum);
}

template <bool NT>
__global__ void __launch_bounds__(64) k_bare_x4(const Args a)
{
    const size_t col = ((size_t)blockIdx.x * 64 + threadIdx.x) * 4;
    size_t o = col;
    for (int t = 0; t < a.T; ++t, o += a.N) {
        const int4v v = {t, t + 1, t + 2, t + 3};
        if (NT) {
            __builtin_nontemporal_store(v, (int4v *)(a.obs + o));
            __builtin_nontemporal_store(v, (int4v *)(a.rew + o));
            __builtin_nontemporal_store(v, (int4v *)(a.don + o));
        } else {
            *(int4v *)(a.obs + o) = v;
            *(int4v *)(a.rew + o) = v;
            *(int4v *)(a.don + o) = v;
        }
        if ((t & 3) == 3 && a.pace) idle(a.pace);
    }
}

__global__ void k_compare(const int *x, const int *y, size_t n, unsigned long long *diff)
{
    unsigned long long d = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d += x[i] != y[i];
    if (d) atomicAdd(diff, d);
}

int main(int argc, char **argv)
{
    const int N = 65536, T = 1000;
    const int buffers = argc > 1 ? atoi(argv[1]) : 6;
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
    CK(hipMalloc(&dc, 2 * S)); CK(hipMalloc(&dsum, 16));
    CK(hipMemcpy(dc, cells.data(), 2 * S, hipMemcpyHostToDevice));
    const size_t plane = (size_t)N * T;
    std::vector<int *> bufs;
    for (int b = 0; b < buffers + 1; ++b) {
        int *p = nullptr;
        if (hipMalloc(&p, 3 * plane * 4) != hipSuccess) break;
        bufs.push_back(p);
    }
    int *ref = bufs.back();  // `tail`'s rows, for the comparison
    bufs.pop_back();
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    Args a{};
    a.cells = dc; a.checksum = dsum; a.N = N; a.T = T; a.start = 0;
    struct Variant { const char *name; bool rollout; int every; std::function<void()> launch; };
    const std::vector<Variant> v = {
        {"tail (product form)", true, 4, [&] { k_tail<<<dim3(N / 256), dim3(256)>>>(a); }},
        {"tail nt", true, 4, [&] { k_tailv<1, true, false><<<dim3(N / 256), dim3(256)>>>(a); }},
        {"tail + barrier", true, 4, [&] { k_tailv<1, false, true><<<dim3(N / 256), dim3(256)>>>(a); }},
        {"tail nt + barrier", true, 4, [&] { k_tailv<1, true, true><<<dim3(N / 256), dim3(256)>>>(a); }},
        {"2 envs/lane, 512 waves", true, 4, [&] { k_tailv<2, false, false><<<dim3(N / 256), dim3(128)>>>(a); }},
        {"2 envs/lane nt", true, 4, [&] { k_tailv<2, true, false><<<dim3(N / 256), dim3(128)>>>(a); }},
        {"2 envs/lane + barrier", true, 4, [&] { k_tailv<2, false, true><<<dim3(N / 256), dim3(128)>>>(a); }},
        {"4 envs/lane, 256 waves", true, 4, [&] { k_tailv<4, false, false><<<dim3(N / 256), dim3(64)>>>(a); }},
        {"4 envs/lane nt", true, 4, [&] { k_tailv<4, true, false><<<dim3(N / 256), dim3(64)>>>(a); }},
        {"handoff<2> 5th wave", true, 2, [&] { k_handoff<2, 0, false><<<dim3(N / 256), dim3(320)>>>(a); }},
        {"handoff<4> 5th wave", true, 4, [&] { k_handoff<4, 0, false><<<dim3(N / 256), dim3(320)>>>(a); }},
        {"handoff<4> 5th wave nt", true, 4, [&] { k_handoff<4, 0, true><<<dim3(N / 256), dim3(320)>>>(a); }},
        {"handoff<8> 5th wave", true, 8, [&] { k_handoff<8, 0, false><<<dim3(N / 256), dim3(320)>>>(a); }},
        {"handoff<2> dry", false, 2, [&] { k_handoff<2, 0, false, true><<<dim3(N / 256), dim3(320)>>>(a); }},
        {"handoff<4> dry", false, 4, [&] { k_handoff<4, 0, false, true><<<dim3(N / 256), dim3(320)>>>(a); }},
        {"handoff<8> dry", false, 8, [&] { k_handoff<8, 0, false, true><<<dim3(N / 256), dim3(320)>>>(a); }},
        {"rotate<4> dry", false, 4, [&] { k_handoff<4, 1, false, true><<<dim3(N / 256), dim3(256)>>>(a); }},
        {"rotate<4>", true, 4, [&] { k_handoff<4, 1, false><<<dim3(N / 256), dim3(256)>>>(a); }},
        {"bare x4, 256 waves", false, 4, [&] { k_bare_x4<false><<<dim3(N / 256), dim3(64)>>>(a); }},
        {"bare x4 nt, 256 waves", false, 4, [&] { k_bare_x4<true><<<dim3(N / 256), dim3(64)>>>(a); }},
    };
    auto point = [&](int *buf) { a.obs = buf; a.rew = buf + plane; a.don = buf + 2 * plane; };
    // correctness: same checksum, same bytes as `tail`
    unsigned long long want = 0;
    for (size_t w = 0; w < v.size(); ++w) {
        if (!v[w].rollout) continue;
        point(w == 0 ? ref : bufs[0]);
        a.pace = w == 0 ? 0u : 3u;
        CK(hipMemset(dsum, 0, 16));
        if (w) CK(hipMemset(bufs[0], 0xff, 3 * plane * 4));
        v[w].launch();
        CK(hipDeviceSynchronize());
        if (w) k_compare<<<dim3(2048), dim3(256)>>>(ref, bufs[0], 3 * plane, dsum + 1);
        unsigned long long got[2];
        CK(hipMemcpy(got, dsum, 16, hipMemcpyDeviceToHost));
        if (w == 0) want = got[0];
        if (got[0] != want || got[1]) { printf("%s: checksum %llu (want %llu), %llu words differ\n", v[w].name, got[0], want, got[1]); return 1; }
    }
    printf("every rollout variant leaves tail's bytes and checksum (%llu); %zu buffers of %.0f MB\n", want, bufs.size(), 3 * plane * 4 / 1e6);
    printf("per variant and buffer: unpaced us -> best us of the scan / the same amount over 8 more launches @ idle turns per hand-over (TB/s)\n");
    auto timed = [&](const Variant &x, int turns, int reps) {
        a.pace = pace_word(turns);
        x.launch();
        x.launch();
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) x.launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms / reps * 1e3f;
    };
    point(bufs[0]);
    for (int i = 0; i < 300; ++i) v[0].launch();  // working clocks
    CK(hipDeviceSynchronize());
    for (const Variant &x : v) {
        printf("%-24s", x.name);
        for (int *buf : bufs) {
            point(buf);
            const float t0 = timed(x, 0, 4);
            float best = t0;
            int at = 0;
            for (int turns = 10 * x.every; turns >= 1; --turns) {  // down from the healthy side
                const float t = timed(x, turns, 3);
                if (t < best) best = t, at = turns;
            }
            const float again = at ? timed(x, at, 8) : t0;  // (near the cliff the stream is bistable: the scan's minimum and 8 more launches)
            printf(" | %6.1f -> %6.1f / %6.1f @%3d (%.2f)", t0, best, again, at, 3 * plane * 4 / (std::min(best, again) * 1e-6) / 1e12);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
