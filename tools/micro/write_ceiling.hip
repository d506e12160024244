// write_ceiling.hip -- what is the highest rate at which ANY rate-limited store stream writes HBM on this device?  (tuning evidence,
// not product code; round 3)
//
// Round 1's store-ceiling table (profiles/r01b_store_ceiling.txt: 5.0 .. 6.0 TB/s whatever the store width, block size or cache
// policy) was measured before round 3 found that an over-driven store stream collapses: all of its rows are
// collapsed streams.  This file repeats the question with the limiter in place.  One kernel writes 786 MB (the bench launch's
// bytes) as `planes` planes of [T][N] int32, a lane owning VEC adjacent columns (one dword / dwordx2 / dwordx4 store per plane and
// step -- a wave puts down 256 B / 512 B / 1 KB per instruction), and idles `turns` x ~33 clocks every 4 steps (the product's
// gu_idle); the host scans the idle amount per shape and buffer and prints the unpaced time, the best time and where it lies.
// Shapes: the product's (3 planes, dword, one wave per SIMD), wider stores at the same wave count, a single plane (a plain fill),
// two and four waves per SIMD, one or two waves per CU with 1 KB / 512 B stores, non-temporal stores.
//   hipcc --offload-arch=gfx950 -O3 -o write_ceiling write_ceiling.hip && ./write_ceiling [buffers]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

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

template <int VEC> struct Vec;
template <> struct Vec<1> { typedef int type; };
template <> struct Vec<2> { typedef int type __attribute__((ext_vector_type(2))); };
template <> struct Vec<4> { typedef int type __attribute__((ext_vector_type(4))); };

// ROT: the four waves of a 256-lane block take turns -- wave w puts down rows w, w + 4, ... for the block's 256 columns with one
// dwordx4 store per plane (what a kernel that hands its rows over through LDS would issue: every wave stores, a quarter as often)
template <bool NT>
__global__ void __launch_bounds__(256) k_rot(int *base, size_t plane, int N, int T, uint32_t pace)
{
    typedef typename Vec<4>::type V;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    int *p = base + (size_t)blockIdx.x * 256 + l * 4 + (size_t)w * N;
    for (int t = w; t < T; t += 4, p += 4 * (size_t)N) {
        V v = {t, t + 1, t + 2, t + 3};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            if (NT) __builtin_nontemporal_store(v, (V *)(p + q * plane));
            else *(V *)(p + q * plane) = v;
        }
        if (pace) idle(pace);
    }
}

template <int VEC, int PLANES, bool NT>
__global__ void __launch_bounds__(256) k_rows(int *base, size_t plane, int N, int T, uint32_t pace, int ncols)
{
    typedef typename Vec<VEC>::type V;
    const size_t col = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
    if (col >= (size_t)ncols) return;
    int *p = base + col;
    for (int t = 0; t < T; ++t, p += N) {
        V v;
        if constexpr (VEC == 1) v = t;
        else {
#pragma unroll
            for (int k = 0; k < VEC; ++k) v[k] = t + k;
        }
#pragma unroll
        for (int q = 0; q < PLANES; ++q) {
            if (NT) __builtin_nontemporal_store(v, (V *)(p + q * plane));
            else *(V *)(p + q * plane) = v;
        }
        if ((t & 3) == 3 && pace) idle(pace);
    }
}

// the same stream through buffer stores with a cache-policy modifier (aux: 1 = sc0, 2 = nt, 16 = sc1; sc1 = the store is written
// through at device scope instead of staying dirty in the L2 until it is evicted)
template <int VEC, int AUX>
__global__ void __launch_bounds__(256) k_rows_aux(int *base, size_t plane, int N, int T, uint32_t pace)
{
    const uint32_t off = (blockIdx.x * blockDim.x + threadIdx.x) * VEC * 4u;
    if (off >= (uint32_t)N * 4u) return;
    char *p = (char *)base;
    for (int t = 0; t < T; ++t, p += (size_t)N * 4) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p + q * plane * 4, 0, 0xFFFFFFFFu, 0x00020000);
            if constexpr (VEC == 1) __builtin_amdgcn_raw_buffer_store_b32(t, r, off, 0, AUX);
            else if constexpr (VEC == 2) {
                typedef int v2 __attribute__((ext_vector_type(2)));
                __builtin_amdgcn_raw_buffer_store_b64(v2{t, t + 1}, r, off, 0, AUX);
            } else {
                typedef int v4 __attribute__((ext_vector_type(4)));
                __builtin_amdgcn_raw_buffer_store_b128(v4{t, t + 1, t + 2, t + 3}, r, off, 0, AUX);
            }
        }
        if ((t & 3) == 3 && pace) idle(pace);
    }
}

// ... and with a dependent chain of CHAIN LDS look-ups per step in front of the stores (the product's step is such a chain: the
// next position needs the cell record of this one): does a wave that cannot catch up after a stall cost write rate?
template <int CHAIN, int AUX>
__global__ void __launch_bounds__(256) k_chain(int *base, size_t plane, int N, int T, uint32_t pace)
{
    __shared__ int next[1024];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) next[i] = (i * 677 + 131) & 1023;  // a permutation of 0..1023 (677 is odd)
    __syncthreads();
    const uint32_t off = (blockIdx.x * blockDim.x + threadIdx.x) * 4u;
    int s = threadIdx.x;
    char *p = (char *)base;
    for (int t = 0; t < T; ++t, p += (size_t)N * 4) {
#pragma unroll
        for (int k = 0; k < CHAIN; ++k) s = next[s];
#pragma unroll
        for (int q = 0; q < 3; ++q)
            __builtin_amdgcn_raw_buffer_store_b32(s + q, __builtin_amdgcn_make_buffer_rsrc(p + q * plane * 4, 0, 0xFFFFFFFFu, 0x00020000), off, 0, AUX);
        if ((t & 3) == 3 && pace) idle(pace);
    }
}

struct Shape {
    const char *name;
    int vec, planes, N, block;
    bool nt;
    int parts, rot;  // parts: the columns as that many launches in a row (N / parts columns each); rot: k_rot
    int auxp1;       // > 0: k_rows_aux (buffer stores) with modifier auxp1 - 1
    int chain;       // > 0: k_chain with that many dependent LDS look-ups per step (dword, 1024 waves)
};

template <int VEC, int PLANES, bool NT>
static void go(const Shape &s, int *buf, uint32_t pace)
{
    const size_t total = (size_t)65536 * 1000 * 3;  // dwords of the bench launch
    const int T = (int)(total / ((size_t)s.N * PLANES));
    const int parts = s.parts > 0 ? s.parts : 1, ncols = s.N / parts;
    const size_t lanes = (size_t)ncols / VEC;
    for (int q = 0; q < parts; ++q)
        k_rows<VEC, PLANES, NT><<<dim3((unsigned)((lanes + s.block - 1) / s.block)), dim3(s.block)>>>(buf + (size_t)q * ncols, (size_t)s.N * T, s.N, T, pace, ncols);
}

static void launch(const Shape &s, int *buf, uint32_t pace)
{
    if (s.chain) {
        const dim3 g(256), b(256);
        const size_t plane = (size_t)65536 * 1000;
#define CHAINCASE(C, A) if (s.chain == C && s.auxp1 - 1 == A) k_chain<C, A><<<g, b>>>(buf, plane, 65536, 1000, pace)
        CHAINCASE(1, 16); CHAINCASE(2, 16); CHAINCASE(3, 16); CHAINCASE(2, 0);
    } else if (s.auxp1) {
        const int T = 1000 * 65536 / s.N, aux = s.auxp1 - 1;
        const dim3 g((unsigned)(s.N / s.vec / s.block)), b(s.block);
        const size_t plane = (size_t)s.N * T;
#define AUXCASE(V, A) if (s.vec == V && aux == A) k_rows_aux<V, A><<<g, b>>>(buf, plane, s.N, T, pace)
        AUXCASE(1, 0); AUXCASE(1, 16); AUXCASE(1, 17); AUXCASE(1, 2); AUXCASE(1, 18);
        AUXCASE(2, 0); AUXCASE(2, 16); AUXCASE(2, 17);
        AUXCASE(4, 0); AUXCASE(4, 16); AUXCASE(4, 17); AUXCASE(4, 18);
    } else if (s.rot) {
        const int T = 1000 * 65536 / s.N;
        if (s.nt) k_rot<true><<<dim3(s.N / 256), dim3(256)>>>(buf, (size_t)s.N * T, s.N, T, pace);
        else k_rot<false><<<dim3(s.N / 256), dim3(256)>>>(buf, (size_t)s.N * T, s.N, T, pace);
    } else if (s.nt) {
        if (s.vec == 1) go<1, 3, true>(s, buf, pace);
        else go<4, 3, true>(s, buf, pace);  // (x2 nt: not built)
    } else if (s.planes == 1) {
        if (s.vec == 1) go<1, 1, false>(s, buf, pace);
        else go<4, 1, false>(s, buf, pace);
    } else {
        if (s.vec == 1) go<1, 3, false>(s, buf, pace);
        else if (s.vec == 2) go<2, 3, false>(s, buf, pace);
        else go<4, 3, false>(s, buf, pace);
    }
}

int main(int argc, char **argv)
{
    const int buffers = argc > 1 ? atoi(argv[1]) : 6;
    const size_t bytes = (size_t)65536 * 1000 * 12;
    const Shape shapes[] = {
        {"dword 1024 waves", 1, 3, 65536, 256, false, 0, 0, 1, 0},
        {"dword 1024 waves sc1", 1, 3, 65536, 256, false, 0, 0, 17, 0},
        {"dword 1024 waves sc1 + 1 look-up per step", 1, 3, 65536, 256, false, 0, 0, 17, 1},
        {"dword 1024 waves sc1 + 2 look-ups per step", 1, 3, 65536, 256, false, 0, 0, 17, 2},
        {"dword 1024 waves sc1 + 3 look-ups per step", 1, 3, 65536, 256, false, 0, 0, 17, 3},
        {"dword 1024 waves     + 2 look-ups per step", 1, 3, 65536, 256, false, 0, 0, 1, 2},
        {"x4 256 waves sc1 nt", 4, 3, 65536, 64, false, 0, 0, 19, 0},
    };
    std::vector<int *> bufs;
    for (int b = 0; b < buffers; ++b) {
        int *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) break;
        bufs.push_back(p);
    }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timed = [&](const Shape &s, int *buf, int turns, int reps) {
        const uint32_t w = pace_word(turns);
        launch(s, buf, w);
        launch(s, buf, w);
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) launch(s, buf, w);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms / reps * 1e3f;
    };
    for (int i = 0; i < 300; ++i) launch(shapes[0], bufs[0], 0u);  // working clocks
    CK(hipDeviceSynchronize());
    printf("%zu buffers of %.0f MB; per shape and buffer: unpaced us -> best us @ idle turns per 4 steps (TB/s at best)\n", bufs.size(), bytes / 1e6);
    for (const Shape &s : shapes) {
        printf("%-58s", s.name);
        for (int *buf : bufs) {
            const float t0 = timed(s, buf, 0, 4);
            float best = t0;
            int at = 0;
            std::vector<int> ladder;  // geometric, down from the healthy side (like the product's calibration), then every value around the best
            for (double g = 700.0; g >= 1.0; g /= 1.18)
                if (ladder.empty() || (int)g < ladder.back()) ladder.push_back((int)g);
            for (int turns : ladder) {
                const float t = timed(s, buf, turns, 3);
                if (t < best) best = t, at = turns;
            }
            if (at) {
                const int lo = (int)(at / 1.18), hi = (int)(at * 1.18) + 1, step = (hi - lo) / 12 + 1, coarse = at;
                for (int turns = hi; turns >= lo && turns >= 1; turns -= step) {
                    if (turns == coarse) continue;
                    const float t = timed(s, buf, turns, 3);
                    if (t < best) best = t, at = turns;
                }
                best = timed(s, buf, at, 8);
            }
            printf(" | %6.1f -> %6.1f @%3d (%.2f)", t0, best, at, bytes / (best * 1e-6) / 1e12);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
