// store_placement.hip -- does the SAME store kernel run at different speeds on different allocations of one process?
// (The unchanged bench kernel takes 119 us in one session and 138 us in the next ON THE SAME GPU, at identical clocks.)
// 65 536 lanes x 1000 steps x 3 dwords = 786 MB per launch, the rollout's store shape, into each of several hipMalloc'ed
// buffers in turn.  hipcc --offload-arch=gfx950 -O3 -o store_placement store_placement.hip && ./store_placement [buffers]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// `skew`: extra elements between consecutive planes; `planes`: how many of the three planes the launch writes
__global__ void __launch_bounds__(256) k_var(int* __restrict__ buf, int N, int T, size_t skew, int planes, int first)
{
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t plane = (size_t)N * T + skew;
    int s = e;
    size_t o = e + (size_t)first * plane;
    for (int t = 0; t < T; ++t) {
        s = s * 1664525 + 1013904223;
        buf[o] = s;
        if (planes > 1) buf[plane + o] = s >> 3;
        if (planes > 2) buf[2 * plane + o] = s & 1;
        o += N;
    }
}

// three planes, the three stores of a step SPACED by dependent integer work instead of issued back to back
// (GAP = s_sleep argument between the stores: 64 clocks each; a chain of multiply-adds would be folded by the compiler)
template <int GAP>
__global__ void __launch_bounds__(256) k_spaced(int* __restrict__ buf, int N, int T)
{
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t plane = (size_t)N * T;
    int s = e;
    size_t o = e;
    for (int t = 0; t < T; ++t) {
        s = s * 1664525 + 1013904223;
        buf[o] = s;
        __builtin_amdgcn_s_sleep(GAP);
        buf[plane + o] = s >> 3;
        __builtin_amdgcn_s_sleep(GAP);
        buf[2 * plane + o] = s & 1;
        __builtin_amdgcn_s_sleep(GAP);
        o += N;
    }
}

// three planes, but B steps are kept in registers and then stored PLANE BY PLANE: B rows of plane 0, B of plane 1, B of plane 2
template <int B>
__global__ void __launch_bounds__(256) k_burst(int* __restrict__ buf, int N, int T)
{
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t plane = (size_t)N * T;
    int s = e;
    size_t o = e;
    for (int t = 0; t < T; t += B) {
        int v[B];
#pragma unroll
        for (int j = 0; j < B; ++j) { s = s * 1664525 + 1013904223; v[j] = s; }
#pragma unroll
        for (int j = 0; j < B; ++j) buf[o + (size_t)j * N] = v[j];
#pragma unroll
        for (int j = 0; j < B; ++j) buf[plane + o + (size_t)j * N] = v[j] >> 3;
#pragma unroll
        for (int j = 0; j < B; ++j) buf[2 * plane + o + (size_t)j * N] = v[j] & 1;
        o += (size_t)B * N;
    }
}

// data dependence: MODE 0 = three planes, all three values full-entropy; MODE 1 = one stream over the whole buffer writing
// the low-entropy pattern (s, s >> 3, s & 1) row after row; MODE 2 = three planes, all zeros
template <int MODE>
__global__ void __launch_bounds__(256) k_data(int* __restrict__ buf, int N, int T)
{
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t plane = (size_t)N * T;
    int s = e;
    size_t o = e;
    if (MODE == 1) {
        for (int t = 0; t < 3 * T; ++t) {
            s = s * 1664525 + 1013904223;
            const int k = t % 3;
            buf[o] = k == 0 ? s : k == 1 ? (s >> 3) : (s & 1);
            o += N;
        }
        return;
    }
    for (int t = 0; t < T; ++t) {
        s = s * 1664525 + 1013904223;
        buf[o] = MODE == 2 ? 0 : s; buf[plane + o] = MODE == 2 ? 0 : s * 7 + 3; buf[2 * plane + o] = MODE == 2 ? 0 : s * 13 + 5;
        o += N;
    }
}

// three planes, ONE store per (non-unrolled) loop iteration: iteration k writes plane k % 3 of step k / 3
__global__ void __launch_bounds__(256) k_cycle(int* __restrict__ buf, int N, int T)
{
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t plane = (size_t)N * T;
    int s = e;
    size_t o = e, po = 0;
    int k = 0;
#pragma unroll 1
    for (int i = 0; i < 3 * T; ++i) {
        s = s * 1664525 + 1013904223;
        buf[po + o] = s;
        ++k;
        po += plane;
        if (k == 3) { k = 0; po = 0; o += N; }
    }
}

// plane-specialised waves: a 768-thread workgroup handles 256 envs, threads 0..255 store plane 0, 256..511 plane 1,
// 512..767 plane 2 (each recomputes the cheap value chain): every WAVE writes one stream
__global__ void __launch_bounds__(768) k_split3(int* __restrict__ buf, int N, int T)
{
    const unsigned which = threadIdx.x >> 8;
    const unsigned e = blockIdx.x * 256 + (threadIdx.x & 255);
    const size_t plane = (size_t)N * T;
    int s = e;
    size_t o = e + which * plane;
    for (int t = 0; t < T; ++t) {
        s = s * 1664525 + 1013904223;
        buf[o] = which == 0 ? s : which == 1 ? (s >> 3) : (s & 1);
        o += N;
    }
}

// the same with the three planes' waves in DIFFERENT workgroups (grid = 3 x N / 256)
__global__ void __launch_bounds__(256) k_split3_blocks(int* __restrict__ buf, int N, int T)
{
    const unsigned nb = N / 256;
    const unsigned which = blockIdx.x / nb;
    const unsigned e = (blockIdx.x % nb) * 256 + threadIdx.x;
    const size_t plane = (size_t)N * T;
    int s = e;
    size_t o = e + which * plane;
    for (int t = 0; t < T; ++t) {
        s = s * 1664525 + 1013904223;
        buf[o] = which == 0 ? s : which == 1 ? (s >> 3) : (s & 1);
        o += N;
    }
}

// [t][3][N] with the three stores of a step spaced by dependent integer work
template <int GAP>
__global__ void __launch_bounds__(256) k_rows3_spaced(int* __restrict__ buf, int N, int T)
{
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    int s = e;
    size_t o = e;
    for (int t = 0; t < T; ++t) {
        s = s * 1664525 + 1013904223;
        buf[o] = s;
        __builtin_amdgcn_s_sleep(GAP);
        buf[o + N] = s >> 3;
        __builtin_amdgcn_s_sleep(GAP);
        buf[o + 2 * (size_t)N] = s & 1;
        __builtin_amdgcn_s_sleep(GAP);
        o += 3 * (size_t)N;
    }
}

// array of structs: buf[t][N][3] -- every lane stores its (obs, reward, done) as ONE 12-byte store, a wave 768 contiguous bytes
struct __attribute__((packed, aligned(4))) Triple { int a, b, c; };
__global__ void __launch_bounds__(256) k_aos(int* __restrict__ buf, int N, int T)
{
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    int s = e;
    Triple* p = reinterpret_cast<Triple*>(buf) + e;
    for (int t = 0; t < T; ++t) {
        s = s * 1664525 + 1013904223;
        *p = Triple{s, s >> 3, s & 1};
        p += N;
    }
}

// the three rows of a step ADJACENT: buf[t][3][N] -- one 768 KB window per step instead of three windows 250 MiB apart
__global__ void __launch_bounds__(256) k_rows3(int* __restrict__ buf, int N, int T)
{
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    int s = e;
    size_t o = e;
    for (int t = 0; t < T; ++t) {
        s = s * 1664525 + 1013904223;
        buf[o] = s; buf[o + N] = s >> 3; buf[o + 2 * (size_t)N] = s & 1;
        o += 3 * (size_t)N;
    }
}

__global__ void __launch_bounds__(256) k_rows(int* __restrict__ buf, int N, int T)
{
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t plane = (size_t)N * T;
    int s = e;
    size_t o = e;
    for (int t = 0; t < T; ++t) {
        s = s * 1664525 + 1013904223;
        buf[o] = s; buf[plane + o] = s >> 3; buf[2 * plane + o] = s & 1;
        o += N;
    }
}

int main(int argc, char** argv)
{
    const int nbuf = argc > 1 ? atoi(argv[1]) : 12, N = 65536, T = 1000, reps = 20;
    const size_t bytes = (size_t)N * T * 4 * 3;
    std::vector<int*> bufs(nbuf);
    for (auto& b : bufs) CK(hipMalloc(&b, bytes + (64 << 20)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int round = 0; round < 3; ++round)
        for (int i = 0; i < nbuf; ++i) {
            for (int r = 0; r < 3; ++r) k_rows<<<N / 256, 256>>>(bufs[i], N, T);
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; ++r) k_rows<<<N / 256, 256>>>(bufs[i], N, T);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("round %d buffer %2d @%p : %.1f us/launch  %.2f TB/s\n", round, i, (void*)bufs[i], ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
        }
    // one stream over the whole buffer (3T rows of one "plane") against the three concurrent streams, per buffer; and reads
    for (int i = 0; i < nbuf; ++i) {
        int* b = bufs[i];
        float ms;
        for (int r = 0; r < 2; ++r) k_var<<<N / 256, 256>>>(b, N, 3 * T, 0, 1, 0);
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) k_var<<<N / 256, 256>>>(b, N, 3 * T, 0, 1, 0);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        const float one = ms / reps * 1e3f;
        for (int r = 0; r < 2; ++r) k_rows<<<N / 256, 256>>>(b, N, T);
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) k_rows<<<N / 256, 256>>>(b, N, T);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        const float three = ms / reps * 1e3f;
        for (int r = 0; r < 2; ++r) k_aos<<<N / 256, 256>>>(b, N, T);
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) k_aos<<<N / 256, 256>>>(b, N, T);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        const float aos = ms / reps * 1e3f;
        float sp[3];
        for (int v = 0; v < 3; ++v) {
            auto launch = [&] { if (v == 0) k_spaced<1><<<N / 256, 256>>>(b, N, T); else if (v == 1) k_spaced<2><<<N / 256, 256>>>(b, N, T); else k_spaced<4><<<N / 256, 256>>>(b, N, T); };
            for (int r = 0; r < 2; ++r) launch();
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; ++r) launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            sp[v] = ms / reps * 1e3f;
        }
        float r3[3];
        for (int v = 0; v < 3; ++v) {
            auto launch = [&] { if (v == 0) k_rows3_spaced<1><<<N / 256, 256>>>(b, N, T); else if (v == 1) k_rows3_spaced<2><<<N / 256, 256>>>(b, N, T); else k_rows3_spaced<4><<<N / 256, 256>>>(b, N, T); };
            for (int r = 0; r < 2; ++r) launch();
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; ++r) launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            r3[v] = ms / reps * 1e3f;
        }
        printf("buffer %2d: [t][3][N] spaced by s_sleep 1 / 2 / 4: %.1f / %.1f / %.1f us\n", i, r3[0], r3[1], r3[2]);
        float da[3];
        for (int v = 0; v < 3; ++v) {
            auto launch = [&] { if (v == 0) k_data<0><<<N / 256, 256>>>(b, N, T); else if (v == 1) k_data<1><<<N / 256, 256>>>(b, N, T); else k_data<2><<<N / 256, 256>>>(b, N, T); };
            for (int r = 0; r < 2; ++r) launch();
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; ++r) launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            da[v] = ms / reps * 1e3f;
        }
        printf("buffer %2d: data: 3 planes full-entropy %.1f us | one stream, low-entropy rows %.1f us | 3 planes all zeros %.1f us\n", i, da[0], da[1], da[2]);
        {
            for (int r = 0; r < 2; ++r) k_cycle<<<N / 256, 256>>>(b, N, T);
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; ++r) k_cycle<<<N / 256, 256>>>(b, N, T);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("buffer %2d: three planes, one store per loop iteration (plane cycling): %.1f us\n", i, ms / reps * 1e3f);
        }
        {
            float t768, t3b;
            for (int r = 0; r < 2; ++r) k_split3<<<N / 256, 768>>>(b, N, T);
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; ++r) k_split3<<<N / 256, 768>>>(b, N, T);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            t768 = ms / reps * 1e3f;
            for (int r = 0; r < 2; ++r) k_split3_blocks<<<3 * N / 256, 256>>>(b, N, T);
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; ++r) k_split3_blocks<<<3 * N / 256, 256>>>(b, N, T);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            t3b = ms / reps * 1e3f;
            printf("buffer %2d: plane-specialised waves: one 768-thread workgroup per 256 envs %.1f us | three 256-thread workgroups %.1f us\n", i, t768, t3b);
        }
        float bu[3];
        for (int v = 0; v < 3; ++v) {
            auto launch = [&] { if (v == 0) k_burst<4><<<N / 256, 256>>>(b, N, T); else if (v == 1) k_burst<8><<<N / 256, 256>>>(b, N, T); else k_burst<20><<<N / 256, 256>>>(b, N, T); };
            for (int r = 0; r < 2; ++r) launch();
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; ++r) launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            bu[v] = ms / reps * 1e3f;
        }
        printf("buffer %2d: 3 planes, stored plane by plane in bursts of 4 / 8 / 20 rows: %.1f / %.1f / %.1f us\n", i, bu[0], bu[1], bu[2]);
        printf("buffer %2d: three streams %.1f us   one stream %.1f us   [t][N][3] %.1f us   3 planes, stores spaced by s_sleep 1 / 2 / 4: %.1f / %.1f / %.1f us\n", i, three, one, aos, sp[0], sp[1], sp[2]);
    }
    // does the speed depend on where INSIDE one allocation the 786 MB window starts?
    {
        int* slab;
        CK(hipMalloc(&slab, bytes + ((size_t)520 << 20)));
        for (size_t off_mb : {0, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512}) {
            int* b = slab + (off_mb << 20) / 4;
            for (int r = 0; r < 2; ++r) k_rows<<<N / 256, 256>>>(b, N, T);
            CK(hipEventRecord(e0));
            for (int r = 0; r < reps; ++r) k_rows<<<N / 256, 256>>>(b, N, T);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("slab %p window at +%3zu MiB : %.1f us/launch  %.2f TB/s\n", (void*)slab, off_mb, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
        }
        CK(hipFree(slab));
    }
    // variants on every buffer: plane skews, and one plane at a time
    auto timed = [&](auto launch, double nbytes, const char* what, int i) {
        for (int r = 0; r < 2; ++r) launch();
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("buffer %2d %-34s: %.1f us/launch  %.2f TB/s\n", i, what, ms / reps * 1e3, nbytes / (ms / reps * 1e-3) / 1e12);
    };
    for (int i = 0; i < nbuf; ++i) {
        int* b = bufs[i];
        timed([&] { k_var<<<N / 256, 256>>>(b, N, T, 0, 3, 0); }, bytes, "3 planes, no skew", i);
        timed([&] { k_rows3<<<N / 256, 256>>>(b, N, T); }, bytes, "rows of a step adjacent [t][3][N]", i);
        for (size_t skew : {(size_t)1024, (size_t)16384, (size_t)(1 << 18) + 4096, (size_t)(1 << 20) + 65536 + 1024}) {
            char name[64]; snprintf(name, sizeof name, "3 planes, skew %zu elems", skew);
            timed([&] { k_var<<<N / 256, 256>>>(b, N, T, skew, 3, 0); }, bytes, name, i);
        }
        for (int p = 0; p < 3; ++p) {
            char name[64]; snprintf(name, sizeof name, "plane %d alone", p);
            timed([&] { k_var<<<N / 256, 256>>>(b, N, T, 0, 1, p); }, bytes / 3.0, name, i);
        }
    }
    return 0;
}
