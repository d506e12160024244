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
