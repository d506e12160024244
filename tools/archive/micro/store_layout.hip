// store_layout.hip -- does the ORDER in which the rollout's trajectory lands in HBM matter?  Tuning aid, not part of the
// product.  65 536 lanes (1024 waves, one per SIMD), 1000 steps, three dwords per lane and step, trivial compute:
//   rows    : plane[t][e]                     the product's layout: every wave-store is 256 B of a 256 KiB row
//   wave    : plane[e / 64][t][64]            each wave appends to its OWN contiguous stream, one per plane
//   block   : plane[e / 256][t][256]          each 4-wave workgroup appends 1 KiB per step to its own stream
//   wave3   : buf[e / 64][t][3][64]           the three planes interleaved: ONE 768-B-per-step stream per wave
//   block3  : buf[e / 256][t][3][256]         ONE 3-KiB-per-step stream per workgroup
// hipcc --offload-arch=gfx950 -O3 -o store_layout store_layout.hip && ./store_layout [N]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// rows written first-to-last (dir = +1) or last-to-first (dir = -1): does a launch that starts on the rows the previous
// launch wrote LAST -- still dirty in L2 / Infinity Cache -- save their write-back?
__global__ void __launch_bounds__(256) k_rows_dir(int* __restrict__ buf, int N, int T, int dir)
{
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t plane = (size_t)N * T;
    int s = e;
    long long o = dir > 0 ? (long long)e : (long long)(T - 1) * N + e;
    const long long step = dir > 0 ? N : -(long long)N;
    for (int t = 0; t < T; ++t) {
        s = s * 1664525 + 1013904223;
        buf[o] = s; buf[plane + o] = s >> 3; buf[2 * plane + o] = s & 1;
        o += step;
    }
}

template <int LAYOUT>
__global__ void __launch_bounds__(256) k_store(int* __restrict__ buf, int N, int T)
{
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t plane = (size_t)N * T;
    int s = e;
    size_t o0, o1, o2, step;
    if (LAYOUT == 0) { o0 = e; o1 = plane + e; o2 = 2 * plane + e; step = N; }
    else if (LAYOUT == 1) { o0 = (size_t)(e >> 6) * T * 64 + (e & 63); o1 = o0 + plane; o2 = o1 + plane; step = 64; }
    else if (LAYOUT == 2) { o0 = (size_t)(e >> 8) * T * 256 + (e & 255); o1 = o0 + plane; o2 = o1 + plane; step = 256; }
    else if (LAYOUT == 3) { o0 = (size_t)(e >> 6) * T * 192 + (e & 63); o1 = o0 + 64; o2 = o0 + 128; step = 192; }
    else { o0 = (size_t)(e >> 8) * T * 768 + (e & 255); o1 = o0 + 256; o2 = o0 + 512; step = 768; }
    for (int t = 0; t < T; ++t) {
        s = s * 1664525 + 1013904223;  // a little dependent integer work
        buf[o0] = s; buf[o1] = s >> 3; buf[o2] = s & 1;
        o0 += step; o1 += step; o2 += step;
    }
}

int main(int argc, char** argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 65536, T = 1000, reps = 20;
    int* buf;
    const size_t bytes = (size_t)N * T * 4 * 3;
    CK(hipMalloc(&buf, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[] = {"rows", "wave", "block", "wave3", "block3", "rows-alternating", "rows-backward"};
    int flip = 1;
    for (int round = 0; round < 3; ++round)
        for (int v = 0; v < 7; ++v) {
            dim3 g(N / 256), blk(256);
            auto launch = [&]() {
                switch (v) {
                case 5: k_rows_dir<<<g, blk>>>(buf, N, T, flip); flip = -flip; break;
                case 6: k_rows_dir<<<g, blk>>>(buf, N, T, -1); break;
                case 0: k_store<0><<<g, blk>>>(buf, N, T); break;
                case 1: k_store<1><<<g, blk>>>(buf, N, T); break;
                case 2: k_store<2><<<g, blk>>>(buf, N, T); break;
                case 3: k_store<3><<<g, blk>>>(buf, N, T); break;
                default: k_store<4><<<g, blk>>>(buf, N, T); break;
                }
            };
            for (int i = 0; i < 3; ++i) launch();
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) launch();
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("N=%d %-6s : %.4f ms/launch  %.2f TB/s\n", N, names[v], ms / reps, bytes / (ms / reps * 1e-3) / 1e12);
        }
    return 0;
}
