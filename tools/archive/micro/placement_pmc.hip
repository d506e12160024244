// placement_pmc.hip -- the rollout-shaped store loop on several allocations of one process (3 GiB spacers in between), FOUR
// dispatches per buffer (one warm-up, three timed with events), meant to run under `rocprofv3 --pmc ...` so that the counters
// of every dispatch can be set against the write-rate class of the buffer it wrote (tools/placement_pmc.sh).
// Prints "buffer i <us per launch>"; dispatches 4 i .. 4 i + 3 belong to buffer i.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k3(int* __restrict__ buf, int N, int T)
{
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t plane = (size_t)N * T;
    int s = e;
    size_t o = e;
    for (int t = 0; t < T; ++t) {
        s = s * 1664525 + 1013904223;
        buf[o] = s;
        buf[plane + o] = s >> 3;
        buf[2 * plane + o] = s & 1;
        o += N;
    }
}

int main(int argc, char** argv)
{
    const int buffers = argc > 1 ? atoi(argv[1]) : 10;
    const int N = 65536, T = 1000;
    const size_t bytes = (size_t)3 * N * T * 4;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    for (int i = 0; i < buffers; ++i) {
        int* buf = nullptr;
        CK(hipMalloc(&buf, bytes));
        hipLaunchKernelGGL(k3, dim3(N / 256), dim3(256), 0, 0, buf, N, T);
        CK(hipDeviceSynchronize());
        float best = 1e9f;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(k3, dim3(N / 256), dim3(256), 0, 0, buf, N, T);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        printf("buffer %d %.1f\n", i, best * 1e3);
        fflush(stdout);
        void* sp = nullptr;
        if (hipMalloc(&sp, (size_t)3 << 30) != hipSuccess) (void)hipGetLastError();
    }
    return 0;
}
