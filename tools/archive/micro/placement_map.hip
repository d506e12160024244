// placement_map.hip -- how are the fast and the slow write-rate classes (store_placement.hip) spread over the whole HBM?
// A 786 MB candidate is written in the rollout's store shape, then a spacer of `gap` GiB is allocated and HELD, and so on
// until `span` GiB are held or hipMalloc fails.  Prints us per launch per candidate next to the GiB allocated before it,
// plus the device virtual address.  hipcc --offload-arch=gfx950 -O3 -o placement_map placement_map.hip && ./placement_map [gap GiB] [span GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

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
    const double gap = argc > 1 ? atof(argv[1]) : 3.0, span = argc > 2 ? atof(argv[2]) : 200.0;
    const int N = 65536, T = 1000;
    const size_t bytes = (size_t)3 * N * T * 4;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    size_t fr, tot;
    CK(hipMemGetInfo(&fr, &tot));
    printf("free %.1f GiB of %.1f; candidate %.0f MB, spacer %.2f GiB\n", fr / 1073741824.0, tot / 1073741824.0, bytes / 1e6, gap);
    double held = 0;
    std::vector<void*> keep;
    for (int i = 0; held < span; ++i) {
        int* buf = nullptr;
        const double t0 = now_ms();
        if (hipMalloc(&buf, bytes) != hipSuccess) { printf("candidate %d: hipMalloc failed\n", i); break; }
        keep.push_back(buf);
        const double t_cand = now_ms() - t0;
        double t_sp = 0;
        float best = 1e9f;
        for (int r = 0; r < 4; ++r) {
            CK(hipEventRecord(a));
            for (int j = 0; j < 3; ++j) hipLaunchKernelGGL(k3, dim3(N / 256), dim3(256), 0, 0, buf, N, T);
            CK(hipEventRecord(b));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            if (r && ms / 3 < best) best = ms / 3;
        }
        const double at = held;
        held += bytes / 1073741824.0;
        if (gap > 0) {
            void* sp = nullptr;
            const double t1 = now_ms();
            if (hipMalloc(&sp, (size_t)(gap * 1073741824.0)) != hipSuccess) { printf("spacer: hipMalloc failed\n"); break; }
            t_sp = now_ms() - t1;
            keep.push_back(sp);
            held += gap;
        }
        printf("at %6.1f GiB  va %p  %.1f us %s  (hipMalloc: candidate %.2f ms, spacer %.2f ms)\n", at, (void*)buf, best * 1e3, best * 1e3 < 125 ? "FAST" : "", t_cand, t_sp);
        fflush(stdout);
    }
    const double t2 = now_ms();
    for (void* p : keep) (void)hipFree(p);
    printf("hipFree of all %zu blocks: %.1f ms\n", keep.size(), now_ms() - t2);
    return 0;
}
