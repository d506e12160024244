// placement_flags.hip -- the store loop on buffers from hipExtMallocWithFlags: default, uncached (MTYPE_UC), fine-grained,
// contiguous; several of each, interleaved, with spacers.  hipcc --offload-arch=gfx950 -O3 -o placement_flags placement_flags.hip
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

static const int N = 65536, T = 1000;
static hipEvent_t ev_a, ev_b;

static float probe(int* buf)
{
    float best = 1e9f;
    for (int r = 0; r < 4; ++r) {
        CK(hipEventRecord(ev_a));
        for (int j = 0; j < 3; ++j) hipLaunchKernelGGL(k3, dim3(N / 256), dim3(256), 0, 0, buf, N, T);
        CK(hipEventRecord(ev_b));
        CK(hipEventSynchronize(ev_b));
        float ms;
        CK(hipEventElapsedTime(&ms, ev_a, ev_b));
        if (r && ms / 3 < best) best = ms / 3;
    }
    CK(hipGetLastError());
    return best * 1e3f;
}

int main(int argc, char** argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 6;
    const size_t bytes = (size_t)3 * N * T * 4;
    CK(hipEventCreate(&ev_a));
    CK(hipEventCreate(&ev_b));
    struct { const char* name; unsigned flags; } kinds[] = {{"default", hipDeviceMallocDefault}, {"uncached", hipDeviceMallocUncached},
                                                            {"finegrained", hipDeviceMallocFinegrained}, {"contiguous", hipDeviceMallocContiguous}};
    for (int r = 0; r < rounds; ++r) {
        printf("round %d", r);
        for (auto& k : kinds) {
            int* p = nullptr;
            hipError_t e = hipExtMallocWithFlags((void**)&p, bytes, k.flags);
            if (e != hipSuccess) { (void)hipGetLastError(); printf("   %s: %s", k.name, hipGetErrorString(e)); continue; }
            printf("   %s %.1f us", k.name, probe(p));
            fflush(stdout);
        }
        printf("\n");
        void* sp = nullptr;
        if (hipMalloc(&sp, (size_t)3 << 30) != hipSuccess) (void)hipGetLastError();
    }
    return 0;
}
