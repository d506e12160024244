// placement_pieces.hip -- a contiguous physical block is reliably in the slow write-rate class (placement_vmm.hip,
// placement_stride.hip).  Is a buffer pieced together from SEPARATE physical handles (mapped back to back in one virtual range)
// in the fast one?  Piece sizes 2 MiB .. 250 MiB; each variant is built several times.
// hipcc --offload-arch=gfx950 -O3 -o placement_pieces placement_pieces.hip && ./placement_pieces [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <chrono>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

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

// `bytes` of virtual range backed by handles of `piece` bytes each; `shuffle`: map the pieces in a scrambled order
static int* build(size_t bytes, size_t piece, bool shuffle, double* ms)
{
    const double t0 = now_ms();
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    const size_t n = (bytes + piece - 1) / piece;
    void* va = nullptr;
    CK(hipMemAddressReserve(&va, n * piece, 2ull << 20, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> hs(n);
    for (size_t i = 0; i < n; ++i) CK(hipMemCreate(&hs[i], piece, &prop, 0));
    for (size_t i = 0; i < n; ++i) {
        const size_t slot = shuffle ? (i * 7919) % n : i;  // (7919 is prime: a permutation whenever n is not a multiple of it)
        CK(hipMemMap((char*)va + slot * piece, piece, 0, hs[i], 0));
    }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, n * piece, &acc, 1));
    *ms = now_ms() - t0;
    return (int*)va;
}

int main(int argc, char** argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 4;
    const size_t bytes = (size_t)3 * N * T * 4, M = 1 << 20;
    CK(hipEventCreate(&ev_a));
    CK(hipEventCreate(&ev_b));
    for (int r = 0; r < rounds; ++r) {
        int* a = nullptr;
        CK(hipMalloc(&a, bytes));
        printf("round %d  hipMalloc %.1f us", r, probe(a));
        const size_t pieces[] = {250 * M, 50 * M, 16 * M, 2 * M};
        for (size_t p : pieces)
            for (int sh = 0; sh < 2; ++sh) {
                double ms;
                int* b = build(bytes, p, sh, &ms);
                printf("   %zu MiB pieces%s %.1f us (built in %.1f ms)", p / M, sh ? " shuffled" : "", probe(b), ms);
                fflush(stdout);
            }
        printf("\n");
        void* sp = nullptr;
        if (hipMalloc(&sp, (size_t)2 << 30) != hipSuccess) (void)hipGetLastError();
    }
    return 0;
}
