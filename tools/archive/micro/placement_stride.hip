// placement_stride.hip -- on an allocation that is RELIABLY in the slow write-rate class (a 4 GiB physical handle mapped at a
// 4 GiB-aligned address, cf. placement_vmm.hip): does the distance between the three planes, the row pitch, or the base offset
// inside the block decide the rate?  65 536 lanes x 1000 steps x 3 planes, the rollout's store shape.
// hipcc --offload-arch=gfx950 -O3 -o placement_stride placement_stride.hip && ./placement_stride
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// plane stride and row pitch in BYTES
__global__ void __launch_bounds__(256) k3(char* __restrict__ buf, int T, size_t plane, size_t pitch)
{
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    int s = e;
    char* o = buf + (size_t)e * 4;
    for (int t = 0; t < T; ++t) {
        s = s * 1664525 + 1013904223;
        *(int*)o = s;
        *(int*)(o + plane) = s >> 3;
        *(int*)(o + 2 * plane) = s & 1;
        o += pitch;
    }
}

static const int N = 65536, T = 1000;
static hipEvent_t ev_a, ev_b;

static float probe(char* buf, size_t plane, size_t pitch)
{
    float best = 1e9f;
    for (int r = 0; r < 4; ++r) {
        CK(hipEventRecord(ev_a));
        for (int j = 0; j < 3; ++j) hipLaunchKernelGGL(k3, dim3(N / 256), dim3(256), 0, 0, buf, T, plane, pitch);
        CK(hipEventRecord(ev_b));
        CK(hipEventSynchronize(ev_b));
        float ms;
        CK(hipEventElapsedTime(&ms, ev_a, ev_b));
        if (r && ms / 3 < best) best = ms / 3;
    }
    CK(hipGetLastError());
    return best * 1e3f;
}

int main()
{
    CK(hipEventCreate(&ev_a));
    CK(hipEventCreate(&ev_b));
    const size_t phys = 4ull << 30;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemGenericAllocationHandle_t handle;
    void* va = nullptr;
    CK(hipMemCreate(&handle, phys, &prop, 0));
    CK(hipMemAddressReserve(&va, phys, phys, nullptr, 0));
    CK(hipMemMap(va, phys, 0, handle, 0));
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, phys, &acc, 1));
    char* base = (char*)va;
    const size_t K = 1024, M = 1024 * K, row = (size_t)N * 4, plane0 = row * T;  // 250 MiB
    printf("block %p; row %zu KiB, plane %zu MiB\n", va, row / K, plane0 / M);
    const size_t extra[] = {0, 4 * K, 64 * K, 256 * K, 1 * M, 2 * M, 6 * M, 16 * M, 50 * M, 100 * M, 250 * M, 774 * M, 1000 * M};
    for (size_t x : extra) printf("plane stride 250 MiB + %8zu KiB : %.1f us\n", x / K, probe(base, plane0 + x, row));
    const size_t pitches[] = {row, row + 256, row + 4 * K, row + 64 * K, 2 * row, 3 * row / 2};
    for (size_t p : pitches) printf("row pitch %7zu B (planes %zu MiB apart) : %.1f us\n", p, (p * T + 2 * M - 1) / (2 * M) * 2, probe(base, (p * T + 2 * M - 1) / (2 * M) * (2 * M), p));
    const size_t offs[] = {0, 2 * M, 250 * M, 512 * M, 1000 * M, 1024 * M, 2048 * M, 3000 * M};
    for (size_t o : offs) printf("base offset %5zu MiB : %.1f us\n", o / M, probe(base + o, plane0, row));
    // one plane only, and two planes
    return 0;
}
