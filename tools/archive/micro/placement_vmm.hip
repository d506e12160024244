// placement_vmm.hip -- is the write-rate class of an allocation (store_placement.hip, placement_map.hip) a matter of how its
// virtual and physical addresses are aligned TO EACH OTHER (page-table fragment size -> TLB reach)?  The same store loop on
//   A  hipMalloc buffers (virtual addresses come 2 MiB-aligned),
//   B  virtual-memory-API buffers: physical handle of 1 GiB (a power of two: naturally aligned in a buddy allocator) mapped at a
//      1 GiB-aligned virtual address,
//   C  the same handle mapped 2 MiB off a 1 GiB boundary (virtual and physical congruent modulo 2 MiB only),
//   D  physical handle of exactly 750 MiB mapped at a 1 GiB-aligned address.
// hipcc --offload-arch=gfx950 -O3 -o placement_vmm placement_vmm.hip && ./placement_vmm [rounds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

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

struct Vmm {
    hipMemGenericAllocationHandle_t handle;
    void* va_base;
    size_t va_size, phys;
    int* ptr;
};

// physical size `phys`, mapped at (1 GiB-aligned address) + off
static bool vmm_alloc(size_t phys, size_t off, Vmm* out)
{
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    out->phys = phys;
    out->va_size = ((phys + off + (1ull << 30) - 1) >> 30) << 30;
    if (hipMemCreate(&out->handle, phys, &prop, 0) != hipSuccess) return false;
    if (hipMemAddressReserve(&out->va_base, out->va_size, 1ull << 30, nullptr, 0) != hipSuccess) return false;
    if (hipMemMap((char*)out->va_base + off, phys, 0, out->handle, 0) != hipSuccess) return false;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess((char*)out->va_base + off, phys, &acc, 1) != hipSuccess) return false;
    out->ptr = (int*)((char*)out->va_base + off);
    return true;
}

int main(int argc, char** argv)
{
    const int rounds = argc > 1 ? atoi(argv[1]) : 10;
    const size_t bytes = (size_t)3 * N * T * 4;
    CK(hipEventCreate(&ev_a));
    CK(hipEventCreate(&ev_b));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("buffer %zu bytes = %.1f MiB; recommended granularity %zu\n", bytes, bytes / 1048576.0, gran);
    std::vector<void*> keep;
    std::vector<Vmm> keepv;
    for (int i = 0; i < rounds; ++i) {
        int* a = nullptr;
        CK(hipMalloc(&a, bytes));
        keep.push_back(a);
        const float ta = probe(a);
        Vmm b, c, d;
        float tb = -1, tc = -1, td = -1;
        if (vmm_alloc(1ull << 30, 0, &b)) { tb = probe(b.ptr); keepv.push_back(b); } else (void)hipGetLastError();
        if (vmm_alloc(1ull << 30, 2ull << 20, &c)) { tc = probe(c.ptr); keepv.push_back(c); } else (void)hipGetLastError();
        if (vmm_alloc(bytes, 0, &d)) { td = probe(d.ptr); keepv.push_back(d); } else (void)hipGetLastError();
        printf("round %2d  A hipMalloc %p %.1f us   B vmm 1 GiB aligned %.1f us   C vmm 1 GiB at +2 MiB %.1f us   D vmm 750 MiB aligned %.1f us\n", i, (void*)a, ta, tb, tc,
               td);
        fflush(stdout);
        // a spacer so that successive rounds land in different neighbourhoods
        void* sp = nullptr;
        if (hipMalloc(&sp, (size_t)3 << 30) == hipSuccess) keep.push_back(sp); else (void)hipGetLastError();
    }
    return 0;
}
