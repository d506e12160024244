// store_cumask.hip -- is the sustained write rate of the rollout's store shape higher with FEWER CUs writing?
// (49 152 lanes = 192 workgroups reach 6.2 TB/s where 65 536 lanes = 256 workgroups reach 5.7.)  The same 65 536-lane
// launch on streams whose CU mask enables 256 / 224 / 192 / 160 / 128 CUs.  Tuning aid, not part of the product.
// hipcc --offload-arch=gfx950 -O3 -o store_cumask store_cumask.hip && ./store_cumask
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(1024) k(int *a, int *b, int *c, int N, int T)
{
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    int s = e;
    char *pa = (char *)a, *pb = (char *)b, *pc = (char *)c;
    const unsigned e4 = e * 4u, row = (unsigned)N * 4u;
    for (int t = 0; t < T; t += 8) {
        __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(pa, 0, 0xFFFFFFFFu, 0x00020000);
        __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(pb, 0, 0xFFFFFFFFu, 0x00020000);
        __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(pc, 0, 0xFFFFFFFFu, 0x00020000);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s = s * 1664525 + 1013904223;
            __builtin_amdgcn_raw_buffer_store_b32(s, ra, e4, j * row, 0);
            __builtin_amdgcn_raw_buffer_store_b32(s >> 3, rb, e4, j * row, 0);
            __builtin_amdgcn_raw_buffer_store_b32(s & 1, rc, e4, j * row, 0);
        }
        pa += 8 * (size_t)row, pb += 8 * (size_t)row, pc += 8 * (size_t)row;
    }
}

int main()
{
    const int N = 65536, T = 1000, reps = 50;
    const size_t bytes = (size_t)N * T * 4;
    int *a, *b, *c;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMalloc(&c, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    struct M { const char *name; uint32_t word; };
    // the same 32-bit pattern in all eight mask words: whatever the bit -> (XCD, CU) mapping is, every eighth of the
    // mask loses the same share
    const M ms[] = {{"256 CUs", 0xFFFFFFFFu}, {"224 CUs", 0x0FFFFFFFu}, {"192 CUs", 0x00FFFFFFu}, {"160 CUs", 0x000FFFFFu}, {"128 CUs", 0x0000FFFFu}};
    std::vector<hipStream_t> streams;
    for (const M &m : ms) {
        uint32_t mask[8];
        for (int i = 0; i < 8; ++i) mask[i] = m.word;
        hipStream_t s;
        CK(hipExtStreamCreateWithCUMask(&s, 8, mask));
        streams.push_back(s);
    }
    for (int round = 0; round < 3; ++round)
        for (size_t v = 0; v < streams.size(); ++v) {
            hipStream_t s = streams[v];
            for (int i = 0; i < 3; ++i) k<<<N / 256, 256, 0, s>>>(a, b, c, N, T);
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < reps; ++i) k<<<N / 256, 256, 0, s>>>(a, b, c, N, T);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float msf;
            CK(hipEventElapsedTime(&msf, e0, e1));
            printf("%s : %7.2f us/launch  %.2f TB/s\n", ms[v].name, msf / reps * 1e3, 3.0 * bytes / (msf / reps * 1e-3) / 1e12);
        }
    printf("-- no CU mask, workgroup size (65 536 lanes: 256 / 128 / 64 workgroups; the dispatcher gives each its own CU)\n");
    const int sizes[] = {256, 512, 1024};
    for (int round = 0; round < 3; ++round)
        for (int bs : sizes) {
            for (int i = 0; i < 3; ++i) k<<<N / bs, bs>>>(a, b, c, N, T);
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) k<<<N / bs, bs>>>(a, b, c, N, T);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float msf;
            CK(hipEventElapsedTime(&msf, e0, e1));
            printf("block %4d : %7.2f us/launch  %.2f TB/s\n", bs, msf / reps * 1e3, 3.0 * bytes / (msf / reps * 1e-3) / 1e12);
        }
    return 0;
}
