// store_policy.hip -- does a cache-policy modifier on the trajectory stores change the SUSTAINED write rate?
// 65 536 lanes, 1000 x 3 buffer_store_dword at [t][e] (the rollout kernel's shape), 20 back-to-back launches per
// variant; aux bits of the raw buffer store on gfx950: 1 = sc0, 2 = nt, 16 = sc1.  Tuning aid, not part of the product.
// hipcc --offload-arch=gfx950 -O3 -o store_policy store_policy.hip && ./store_policy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int AUX>
__global__ void __launch_bounds__(256) k(int *a, int *b, int *c, int N, int T)
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
            s = s * 1664525 + 1013904223;  // a little dependent integer work
            __builtin_amdgcn_raw_buffer_store_b32(s, ra, e4, j * row, AUX);
            __builtin_amdgcn_raw_buffer_store_b32(s >> 3, rb, e4, j * row, AUX);
            __builtin_amdgcn_raw_buffer_store_b32(s & 1, rc, e4, j * row, AUX);
        }
        pa += 8 * (size_t)row, pb += 8 * (size_t)row, pc += 8 * (size_t)row;
    }
}

template <int AUX>
static void run(int *a, int *b, int *c, int N, int T, size_t bytes)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int reps = 20;
    for (int i = 0; i < 3; ++i) k<AUX><<<N / 256, 256>>>(a, b, c, N, T);
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) k<AUX><<<N / 256, 256>>>(a, b, c, N, T);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("aux=%2d (%s%s%s) : %.2f us/launch  %.2f TB/s\n", AUX, AUX & 1 ? "sc0 " : "", AUX & 2 ? "nt " : "", AUX & 16 ? "sc1" : "",
           ms / reps * 1e3, 3.0 * bytes / (ms / reps * 1e-3) / 1e12);
}

int main()
{
    const int N = 65536, T = 1000;
    int *a, *b, *c;
    const size_t bytes = (size_t)N * T * 4;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMalloc(&c, bytes));
    for (int round = 0; round < 2; ++round) {
        run<0>(a, b, c, N, T, bytes);
        run<1>(a, b, c, N, T, bytes);
        run<2>(a, b, c, N, T, bytes);
        run<3>(a, b, c, N, T, bytes);
        run<16>(a, b, c, N, T, bytes);
        run<17>(a, b, c, N, T, bytes);
        run<18>(a, b, c, N, T, bytes);
        run<19>(a, b, c, N, T, bytes);
    }
    return 0;
}
