// store_pitch.hip -- does the ROW PITCH of the [T][N] trajectory planes matter?  65 536 lanes write 1000 rows of three
// planes (the rollout kernel's store shape); the pitch between rows is N + pad elements.  With pad = 0 the pitch is
// 256 KiB, a power of two: every row starts on the same HBM channel / bank pattern.  50 back-to-back launches per
// variant, interleaved rounds.  Tuning aid, not part of the product.
// hipcc --offload-arch=gfx950 -O3 -o store_pitch store_pitch.hip && ./store_pitch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void __launch_bounds__(256) k(int *a, int *b, int *c, int N, int pitch, int T)
{
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (unsigned)N) return;
    int s = e;
    char *pa = (char *)a, *pb = (char *)b, *pc = (char *)c;
    const unsigned e4 = e * 4u, row = (unsigned)pitch * 4u;
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

int main(int argc, char **argv)
{
    const int T = 1000, reps = 50;
    const size_t cap = (size_t)(131072 + 32768) * T * 4;
    int *a, *b, *c;
    CK(hipMalloc(&a, cap));
    CK(hipMalloc(&b, cap));
    CK(hipMalloc(&c, cap));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    struct V { int N, pad; };
    const V vs[] = {{65536, 0}, {65536, 64}, {65536, 256}, {65536, 1024}, {65536, 4096}, {65536, 16384}, {65536, 16448},
                    {49152, 0}, {49152, 16384}, {57344, 0}, {81920, 0}, {98304, 0}, {131072, 0}, {131072, 4096}};
    if (argc > 2) {  // store_pitch N pad pad pad ...: scan pads for one batch size
        const int N = atoi(argv[1]);
        for (int round = 0; round < 2; ++round)
            for (int i = 2; i < argc; ++i) {
                const int pad = atoi(argv[i]), pitch = N + pad;
                for (int w = 0; w < 3; ++w) k<<<(N + 255) / 256, 256>>>(a, b, c, N, pitch, T);
                CK(hipEventRecord(e0));
                for (int w = 0; w < reps; ++w) k<<<(N + 255) / 256, 256>>>(a, b, c, N, pitch, T);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                printf("N=%6d pad=%6d (pitch %8.1f KiB) : %7.2f us/launch  %.2f TB/s\n", N, pad, pitch * 4 / 1024.0, ms / reps * 1e3,
                       3.0 * N * T * 4 / (ms / reps * 1e-3) / 1e12);
            }
        return 0;
    }
    for (int round = 0; round < 2; ++round)
        for (const V &v : vs) {
            const int pitch = v.N + v.pad;
            for (int i = 0; i < 3; ++i) k<<<(v.N + 255) / 256, 256>>>(a, b, c, v.N, pitch, T);
            CK(hipEventRecord(e0));
            for (int i = 0; i < reps; ++i) k<<<(v.N + 255) / 256, 256>>>(a, b, c, v.N, pitch, T);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            printf("N=%6d pad=%5d (pitch %7d el = %8.1f KiB) : %7.2f us/launch  %.2f TB/s\n", v.N, v.pad, pitch, pitch * 4 / 1024.0,
                   ms / reps * 1e3, 3.0 * v.N * T * 4 / (ms / reps * 1e-3) / 1e12);
        }
    return 0;
}
