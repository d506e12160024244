// Issue rate and dependent latency of the 32-bit integer VALU operations the per-env RNG (MurmurHash3) is made of (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -o int_rates int_rates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// OP: 0 v_mul_lo_u32, 1 v_mad_u64_u32 (h * 5 + c as the compiler emits it), 2 v_lshl_add_u32 + v_add (h * 5 + c by shift),
//     3 v_mul_u32_u24, 4 v_alignbit (rotate), 5 v_xor, 6 the whole MurmurHash3 word (block + finaliser)
template <int OP, int CH>
__global__ void __launch_bounds__(256) k(uint32_t *out, uint32_t seed, int iters)
{
    uint32_t acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = seed + threadIdx.x * 2654435761u + c;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            uint32_t x = acc[c];
            if (OP == 0) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(x) : "v"(x), "s"(0xCC9E2D51u));
            if (OP == 1) { uint64_t r; asm volatile("v_mad_u64_u32 %0, vcc, %1, 5, %2" : "=v"(r) : "v"(x), "v"((uint64_t)0xE6546B64u) : "vcc"); x = (uint32_t)r; }
            if (OP == 2) { asm volatile("v_lshl_add_u32 %0, %1, 2, %1" : "=v"(x) : "v"(x)); asm volatile("v_add_u32 %0, %1, %2" : "=v"(x) : "v"(x), "s"(0xE6546B64u)); }
            if (OP == 3) asm volatile("v_mul_u32_u24 %0, %1, %2" : "=v"(x) : "v"(x), "s"(0x9E2D51u));
            if (OP == 4) asm volatile("v_alignbit_b32 %0, %1, %1, 17" : "=v"(x) : "v"(x));
            if (OP == 5) asm volatile("v_xor_b32 %0, %1, %2" : "=v"(x) : "v"(x), "s"(0x85EBCA6Bu));
            if (OP == 6) {
                uint32_t kx = x * 0xCC9E2D51u;
                kx = (kx << 15) | (kx >> 17);
                kx *= 0x1B873593u;
                uint32_t h = seed ^ kx;
                h = (h << 13) | (h >> 19);
                h = h * 5u + 0xE6546B64u;
                h ^= 16u;
                h ^= h >> 16;
                h *= 0x85EBCA6Bu;
                h ^= h >> 13;
                h *= 0xC2B2AE35u;
                h ^= h >> 16;
                x = h;
            }
            acc[c] = x;
        }
    }
    uint32_t s = 0;
    for (int c = 0; c < CH; ++c) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP, int CH>
static void run(const char *name, int blocks_per_cu, uint32_t *out)
{
    const int iters = 20000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<OP, CH>), dim3(blocks), dim3(256), 0, 0, out, 123u, 10);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<OP, CH>), dim3(blocks), dim3(256), 0, 0, out, 123u, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double ns = ms * 1e6 / ((double)blocks_per_cu * iters * CH);
    printf("%-28s chains=%d waves/SIMD=%d  %.2f ns per wave-op per SIMD  (%.1f cycles at 2.4 GHz)\n", name, CH, blocks_per_cu, ns, ns * 2.4);
}

int main()
{
    uint32_t *out;
    CHECK(hipMalloc(&out, 256 * 8 * 256 * sizeof(uint32_t)));
    printf("-- one wave per SIMD, one dependent chain: latency\n");
    run<0, 1>("v_mul_lo_u32", 1, out);
    run<1, 1>("v_mad_u64_u32", 1, out);
    run<2, 1>("v_lshl_add_u32 + v_add_u32", 1, out);
    run<3, 1>("v_mul_u32_u24", 1, out);
    run<4, 1>("v_alignbit_b32", 1, out);
    run<5, 1>("v_xor_b32", 1, out);
    run<6, 1>("MurmurHash3 word", 1, out);
    printf("-- one wave per SIMD, 8 independent chains: issue rate\n");
    run<0, 8>("v_mul_lo_u32", 1, out);
    run<1, 8>("v_mad_u64_u32", 1, out);
    run<2, 8>("v_lshl_add_u32 + v_add_u32", 1, out);
    run<3, 8>("v_mul_u32_u24", 1, out);
    run<4, 8>("v_alignbit_b32", 1, out);
    run<6, 8>("MurmurHash3 word", 1, out);
    return 0;
}
