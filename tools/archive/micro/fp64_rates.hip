// Issue rate and dependent latency of the float64 VALU operations the DP / Monte-Carlo kernels are made of (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o fp64_rates fp64_rates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// OP: 0 add, 1 mul, 2 fma, 3 cvt f32->f64 (+add to keep it alive), 4 div.  CH independent chains per lane.
template <int OP, int CH>
__global__ void __launch_bounds__(256) k(double *out, const float *fin, int iters)
{
    double acc[CH];
    float f[CH];
    for (int c = 0; c < CH; ++c) {
        acc[c] = 1.0 + threadIdx.x * 1e-9 + c;
        f[c] = fin[(threadIdx.x + c) & 255];
    }
    const double m = 1.0000000001, a = 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            if (OP == 0) acc[c] = __dadd_rn(acc[c], a);
            if (OP == 1) acc[c] = __dmul_rn(acc[c], m);
            if (OP == 2) acc[c] = __fma_rn(acc[c], m, a);
            if (OP == 3) {
                double d;
                asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d) : "v"(f[c]));
                acc[c] = __longlong_as_double(__double_as_longlong(acc[c]) ^ __double_as_longlong(d));
            }
            if (OP == 4) acc[c] = __ddiv_rn(acc[c], m);
        }
    }
    double s = 0;
    for (int c = 0; c < CH; ++c) s += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP, int CH>
static void run(const char *name, int blocks_per_cu, double *out, float *fin)
{
    const int iters = OP == 4 ? 2000 : 20000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<OP, CH>), dim3(blocks), dim3(256), 0, 0, out, fin, 10);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<OP, CH>), dim3(blocks), dim3(256), 0, 0, out, fin, iters);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    // wave-instructions per SIMD = blocks_per_cu (one wave of each block lands on each SIMD) * iters * CH
    const double ns_per_wave_instr = ms * 1e6 / ((double)blocks_per_cu * iters * CH);
    printf("%-8s chains=%d waves/SIMD=%d  %.2f ms  %.2f ns per wave-instruction per SIMD  (%.1f cycles at 2.4 GHz)\n", name, CH,
           blocks_per_cu, ms, ns_per_wave_instr, ns_per_wave_instr * 2.4);
}

int main()
{
    double *out;
    float *fin;
    CHECK(hipMalloc(&out, 256 * 8 * 256 * sizeof(double)));
    CHECK(hipMalloc(&fin, 256 * sizeof(float)));
    CHECK(hipMemset(fin, 0, 256 * sizeof(float)));
    printf("-- one wave per SIMD, one dependent chain: latency\n");
    run<0, 1>("add", 1, out, fin);
    run<1, 1>("mul", 1, out, fin);
    run<2, 1>("fma", 1, out, fin);
    run<3, 1>("cvt+xor", 1, out, fin);
    run<4, 1>("div", 1, out, fin);
    printf("-- one wave per SIMD, 8 independent chains: issue rate\n");
    run<0, 8>("add", 1, out, fin);
    run<1, 8>("mul", 1, out, fin);
    run<2, 8>("fma", 1, out, fin);
    run<3, 8>("cvt+xor", 1, out, fin);
    printf("-- 8 waves per SIMD, 8 chains\n");
    run<0, 8>("add", 8, out, fin);
    run<1, 8>("mul", 8, out, fin);
    run<2, 8>("fma", 8, out, fin);
    run<3, 8>("cvt+xor", 8, out, fin);
    return 0;
}
