// store_ceiling.hip -- what can 65 536 lanes (1024 waves, ONE per SIMD) write per second in the
// rollout kernel's store shape?  Tuning aid, not part of the product.
//   dword   : per step, 3 x global_store_dword at [t][e]            (256 B per wave-instruction)
//   x4      : per 4 steps, 3 x global_store_dwordx4 at [t+j][4g..]  (4 x 256 B per wave-instruction)
//   *_nt    : same with non-temporal stores
// hipcc --offload-arch=gfx950 -O3 -o store_ceiling store_ceiling.hip && ./store_ceiling
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <bool NT>
__global__ void k_dword(int* __restrict__ a, int* __restrict__ b, int* __restrict__ c, int N, int T)
{
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    int s = e;
    for (int t = 0; t < T; ++t) {
        s = s * 1664525 + 1013904223;  // a little dependent integer work
        const size_t o = (size_t)t * N + e;
        if (NT) { __builtin_nontemporal_store(s, a + o); __builtin_nontemporal_store(s >> 3, b + o); __builtin_nontemporal_store(s & 1, c + o); }
        else { a[o] = s; b[o] = s >> 3; c[o] = s & 1; }
    }
}

template <bool NT>
__global__ void k_x4(int* __restrict__ a, int* __restrict__ b, int* __restrict__ c, int N, int T)
{
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned j = e & 3, e0 = e & ~3u;
    int s = e;
    for (int t = 0; t < T; t += 4) {
        int4 v;
        s = s * 1664525 + 1013904223; v.x = s;
        s = s * 1664525 + 1013904223; v.y = s;
        s = s * 1664525 + 1013904223; v.z = s;
        s = s * 1664525 + 1013904223; v.w = s;
        const size_t o = (size_t)(t + j) * N + e0;  // lane j of each quad writes row t+j, 4 envs wide
        int4 w = make_int4(v.x >> 3, v.y >> 3, v.z >> 3, v.w >> 3), u = make_int4(v.x & 1, v.y & 1, v.z & 1, v.w & 1);
        if (NT) {
            v4i nv = {v.x, v.y, v.z, v.w}, nw = {w.x, w.y, w.z, w.w}, nu = {u.x, u.y, u.z, u.w};
            __builtin_nontemporal_store(nv, (v4i*)(a + o)); __builtin_nontemporal_store(nw, (v4i*)(b + o)); __builtin_nontemporal_store(nu, (v4i*)(c + o));
        } else { *(int4*)(a + o) = v; *(int4*)(b + o) = w; *(int4*)(c + o) = u; }
    }
}

int main(int argc, char** argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 65536, T = 1000, reps = 20;
    int *a, *b, *c;
    const size_t bytes = (size_t)N * T * 4;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct V { const char* name; int kind; bool nt; int bs; };
    std::vector<V> vs;
    for (int bs : {64, 128, 256}) for (int kind : {0, 1}) for (bool nt : {false, true}) vs.push_back({kind ? "x4" : "dword", kind, nt, bs});
    for (int round = 0; round < 2; ++round)
    for (auto& v : vs) {
        dim3 g(N / v.bs), blk(v.bs);
        auto launch = [&]() {
            if (v.kind == 0) { if (v.nt) k_dword<true><<<g, blk>>>(a, b, c, N, T); else k_dword<false><<<g, blk>>>(a, b, c, N, T); }
            else { if (v.nt) k_x4<true><<<g, blk>>>(a, b, c, N, T); else k_x4<false><<<g, blk>>>(a, b, c, N, T); }
        };
        for (int i = 0; i < 3; ++i) launch();
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("N=%d bs=%3d %-5s nt=%d : %.4f ms/launch  %.2f TB/s\n", N, v.bs, v.name, (int)v.nt, ms / reps, 3.0 * bytes / (ms / reps * 1e-3) / 1e12);
    }
    return 0;
}
