// store_deadline.hip -- a rate limiter that keeps the waves ON A SCHEDULE instead of idling a fixed amount (tuning evidence, not
// product code; round 3)
//
// write_ceiling.hip: rate-limited, the bare store stream of the bench launch sustains 7.4 .. 7.6 TB/s on every buffer while it is
// blocked at its stores half of the time; the product-shaped step sustains 6.7 .. 7.0 and only when it is never blocked (its best idle
// amount is the one at which the waves' own pace equals the memory's; one turn less and the launch collapses).  Idea: what a
// blocked product-shaped kernel loses is the lockstep of its waves (every wave idles the same amount whether it is ahead or behind,
// so waves that were held up stay behind, the rows in flight spread out, and the memory sees ever more rows at once).  A deadline
// does not have that problem: wave w may start group k (16 steps) no earlier than start_w + k * period on the shader clock, and
// does not wait at all when it is late.
//   turns    : gu_idle every 4 steps (the product's limiter)
//   deadline : one clock read per 16 steps (s_memtime / s_memrealtime) and s_sleep until the group's time has come
// The step is write_ceiling.hip's k_prod<15> (byte cell table, the product's arithmetic, 16 steps unrolled, scalar row offsets, sc1).
//   hipcc --offload-arch=gfx950 -O3 -o store_deadline store_deadline.hip && ./store_deadline [buffers]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void idle(uint32_t pace)
{
    uint32_t c;
    asm volatile("s_and_b32 %0, %1, 0xff\n s_cmp_eq_u32 %0, 0\n s_cbranch_scc1 2f\n 1:\n s_sub_u32 %0, %0, 1\n s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 1b\n 2:\n"
                 "s_lshr_b32 %0, %1, 8\n s_cmp_eq_u32 %0, 0\n s_cbranch_scc1 4f\n 3:\n s_sleep 1\n s_sub_u32 %0, %0, 1\n s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 3b\n 4:"
                 : "=&s"(c) : "s"(pace) : "scc", "memory");
}

// MODE 0: turns every 4 steps; 1: deadline per 16 steps on s_memtime; 2: deadline per 16 steps on s_memrealtime (100 MHz);
// 3: deadline per 4 steps on s_memtime
template <int MODE>
__device__ __forceinline__ uint64_t now()
{
    return MODE == 2 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime();
}

__device__ unsigned long long *g_marks;  // "spread": per wave, the 100 MHz clock when it has done half / all of its steps

template <int MODE>
__global__ void __launch_bounds__(256) k_prod(int *base, size_t plane, int N, int T, uint32_t pace_in, unsigned long long *ticks)
{
    uint32_t pace = pace_in;
    __shared__ __attribute__((aligned(16))) uint8_t cell[2048];
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) {
        const int x = i & 31, y = i >> 5;
        const bool term = (x == 31 && y == 31) || (x == 16 && y > 3 && y < 28);
        uint8_t open = (y > 0 ? 1 : 0) | (x < 31 ? 2 : 0) | (y < 31 ? 4 : 0) | (x > 0 ? 8 : 0);
        if (((i * 2654435761u) >> 24) < 64 && i > 0) open &= 0x5;
        cell[i] = term ? 16 : open;
        cell[1024 + i] = (uint8_t)(int8_t)(term ? 10 : -1);
    }
    __syncthreads();
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x, off = e * 4u;
    const uint64_t lut = (uint64_t)(uint16_t)(int16_t)(-32) | (1ull << 16) | (32ull << 32) | (0xFFFFull << 48);
    int s = 0, r = -1, d = 0;
    uint32_t flags = cell[0], word = 0;
    char *p = (char *)base;
    const uint32_t row = (uint32_t)N * 4u;
    const uint64_t t0 = now<MODE>();
    const uint32_t credit = (pace >> 20) * 8u;  // ticks the schedule starts in the past (experiment "credit": a launch begins by catching up)
    pace &= 0xFFFFFu;
    uint64_t due = t0 - credit;
    auto wait_until_due = [&]() {
        due += pace;
        while ((int64_t)(now<MODE>() - due) < 0) __builtin_amdgcn_s_sleep(1);
    };
    for (int t = 0; t < T; t += 16, p += (size_t)row * 16) {
        const __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(p, 0, 0xFFFFFFFFu, 0x00020000),
                                     rr = __builtin_amdgcn_make_buffer_rsrc(p + plane * 4, 0, 0xFFFFFFFFu, 0x00020000),
                                     rd = __builtin_amdgcn_make_buffer_rsrc(p + 2 * plane * 4, 0, 0xFFFFFFFFu, 0x00020000);
        uint32_t h = (e * 0x9E3779B9u) ^ (uint32_t)(t >> 4);
        h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
        word = h;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const uint32_t act = __builtin_amdgcn_ubfe(word, 2 * j, 2);
            const int delta = __builtin_amdgcn_sbfe((int)(uint32_t)(lut >> (act << 4)), 0, 16);
            const bool was_done = flags & 16u;
            s = was_done ? 0 : s;
            flags = was_done ? (uint32_t)cell[0] : flags;
            s = __mul24((int)__builtin_amdgcn_ubfe(flags, act, 1), delta) + s;
            flags = cell[s];
            r = (int8_t)cell[1024 + s];
            d = (int)((flags >> 4) & 1u);
            __builtin_amdgcn_raw_buffer_store_b32(s, ro, off, j * row, 16);
            __builtin_amdgcn_raw_buffer_store_b32(r, rr, off, j * row, 16);
            __builtin_amdgcn_raw_buffer_store_b32(d, rd, off, j * row, 16);
            if (MODE == 0 && (j & 3) == 3 && pace) idle(pace);
            if (MODE == 3 && (j & 3) == 3 && pace) wait_until_due();
        }
        if ((MODE == 1 || MODE == 2) && pace) wait_until_due();
        if (g_marks && t + 16 == (T / 32) * 16 && (threadIdx.x & 63) == 0) g_marks[2 * (e >> 6)] = __builtin_amdgcn_s_memrealtime();
    }
    if (g_marks && (threadIdx.x & 63) == 0) g_marks[2 * (e >> 6) + 1] = __builtin_amdgcn_s_memrealtime();
    if (e == 0) *ticks = now<MODE>() - t0;
}

int main(int argc, char **argv)
{
    const int N = 65536, T = 992;
    const int buffers = argc > 1 ? atoi(argv[1]) : 5;
    size_t plane = (size_t)N * 1000;  // (ints; + a skew when argv[2] = "skew")
    const size_t bytes = 3 * (size_t)N * T * 4;
    const bool skew_mode = argc > 2 && std::string(argv[2]) == "skew";
    std::vector<int *> bufs;
    for (int b = 0; b < buffers; ++b) {
        int *p = nullptr;
        if (hipMalloc(&p, 3 * plane * 4 + (64u << 20)) != hipSuccess) break;
        bufs.push_back(p);
    }
    unsigned long long *dticks;
    CK(hipMalloc(&dticks, 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto launch = [&](int mode, int *buf, uint32_t pace) {
        const dim3 g(N / 256), b(256);
        if (mode == 0) k_prod<0><<<g, b>>>(buf, plane, N, T, pace, dticks);
        else if (mode == 1) k_prod<1><<<g, b>>>(buf, plane, N, T, pace, dticks);
        else if (mode == 2) k_prod<2><<<g, b>>>(buf, plane, N, T, pace, dticks);
        else k_prod<3><<<g, b>>>(buf, plane, N, T, pace, dticks);
    };
    auto timed = [&](int mode, int *buf, uint32_t pace, int reps) {
        launch(mode, buf, pace);
        launch(mode, buf, pace);
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) launch(mode, buf, pace);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms / reps * 1e3f;
    };
    for (int i = 0; i < 300; ++i) launch(0, bufs[0], 0u);
    CK(hipDeviceSynchronize());
    printf("%zu buffers; %.0f MB per launch; us per launch (TB/s)\n", bufs.size(), bytes / 1e6);
    const char *names[] = {"turns per 4 steps", "deadline per 16 steps, s_memtime", "deadline per 16 steps, s_memrealtime", "deadline per 4 steps, s_memtime"};
    if (argc > 2 && std::string(argv[2]) == "credit") {
        // the schedule of every launch begins `credit` in the past: its first groups run unthrottled until they have caught up, which
        // is what a schedule that continues across launches would do after the gap between two launches
        const int credits[] = {0, 100, 200, 300, 400, 600};  // ticks of 10 ns
        for (size_t b = 0; b < bufs.size(); ++b) {
            int *buf = bufs[b];
            const float t0 = timed(0, buf, 0u, 8);
            launch(2, buf, 0u);
            CK(hipDeviceSynchronize());
            unsigned long long ticks = 0;
            CK(hipMemcpy(&ticks, dticks, 8, hipMemcpyDeviceToHost));
            const double per_group = (double)ticks / (T / 16);
            printf("buffer %zu: unpaced %.1f us; per credit (us) and period: us per launch over 30 launches\n", b, t0);
            for (int credit : credits) {
                printf("  credit %3.0f us:", credit / 100.0);
                for (double f = 1.12; f >= 0.959; f -= 0.02) {
                    const uint32_t period = (uint32_t)(per_group * f + 0.5);
                    const float t = timed(2, buf, period | ((uint32_t)(credit / 8) << 20), 30);
                    printf(" %u:%.1f", period, t);
                }
                printf("\n");
            }
            fflush(stdout);
        }
        return 0;
    }
    if (argc > 2 && std::string(argv[2]) == "pitch") {
        // the row pitch: 65 536 columns x 4 B = 2^18 B (the product's) against padded pitches -- does the power-of-two distance between
        // the rows in flight cost write rate?  Schedule limiter, period scanned, 30 launches per figure
        const int pads[] = {0, 64, 256, 1024, 4096};  // ints (3 planes x 1000 rows x pad must fit the 64 MB of slack behind each buffer)
        printf("%-10s", "pad B");
        for (size_t b = 0; b < bufs.size(); ++b) printf("  buffer %zu: unpaced -> best @period", b);
        printf("\n");
        for (int pad : pads) {
            const int pitch = N + pad;
            plane = (size_t)pitch * 1000;
            printf("%-10d", pad * 4);
            for (int *buf : bufs) {
                auto launch_p = [&](int mode, uint32_t pace) {
                    if (mode == 0) k_prod<0><<<dim3(N / 256), dim3(256)>>>(buf, plane, pitch, T, pace, dticks);
                    else k_prod<2><<<dim3(N / 256), dim3(256)>>>(buf, plane, pitch, T, pace, dticks);
                };
                auto timed_p = [&](int mode, uint32_t pace, int reps) {
                    launch_p(mode, pace);
                    launch_p(mode, pace);
                    CK(hipEventRecord(e0));
                    for (int i = 0; i < reps; ++i) launch_p(mode, pace);
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    CK(hipGetLastError());
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    return ms / reps * 1e3f;
                };
                const float t0 = timed_p(0, 0u, 8);
                launch_p(2, 0u);
                CK(hipDeviceSynchronize());
                unsigned long long ticks = 0;
                CK(hipMemcpy(&ticks, dticks, 8, hipMemcpyDeviceToHost));
                const double per_group = (double)ticks / (T / 16);
                float best = 1e9f;
                double at = 0;
                for (double f = 1.16; f >= 0.939; f -= 0.02) {
                    const float t = timed_p(2, (uint32_t)(per_group * f + 0.5), 30);
                    if (t < best) best = t, at = per_group * f;
                }
                printf("  %6.1f -> %6.1f @%3.0f       ", t0, best, at);
            }
            printf("\n");
            fflush(stdout);
        }
        return 0;
    }
    if (argc > 2 && std::string(argv[2]) == "spread") {
        // how far apart are the waves?  Per wave the clock at half time and at the end; per setting the range (max - min) over the
        // 1024 waves of the last of 12 launches, in us and in rows (a row of all three planes is written every launch time / 992)
        unsigned long long *marks = nullptr;
        CK(hipMalloc(&marks, 2 * 1024 * 8));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_marks), &marks, sizeof marks));
        std::vector<unsigned long long> hm(2048);
        for (size_t b = 0; b < bufs.size(); ++b) {
            int *buf = bufs[b];
            launch(2, buf, 0u);
            CK(hipDeviceSynchronize());
            unsigned long long ticks = 0;
            CK(hipMemcpy(&ticks, dticks, 8, hipMemcpyDeviceToHost));
            const double per_group = (double)ticks / (T / 16);
            struct { const char *name; int mode; uint32_t pace; } settings[] = {
                {"unthrottled", 0, 0u}, {"idle turns 10 per 4 steps", 0, 10u}, {"idle turns 8", 0, 8u},
                {"schedule 1.08", 2, (uint32_t)(per_group * 1.08)}, {"schedule 1.02", 2, (uint32_t)(per_group * 1.02)}, {"schedule 0.94 (too short)", 2, (uint32_t)(per_group * 0.94)}};
            printf("buffer %zu\n", b);
            for (auto &st : settings) {
                const float t = timed(st.mode, buf, st.pace, 12);
                CK(hipMemcpy(hm.data(), marks, 2048 * 8, hipMemcpyDeviceToHost));
                unsigned long long lo[2] = {~0ull, ~0ull}, hi[2] = {0, 0};
                for (int w = 0; w < 1024; ++w)
                    for (int k = 0; k < 2; ++k) {
                        lo[k] = std::min(lo[k], hm[2 * w + k]);
                        hi[k] = std::max(hi[k], hm[2 * w + k]);
                    }
                const double us_mid = (hi[0] - lo[0]) / 100.0, us_end = (hi[1] - lo[1]) / 100.0, row_us = t / T;
                printf("  %-28s %6.1f us per launch; waves apart at half time %5.1f us = %4.0f rows, at the end %5.1f us = %4.0f rows\n", st.name, t, us_mid, us_mid / row_us,
                       us_end, us_end / row_us);
            }
            fflush(stdout);
        }
        return 0;
    }
    if (skew_mode) {
        // the three planes 250 MiB apart (the product's layout) against planes skewed by a few KB .. MB: does it matter which
        // channels the three rows of a step fall on?  Schedule limiter (s_memrealtime per 16 steps), period scanned, 30 launches each
        const size_t skews[] = {0, 1024, 4096, 4096 + 256, 16384 + 1024, 65536 + 4096, 262144 + 16384, (1u << 20) + 65536, (2u << 20) / 3 / 256 * 256, (5u << 20) + 4096};
        printf("%-10s", "skew B");
        for (size_t b = 0; b < bufs.size(); ++b) printf("  buffer %zu: unpaced -> best @period", b);
        printf("\n");
        for (size_t skew : skews) {
            plane = (size_t)N * 1000 + skew / 4;
            printf("%-10zu", skew);
            for (int *buf : bufs) {
                const float t0 = timed(0, buf, 0u, 8);
                launch(2, buf, 0u);
                CK(hipDeviceSynchronize());
                unsigned long long ticks = 0;
                CK(hipMemcpy(&ticks, dticks, 8, hipMemcpyDeviceToHost));
                const double per_group = (double)ticks / (T / 16);
                float best = 1e9f;
                double at = 0;
                for (double f = 1.16; f >= 0.939; f -= 0.02) {
                    const float t = timed(2, buf, (uint32_t)(per_group * f + 0.5), 30);
                    if (t < best) best = t, at = per_group * f;
                }
                printf("  %6.1f -> %6.1f @%3.0f       ", t0, best, at);
            }
            printf("\n");
            fflush(stdout);
        }
        return 0;
    }
    for (size_t b = 0; b < bufs.size(); ++b) {
        int *buf = bufs[b];
        const float t0 = timed(0, buf, 0u, 8);
        printf("buffer %zu: unpaced %.1f us\n", b, t0);
        {  // the product's limiter: sustained time per idle amount
            printf("  %-40s", names[0]);
            float best = 1e9f;
            int at = 0;
            for (int turns = 14; turns >= 6; --turns) {
                const float t = timed(0, buf, (uint32_t)turns, 30);
                printf(" %d:%.1f", turns, t);
                if (t < best) best = t, at = turns;
            }
            printf("  -> %.1f @%d (%.2f)\n", best, at, bytes / (best * 1e-6) / 1e12);
        }
        for (int mode = 1; mode <= 3; ++mode) {
            // ticks of an unpaced launch on this clock -> ticks per group at that rate; scan the period downwards from 1.05 x
            launch(mode, buf, 0u);
            CK(hipDeviceSynchronize());
            unsigned long long ticks = 0;
            CK(hipMemcpy(&ticks, dticks, 8, hipMemcpyDeviceToHost));
            const int groups = mode == 3 ? T / 4 : T / 16;
            const double per_group = (double)ticks / groups;  // ticks per group of wave 0 in an unpaced launch
            printf("  %-40s (%llu ticks per unpaced launch)", names[mode], ticks);
            float best = 1e9f;
            double at = 0;
            for (double f = 1.30; f >= 0.699; f -= 0.02) {  // the period as a fraction of that
                const uint32_t period = (uint32_t)(per_group * f + 0.5);
                if (!period) continue;
                const float t = timed(mode, buf, period, 30);
                printf(" %.2f:%.1f", f, t);
                if (t < best) best = t, at = f;
            }
            printf("  -> %.1f @%.2f (%.2f)\n", best, at, bytes / (best * 1e-6) / 1e12);
        }
        fflush(stdout);
    }
    return 0;
}
