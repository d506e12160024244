// gu_rollout_multi.hip -- K env-steps per LDS round trip: the uniform-policy (or caller-supplied-stream) rollout that keeps
// only per-env statistics (no trajectory), on a K-STEP transition table.
//
// gu_rollout_rows.hip brought a step down to one v_and_or_b32 + one ds_read_b32; at one wave per SIMD that round trip (~90
// clocks) IS the step.  The uniform policy's actions do not depend on the env state -- they are two-bit fields of a counter-RNG
// word (a caller-supplied stream is packed into the same shape by gu_upload_actions) -- so the transitions of K consecutive
// steps can be composed ahead of time:
//
//     rowK[s][a1 | a2 << 2 | ..] = { LDS byte address of rowK[cell after the K steps]  : bits 0..17
//                                    number of done flags raised by the K steps        : bits 18..20
//                                    done flag of the last of them                     : bit 23
//                                    sum of the K rewards, int8                        : bits 24..31 }
//
// with the lazy `if done: env.reset()` of the harness (core/algorithms/monte_carlo.py:19-25, env:187-193) folded into every
// one of the K steps exactly as in the one-step table.  K = 4 for grids of up to 64 cells (4^4 entries of 4 bytes per cell),
// K = 2 up to ~2000 cells; one round trip then advances an env by K steps, and the per-env return / episode count / final
// state are all that is kept.  Steps that do not fill a group -- the first step of a launch (on the per-cell planes: a stored
// state may disagree with its cell), the steps up to the next multiple of K of the env's step counter, the last T mod K --
// run on a one-step table of the same format that sits behind the K-step table in LDS.
// Results are bit-identical to the other two kernels (tests/test_gpu_kstep_kernel.py).
#include "gu_rollout.hpp"

#define GU_MROW_ADDR_MASK 0x3FFFFu
#define GU_MROW_DCOUNT_SHIFT 18
#define GU_MROW_DONE_BIT 23

struct BuildMultiArgs {
    const uint8_t *cell;  // absorbing-aware planes [flags | reward]
    int32_t cell_bytes, S, W, start0, auto_reset, row_shift;
    uint32_t *rows;       // [S][4^K]
};

template <int K>
__global__ void __launch_bounds__(256) gu_build_multi_rows_kernel(const BuildMultiArgs a)
{
    constexpr int G = 1 << (2 * K);
    const int32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= a.S * G) return;
    int32_t cur = idx >> (2 * K);
    const uint32_t g = (uint32_t)idx & (G - 1);
    int32_t rsum = 0;
    uint32_t dcount = 0, dlast = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const uint32_t act = (g >> (2 * k)) & 3u;
        const int32_t base = (a.auto_reset && (a.cell[cur] & GU_CELL_TERM)) ? a.start0 : cur;  // lazy `if done: env.reset()` (env:187-193)
        const uint32_t fb = a.cell[base];  // OPEN bits of the absorbing map: a terminal cell does not move (env:145-146)
        cur = base + (((fb >> act) & 1u) ? gu_delta<false>(act, 0, a.W) : 0);
        dlast = (a.cell[cur] >> GU_CELL_TERM_BIT) & 1u;
        dcount += dlast;
        rsum += (int8_t)a.cell[a.cell_bytes + cur];
    }
    a.rows[idx] = ((uint32_t)cur << a.row_shift) | (dcount << GU_MROW_DCOUNT_SHIFT) | (dlast << GU_MROW_DONE_BIT) | ((uint32_t)(uint8_t)(int8_t)rsum << 24);
}

typedef __attribute__((address_space(3))) const uint32_t *lds_u32_ptr;

template <int POLICY, int K, bool STATS>
__global__ void __launch_bounds__(GU_MAX_BLOCK) gu_rollout_multi_kernel(const RolloutArgs a, const int32_t auto_reset, const uint32_t *__restrict__ rows1,
                                                                        const int32_t shiftK)
{
    constexpr int ROWK_LOG2 = 2 * K + 2;  // bytes per cell and copy of the K-step table
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t baseK = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)smem;
    const uint32_t base1 = baseK + ((uint32_t)a.S << shiftK);
    const int32_t copies_log2 = shiftK - ROWK_LOG2;
    {
        // staging: 16-byte units, consecutive threads -> consecutive LDS addresses; the LDS base is folded into every entry
        uint4 *dst = reinterpret_cast<uint4 *>(smem);
        const uint4 *gK = reinterpret_cast<const uint4 *>(a.rows);
        const int32_t unitsK = (a.S << shiftK) >> 4;
#pragma unroll 8
        for (int32_t u = threadIdx.x; u < unitsK; u += blockDim.x) {
            const int32_t cell = u >> (shiftK - 4), within = u & ((1 << (ROWK_LOG2 - 4)) - 1);  // (the copy index drops out)
            uint4 v = gK[(cell << (ROWK_LOG2 - 4)) + within];
            v.x += baseK, v.y += baseK, v.z += baseK, v.w += baseK;
            dst[u] = v;
        }
        const uint4 *g1 = reinterpret_cast<const uint4 *>(rows1);
        uint4 *dst1 = reinterpret_cast<uint4 *>(smem + ((size_t)a.S << shiftK));
        for (int32_t u = threadIdx.x; u < a.S; u += blockDim.x) {
            uint4 v = g1[u];
            v.x += base1, v.y += base1, v.z += base1, v.w += base1;
            dst1[u] = v;
        }
        __syncthreads();
    }
    const int64_t e64 = (int64_t)gu_env_block(a.xcd_remap) * blockDim.x + threadIdx.x;
    if (e64 >= a.N) return;
    const uint32_t e = (uint32_t)e64;
    const uint32_t lane_copy = (threadIdx.x & ((1u << copies_log2) - 1u)) << ROWK_LOG2;

    int32_t s = a.pos[e];
    const uint32_t d_entry = (uint32_t)a.done[e];
    uint32_t ep = a.episode[e];
    uint32_t t = a.tcount[e] + a.steps_taken;
    const uint32_t prefix = gu_rng_prefix(a.seed_prefix, a.env_id0 + e);
    int32_t ret = 0;
    uint32_t fin = 0;

    // `rec` always holds the entry fetched last, NOT yet added to the statistics: they hang off the entry beside the chain and
    // are added after the next read has been issued
    uint32_t rec;
    auto account = [&](uint32_t x) {
        ret += (int32_t)x >> 24;
        fin += __builtin_amdgcn_ubfe(x, GU_MROW_DCOUNT_SHIFT, 3);
    };
    auto single = [&](uint32_t act) {  // one step on the one-step table
        const uint32_t prev = rec;
        uint32_t off = act << 2;
        asm("" : "+v"(off));
        rec = *(lds_u32_ptr)(uintptr_t)((prev & GU_MROW_ADDR_MASK) | off);
        account(prev);
    };
    // `between`: work that does not depend on the env state (half of the next RNG word), placed between the issue of the read
    // and the first use of its result
    auto group_with = [&](uint32_t g, auto between) {  // K steps on the K-step table
        const uint32_t prev = rec;
        uint32_t off = lane_copy | (g << 2);
        asm("" : "+v"(off));  // keep it ONE value off the chain: the chain is v_and_or_b32 + ds_read_b32
        rec = *(lds_u32_ptr)(uintptr_t)((prev & GU_MROW_ADDR_MASK) | off);
        between();
        account(prev);
    };
    auto group = [&](uint32_t g) { group_with(g, [] {}); };
    auto to_multi = [&] {
        const uint32_t cell = ((rec & GU_MROW_ADDR_MASK) - base1) >> 4;
        rec = (rec & ~GU_MROW_ADDR_MASK) | ((cell << shiftK) + baseK);
    };
    auto to_single = [&] {
        const uint32_t cell = ((rec & GU_MROW_ADDR_MASK) - baseK) >> shiftK;
        rec = (rec & ~GU_MROW_ADDR_MASK) | ((cell << 4) + base1);
    };

    const char *pa = (const char *)a.actions;  // GU_POLICY_STREAM: packed words [ceil(T / 16) + pad][N], always from row 0
    const uint32_t e4 = e * 4u;
    uint32_t word = POLICY == GU_POLICY_STREAM
                        ? (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(__builtin_amdgcn_make_buffer_rsrc((void *)pa, 0, 0xFFFFFFFFu, 0x00020000), e4, 0, 0)
                        : gu_rng_word(prefix, GU_RNG_STREAM_ACTION, t >> 4);
    {
        // first step of the launch, on the per-cell planes: the stored done flag decides the lazy reset (it may disagree with
        // the cell: fresh reset onto a terminal start, gu_set_state)
        const uint32_t act = POLICY == GU_POLICY_STREAM ? (word & 3u) : ((word >> (2u * (t & 15u))) & 3u);
        if (auto_reset && d_entry) s = a.starts[0];
        const uint32_t f0 = a.cell[s];
        s += ((f0 >> act) & 1u) ? gu_delta<false>(act, 0, a.W) : 0;
        const uint32_t dn = (a.cell[s] >> GU_CELL_TERM_BIT) & 1u;
        const uint32_t r = (uint8_t)a.cell[a.cell_bytes + s];
        rec = (((uint32_t)s << 4) + base1) | (dn << GU_MROW_DCOUNT_SHIFT) | (dn << GU_MROW_DONE_BIT) | (r << 24);
        ++t;
    }
    int32_t rem = (int32_t)a.T - 1;  // (32-bit: gu_rollout caps T at 1e8; a 64-bit count costs the loop's scalar unit two instructions per group)
    const uint32_t t_first = __builtin_amdgcn_readfirstlane(t);
    if (POLICY == GU_POLICY_STREAM) {
        // the caller's stream: row i of the stream is step i of the launch, so groups are aligned to the launch, and every
        // lane is at the same position; words arrive four ahead of their steps (gu_stream_run)
        bool in_multi = false;
        gu_stream_run<true>(
            pa, a.N * 4, e4, a.T, 1,
            [&](uint32_t w16) {
                if (!in_multi) to_multi();
                in_multi = true;
#pragma unroll
                for (uint32_t j = 0; j < 16 / K; ++j) group(__builtin_amdgcn_ubfe(w16, 2 * K * j, 2 * K));
            },
            [&](uint32_t act) {
                if (in_multi) to_single();
                in_multi = false;
                single(act);
            });
        if (in_multi) to_single();
    } else if (__all(t == t_first)) {
        // every lane of the wave is at the same step count (always, unless gu_set_state installed per-env counters): the
        // position inside the RNG word is wave-uniform, the bit-field offsets of the unrolled body are constants
        uint32_t tu = t_first, have = (tu - 1u) >> 4;  // `have`: index of the word held in `word`
        auto need = [&](uint32_t idx) {
            if (idx != have) {
                word = gu_rng_word(prefix, GU_RNG_STREAM_ACTION, idx);
                have = idx;
            }
        };
        for (; rem > 0 && (tu & (K - 1)); ++tu, --rem) {  // up to the next multiple of K
            need(tu >> 4);
            single((word >> (2u * (tu & 15u))) & 3u);
        }
        if (rem >= K) {
            to_multi();
            for (; rem >= K && (tu & 15u); tu += K, rem -= K) {  // the rest of the word the launch starts in
                need(tu >> 4);
                group((word >> (2u * (tu & 15u))) & ((1u << (2 * K)) - 1u));
            }
            if (rem >= 16) {
                uint32_t next = gu_rng_word(prefix, GU_RNG_STREAM_ACTION, tu >> 4);
                for (; rem >= 16; tu += 16, rem -= 16) {
                    word = next;
                    // the next word is hashed in the shadow of this word's first two round trips, half and half
                    uint32_t half = 0;
                    group_with(__builtin_amdgcn_ubfe(word, 0, 2 * K), [&] {
                        __builtin_amdgcn_sched_barrier(0);
                        half = gu_rng_word_begin(prefix, GU_RNG_STREAM_ACTION, (tu >> 4) + 1u);
                        __builtin_amdgcn_sched_barrier(0);
                    });
                    group_with(__builtin_amdgcn_ubfe(word, 2 * K, 2 * K), [&] {
                        __builtin_amdgcn_sched_barrier(0);
                        next = gu_rng_word_finish(half);
                        __builtin_amdgcn_sched_barrier(0);
                    });
#pragma unroll
                    for (uint32_t j = 2; j < 16 / K; ++j) group(__builtin_amdgcn_ubfe(word, 2 * K * j, 2 * K));
                }
                word = next;
                have = tu >> 4;
            }
            for (; rem >= K; tu += K, rem -= K) {  // whole groups of the last, partial word
                need(tu >> 4);
                group((word >> (2u * (tu & 15u))) & ((1u << (2 * K)) - 1u));
            }
            to_single();
        }
        for (; rem > 0; ++tu, --rem) {  // fewer than K steps are left
            need(tu >> 4);
            single((word >> (2u * (tu & 15u))) & 3u);
        }
    } else {
        for (; rem > 0; --rem, ++t) {
            if ((t & 15u) == 0u) word = gu_rng_word(prefix, GU_RNG_STREAM_ACTION, t >> 4);
            single((word >> (2u * (t & 15u))) & 3u);
        }
    }
    account(rec);
    const uint32_t d_last = __builtin_amdgcn_ubfe(rec, GU_MROW_DONE_BIT, 1);
    const int32_t s_last = (int32_t)(((rec & GU_MROW_ADDR_MASK) - base1) >> 4);
    // resets performed = steps that started from a done env = done at entry + done flags raised by every step but the last
    if (auto_reset) ep += d_entry + fin - d_last;
    a.pos[e] = s_last;
    a.reward[e] = (int8_t)a.cell[a.cell_bytes + s_last];  // the last step's reward = reward_matrix[its next cell] (env:152-155)
    a.done[e] = (int32_t)d_last;
    a.episode[e] = ep;
    if (STATS) {
        a.ret[e] = ret;
        a.episodes_fin[e] = (int32_t)fin;
    }
    const uint64_t bits = __ballot(d_last != 0);
    if ((threadIdx.x & 63) == 0) a.done_bits[e >> 6] = bits;
}

// ------------------------------------------------------------------------------------ host side
static int multi_mode(const gu_engine *h) { return (int)gu_opt(h, GU_OPT_ROLLOUT_MULTI); }

// (K, workgroup size, copies) or false: (4^K * 4 * copies + 16) bytes per cell and workgroup, as many workgroups per CU as the
// batch needs on 256 CUs
static bool multi_shape(const gu_engine *h, int *K, int *block, int *copies)
{
    // diagnostics: force K, replicate the table (a second copy across the banks buys nothing: profiles/archive/r02h_multi_ab.txt)
    const int only = (int)gu_opt(h, GU_OPT_ROLLOUT_MULTI_K), max_copies = gu_opt(h, GU_OPT_ROLLOUT_MULTI_COPIES) == 2 ? 2 : 1;
    for (int bs = 256; bs <= GU_MAX_BLOCK; bs <<= 1) {
        const int64_t blocks = (h->N + bs - 1) / bs, per_cu = (blocks + h->n_cu - 1) / h->n_cu;
        for (int k = 4; k >= 2; k -= 2) {
            if (only && only != k) continue;
            const int64_t row = (int64_t)4 << (2 * k);
            for (int c = max_copies; c >= 1; c >>= 1) {
                if (((int64_t)h->S * (row * c + 16)) * per_cu <= h->lds_per_cu - 2048) {
                    *K = k, *block = bs, *copies = c;
                    return true;
                }
            }
        }
    }
    return false;
}

// Returns true when the launch was taken by the K-step kernel.
bool gu_rollout_multi(gu_engine *h, RolloutArgs a, int32_t policy, int auto_mode, int traj, bool stats)
{
    if ((policy != GU_POLICY_UNIFORM && policy != GU_POLICY_STREAM) || traj != 0 || auto_mode == 2 || h->n_grids != 1) return false;
    const int mode = multi_mode(h);
    if (mode == 0 || (mode != 1 && a.T < 64)) return false;  // (short launches: the second table's staging is not worth it)
    int K = 0, bs = 0, copies = 0;
    if (!multi_shape(h, &K, &bs, &copies)) return false;
    const int row_log2 = 2 * K + 2;
    const int shift = row_log2 + (copies == 2 ? 1 : 0);
    const int which = auto_mode ? 1 : 0;
    if (h->mrows_K[which] != K || h->mrows_shift[which] != shift) {
        for (uint32_t **p : {&h->d_mrows[which], &h->d_mrows1[which]}) {
            if (*p) (void)hipFree(*p);
            *p = nullptr;
        }
        h->mrows_K[which] = 0;
        if (hipMalloc(&h->d_mrows[which], ((size_t)h->S << row_log2)) != hipSuccess || hipMalloc(&h->d_mrows1[which], (size_t)h->S * 16) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        BuildMultiArgs bK{h->d_cell, h->cell_bytes, h->S, h->W, h->start0, which, shift, h->d_mrows[which]};
        const unsigned nK = (unsigned)(((int64_t)h->S << (2 * K)) + 255) / 256;
        if (K == 4) hipLaunchKernelGGL(gu_build_multi_rows_kernel<4>, dim3(nK), dim3(256), 0, h->stream, bK);
        else hipLaunchKernelGGL(gu_build_multi_rows_kernel<2>, dim3(nK), dim3(256), 0, h->stream, bK);
        BuildMultiArgs b1{h->d_cell, h->cell_bytes, h->S, h->W, h->start0, which, 4, h->d_mrows1[which]};
        hipLaunchKernelGGL(gu_build_multi_rows_kernel<1>, dim3((unsigned)(((int64_t)h->S * 4 + 255) / 256)), dim3(256), 0, h->stream, b1);
        h->mrows_K[which] = K;
        h->mrows_shift[which] = shift;
    }
    a.rows = h->d_mrows[which];
    const size_t lds = ((size_t)h->S << shift) + (size_t)h->S * 16;
    const dim3 grid(gu_blocks(h->N, bs)), block(bs);
    a.xcd_remap = a.xcd_remap && grid.x % 8 == 0;
#define GU_MULTI_LAUNCH_P(PP, KK, ST)                                                                                        \
    do {                                                                                                                     \
        auto kern = gu_rollout_multi_kernel<PP, KK, ST>;                                                                     \
        static std::atomic<uint64_t> raised{0}; /* per instantiation AND per device (gu_allow_lds) */                        \
        gu_allow_lds(kern, raised, h->device, lds, (size_t)h->lds_per_cu);                                                   \
        hipLaunchKernelGGL(kern, grid, block, lds, h->stream, a, which, h->d_mrows1[which], shift);                          \
    } while (0)
#define GU_MULTI_LAUNCH(KK, ST)                                                          \
    do {                                                                                 \
        if (policy == GU_POLICY_STREAM) GU_MULTI_LAUNCH_P(GU_POLICY_STREAM, KK, ST);     \
        else GU_MULTI_LAUNCH_P(GU_POLICY_UNIFORM, KK, ST);                               \
    } while (0)
    if (K == 4) {
        if (stats) GU_MULTI_LAUNCH(4, true); else GU_MULTI_LAUNCH(4, false);
    } else {
        if (stats) GU_MULTI_LAUNCH(2, true); else GU_MULTI_LAUNCH(2, false);
    }
#undef GU_MULTI_LAUNCH
#undef GU_MULTI_LAUNCH_P
    return true;
}
