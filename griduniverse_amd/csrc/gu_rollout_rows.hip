// gu_rollout_rows.hip -- the fused rollout on a TRANSITION-ROW table: the latency-bound form of the caller loop of
// core/algorithms/monte_carlo.py:7-26 for one small grid with a single start cell (or no auto-reset).
//
// The general kernel (gu_rollout.hpp) walks the per-cell records: per step  record -> TERM test + two selects (lazy reset)
// -> OPEN bit -> multiply-add -> LDS gather, i.e. ~6 dependent vector ops around one LDS round trip.  Here the whole
// transition of core/envs/griduniverse_env.py:136-155 INCLUDING the harness's `if done: env.reset()` is tabulated per
// (cell, action) once per grid by gu_build_rows_kernel:
//
//     row[s][a] = { LDS byte address of row[next] : bits 0..19 | done(next) : bit 23 | reward_matrix[next] int8 : bits 24..31 }
//     next      = look_step_ahead(base, a).next,   base = start cell if (auto-reset and s is terminal) else s
//
// so that one env-step is   addr = (rec & 0xFFFFF) | (copy * 16 + action * 4);   rec = LDS[addr]   -- ONE v_and_or_b32 and one
// ds_read_b32 on the dependent chain; cell index, reward, done flag, episode count and the trajectory stores all hang off
// `rec` beside the chain.  The 16-byte rows are replicated `copies` (<= 8) times so that the lanes of a half-wave spread
// over the LDS banks (bank = copy * 4 + action).  The invariant the table relies on -- "the env is done exactly when its
// cell is terminal" -- holds after any step but not for an arbitrary stored state (gu_set_state, a reset onto a terminal
// start), so every lane takes its FIRST step of a launch on the per-cell planes, like the general kernel.
//
// Results are bit-identical to the general kernel (tests/test_gpu_round2.py runs both on the same seeds); the launcher
// picks this one where it is faster (profiles/r02b_map_ab.txt).
#include "gu_rollout.hpp"

#define GU_ROW_ADDR_MASK 0xFFFFFu
#define GU_ROW_DONE_BIT 23

struct BuildRowsArgs {
    const uint8_t *cell;  // absorbing-aware planes [flags | reward]
    int32_t cell_bytes, S, W, start0, auto_reset, row_shift;
    uint32_t *rows;       // [S][4]
};

__global__ void __launch_bounds__(256) gu_build_rows_kernel(const BuildRowsArgs a)
{
    const int32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= a.S) return;
    const int32_t base = (a.auto_reset && (a.cell[s] & GU_CELL_TERM)) ? a.start0 : s;  // lazy `if done: env.reset()` (env:187-193)
    const uint32_t fb = a.cell[base];  // OPEN bits of the absorbing map: a terminal cell does not move (env:145-146)
    uint32_t out[4];
#pragma unroll
    for (uint32_t act = 0; act < 4; ++act) {
        const int32_t n = base + (((fb >> act) & 1u) ? gu_delta<false>(act, 0, a.W) : 0);
        const uint32_t d = (a.cell[n] >> GU_CELL_TERM_BIT) & 1u;
        const uint32_t r = (uint8_t)a.cell[a.cell_bytes + n];
        out[act] = ((uint32_t)n << a.row_shift) | (d << GU_ROW_DONE_BIT) | (r << 24);
    }
    *reinterpret_cast<uint4 *>(a.rows + 4 * (int64_t)s) = make_uint4(out[0], out[1], out[2], out[3]);
}

typedef __attribute__((address_space(3))) const uint32_t *lds_u32_ptr;

template <int POLICY, int TRAJ, bool STATS>
__global__ void __launch_bounds__(GU_MAX_BLOCK) gu_rollout_rows_kernel(const RolloutArgs a, const int32_t auto_reset)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int32_t shift = a.row_shift;         // log2(16 * copies)
    const int32_t copies_log2 = shift - 4;
    // LDS address of the staged table: folded into every record (0 in practice: this kernel has no static LDS), so that a
    // record's address bits are the raw ds_read address and no base is added on the dependent chain
    const uint32_t lds_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t *)smem;
    {
        const uint4 *g = reinterpret_cast<const uint4 *>(a.rows);
        const int32_t n_rows = a.S << copies_log2;
        for (int32_t i = threadIdx.x; i < n_rows; i += blockDim.x) {
            uint4 row = g[i >> copies_log2];
            row.x += lds_base, row.y += lds_base, row.z += lds_base, row.w += lds_base;
            reinterpret_cast<uint4 *>(smem)[i] = row;
        }
        __syncthreads();
    }
    const int64_t e64 = (int64_t)gu_env_block(a.xcd_remap) * blockDim.x + threadIdx.x;
    if (e64 >= a.N) return;
    const uint32_t e = (uint32_t)e64;
    const uint32_t lane_copy = (threadIdx.x & ((1u << copies_log2) - 1u)) << 4;  // this lane's copy of every row

    int32_t s = a.pos[e];
    uint32_t d = (uint32_t)a.done[e];
    uint32_t ep = a.episode[e];
    const uint32_t t_lane = a.tcount[e] + a.steps_taken;
    const uint32_t prefix = gu_rng_prefix(a.seed_prefix, a.env_id0 + e);
    int32_t ret = 0;
    uint32_t fin = 0;
    uint32_t rec;

    char *po = (char *)a.tr_obs, *pr = (char *)a.tr_reward, *pd = (char *)a.tr_done;
    const char *pa = (const char *)a.actions;
    const uint32_t e4 = e * 4u;
    const int64_t row = a.N * 4;
    const uint32_t row32 = (uint32_t)row;
    __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(po, 0, 0xFFFFFFFFu, 0x00020000);
    __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(pr, 0, 0xFFFFFFFFu, 0x00020000);
    __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(pd, 0, 0xFFFFFFFFu, 0x00020000);
    auto rebase = [&](int64_t rows) {
        po += rows * row;
        pr += rows * row;
        pd += rows * row;
        ro = __builtin_amdgcn_make_buffer_rsrc(po, 0, 0xFFFFFFFFu, 0x00020000);
        rr = __builtin_amdgcn_make_buffer_rsrc(pr, 0, 0xFFFFFFFFu, 0x00020000);
        rd = __builtin_amdgcn_make_buffer_rsrc(pd, 0, 0xFFFFFFFFu, 0x00020000);
    };
    // everything a step leaves behind hangs off the record that the step fetched -- none of it is on the chain.  It is
    // emitted one iteration late, i.e. AFTER the next step's LDS read has been issued, so that it fills that read's latency
    // instead of delaying its issue: iteration i fetches record i and emits record i - 1 into trajectory row i - 1.
    auto emit = [&](uint32_t rec, uint32_t soff) {
        const int32_t r = (int32_t)rec >> 24;
        const uint32_t dn = __builtin_amdgcn_ubfe(rec, GU_ROW_DONE_BIT, 1);
        fin += dn;
        if (STATS) ret += r;
        if (TRAJ) {
            const int32_t cell = (int32_t)__builtin_amdgcn_ubfe(rec - lds_base, (uint32_t)shift, 20u - (uint32_t)shift);
            if (TRAJ == 1) {
                __builtin_amdgcn_raw_buffer_store_b32(cell, ro, e4, soff, 0);
                __builtin_amdgcn_raw_buffer_store_b32(r, rr, e4, soff, 0);
                __builtin_amdgcn_raw_buffer_store_b32((int32_t)dn, rd, e4, soff, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b32((int32_t)((uint32_t)cell | (((uint32_t)r & 0xFFu) << 16) | (dn << 24)), ro, e4, soff, 0);
            }
        }
    };
    // first step of the launch, on the per-cell planes: the stored done flag decides the lazy reset (it may disagree
    // with the cell: fresh reset onto a terminal start, gu_set_state)
    auto first_step = [&](uint32_t act) {
        const int8_t *rew = reinterpret_cast<const int8_t *>(a.cell + a.cell_bytes);
        if (auto_reset && d) {
            s = a.starts[0];
            ++ep;
        }
        const uint32_t f0 = a.cell[s];
        s += ((f0 >> act) & 1u) ? gu_delta<false>(act, 0, a.W) : 0;
        const uint32_t dn = (a.cell[s] >> GU_CELL_TERM_BIT) & 1u;
        rec = (((uint32_t)s << shift) + lds_base) | (dn << GU_ROW_DONE_BIT) | ((uint32_t)(uint8_t)rew[s] << 24);
    };
    auto step = [&](uint32_t act, uint32_t soff) {
        uint32_t actoff = lane_copy | (act << 2);  // off the chain (the action does not depend on the env state)
        asm("" : "+v"(actoff));                    // keep it ONE value: otherwise the three-way OR is re-associated onto the chain
        const uint32_t prev = rec;
        const uint32_t addr = (prev & GU_ROW_ADDR_MASK) | actoff;  // v_and_or_b32: the only vector op between two LDS reads
        rec = *(lds_u32_ptr)(uintptr_t)addr;
        emit(prev, soff);
    };
    auto step1 = [&](uint32_t act) {
        step(act, 0);
        if (TRAJ) rebase(1);
    };

    if (POLICY == GU_POLICY_UNIFORM) {
        uint32_t t = t_lane;
        uint32_t word = gu_rng_word(prefix, GU_RNG_STREAM_ACTION, t >> 4);
        first_step((word >> (2u * (t & 15u))) & 3u);
        ++t;
        int64_t i = 1;
        const uint32_t t_first = __builtin_amdgcn_readfirstlane(t);
        if (__all(t == t_first)) {  // every lane at the same step count: the 16-actions-per-word schedule is wave-uniform
            t = t_first;
            if (t & 15u) {  // head: finish the current word
                word = gu_rng_word(prefix, GU_RNG_STREAM_ACTION, t >> 4);
                for (; i < a.T && (t & 15u); ++i, ++t) step1((word >> (2u * (t & 15u))) & 3u);
            }
            for (; i + 16 <= a.T; i += 16, t += 16) {
                word = gu_rng_word(prefix, GU_RNG_STREAM_ACTION, t >> 4);
#pragma unroll
                for (uint32_t j = 0; j < 16; ++j) step(__builtin_amdgcn_ubfe(word, 2 * j, 2), j * row32);
                if (TRAJ) rebase(16);
            }
            if (i < a.T) {
                word = gu_rng_word(prefix, GU_RNG_STREAM_ACTION, t >> 4);
                for (uint32_t j = 0; i < a.T; ++i, ++j) step1((word >> (2u * j)) & 3u);
            }
        } else {
            if ((t & 15u) == 0u) word = gu_rng_word(prefix, GU_RNG_STREAM_ACTION, t >> 4);
            for (; i < a.T; ++i) {
                step1((word >> (2u * (t & 15u))) & 3u);
                ++t;
                if ((t & 15u) == 0u) word = gu_rng_word(prefix, GU_RNG_STREAM_ACTION, t >> 4);
            }
        }
    } else {  // GU_POLICY_STREAM: action rows loaded 8 at a time, one chunk ahead of the steps that consume them
        constexpr int CH = 8;
        first_step((uint32_t)(*(const int32_t *)(pa + e4)) & 3u);
        pa += row;
        int64_t i = 1;
        uint32_t cur[CH], nxt[CH];
        auto load_chunk = [&](uint32_t (&dst)[CH], const char *base) {
            const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, 0xFFFFFFFFu, 0x00020000);
#pragma unroll
            for (int j = 0; j < CH; ++j) dst[j] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(ra, e4, j * row32, 0);
        };
        if (i + CH <= a.T) load_chunk(cur, pa);
        for (; i + CH <= a.T; i += CH) {
            pa += CH * row;
            if (i + 2 * CH <= a.T) load_chunk(nxt, pa);
#pragma unroll
            for (int j = 0; j < CH; ++j) step(cur[j] & 3u, j * row32);
            if (TRAJ) rebase(CH);
#pragma unroll
            for (int j = 0; j < CH; ++j) cur[j] = nxt[j];
        }
        for (; i < a.T; ++i) {
            const uint32_t act = (uint32_t)(*(const int32_t *)(pa + e4)) & 3u;
            pa += row;
            step1(act);
        }
    }
    emit(rec, 0);  // the last step's record (row T - 1: every loop above leaves the row base one step behind)
    // resets performed = steps that started from a done env = (done at entry: counted in first_step) + done flags seen
    // on every step but the last
    const uint32_t d_last = __builtin_amdgcn_ubfe(rec, GU_ROW_DONE_BIT, 1);
    if (auto_reset) ep += fin - d_last;
    a.pos[e] = (int32_t)__builtin_amdgcn_ubfe(rec - lds_base, (uint32_t)shift, 20u - (uint32_t)shift);
    a.reward[e] = (int32_t)rec >> 24;
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
// (block size, copies) for the row table, or false when it does not fit: 16 * copies bytes per cell and block, one table
// shared by all waves of a block; the batch must fit in (blocks per CU the LDS admits) x 256 CUs blocks.
static bool rows_shape(const gu_engine *h, int *block, int *copies)
{
    if (h->n_grids != 1) return false;
    int best_bs = 0, best_c = 0;
    for (int bs = 256; bs <= GU_MAX_BLOCK; bs <<= 1) {
        const int64_t blocks = (h->N + bs - 1) / bs, per_cu = (blocks + 255) / 256;
        for (int c = 8; c >= 1; c >>= 1) {
            if ((int64_t)h->S * 16 * c * per_cu <= 160 * 1024 - 2048) {
                if (c > best_c) best_c = c, best_bs = bs;
                break;
            }
        }
    }
    if (!best_c) return false;
    *block = best_bs;
    *copies = best_c;
    return true;
}

static int rows_mode()
{
    const char *s = std::getenv("GU_ROLLOUT_ROWS");  // read per launch: A/B runs switch it inside one process
    return s ? std::atoi(s) : -1;
}

template <int POLICY>
static void rows_dispatch(const RolloutArgs &a, int traj, bool stats, int auto_reset, dim3 grid, dim3 block, size_t lds, hipStream_t stream)
{
#define GU_ROWS_LAUNCH(TR, ST)                                                                                           \
    do {                                                                                                                 \
        auto kern = gu_rollout_rows_kernel<POLICY, TR, ST>;                                                              \
        static size_t allowed = 64 * 1024; /* per instantiation: raise the dynamic-LDS limit once, not per launch */     \
        if (lds > allowed) {                                                                                             \
            (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);        \
            allowed = 160 * 1024;                                                                                        \
        }                                                                                                                \
        hipLaunchKernelGGL(kern, grid, block, lds, stream, a, auto_reset);                                               \
    } while (0)
    if (traj == 1) {
        if (stats) GU_ROWS_LAUNCH(1, true); else GU_ROWS_LAUNCH(1, false);
    } else if (traj == 2) {
        if (stats) GU_ROWS_LAUNCH(2, true); else GU_ROWS_LAUNCH(2, false);
    } else {
        if (stats) GU_ROWS_LAUNCH(0, true); else GU_ROWS_LAUNCH(0, false);
    }
#undef GU_ROWS_LAUNCH
}

// Returns true when the launch was taken by the row-table kernel.
bool gu_rollout_rows(gu_engine *h, RolloutArgs a, int32_t policy, int auto_mode, int traj, bool stats)
{
    if (policy != GU_POLICY_UNIFORM && policy != GU_POLICY_STREAM) return false;
    if (auto_mode == 2) return false;  // several start cells: the reset draws from the RNG, it cannot be tabulated
    const int mode = rows_mode();
    // Default policy (profiles/r02b_map_ab.txt, profiles/r02d_rows_crossover.txt; interleaved A/B in one process): every
    // launch that is bound by the dependent chain rather than by the HBM write path --
    //   stats only           : every batch size (62 -> 40 us at 65 536 envs, 68 -> 35 us at 262 144)
    //   packed rows (4 B)    : up to one 256-env workgroup per CU (83 -> 51 us at 65 536 envs; 103 against 111 us at 131 072)
    //   int32 rows (12 B)    : up to 32 768 envs (80 -> 59 us at 4096..16 384 envs, 84 -> 74 us at 32 768 -- config 2, and a
    //                          config-4 shard; beyond that the general kernel's store timing is the better one: 124 against 133 us
    //                          at 65 536 envs)
    if (mode == 0) return false;
    if (mode != 1) {
        const unsigned blocks = gu_blocks(h->N, 256);
        if ((traj == 1 && blocks > 128) || (traj == 2 && blocks > 256)) return false;
    }
    int bs = 0, copies = 0;
    if (!rows_shape(h, &bs, &copies)) return false;
    int shift = 4;
    while ((1 << (shift - 4)) < copies) ++shift;
    const int which = auto_mode ? 1 : 0;
    if (!h->d_rows[which]) {
        if (hipMalloc(&h->d_rows[which], (size_t)h->S * 4 * sizeof(uint32_t)) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        h->rows_shift[which] = -1;
    }
    if (h->rows_shift[which] != shift) {
        BuildRowsArgs b{h->d_cell, h->cell_bytes, h->S, h->W, h->start0, which, shift, h->d_rows[which]};
        hipLaunchKernelGGL(gu_build_rows_kernel, dim3((unsigned)((h->S + 255) / 256)), dim3(256), 0, h->stream, b);
        h->rows_shift[which] = shift;
    }
    a.rows = h->d_rows[which];
    a.row_shift = shift;
    const size_t lds = ((size_t)h->S * 16) << (shift - 4);
    const dim3 grid(gu_blocks(h->N, bs)), block(bs);
    a.xcd_remap = a.xcd_remap && grid.x % 8 == 0;
    if (policy == GU_POLICY_UNIFORM) rows_dispatch<GU_POLICY_UNIFORM>(a, traj, stats, which, grid, block, lds, h->stream);
    else rows_dispatch<GU_POLICY_STREAM>(a, traj, stats, which, grid, block, lds, h->stream);
    return true;
}
