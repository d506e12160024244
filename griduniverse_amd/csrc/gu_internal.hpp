// gu_internal.hpp -- engine object shared by the translation units of libgu.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <string>
#include <vector>

#include "../../include/gu.h"
#include "../../include/gu_diag.h"

// ---- per-cell records (two byte planes, staged in LDS by every kernel) ----------------
// Compiled on the host from the row bit-planes handed to gu_set_grid().  Together they
// encode the reference transition core/envs/griduniverse_env.py:136-155 for a cell:
//   flags[s]  bits 0..3  OPEN[a]: move a (UP,RIGHT,DOWN,LEFT; env:51-56) changes the position,
//                        i.e. NOT (grid edge (env:51-54) | wall at the candidate cell (env:149)
//                        | -- in the "absorbing" map only -- the cell itself is terminal (env:145-146))
//             bit  4     is_terminal(cell)            (env:163-168)
//             bit  7     the cell itself is a wall (env:157-161); used by the path search only
//             bits 5, 6  reward_matrix[cell] == +10 / == -10 (redundant with the reward plane; lets a kernel
//                        that keeps a PRIVATE copy of the flags plane per lane do without the second plane)
//   reward[s] int8       reward_matrix[cell]: -1, +10 or -10   (env:80-90)
// so that one env-step is   s += OPEN[a] * delta[a];  flags = F[s];  reward = R[s];  done = TERM.
// Device layout: one buffer [flags: cell_bytes | reward: cell_bytes], cell_bytes = S rounded up to 16.
#define GU_CELL_OPEN_MASK 0x0Fu
#define GU_CELL_TERM 0x10u
#define GU_CELL_TERM_BIT 4
#define GU_CELL_RPLUS 0x20u
#define GU_CELL_RMINUS 0x40u
// Actions: 0..3 = UP, RIGHT, DOWN, LEFT (env:56); -4..-1 address the same list from its end (env:148 indexes a Python list: -1 is
// LEFT), i.e. action & 3; anything else is an IndexError in the reference and rejected here.
// A buffer store of more than 64 bits reads its data registers late: a vector instruction that overwrites them in the slot behind
// the store wins for the last lanes of every row of sixteen.  The compiler pads that hazard only for stores WITHOUT a scalar offset
// (the rule of earlier chips); on gfx950 it bites with one too -- wrong first words of 12-byte rows in lanes 12 .. 15 of every
// sixteen (tools/triples_race.py, profiles/r06t_triples_race.txt: every workgroup size but one).  Two wait states behind such a store.
#define GU_WIDE_STORE_PAD()                         \
    do {                                            \
        __builtin_amdgcn_sched_barrier(0);          \
        asm volatile("s_nop 1" ::: "memory");       \
        __builtin_amdgcn_sched_barrier(0);          \
    } while (0)
#define GU_ACTION_OK(raw) ((uint32_t)((int32_t)(raw) + 4) < 8u)
#define GU_CELL_WALL 0x80u  /* the cell itself is a wall (only the path search needs it: wall nodes have no edges) */

#define GU_MAX_LDS_CELLS 32767 /* both planes of grids up to 32 767 cells (64 KiB) are LDS-resident; larger read L2 */

#define GU_HOST_ERR_WORD 4
#define GU_HOST_COUNT_WORD 8

#define GU_STREAM_PAD_WORDS 4  // spare rows behind the packed action stream: the rollout kernels read up to four words ahead

// ---- closed-loop store pacing (gu_rollout.hpp: GuPacer, gu_pace_next) -----------------
struct GuPaceEntry {   // one launch of one kind: how it runs, the loop's state behind it, and -- afterwards -- what became of it; 128 bytes
    uint32_t period_q;    // the period of launch `seq`, in 1/64 ticks (the schedule uses the rounded tick count): written by the
    uint32_t seq;         // first wave of launch seq - 1 (gu_rollout.hpp: GuPacer::decide); an entry with another number is stale
    uint64_t t_start;     // 100 MHz clock when the launch's first workgroup began (written by that launch)
    uint32_t unpaced;     // 1: launch `seq` runs WITHOUT the limiter (a probe, or because the limiter does not pay for this kind)
    uint32_t phase;       // GU_PACE_NORMAL .. : where the loop's "does the limiter pay at all" cycle stands (GuPacer::decide)
    uint32_t left;        // launches left in this phase
    uint32_t ema_paced;   // ticks: launches WITH the limiter, from one launch's start to the next one's (running mean)
    uint32_t ema_unpaced; // ... and WITHOUT it (the mean of the last probe's launches)
    // the log: what the waves of the launch reported, summed by the first wave of launch seq + 1
    uint32_t verdict;     // 0 not summed yet, 1 on schedule, 2 behind (the log's bar: GuPaceArgs::bar_num)
    uint32_t waves;       // waves that reported
    uint32_t elapsed;     // ticks from start to report, the slowest wave
    uint32_t ended_late;  // waves that were more than two periods behind their schedule when they reported
    uint32_t max_behind;  // ticks: the most any of them was behind (0 when none was more than two periods behind)
    uint32_t report_steps; // the step count behind which this launch's waves reported (launches of one kind may differ in length)
    uint32_t groups;      // 16-step groups of this launch
    // the slow loop around the rule: which SHARE of waves behind is the best one to aim for on this buffer (GuPacer::decide)
    uint32_t dec_q;       // what the period comes down by per launch, 1/64 ticks: the rule aims for a share of dec_q / gain_q
    uint32_t block_left;  // launches left in this block (the first GU_PACE_BLOCK_SKIP of a block are not counted: the period is on its way)
    uint32_t block_sum;   // ticks: start-to-start intervals of the block's counted launches, summed
    uint32_t block_n;     // ... and how many
    uint32_t last_mean;   // ticks: the mean of the block before (0: none yet)
    uint32_t up;          // 1: the last change of dec_q was upwards
    uint32_t quiet;       // 1: the limiter is off and stays off for now -- the waves of launch `seq` report nothing, its first wave only
                          // counts the launch and keeps the start-to-start mean (what a kind that is better off unpaced pays: ~nothing)
    uint32_t reserved2[9];
};
static_assert(sizeof(GuPaceEntry) == 128, "GuPaceEntry is two cache lines' halves: 128 bytes");
#define GU_PACE_BLOCK 192u      /* launches per block of the slow loop ...                       */
#define GU_PACE_BLOCK_SKIP 64u  /* ... of which the first ones only let the period settle        */
#define GU_PACE_NORMAL 0   /* the limiter is on, the period follows the rule                                                      */
#define GU_PACE_PROBE_OFF 1 /* a few launches without the limiter: how long do they take?                                          */
#define GU_PACE_OFF 2      /* the launches without it were quicker: the kind runs without a limiter                               */
#define GU_PACE_PROBE_ON 3 /* ... until it is tried again for a few launches                                                      */
// What a wave leaves behind for the loop: ONE 8-byte plain store into its own slot of its launch's set (two sets per kind, by launch
// parity), a few groups before the end of the launch.  bit 63: reported; bits 0 .. 30: ticks from the wave's start to this moment.
// (NOT atomics.  Round 5's first version had every wave add its counts to one word of its launch's record: the 3072 agent-scope
// atomics of a launch were executed one after the other at ~11 ns each and added 34 us to every 105 us launch.  Spread over 32
// neighbouring cache lines they still cost 6 .. 9 us per launch -- the lines share a memory channel, and that is where device-scope
// atomics are executed; profiles/archive/r05a_pace_c3.txt, r05c_pace_c3.txt "held ... WITHOUT records".)
#define GU_PACE_RING 64u  /* entries per launch kind: the last 61 launches can be read back (gu_rollout_pace_log) */
struct GuPaceArgs {
    GuPaceEntry *ring;      // nullptr: `period` as it is (0 = no limiter), nothing recorded
    uint64_t *slots;        // [2][slot_stride]: the waves' reports of this launch (set seq & 1) and of the launch before it
    uint32_t seq;           // this launch's number within its kind (the host counts)
    uint32_t period;        // ring == nullptr or `fixed`: the period in ticks; else the period of a kind's first launch (the model)
    uint32_t lo, hi;        // the loop keeps the period within [lo, hi] ticks
    uint32_t groups;        // 16-step groups of this launch (T / 16): the length of its schedule in periods
    uint32_t report_at;     // the waves report once they have done this many steps (a few groups before the end: gu_rollout.hpp, GuPacer::report)
    uint32_t n_waves, slot_stride;  // waves of this launch; slots per set
    uint16_t bar_num;       // of 256: the log calls a launch BEHIND when a wave was more than bar_num / 256 of the schedule behind it
    uint16_t fixed;         // 1: run with `period` and only record (measurement aid: tools/pace_loop.py)
    uint32_t gain_q;        // 1/64 ticks: what the period goes up by after a launch in which EVERY wave fell behind (a share of the waves: that share of it)
    uint32_t dec_q;         // 1/64 ticks: what it comes down by, every launch
    uint32_t probe_every;   // launches between two looks at the other side (limiter off while it is on, on while it is off); 0 = never
    uint32_t adapt;         // 1: the slow loop moves the rule's aim (dec_q) by what the launches' start-to-start time says
};

struct gu_engine {
    int device = -1;
    int n_cu = 256;                  // compute units of the device (hipDeviceAttributeMultiprocessorCount)
    int64_t lds_per_cu = 160 * 1024; // LDS bytes of one CU (the most one workgroup can ask for)
    bool gfx950 = true;              // the device the absolute figures in this library were measured on
    int64_t opt[GU_OPT_COUNT];       // gu_set_option values of this engine (GU_OPT_UNSET: the process default applies)
    int64_t opt_x[GU_OPT_X_COUNT];   // experiment switches (only a -DGU_EXPERIMENTS build ever sets them)
    hipStream_t stream = nullptr;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr, ev_sync = nullptr;  // ev_sync: gu_sync's marker (no timing)
    std::vector<hipEvent_t> ev_marks;  // gu_timer_mark pool (grows on demand, reused)
    size_t n_marks = 0;

    int64_t N = 0;        // envs on this device
    int64_t env_id0 = 0;  // global id of env 0

    // grid
    bool has_grid = false;
    int32_t W = 0, H = 0, S = 0;
    int32_t cell_bytes = 0;          // S rounded up to 16
    uint8_t *d_cell = nullptr;       // absorbing-aware planes    [2 * cell_bytes]
    uint8_t *d_cell_raw = nullptr;   // care_about_terminal=False [2 * cell_bytes]
    uint8_t *d_kind = nullptr;       // [n_grids][cell_bytes] texture class of every cell as the reference's viewer picks it
                                     // (rendering.py:119-133: goal, else lava, else wall, else ground = 3, 2, 1, 0) -- the flags
                                     // cannot tell a goal+lava cell from a lava cell; nullptr for device-generated mazes (no
                                     // overlaps there: the flags decide)
    uint64_t delta_lut = 0;          // int16 x 4: {-W, +1, +W, -1} (valid when W <= 32767)
    int32_t *d_starts = nullptr;     // [n_grids][max_starts]
    int32_t *d_nstarts = nullptr;    // [n_grids]
    int32_t n_starts = 0;            // of grid 0 (the only grid unless n_grids > 1)
    int32_t start0 = 0;              // first start cell of grid 0 (host copy)
    int32_t max_starts = 0;
    bool all_single_start = true;
    // several distinct grids of one shape: env e uses grid e / group (contiguous equal groups)
    int32_t n_grids = 1;
    int64_t group = 0;               // envs per grid = N / n_grids

    // SoA env state
    int32_t *d_out3 = nullptr;  // pos[N] | reward[N] | done[N]
    uint32_t *d_episode = nullptr;
    int32_t *d_out3_alt = nullptr;      // a second set of the env-state arrays: config 5's per-XCD launch writes its results there and
    uint32_t *d_episode_alt = nullptr;  // the sets change places when it did not give up (gu_vi_xcd.hip: gu_vi_xcd_fused_run)
    uint64_t *d_done_bits_alt = nullptr;
    uint32_t *d_nib = nullptr;     // [ceil(N / 64)][nib_dwords][64]: every env's grid at four bits per cell (gu_nibble_planes; rollout MAP 5)
    bool nib_valid = false;
    uint32_t *d_tcount = nullptr;  // per-env step-count OFFSET (read as int32); effective 64-bit count = steps_taken + offset
    uint64_t steps_taken = 0;      // lock-step counter since gu_seed (all envs step together)
    int64_t off_lo = 0, off_hi = 0;  // what the host knows of the offsets: off_lo <= every offset <= off_hi (gu_set_state, rejected actions)
    bool off_exact = true;           // ... and whether both bounds are attained (false behind a rejected action: read back when it matters)
    uint64_t seed = 0;
    uint32_t seed_prefix = 0;

    // device-resident action stream
    int32_t *d_actions = nullptr;
    uint32_t *d_actions_packed = nullptr;  // the same stream, 16 two-bit actions per word: [ceil(T / 16)][N], what the rollout kernels read
    int64_t actions_T = 0;    // rows of the stream uploaded last (0: none / rejected)
    int64_t actions_cap = 0;  // rows the two buffers can hold

    // trajectory buffers obs|reward|done, each [traj_T][N]
    int32_t *d_traj = nullptr;
    int64_t traj_T = 0;
    int traj_kind = 0;  // what the last rollout left in the buffer: 0 nothing, 1 int32 rows (three planes), 2 packed rows, 3 int32 triples
    int traj_written = 0;  // ... as gu_launch_rollout chose it for the launch it issued last
    int32_t traj_candidates = 0;                // allocations tried for the buffer (gu_alloc_trajectory)
    float traj_probe_ms_best = 0.0f, traj_probe_ms_worst = 0.0f;
    std::vector<float> traj_probe_ms;           // per candidate, in the order tried
    std::vector<uint64_t> traj_probe_addr;
    int32_t traj_kept = -1;                     // index of the kept candidate
    float traj_search_ms = 0.0f;                // wall time of the search
    uint64_t traj_peak_bytes = 0;               // most device memory the search held at once
    bool traj_registered = false;               // counted in the per-device registry of chosen buffers

    // store pacing of the launches that write rows (gu_rollout.hpp: GuPacer): one ring of launch records per launch kind
    // [policy * 3 + auto mode] for the general kernel, + 12 for the transition-row kernel's int32 rows, + 24 for its packed rows
    struct PaceKind {
        bool active = false;           // the kind's ring is in use
        const void *buffer = nullptr;  // the launch shape the ring belongs to: trajectory buffer, workgroups, launch length (within
        unsigned blocks = 0;           // a factor of two), bytes per env-step -- another shape starts the ring over
        int64_t T = 0;
        int row_bytes = 0;
        uint32_t seq = 0;              // launches of the kind recorded in the ring so far
        uint32_t model = 0;            // the period its first launch started from
        uint32_t n_waves = 0;          // waves of the kind's last launch (its reports: one word per wave)
    } pace[36];
    GuPaceEntry *d_pace_ring = nullptr;  // [36][GU_PACE_RING]
    uint64_t *d_pace_slots = nullptr;    // [36][2][pace_slot_stride] the waves' reports (allocated with the ring)
    int64_t pace_slot_stride = 0;
    hipEvent_t ev_cal[2] = {nullptr, nullptr};

    // transition-row tables of the latency-bound rollout (gu_rollout_rows.hip): [0] absorbing, [1] auto-reset folded in
    uint32_t *d_rows[2] = {nullptr, nullptr};
    int rows_shift[2] = {-1, -1};   // log2(16 * copies) the table was built for (-1: not built)
    uint32_t *d_rows2[2] = {nullptr, nullptr};  // pair tables (two steps per round trip) + the one-step table that goes with them
    bool rows2_built[2] = {false, false};
    // K-step tables of the statistics-only uniform rollout (gu_rollout_multi.hip) and the one-step tables that go with them
    uint32_t *d_mrows[2] = {nullptr, nullptr}, *d_mrows1[2] = {nullptr, nullptr};
    int mrows_K[2] = {0, 0}, mrows_shift[2] = {-1, -1};  // what they were built for (K = 0: not built)
    uint32_t *d_prow = nullptr;     // rows of the table policies (greedy: [S], sampled: [S][8]), rebuilt by every launch

    // rollout stats
    int32_t *d_ret = nullptr;
    int32_t *d_episodes_fin = nullptr;
    bool stats_valid = false;

    // done compaction: one ballot word per wave, written by every kernel that writes done[]
    uint64_t *d_done_bits = nullptr;
    bool entry_table_ok = false;    // the env state was left by a rollout (done flag == TERM bit of the cell, reward == the cell's): gu_rollout_rows.hip
    bool done_bits_valid = false;   // false only after the host installed done[] (gu_set_state)

    // scratch for masks / start choices / look_step_ahead
    void *d_scratch = nullptr;
    size_t scratch_bytes = 0;

    // pinned host staging (4*N int32)
    int32_t *h_pin = nullptr;
    double *h_tables = nullptr;     // page-locked copy of the DP tables [S] + [S][4]: the per-XCD launches of the tables alone write their results here too
    size_t h_tables_bytes = 0;      //   (gu_vi_xcd.hip), and a gu_vi_get behind such a call is two memcpys -- no launch, no wait
    bool h_tables_valid = false;    //   ... while nothing else has touched the tables since (every gu_vi_* call that may write them withdraws it)
    unsigned long long *h_ctl = nullptr;  // page-locked landing area of small results (gu_read_back): GU_CTL_WORDS 64-bit words
    char *h_up = nullptr;                 // page-locked staging of small uploads (GU_UP_BYTES): a copy from it is one DMA the stream orders,
                                          // a copy from the caller's pageable array is staged and waited for by the runtime
    uint32_t *h_seq = nullptr;      // page-locked control words (64 bytes): [0] completion word of gu_step's host-visible
                                    // paths, [GU_HOST_ERR_WORD] raised by a kernel that met an invalid action / state,
                                    // [GU_HOST_COUNT_WORD] number of done envs written by the compaction kernel
    uint32_t *d_blocks_done = nullptr;  // its device-side block counter
    uint32_t seq = 0, seq_since_sync = 0;

    // hipGraph cache for gu_step_graph
    hipGraphExec_t graph_exec = nullptr;
    int64_t graph_t0 = -1, graph_T = -1;
    uint32_t graph_flags = 0;

    // tabular DP
    double *d_v[2] = {nullptr, nullptr};
    double *d_pi[2] = {nullptr, nullptr};
    uint4 *d_pi_thr = nullptr;      // [S] inverse-CDF thresholds of d_pi[vi_cur], rebuilt by every GU_POLICY_SAMPLE rollout
    int vi_cur = 0;
    bool has_vi = false;
    double *d_delta = nullptr;      // per-block maxima + final
    uint8_t *d_greedy = nullptr;    // first-argmax action per state [cell_bytes]
    bool greedy_valid = false;
    int32_t vi_xcd_members[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // workgroups that registered per XCC in the last per-XCD launch of gu_vi_sweep_step_run
    int vi_run_form = 0;            // which form the last gu_vi_sweep_step_run took: 1 per XCD, 2 chip-wide cluster, 3 one launch per round
    void *d_vi_xcd_ctl = nullptr, *d_vi_xcd_work = nullptr;  // the per-XCD launches' own headers / keys / slots and granule buffers (gu_vi_xcd.hip)
    size_t vi_xcd_ctl_bytes = 0, vi_xcd_work_bytes = 0;
    uint32_t vi_xcd_epoch = 1;      // number of the next per-XCD launch (its tags carry it)
    int64_t vi_xcd_torn = 0;        // -DGU_VI_XCD_TORN builds: exchange words found with the right tag and the wrong payload, summed over the per-XCD launches
    int vi_dp_form = 0;             // ... and the last gu_vi_sweep / gu_vi_run / gu_vi_eval_run: 1 per XCD, 2 one workgroup, 3 chip-wide cluster, 4 one launch per round

    // agent trail (gu_trail.hip): off unless gu_trail_enable was called
    int32_t trail_cap = 0;             // entries per env (0: off)
    int32_t *d_trail = nullptr;        // [N][trail_cap] ring of cells
    int32_t *d_trail_len = nullptr, *d_trail_head = nullptr;  // [N] each (one allocation)
    uint8_t *d_trail_done = nullptr;   // [N] done flag behind the last append
    uint32_t *d_trail_alpha = nullptr; // [trail_cap] alpha of the i-th newest entry, 16 fractional bits

    // RCCL
    void *comm = nullptr;  // ncclComm_t
    int32_t nranks = 0, rank = 0;
    int32_t *d_gather = nullptr;

    int32_t *pos() const { return d_out3; }
    int32_t *reward() const { return d_out3 + N; }
    int32_t *done() const { return d_out3 + 2 * N; }
};

// what the kernels need to find a lane's grid (by value in the kernel arguments)
struct GridSel {
    int64_t group;           // envs per grid
    int64_t grid_stride;     // bytes between consecutive grids' plane pairs (2 * cell_bytes)
    const int32_t *n_starts; // [n_grids]
    int32_t n_grids, max_starts;
    int32_t per_wave;        // LDS variants: every WAVE of a workgroup stages the planes of its own grid (groups of 64 .. 192 envs under
                             // workgroups of 256: the launch shape of the shared grid); 0: one grid per workgroup
};

inline GridSel gu_grid_sel(const gu_engine *h)
{
    return GridSel{h->group, 2 * (int64_t)h->cell_bytes, h->d_nstarts, h->n_grids, h->max_starts, 0};
}

// ---- error plumbing --------------------------------------------------------------
void gu_set_error(const char *fmt, ...);
int gu_fail(int code, const char *fmt, ...);

#define GU_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess)                                                                     \
            return gu_fail(_e == hipErrorOutOfMemory ? GU_ERR_NOMEM : GU_ERR_HIP, "%s failed: %s", \
                           #expr, hipGetErrorString(_e));                                         \
    } while (0)

#define GU_REQUIRE(cond, code, ...) \
    do {                            \
        if (!(cond)) return gu_fail(code, __VA_ARGS__); \
    } while (0)

int gu_use_device(gu_engine *h);
int gu_ensure_scratch(gu_engine *h, size_t bytes);

// ---- options (gu_options.hip) ----------------------------------------------------
// The value in force for `option`: the engine's own, else the process default, else the built-in one.  A few loads; called
// per launch.  (A -DGU_EXPERIMENTS build also consults the environment variable of the same name on every call, for the
// A/B tools.)
int64_t gu_opt(const gu_engine *h, int option);
int gu_debug();  // GU_DEBUG's level when the library was first asked (read once per process): 1 = one line per search / calibration, 2 = every candidate

// Raise the dynamic-LDS limit of one kernel instantiation once PER DEVICE (HIP keeps the attribute per device: a process
// that drives several GPUs must set it on each).  `mask` is a per-instantiation static.
template <typename K>
static inline void gu_allow_lds(K kern, std::atomic<uint64_t> &mask, int device, size_t bytes, size_t wanted)
{
    if (bytes <= 64 * 1024) return;
    const uint64_t bit = device < 64 ? (1ull << device) : 0;
    if (bit && (mask.load(std::memory_order_relaxed) & bit)) return;
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)wanted);
    if (bit) mask.fetch_or(bit, std::memory_order_relaxed);
}

// ---- kernel launchers (gu_kernels.hip) -------------------------------------------
int gu_launch_reset(gu_engine *h, const uint8_t *d_mask, const int32_t *d_choice, bool only_done);
int gu_launch_step(gu_engine *h, const int32_t *d_actions_row, uint32_t flags, int32_t *host_obs = nullptr,
                   int32_t *host_reward = nullptr, int32_t *host_done = nullptr, uint32_t *host_seq = nullptr, uint32_t seq = 0,
                   uint32_t *host_err = nullptr);
int gu_launch_validate_actions(gu_engine *h, const int32_t *d_actions, int64_t count);
int gu_launch_pack_actions(gu_engine *h, int64_t T);  // validates h->d_actions[0 .. T) and fills h->d_actions_packed
int gu_launch_rollout(gu_engine *h, int64_t T, int32_t policy, uint32_t flags);
int gu_launch_lookahead(gu_engine *h, int64_t n, const int32_t *d_states, const int32_t *d_actions, bool care,
                        int32_t *d_next, int32_t *d_reward, int32_t *d_done);
int gu_launch_done_compact(gu_engine *h);
int gu_launch_deinterleave(gu_engine *h, const int32_t *triples, int32_t *planes, int64_t count);  // [count][3] -> [3][count]
int gu_probe_trajectory_buffer(gu_engine *h, int32_t *buf, int64_t T, float *ms);  // one timed full write, rollout store shape
// Device-to-device copy by a kernel on the engine's stream (bytes a multiple of 4): the snapshots that the pace calibration and the
// DP cluster launches take of the engine's state, and put back, stay on the kernel path (ordered with the launches around them,
// no copy-engine round trip for a few hundred KB).
int gu_device_copy(gu_engine *h, void *dst, const void *src, size_t bytes);
#define GU_MAX_SEGMENTS 8
struct GuSegments {  // gu_device_segments: copies (src != nullptr) and fills with zero (src == nullptr) of whole 32-bit words, one launch
    void *dst[GU_MAX_SEGMENTS];
    const void *src[GU_MAX_SEGMENTS];
    size_t words[GU_MAX_SEGMENTS];
    int n = 0;
    void add(void *d, const void *s, size_t bytes)
    {
        dst[n] = d, src[n] = s, words[n] = bytes / 4;
        ++n;
    }
};
int gu_device_segments(gu_engine *h, const GuSegments &s);
// Small results back to the host through the engine's page-locked landing area (a device-to-host copy into pageable memory is
// staged and waited for by the runtime; into pinned memory it is one DMA): copies `bytes` from `src` on the device to `dst`, behind
// everything queued on the engine's stream, and waits.  Larger than the area: the plain copy.
#define GU_CTL_WORDS (8 + 4096 + 20480)  /* ... and the tables of grids of up to 4096 states (v + pi: 40 bytes per state) in one go */
#define GU_UP_BYTES (40 * 4096)         /* page-locked staging of small uploads (gu_vi_set) */
int gu_read_back(gu_engine *h, void *dst, const void *src, size_t bytes);

// ---- tabular DP launchers (gu_vi.hip) --------------------------------------------
int gu_vi_alloc(gu_engine *h);
void gu_vi_free(gu_engine *h);
int gu_launch_greedy_table(gu_engine *h);

// ---- config 5 synchronised per XCD (gu_vi_xcd.hip) --------------------------------
struct GuXcdPlan {
    int block = 0, K = 0, NB = 1;  // threads per workgroup, states per thread at most, 16-byte exchange loads per thread (1 or 4)
    unsigned G = 0;            // workgroups
    uint32_t values = 0;       // doubles of a workgroup's value window in LDS
    size_t lds = 0, slots_bytes = 0, work_bytes = 0;  // dynamic LDS; scratch: delta-key slots (all XCCs), granule buffers (per XCC)
};
struct ViStepXcdArgs;
#define GU_VI_FALLBACK 1  /* internal: a one-launch DP form did not apply or gave up, the tables are as they were -- take the next form */
bool gu_vi_xcd_plan(const gu_engine *h, bool agents, GuXcdPlan *plan);
int gu_vi_xcd_launch(gu_engine *h, const GuXcdPlan &plan, const ViStepXcdArgs &a, bool agents, bool greedy);
int gu_vi_xcd_buffers(gu_engine *h, const GuXcdPlan &xp);  // the per-XCD launches' own buffers, wiped when they must be
uint32_t gu_vi_xcd_tag0(const gu_engine *h);                // the launch's number, where its tags carry it
void gu_vi_xcd_free(gu_engine *h);
int gu_vi_xcd_fused_run(gu_engine *h, const GuXcdPlan &xp, double gamma, int32_t iters, uint32_t flags, double *deltas);
int gu_vi_xcd_dp_run(gu_engine *h, double gamma, double threshold, bool use_threshold, bool greedy, int32_t max_rounds, int32_t *rounds_done,
                     double *deltas);

// ---- agent trail (gu_trail.hip): no-ops while the trail is off ----------------------
int gu_trail_after_step(gu_engine *h, uint32_t flags);
int gu_trail_before_reset(gu_engine *h, const uint8_t *d_mask, bool only_done);
int gu_trail_after_rollout(gu_engine *h, int64_t T, int traj, bool auto_reset);
int gu_trail_after_set_state(gu_engine *h, bool moved, bool done_given);
void gu_trail_free(gu_engine *h);

// ---- grids (gu_api.hip / gu_maze.hip) ----------------------------------------------
int gu_install_grids(gu_engine *h, int32_t n_grids, int32_t W, int32_t H, const std::vector<uint8_t> &cell,
                     const std::vector<uint8_t> &raw, const std::vector<uint8_t> &kind, const std::vector<int32_t> &starts,
                     const std::vector<int32_t> &n_starts, int32_t max_starts);

// ---- RCCL (gu_comm.hip) ----------------------------------------------------------
void gu_comm_free(gu_engine *h);
