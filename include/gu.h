/* gu.h -- C ABI of libgu.so, the MI355X (gfx950) GridUniverse step/reset engine.
 *
 * The reference (TheMTank/GridUniverse) is pure Python and has no FFI layer; its
 * boundary for this path is the old-gym Env API of
 * core/envs/griduniverse_env.py:14-321.  Every entry point below names the
 * reference interface (file:line, `env:` = core/envs/griduniverse_env.py) whose
 * per-instance work it replaces for a whole batch of env instances.  The
 * reference-side binding (ctypes) is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C, no C++/torch types; every function returns 0 (GU_OK) or a negative
 *     GU_ERR_* code and never throws; gu_last_error() gives the message of the last
 *     failure on the calling thread.
 *   - the caller owns every host buffer (C-contiguous, int32 unless noted);
 *     the library owns all device memory and frees it in gu_destroy().
 *   - a handle is bound to ONE device and ONE HIP stream and is not thread-safe;
 *     different handles may be driven from different threads / processes.
 *   - calls that take host buffers are synchronous; calls documented "async" only
 *     enqueue work on the handle's stream (gu_sync() waits for it).
 *   - env state is struct-of-arrays in HBM: pos[N] | reward[N] | done[N] (one
 *     contiguous int32[3N] block, so the gathered view is one collective),
 *     episode[N] (uint32), a 64-bit step count per env.  Actions are 0..3 =
 *     UP,RIGHT,DOWN,LEFT (env:56); -4..-1 are the same list addressed from its end, as
 *     in the reference (env:148: -1 is LEFT); anything else is
 *     rejected (the reference raises IndexError).  Caller-supplied actions and
 *     states are validated by the kernels that consume them (an error word in
 *     page-locked memory), not by host loops.
 */
#ifndef GU_H
#define GU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GU_ABI_VERSION 1

#define GU_OK 0
#define GU_ERR_INVALID (-1)     /* bad argument (NULL, out of range, wrong size)        */
#define GU_ERR_HIP (-2)         /* a HIP runtime call failed (no device, launch error)   */
#define GU_ERR_NOMEM (-3)       /* host or device allocation failed                      */
#define GU_ERR_STATE (-4)       /* call order: no grid set, no trajectory reserved, ...  */
#define GU_ERR_COMM (-5)        /* RCCL failure                                          */
#define GU_ERR_UNSUPPORTED (-6) /* grid too large for this build, etc.                   */

/* gu_rollout / gu_step_device flags */
#define GU_F_AUTO_RESET 1u  /* harness `if done: env.reset()` applied before the next step */
#define GU_F_TRAJECTORY 2u  /* write (obs,reward,done)[t][env] for every step              */
#define GU_F_STATS 4u       /* per-env sum of rewards and finished-episode count           */
#define GU_F_PACKED 16u     /* gu_rollout: write ONE packed uint32 per env-step instead of three int32 rows:
                               obs | (reward & 0xFF) << 16 | done << 24  (4 B per env-step; grids <= 65 536 cells);
                               read back with gu_read_trajectory_packed.  Excludes GU_F_TRAJECTORY. */
#define GU_F_PINNED_IO 8u   /* gu_step: the caller's buffers are page-locked (gu_host_alloc): the kernel reads and
                               writes them itself instead of going through the library's staging block */

/* gu_rollout policy kinds */
#define GU_POLICY_UNIFORM 0 /* a ~ U{0..3} from the per-env counter RNG (stream 0)           */
#define GU_POLICY_STREAM 1  /* a = actions[t][env] uploaded with gu_upload_actions           */
#define GU_POLICY_GREEDY 2  /* a = first argmax of pi[pos] (np.argmax; examples/griduniverse_alg_examples.py:76) */
#define GU_POLICY_SAMPLE 3  /* a ~ pi[pos]: inverse CDF on one 32-bit word of RNG stream 2 per step, the batched form of
                               np.random.choice(4, p=policy[obs]) in core/algorithms/monte_carlo.py:20.  The stream is the
                               build's own (oracle/gu_rng.py is its specification): one MurmurHash3 word per sixteen steps
                               of an env, the fifteen behind it by xorshift32 + a Weyl increment -- sampled sequences
                               differ from those of libraries built before ABI round 4, their distribution does not */

typedef struct gu_engine *gu_handle;

/* ---- library ----------------------------------------------------------------- */
int gu_version(void);                         /* GU_ABI_VERSION */
int gu_last_error(char *buf, size_t len);     /* copies the thread's last message, returns its length */
int gu_device_count(int *count);              /* hipGetDeviceCount */
int gu_source_hash(char *buf, size_t len);    /* 16 hex digits identifying the sources the library was built from
                                                 (sha256 over every .hip / .hpp under csrc and include/gu.h); returns the length */
/* "key=value;..." description of device `device_id` (name, arch, pci bus id, cus, lds_per_cu, sclk_khz, mclk_khz,
 * bus_bits, l2_bytes, hbm_bytes, hbm_free): what bench.py prints next to its numbers.  Returns the length. */
int gu_device_info(int device_id, char *buf, size_t len);

/* ---- options -------------------------------------------------------------------
 * Launch-shape and search parameters, per engine (h) or -- h == NULL -- as the process default every engine without a value
 * of its own uses.  GU_OPT_UNSET as value returns an option to the built-in default.  RESULTS NEVER DEPEND ON THEM (the
 * tests run every kernel path against the oracle by flipping them); they exist for tests and measurements.  The library
 * reads only two environment variables: GU_RCCL_LIB (which librccl to dlopen) and GU_DEBUG (reports on stderr).
 * Options 100+ select known-unsafe experiments and are refused (GU_ERR_UNSUPPORTED) unless the library was built with
 * -DGU_EXPERIMENTS (make exp -> libgu_exp.so, used by tools/ only). */
#define GU_OPT_UNSET INT64_MIN
#define GU_OPT_ROLLOUT_BLOCK 1        /* workgroup size of the general rollout kernel: 64 .. 1024 (256)                         */
#define GU_OPT_ROLLOUT_ROWS 2         /* transition-row kernel: 0 never, 1 wherever eligible, 2 = 1 without its pair tables
                                         (default: by launch shape)                                                           */
#define GU_OPT_ROWS_COPIES 3          /* copies of its table across the LDS banks: 1, 2, .. 32 (default: by policy)           */
#define GU_OPT_ROLLOUT_MULTI 4        /* K-step kernel: 0 never, 1 whenever the table fits (default: launches of >= 64 steps)  */
#define GU_OPT_ROLLOUT_MULTI_K 5      /* force K = 2 or 4                                                                     */
#define GU_OPT_ROLLOUT_MULTI_COPIES 6 /* 2 = replicate its table across the banks                                             */
#define GU_OPT_ROLLOUT_XCD 7          /* 1 = XCD-aware env-block order (measured slower; off)                                 */
#define GU_OPT_VI_PATH 8              /* DP: 1 = no workgroup-cluster kernel, 2 = one launch per round on every grid size,
                                         3 = chip-wide cluster kernel with an INJECTED grid-barrier timeout (tests of the
                                         fallback), 4 = no per-XCD form of gu_vi_sweep_step_run (the chip-wide cluster kernel
                                         instead), 5 = its per-XCD form gives up at once (INJECTED; tests of the fallback),
                                         6 = the per-XCD form of gu_vi_sweep_step_run for calls of ONE round too (by default
                                         those take the single fused launch, which starts ~10 us quicker; tests, measurements) */
#define GU_OPT_MC_SCRATCH_MB 9        /* scratch budget of gu_mc_evaluate (2048)                                              */
#define GU_OPT_MC_LANE_RETURNS 10     /* 1 = return sums by the per-lane kernel instead of the LDS-tiled one                  */
#define GU_OPT_MC_GLOBAL_WALK 11      /* 1 = history walk with its counters in global memory instead of LDS                   */
#define GU_OPT_STEP_SYNC 12           /* 1 = gu_step always waits with the stream synchronisation                             */
#define GU_OPT_TRAJ_CANDIDATES 13     /* back-to-back candidates of the trajectory placement search (4; 1 = take the first)   */
#define GU_OPT_TRAJ_FAR_CANDIDATES 14 /* candidates behind spacers (0 = none, the default)                                       */
#define GU_OPT_TRAJ_STRIDE_MIB 15     /* spacer size (3072)                                                                   */
#define GU_OPT_TRAJ_FAR_MIB 16        /* most memory the search may hold at once (49152)                                      */
#define GU_OPT_ROLLOUT_PACE 18        /* store pacing of launches that write rows: -1 = closed loop (default, see gu_rollout_pacing),
                                         0 = none, n = fixed period of n 10 ns ticks per 16 steps (-2: accepted, same as -1) */
#define GU_OPT_VI_XCD_BLOCK 19        /* workgroup size of the per-XCD form of gu_vi_sweep_step_run: 256, 512, 1024 (0 = by batch size) */
#define GU_OPT_TRAJ_LAYOUT 24         /* device layout of the int32 trajectory: 0 = three planes [T][N], 1 = one plane of (obs, reward,
                                         done) triples [T][N][3] written with one 12-byte store per lane and step, -1 (default) =
                                         triples where they are faster (small batches under the uniform policy); what
                                         gu_read_trajectory, gu_mc_evaluate and the host see does not change                   */
#define GU_OPT_ROLLOUT_HALF_WAVES 28   /* transition-row kernel, launches that write rows: 32 envs per wave and twice the waves; -1 = where
                                         measured faster (triples + pair tables at 8192 .. 16 384 envs; default), 0 never, 1 always   */
#define GU_OPT_ROLLOUT_ENTRY 29        /* transition-row kernel: 1 (default) = a launch that follows another rollout of the same engine takes
                                         its FIRST step on the staged table too (the state a rollout leaves behind always agrees with its cell);
                                         0 = always on the per-cell planes, as a launch behind gu_reset / gu_set_state / gu_step must       */
#define GU_OPT_SYNC_SPIN_US 30         /* gu_sync / gu_timer_end poll the event for up to this many microseconds (5000) before the runtime's
                                         blocking wait; 0 = blocking at once (ranks or engines that outnumber the host cores)             */
#define GU_OPT_COUNT 31
int gu_set_option(gu_handle h, int32_t option, int64_t value);
int gu_get_option(gu_handle h, int32_t option, int64_t *value);   /* the value in force (own, process default or built-in) */

/* ---- lifetime ----------------------------------------------------------------
 * One engine = `num_envs` lock-stepped instances of one grid on device `device_id`;
 * `env_id0` is the global index of its first env (RNG streams are keyed by global
 * id, so a sharded batch reproduces the single-device batch byte for byte).
 * Replaces num_envs x GridUniverseEnv.__init__ (env:17-107) state setup. */
int gu_create(int device_id, int64_t num_envs, int64_t env_id0, gu_handle *out);
int gu_destroy(gu_handle h);

/* ---- grid --------------------------------------------------------------------
 * Row bit-planes, `words_per_row` = ceil(W/32) uint32 words per row, bit (x & 31) of
 * word (x >> 5) of row y describes cell s = y*W + x (env:116-117).
 *   wall_rows  : wall_grid[s] == 1          (env:120-134, env:157-161)
 *   goal_rows  : s in goal_states           (env:173-174)
 *   lava_rows  : s in lava_states           (env:170-171)
 *   rplus_rows / rminus_rows : reward_matrix[s] == +10 / == -10 (env:80-90, 312-316),
 *       or both NULL to derive them as (lava ? -10 : goal ? +10 : -1).  They are
 *       separate planes because the reference's reward matrix and terminal test
 *       can disagree (negative indices wrap only in the former; quirk 5).
 *   starts     : starting_states (env:61-63), n_starts >= 1.
 * W * H <= 2^30 cells and W <= 8 388 607 columns (GU_ERR_UNSUPPORTED beyond: the move is a 24-bit multiply-add).
 * The library compiles the planes into two bytes per cell -- flags (per action: does the
 * move change the position, incl. the absorbing-terminal rule; terminal bit; reward code;
 * wall bit) and the int8 reward -- which the kernels stage in LDS (gu_get_cells reads
 * them back). */
int gu_set_grid(gu_handle h, int32_t W, int32_t H, int32_t words_per_row,
                const uint32_t *wall_rows, const uint32_t *goal_rows, const uint32_t *lava_rows,
                const uint32_t *rplus_rows, const uint32_t *rminus_rows,
                const int32_t *starts, int32_t n_starts);

/* Several DISTINCT grids of one shape in one engine (SURVEY.md 8(d) C3 variant, 8(f) rank 3): env e uses grid
 * e / (num_envs / n_grids) -- contiguous equal groups, n_grids must divide num_envs.  Planes are
 * [n_grids][H][words_per_row]; starts is [n_grids][max_starts] with n_starts[g] valid entries per grid.
 * The rollout kernels stage one grid per block in LDS when the group size is a multiple of 64, keep a private
 * copy of its grid per lane in LDS otherwise (e.g. one grid per env), and read L2 for grids too large for either.
 * Multi-grid engines do not support the DP / Monte-Carlo tables (one value table per engine). */
int gu_set_grids(gu_handle h, int32_t n_grids, int32_t W, int32_t H, int32_t words_per_row,
                 const uint32_t *wall_rows, const uint32_t *goal_rows, const uint32_t *lava_rows,
                 const uint32_t *rplus_rows, const uint32_t *rminus_rows,
                 const int32_t *starts, const int32_t *n_starts, int32_t max_starts);
/* Generate n_grids random mazes ON THE DEVICE (one lane carves one maze): the algorithm of
 * core/envs/maze_generation.py:41-149 (recursive backtracker, one 'x', one 'G'), draws from RNG stream 3 keyed
 * by (maze_seed, global grid id).  Replaces n_grids x GridUniverseEnv(random_maze=True) (env:318-321). */
int gu_generate_mazes(gu_handle h, int32_t n_grids, int32_t W, int32_t H, uint64_t maze_seed);
/* Read back grid `grid_index` as compiled: flags[S] (OPEN bits 0-3 = move changes the position, bit 4 terminal,
 * bits 5-6 reward code, bit 7 the cell is a wall),
 * reward[S] (int8), and its start table (any pointer may be NULL). */
int gu_get_cells(gu_handle h, int32_t grid_index, uint8_t *flags, int8_t *reward, int32_t *starts, int32_t *n_starts);

/* ---- RNG ---------------------------------------------------------------------
 * Keys the per-env counter RNG (MurmurHash3 of seed, global env id, stream,
 * counter -- 32-bit counters, no stream repeats before 2^32 draws; host view:
 * griduniverse_amd/rng.py, restated for the tests in oracle/gu_rng.py) and zeroes
 * episode[] and tcount[].
 * The reference has no per-env RNG (env:242-244 stores one and never uses it). */
int gu_seed(gu_handle h, uint64_t seed);

/* ---- reset: GridUniverseEnv._reset, env:187-193 --------------------------------
 * mask (N bytes, NULL = all): which envs to reset.  start_choice (N int32 indices
 * into starts[], NULL = draw from RNG stream 1 at the env's episode counter).
 * Sets done = 0, bumps episode[].  obs_out (N, optional) receives pos[]. */
int gu_reset(gu_handle h, const uint8_t *mask, const int32_t *start_choice, int32_t *obs_out);
/* Device-side: reset exactly the envs whose done flag is set (async). */
int gu_reset_done(gu_handle h);

/* ---- step: GridUniverseEnv._step, env:176-185 (+ look_step_ahead env:136-155) ---
 * Synchronous, host buffers: actions in, (obs, reward, done) out (each N int32,
 * outputs optional).  flags: GU_F_AUTO_RESET; GU_F_PINNED_IO when every buffer passed
 * is page-locked (gu_host_alloc): the kernel then reads / writes them directly over
 * PCIe and no copy command is issued (pageable buffers take the same route through the
 * library's own page-locked staging block, plus one memcpy each way).  Batches of up to
 * 8192 envs return as soon as the kernel has published a completion word in page-locked
 * memory -- the results are in place, the stream may still be draining; every later call on
 * the handle is ordered behind it as usual.
 * An action outside -4..3 is detected BY THE KERNEL (env:148 raises IndexError before it touches the
 * instance): that env does not step -- position, reward, done flag, pending lazy reset and step
 * count stay as they were, its outputs repeat its current state -- every env with a valid action
 * steps, and the call returns GU_ERR_INVALID naming the first offender.
 * With GU_F_PINNED_IO every pointer is checked (once per allocation) to be page-locked host memory;
 * pageable memory is GU_ERR_INVALID, not a GPU fault. */
int gu_step(gu_handle h, const int32_t *actions, uint32_t flags,
            int32_t *obs, int32_t *reward, int32_t *done);

/* Device-resident action stream [T][N] for gu_step_device / GU_POLICY_STREAM.  Values are validated on the
 * device after the copy; a stream holding anything outside -4..3 is rejected as a whole (GU_ERR_INVALID).
 * The upload REPLACES the stream: afterwards it holds exactly rows 0 .. T-1 (a rejected upload leaves none).
 * Besides the int32 rows (read by the single-step launches) the device keeps the stream packed to two bits
 * per action, 16 steps per word and env; gu_rollout(GU_POLICY_STREAM) reads that: 0.25 B, not 4 B, per env-step. */
int gu_upload_actions(gu_handle h, const int32_t *actions, int64_t T);
/* One step with actions row `t` of the uploaded stream; results stay in HBM (async). */
int gu_step_device(gu_handle h, int64_t t, uint32_t flags);
/* The same T single-step launches (rows t0 .. t0+T-1) replayed from one hipGraph (async). */
int gu_step_graph(gu_handle h, int64_t t0, int64_t T, uint32_t flags);
/* Copy the current (obs, reward, done) block to the host (any pointer may be NULL). */
int gu_read_outputs(gu_handle h, int32_t *obs, int32_t *reward, int32_t *done);

/* ---- rollout: the caller loop of core/algorithms/monte_carlo.py:7-26 fused -----
 * T env-steps per env in ONE launch (async).  GU_F_TRAJECTORY: the (obs, reward, done) of step i of this call land in row i of the
 * trajectory buffer (gu_reserve_trajectory(T) first; a buffer that is already large enough is kept).  GU_F_STATS: the per-env
 * reward sum and the number of episodes finished during this call are kept for gu_read_stats. */
/* Placement: where a buffer lands in HBM changes its write rate by a few per cent, so buffers of 64 MB and more are CHOSEN: up to
 * GU_OPT_TRAJ_CANDIDATES (4) allocations are written once in the rollout's store shape and the fastest is kept; never more than
 * a tenth of the free memory is held.  (what the search did: gu_trajectory_placement*, include/gu_diag.h.) */
int gu_reserve_trajectory(gu_handle h, int64_t T);
int gu_rollout(gu_handle h, int64_t T, int32_t policy_kind, uint32_t flags);
/* Store pacing: launches that write 128 MB of rows and more keep a schedule -- a wave begins its next 16 steps no earlier than
 * `period` ticks of 10 ns after the last ones were due.  The launches choose the period themselves, closed loop, on the device;
 * results never depend on it.  GU_OPT_ROLLOUT_PACE = 0: no limiter; n > 0: that period, fixed.  (where the
 * loop stands: gu_rollout_pacing, include/gu_diag.h.)  gu_rollout_calibrate is gu_rollout, kept for callers of rounds 3 and 4. */
int gu_rollout_calibrate(gu_handle h, int64_t T, int32_t policy_kind, uint32_t flags);
int gu_read_trajectory(gu_handle h, int64_t t0, int64_t T, int32_t *obs, int32_t *reward, int32_t *done);
int gu_read_trajectory_packed(gu_handle h, int64_t t0, int64_t T, uint32_t *packed);   /* [T][N] after GU_F_PACKED */
int gu_read_stats(gu_handle h, int64_t *reward_sum, int32_t *episodes);

/* ---- state (checkpoint / parity harness) --------------------------------------
 * Any pointer may be NULL.  pos/done int32[N], episode uint32[N], tcount uint64[N]: the steps every env has taken since gu_seed
 * (64 bits: at 4e7 steps per second and env a 32-bit count would wrap, and the action stream repeat, after 97 s; the step counts
 * of one engine must lie within 2^31 of each other -- envs step in lock step, apart only by gu_set_state and rejected actions). */
int gu_get_state(gu_handle h, int32_t *pos, int32_t *done, uint32_t *episode, uint64_t *tcount);
int gu_set_state(gu_handle h, const int32_t *pos, const int32_t *done, const uint32_t *episode, const uint64_t *tcount);

/* Ascending indices of envs whose done flag is set.  Every kernel that writes done[] (step, rollout, reset,
 * sweep-step) also writes its waves' 64-bit ballot of the new flags -- 8 KB at 65 536 envs -- so this call is ONE
 * launch: a single-workgroup scan + expansion of those words straight into page-locked memory (only a done[]
 * installed with gu_set_state needs a ballot pass first).  Batched form of the harness's `if done: env.reset()`
 * bookkeeping (core/algorithms/monte_carlo.py:19-25).  idx has room for N entries. */
int gu_done_indices(gu_handle h, int32_t *idx, int32_t *count);

/* ---- look_step_ahead table queries: env:136-155 for n (state, action) pairs (grid 0 of a multi-grid engine) ---- */
int gu_look_step_ahead(gu_handle h, int64_t n, const int32_t *states, const int32_t *actions,
                       int32_t care_about_terminal, int32_t *next, int32_t *reward, int32_t *done);

/* ---- tabular DP on the engine's grid (float64, bit-exact, no FMA contraction) ----
 * gu_vi_set   : upload v[S] and pi[S][4]                (dynamic_programming.py:12-13)
 * gu_vi_sweep : `iters` x { V1 policy-evaluation sweep  core/algorithms/utils.py:15-27;
 *               optionally V2 greedy improvement        core/algorithms/utils.py:55-72 }
 *               deltas[i] = max(v - v') of sweep i      dynamic_programming.py:17 (may be NULL)
 * gu_vi_run   : the whole value_iteration loop (dynamic_programming.py:14-27) queued in one call: up to
 *               max_steps rounds {V1, delta, V2}; the stopping rule `delta < threshold` is evaluated ON THE
 *               DEVICE after each round and turns the launches queued behind it into no-ops, so the host
 *               synchronises once.  steps_done = rounds executed; deltas[0..steps_done) optional.
 * gu_vi_eval_run : policy_iteration's evaluation loop (dynamic_programming.py:40-42): V1 sweeps with the policy
 *               fixed until `delta < threshold` or max_steps sweeps; steps_done / deltas as for gu_vi_run.
 *               (Grids of up to 4096 states run gu_vi_run / gu_vi_eval_run / gu_vi_sweep as ONE launch of one
 *               workgroup with v in LDS; larger grids take one launch per round.)
 * gu_vi_greedy: V2 alone on the current v (policy improvement without an evaluation sweep)
 * gu_vi_get   : download v / pi (either may be NULL).  Behind a gu_vi_run / gu_vi_sweep / gu_vi_eval_run that ran as the one
 *               launch of one XCD's workgroups this is two memcpys: that launch leaves its final tables in a page-locked copy
 *               on the host, valid until the next call that may write a table.
 * gu_vi_sweep_step : config 5 -- ONE launch that performs one V1+V2 sweep AND one env
 *               step in which every agent acts greedily on the updated policy.
 * gu_vi_sweep_step_run : `iters` such rounds.  Three forms, fastest first: (1) ONE launch synchronised per XCD -- every XCD
 *               sweeps the whole table with its own workgroups and steps its share of the agents, nothing a round needs
 *               leaves the XCD's L2 (table + planes within one workgroup's LDS, one workgroup per CU at most); (2) ONE launch
 *               of a workgroup cluster with a chip-wide barrier per round (max(S, N) <= 1024 x the device's CUs,
 *               S <= 32 767); (3) one launch per round.  A form that cannot hold its workgroups resident together gives up
 *               (bounded spins), the state is put back, the next form runs.  deltas[iters] optional.
 * gu_vi_last_form : which of the three the last gu_vi_sweep_step_run of this engine took (1, 2, 3; 0 = none yet).
 * gu_vi_last_clusters : members[8] = how many workgroups of its last per-XCD launch read each HW_REG_XCC_ID (the clusters as the
 *               hardware reported them; equal under the round-robin placement observed on MI355X -- speed only, any split is correct). */
int gu_vi_set(gu_handle h, const double *v, const double *pi);
int gu_vi_sweep(gu_handle h, double gamma, int32_t iters, int32_t greedy_update, double *deltas);
int gu_vi_run(gu_handle h, double gamma, double threshold, int32_t max_steps, int32_t *steps_done, double *deltas);
int gu_vi_eval_run(gu_handle h, double gamma, double threshold, int32_t max_steps, int32_t *steps_done, double *deltas);
int gu_vi_greedy(gu_handle h, double gamma);
int gu_vi_get(gu_handle h, double *v, double *pi);
int gu_vi_sweep_step(gu_handle h, double gamma, uint32_t flags, double *delta);
int gu_vi_sweep_step_run(gu_handle h, double gamma, int32_t iters, uint32_t flags, double *deltas);

/* ---- Monte-Carlo policy evaluation: core/algorithms/monte_carlo.py:29-99 ----------------
 * Consumes the trajectory rows 0..T-1 of the last gu_rollout (run WITHOUT auto-reset: env e is
 * episode e; its episode ends at its first done row, or after T steps -- run_episode, :7-26) and
 * applies the reference's arithmetic in the reference's order (episode 0, 1, ... N-1):
 *   per episode  : visit counts and summed returns per state, first-visit or every-visit (:54-71);
 *                  return from step idx = sum_i discount_pow[i] * r[idx+i] over the i with keep[i]
 *                  (the caller passes discount_factor**i and (discount_factor**i > threshold) so
 *                  that pow is evaluated by the host language exactly like the reference, :69-70)
 *   across episodes, per state: incremental mean / running mean with alpha / batch mean (:73-97)
 * first_state[N]: the state each episode started in (what reset() returned).  value[S] in/out is
 * NOT read: evaluation starts from zeros like the reference; value_out[S] and visits_out[S]
 * (total_visit_counter, optional) are written.  float64, bit-exact. */
int gu_mc_evaluate(gu_handle h, int64_t T, const int32_t *first_state, int32_t every_visit, int32_t incremental_mean,
                   int32_t stationary_env, double alpha, const double *discount_pow, const uint8_t *keep,
                   double *value_out, double *visits_out);
/* The REFERENCE'S OWN episodes for gu_mc_evaluate: run_episode (core/algorithms/monte_carlo.py:7-26) draws one
 * np.random.choice(4, p=policy[obs]) -- one uniform of numpy's global stream -- per step, episode after episode.  The host
 * pre-draws the uniforms u[K] in bulk; the episode that begins at uniform i in start cell c is a pure function of (i, c), so
 *   gu_mc_walk_lengths  walks it for EVERY offset i < n_offsets and every start cell start_states[n_starts], one lane each:
 *                       lengths[c][i] = steps until done or `cap` (run_episode's max_steps_per_episode; 0xFFFF: the uniforms ran
 *                       out first); cdf[S][4] = the rows np.random.choice builds (p.cumsum() / p.sum()).  The caller follows the chain
 *                       offset -> offset + length through the table, one look-up per episode;
 *   gu_mc_walk_episodes walks episode e (env e of the engine) from uniform offsets[e] and cell first_state[e] and writes its
 *                       (obs, reward, done) rows 0 .. T-1 into the trajectory buffer (rows past its end: the absorbing state, as a
 *                       rollout without auto-reset leaves them), for gu_mc_evaluate.  The engine's env state is not touched. */
int gu_mc_walk_lengths(gu_handle h, int64_t K, const double *u, int64_t n_offsets, int32_t n_starts, const int32_t *start_states,
                       int32_t cap, const double *cdf, uint16_t *lengths);
int gu_mc_walk_episodes(gu_handle h, int64_t K, const double *u, const double *cdf, const int64_t *offsets, const int32_t *first_state,
                        int32_t cap, int64_t T);

/* ---- shortest paths: the breadth-first search of core/algorithms/maze_solving.py:43-50, 123-193 for EVERY grid ----
 * One lane per grid searches from the grid's first start cell over the care_about_terminal=False move graph
 * (children in action order, FIFO), stops at the first terminal state (goal or lava) it dequeues and writes the
 * action list.  path[n_grids][max_path] int8, path_len[n_grids] (-1 no terminal reachable, -2 longer than
 * max_path), terminal[n_grids] (optional) the state reached. */
int gu_shortest_paths(gu_handle h, int32_t max_path, int8_t *path, int32_t *path_len, int32_t *terminal);

/* ---- headless RGB frames (stands in for the pyglet window of core/envs/rendering.py:236-343) -----------------
 * rgb[n_envs][H*cell_px][W*cell_px][3] uint8 for envs env0 .. env0+n_envs-1: ground / wall / goal / lava tiles -- which of
 * the four a cell gets follows the viewer's rule (rendering.py:119-133: goal, else lava, else wall, else ground); the flat
 * colours that stand in for its textures are build-defined -- a grid line on the top and left edge of each cell
 * (cell_px >= 4), the agent as an inset square on its cell. */
int gu_render_rgb(gu_handle h, int64_t env0, int64_t n_envs, int32_t cell_px, uint8_t *rgb);
/* The agent trail: core/envs/griduniverse_env.py:92-93, 182-184, 190 (`last_n_states`: the cell the agent is on after every
 * step, newest 500 kept, emptied by reset) + core/envs/rendering.py:287-311 (every frame: newest first, a quad over the entry's
 * tile with alpha 0.3 * 0.96^(i + 1), entries on the agent's current cell skipped).  OFF by default; gu_trail_enable(capacity
 * 1 .. 500; 0 = off again) makes the step / reset / rollout launches keep a ring per env -- small kernels of their own behind the
 * launch; a rollout must then write rows (GU_F_TRAJECTORY or GU_F_PACKED), the fused sweep + step launches are refused --
 * and gu_render_rgb blend it over the tiles: corner colours red, yellow, green, blue (bottom-left, counter-clockwise) interpolated
 * bilinearly at the pixel centre, c = (c * (65536 - A) + colour * A + 32768) >> 16 per entry with A = round(alpha * 65536).
 * gu_trail_read: cells[n_envs][capacity] oldest first (-1 beyond length[k]), like the reference's list of states. */
int gu_trail_enable(gu_handle h, int32_t capacity);
int gu_trail_read(gu_handle h, int64_t env0, int64_t n_envs, int32_t *cells, int32_t *length);
/* The policy-arrow figure of Viewer.render_policy_arrows (core/envs/rendering.py:159-212) for the current policy table
 * (gu_vi_set / gu_vi_run ...): rgb[H*cell_px][W*cell_px][3], the tiles of grid 0 plus, on every state that is neither
 * terminal nor a wall, one arrow per action with probability >= 0.1: shaft from the tile centre, round(p*20) long
 * (half to even), head a triangle of half-width 5 and height 5 on its end, on the reference's 52-pixel tile.
 * Rasterisation (the reference leaves it to OpenGL): coordinates scale by cell_px / 52 about the tile centre; a pixel is
 * painted when its centre lies within max(1, cell_px / 26) / 2 of the shaft segment, or inside the head triangle (edges
 * included, the base edge excluded).  Probabilities are expected in [0, 1] (an arrow never leaves its tile then). */
int gu_render_policy_rgb(gu_handle h, int32_t cell_px, uint8_t *rgb);

/* ---- page-locked host memory -----------------------------------------------------
 * Buffers from gu_host_alloc make gu_step (with GU_F_PINNED_IO), gu_read_outputs and gu_read_trajectory copy at
 * the full PCIe rate; numpy arrays can be built on them (np.ctypeslib / np.frombuffer). */
int gu_host_alloc(size_t bytes, void **ptr);
int gu_host_free(void *ptr);

/* ---- stream ---------------------------------------------------------------------
 * (HIP-event timers on the handle's own stream -- torch.cuda.Event cannot see it --: gu_timer_*, include/gu_diag.h.) */
int gu_sync(gu_handle h);

/* ---- multi-GPU gathered view (RCCL over xGMI) ----------------------------------
 * One process per GPU.  Rank 0 calls gu_comm_unique_id and ships the 128 bytes to
 * the other ranks by any host channel; every rank then calls gu_comm_init.  The
 * data path (step / rollout) never communicates; only gu_allgather_view does:
 * one ncclAllGather of the packed int32[3N] (obs|reward|done) block per rank, then
 * a D2H copy into three host arrays of nranks*N.  All ranks must hold the same N. */
#define GU_COMM_ID_BYTES 128
int gu_comm_unique_id(uint8_t id[GU_COMM_ID_BYTES]);
int gu_comm_init(gu_handle h, int32_t nranks, int32_t rank, const uint8_t id[GU_COMM_ID_BYTES]);
int gu_comm_destroy(gu_handle h);
int gu_allgather_view(gu_handle h, int32_t *obs_all, int32_t *reward_all, int32_t *done_all);
/* The same for ONE process that holds one handle per device (handles[i] on a distinct device, equal N):
 * ncclCommInitAll, then one grouped ncclAllGather; the view is copied out of handles[0]'s device. */
int gu_comm_init_all(gu_handle *handles, int32_t n);
int gu_allgather_view_all(gu_handle *handles, int32_t n, int32_t *obs_all, int32_t *reward_all, int32_t *done_all);

#ifdef __cplusplus
}
#endif
#endif /* GU_H */
