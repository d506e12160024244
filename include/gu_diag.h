/* gu_diag.h -- introspection and measurement aids of libgu.so: NOT part of the surface a reference maintainer binds
 * (include/gu.h is).  Everything here reports what the engine did -- which buffer the placement search kept,
 * where the closed loop of the store pacing stands, which form a DP call took, HIP-event timers on the engine's own stream --
 * or tunes it for an experiment; nothing here changes a result.  bench.py, tools/ and the tests use it; same conventions as gu.h.
 */
#ifndef GU_DIAG_H
#define GU_DIAG_H

#include "gu.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- trajectory placement: what gu_reserve_trajectory's search did ------------------
 * gu_trajectory_placement reports the outcome of the search include/gu.h describes, gu_trajectory_placement_detail everything that was tried: per candidate its
 * probe time (ms per full write) and device address (capacity entries; *count = number tried), the index of the kept one,
 * the wall time the search took and the largest number of bytes it held at once.  Any pointer may be NULL.
 * gu_probe_trajectory re-runs the same timed write on the buffer the engine holds NOW (it overwrites the rows: for
 * measurements -- bench.py runs it right after its timed region to tell a drifting device from a slow kernel). */
int gu_trajectory_placement(gu_handle h, int32_t *candidates, float *best_ms, float *worst_ms);
int gu_trajectory_placement_detail(gu_handle h, int32_t capacity, float *probe_ms, uint64_t *address, int32_t *count,
                                   int32_t *kept, float *search_ms, uint64_t *peak_bytes);
int gu_probe_trajectory(gu_handle h, float *milliseconds);

/* ---- store pacing: where the closed loop stands (include/gu.h, "Store pacing") ---------
 * gu_rollout_pacing reports a launch kind's current period, the length of its schedule (ms_paced; ms_unpaced and
 * calibration_ms are 0), and how many launches of the kind have run on the current shape (`evaluated`); GU_ERR_STATE when the kind
 * keeps no schedule.  gu_rollout_pacing_totals: kinds with a schedule; calibration_ms and launches_spent are 0 (no launch
 * is ever spent on a search; the open-loop search of rounds 3 and 4 left the library in round 6). */
int gu_rollout_pacing_totals(gu_handle h, float *calibration_ms, int32_t *launches_spent, int32_t *kinds_paced, int32_t *kinds_from_cache,
                             int32_t *kinds_waiting);
int gu_rollout_pacing(gu_handle h, int32_t policy_kind, uint32_t flags, int32_t *period, float *ms_unpaced, float *ms_paced,
                      int32_t *evaluated, float *calibration_ms);
/* The records of the last launches of a kind, oldest first (at most 61; waits for the stream): per launch eight 64-bit words --
 * launch number; period in 1/64 ticks (0: the launch ran without the limiter); verdict of the launch behind it (0 none yet, 1 on
 * schedule, 2 behind) | phase of the loop << 8 (0 limiter on, 1 three launches without it, 2 limiter off, 3 six launches with it);
 * waves that reported; ticks from start to report of the slowest wave; waves that were more than two periods behind; the most a
 * wave was behind (ticks); ticks from this launch's start to the next one's (0 for the last).  *launches = launches of the kind
 * on the current shape. */
int gu_rollout_pace_log(gu_handle h, int32_t policy_kind, uint32_t flags, int32_t capacity, uint64_t *entries, int32_t *count, uint32_t *launches);
/* MEASUREMENT AID: what every wave of the kind's LAST launch reported -- ticks (10 ns) from the wave's start to its report, a few
 * groups before the end of the launch (0: the wave did not report) -- before the next launch sums and clears it. */
int gu_rollout_pace_waves(gu_handle h, int32_t policy_kind, uint32_t flags, int32_t capacity, uint32_t *elapsed, int32_t *count);

/* ---- tabular DP: which form ran ----------------------------------------------------
 * gu_vi_last_form : which of its three forms the last gu_vi_sweep_step_run of this engine took (1 per XCD, 2 chip-wide cluster,
 *   3 one launch per round; 0 = none yet).
 * gu_vi_last_clusters : members[8] = how many workgroups of its last per-XCD launch read each HW_REG_XCC_ID (the clusters as the
 *   hardware reported them). */
int gu_vi_last_form(gu_handle h);
/* ... and the last gu_vi_sweep / gu_vi_run / gu_vi_eval_run: 1 per XCD (one cluster's workgroups), 2 one workgroup, 3 chip-wide
 * cluster, 4 one launch per round (the form that finished the call; 0 = none yet). */
int gu_vi_last_dp_form(gu_handle h);
int gu_vi_last_clusters(gu_handle h, int32_t *members);
/* A -DGU_VI_XCD_TORN build of the library (tools/xcd_stress.py) counts the words of the per-XCD exchange that arrived with the right
 * round tag and the WRONG payload -- a 16-byte store whose 8-byte half was torn at four bytes -- over all launches of this engine.
 * The product library returns GU_ERR_UNSUPPORTED (*count = -1). */
int gu_vi_xcd_torn_words(gu_handle h, int64_t *count);

/* ---- timing: HIP events on the handle's own stream (torch.cuda.Event cannot see it) ----- */
int gu_timer_begin(gu_handle h);
int gu_timer_end(gu_handle h, float *milliseconds); /* records, waits, returns elapsed */
/* Lap timing: gu_timer_mark records one event on the stream per call (async); gu_timer_laps waits for the last mark,
 * writes the count-1 intervals between consecutive marks (capacity = room in `milliseconds`, which may be NULL to
 * discard) and forgets the marks. */
int gu_timer_mark(gu_handle h);
int gu_timer_laps(gu_handle h, float *milliseconds, int32_t capacity, int32_t *count);

/* ---- options that only measurements and tuning touch (gu_set_option, include/gu.h) -------- */
#define GU_OPT_TRAJ_PROBE_ALL 17      /* 1 = probe every candidate, no early stop (measurement aid)                           */
#define GU_OPT_PACE_TARGET 20         /* closed-loop store pacing: GB/s of rows the first launch of a kind is scheduled for (7200)  */
#define GU_OPT_PACE_BAR_NUM 21        /* ... the log calls a launch BEHIND when a wave was more than this many 256ths of the schedule late (20) */
#define GU_OPT_PACE_GAIN_Q 22         /* ... 1/64 ticks: the period's step up after a launch whose waves ALL fell behind (128); a share of
                                         the waves: that share of it                                                          */
#define GU_OPT_PACE_DEC_Q 23          /* ... 1/64 ticks: what the period comes down by, every launch (8; the slow loop moves it)    */
#define GU_OPT_PACE_RECORD 25         /* 0 = launches with a FIXED period keep no record (measurement aid: what the records cost)   */
#define GU_OPT_PACE_PROBE_EVERY 26    /* launches between two looks at whether the limiter pays at all: three launches without it while it
                                         is on, six with it (every 2 x this) while it is off (1024; 0 = never: the limiter stays on)   */
#define GU_OPT_PACE_ADAPT 27          /* 1 (default): the share of waves behind that the rule aims for follows the measured start-to-start
                                         time of the launches (blocks of 192); 0: it stays at GU_OPT_PACE_DEC_Q / GU_OPT_PACE_GAIN_Q   */
#define GU_OPT_X_TRAJ_UNCACHED 100    /* EXPERIMENT: uncached memory type for the trajectory (readers may see stale bytes)    */
#define GU_OPT_X_TRAJ_POISON 101      /* EXPERIMENT: fill a fresh trajectory buffer with 0x5A                                 */
#define GU_OPT_X_MC_POISON 102        /* EXPERIMENT: fill the Monte-Carlo scratch with 0x5A before every evaluation           */
#define GU_OPT_X_COUNT 3

#ifdef __cplusplus
}
#endif
#endif /* GU_DIAG_H */
