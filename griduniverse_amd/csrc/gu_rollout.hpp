// gu_rollout.hpp -- the fused rollout kernel template and its launch dispatch.  Included by one small translation
// unit per policy kind (gu_rollout_*.hip), so that the template grid (policy x auto-reset x trajectory x stats x map)
// compiles in parallel; gu_kernels.hip only sees the four entry points declared at the end.
#pragma once
#include "gu_map.hpp"

#include <cstdlib>
#include <functional>
#include <type_traits>

// ------------------------------------------------------------------------------------
// fused rollout: T env-steps per lane in one launch
//   algorithmic HBM bytes per env-step with GU_F_TRAJECTORY: 3 x 4 B row writes = 12 B
//   (+0.25 B for GU_POLICY_STREAM: the caller's actions are read packed, two bits each); state is loaded/stored once per launch.
// ------------------------------------------------------------------------------------
// Inverse CDF on one 32-bit word: a = #{k < 3 : word >= t_k} over the thresholds that a word can reach.  A threshold that no
// word reaches (cumulative probability >= 1: every one-hot or zero-tailed row has some) is stored as 0 -- which every word
// passes -- and counted in q.w, so the three compares need no per-threshold mask: three v_cmp + carry adds and one subtract.
__device__ __forceinline__ uint32_t gu_sample_action(uint32_t word, const uint4 q)
{
    return (uint32_t)(word >= q.x) + (uint32_t)(word >= q.y) + (uint32_t)(word >= q.z) - q.w;
}


// The rate limiter of the trajectory store stream (int32 rows; packed rows on the transition-row kernel): every wave keeps a SCHEDULE.  `pace` = ticks of the constant 100 MHz clock
// (s_memrealtime, 10 ns) per 16 steps; a wave may begin its next group of steps no earlier than its own start + steps done x
// pace / 16, and does not wait at all when it is late.
//
// WHY (round 3; tools/archive/micro/store_pacing.hip, write_ceiling.hip, store_deadline.hip, tools/pace_ab.py, profiles/archive/r03d_*, r03k_*):
// the HBM write path of an MI355X shows CONGESTION COLLAPSE.  65 536 lanes that hand their three rows per step to the memory
// system as fast as it will take them keep every queue on the way full, and the sustained rate then DROPS: to 5.7 TB/s on most
// allocations (round 2's "slow class"), 6.6 on some.  The same stores offered just below the memory's capacity go through at
// 7.2 .. 7.5 TB/s on EVERY allocation.  The first limiter (until r03j) idled a fixed number of scalar-loop turns every four
// steps: it works only while no wave is ever held up -- every wave idles the same amount whether it is ahead or behind, so
// waves that were blocked stay behind, the rows in flight spread out and the collapse feeds itself; its best setting was the
// one at which the waves' own pace equalled the memory's (113 .. 116 us per 65 536 x 1000 launch on slow allocations), one turn
// of 33 clocks less and the launch collapsed.  A schedule has neither problem: late waves catch up, the waves stay within a
// few rows of each other, and the period has a resolution of 0.6 % (one tick in ~170).  Same kernel, same buffers: 107 .. 110 us
// (7.2 .. 7.35 TB/s) on slow allocations, 105 .. 107 on fast ones.  One clock read per 16 steps: a read per step costs more than
// it saves (s_memtime per step: 146 us), per 16 steps it is not measurable.
//  * 100 MHz, not the shader clock (s_memtime runs at the engine clock, which moves with load and power).
//  * WHO CHOOSES THE PERIOD (round 5): the launches themselves, closed loop, on the device.  Rounds 3 and 4 searched it with a few
//    hundred dedicated full-size launches on a snapshot of the engine's state (on request, or after 1024 launches of a kind) and
//    then held it open loop.  Now every launch kind of an engine owns a ring of launch records and two sets of per-wave slots in
//    device memory (gu_internal.hpp: GuPaceEntry).  A few groups before its end every wave stores ONE word into its own slot: was
//    it more than two periods behind its schedule, and by how much.  The FIRST WAVE of the next launch of the kind -- in the time
//    before its own first step -- sums the slots and writes the period of the launch after it into that launch's record
//    (GuPacer::decide); every wave of a launch reads its launch's record when it starts.  No host round
//    trip, no dedicated launch, no snapshot, no atomics, nobody waits for anybody; what a launch does is felt two launches later.
//    The first launch of a kind starts from a model (the rows of 16 steps at 7.2 TB/s).
//  * THE RULE.  Per launch:   period += gain x (share of the waves more than two periods behind)  -  dec     (1/64 ticks)
//    -- stochastic approximation (Robbins-Monro): the period settles where the MEAN share of waves behind is dec / gain (the
//    defaults: 8 / 128 = 6 %).  Why the share of waves: near the cliff of this memory launches fall behind sporadically -- a
//    few per cent of them even 10 ticks above it, a workgroup or two each time, 4 .. 8 us late (0.5 % of the waves on average) --
//    while below it most waves of every launch do; the mean share rises from 0.5 % to 3 % to > 50 % within ten ticks, so 6 % is
//    reached within a tick or two of the period at which the mean launch time is shortest, on every allocation, and five times the
//    background keeps the loop from creeping up on a noisy device (profiles/archive/r05c_pace_c3.txt).  The first launches come down
//    faster: dec is at least 8 / (8 + seq) tick.  And the loop keeps asking whether the limiter pays at all (decide()).
// (History, all in profiles/archive/r05*_pace_*.txt.  Version 1 moved one whole tick per launch and kept a "period known to fail" with
// exponential back-off: two unlucky launches in a row doubled the back-off twice, and the period drifted up by 8 ticks in 300
// launches and stayed there.  Version 2 stepped up by two ticks per launch behind and down by 1/32 tick: 4 ticks = 2.3 % above
// the best fixed period, because a launch behind costs 8 us here, not the 35 us that ratio was chosen for.  Version 3 stepped up
// in proportion to the slowest wave's distance behind: the background events alone held it 6 ticks above the best period.)
// sum / maximum of a 32-bit value over the wave, in lane 63: DPP row shifts inside each row of 16 lanes, then the two row
// broadcasts of the gfx9 family -- register moves inside the SIMD (six instructions per reduction; a __shfl_xor butterfly is six
// dependent ds_bpermute round trips, ~0.2 us each way of a wave that has its SIMD alone)
#define GU_DPP_REDUCE(x, OP)                                                                                     \
    do {                                                                                                         \
        uint32_t o_;                                                                                             \
        o_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), 0x111, 0xF, 0xF, false), (x) = OP((x), o_);      \
        o_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), 0x112, 0xF, 0xF, false), (x) = OP((x), o_);      \
        o_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), 0x114, 0xF, 0xF, false), (x) = OP((x), o_);      \
        o_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), 0x118, 0xF, 0xF, false), (x) = OP((x), o_);      \
        o_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), 0x142, 0xA, 0xF, false), (x) = OP((x), o_);      \
        o_ = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), 0x143, 0xC, 0xF, false), (x) = OP((x), o_);      \
    } while (0)
#define GU_OP_ADD(a_, b_) ((a_) + (b_))
#define GU_OP_MAX(a_, b_) ((a_) > (b_) ? (a_) : (b_))

struct GuPacer {
    uint64_t due, t0;
    uint32_t ticks;
    uint32_t report_at;  // the wave reports behind the group that completes this many steps (0: it has, or there is nothing to report to)
    uint32_t n_steps;
    uint32_t p_q;
    uint64_t *slot;      // this wave's slot in the launch's set
    // what the launch's first wave asks for at the very top of the kernel, to have it by the time it decides (start()): the first
    // 1024 slots of the launch before (sixteen per lane), and the loop's state in this launch's and that launch's records
    uint64_t w[16];
    uint32_t c_seq, c_phase, c_left, c_ema_p, c_ema_u, v_period_q, v_seq, v_unpaced, v_report_steps, v_groups;
    uint32_t c_dec_q, c_block_left, c_block_sum, c_block_n, c_last_mean, c_up, c_quiet;
    uint64_t v_t_start;
    bool first_wave;
    // at the very top of the kernel: ask for this launch's record, so that the answer is there when start() wants it
    __device__ __forceinline__ void fetch(const GuPaceArgs &pa, bool on)
    {
        ticks = on ? pa.period : 0u;
        p_q = pa.period << 6;
        first_wave = false;
        if (on && pa.ring) {
            const GuPaceEntry *entry = pa.ring + (pa.seq & (GU_PACE_RING - 1u));
            const GuPaceEntry *prev = pa.ring + ((pa.seq - 1u) & (GU_PACE_RING - 1u));
            c_seq = entry->seq;  // (scalar loads, all of them)
            const bool mine = c_seq == pa.seq;
            const uint32_t have = mine ? entry->period_q : 0u;
            const uint32_t unpaced = mine ? entry->unpaced : 0u;
            c_phase = entry->phase, c_left = entry->left, c_ema_p = entry->ema_paced, c_ema_u = entry->ema_unpaced;
            v_period_q = prev->period_q, v_seq = prev->seq, v_unpaced = prev->unpaced, v_t_start = prev->t_start;
            v_report_steps = prev->report_steps, v_groups = prev->groups;
            if (have && !pa.fixed) p_q = have;
            p_q = p_q < (pa.lo << 6) ? (pa.lo << 6) : p_q;
            p_q = p_q > (pa.hi << 6) ? (pa.hi << 6) : p_q;
            if (pa.fixed) p_q = pa.period << 6;
            ticks = (unpaced && !pa.fixed) ? 0u : (p_q + 32u) >> 6;
            c_quiet = (mine && !pa.fixed) ? entry->quiet : 0u;
            first_wave = blockIdx.x == 0 && __builtin_amdgcn_readfirstlane(threadIdx.x) < 64u;
            if (first_wave && !c_quiet) {
                c_dec_q = entry->dec_q, c_block_left = entry->block_left, c_block_sum = entry->block_sum, c_block_n = entry->block_n;
                c_last_mean = entry->last_mean, c_up = entry->up;
                const uint64_t *set = pa.slots + (size_t)((pa.seq - 1u) & 1u) * pa.slot_stride;
#pragma unroll
                for (uint32_t j = 0; j < 16u; ++j) {
                    const uint32_t i = j * 64u + threadIdx.x;
                    w[j] = i < pa.n_waves ? __builtin_nontemporal_load(set + i) : 0ull;
                }
            }
        }
    }
    // right before the first step.  The launch's FIRST WAVE sums what the launch before this one reported and writes the record of
    // the launch behind this one (decide).  Here, not in the step loop: inlined into the loop's back edge the same code cost
    // every kernel 30 .. 50 registers and spills (the row-table kernel: from 70 SGPRs and none spilled to 106 and 28 spilled, its
    // launch without a limiter from 42.7 to 48.1 us); and with
    // everything it reads asked for at the top of the kernel, because a wave that starts its steps 2 us late ends 2 us late when
    // the launch runs without the limiter (nobody waits, nobody catches up: packed rows 42 -> 45 us).
    __device__ __forceinline__ void start(const GuPaceArgs &pa, bool on)
    {
        slot = nullptr;
        n_steps = 0;
        report_at = 0;
        t0 = 0;
        if (on && pa.ring) {
            const uint64_t now = __builtin_amdgcn_s_memrealtime();
            due = t0 = now;
            if (c_quiet) {  // the limiter is off and stays off: nobody reports, the first wave counts the launch
                if (first_wave) decide_quiet(pa, now);
                return;
            }
            const uint32_t wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
            slot = pa.slots + (size_t)(pa.seq & 1u) * pa.slot_stride + wave;
            report_at = pa.report_at;
            if (first_wave) decide(pa, now);  // (unless decide_early has done it)
            return;
        }
        due = ticks ? __builtin_amdgcn_s_memrealtime() : 0ull;
    }
    // a kernel some of whose lanes leave before start() (the transition-row kernel's half waves): the first wave decides while it is
    // still whole
    __device__ __forceinline__ void decide_early(const GuPaceArgs &pa)
    {
        if (first_wave && !c_quiet) decide(pa, __builtin_amdgcn_s_memrealtime());
    }
    // A launch of a kind whose limiter is off (GU_PACE_OFF) and stays off: the record of the launch behind this one is this one's,
    // one launch further along; the mean start-to-start time of the launches without the limiter is kept up (the next comparison
    // with the limiter wants it fresh).  One lane, a dozen scalar operations, two stores.
    __device__ __forceinline__ void decide_quiet(const GuPaceArgs &pa, uint64_t now)
    {
        first_wave = false;
        if (threadIdx.x == 0) {
            GuPaceEntry *cur = pa.ring + (pa.seq & (GU_PACE_RING - 1u));
            GuPaceEntry *next = pa.ring + ((pa.seq + 1u) & (GU_PACE_RING - 1u));
            uint32_t ema_u = c_ema_u;
            const uint64_t gap = now - v_t_start;
            const uint32_t took = gap > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)gap;
            if (v_seq + 1u == pa.seq && v_unpaced && v_t_start && v_groups == pa.groups && (!ema_u || took < 2u * ema_u)) ema_u = ema_u ? (ema_u * 7u + took) >> 3 : took;
            cur->t_start = now, cur->groups = pa.groups, cur->report_steps = 0;
            const uint32_t left = c_left > 0u ? c_left - 1u : 0u;
            next->period_q = p_q, next->seq = pa.seq + 1u, next->t_start = 0, next->unpaced = 1u;
            next->phase = GU_PACE_OFF, next->left = left, next->ema_paced = c_ema_p, next->ema_unpaced = ema_u;
            next->verdict = 0, next->waves = 0, next->elapsed = 0, next->ended_late = 0, next->max_behind = 0, next->report_steps = 0, next->groups = 0;
            next->dec_q = pa.dec_q, next->block_left = GU_PACE_BLOCK, next->block_sum = 0, next->block_n = 0, next->last_mean = 0, next->up = 0;
            next->quiet = left > 1u ? 1u : 0u;  // (the launch that ends the phase is a full one: it sets the probe up)
        }
    }
    // The launch's first wave, once: what the waves of the launch BEFORE this one reported (set (seq - 1) & 1), summed over the
    // wave's lanes; that launch's log; how the launch BEHIND this one runs.  The set is cleared as it is read: the launch behind
    // this one reports into it.
    __device__ __forceinline__ void decide(const GuPaceArgs &pa, uint64_t now)
    {
        first_wave = false;
        uint64_t *set = pa.slots + (size_t)((pa.seq - 1u) & 1u) * pa.slot_stride;
        GuPaceEntry *cur = pa.ring + (pa.seq & (GU_PACE_RING - 1u));
        GuPaceEntry *prev = pa.ring + ((pa.seq - 1u) & (GU_PACE_RING - 1u));
        const bool fresh = c_seq != pa.seq;  // the kind's first launch on this shape: nobody wrote its record
        // the launch before: its schedule up to the report (0: it ran without the limiter, nobody was "behind")
        const bool prev_ok = v_seq + 1u == pa.seq;
        const uint32_t prev_ticks = (prev_ok && !v_unpaced) ? (pa.fixed ? pa.period : (v_period_q + 32u) >> 6) : 0u;
        const uint32_t scheduled = (prev_ticks * ((v_report_steps + 15u) & ~15u)) >> 4;  // (ITS report point: launches of a kind may differ in length)
        const bool same_length = v_groups == pa.groups;  // (times of launches of different lengths are not compared)
        uint32_t counts = 0, most = 0, longest = 0;  // counts: waves that reported << 16 | waves that were far behind
        for (uint32_t base = 0; base < pa.n_waves; base += 64u * 16u) {
            if (base) {  // (batches of more than 1024 waves: the slots behind the first 1024, sixteen per lane and round trip)
#pragma unroll
                for (uint32_t j = 0; j < 16u; ++j) {
                    const uint32_t i = base + j * 64u + threadIdx.x;
                    w[j] = i < pa.n_waves ? __builtin_nontemporal_load(set + i) : 0ull;
                }
            }
#pragma unroll
            for (uint32_t j = 0; j < 16u; ++j) {
                const uint32_t i = base + j * 64u + threadIdx.x;
                if (i < pa.n_waves) set[i] = 0;
                const uint32_t reported = (uint32_t)(w[j] >> 63), elapsed = (uint32_t)w[j] & 0x7FFFFFFFu;
                const uint32_t behind = (prev_ticks && elapsed > scheduled) ? elapsed - scheduled : 0u;
                counts += (reported << 16) + ((reported && behind > 2u * prev_ticks) ? 1u : 0u);
                most = behind > most ? behind : most;
                longest = elapsed > longest ? elapsed : longest;
            }
        }
        GU_DPP_REDUCE(counts, GU_OP_ADD);
        GU_DPP_REDUCE(most, GU_OP_MAX);
        GU_DPP_REDUCE(longest, GU_OP_MAX);
        if (threadIdx.x == 63) {  // (the lane the reductions end in)
            const uint32_t waves = counts >> 16, far = counts & 0xFFFFu;
            GuPaceEntry *next = pa.ring + ((pa.seq + 1u) & (GU_PACE_RING - 1u));
            uint32_t n_q = p_q, phase = GU_PACE_NORMAL, left = pa.probe_every ? (pa.probe_every < 128u ? pa.probe_every : 128u) : 0u, ema_p = 0, ema_u = 0;
            uint32_t dec_q = pa.dec_q, block_left = GU_PACE_BLOCK, block_sum = 0, block_n = 0, last_mean = 0, up = 0;
            if (!fresh) {
                phase = c_phase, left = c_left, ema_p = c_ema_p, ema_u = c_ema_u;
                dec_q = c_dec_q, block_left = c_block_left, block_sum = c_block_sum, block_n = c_block_n, last_mean = c_last_mean, up = c_up;
            } else {
                cur->unpaced = 0, cur->phase = phase, cur->left = left, cur->ema_paced = 0, cur->ema_unpaced = 0;
                cur->dec_q = dec_q, cur->block_left = block_left, cur->block_sum = 0, cur->block_n = 0, cur->last_mean = 0, cur->up = up;
            }
            cur->period_q = p_q, cur->seq = pa.seq, cur->t_start = now, cur->report_steps = pa.report_at, cur->groups = pa.groups;
            cur->verdict = 0, cur->waves = 0, cur->elapsed = 0, cur->ended_late = 0, cur->max_behind = 0;
            if (waves && prev_ok) {
                prev->verdict = (uint64_t)most * 256u > (uint64_t)prev_ticks * v_groups * pa.bar_num ? 2u : 1u;
                prev->waves = waves, prev->elapsed = longest, prev->ended_late = far, prev->max_behind = far ? most : 0u;
                // the launch before, from its start to this launch's: what a caller who launches back to back pays per launch.  (A
                // gap on the host stretches it: an interval of more than twice the mean is not taken in.)
                const uint64_t gap = now - v_t_start;
                const uint32_t mean = v_unpaced ? ema_u : ema_p;
                const uint32_t took = gap > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)gap;
                const bool plausible = same_length && v_t_start && (!mean || took < 2u * mean);
                if (v_unpaced) {
                    if (plausible) ema_u = ema_u ? (ema_u + took) >> 1 : took;
                } else {
                    if (plausible) ema_p = ema_p ? (ema_p * 7u + took) >> 3 : took;
                    // THE RULE (above): up by the share of waves that fell behind, down by a fraction of a tick
                    const uint32_t step_up = (uint32_t)(((uint64_t)pa.gain_q * far) / waves);
                    const uint32_t early = 512u / (8u + (pa.seq > 100000u ? 100000u : pa.seq));  // 64 x 8 / (8 + seq)
                    const uint32_t dec = early > dec_q ? early : dec_q;
                    n_q += step_up;
                    n_q = n_q > dec ? n_q - dec : n_q;
                    // THE SLOW LOOP AROUND IT: which share of waves behind is the best one to AIM for differs from buffer to buffer
                    // (how often a launch falls behind at a given distance from the cliff, and what that costs: on one allocation
                    // the rule's 6 % held the period 8 ticks = 3 % above the best one, profiles/archive/r05f_pytest_slow_buffer.txt).  What
                    // counts is the time from one launch's start to the next one's, and the first wave sees it: blocks of 192
                    // launches (the last 128 counted), each with its own aim -- dec_q moves by a factor of 4/3 per block, in the
                    // direction of the last move while the block's mean interval got shorter, the other way when it got longer.
                    // The first move is DOWN, towards the longer period: the launch time rises by 0.6 us per two ticks above the best
                    // period and by 2 .. 7 us per two ticks below it (profiles/archive/r05y_pace_aim.txt; until late in round 5 the first move
                    // was up, by 3/2, and the launches 192 .. 384 of a kind -- a benchmark's -- ran with 10 % of their waves behind).
                    if (phase == GU_PACE_NORMAL && !pa.fixed && pa.adapt) {
                        if (block_left <= GU_PACE_BLOCK - GU_PACE_BLOCK_SKIP && plausible) block_sum += took, ++block_n;
                        if (block_left) --block_left;
                        if (!block_left) {
                            const uint32_t mean = block_n ? block_sum / block_n : 0u;
                            if (mean && last_mean && mean > last_mean) up ^= 1u;
                            if (mean) last_mean = mean;
                            dec_q = up ? (dec_q * 4u + 2u) / 3u : (dec_q * 3u) / 4u;
                            dec_q = dec_q < 4u ? 4u : dec_q;
                            dec_q = dec_q > pa.gain_q / 8u ? pa.gain_q / 8u : dec_q;  // (aims of 4 / gain .. 12.5 %)
                            block_left = GU_PACE_BLOCK, block_sum = 0, block_n = 0;
                        }
                    }
                }
            }
            n_q = n_q < (pa.lo << 6) ? (pa.lo << 6) : n_q;
            n_q = n_q > (pa.hi << 6) ? (pa.hi << 6) : n_q;
            // DOES THE LIMITER PAY AT ALL?  A kind bound by its own dependent chain, not by the memory (packed rows at one wave per
            // SIMD: 42 us without the limiter, 48 at the period the rule settles on -- the waves' natural pace) is better off
            // without it.  So every `probe_every` launches (the first time after 128, when the period has come down from the model)
            // four launches run without the limiter; when they come round quicker (by 2 %, start to start) than the launches with
            // it do on average, the kind runs WITHOUT it, and the limiter gets eight launches to prove itself again every
            // 2 x probe_every.  Four launches in a thousand: 0.05 % of a kind that is better off with the limiter.
            if (pa.probe_every && !pa.fixed && left && --left == 0u) {
                const bool off_wins = ema_u && ema_p && (uint64_t)ema_u * 100u < (uint64_t)ema_p * 98u;
                if (phase != GU_PACE_NORMAL || off_wins) block_left = GU_PACE_BLOCK, block_sum = 0, block_n = 0, last_mean = 0;  // (a block starts over behind a probe)
                if (phase == GU_PACE_NORMAL) phase = GU_PACE_PROBE_OFF, left = 4u, ema_u = 0u;
                else if (phase == GU_PACE_OFF) phase = GU_PACE_PROBE_ON, left = 8u, ema_p = 0u;
                else if (off_wins) phase = GU_PACE_OFF, left = 2u * pa.probe_every;
                else phase = GU_PACE_NORMAL, left = pa.probe_every;
            }
            next->period_q = n_q, next->seq = pa.seq + 1u, next->t_start = 0;
            next->unpaced = (phase == GU_PACE_PROBE_OFF || phase == GU_PACE_OFF) ? 1u : 0u;
            next->phase = phase, next->left = left, next->ema_paced = ema_p, next->ema_unpaced = ema_u;
            next->dec_q = dec_q, next->block_left = block_left, next->block_sum = block_sum, next->block_n = block_n, next->last_mean = last_mean, next->up = up;
            next->quiet = (phase == GU_PACE_OFF && left > 1u) ? 1u : 0u;
            next->verdict = 0, next->waves = 0, next->elapsed = 0, next->ended_late = 0, next->max_behind = 0, next->report_steps = 0, next->groups = 0;
        }
    }
    // The wave's report: ONE plain 8-byte store into its own slot (gu_internal.hpp).  Made `report_at` steps into the launch, i.e.
    // a few groups BEFORE its end, so that nothing of it is in flight when the kernel wants to complete.
    __device__ __forceinline__ void report(uint64_t now)
    {
        if ((threadIdx.x & 63u) == 0u) {
            const uint64_t elapsed = now - t0;
            *slot = (1ull << 63) | (elapsed > 0x7FFFFFFFull ? 0x7FFFFFFFull : elapsed);
        }
        report_at = 0;
    }
    // a launch that keeps no schedule and reports nothing: after() does nothing for it.  (The transition-row kernel runs such a launch
    // on a copy of its loops WITHOUT the call: config 2 and the packed rows at one wave per SIMD are bound by the issue of their own
    // instructions, one per four clocks, and the test alone -- a taken branch around the schedule's code -- cost them ~60 clocks per
    // group of sixteen steps.)
    __device__ __forceinline__ bool idle() const { return ticks == 0u && slot == nullptr; }
    // `steps` steps have just been done (rows stored): wait until their time is up
    __device__ __forceinline__ void after(uint32_t steps)
    {
        if (ticks) {
            due += (ticks * steps) >> 4;
            const uint64_t now = __builtin_amdgcn_s_memrealtime();
            n_steps += steps;
            if (__builtin_expect(report_at != 0u && n_steps >= report_at, 0)) report(now);
            // (bounded: an `s_sleep 1` takes ~30 ns = 3 ticks, so a wait of one period ends within ticks / 3 turns; a clock that does
            // not advance must slow the launch down, not hang it)
            if ((int64_t)(now - due) < 0)
                for (uint32_t turn = 0; turn < 2u * ticks + 64u && (int64_t)(__builtin_amdgcn_s_memrealtime() - due) < 0; ++turn)
                    __builtin_amdgcn_s_sleep(1);
        } else if (slot) {  // a launch without the limiter that is part of the loop: no clock but for its one report
            n_steps += steps;
            if (__builtin_expect(report_at != 0u && n_steps >= report_at, 0)) report(__builtin_amdgcn_s_memrealtime());
        }
    }
    // the wave leaves (a launch too short, or too oddly aligned, to have reached report_at reports now)
    __device__ __forceinline__ void finish()
    {
        if (slot && report_at) report(__builtin_amdgcn_s_memrealtime());
    }
};

struct RolloutArgs {
    const uint8_t *cell;
    const uint8_t *greedy;  // first-argmax action per state (GU_POLICY_GREEDY)
    const uint4 *pi_thr;    // [S] inverse-CDF thresholds of the action probabilities (GU_POLICY_SAMPLE)
    int32_t S, pi_lds;      // pi_lds: the threshold table fits in LDS behind the two grid planes
    int32_t cell_bytes, W;
    uint64_t lut;
    int32_t *pos, *reward, *done;
    uint32_t *episode;
    const uint32_t *tcount;  // per-env offsets
    const int32_t *starts;
    const uint32_t *actions;  // GU_POLICY_STREAM: [ceil(T / 16)][N] words of 16 two-bit actions (gu_pack_actions_kernel)
    int32_t *tr_obs, *tr_reward, *tr_done;  // [T][N] each
    int32_t *ret, *episodes_fin;
    uint64_t *done_bits;  // [ceil(N/64)] wave ballots of the final done flags (episode-done compaction)
    uint32_t n_starts, seed_prefix, env_id0, steps_taken;  // seed_prefix: the epoch of this launch folded in; steps_taken: low word
    uint32_t seed_prefix0, steps_hi;  // the seed prefix without an epoch (start choices; straddling launches); high word of the lock-step count
    const uint32_t *nib;    // MAP 5: [waves][nib_dwords][64] four bits per cell of the padded grid, per env (gu_nibble_planes)
    int32_t nib_dwords;     // dwords per env of that image (a multiple of four)
    uint64_t lut_p;         // MAP 5: the action -> delta LUT of the padded image, times 32: -(W + 1), +1, +(W + 1), -1
    int32_t straddle;       // some env passes a multiple of 2^32 steps during this launch: the general kernel asks per lane and step
    int64_t N, T;
    GridSel gs;
    const uint32_t *rows;   // transition-row table [S][4] (gu_rollout_rows.hip)
    const uint32_t *rows2;  // ... or its pair table [S][16][2] followed by a one-step table [S][4] (two steps per LDS round trip)
    int32_t row_shift;      // log2(16 * copies) of that table
    int32_t stream_lds_off; // GU_POLICY_STREAM, MAP 1: byte offset in LDS of the staged action words [stream_lds_words][blockDim.x] ...
    int32_t stream_lds_words;  // ... and how many words per lane fit (0: every word is read from HBM when its steps are due)
    int32_t entry_table;    // transition-row kernel: the state at entry was left by a rollout -- the first step runs on the table too
    int32_t half_waves;     // transition-row kernel: lanes 0 .. 31 of every wave carry an env, twice the waves (gu_rollout_rows.hip)
    int32_t xcd_remap;      // workgroup b works on env block (b % 8) * (blocks / 8) + b / 8: one XCD = one contiguous env range
    GuPaceArgs pace;        // launches that write rows: the waves' schedule (GuPacer), the rate limiter of the store stream
};

// Workgroups are handed to the 8 XCDs round-robin (workgroup b -> XCD b % 8), so neighbouring env blocks would be
// written by different XCDs, through different L2s.  With the remap every XCD owns one contiguous eighth of the batch:
// its L2 then writes back 8x longer contiguous runs of every trajectory row.  (No reuse is at stake -- this is about the
// write stream's locality.)  Needs a block count divisible by 8.
__device__ __forceinline__ uint32_t gu_env_block(int32_t xcd_remap)
{
    const uint32_t b = blockIdx.x;
    return xcd_remap ? (b & 7u) * (gridDim.x >> 3) + (b >> 3) : b;
}

// GU_POLICY_STREAM without LDS staging: steps [i0, T) of the packed stream (gu_pack_actions_kernel: word [k][env] = the two-bit
// actions of steps 16 k .. 16 k + 15; the buffer carries GU_STREAM_PAD_WORDS spare rows, so the look-ahead never leaves it).
// Four words are in flight per lane, each re-loaded right after it has been consumed: a word arrives three words (48 steps)
// before its steps are due -- one word ahead is not enough for the stats-only launches (37 ns per step against ~1 us of HBM
// latency).  `step16(word)`: sixteen unrolled steps; `step1(act)`: one step.
// TAIL16: the full words among the last < 64 steps also go through step16 (one more inlined copy of it) instead of step1.
template <bool TAIL16 = false, class S16, class S1>
__device__ __forceinline__ void gu_stream_run(const char *pa, int64_t row, uint32_t e4, int64_t T, int64_t i0, S16 &&step16, S1 &&step1)
{
    const uint32_t row32 = (uint32_t)row;
    auto word_at = [&](int64_t k, uint32_t c) {
        return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(__builtin_amdgcn_make_buffer_rsrc((void *)(pa + k * row), 0, 0xFFFFFFFFu, 0x00020000), e4,
                                                              c * row32, 0);
    };
    int64_t i = i0, k = i0 >> 4;
    if (i & 15) {  // head: the rest of a word that an earlier step began
        const uint32_t word = word_at(k, 0);
        for (; i < T && (i & 15); ++i) step1((word >> (2u * (uint32_t)(i & 15))) & 3u);
        ++k;
    }
    uint32_t w[4];
#pragma unroll
    for (uint32_t c = 0; c < 4; ++c) w[c] = word_at(k, c);
    for (const int64_t full = T >> 4; k + 4 <= full; k += 4, i += 64) {
#pragma unroll
        for (uint32_t c = 0; c < 4; ++c) {
            const uint32_t word = w[c];
            asm volatile("" ::"v"(word));  // the wait for THIS word lands here, not behind the load that follows
            w[c] = word_at(k + 4, c);
            step16(word);
        }
    }
    uint32_t q = 0;  // fewer than 64 steps are left: words w[0 ..]
    if (TAIL16) {
        for (; i + 16 <= T; i += 16, q += 16) {
            const uint32_t sel = q >> 4;
            step16(sel == 0 ? w[0] : sel == 1 ? w[1] : w[2]);
        }
    }
    for (; i < T; ++i, ++q) {
        const uint32_t sel = q >> 4, word = sel == 0 ? w[0] : sel == 1 ? w[1] : sel == 2 ? w[2] : w[3];
        step1((word >> (2u * (q & 15u))) & 3u);
    }
}

// AUTO: 0 = no reset; 1 = auto-reset, single start cell (branch-free selects keyed on the TERM bit of the
//       register copy of flags); 2 = auto-reset, several start cells (RNG stream 1, rare divergent branch)
// MAP : 0 = records read from L2 (any grid size, any grid-per-env assignment)
//       1 = the block's grid staged in LDS, shared by its lanes
//       5 = every lane keeps its own grid in LDS as FOUR BITS PER CELL -- {reward +10, reward -10, terminal, wall}: the upper half
//           of the cell record, reward code first -- for multi-grid engines whose groups do not align with blocks, e.g. one maze per env.  The image
//           is PADDED with wall cells: one column on the left of every row, one row above and below (cell (x, y) at index
//           (y + 1)(W + 1) + x + 1), so that a move off the grid meets a wall like any other and the step tests nothing but the
//           CANDIDATE cell's wall bit (env:136-155 literally, env:51-54 folded into the padding).  (P + 7) / 8 dwords per env, P =
//           (H + 2)(W + 1) + 1: 576 bytes at 32 x 32 -- a wave's 36 KB, four waves per CU, the occupancy of the single-grid
//           launch --, dword j of lane l at word j * 64 + l: the 64 lanes of a gather hit 64 different banks whatever cells they
//           ask for.  One LDS round trip per step as in the other variants.  Staged from a per-wave image that gu_nibble_planes
//           builds once per grid installation (gu_kernels.hip).  Replaces round 2's private BYTE plane per lane (MAP 2: 66.5 KB
//           per wave, two waves per CU, 0.53 of the HBM peak).
//       3 = like 1 with the flags plane only (it carries the reward code): grids of 32 768 .. ~160 000 cells, one
//           block per CU.  A global read per step would wait for every trajectory store in flight (vmcnt counts both).
//       (a fifth variant -- one dword record per cell replicated 32 times, so that the per-step gather is free of LDS bank
//       conflicts -- was built and measured SLOWER than variant 1 at every batch size but one, 66.8 against 62.3 us for the
//       stats-only launch of config 3: the conflicts (SQ_LDS_BANK_CONFLICT ~ 7 cycles per gather) are not what bounds the
//       latency-bound modes, the length of the dependent chain is; profiles/archive/r02b_map_ab.txt.  What shortens the chain is the
//       transition-row table of gu_rollout_rows.hip.)
// Cache-policy bits of the trajectory stores (buffer_store aux: 1 = sc0, 2 = nt, 16 = sc1).  The int32 rows go out with sc1 + nt:
// written through at device scope instead of staying dirty in the L2 until they are evicted, and marked as streaming -- the rows
// are never read again by the launch, and the memory then sees them in the order the waves issue them, not in the L2's eviction
// order.  Measured with library variants, 24 buffers each, processes interleaved.  Under the first (idle-turn) limiter
// (profiles/archive/r03k_store_scope.txt): 119.4 us without, 116.4 with sc1, 115.3 with sc0 + sc1, 119.2 with nt alone, 116.1 with all
// three.  Under the schedule limiter (profiles/archive/r03p_store_scope_schedule.txt, two boxes): default policy 114.4, nt alone 110.5, sc1
// 110.7 / 109.0, sc0 + sc1 110.3, sc1 + nt 107.6 / 108.1, all three 108.2 -- on slow allocations sc1 111.2 .. 111.5 against 107.8 ..
// 108.2 with sc1 + nt, on fast ones no difference (107.1 / 107.7).  The packed row (4 B per env-step, transition-row kernel: 45 us
// per 65 536 x 1000 launch) gains from sc1 too: 1.35e12 -> 1.44e12 env-steps/s, three runs each (profiles/archive/r03m_packed_sc1.txt).
#ifndef GU_STORE_AUX
#define GU_STORE_AUX 18
#endif
#ifndef GU_STORE_AUX_PACKED
#define GU_STORE_AUX_PACKED 16
#endif
#define GU_MAX_BLOCK 1024
typedef uint32_t gu_v3u __attribute__((ext_vector_type(3)));
// TRAJ: 0 = no trajectory; 1 = int32 obs / reward / done rows (12 B per env-step), three planes [T][N]; 3 = the same int32 words as
//       ONE plane of triples [T][N][3] (one 12-byte store per lane and step instead of three 4-byte ones);
//       2 = ONE packed uint32 row: obs | (reward & 0xFF) << 16 | done << 24 (4 B per env-step, grids up to 65 536 cells)
template <int POLICY, int AUTO, int TRAJ, bool STATS, int MAP>
__global__ void __launch_bounds__(GU_MAX_BLOCK) gu_rollout_kernel(const RolloutArgs a)
{
    constexpr bool LDS = MAP == 1 || MAP == 3;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    // this lane's state is asked for BEFORE the grid is staged, so that the two round trips to memory overlap (at the start of a
    // launch the loads queue behind what is left of the previous launch's stores: ~1 us each)
    GuPacer pacer;
    pacer.fetch(a.pace, TRAJ == 1 || TRAJ == 3);
    const int64_t e64 = (int64_t)gu_env_block(a.xcd_remap) * blockDim.x + threadIdx.x;  // (remap only on single-grid engines)
    const bool live = e64 < a.N;
    const uint32_t e = (uint32_t)e64;
    int32_t s = 0, r = 0;
    uint32_t d = 0, ep = 0, tcount0 = 0;
    if (live) {
        s = a.pos[e];
        r = a.reward[e];
        d = (uint32_t)a.done[e];
        ep = a.episode[e];
        tcount0 = a.tcount[e];
    }
    CellMap m = gu_stage_map<LDS>(a.cell, a.cell_bytes, smem, a.gs, MAP == 3 ? 1 : 2);
    // MAP 5: every wave of the block stages the image of ITS 64 envs (one contiguous piece) into its own region of the block's LDS
    const uint32_t nib_region = MAP == 5 ? (threadIdx.x >> 6) * (uint32_t)a.nib_dwords * 256u : 0u;  // byte offset of that region
    if (MAP == 5) {
        const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
        const uint4 *src = reinterpret_cast<const uint4 *>(a.nib + wave * (size_t)a.nib_dwords * 64u);
        uint4 *dst = reinterpret_cast<uint4 *>(smem + nib_region);
        if (wave * 64u < (size_t)a.N)
            for (int32_t i = threadIdx.x & 63; i < a.nib_dwords * 16; i += 64) dst[i] = src[i];
        __syncthreads();
    }
    const uint8_t *greedy = a.greedy;
    if (MAP == 1 && POLICY == GU_POLICY_GREEDY) {
        uint8_t *dst = smem + 2 * a.cell_bytes;
        for (int32_t i = threadIdx.x * 16; i < a.cell_bytes; i += blockDim.x * 16)
            *reinterpret_cast<uint4 *>(dst + i) = *reinterpret_cast<const uint4 *>(a.greedy + i);
        __syncthreads();
        greedy = dst;
    }
    // the LDS copy keeps its own pointer (never merged with the global one): a pointer that may be either becomes a
    // FLAT load, whose wait also covers every trajectory store still in flight
    const bool thr_in_lds = MAP == 1 && POLICY == GU_POLICY_SAMPLE && a.pi_lds;
    uint4 *thr_lds = reinterpret_cast<uint4 *>(smem + 2 * a.cell_bytes);
    if (thr_in_lds) {
        for (int32_t i = threadIdx.x; i < a.S; i += blockDim.x) thr_lds[i] = a.pi_thr[i];
        __syncthreads();
    }
    if (!live) return;
    LaneGrid lg = gu_lane_grid<LDS>(a.gs, a.starts, a.n_starts, e, m);

    const uint32_t t_lane = tcount0 + a.steps_taken;
    const uint32_t prefix = gu_rng_prefix(a.seed_prefix, a.env_id0 + e);
    // start choices (stream 1) are keyed by the episode count alone: no epoch in their prefix
    const uint32_t prefix0 = AUTO == 2 ? gu_rng_prefix(a.seed_prefix0, a.env_id0 + e) : prefix;
    // a launch during which an env passes a multiple of 2^32 steps (once in 2^32 steps; the launcher sends it here, whatever its
    // shape): the words of streams 0 and 2 come from the prefix of the epoch of THEIR step, asked per lane and step
    const uint32_t epoch0 = (uint32_t)((uint64_t)((int64_t)(((uint64_t)a.steps_hi << 32) | a.steps_taken) + (int64_t)(int32_t)tcount0) >> 32);
    auto prefix_at = [&](uint32_t t) {  // (t: the low word of the step count; a launch is shorter than 2^32 steps)
        return gu_rng_prefix(gu_rng_seed_prefix_epoch(a.seed_prefix0, epoch0 + (t < t_lane ? 1u : 0u)), a.env_id0 + e);
    };
    // MAP 5 keeps in `flags` what the four bits of the agent's cell say, shifted up by three: RPLUS 0x08, RMINUS 0x10 -- so that the
    // register itself is the bit offset of the cell's reward in a constant (v_bfe_i32 reads the low five bits of its offset) --, TERM
    // 0x20, WALL 0x40.  A terminal cell absorbs every action (env:145-146): the move asks the TERM bit of the cell the agent is on, the
    // lazy reset asks the done flag `d` -- so the register's TERM bit is never overwritten at entry, as the other maps do.
    constexpr uint32_t TERM5 = 0x20u, TERM5_BIT = 5u;
    auto flags5 = [](uint32_t b) { return (((b >> 5) & 3u) << 3) | (((b >> GU_CELL_TERM_BIT) & 1u) << TERM5_BIT) | ((b >> 7) << 6); };
    uint32_t flags = MAP == 5 ? flags5(m.f[s]) : m.f[s];
    int32_t ret = 0, fin = 0;
    const int32_t W = a.W;
    const uint64_t lut = a.lut;
    const int32_t start0 = lg.starts[0];
    const uint32_t start0_flags = MAP == 5 ? flags5(m.f[start0]) : m.f[start0];
    // `sq`: the agent's cell in the padded image (cell (x, y) -> (y + 1)(W + 1) + x + 1), times 32 -- bits 8 and up are then the byte
    // offset of its dword's row in the [dword][lane] image, bits 5 .. 7 the cell inside the dword; a.lut_p is scaled alike -- plus the
    // byte offset of the wave's region in the block's LDS (a multiple of 256: it moves the row, nothing else)
    auto padded = [&](int32_t c) { return (int32_t)(((uint32_t)(c + c / W + W + 2) << 5) + nib_region); };  // (+ the wave's region: whole rows)
    int32_t sq = MAP == 5 ? padded(s) : 0;
    const int32_t start0_q = MAP == 5 ? padded(start0) : 0;
    // The image sits at the START of the workgroup's LDS (this kernel has no static LDS; the launcher checks it and takes another
    // map otherwise), in rows of 256 bytes: the address of a gather is row offset | lane * 4 -- ONE v_and_or_b32.
    typedef __attribute__((address_space(3))) const uint32_t *lds_word_ptr;
    const uint32_t lane4 = (threadIdx.x & 63u) << 2;
    auto word_at = [&](int32_t cq) { return *(lds_word_ptr)(uintptr_t)(((uint32_t)cq & 0xFFFFFF00u) | lane4); };
    // (the low five bits of a bit-field offset count: (cell & 7) * 4, cq being a multiple of 32)
    auto four_of = [](uint32_t word, int32_t cq) { return __builtin_amdgcn_ubfe(word, (uint32_t)cq >> 3, 4); };
    auto four_at = [&](int32_t cq) { return four_of(word_at(cq), cq); };
    auto cell5 = [&](int32_t cq) { return four_at(cq) << 3; };  // the record of padded cell cq / 32 from the four-bit image
    // one move on the four-bit image: the candidate cell, ONE gather, a wall?  (env:136-155; moves off the grid meet the padding)
    auto move5 = [&](uint32_t act, int32_t delta, auto &&in_the_shadow) {
        int32_t cand = s + delta;
        asm volatile("" : "+v"(cand));  // (kept as the sum it is: select(go, s + delta, s), not s + select(go, delta, 0) with its sign extension)
        const int32_t candq = sq + gu_delta<true>(act, a.lut_p, 0);
        const uint32_t word = word_at(candq);
        in_the_shadow();  // (what the caller has to issue that does not depend on the gather: the row stores of the step before)
        const uint32_t four = four_of(word, candq);
        const bool go = ((flags & TERM5) | four) < 8u;  // the agent's cell is not terminal and the candidate (bit 3: WALL) is no wall
        s = go ? cand : s;
        sq = go ? candq : sq;
        flags = go ? four << 3 : flags;
    };
    // Trajectory rows are addressed as buffer resource (wave-uniform base, rebuilt per 16-step chunk)
    // + lane byte offset e4 (VGPR) + scalar row offset (SGPR): buffer_store_dword ... offen, so that
    // advancing a row costs SALU only and no per-lane 64-bit address arithmetic.
    char *po = (char *)a.tr_obs, *pr = (char *)a.tr_reward, *pd = (char *)a.tr_done;
    const char *pa = (const char *)a.actions;
    const uint32_t e4 = e * 4u;
    const int64_t row = a.N * 4;
    const uint32_t row32 = (uint32_t)row;  // gu_create caps N at 2^25, so lane offset + 15 rows < 2^31 bytes
    // the trajectory's own row pitch and lane offset: TRAJ == 3 keeps ONE row of (obs, reward, done) triples per step
    // (the launcher takes that layout for batches of up to 2^24 envs only: lane offset + 15 rows of 12 N bytes < 2^32)
    const int64_t trow = TRAJ == 3 ? a.N * 12 : row;
    const uint32_t trow32 = (uint32_t)trow;
    const uint32_t te = TRAJ == 3 ? e * 12u : e4;
    __amdgpu_buffer_rsrc_t ro = __builtin_amdgcn_make_buffer_rsrc(po, 0, 0xFFFFFFFFu, 0x00020000);
    __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(pr, 0, 0xFFFFFFFFu, 0x00020000);
    __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(pd, 0, 0xFFFFFFFFu, 0x00020000);
    // the row of one step (`soff`: wave-uniform byte offset of the row from the resource base)
    auto emit = [&](int32_t s_, int32_t r_, uint32_t d_, uint32_t soff) {
        if (TRAJ == 1) {
            __builtin_amdgcn_raw_buffer_store_b32(s_, ro, e4, soff, GU_STORE_AUX);
            __builtin_amdgcn_raw_buffer_store_b32(r_, rr, e4, soff, GU_STORE_AUX);
            __builtin_amdgcn_raw_buffer_store_b32((int32_t)d_, rd, e4, soff, GU_STORE_AUX);
        } else if (TRAJ == 2) {
            __builtin_amdgcn_raw_buffer_store_b32((int32_t)((uint32_t)s_ | (((uint32_t)r_ & 0xFFu) << 16) | (d_ << 24)), ro, e4, soff, GU_STORE_AUX_PACKED);
        } else if (TRAJ == 3) {  // one 12-byte store per lane and step: the wave's 768 bytes are contiguous
            const gu_v3u triple = {(uint32_t)s_, (uint32_t)r_, d_};
            __builtin_amdgcn_raw_buffer_store_b96(triple, ro, te, soff, GU_STORE_AUX);
            GU_WIDE_STORE_PAD();
        }
    };
    // MAP 5 holds a step's row back until the NEXT step's gather has been issued: LDS and vector-memory instructions of a wave go
    // out through one port, in order, and a gather queued behind three row stores waits for them (~25 clocks each) on top of its own
    // latency -- on the dependent chain of a loop that has its SIMD alone.  Issued behind the gather, the stores fill its shadow.
    bool held = false;
    int32_t held_s = 0, held_r = 0;
    uint32_t held_d = 0, held_soff = 0;
    auto release = [&]() {
        if (MAP == 5 && TRAJ && held) emit(held_s, held_r, held_d, held_soff);
        held = false;
    };
    auto rebase = [&](int64_t rows) {
        release();  // (its row offset counts from the base that is about to move)
        po += rows * trow;
        ro = __builtin_amdgcn_make_buffer_rsrc(po, 0, 0xFFFFFFFFu, 0x00020000);
        if (TRAJ == 1) {
            pr += rows * trow;
            pd += rows * trow;
            rr = __builtin_amdgcn_make_buffer_rsrc(pr, 0, 0xFFFFFFFFu, 0x00020000);
            rd = __builtin_amdgcn_make_buffer_rsrc(pd, 0, 0xFFFFFFFFu, 0x00020000);
        }
    };

    // AUTO == 1 keeps the invariant "d == TERM bit of the REGISTER copy of flags", so the lazy reset needs no
    // separate test on the dependent chain; at entry the stored done flag may disagree with the cell (fresh reset
    // onto a terminal start, gu_set_state), so the register copy takes its TERM bit from the stored flag.
    if (AUTO == 1 && MAP != 5) flags = (flags & ~GU_CELL_TERM) | (d << GU_CELL_TERM_BIT);

    // `soff`: wave-uniform byte offset of this step's row from the resource base
    auto step = [&](uint32_t act, uint32_t soff) {
        const int32_t delta = gu_delta<MAP != 0>(act, lut, W);
        if (AUTO == 1) {
            // lazy `if done: env.reset()` (env:187-193) with a single start cell: two selects keyed directly on the
            // TERM bit of the record that just arrived (no separate done register on the dependent chain).  A variant
            // that precomputes the move from the start cell off the chain was measured slower at every occupancy
            // (profiles/archive/r01e_auto_form_ab.txt).
            // (MAP 5 asks the done bit it has extracted for the row anyway: one instruction less in a loop bound by their number)
            const bool was_done = MAP == 5 ? d != 0u : (flags & GU_CELL_TERM) != 0u;
            ep += was_done;
            s = was_done ? start0 : s;
            flags = was_done ? start0_flags : flags;
            if (MAP == 5) sq = was_done ? start0_q : sq;
        } else if (AUTO == 2) {
            if (d) {
                s = lg.starts[gu_rng_start_index(prefix0, ep, lg.n_starts)];
                ++ep;
                if (MAP == 5) sq = padded(s);
                flags = MAP == 5 ? cell5(sq) : m.f[s];  // (no global read inside the loop: it would wait for every row store in flight)
            }
        }
        if (MAP == 5) {
            move5(act, delta, [&]() {
                if (TRAJ && held) {
                    __builtin_amdgcn_sched_barrier(0);
                    emit(held_s, held_r, held_d, held_soff);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
        } else {
            s = gu_move(s, flags, act, delta);
            flags = m.f[s];
        }
        r = MAP == 5 ? __builtin_amdgcn_sbfe((int32_t)0xF6F60AFFu, flags, 8) : (MAP >= 2) ? gu_reward_packed(flags) : (int32_t)m.r[s];
        d = __builtin_amdgcn_ubfe(flags, MAP == 5 ? TERM5_BIT : GU_CELL_TERM_BIT, 1);
        if (STATS) {
            ret += r;
            fin += (int32_t)d;
        }
        if (MAP == 5 && TRAJ) held = true, held_s = s, held_r = r, held_d = d, held_soff = soff;
        else emit(s, r, d, soff);
    };
    auto step1 = [&](uint32_t act) {  // one step, then advance the resource base by one row
        step(act, 0);
        if (TRAJ) rebase(1);
    };

    pacer.start(a.pace, TRAJ == 1 || TRAJ == 3);
    if (POLICY == GU_POLICY_UNIFORM && a.straddle) {
        uint32_t t = t_lane;
#pragma nounroll
        for (int64_t i = 0; i < a.T; ++i, ++t) step1((gu_rng_word(prefix_at(t), GU_RNG_STREAM_ACTION, t >> 4) >> (2u * (t & 15u))) & 3u);
    } else if (POLICY == GU_POLICY_UNIFORM) {
        // Fast path: every lane of the wave is at the same step count (always true unless
        // gu_set_state installed per-env counters), so the 16-actions-per-word schedule is
        // wave-uniform: constant bit-field offsets, one hash per 16 steps.
        const uint32_t t_first = __builtin_amdgcn_readfirstlane(t_lane);
        if (__all(t_lane == t_first)) {
            uint32_t t = t_first;
            int64_t i = 0;
            if (t & 15u) {  // head: finish the current word
                const uint32_t word = gu_rng_word(prefix, GU_RNG_STREAM_ACTION, t >> 4);
                for (; i < a.T && (t & 7u); ++i, ++t) step1((word >> (2u * (t & 15u))) & 3u);
                if ((t & 15u) == 8u && i + 8 <= a.T) {  // the word's second half as one unrolled block (launches of 1000 steps begin here every other time)
#pragma unroll
                    for (uint32_t j = 0; j < 8; ++j) step(__builtin_amdgcn_ubfe(word, 16 + 2 * j, 2), j * trow32);
                    if (TRAJ) rebase(8);
                    i += 8, t += 8;
                }
                for (; i < a.T && (t & 15u); ++i, ++t) step1((word >> (2u * (t & 15u))) & 3u);
                pacer.after((uint32_t)i);
            }
            for (; i + 16 <= a.T; i += 16, t += 16) {  // body: 16 steps per word, fully unrolled
                uint32_t word = gu_rng_word(prefix, GU_RNG_STREAM_ACTION, t >> 4);
                // (MAP 5 is bound by the NUMBER of its instructions: the word is kept as it is -- the compiler would otherwise redo the
                // hash's last multiply in every step whose bits it can take from the product)
                if (MAP == 5) asm volatile("" : "+v"(word));
#pragma unroll
                for (uint32_t j = 0; j < 16; ++j) step(__builtin_amdgcn_ubfe(word, 2 * j, 2), j * trow32);
                if (TRAJ) rebase(16);
                if (i + 16 < a.T) pacer.after(16);  // (nothing to wait for behind the last group)
            }
            if (i < a.T) {  // tail
                const uint32_t word = gu_rng_word(prefix, GU_RNG_STREAM_ACTION, t >> 4);
                uint32_t j = 0;
                if (i + 8 <= a.T) {  // its first half as one unrolled block
#pragma unroll
                    for (uint32_t k = 0; k < 8; ++k) step(__builtin_amdgcn_ubfe(word, 2 * k, 2), k * trow32);
                    if (TRAJ) rebase(8);
                    i += 8, j = 8;
                }
                for (; i < a.T; ++i, ++j) step1((word >> (2u * j)) & 3u);
            }
        } else {
            uint32_t t = t_lane;
            uint32_t word = gu_rng_word(prefix, GU_RNG_STREAM_ACTION, t >> 4);
            for (int64_t i = 0; i < a.T; ++i) {
                step1((word >> (2u * (t & 15u))) & 3u);
                ++t;
                if ((t & 15u) == 0u) word = gu_rng_word(prefix, GU_RNG_STREAM_ACTION, t >> 4);
            }
        }
    } else if (POLICY == GU_POLICY_STREAM) {
        // The uploaded stream arrives packed like the uniform policy's RNG words (16 two-bit actions per env and word,
        // gu_pack_actions_kernel), always from row 0.
        const char *pw = pa;  // the word's row base moves (64-bit), the lane offset stays e4
        auto load_word = [&](uint32_t soff) {
            return (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(__builtin_amdgcn_make_buffer_rsrc((void *)pw, 0, 0xFFFFFFFFu, 0x00020000), e4, soff, 0);
        };
        const int64_t KW = (MAP == 1) ? a.stream_lds_words : 0;
        if (KW) {
            // A global load in a loop that also streams trajectory stores waits for ALL of them (one counter, out-of-order
            // completion between the two kinds: s_waitcnt vmcnt(0)), i.e. drains the store pipeline.  So the words are
            // fetched KW at a time (up to 1024 steps) into this lane's LDS column, and the loop reads them from there
            // (lgkmcnt).  A lane reads only what it wrote itself: no barrier.
            uint32_t *sw = reinterpret_cast<uint32_t *>(smem + a.stream_lds_off) + threadIdx.x;
            const uint32_t bd = blockDim.x;
            int64_t words_left = (a.T + 15) >> 4, i = 0;
            while (i < a.T) {
                const int64_t cnt = words_left < KW ? words_left : KW;
                int64_t k = 0;
                for (; k + 8 <= cnt; k += 8) {
                    uint32_t w[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) w[j] = load_word(j * row32);
#pragma unroll
                    for (int j = 0; j < 8; ++j) sw[(k + j) * bd] = w[j];
                    pw += 8 * row;
                }
                for (; k < cnt; ++k, pw += row) sw[k * bd] = load_word(0);
                words_left -= cnt;
                const int64_t steps = (a.T - i) < cnt * 16 ? (a.T - i) : cnt * 16;
                uint32_t word = sw[0];
                for (k = 0; k * 16 + 16 <= steps; ++k) {
                    const uint32_t next = k + 1 < cnt ? sw[(k + 1) * bd] : 0u;  // one word ahead of its steps
#pragma unroll
                    for (uint32_t j = 0; j < 16; ++j) step(__builtin_amdgcn_ubfe(word, 2 * j, 2), j * trow32);
                    if (TRAJ) rebase(16);
                    pacer.after(16);
                    word = next;
                }
                for (int64_t q = k * 16, j = 0; q < steps; ++q, ++j) step1((word >> (2u * (uint32_t)j)) & 3u);  // tail of the stream
                i += steps;
            }
        } else {
            gu_stream_run(
                pa, row, e4, a.T, 0,
                [&](uint32_t word) {
#pragma unroll
                    for (uint32_t j = 0; j < 16; ++j) step(__builtin_amdgcn_ubfe(word, 2 * j, 2), j * trow32);
                    if (TRAJ) rebase(16);
                    pacer.after(16);
                },
                step1);
        }
    } else {
        // Table policies: greedy[] / the sampling thresholds are read at the post-reset position, so the lazy reset
        // is explicit here.  8 steps per resource rebase (scalar row offsets, as on the uniform path); the sampling
        // word of the NEXT step is hashed while this step's threshold read is in flight (it does not depend on s).
        uint32_t t = t_lane;
        uint32_t word = POLICY == GU_POLICY_SAMPLE ? gu_rng_sample_word(prefix, t) : 0u;
        auto run = [&](auto thr_at) {
            // fresh: 0 = step t + 1 starts no sampling word, 1 = it does (both: t the same in every lane, known at compile time),
            // 2 = ask the lanes (gu_rng.hpp)
            auto tstep = [&](uint32_t soff, auto fresh_tag) {
                constexpr int FRESH = decltype(fresh_tag)::value;
                if (AUTO == 1) {
                    const bool was_done = flags & GU_CELL_TERM;
                    s = was_done ? start0 : s;
                    ep += was_done;
                    flags = was_done ? (start0_flags & ~GU_CELL_TERM) : flags;
                    d = 0;
                } else if (AUTO == 2) {
                    if (d) {
                        s = lg.starts[gu_rng_start_index(prefix0, ep, lg.n_starts)];
                        ++ep;
                        flags = m.f[s];
                        d = 0;
                    }
                }
                uint32_t act;
                if (POLICY == GU_POLICY_GREEDY) {
                    act = greedy[s];
                } else {
                    // inverse CDF of pi[s] on one uniform 32-bit word (RNG stream 2, counter = step count), as integer
                    // thresholds (gu_pi_threshold_kernel)
                    const uint4 q = thr_at(s);
                    const uint32_t next_word = FRESH == 2 ? gu_rng_sample_advance(prefix, t, word)
                                               : FRESH == 1 ? gu_rng_sample_advance_at<true>(prefix, t, word) : gu_rng_sample_advance_at<false>(prefix, t, word);
                    act = gu_sample_action(word, q);
                    word = next_word;
                }
                ++t;
                step(act, soff);
            };
            const std::integral_constant<int, 0> same_word{};
            const std::integral_constant<int, 1> new_word{};
            const std::integral_constant<int, 2> ask{};
            int64_t i = 0;
            if (POLICY == GU_POLICY_SAMPLE && a.straddle) {  // (see prefix_at: every step's word from the prefix of its own epoch)
#pragma nounroll
                for (; i < a.T; ++i) {
                    word = gu_rng_sample_word(prefix_at(t), t);
                    tstep(0, ask);
                    if (TRAJ) rebase(1);
                }
            }
            const uint32_t t_first = __builtin_amdgcn_readfirstlane(t);
            if (POLICY == GU_POLICY_SAMPLE && __all(t == t_first)) {
                // every lane at the same step count: single steps up to a multiple of four, then groups of eight in which the
                // fourth and the eighth step start a word -- no ballot, no branch (a taken branch costs ~60 clocks at one wave per SIMD)
                t = t_first;
                for (; i < a.T && (t & GU_RNG_SAMPLE_MASK); ++i) {
                    tstep(0, ask);
                    if (TRAJ) rebase(1);
                }
                if (i) pacer.after((uint32_t)i);
                constexpr uint32_t G = GU_RNG_SAMPLE_MASK + 1u < 8u ? 8u : GU_RNG_SAMPLE_MASK + 1u;  // steps per unrolled group
                for (; i + G <= a.T; i += G) {
#pragma unroll
                    for (uint32_t j = 0; j < G; ++j) {
                        if ((j & GU_RNG_SAMPLE_MASK) == GU_RNG_SAMPLE_MASK) tstep(j * trow32, new_word);
                        else tstep(j * trow32, same_word);
                    }
                    if (TRAJ) rebase(G);
                    if (i + G < a.T) pacer.after(G);
                }
            } else {
                for (; i + 8 <= a.T; i += 8) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) tstep(j * trow32, ask);
                    if (TRAJ) rebase(8);
                    if (i + 8 < a.T) pacer.after(8);
                }
            }
            for (; i < a.T; ++i) {
                tstep(0, ask);
                if (TRAJ) rebase(1);
            }
        };
        if (thr_in_lds) run([thr_lds](int32_t at) { return thr_lds[at]; });
        else run([&a](int32_t at) { return a.pi_thr[at]; });
    }
    release();
    pacer.finish();
    a.pos[e] = s;
    a.reward[e] = r;
    a.done[e] = (int32_t)d;
    a.episode[e] = ep;
    if (STATS) {
        a.ret[e] = ret;
        a.episodes_fin[e] = fin;
    }
    // episode-done compaction, first half: the wave's done flags as one 64-bit word (lanes past N have left; every
    // workgroup size used is a multiple of 64, so lane 0 of a wave holds its lowest env)
    const uint64_t bits = __ballot(d != 0);
    if ((threadIdx.x & 63) == 0) a.done_bits[e >> 6] = bits;
}

// ------------------------------------------------------------------------------------
// launch helpers
// ------------------------------------------------------------------------------------
static inline unsigned gu_blocks(int64_t n, int block) { return (unsigned)((n + block - 1) / block); }

// Largest block size <= preferred for which every block uses one grid (0 = none: use the L2 variant)
static inline int gu_lds_block(const gu_engine *h, int preferred, int planes)
{
    if (h->S > GU_MAX_LDS_CELLS || (size_t)planes * h->cell_bytes > 65536) return 0;
    if (h->n_grids == 1) return preferred;
    for (int bs = preferred; bs >= 64; bs >>= 1)
        if (h->group % bs == 0) return bs;
    return 0;
}

static inline int gu_rollout_block(const gu_engine *h) { return (int)gu_opt(h, GU_OPT_ROLLOUT_BLOCK); }

// MAP 5: dwords per env of the padded four-bits-per-cell image (eight cells per dword, rounded up to whole uint4 per lane), bytes per wave
static inline int32_t gu_nibble_cells(const gu_engine *h) { return (h->H + 2) * (h->W + 1) + 1; }
static inline int32_t gu_nibble_dwords(const gu_engine *h) { return (((gu_nibble_cells(h) + 7) / 8) + 3) & ~3; }
static inline size_t gu_nibble_bytes_per_wave(const gu_engine *h) { return (size_t)gu_nibble_dwords(h) * 256u; }
int gu_nibble_planes(gu_engine *h);  // builds h->d_nib if the installed grids have none yet (gu_kernels.hip)

template <int POLICY, int AUTO, int TRAJ, bool STATS>
static void gu_rollout_launch(gu_engine *h, const RolloutArgs &a_in, int bs)
{
    RolloutArgs a = a_in;
    auto blocks_ok = [&](int block) { return gu_blocks(h->N, block) % 8 == 0; };
    auto n_blocks = [&](int block) { return gu_blocks(h->N, block); };
    const int planes = POLICY == GU_POLICY_GREEDY ? 3 : 2;
    int lds_bs = gu_lds_block(h, bs, planes);
    // Groups of 64 .. 192 envs (a multiple of 64, smaller than the workgroup): every wave stages its own grid's planes, so the launch
    // keeps the workgroup size of the shared-grid launch -- whose store stream the memory takes at a shorter period than that of
    // one-wave workgroups (profiles/r06m_multigrid_ab.txt).  Uniform and stream policies (the table policies keep one grid anyway).
    bool per_wave = false;
    if ((POLICY == GU_POLICY_UNIFORM || POLICY == GU_POLICY_STREAM) && lds_bs && lds_bs < bs && h->n_grids > 1 && h->group % 64 == 0 &&
        (size_t)(bs / 64) * planes * h->cell_bytes <= 32768)
        per_wave = true, lds_bs = bs;
    if (lds_bs) {
        size_t lds = (size_t)planes * h->cell_bytes * (per_wave ? (size_t)(bs / 64) : 1);
        RolloutArgs b = a;
        b.gs.per_wave = per_wave ? 1 : 0;
        b.xcd_remap = a.xcd_remap && blocks_ok(lds_bs);
        if (POLICY == GU_POLICY_SAMPLE && lds + (size_t)h->S * sizeof(uint4) <= 65536) {
            b.pi_lds = 1;
            lds += (size_t)h->S * sizeof(uint4);
        }
        auto kern = gu_rollout_kernel<POLICY, AUTO, TRAJ, STATS, 1>;
        if (POLICY == GU_POLICY_STREAM && TRAJ != 0) {
            // staged action words: as many per lane as the LDS share of a block admits at the occupancy this batch needs
            const int64_t per_cu = std::min<int64_t>(8, std::max<int64_t>(1, ((int64_t)n_blocks(lds_bs) + h->n_cu - 1) / h->n_cu));
            const int64_t room = h->lds_per_cu / per_cu - (int64_t)lds - 512;
            int64_t kw = std::min<int64_t>({room / ((int64_t)lds_bs * 4), (int64_t)64, (a.T + 15) / 16});
            if (kw >= 4) {
                b.stream_lds_off = (int32_t)lds;
                b.stream_lds_words = (int32_t)kw;
                lds += (size_t)kw * lds_bs * 4;
            }
        }
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_per_cu);
        hipLaunchKernelGGL(kern, dim3(n_blocks(lds_bs)), dim3(lds_bs), lds, h->stream, b);
        return;
    }
    if constexpr (POLICY == GU_POLICY_UNIFORM || POLICY == GU_POLICY_STREAM) {
        // one grid too big for two planes in 64 KiB: its flags plane alone, up to the whole 160 KB of a CU
        if (h->n_grids == 1 && h->W <= 32767 && (int64_t)h->cell_bytes <= h->lds_per_cu - 512) {
            a.xcd_remap = a.xcd_remap && blocks_ok(bs);
            auto kern = gu_rollout_kernel<POLICY, AUTO, TRAJ, STATS, 3>;
            if (h->cell_bytes > 64 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->cell_bytes);
            hipLaunchKernelGGL(kern, dim3(n_blocks(bs)), dim3(bs), (size_t)h->cell_bytes, h->stream, a);
            return;
        }
        // misaligned multi-grid engine (e.g. one maze per env): every lane's grid at four bits per cell in LDS, if a wave's 64 fit
        if (h->n_grids > 1 && h->W <= 1022 && gu_nibble_bytes_per_wave(h) <= (size_t)h->lds_per_cu - 512 && a.nib) {
            // workgroups of four waves where four images fit a CU's LDS (32 x 32: 144 KB): the launch shape of the shared-grid kernel,
            // whose store stream the memory takes at a shorter period than that of 1024 one-wave workgroups (profiles/r06m_multigrid_ab.txt)
            int mbs = 256;
            while (mbs > 64 && (size_t)(mbs / 64) * gu_nibble_bytes_per_wave(h) > (size_t)h->lds_per_cu - 512) mbs >>= 1;
            const size_t lds = (size_t)(mbs / 64) * gu_nibble_bytes_per_wave(h);
            auto kern = gu_rollout_kernel<POLICY, AUTO, TRAJ, STATS, 5>;
            static std::atomic<int> base_zero{0};  // 1: no static LDS, the image starts at LDS address 0 (the kernel relies on it)
            if (!base_zero.load(std::memory_order_relaxed)) {
                hipFuncAttributes fa{};
                const bool got = hipFuncGetAttributes(&fa, (const void *)kern) == hipSuccess;
                base_zero.store(got && fa.sharedSizeBytes == 0 ? 1 : 2, std::memory_order_relaxed);
            }
            if (base_zero.load(std::memory_order_relaxed) == 1) {
                if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL(kern, dim3(gu_blocks(h->N, mbs)), dim3(mbs), lds, h->stream, a);
                return;
            }
        }
    }
    a.xcd_remap = a.xcd_remap && h->n_grids == 1 && blocks_ok(bs);
    hipLaunchKernelGGL((gu_rollout_kernel<POLICY, AUTO, TRAJ, STATS, 0>), dim3(n_blocks(bs)), dim3(bs), 0, h->stream, a);
}

template <int POLICY, int AUTO>
static void gu_rollout_dispatch2(gu_engine *h, const RolloutArgs &a, int traj, bool stats, int bs)
{
    if (traj == 1) {
        if (stats) gu_rollout_launch<POLICY, AUTO, 1, true>(h, a, bs);
        else gu_rollout_launch<POLICY, AUTO, 1, false>(h, a, bs);
    } else if (traj == 2) {
        if (stats) gu_rollout_launch<POLICY, AUTO, 2, true>(h, a, bs);
        else gu_rollout_launch<POLICY, AUTO, 2, false>(h, a, bs);
    } else if (traj == 3) {
        if (stats) gu_rollout_launch<POLICY, AUTO, 3, true>(h, a, bs);
        else gu_rollout_launch<POLICY, AUTO, 3, false>(h, a, bs);
    } else {
        if (stats) gu_rollout_launch<POLICY, AUTO, 0, true>(h, a, bs);
        else gu_rollout_launch<POLICY, AUTO, 0, false>(h, a, bs);
    }
}

template <int POLICY>
static void gu_rollout_dispatch(gu_engine *h, const RolloutArgs &a, int auto_mode, int traj, bool stats, int bs)
{
    switch (auto_mode) {
    case 0: gu_rollout_dispatch2<POLICY, 0>(h, a, traj, stats, bs); break;
    case 1: gu_rollout_dispatch2<POLICY, 1>(h, a, traj, stats, bs); break;
    default: gu_rollout_dispatch2<POLICY, 2>(h, a, traj, stats, bs); break;
    }
}

// one per policy kind, each in its own translation unit
void gu_rollout_uniform(gu_engine *h, const RolloutArgs &a, int auto_mode, int traj, bool stats, int bs);
void gu_rollout_stream(gu_engine *h, const RolloutArgs &a, int auto_mode, int traj, bool stats, int bs);
void gu_rollout_greedy(gu_engine *h, const RolloutArgs &a, int auto_mode, int traj, bool stats, int bs);
void gu_rollout_sample(gu_engine *h, const RolloutArgs &a, int auto_mode, int traj, bool stats, int bs);
// the transition-row kernel (gu_rollout_rows.hip): true when it took the launch (*rc: what its pace calibration returned)
bool gu_rollout_rows(gu_engine *h, RolloutArgs a, int32_t policy, int auto_mode, int traj, bool stats, int *rc);
bool gu_rows_pairs_fit(const gu_engine *h);  // its pair tables fit this engine's grid (and GU_OPT_ROLLOUT_ROWS does not forbid them)
// store pacing (gu_kernels.hip): the schedule of a launch that writes rows -- the launch kind's ring of records, or a fixed period
int gu_pace_for(gu_engine *h, int slot, int64_t T, unsigned blocks, int block_size, int row_bytes, GuPaceArgs *pace);
// the K-step kernel (gu_rollout_multi.hip; uniform policy, no trajectory): true when it took the launch
bool gu_rollout_multi(gu_engine *h, RolloutArgs a, int32_t policy, int auto_mode, int traj, bool stats);
