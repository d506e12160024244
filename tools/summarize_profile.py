#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (tools/gpu_profile.sh) into small text/JSON summaries that are
committed under profiles/.  Usage: summarize_profile.py <prof_dir> <tag>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


# the dominant kernel of bench.py (the general rollout kernel; NOT the row-table / K-step kernels of its `other_modes` launches)
BENCH_KERNEL = 'gu_rollout_kernel<'

# every launch form the bench line reports, recognised by kernel name and launch size (threads): the headline (config 3,
# 65 536 envs), the strong-scaling config-4 line (262 144 envs on one GPU), packed rows (row-table kernel), statistics only
# (K-step kernel)
MODES = {
    'headline': lambda name, grid: 'gu_rollout_kernel<' in name and grid == 65536,
    'strong_c4': lambda name, grid: 'gu_rollout_kernel<' in name and grid == 262144,
    'packed_rows': lambda name, grid: 'gu_rollout_rows_kernel<0, 2,' in name and grid == 65536,   # <uniform, packed rows, ...>
    'stats_only': lambda name, grid: 'gu_rollout_multi_kernel<' in name and grid == 65536,
    'rollout_sample_policy_traj': lambda name, grid: 'gu_rollout_rows_kernel<3, 1,' in name and grid == 65536,  # <sampled, int32 rows, ...>
    'rollout_sample_policy_stats_only': lambda name, grid: 'gu_rollout_rows_kernel<3, 0,' in name and grid == 65536,
    'c5_rounds_in_one_launch': lambda name, grid: 'gu_vi_xcd_kernel<' in name or 'gu_vi_sweep_step_xcd_kernel<' in name,
    'c3_distinct_65536': lambda name, grid: 'gu_rollout_kernel<' in name and name.rstrip().endswith(', 5>(RolloutArgs)') and grid == 65536,
}


def mode_of(name, grid):
    for mode, match in MODES.items():
        if match(name, int(grid)):
            return mode
    return None

def rows(pattern):
    out = []
    for p in glob.glob(pattern, recursive=True):
        with open(p, newline='') as f:
            out.extend(csv.DictReader(f))
    return out


def main():
    prof, tag = sys.argv[1], sys.argv[2]
    dest = os.path.join(prof, 'summary')
    os.makedirs(dest, exist_ok=True)
    summary = {'tag': tag}
    # kernel trace: per-kernel duration statistics
    kt = rows(os.path.join(prof, 'kt', '**', '*kernel_trace.csv'))
    per = defaultdict(list)
    for r in kt:
        per[r['Kernel_Name']].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    stats = {}
    for k, v in per.items():
        v.sort()
        stats[k] = dict(calls=len(v), total_ns=sum(v), avg_ns=sum(v) / len(v), min_ns=v[0], median_ns=v[len(v) // 2], max_ns=v[-1])
    summary['kernel_trace'] = stats
    if kt:
        r = next(x for x in kt if BENCH_KERNEL in x['Kernel_Name'])
        summary['rollout_dispatch'] = {k: r.get(k) for k in ('Workgroup_Size_X', 'Grid_Size_X', 'VGPR_Count', 'Accum_VGPR_Count',
                                                             'SGPR_Count', 'LDS_Block_Size', 'Scratch_Size') if k in r}
    # counters: average per rollout dispatch (skip the first, which includes cold caches)
    pmc = defaultdict(list)
    for d in glob.glob(os.path.join(prof, 'pmc_*')):
        for r in rows(os.path.join(d, '**', '*counter_collection.csv')):
            if BENCH_KERNEL in r['Kernel_Name']:
                pmc[r['Counter_Name']].append(float(r['Counter_Value']))
    summary['rollout_pmc_avg_per_dispatch'] = {k: sum(v) / len(v) for k, v in sorted(pmc.items())}
    summary['rollout_pmc_samples'] = {k: len(v) for k, v in sorted(pmc.items())}
    avg = summary['rollout_pmc_avg_per_dispatch']
    # ---- per bench mode: HBM traffic from the WRITE_SIZE / FETCH_SIZE passes, kernel duration from the kernel-trace pass
    # MI355X_MICROARCH.md "HBM": WRITE_SIZE and FETCH_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a coalesced
    # read stream, so it is doubled; WRITE_SIZE reads the bytes exactly.  Calibration in our own access pattern: the
    # headline kernel's known write volume is 12 B x N x T + state write-back.
    # bench.py's order of work tells the launch forms apart where kernel name and launch size do not: the headline launches come
    # first, then `other_modes` (K-step and row-table kernels), then strong_c4 -- whose pacing calibration also tries the batch as
    # several launches of one wave per SIMD in a row (65 536 lanes each, like the headline's).  Per form the LAST dispatches count
    # (same launch size as the very last one): calibration launches, over-idled on purpose, and strong_c4's shorter first launch
    # come before them.  (Until r03o a strong_c4 batch could run as several launches in a row: its part x the number of parts.)
    C4_ENVS = 262144
    TAIL = 8

    def classify(records, tail=TAIL):
        """records: [(dispatch id, kernel name, grid size, value)] -> {mode: [values per batch launch, steady launches only]}"""
        records = sorted(records)
        marker = next((i for i, n, g, v in records if 'gu_rollout_multi_kernel<' in n or 'gu_rollout_rows_kernel<' in n), None)
        seq = defaultdict(list)
        for i, n, g, v in records:
            if 'gu_rollout_multi_kernel<' in n:
                names['stats_only'] = n
                if g == 65536:
                    seq['stats_only'].append((g, v))
            elif 'gu_rollout_rows_kernel<' in n:
                # <policy, rows, ...>: uniform with packed rows / sampled with int32 rows / sampled, statistics only (other_modes);
                # the int32-row launches of configs 2 and 4-shard run on this kernel at other batch sizes
                form = ('packed_rows' if '<0, 2,' in n else 'rollout_sample_policy_traj' if '<3, 1,' in n
                        else 'rollout_sample_policy_stats_only' if '<3, 0,' in n else None)
                if form and g == 65536:
                    names[form] = n
                    seq[form].append((g, v))
            elif 'gu_vi_xcd_kernel<' in n or 'gu_vi_sweep_step_xcd_kernel<' in n:
                names['c5_rounds_in_one_launch'] = n
                seq['c5_rounds_in_one_launch'].append((g, v))
            elif 'gu_rollout_kernel<' in n:
                if n.rstrip().endswith(', 5>(RolloutArgs)'):  # MAP 5: one grid per env (bench.py: configs.c3_distinct_65536)
                    names['c3_distinct_65536'] = n
                    if g == 65536:
                        seq['c3_distinct_65536'].append((g, v))
                    continue
                names.setdefault('general', n)
                if marker is None or i < marker:
                    if g == 65536:
                        seq['headline'].append((g, v))
                elif g == C4_ENVS:  # (the distinct-grid configs also run 65 536 lanes on this kernel, after the marker)
                    seq['strong_c4'].append((g, v))
        out = {}
        for mode, items in seq.items():
            last_grid = items[-1][0]
            steady = []
            for g, v in reversed(items):
                if g != last_grid or len(steady) == tail:
                    break
                steady.append(v)
            factor = C4_ENVS // last_grid if mode == 'strong_c4' else 1
            if mode == 'strong_c4':
                names['strong_c4_parts'] = factor
            out[mode] = [v * factor for v in reversed(steady)]
        return out

    def median(v):
        v = sorted(v)
        return v[len(v) // 2]

    names = {}
    per_mode = defaultdict(dict)
    for counter, d in (('WRITE_SIZE', 'pmc_write'), ('FETCH_SIZE', 'pmc_fetch')):
        recs = [(int(r['Dispatch_Id']), r['Kernel_Name'], int(r['Grid_Size']), float(r['Counter_Value']))
                for r in rows(os.path.join(prof, d, '**', '*counter_collection.csv')) if r['Counter_Name'] == counter]
        for mode, values in classify(recs).items():
            per_mode[mode][counter] = values
    dur = classify([(int(r['Dispatch_Id']), r['Kernel_Name'], int(r['Grid_Size_X']), float(int(r['End_Timestamp']) - int(r['Start_Timestamp']))) for r in kt], tail=400)
    durations = {mode: [int(x) for x in v] for mode, v in dur.items()}
    for mode in ('headline', 'strong_c4'):
        names[mode] = names.get('general', '')
    modes = {}
    for mode, counters in per_mode.items():
        if 'WRITE_SIZE' not in counters:
            continue
        wr = median(counters['WRITE_SIZE']) * 1024.0
        fetch = counters.get('FETCH_SIZE', [])
        rd = 2.0 * (median(fetch) if fetch else 0.0) * 1024.0
        entry = dict(write_bytes=wr, read_bytes_corrected=rd, hbm_bytes_per_launch=wr + rd, kernel=names.get(mode, '')[:120],
                     dispatches_counted=len(counters['WRITE_SIZE']), statistic='median of the last %d dispatches of the form' % TAIL)
        if mode == 'strong_c4' and names.get('strong_c4_parts', 1) > 1:
            entry['launches_in_a_row'] = names['strong_c4_parts']
        if durations.get(mode):
            entry['kernel_avg_us'] = sum(durations[mode]) / len(durations[mode]) / 1e3
            entry['kernel_median_us'] = median(durations[mode]) / 1e3
            entry['kernel_dispatches_timed'] = len(durations[mode])
        modes[mode] = entry
    if modes:
        import datetime
        source = ('rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE (separate passes over bench.py, every launch form of the line), tag %s; '
                  'bytes = WRITE_SIZE*1024 + 2*FETCH_SIZE*1024; kernel_avg_us from the --kernel-trace pass of the same tag' % tag)
        table = dict(tag=tag, date=datetime.date.today().isoformat(), source=source, modes=modes)
        if 'headline' in modes:  # (the flat form older readers of the file expect)
            table.update({k: modes['headline'][k] for k in ('write_bytes', 'read_bytes_corrected', 'hbm_bytes_per_launch')})
        for entry in modes.values():
            entry['source'] = source
        summary['hbm'] = table
        json.dump(table, open(os.path.join(dest, 'rollout_pmc_latest.json'), 'w'), indent=1)
    for p in glob.glob(os.path.join(prof, 'kt', '**', '*kernel_stats.csv'), recursive=True):
        open(os.path.join(dest, '%s_rocprofv3_kernel_stats.csv' % tag), 'w').write(open(p).read())
    json.dump(summary, open(os.path.join(dest, '%s_rocprof_summary.json' % tag), 'w'), indent=1)
    with open(os.path.join(dest, '%s_kernel_stats.txt' % tag), 'w') as f:
        # the rollout kernel serves two launch sizes of the line (65 536 envs: the headline; 262 144: strong_c4): split first
        f.write('%-90s %6s %12s %12s %12s %12s\n' % ('bench launch form (steady launches: the last of each form)', 'calls', 'avg_us', 'min_us', 'median_us', 'max_us'))
        for mode in MODES:
            v = sorted(durations.get(mode, []))
            if v:
                f.write('%-90s %6d %12.2f %12.2f %12.2f %12.2f\n' % ('%s: %s' % (mode, str(names.get(mode, '')))[:90], len(v), sum(v) / len(v) / 1e3,
                                                                   v[0] / 1e3, v[len(v) // 2] / 1e3, v[-1] / 1e3))
        f.write('\n')
        f.write('%-90s %6s %12s %12s %12s %12s\n' % ('kernel', 'calls', 'avg_us', 'min_us', 'median_us', 'max_us'))
        for k, s in sorted(stats.items(), key=lambda kv: -kv[1]['total_ns']):
            f.write('%-90s %6d %12.2f %12.2f %12.2f %12.2f\n' % (k[:90], s['calls'], s['avg_ns'] / 1e3, s['min_ns'] / 1e3,
                                                               s['median_ns'] / 1e3, s['max_ns'] / 1e3))
    print(open(os.path.join(dest, '%s_kernel_stats.txt' % tag)).read())
    print(json.dumps(summary['rollout_pmc_avg_per_dispatch'], indent=1))


if __name__ == '__main__':
    main()
