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
    'packed_rows': lambda name, grid: 'gu_rollout_rows_kernel<' in name and grid == 65536,
    'stats_only': lambda name, grid: 'gu_rollout_multi_kernel<' in name and grid == 65536,
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
    per_mode = defaultdict(lambda: defaultdict(list))
    names = {}
    for d in glob.glob(os.path.join(prof, 'pmc_write')) + glob.glob(os.path.join(prof, 'pmc_fetch')):
        for r in rows(os.path.join(d, '**', '*counter_collection.csv')):
            mode = mode_of(r['Kernel_Name'], r['Grid_Size'])
            if mode and r['Counter_Name'] in ('WRITE_SIZE', 'FETCH_SIZE'):
                per_mode[mode][r['Counter_Name']].append(float(r['Counter_Value']))
                names[mode] = r['Kernel_Name']
    durations = defaultdict(list)
    for r in kt:
        mode = mode_of(r['Kernel_Name'], r['Grid_Size_X'])
        if mode:
            durations[mode].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
            names.setdefault(mode, r['Kernel_Name'])
    modes = {}
    for mode, counters in per_mode.items():
        if 'WRITE_SIZE' not in counters:
            continue
        wr = sum(counters['WRITE_SIZE']) / len(counters['WRITE_SIZE']) * 1024.0
        fetch = counters.get('FETCH_SIZE', [])
        rd = 2.0 * (sum(fetch) / len(fetch) if fetch else 0.0) * 1024.0
        entry = dict(write_bytes=wr, read_bytes_corrected=rd, hbm_bytes_per_launch=wr + rd, kernel=names[mode][:120],
                     dispatches_counted=len(counters['WRITE_SIZE']))
        if durations.get(mode):
            entry['kernel_avg_us'] = sum(durations[mode]) / len(durations[mode]) / 1e3
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
        f.write('%-90s %6s %12s %12s %12s %12s\n' % ('bench launch form (kernel x launch size)', 'calls', 'avg_us', 'min_us', 'median_us', 'max_us'))
        for mode in MODES:
            v = sorted(durations.get(mode, []))
            if v:
                f.write('%-90s %6d %12.2f %12.2f %12.2f %12.2f\n' % ('%s: %s' % (mode, names.get(mode, ''))[:90], len(v), sum(v) / len(v) / 1e3,
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
