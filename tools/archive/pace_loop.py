#!/usr/bin/env python3
"""The closed-loop store pacing (gu_rollout.hpp: GuPacer / gu_pace_next) looked at from outside, on one engine shape:
  sweep  : fixed periods, each entered from a healthy stream and from a collapsed one -- device-clock time per launch and the
           share of wave-groups begun behind schedule (what the controller steers by), so the thresholds can be read off;
  loop   : a FRESH engine, launches back to back from the first one on: period, late share and device time of every launch
           (read from the kind's ring every 60 launches), event-timed wall per chunk;
  search : what the open-loop search of rounds 3 and 4 finds on the same buffer, and launches held at that period.
    python tools/pace_loop.py [--kind c3|c4|sample|stream|packed] [--launches 600] [--bar 20 --inc <gain of 256> --dec <1/64 ticks>] [--waves 200] [--sweep lo hi step] [--json out]"""
import argparse
import json
import os
import random
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit('/', 2)[0])
import griduniverse_amd as gua  # noqa: E402
from griduniverse_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--kind', default='c3')
ap.add_argument('--launches', type=int, default=600)
ap.add_argument('--bar', type=int, default=None)
ap.add_argument('--inc', type=int, default=None)
ap.add_argument('--dec', type=int, default=None)
ap.add_argument('--waves', type=int, default=0)
ap.add_argument('--adapt', type=int, default=None)
ap.add_argument('--summary', action='store_true', help='one line: mean wall us per launch over the second half of the run')
ap.add_argument('--target', type=int, default=None)
ap.add_argument('--sweep', type=int, nargs=3, default=None)
ap.add_argument('--no-search', action='store_true')
ap.add_argument('--candidates', type=int, default=None)
ap.add_argument('--json', default=None)
args = ap.parse_args()

T = 1000
random.seed(123)
np.random.seed(123)
if args.kind == 'c4':
    env = gua.GridUniverseEnv(grid_shape=(32, 32), lava_states=[16 + 32 * r for r in range(24)], goal_states=[1023])
    N = 32768
else:
    env = gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True)
    N = 65536
spec = gua.GridSpec.from_env(env)
policy = {'c3': 'uniform', 'c4': 'uniform', 'sample': 'sample', 'stream': 'stream', 'packed': 'uniform'}[args.kind]
traj = 'packed' if args.kind == 'packed' else True
if args.candidates:
    _lib.set_default_option('traj_candidates', args.candidates)
for name, v in (('pace_adapt', args.adapt), ('pace_bar_num', args.bar), ('pace_gain_q', args.inc), ('pace_dec_q', args.dec), ('pace_target', args.target)):
    if v is not None:
        _lib.set_default_option(name, v)


def fresh(seed=123):
    eng = gua.Engine(N, spec, seed=seed)
    eng.reset()
    eng.reserve_trajectory(T)
    if policy == 'sample':
        S = spec.W * spec.H
        eng.vi_set(np.zeros(S), np.random.RandomState(1).dirichlet(np.ones(4), S))
    if policy == 'stream':
        eng.upload_actions(np.random.RandomState(2).randint(0, 4, (T, N)).astype(np.int32))
    return eng


def go(eng, n):
    for _ in range(n):
        eng.rollout(T, policy, True, traj)


def log_of(eng):
    return eng.rollout_pace_log(policy, True, packed=(traj == 'packed'))


out = dict(kind=args.kind, N=N, T=T, options=dict(bar=args.bar, inc=args.inc, dec=args.dec, target=args.target))


def chunk(eng, n, trace=None):
    """n launches back to back; wall us per launch (events) and, from the kind's ring, per launch: period, start-to-start us, verdict."""
    eng.sync()
    eng.timer_begin()
    go(eng, n)
    wall = eng.timer_end() / n * 1e3
    lg = log_of(eng)
    k = min(n, len(lg['seq']))
    if trace is not None:
        trace['period'] += [round(float(x), 2) for x in lg['period'][-k:]]
        trace['us'] += [round(float(x) / 100.0, 2) for x in lg['interval'][-k:]]
        trace['verdict'] += [int(x) for x in lg['verdict'][-k:]]
        trace['late_share'] += [int(x) for x in lg['phase'][-k:]]
        trace['behind_us'] += [round(float(x) / 100.0, 1) for x in lg['max_behind'][-k:]]
        trace['ended_late'] += [int(x) for x in lg['ended_late'][-k:]]
        trace.setdefault('dec_q', []).extend(int(x) for x in lg['dec_q'][-k:])
    return wall, lg


# ---- loop: a fresh engine, nothing asked for ------------------------------------------------------------
eng = fresh()
trace = dict(period=[], us=[], verdict=[], late_share=[], behind_us=[], ended_late=[])
chunks = []
done = 0
while done < args.launches:
    n = min(58, args.launches - done)
    wall, lg = chunk(eng, n, trace)
    if lg is None:
        print('this launch kind keeps no schedule')
        sys.exit(1)
    chunks.append(round(wall, 2))
    done += n
if args.summary:
    half = chunks[len(chunks) // 2:]
    per = np.array(trace['period'])
    per = per[per > 0]
    print('SUMMARY %s gain %s dec %s adapt %s: wall us per launch, second half of %d launches: mean %.2f min %.2f max %.2f | period p10 %.1f median %.1f p90 %.1f | probe %.4f' % (
        args.kind, args.inc, args.dec, args.adapt, args.launches, np.mean(half), np.min(half), np.max(half), np.percentile(per[len(per) // 2:], 10),
        np.median(per[len(per) // 2:]), np.percentile(per[len(per) // 2:], 90), eng.trajectory_placement()[1]))
print('== loop: %s, %d envs x %d steps, fresh engine, %d launches' % (args.kind, N, T, args.launches))
print('   placement', eng.trajectory_placement(), 'totals', eng.rollout_pacing_totals())
print('   first 40 launches: period       ', trace['period'][:40])
print('                      start-start us', trace['us'][:40])
print('                      verdict       ', trace['verdict'][:40])
print('                      behind us     ', trace['behind_us'][:40])
print('   phases seen (0 limiter on, 1 probing without, 2 off, 3 probing with):', sorted(set(trace['late_share'])), ' launches without the limiter:', sum(1 for x in trace['period'] if x == 0))
us = np.array(trace['us'])
per = np.array(trace['period'])
ver = np.array(trace['verdict'])
for a, b in ((0, 6), (6, 32), (32, 100), (100, 300), (300, 600), (600, 1200), (1200, len(us))):
    b = min(b, len(us))
    if b > a:
        seg = us[a:b]
        seg = seg[seg > 0]
        print('   launches %4d .. %4d: start-to-start us median %.2f mean %.2f max %.2f   periods %.1f .. %.1f   launches behind: %d' % (
            a, b - 1, np.median(seg), seg.mean(), seg.max(), per[a:b][per[a:b] > 0].min() if (per[a:b] > 0).any() else 0, per[a:b].max(), int((ver[a:b] == 2).sum())))
print('   wall us per launch by chunk of 60 (events):', chunks)
print('   the aim (dec_q, 1/64 ticks per launch) every 192 launches:', trace['dec_q'][::192])
out['loop'] = dict(trace=trace, wall_us_by_chunk=chunks)

# ---- what a launch that is behind looks like, wave by wave (launches one at a time) ----------------------------
if args.waves:
    shown = 0
    mb = []
    for i in range(args.waves):
        eng.rollout(T, policy, True, traj)
        el = eng.rollout_pace_waves(policy, True, packed=(traj == 'packed'))
        el = el[el > 0]
        mb.append(int(el.max() - np.median(el)))
        if mb[-1] > 500 and shown < 6:
            shown += 1
            pc = np.percentile(el, [0, 10, 50, 90, 99, 100]).astype(int)
            order = np.argsort(-el)[:8]
            print('   launch %d one at a time: ticks from start to report over %d waves: min %d p10 %d median %d p90 %d p99 %d max %d; worst waves %s (workgroup %s)' % (
                i, len(el), pc[0], pc[1], pc[2], pc[3], pc[4], pc[5], order.tolist(), (order // 4).tolist()))
    mb = np.array(mb)
    print('== one at a time, %d launches: slowest wave minus median wave, ticks: median %d p90 %d max %d; launches > 500: %d' % (
        args.waves, np.median(mb), np.percentile(mb, 90), mb.max(), int((mb > 500).sum())))

# ---- sweep: fixed periods on the same engine ----------------------------------------------------------------
if args.sweep:
    lo, hi, step = args.sweep
    rows = []
    print('== sweep (fixed periods, 40 launches each, back to back): period | start-to-start us median, mean | launches behind | wall us')
    for p in range(hi, lo - 1, -step):
        eng.set_option('rollout_pace', p)
        go(eng, 4)
        tr = dict(period=[], us=[], verdict=[], late_share=[], behind_us=[], ended_late=[])
        wall, lg = chunk(eng, 40, tr)
        u = np.array(tr['us'][:-1])
        row = [p, round(float(np.median(u)), 2), round(float(u.mean()), 2), int((np.array(tr['verdict']) == 2).sum()), round(wall, 2),
               round(float(np.median(tr['late_share'])), 3)]
        rows.append(row)
        print('   %4d | %7.2f %7.2f | %2d | %7.2f | phase/late %.3f' % tuple(row))
    eng.set_option('rollout_pace', None)
    out['sweep'] = rows

# (the open-loop search of rounds 3 and 4, which this tool held the loop against until round 5, left the library in round 6)
eng.close()
if args.json:
    os.makedirs(os.path.dirname(args.json) or '.', exist_ok=True)
    with open(args.json, 'w') as f:
        json.dump(out, f)
