#!/usr/bin/env python3
"""A/B of the rollout kernel's cell-record maps in ONE process (the switches are read per launch), config-3 grid:
GU_ROLLOUT_REP=0 shared byte planes in LDS (MAP 1) / =1 replicated conflict-free dword records (MAP 4), GU_ROLLOUT_ROWS=0/1
the transition-row table (gu_rollout_rows.hip), GU_ROLLOUT_XCD=0/1 the XCD-aware env-block order, GU_ROLLOUT_BLOCK the
workgroup size.  Modes: stats-only, packed rows, int32 rows; uniform and
stream policies; two batch sizes.  Every variant's final state and stats are compared with the first variant's (they
must be identical), then launches are timed interleaved."""
import os
import random
import statistics
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import griduniverse_amd as gua  # noqa: E402

def sw(rep='0', rows='0', xcd='0', block='256'):
    return dict(GU_ROLLOUT_REP=rep, GU_ROLLOUT_ROWS=rows, GU_ROLLOUT_XCD=xcd, GU_ROLLOUT_BLOCK=block)


VARIANTS = [('planes', sw()), ('planes+xcd', sw(xcd='1')), ('planes/1024', sw(block='1024')), ('planes/1024+xcd', sw(block='1024', xcd='1')),
            ('rows', sw(rows='1')), ('rows+xcd', sw(rows='1', xcd='1'))]
# ('replicated', GU_ROLLOUT_REP=1: one dword record per cell replicated 32x, conflict-free) was part of this A/B in round 2 and
# lost to 'planes' (profiles/archive/r02b_map_ab.txt); the variant and its switch were removed from the library afterwards.
if len(sys.argv) > 1:
    VARIANTS = [v for v in VARIANTS if v[0] in sys.argv[1:]]


def use(env):
    os.environ.update(env)


def main():
    random.seed(123)
    np.random.seed(123)
    env = gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True)
    spec = gua.GridSpec.from_env(env)
    for N, T in ((65536, 1000), (262144, 500), (1 << 20, 125)):
        engines = {}
        for name, sw in VARIANTS:
            eng = gua.Engine(N, spec, seed=1)
            eng.reset()
            eng.reserve_trajectory(T)
            engines[name] = eng
        acts = np.random.RandomState(1).randint(0, 4, (T, N)).astype(np.int32) if N <= 65536 else None
        for policy in ('uniform', 'stream'):
            if policy == 'stream':
                if acts is None:
                    continue
                for eng in engines.values():
                    eng.upload_actions(acts)
            for mode, kw in (('stats-only', dict(trajectory=False, stats=True)), ('packed', dict(trajectory='packed', stats=False)),
                             ('int32 rows', dict(trajectory=True, stats=False)), ('int32 rows, no reset', dict(trajectory=True, stats=False, auto_reset=False))):
                auto = kw.pop('auto_reset', True)
                ref = None
                for name, sw in VARIANTS:  # identical results first
                    use(sw)
                    eng = engines[name]
                    eng.seed(1)
                    eng.reset()
                    eng.rollout(T, policy, auto, kw['trajectory'], kw['stats'])
                    st = eng.get_state()
                    sig = [st['pos'], st['done'], st['episode']]
                    if kw['stats']:
                        sig += list(eng.read_stats())
                    if kw['trajectory'] is True:
                        tr = eng.read_trajectory(T - 3, 3)
                        sig += [tr['obs'], tr['reward'], tr['done']]
                    elif kw['trajectory'] == 'packed':
                        sig += [eng.read_trajectory_packed(T - 3, 3, unpack=False)]
                    if ref is None:
                        ref = sig
                    else:
                        assert all(np.array_equal(a, b) for a, b in zip(ref, sig)), (name, policy, mode)
                times = {name: [] for name, _ in VARIANTS}
                for rnd in range(7):
                    for name, sw in VARIANTS:
                        use(sw)
                        eng = engines[name]
                        for _ in range(2):
                            eng.rollout(T, policy, auto, kw['trajectory'], kw['stats'])
                        eng.sync()
                        eng.timer_begin()
                        for _ in range(10):
                            eng.rollout(T, policy, auto, kw['trajectory'], kw['stats'])
                        times[name].append(eng.timer_end() / 10 * 1e3)
                line = '  '.join('%s %.1f us (min %.1f)' % (n, statistics.median(t), min(t)) for n, t in times.items())
                print('N %7d T %4d %-8s %-22s %s' % (N, T, policy, mode, line), flush=True)
        for eng in engines.values():
            eng.close()


if __name__ == '__main__':
    main()
