#!/usr/bin/env python3
"""Where a round of the per-XCD form of gu_vi_sweep_step_run spends its time: run against the diagnostic variant library
(make -C griduniverse_amd/csrc variant VARIANT=_stamps EXTRA=-DGU_VI_XCD_STAMPS; GU_LIB_PATH=.../libgu_stamps.so), which
returns per-phase shader-clock sums of ONE wave of one member (GU_VI_STAMP_RANK / GU_VI_STAMP_WAVE, set here per run) in place of
the first twelve deltas.  The stamped wave is ~250 clocks per round slower than it is unstamped; the others are not.
Usage: GU_LIB_PATH=griduniverse_amd/lib/libgu_stamps.so python tools/c5_stamps.py [--envs 65536] [--rounds 2000]"""
import argparse
import json
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import griduniverse_amd as gua  # noqa: E402
from griduniverse_amd import _lib  # noqa: E402

# indexed by stamp number; ORDER is the order in which a round passes them
PHASES = ['V1 from registers, granules stored', 'delta keys to LDS', 'own values to LDS, exchange loads issued (the tables alone: + the workgroup\'s key reduced and posted)',
          'workgroup barrier 1', 'exchange: halo granules + action words, reloaded until tagged', 'a workgroup whose waves all own states: key posted, deltas collected',
          'exchange -> LDS', 'workgroup barrier 2', 'V2, action words published',
          'stateless waves: deltas collected (workgroup 0, last wave), key reduced and posted (first of them), agent step; others: end of V2']
ORDER = [0, 1, 3, 2, 4, 6, 5, 7, 8, 9]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=65536)
    ap.add_argument('--size', type=int, default=64)
    ap.add_argument('--rounds', type=int, default=2000)
    ap.add_argument('--blocks', type=int, nargs='+', default=[256])
    ap.add_argument('--ranks', type=int, nargs='+', default=[0, 7], help='members of the writing cluster to stamp')
    ap.add_argument('--waves', type=int, nargs='+', default=[0, 1, 2, 3], help='waves of those members to stamp (one per run)')
    a = ap.parse_args()
    random.seed(5)
    np.random.seed(5)
    env = gua.GridUniverseEnv(grid_shape=(a.size, a.size), random_maze=True)
    S = env.world.size
    out = {'phases_in_time_order': [PHASES[i] for i in ORDER], 'runs': []}
    for block in a.blocks:
        _lib.set_default_option('vi_xcd_block', block)
        for rank in a.ranks:
            for wave in a.waves:
                if wave * 64 >= block:
                    continue
                os.environ['GU_VI_STAMP_RANK'], os.environ['GU_VI_STAMP_WAVE'] = str(rank), str(wave)  # (read by the variant library at launch)
                with gua.Engine(a.envs, gua.GridSpec.from_env(env), seed=5) as eng:
                    for rep in range(2):
                        eng.reset()
                        eng.vi_set(np.zeros(S), np.ones((S, 4)) / 4)
                        d = eng.vi_sweep_step_run(1.0, a.rounds, True)
                    assert eng.vi_last_form() == 1
                bits = d[:12].view(np.uint64)  # undo vi_unkey: the kernel stored raw sums where the keys go
                keys = []
                for b in bits.tolist():  # key -> double was: top bit set ? clear it : ~key; invert
                    keys.append((b | (1 << 63)) if not (b >> 63) else (~b) & ((1 << 64) - 1))
                life = None
                if a.rounds >= 20:  # the kernel's life beside the loop, 100 MHz ticks since entry -> us
                    lk = []
                    for b in d[13:19].view(np.uint64).tolist():
                        lk.append((b | (1 << 63)) if not (b >> 63) else (~b) & ((1 << 64) - 1))
                    life = dict(zip(['planes_staged', 'registered', 'loop_begins', 'loop_ends', 'last_exchange_done', 'results_written'], [round(x / 100.0, 2) for x in lk]))
                cyc = [k / a.rounds for k in keys[:10]]
                ticks = keys[11]
                out['runs'].append({'block': block, 'rank': rank, 'wave': wave, 'cycles_per_round': [round(cyc[i], 1) for i in ORDER],
                                    'cycles_per_round_total': round(sum(cyc), 1), 'poll_turns_per_round': keys[10] / a.rounds,
                                    'us_per_round_100MHz_clock': ticks / 100.0 / a.rounds, 'kernel_life_us_since_entry': life, 'shader_clock_GHz': sum(cyc) * a.rounds / (ticks * 10.0)})
    print(json.dumps(out, indent=1))
    print('\n'.join('block %4d rank %2d wave %d: ' % (r['block'], r['rank'], r['wave']) + ' '.join('%6.0f' % c for c in r['cycles_per_round'])
                    + '  | total %6.0f, polls %.2f' % (r['cycles_per_round_total'], r['poll_turns_per_round']) for r in out['runs']), file=sys.stderr)


if __name__ == '__main__':
    main()
