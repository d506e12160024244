#!/usr/bin/env python3
"""Where a round of the per-XCD form of gu_vi_sweep_step_run spends its time: run against the diagnostic variant library
(make -C griduniverse_amd/csrc variant VARIANT=_stamps EXTRA=-DGU_VI_XCD_STAMPS; GU_LIB_PATH=.../libgu_stamps.so), which
returns per-phase shader-clock sums of one workgroup's first wave in place of the first twelve deltas.
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

PHASES = ['V1 from registers, granules stored', 'delta keys to LDS', 'own values to LDS, exchange loads issued, key wave: the workgroup\'s key reduced', 'workgroup barrier 1',
          'exchange: halo granules + action words, reloaded until tagged', 'delta collected (workgroup 0, wave 0)', 'exchange -> LDS; key wave: key posted',
          'workgroup barrier 2', 'V2, action words published', 'agent step (none in a wave that owns states)']


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=65536)
    ap.add_argument('--size', type=int, default=64)
    ap.add_argument('--rounds', type=int, default=2000)
    ap.add_argument('--blocks', type=int, nargs='+', default=[256, 512, 1024])
    a = ap.parse_args()
    random.seed(5)
    np.random.seed(5)
    env = gua.GridUniverseEnv(grid_shape=(a.size, a.size), random_maze=True)
    S = env.world.size
    out = {}
    for block in a.blocks:
        _lib.set_default_option('vi_xcd_block', block)
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
        cyc = [k / a.rounds for k in keys[:10]]
        ticks = keys[11]
        out[block] = {'cycles_per_round': {PHASES[i]: round(cyc[i], 1) for i in range(10)}, 'cycles_per_round_total': round(sum(cyc), 1),
                      'poll_turns_per_round': keys[10] / a.rounds, 'us_per_round_100MHz_clock': ticks / 100.0 / a.rounds,
                      'shader_clock_GHz': sum(cyc) * a.rounds / (ticks * 10.0)}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
