#!/usr/bin/env python3
"""Rows, state and statistics of two consecutive rollouts under every (traj_layout, rollout_rows) pair, with process-wide option
defaults from the command line (name=value ...): which pairs differ from (0, default), and where.  A debugging aid for the option
sessions of tools/gpu_soak_switches.sh.  Usage: python tools/variant_parity.py rollout_block=1024 [--root DIR]"""
import os
import sys

import numpy as np

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:]]
if '--root' in args:
    root = os.path.abspath(args[args.index('--root') + 1])
    del args[args.index('--root'):args.index('--root') + 2]
sys.path.insert(0, root)
from griduniverse_amd import Engine, GridSpec, _lib  # noqa: E402
from benchlib.workloads import build_workload  # noqa: E402

for item in args:
    k, v = item.split('=')
    _lib.set_default_option(k, int(v))
print('root', root, 'options', args)


def spec_of(name):
    return GridSpec.from_env(build_workload(name)[0])


for name, N, T in (('c2', 4096, 300), ('c4', 32768, 160), ('c3', 65536, 120)):
    for policy in ('uniform', 'greedy'):
        S = spec_of(name).S
        outs = {}
        for layout, rows in ((0, None), (1, None), (-1, None), (1, 0), (1, 3), (0, 0)):
            _lib.set_default_option('traj_layout', layout)
            _lib.set_default_option('rollout_rows', rows)
            with Engine(N, spec_of(name), seed=11, env_id0=5) as eng:
                eng.reset()
                eng.reserve_trajectory(T)
                if policy == 'greedy':
                    eng.vi_set(np.zeros(S), np.random.RandomState(1).dirichlet(np.ones(4), S))
                eng.rollout(T // 3, policy, True, True, stats=True)
                a = eng.read_trajectory(0, T // 3)
                eng.rollout(T, policy, True, True, stats=True)
                b = eng.read_trajectory(0, T)
                st = eng.get_state()
                outs[(layout, rows)] = dict(a_obs=a['obs'], a_rew=a['reward'], a_done=a['done'], b_obs=b['obs'], b_rew=b['reward'], b_done=b['done'],
                                            pos=st['pos'], done=st['done'], episode=st['episode'], ret=eng.read_stats()[0], fin=eng.read_stats()[1])
        ref = outs[(0, None)]
        for key, got in outs.items():
            bad = [k for k in ref if not np.array_equal(ref[k], got[k])]
            if bad:
                k = bad[0]
                where = np.argwhere(ref[k] != got[k])
                print('  %s %s N=%d T=%d: (layout, rows) = %s differs in %s; first at %s (%d places): %s vs %s'
                      % (name, policy, N, T, key, bad, where[0].tolist(), len(where), ref[k][tuple(where[0])], got[k][tuple(where[0])]))
        print('%s %s: compared' % (name, policy), flush=True)
