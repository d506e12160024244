#!/usr/bin/env python3
"""Stress of the per-XCD launches (csrc/gu_vi_xcd.hip: no barrier between workgroups, every word that crosses them carries its
round) under UNEVEN load: while a second engine on the same device streams rollouts from another thread -- so that workgroups of
the launch under test start late, share CUs and memory queues with a neighbour, and the clusters run skewed -- random cases are run
on the per-XCD form and compared, byte for byte (tables, deltas, every env's state), with the chip-wide cluster form of the same
case, which shares no code with it beyond the float64 helpers.  Cases: random grid sizes, batch sizes, round counts, workgroup
sizes, gamma; config 5's fused sweep + step loop and the tables alone (value iteration with a stopping rule, policy evaluation).
Usage: python tools/xcd_stress.py [seconds, default 60] [out.txt]"""
import hashlib
import os
import random
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import griduniverse_amd as gua  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
lines = []


def say(text):
    print(text, flush=True)
    lines.append(text)


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


stop = threading.Event()


def neighbour():
    """A second engine on the device: back-to-back rollouts with int32 rows (the HBM write stream at capacity), in bursts."""
    random.seed(1)
    np.random.seed(1)
    env = gua.GridUniverseEnv(grid_shape=(32, 32), random_maze=True)
    eng = gua.Engine(65536, gua.GridSpec.from_env(env), seed=2)
    eng.set_option('traj_candidates', 1)
    eng.reset()
    eng.reserve_trajectory(500)
    rs = np.random.RandomState(3)
    while not stop.is_set():
        for _ in range(int(rs.randint(1, 30))):
            eng.rollout(int(rs.randint(50, 500)), 'uniform', True, True)
        eng.sync()
        time.sleep(float(rs.rand()) * 0.003)
    eng.close()


load = threading.Thread(target=neighbour, daemon=True)
if os.environ.get('GU_STRESS_NO_NEIGHBOUR') != '1':
    load.start()
rs = np.random.RandomState(11)
t0, cases, forms = time.time(), 0, {1: 0, 2: 0, 3: 0}
torn, torn_build, xcd_rounds = 0, False, 0  # (a -DGU_VI_XCD_TORN build of the library counts halves of the exchange torn at four bytes)
while time.time() - t0 < budget:
    w, h = int(rs.randint(6, 110)), int(rs.randint(6, 110))
    random.seed(int(rs.randint(1 << 30)))
    np.random.seed(int(rs.randint(1 << 30)))
    env = gua.GridUniverseEnv(grid_shape=(w, h), random_maze=True)
    S = env.world.size
    spec = gua.GridSpec.from_env(env)
    gamma = float(rs.choice([1.0, 0.99, 0.9]))
    if rs.rand() < 0.6:  # config 5's loop
        N = int(rs.choice([3, 300, 4096, 20000, 65536, 100000]))
        rounds = int(rs.randint(1, 400))
        out = {}
        for form, path in (('per_xcd', 6), ('chip_wide', 4)):  # (6: the per-XCD form for the short second call too)
            with gua.Engine(N, spec, seed=5) as eng:
                eng.set_option('vi_path', path)
                eng.set_option('vi_xcd_block', int(rs.choice([0, 256, 512, 1024])) if path == 6 else None)
                eng.reset()
                eng.vi_set(np.zeros(S), np.full((S, 4), 0.25))
                d1 = eng.vi_sweep_step_run(gamma, rounds, True)
                d2 = eng.vi_sweep_step_run(gamma, 1 + rounds % 3, True)
                st = eng.get_state()
                out[form] = digest(d1, d2, *eng.vi_get(), st['pos'], st['done'], st['episode'], eng.read_outputs()[1])
                if form == 'per_xcd':
                    forms[eng.vi_last_form()] += 1
                    n_torn = eng.vi_xcd_torn_words()
                    if n_torn is not None:
                        torn_build, torn, xcd_rounds = True, torn + n_torn, xcd_rounds + rounds + 1 + rounds % 3
        assert out['per_xcd'] == out['chip_wide'], ('sweep+step', w, h, N, rounds, gamma)
    else:  # the tables alone
        out = {}
        T = int(rs.randint(5, 300))
        for form, path in (('per_xcd', 6), ('other', 4)):  # (6: whatever the grid's size and the call's length)
            with gua.Engine(8, spec, seed=5) as eng:
                eng.set_option('vi_path', path)
                eng.vi_set(np.zeros(S), np.full((S, 4), 0.25))
                steps, deltas = eng.vi_run(gamma if gamma < 1.0 else 0.97, 1e-4, T)
                v, pi = eng.vi_get()
                steps2, deltas2 = eng.vi_eval_run(0.9, 1e-3, 40)
                out[form] = digest(np.int64(steps), deltas, v, pi, np.int64(steps2), deltas2, eng.vi_get()[0])
                if form == 'per_xcd':
                    n_torn = eng.vi_xcd_torn_words()
                    if n_torn is not None:
                        torn_build, torn, xcd_rounds = True, torn + n_torn, xcd_rounds + int(steps) + int(steps2)
        assert out['per_xcd'] == out['other'], ('tables', w, h, gamma)
    cases += 1
stop.set()
if load.is_alive():
    load.join(timeout=30)
say('%d random cases in %.0f s under a streaming neighbour on the device, every one byte-identical on the per-XCD form and the chip-wide / '
    'single-workgroup form; form taken by the fused sweep + step cases (1 per XCD, 2 chip-wide, 3 per launch): %r' % (cases, time.time() - t0, forms))
if torn_build:
    say('torn-half detector (library built with -DGU_VI_XCD_TORN: every tag word carries a check of the payload it was stored with): %d words '
        'with the right tag and the wrong payload in %d rounds of per-XCD launches' % (torn, xcd_rounds))
else:
    say('(product library: no torn-half detector -- make variant VARIANT=_torn EXTRA=-DGU_VI_XCD_TORN, GU_LIB_PATH=.../libgu_torn.so)')
if len(sys.argv) > 2:
    open(sys.argv[2], 'w').write('\n'.join(lines) + '\n')
