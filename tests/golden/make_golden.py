#!/usr/bin/env python3
"""Capture golden vectors from the REAL reference (TheMTank/GridUniverse).

Runs only in the build container, where /root/reference is mounted.  It imports
the reference under the stub `gym` in tests/golden/gym_stub (the reference's only missing
dependency; none of gym's arithmetic is on the path) and writes DATA fixtures
(inputs + expected outputs) to tests/golden/.  Nothing of the reference's source
is copied; the fixtures are what the oracle and the HIP kernels are pinned to.

    python tests/golden/make_golden.py            # regenerate everything (~2-3 min)

Random-action streams come from the build's own counter RNG (oracle/gu_rng.py,
a restatement of MurmurHash3) and are stored IN the fixtures, so the fixtures
are self-contained.  Start states of multi-start levels are chosen by the same
RNG (stream 1) and forced onto the reference instance right after its own
`reset()`; the auto-reset policy is the reference harness's `if done: reset()`
(examples/griduniverse_env_examples.py:22-24 breaks instead; monte_carlo.py:25).
"""
import contextlib
import hashlib
import io
import json
import os
import random
import sys
import warnings

os.environ.setdefault('MPLBACKEND', 'Agg')
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get('GU_REFERENCE', '/root/reference')
sys.path[:0] = [os.path.join(HERE, 'gym_stub'), REF, REPO]

import numpy as np  # noqa: E402

if not hasattr(np, 'float'):
    np.float = float  # utils.py:71 uses the alias numpy removed in 1.24

import matplotlib.pyplot as pyplot  # noqa: E402
from core.envs.griduniverse_env import GridUniverseEnv  # noqa: E402  (the reference)
from core.algorithms import utils as ref_utils  # noqa: E402
from core.algorithms import dynamic_programming as ref_dp  # noqa: E402
from oracle import gu_rng  # noqa: E402

OUT = HERE
LEVELS = os.path.join(REF, 'core', 'envs', 'maze_text_files')


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()):
        yield
    pyplot.close('all')  # the generator leaks one figure per call (maze_generation.py:105)


def ref_env(**kw):
    with quiet():
        return GridUniverseEnv(**kw)


def seeded_maze_env(w, h, k):
    random.seed(k)
    np.random.seed(k)
    return ref_env(grid_shape=(w, h), random_maze=True)


def spec_of(env):
    """Grid description as plain lists (what the build's engine is configured from)."""
    return dict(W=int(env.x_max), H=int(env.y_max),
                starts=[int(s) for s in env.starting_states],
                goals=[int(s) for s in env.goal_states],
                lava=[int(s) for s in env.lava_states],
                walls=[int(s) for s in env.wall_indices],
                reward=[int(r) for r in env.reward_matrix])


def rollout(env, actions, seed, env_ids, auto_reset):
    """Drive the reference step() for every env id (one instance, time-multiplexed).

    Returns obs, reward, done as int32 [T, N] plus the start index used for every
    episode start, in order, per env."""
    T, N = actions.shape
    obs = np.zeros((T, N), np.int32)
    rew = np.zeros((T, N), np.int32)
    don = np.zeros((T, N), np.int32)
    starts = list(env.starting_states)
    first_state = np.zeros(N, np.int32)
    for j, gid in enumerate(env_ids):
        episode = 0

        def do_reset():
            nonlocal episode
            env.reset()
            s = starts[gu_rng.start_index(seed, gid, episode, len(starts))]
            env.current_state = env.previous_state = env.initial_state = s
            episode += 1
            return s
        first_state[j] = do_reset()
        done = False
        for t in range(T):
            if auto_reset and done:
                do_reset()
            o, r, done, _ = env.step(int(actions[t, j]))
            obs[t, j], rew[t, j], don[t, j] = o, r, done
    return obs, rew, don, first_state


def digest(obs, rew, don):
    h = hashlib.sha256()
    for a in (obs, rew, don):
        h.update(np.ascontiguousarray(a, dtype='<i4').tobytes())
    return h.hexdigest()


def save_traj(name, env, seed, N, T, auto_reset, actions=None, env_id0=0, note=''):
    env_ids = list(range(env_id0, env_id0 + N))
    if actions is None:
        actions = gu_rng.action_stream(seed, env_ids, 0, T)
    obs, rew, don, first = rollout(env, actions, seed, env_ids, auto_reset)
    meta = dict(spec_of(env), seed=int(seed), N=N, T=T, auto_reset=bool(auto_reset), env_id0=env_id0, note=note,
                sha256=digest(obs, rew, don))
    np.savez_compressed(os.path.join(OUT, 'traj_%s.npz' % name), meta=json.dumps(meta),
                        actions=actions.astype(np.int32), obs=obs, reward=rew, done=don, first_state=first)
    print('traj', name, 'N', N, 'T', T, 'done-rate %.4f' % don.mean(), meta['sha256'][:16])
    return meta


def save_digest(store, name, env, seed, N, T, auto_reset):
    env_ids = list(range(N))
    actions = gu_rng.action_stream(seed, env_ids, 0, T)
    obs, rew, don, _ = rollout(env, actions, seed, env_ids, auto_reset)
    store[name] = dict(spec_of(env), seed=int(seed), N=N, T=T, auto_reset=bool(auto_reset),
                       sha256=digest(obs, rew, don), sum_reward=int(rew.sum()), sum_done=int(don.sum()),
                       sha256_final_obs=hashlib.sha256(obs[-1].astype('<i4').tobytes()).hexdigest())
    print('digest', name, store[name]['sha256'][:16])


def lava_column_32():
    # examples/griduniverse_env_examples.py:80 column pattern scaled to 32x32
    return [16 + 32 * r for r in range(24)]


# ----------------------------------------------------------------------------- G1
def capture_kats():
    """The reference's own ten tests (tests/test_griduniverse.py) as data."""
    kats = []

    def run(name, kwargs, actions, reset_first=False, level=None):
        kw = dict(kwargs)
        if level:
            kw['custom_world_fp'] = os.path.join(LEVELS, level)
        results = []
        # a level with several 'x' cells starts at a random one; the KAT must hold for each
        for pick in (range(2) if level else [None]):
            env = ref_env(**kw)
            if reset_first:
                env.reset()
            if pick is not None:
                env.current_state = env.previous_state = env.initial_state = env.starting_states[pick]
            first = int(env.current_state)
            steps = []
            for a in actions:
                o, r, d, _ = env.step(a)
                steps.append([int(o), int(r), bool(d)])
            results.append(dict(first_state=first, steps=steps))
        kats.append(dict(name=name, kwargs=kwargs, level=level, actions=list(actions), runs=results))

    run('wall_not_trespassed', dict(walls=[1]), [1])
    run('default_completion_in_six_steps', {}, [1, 1, 1, 2, 2, 2])
    run('large_completion_in_53_steps', dict(grid_shape=[25, 30]), [1] * 24 + [2] * 29)
    run('custom_from_text_file', {}, [2, 2, 2, 2, 2, 2, 2, 1], level='test_env.txt')
    run('each_boundary_within_default_env', {}, [3, 0, 1, 1, 1, 1, 2, 2, 3, 2, 2, 3, 3, 3], reset_first=True)
    run('lava', dict(lava_states=[1]), [1])
    run('lava_from_text_file', {}, [2, 2, 2, 1, 1], level='test_env.txt')
    return kats


# ----------------------------------------------------------------------------- G6
def capture_errors():
    cases = [
        dict(goal_states=[16]), dict(goal_states=['a']), dict(goal_states=5.0), dict(lava_states='a'),
        dict(walls='aaaa'), dict(grid_shape=[2, 2, 2]), dict(grid_shape='set'), dict(grid_shape=[2, 2.0]),
        dict(grid_shape=2), dict(lava_states=[16]), dict(lava_states=['b']), dict(walls=[16]), dict(walls=[-1]),
        dict(goal_states=[3.5]), dict(goal_states=[-17]), dict(lava_states=[-17]),
    ]
    out = []
    for kw in cases:
        real = dict(kw)
        if real.get('grid_shape') == 'set':
            real['grid_shape'] = set([2, 3])
        try:
            ref_env(**real)
            out.append(dict(kwargs=kw, error=None))
        except Exception as e:  # noqa: BLE001
            out.append(dict(kwargs=kw, error=type(e).__name__, message=str(e)))
    # loader errors (env:279,293,298,300)
    for name, lines in [('not_rectangle', ['xo', 'oGo']), ('bad_char', ['xo', 'oT']),
                        ('no_start', ['oo', 'oG']), ('no_goal', ['xo', 'oo'])]:
        env = ref_env()
        try:
            with quiet():
                env._create_custom_world_from_text(lines)
            out.append(dict(lines=lines, name=name, error=None))
        except Exception as e:  # noqa: BLE001
            out.append(dict(lines=lines, name=name, error=type(e).__name__, message=str(e)))
    # action domain (quirk 6)
    env = ref_env()
    for a in (4, -1, -4, -5):
        env.reset()
        env.current_state = 5
        try:
            o, r, d, _ = env.step(a)
            out.append(dict(step_action=a, from_state=5, error=None, result=[int(o), int(r), bool(d)]))
        except Exception as e:  # noqa: BLE001
            out.append(dict(step_action=a, from_state=5, error=type(e).__name__))
    return out


# ----------------------------------------------------------------------------- G5/G7
def capture_render_and_quirks():
    out = {}
    env = ref_env(walls=[1], lava_states=[2])
    out['render_walls1_lava2'] = env.render(mode='ansi').getvalue()
    env = ref_env()
    frames = [env.render(mode='ansi').getvalue()]
    for a in [1, 2, 2, 1, 1, 2]:
        env.step(a)
        frames.append(env.render(mode='ansi').getvalue())
    out['render_default_walk'] = frames
    env = ref_env(custom_world_fp=os.path.join(LEVELS, 'test_env.txt'))
    env.current_state = env.starting_states[0]
    out['render_test_env'] = env.render(mode='ansi').getvalue()
    env = ref_env(grid_shape=(5, 3), goal_states=[14, 7], lava_states=[7, 3], walls=[6, 14])
    out['render_5x3_overlaps'] = env.render(mode='ansi').getvalue()

    quirks = {}
    # 1 absorbing terminals
    env = ref_env()
    env.current_state = 11
    quirks['absorbing'] = [[int(o), int(r), bool(d)] for o, r, d, _ in (env.step(a) for a in [2, 0, 3, 1])]
    # 2 start on a wall, walk off, cannot walk back
    env = ref_env(walls=[0])
    quirks['start_on_wall'] = [[int(o), int(r), bool(d)] for o, r, d, _ in (env.step(a) for a in [1, 3, 2, 0])]
    # 3 goal that is a wall: unreachable but terminal if you are placed on it
    env = ref_env(goal_states=[5], walls=[5])
    seq = [[int(o), int(r), bool(d)] for o, r, d, _ in (env.step(a) for a in [1, 2, 2, 0])]
    env.current_state = 5
    seq.append([int(x) if i < 2 else bool(x) for i, x in enumerate(env.step(1)[:3])])
    quirks['goal_is_wall'] = seq
    # 4 cell both goal and lava -> -10
    env = ref_env(goal_states=[1, 15], lava_states=[1])
    quirks['goal_and_lava'] = [[int(o), int(r), bool(d)] for o, r, d, _ in (env.step(a) for a in [1, 1])]
    # 5 negative goal index: reward wraps, terminal test does not
    env = ref_env(goal_states=[-1])
    env.current_state = 14
    quirks['negative_goal'] = dict(reward=[int(r) for r in env.reward_matrix], goal_states=[-1],
                                   steps=[[int(o), int(r), bool(d)] for o, r, d, _ in (env.step(a) for a in [1, 1, 3])])
    # 7 stale observation_space after loading
    env = ref_env(custom_world_fp=os.path.join(LEVELS, 'test_env.txt'))
    quirks['stale_observation_space'] = dict(n=int(env.observation_space.n), world_size=int(env.world.size))
    # 8 return types
    env = ref_env()
    o, r, d, i = env.step(1)
    quirks['types'] = [type(o).__name__, type(r).__name__, type(d).__name__, type(i).__name__]
    # care_about_terminal=False (only user: maze_solving.py:48)
    env = ref_env(lava_states=[1])
    quirks['care_about_terminal_false'] = [
        [int(x) if k < 2 else bool(x) for k, x in enumerate(env.look_step_ahead(s, a, c))]
        for (s, a, c) in [(1, 1, True), (1, 1, False), (15, 3, True), (15, 3, False), (1, 2, False), (0, 1, False)]]
    # look_step_ahead over the whole (state, action) table of a busy grid
    env = ref_env(grid_shape=(6, 5), goal_states=[29, 8], lava_states=[13, 8], walls=[7, 14, 20, 29],
                  initial_state=[0, 3])
    table = {}
    for care in (True, False):
        table[str(care)] = [[[int(x) if k < 2 else bool(x) for k, x in enumerate(env.look_step_ahead(s, a, care))]
                             for a in range(4)] for s in range(30)]
    quirks['lsa_table_6x5'] = dict(spec=spec_of(env), table=table)
    out['quirks'] = quirks
    return out


# ----------------------------------------------------------------------------- G3/G8
def capture_mazes_and_levels():
    mazes = {}
    for (w, h) in [(8, 8), (11, 11), (7, 5), (5, 7), (32, 32), (64, 64), (21, 13)]:
        for k in ([0, 1, 123] if w * h <= 1024 else [5, 123]):
            env = seeded_maze_env(w, h, k)
            tail = [random.random(), float(np.random.random())]  # RNG positions after construction
            rows = []
            sp = spec_of(env)
            for y in range(h):
                row = ''
                for x in range(w):
                    s = y * w + x
                    row += '#' if s in set(sp['walls']) else 'x' if s in sp['starts'] else 'G' if s in sp['goals'] else 'o'
                rows.append(row)
            mazes['%dx%d_seed%d' % (w, h, k)] = dict(W=w, H=h, seed=k, rows=rows, start=sp['starts'], goal=sp['goals'],
                                                     n_walls=len(sp['walls']), initial_state=int(env.initial_state),
                                                     rng_tail=tail)
    levels = {}
    for fn in sorted(os.listdir(LEVELS)):
        random.seed(7)
        env = ref_env(custom_world_fp=os.path.join(LEVELS, fn))
        sp = spec_of(env)
        del sp['reward']
        sp['initial_state_seed7'] = int(env.initial_state)
        sp['observation_space_n'] = int(env.observation_space.n)
        levels[fn] = sp
    return mazes, levels


# ----------------------------------------------------------------------------- G4
def capture_dp():
    def uniform(env):
        return np.ones([env.world.size, 4]) / 4

    cases = [('maze8_s1', lambda: seeded_maze_env(8, 8, 1), 1.0, 12),
             ('maze11_s3_g09', lambda: seeded_maze_env(11, 11, 3), 0.9, 15),
             ('lava4x4_g095', lambda: ref_env(lava_states=[5, 6, 9]), 0.95, 12),
             ('rect6x5_g1', lambda: ref_env(grid_shape=(6, 5), goal_states=[29, 8], lava_states=[13], walls=[7, 14, 20],
                                            initial_state=[0, 3]), 1.0, 10),
             ('maze32_s1', lambda: seeded_maze_env(32, 32, 1), 1.0, 12),
             ('maze32_s1_g099', lambda: seeded_maze_env(32, 32, 1), 0.99, 8),
             ('maze64_s5', lambda: seeded_maze_env(64, 64, 5), 1.0, 6),
             ('maze64_s5_g097', lambda: seeded_maze_env(64, 64, 5), 0.97, 6),
             # beyond 4096 states (the DP cluster kernel): the shipped 101x101 level and a 128x128 generator maze
             ('level101_g099', lambda: ref_env(custom_world_fp=os.path.join(LEVELS, 'maze_101x101.txt')), 0.99, 8),
             ('maze128_s7', lambda: seeded_maze_env(128, 128, 7), 1.0, 4)]
    only = [a[3:] for a in sys.argv[1:] if a.startswith('dp=')]
    for name, make, gamma, iters in cases:
        if only and name not in only:
            continue
        env = make()
        S = env.world.size
        arrays = {}
        # (a) repeated V1 under the uniform policy (examples/griduniverse_alg_examples.py:31-35)
        v = np.zeros(S)
        pi0 = uniform(env)
        for k in range(1, 11):
            v = ref_utils.single_step_policy_evaluation(pi0, env, discount_factor=gamma, value_function=v)
            if k in (1, 2, 10):
                arrays['eval_v_%d' % k] = v.copy()
        arrays['greedy_pi_after_10'] = ref_utils.greedy_policy_from_value_function(
            uniform(env), env, v, discount_factor=gamma).copy()
        # (b) value_iteration trace: one V1+V2 per iteration (dynamic_programming.py:15-20)
        v = np.zeros(S)
        pi = uniform(env)
        deltas = []
        for k in range(1, iters + 1):
            v_new = ref_utils.single_step_policy_evaluation(pi, env, discount_factor=gamma, value_function=v)
            deltas.append(float(np.max(v - v_new)))
            v = v_new
            pi = ref_utils.greedy_policy_from_value_function(pi, env, value_function=v, discount_factor=gamma)
            if S <= 4096 or k in (1, 2, iters):  # the big grids keep the first two and the last round only
                arrays['vi_v_%d' % k] = v.copy()
                arrays['vi_pi_%d' % k] = pi.copy()
        # (c) the reference driver itself
        with warnings.catch_warnings(record=True) as wlist:
            warnings.simplefilter('always')
            v3, pi3 = ref_dp.value_iteration(uniform(env), env, np.zeros(S), threshold=1e-3, max_steps=iters,
                                             discount_factor=gamma)
        arrays['vi_driver_v'] = v3
        arrays['vi_driver_pi'] = pi3
        meta = dict(spec_of(env), gamma=gamma, iters=iters, deltas=deltas, driver_warned=len(wlist) > 0,
                    driver_threshold=1e-3)
        if S <= 1024:
            with warnings.catch_warnings(record=True) as wlist:
                warnings.simplefilter('always')
                v4, pi4 = ref_dp.policy_iteration(uniform(env), env, np.zeros(S), threshold=1e-3, max_steps=60,
                                                  discount_factor=gamma)
            arrays['pi_driver_v'] = v4
            arrays['pi_driver_pi'] = pi4
            meta['pi_driver_warned'] = len(wlist) > 0
            meta['pi_driver_max_steps'] = 60
        np.savez_compressed(os.path.join(OUT, 'dp_%s.npz' % name), meta=json.dumps(meta), **arrays)
        print('dp', name, 'S', S, 'gamma', gamma, 'deltas[-1]', deltas[-1])


# ----------------------------------------------------------------------------- G9
def capture_mc():
    """monte_carlo_evaluation (core/algorithms/monte_carlo.py:29-99) of the REAL reference, with its
    run_episode replaced by a replay of episodes generated with the build's RNG (stream 2 sampling):
    episode e = env e of one no-auto-reset rollout of the oracle."""
    import core.algorithms.monte_carlo as ref_mc
    from oracle import c_oracle as C

    rs = np.random.RandomState(11)
    cases = [('maze8_uniform', lambda: seeded_maze_env(8, 8, 1), None, 48, 160, 21),
             ('lava4x4_dirichlet', lambda: ref_env(lava_states=[5, 6, 9]), 'dirichlet', 40, 60, 22),
             ('rect6x5_multistart', lambda: ref_env(grid_shape=(6, 5), goal_states=[29, 8], lava_states=[13], walls=[7, 14, 20],
                                                    initial_state=[0, 3, 27]), 'dirichlet', 64, 80, 23)]
    combos = [dict(every_visit=ev, incremental_mean=im, stationary_env=st)
              for ev in (False, True) for im, st in ((True, True), (True, False), (False, True))]
    for name, make, kind, N, T, seed in cases:
        env = make()
        S = env.world.size
        policy = np.ones((S, 4)) / 4 if kind is None else rs.dirichlet(np.ones(4) * 0.7, S)
        grid = C.Grid.from_env(env)
        st = C.State(N)
        first = C.reset(grid, seed, st)
        traj = C.rollout(grid, seed, st, T, auto_reset=False, pi=policy)
        episodes = []
        for e in range(N):
            hits = np.flatnonzero(traj['done'][:, e])
            L = int(hits[0]) + 1 if hits.size else T
            episodes.append(([int(first[e])] + [int(x) for x in traj['obs'][:L, e]],
                             [np.int64(x) for x in traj['reward'][:L, e]], bool(traj['done'][L - 1, e])))
        arrays = dict(policy=policy, first_state=first, obs=traj['obs'], reward=traj['reward'], done=traj['done'])
        meta = dict(spec_of(env), seed=seed, N=N, T=T, runs=[])
        for gamma, thr, alpha in ((0.99, 1e-4, 0.001), (0.9, 1e-3, 0.05), (1.0, 1e-4, 0.2)):
            for c in combos:
                replay = iter(episodes)
                original = ref_mc.run_episode
                ref_mc.run_episode = lambda policy, env, max_steps_per_episode=1000: next(replay)
                try:
                    with quiet():
                        v = ref_mc.monte_carlo_evaluation(policy, env, discount_factor=gamma, threshold=thr, alpha=alpha,
                                                          num_episodes=N, **c)
                finally:
                    ref_mc.run_episode = original
                key = 'v_%d' % len(meta['runs'])
                arrays[key] = v
                meta['runs'].append(dict(c, discount_factor=gamma, threshold=thr, alpha=alpha, key=key))
        np.savez_compressed(os.path.join(OUT, 'mc_%s.npz' % name), meta=json.dumps(meta), **arrays)
        print('mc', name, 'S', S, 'episodes', N, 'mean length %.1f' % np.mean([len(e[1]) for e in episodes]),
              'terminal %.2f' % np.mean([e[2] for e in episodes]))


def capture_mc_numpy_rng():
    """monte_carlo_evaluation of the REAL reference exactly as a user calls it: its own run_episode, actions drawn with
    np.random.choice from numpy's GLOBAL stream (core/algorithms/monte_carlo.py:20), start cells with the stdlib's
    random.choice (core/envs/griduniverse_env.py:189), both seeded right before the call.  Stored: the returned value function
    (raw float64) per flag combination, plus the next draw of both global streams afterwards (how much each consumed)."""
    import core.algorithms.monte_carlo as ref_mc
    rs = np.random.RandomState(12)
    cases = [('maze8_uniform', lambda: seeded_maze_env(8, 8, 1), None, 25, 31),
             ('lava4x4_dirichlet', lambda: ref_env(lava_states=[5, 6, 9]), 'dirichlet', 40, 32),
             ('rect6x5_multistart', lambda: ref_env(grid_shape=(6, 5), goal_states=[29, 8], lava_states=[13], walls=[7, 14, 20],
                                                    initial_state=[0, 3, 27]), 'sparse', 30, 33)]
    combos = [dict(every_visit=ev, incremental_mean=im, stationary_env=st)
              for ev in (False, True) for im, st in ((True, True), (True, False), (False, True))]
    for name, make, kind, episodes, seed in cases:
        env = make()
        S = env.world.size
        if kind is None:
            policy = np.ones((S, 4)) / 4
        else:
            policy = rs.dirichlet(np.ones(4) * 0.7, S)
            if kind == 'sparse':  # rows with exact zeros and ones: thresholds at both ends of the cumulative table
                policy[::5] = np.eye(4)[rs.randint(0, 4, len(policy[::5]))]
                policy[1::7] = [0.5, 0.0, 0.5, 0.0]
        arrays = dict(policy=policy)
        meta = dict(spec_of(env), seed=seed, num_episodes=episodes, runs=[])
        for gamma, thr, alpha in ((0.99, 1e-4, 0.001), (0.9, 1e-3, 0.05)):
            for c in combos:
                random.seed(seed)
                np.random.seed(seed)
                with quiet():
                    v = ref_mc.monte_carlo_evaluation(policy, env, discount_factor=gamma, threshold=thr, alpha=alpha,
                                                      num_episodes=episodes, **c)
                key = 'v_%d' % len(meta['runs'])
                arrays[key] = v
                meta['runs'].append(dict(c, discount_factor=gamma, threshold=thr, alpha=alpha, key=key,
                                         next_numpy_uniform=float(np.random.random_sample()), next_stdlib_uniform=random.random()))
        np.savez_compressed(os.path.join(OUT, 'mcnp_%s.npz' % name), meta=json.dumps(meta), **arrays)
        print('mcnp', name, 'S', S, 'episodes', episodes, 'runs', len(meta['runs']))


# ----------------------------------------------------------------------------- G12
def capture_driver():
    """tests/golden/driver_flow.py (a whole driver in the style of the reference's examples) on the REAL reference."""
    import core.algorithms.monte_carlo as ref_mc
    sys.path.insert(0, HERE)
    from driver_flow import driver_flow
    with quiet():
        out = driver_flow(GridUniverseEnv, ref_utils, ref_dp, ref_mc)
    np.savez_compressed(os.path.join(OUT, 'driver_alg_examples.npz'), **out)
    print('driver', {k: v.shape for k, v in out.items()})


# ----------------------------------------------------------------------------- G10
def reference_path_search():
    """The reference's breadth-first path search lives inside the `if __name__ == '__main__':` demo loop of
    core/algorithms/maze_solving.py (:43-50 create_graph, :113-127 calculate_action, :129-169 breadth_first_search,
    :171-193 construct_path), next to window / sleep calls, so the module can be neither imported nor run headless.
    Here its four FUNCTION DEFINITIONS are lifted out of the parsed module by name (ast; no GUI statement runs) and
    compiled, unchanged, into a namespace that supplies the free variables the demo loop binds around them
    (`env`, `actions`, `nodes_and_edges`).  What runs is the reference's own code object."""
    import ast
    path = os.path.join(REF, 'core', 'algorithms', 'maze_solving.py')
    tree = ast.parse(open(path).read(), path)
    wanted = ('create_graph', 'calculate_action', 'breadth_first_search', 'construct_path')
    defs = {n.name: n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name in wanted}
    assert sorted(defs) == sorted(wanted), sorted(defs)
    module = ast.Module(body=[defs[k] for k in wanted], type_ignores=[])
    code = compile(ast.fix_missing_locations(module), path, 'exec')

    def search(env, start_state):
        """(action list or None, terminal state or None, exception name or None) of the reference's search."""
        ns = dict(env=env, actions=range(4), nodes_and_edges={})
        exec(code, ns)
        reached = []
        construct = ns['construct_path']

        def recording_construct_path(state, meta):
            reached.append(int(state))
            return construct(state, meta)
        ns['construct_path'] = recording_construct_path
        with quiet():
            ns['create_graph']()
            try:
                path_found = ns['breadth_first_search'](ns['nodes_and_edges'], start_state)
            except KeyError:
                return None, None, 'KeyError', ns['nodes_and_edges']
        if path_found is None:  # the queue ran empty (:167-169 call construct_path and drop its result)
            return None, None, None, ns['nodes_and_edges']
        return [int(a) for a in path_found], reached[-1], None, ns['nodes_and_edges']
    return search


def capture_bfs():
    search = reference_path_search()
    rs = np.random.RandomState(4)
    cases = []

    def add(name, env, start=None, note=''):
        start = int(env.initial_state if start is None else start)
        path_found, terminal, error, graph = search(env, start)
        n_edges = int(sum(len(v) for v in graph.values()))
        cases.append(dict(spec_of(env), name=name, start=start, path=path_found, terminal=terminal, error=error,
                          graph_nodes=len(graph), graph_edges=n_edges, note=note))
        print('bfs', name, 'start', start, 'len', None if path_found is None else len(path_found), 'terminal', terminal, error or '')

    add('default4x4', ref_env())
    add('open8x8_ties', ref_env(grid_shape=(8, 8), goal_states=[63, 36, 7], walls=[9, 10, 17, 27, 35, 43]),
        note='several shortest paths: FIFO order and action order decide')
    add('lava_nearer_than_goal', ref_env(grid_shape=(8, 8), lava_states=[18, 42], walls=[1]), note='terminal reached is lava')
    add('goal_and_lava_same_cell', ref_env(grid_shape=(5, 5), goal_states=[12], lava_states=[12, 3]))
    add('start_is_terminal', ref_env(grid_shape=(6, 4), initial_state=9, goal_states=[9]), note='empty path')
    add('start_walled_in', ref_env(grid_shape=(10, 7), goal_states=[69], walls=[1, 10, 11]), note='queue runs empty: None')
    add('goal_behind_walls', ref_env(grid_shape=(6, 6), goal_states=[35], walls=[29, 34, 14]),
        note='goal unreachable')
    add('start_on_wall', ref_env(grid_shape=(4, 4), walls=[0, 5]), note='wall start is no graph node: KeyError at :140')
    add('column1x9', ref_env(grid_shape=(1, 9)), note='W=1: vertical neighbours differ by 1 -> calculate_action names them LEFT/RIGHT')
    add('row9x1', ref_env(grid_shape=(9, 1)))
    add('two_wide', ref_env(grid_shape=(2, 7), walls=[3, 6]))
    W, H = 25, 30
    cells = rs.permutation(W * H)
    add('rect25x30_busy', ref_env(grid_shape=(W, H), initial_state=int(cells[0]), goal_states=[int(c) for c in cells[30:34]],
                                  lava_states=[int(c) for c in cells[40:52]], walls=[int(c) for c in cells[100:300]]))
    W, H = 40, 12
    cells = rs.permutation(W * H)
    add('wide40x12', ref_env(grid_shape=(W, H), initial_state=int(cells[0]), goal_states=[int(c) for c in cells[2:4]],
                             lava_states=[int(c) for c in cells[6:12]], walls=[int(c) for c in cells[20:150]]))
    for w, h, k in ((15, 15, 0), (15, 15, 1), (8, 8, 3), (11, 11, 2), (21, 13, 4), (32, 32, 123)):
        add('maze%dx%d_s%d' % (w, h, k), seeded_maze_env(w, h, k), note='the demo script itself searches 15x15 random mazes (:18)')
    env = ref_env(custom_world_fp=os.path.join(LEVELS, 'test_env.txt'))
    for st in env.starting_states:
        add('test_env_start%d' % st, env, start=st)
    add('maze_21x21_level', ref_env(custom_world_fp=os.path.join(LEVELS, 'maze_21x21.txt')))
    add('maze_101x101_level', ref_env(custom_world_fp=os.path.join(LEVELS, 'maze_101x101.txt')))
    return cases


# ----------------------------------------------------------------------------- G2
# ----------------------------------------------------------------------------- G11
def capture_arrows():
    """Policy-arrow geometry and tile kinds of the reference's pyglet viewer, without a window.

    core/envs/rendering.py imports pyglet / OpenGL at module level and Viewer.__init__ opens a window, so neither the module nor
    the class can be used headless.  What is lifted out of the PARSED module (ast; nothing of it is imported or copied):
      * the method definitions Viewer.get_x_y_pix_location (:153-157) and Viewer.render_policy_arrows (:159-212), compiled
        unchanged into a class whose instances get the attributes __init__ would have computed (tile_dim = width of the ground
        texture + padding 1, :74-75; num_extra_tiles 4, :89; pix_grid_height, :117; x_distance_to_move set to 0 -- a pure
        translation) and whose free names `Line` / `FilledPolygon` record their arguments instead of drawing;
      * the `for i, (x, y) in enumerate(self.env.world)` statement of Viewer.__init__ (:119-133) that decides which texture
        every cell gets (goal, else lava, else wall, else ground), run with a `pyglet.sprite.Sprite` that records its image.
    Recorded per case: the policy, the geoms in the order the reference adds them (arrow head, then shaft, per state and action;
    coordinates relative to the state's tile origin, y up as in GL), and the tile kind of every state."""
    import ast
    from PIL import Image
    path = os.path.join(REF, 'core', 'envs', 'rendering.py')
    tree = ast.parse(open(path).read(), path)
    viewer = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == 'Viewer')
    methods = {n.name: n for n in viewer.body if isinstance(n, ast.FunctionDef)}
    lifted = ast.ClassDef(name='LiftedViewer', bases=[], keywords=[], decorator_list=[],
                          body=[methods['get_x_y_pix_location'], methods['render_policy_arrows']])
    tile_loop = next(n for n in ast.walk(methods['__init__']) if isinstance(n, ast.For)
                     and isinstance(n.iter, ast.Call) and getattr(n.iter.func, 'id', '') == 'enumerate')
    wrapper = ast.parse('def place_tiles(self, background):\n    pass\n').body[0]
    wrapper.body = [tile_loop]
    module = ast.Module(body=[lifted, wrapper], type_ignores=[])
    code = compile(ast.fix_missing_locations(module), path, 'exec')

    class Line(object):
        def __init__(self, start, end):
            self.kind, self.start, self.end = 'line', start, end

    class FilledPolygon(object):
        def __init__(self, v):
            self.kind, self.v = 'polygon', v

    class Sprite(object):
        def __init__(self, img, x=0, y=0, batch=None, group=None):
            self.img, self.x, self.y = img, x, y

    class Namespace(object):
        pass
    pyglet_stub = Namespace()
    pyglet_stub.sprite = Namespace()
    pyglet_stub.sprite.Sprite = Sprite
    ns = dict(np=np, Line=Line, FilledPolygon=FilledPolygon, pyglet=pyglet_stub)
    exec(code, ns)
    ground_width = Image.open(os.path.join(REF, 'core', 'resources', 'wbs_texture_05_resized.jpg')).size[0]
    padding = 1

    def viewer_for(env):
        v = ns['LiftedViewer']()
        v.env = env
        v.padding = padding
        v.tile_dim = ground_width + padding                  # :74-75
        v.num_extra_tiles = 4                                # :89
        v.x_distance_to_move = 0                             # (:106 centres the grid in the window: a translation)
        v.pix_grid_height = env.y_max * v.tile_dim + (v.num_extra_tiles // 2) * v.tile_dim   # :117
        v.geoms = []
        v.add_geom = lambda geom: v.geoms.append(geom)  # (:232-233; render_policy_arrows rebinds self.geoms first, :161)
        for kind in ('terminal_goal', 'terminal_lava', 'wall', 'ground'):
            setattr(v, kind + '_sprites', [])
            setattr(v, kind + '_img', kind)
        v.batch = None
        return v

    def case(name, env, policy, note=''):
        v = viewer_for(env)
        S = env.world.size
        origin = {}
        for s_, (x, y) in enumerate(env.world):
            origin[s_] = v.get_x_y_pix_location(x, y)
        centre_to_state = {(ox + v.tile_dim // 2, oy + v.tile_dim // 2): s_ for s_, (ox, oy) in origin.items()}
        v.render_policy_arrows(np.asarray(policy, dtype=np.float64))
        geoms, pending = [], None
        for g in v.geoms:  # the reference adds the head, then the shaft of the same arrow
            if g.kind == 'polygon':
                pending = g
                continue
            s_ = centre_to_state[tuple(int(c) for c in g.start)]
            ox, oy = origin[s_]
            rel = lambda pt: [int(pt[0]) - ox, int(pt[1]) - oy]  # noqa: E731
            geoms.append(dict(state=s_, head=[rel(p_) for p_ in pending.v], start=rel(g.start), end=rel(g.end)))
            pending = None
        ns['place_tiles'](v, None)
        kinds = [None] * S
        for kind in ('terminal_goal', 'terminal_lava', 'wall', 'ground'):
            for sp in getattr(v, kind + '_sprites'):
                s_ = next(k for k, o in origin.items() if o == (sp.x, sp.y))
                assert kinds[s_] is None
                kinds[s_] = {'terminal_goal': 'goal', 'terminal_lava': 'lava', 'wall': 'wall', 'ground': 'ground'}[kind]
        assert None not in kinds
        out = spec_of(env)
        out.update(name=name, note=note, policy=[[float(p_) for p_ in row] for row in np.asarray(policy, dtype=np.float64)],
                   geoms=geoms, tiles=kinds)
        print('arrows', name, 'S', S, 'arrows', len(geoms), 'tiles', {k: kinds.count(k) for k in sorted(set(kinds))})
        return out

    rs = np.random.RandomState(52)
    cases = []
    # a 6 x 4 grid with every overlap the texture rule has to decide: goal+lava (7), goal+wall (13), lava+wall (9), plain ones
    quirk = ref_env(grid_shape=(6, 4), goal_states=[23, 7, 13], lava_states=[7, 9, 16], walls=[2, 3, 14, 13, 9])
    S = quirk.world.size
    pi = rs.dirichlet(np.ones(4) * 0.6, S)
    pi[0] = [1, 0, 0, 0]
    pi[1] = [0, 1, 0, 0]
    pi[4] = [0, 0, 0, 1]
    pi[5] = [0.25, 0.25, 0.25, 0.25]
    pi[6] = [0.5, 0.5, 0, 0]
    pi[8] = [0.099, 0.101, 0.4, 0.4]            # just below / above the 0.1 cut
    pi[10] = [0.1, 0.1, 0.1, 0.7]               # p == 0.1 exactly is drawn (the test is `< 0.1`)
    pi[11] = [0.125, 0.375, 0.3, 0.2]           # round half to even: 2.5 -> 2, 7.5 -> 8
    pi[12] = [0.175, 0.225, 0.275, 0.325]       # 3.5, 4.5, 5.5, 6.5 as far as float64 has them
    pi[15] = [0.025, 0.975, 0.0, 0.0]
    pi[17] = [0.0, 0.0, 1.0, 0.0]
    cases.append(case('quirk6x4', quirk, pi, 'overlapping goal / lava / wall cells; the 0.1 cut; round-half-even lengths; one-hot and uniform rows'))
    default = ref_env()
    cases.append(case('default4x4_uniform', default, np.ones((16, 4)) / 4, 'the policy examples/griduniverse_alg_examples.py starts from'))
    with quiet():
        level = GridUniverseEnv(custom_world_fp=os.path.join(LEVELS, 'test_env.txt'))
    S = level.world.size
    cases.append(case('test_env_3x8_dirichlet', level, rs.dirichlet(np.ones(4), S), 'non-square level (two starts, lava, walls)'))
    maze = seeded_maze_env(11, 11, 3)
    S = maze.world.size
    greedy = np.zeros((S, 4))
    greedy[np.arange(S), rs.randint(0, 4, S)] = 1.0
    cases.append(case('maze11_onehot', maze, greedy, 'a deterministic policy on a generator maze'))
    return dict(tile_dim=ground_width + padding, ground_texture_width=ground_width, padding=padding, arrow_base_length_full_prob=20,
                arrow_width=5, arrow_height=5, cases=cases,
                source='core/envs/rendering.py: Viewer.get_x_y_pix_location, Viewer.render_policy_arrows and the tile loop of '
                       'Viewer.__init__, lifted with ast and executed against the real reference env; coordinates are GL pixels '
                       'relative to the tile origin (bottom-left corner of the tile, y up)')


def capture_trail():
    """The agent trail of the reference's viewer (core/envs/rendering.py:287-311), without a window: the statements `a = 0.3`,
    `discount = 0.96`, `padding = 10` and the `for i, (x, y) in enumerate(self.env.last_n_states[::-1])` loop are lifted out of the
    PARSED Viewer.render (ast; nothing is imported or copied) into a function whose free names glColor4f / glVertex2i record their
    arguments, next to the lifted Viewer.get_x_y_pix_location.  It runs against the real reference env, whose own step() / reset()
    keep `last_n_states` (env:92-93, 182-184, 190).  Recorded per case: the actions (-1 = reset()), the env's final state, its
    last_n_states as states, and the quads in the order the reference emits them: (state of the tile, alpha)."""
    import ast
    from PIL import Image
    path = os.path.join(REF, 'core', 'envs', 'rendering.py')
    tree = ast.parse(open(path).read(), path)
    viewer = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == 'Viewer')
    methods = {n.name: n for n in viewer.body if isinstance(n, ast.FunctionDef)}

    def is_trail_loop(n):
        return (isinstance(n, ast.For) and isinstance(n.iter, ast.Call) and getattr(n.iter.func, 'id', '') == 'enumerate'
                and 'last_n_states' in ast.dump(n.iter))
    host = next(m for m in methods.values() if any(is_trail_loop(n) for n in ast.walk(m)))
    loop = next(n for n in ast.walk(host) if is_trail_loop(n))
    setup = [n for n in host.body if isinstance(n, ast.Assign) and len(n.targets) == 1 and getattr(n.targets[0], 'id', '') in ('a', 'discount', 'padding')]
    assert [n.targets[0].id for n in setup] == ['a', 'discount', 'padding'], 'the trail loop\'s constants moved'
    wrapper = ast.parse('def draw_trail(self):\n    pass\n').body[0]
    wrapper.body = setup + [loop]
    lifted = ast.ClassDef(name='LiftedViewer', bases=[], keywords=[], decorator_list=[], body=[methods['get_x_y_pix_location'], wrapper])
    code = compile(ast.fix_missing_locations(ast.Module(body=[lifted], type_ignores=[])), path, 'exec')
    calls = []
    ns = dict(np=np, glColor4f=lambda *c: calls.append(('colour',) + tuple(c)), glVertex2i=lambda *v: calls.append(('vertex',) + tuple(int(k) for k in v)))
    exec(code, ns)
    ground_width = Image.open(os.path.join(REF, 'core', 'resources', 'wbs_texture_05_resized.jpg')).size[0]

    def case(name, env, actions, note=''):
        with quiet():
            start_state = int(env.reset())  # (a multi-start level draws it from the global RNG: recorded, forced in the tests)
            for a_ in actions:
                if a_ < 0:
                    assert len(env.starting_states) == 1
                    env.reset()
                else:
                    env.step(int(a_))
        v = ns['LiftedViewer']()
        v.env = env
        v.tile_dim = ground_width + 1
        v.num_extra_tiles = 4
        v.x_distance_to_move = 0
        v.pix_grid_height = env.y_max * v.tile_dim + (v.num_extra_tiles // 2) * v.tile_dim
        origin = {v.get_x_y_pix_location(x, y): s_ for s_, (x, y) in enumerate(env.world)}
        del calls[:]
        v.draw_trail()
        quads, corner_colours = [], None
        assert len(calls) % 8 == 0
        for k in range(0, len(calls), 8):
            group = calls[k:k + 8]
            assert [g[0] for g in group] == ['colour', 'vertex'] * 4
            colours = [list(g[1:4]) for g in group[0::2]]
            alphas = {g[4] for g in group[0::2]}
            verts = [g[1:] for g in group[1::2]]
            assert len(alphas) == 1
            ox, oy = verts[0]
            assert verts == [(ox, oy), (ox + v.tile_dim, oy), (ox + v.tile_dim, oy + v.tile_dim), (ox, oy + v.tile_dim)]
            assert corner_colours in (None, colours)
            corner_colours = colours
            quads.append([origin[(ox, oy)], float(alphas.pop())])
        world = {(int(x), int(y)): s_ for s_, (x, y) in enumerate(env.world)}
        out = spec_of(env)
        out.update(name=name, note=note, actions=[int(a_) for a_ in actions], start_state=start_state, current_state=int(env.current_state),
                   last_n_states=[world[(int(x), int(y))] for x, y in env.last_n_states], quads=quads, corner_colours=corner_colours)
        print('trail', name, 'steps', len(actions), 'kept', len(env.last_n_states), 'quads', len(quads))
        return out

    rs = np.random.RandomState(77)
    cases = [case('default4x4_walk', ref_env(), [1, 1, 2, 3, 0, 1, 2, 2, 3, 3, 0, 1], 'revisits cells, ends next to where it has been'),
             case('default4x4_bumping', ref_env(), [3, 3, 0, 0, 1, 3, 1, 3], 'steps into the border: the current cell is entered several times and always skipped'),
             case('lava4x4_reset', ref_env(lava_states=[5]), [1, 2, -1, 2, 2, 1, 1], 'walks into lava, the harness resets: the trail starts over'),
             case('open8x8_long', ref_env(grid_shape=(8, 8), goal_states=[63], walls=[27, 28]), [int(a_) for a_ in rs.randint(0, 4, 700) if True][:560],
                  'more than 500 steps without reaching a terminal state is not guaranteed: see last_n_states for what was kept')]
    with quiet():
        level = GridUniverseEnv(custom_world_fp=os.path.join(LEVELS, 'test_env.txt'))
    cases.append(case('test_env_3x8', level, [2, 2, 1, 2, 3, 2, 2, 0, 0, 1], 'non-square level'))
    return dict(tile_dim=ground_width + 1, alpha0=0.3, discount=0.96, cases=cases,
                source='core/envs/rendering.py: the trail loop of Viewer.render and Viewer.get_x_y_pix_location, lifted with ast and executed '
                       'against the real reference env after its own step() / reset() calls; quads = [state of the tile, alpha] in drawing order; '
                       'corner_colours = the glColor4f arguments (r, g, b) at the quad\'s vertices (x0, y0), (x0 + tile, y0), (x0 + tile, y0 + tile), '
                       '(x0, y0 + tile), GL coordinates (y up); GL clamps 0xFF to 1.0')


def capture_trajectories():
    digests = {}
    # C1: run_default_griduniverse() shape -- 1 env, 1000 random steps, reset on done
    env = ref_env()
    acts = np.random.RandomState(0).randint(0, 4, size=(1000, 1)).astype(np.int32)
    save_traj('c1_default4x4', env, 11, 1, 1000, True, actions=acts, note='actions = RandomState(0).randint(0,4,1000)')
    # C2: default 8x8
    env = ref_env(grid_shape=(8, 8))
    save_traj('c2_open8x8', env, 2, 64, 256, True)
    save_traj('c2_open8x8_absorbing', env, 3, 64, 256, False)
    save_digest(digests, 'c2_open8x8_4096x1000', env, 2, 4096, 1000, True)
    # C3: 32x32 generator maze, seed 123
    env = seeded_maze_env(32, 32, 123)
    save_traj('c3_maze32', env, 123, 64, 512, True)
    save_digest(digests, 'c3_maze32_4096x1000', env, 123, 4096, 1000, True)
    # C4: 32x32 open grid with the lava column
    env = ref_env(grid_shape=(32, 32), lava_states=lava_column_32())
    save_traj('c4_lava32', env, 4, 64, 256, True)
    save_traj('c4_lava32_absorbing', env, 4, 64, 256, False)
    save_traj('c4_lava32_shard1', env, 4, 64, 256, True, env_id0=32768, note='env ids of the second of 8 shards of 262144')
    save_digest(digests, 'c4_lava32_4096x1000', env, 4, 4096, 1000, True)
    # C5: 64x64 generator maze, seed 5
    env = seeded_maze_env(64, 64, 5)
    save_traj('c5_maze64', env, 5, 64, 512, True)
    # multi-start level (2 starts, lava, walls), short episodes
    env = ref_env(custom_world_fp=os.path.join(LEVELS, 'test_env.txt'))
    save_traj('multistart_test_env', env, 77, 64, 192, True)
    # non-square, several goals / lava / walls / starts incl. a terminal start and a wall start
    rs = np.random.RandomState(9)
    W, H = 25, 30
    cells = rs.permutation(W * H)
    env = ref_env(grid_shape=(W, H), initial_state=[int(c) for c in cells[:5]] + [int(cells[40])] + [int(cells[100])],
                  goal_states=[int(c) for c in cells[30:41]], lava_states=[int(c) for c in cells[38:60]],
                  walls=[int(c) for c in cells[100:260]])
    save_traj('rect25x30_busy', env, 99, 64, 384, True)
    # wider than one 32-bit row word
    W, H = 40, 12
    cells = rs.permutation(W * H)
    env = ref_env(grid_shape=(W, H), initial_state=[int(cells[0]), int(cells[1])], goal_states=[int(c) for c in cells[2:6]],
                  lava_states=[int(c) for c in cells[6:20]], walls=[int(c) for c in cells[20:120]])
    save_traj('wide40x12', env, 40, 64, 256, True)
    # the 101x101 level: long corridors, 4 words per row
    env = ref_env(custom_world_fp=os.path.join(LEVELS, 'maze_101x101.txt'))
    save_traj('maze101', env, 101, 16, 2048, True)
    # 1x1 and 1xN / Nx1 degenerate grids
    env = ref_env(grid_shape=(1, 1))
    save_traj('grid1x1', env, 1, 4, 8, True)
    env = ref_env(grid_shape=(1, 9), walls=[4])
    save_traj('grid1x9', env, 1, 8, 64, True)
    env = ref_env(grid_shape=(9, 1), lava_states=[6])
    save_traj('grid9x1', env, 1, 8, 64, True)
    return digests


def capture_big_digest():
    """BASELINE config 3 at FULL size straight from the reference: 65 536 envs x 1000 steps of its step() on the
    32x32 generator maze (seed 123) = 65.5 M reference steps (~2.5 min), kept as sha256 digests."""
    path = os.path.join(OUT, 'digests.json')
    store = json.load(open(path)) if os.path.exists(path) else {}
    env = seeded_maze_env(32, 32, 123)
    save_digest(store, 'c3_maze32_65536x1000', env, 123, 65536, 1000, True)
    json.dump(store, open(path, 'w'), indent=1)
    # BASELINE config 4 at its full batch size: 262 144 envs on the lava grid, 250 steps (another 65.5 M steps)
    env = ref_env(grid_shape=(32, 32), lava_states=lava_column_32())
    save_digest(store, 'c4_lava32_262144x250', env, 4, 262144, 250, True)
    json.dump(store, open(path, 'w'), indent=1)


def capture_stream_digests():
    """Caller-supplied action streams at config sizes (`RandomState(seed).randint(0, 4, (T, N))`), stepped by the
    reference itself and kept as sha256 digests: what the STREAM policy (gu_upload_actions + gu_rollout) must reproduce.
    The stream seed is part of the fixture; T = 1000 is not a multiple of the 16-action word of the device's packed form."""
    path = os.path.join(OUT, 'digests.json')
    store = json.load(open(path)) if os.path.exists(path) else {}

    def one(name, env, seed, stream_seed, N, T, auto_reset):
        env_ids = list(range(N))
        actions = np.random.RandomState(stream_seed).randint(0, 4, size=(T, N)).astype(np.int32)
        obs, rew, don, _ = rollout(env, actions, seed, env_ids, auto_reset)
        store[name] = dict(spec_of(env), seed=int(seed), N=N, T=T, auto_reset=bool(auto_reset), stream_seed=int(stream_seed),
                           actions='numpy.random.RandomState(stream_seed).randint(0, 4, size=(T, N)).astype(int32)',
                           sha256=digest(obs, rew, don), sum_reward=int(rew.sum()), sum_done=int(don.sum()),
                           sha256_final_obs=hashlib.sha256(obs[-1].astype('<i4').tobytes()).hexdigest())
        print('digest', name, store[name]['sha256'][:16])

    one('c2_open8x8_stream_4096x1000', ref_env(grid_shape=(8, 8)), 2, 1002, 4096, 1000, True)
    one('c3_maze32_stream_4096x1000', seeded_maze_env(32, 32, 123), 123, 1003, 4096, 1000, True)
    one('c4_lava32_stream_absorbing_4096x1000', ref_env(grid_shape=(32, 32), lava_states=lava_column_32()), 4, 1004, 4096, 1000, False)
    json.dump(store, open(path, 'w'), indent=1)


def reseed():
    """Both global RNG streams back to a fixed point: every capture starts from here, so that a capture run alone writes the
    same fixture as a full run (round 4: capture_trail drew a multi-start level's start cell from whatever the captures before
    it had left of the stdlib stream)."""
    random.seed(0)
    np.random.seed(0)


def same_npz(a, b):
    """Two .npz files hold the same arrays: names, dtypes, shapes and raw bytes."""
    with np.load(a, allow_pickle=False) as x, np.load(b, allow_pickle=False) as y:
        if sorted(x.files) != sorted(y.files):
            return 'array names differ: %s / %s' % (sorted(x.files), sorted(y.files))
        for k in x.files:
            u, v = x[k], y[k]
            if u.dtype != v.dtype or u.shape != v.shape or u.tobytes() != v.tobytes():
                return 'array %r differs' % k
    return None


def check(made, committed):
    """Compare everything under `made` (a fresh regeneration) with the committed fixtures by CONTENT: .json by equality of the
    parsed data, .npz array by array, byte-wise.  digests.json is compared entry by entry -- a regeneration without `big` does not
    hold the two full-size digests.  Returns the list of differences."""
    bad = []
    for name in sorted(os.listdir(made)):
        new, old = os.path.join(made, name), os.path.join(committed, name)
        if not os.path.exists(old):
            bad.append('%s: not among the committed fixtures' % name)
        elif name.endswith('.json'):
            x, y = json.load(open(new)), json.load(open(old))
            if name == 'digests.json':
                bad += ['digests.json[%s] differs' % k for k in x if x[k] != y.get(k)]
            elif x != y:
                keys = [k for k in x if isinstance(x, dict) and isinstance(y, dict) and x[k] != y.get(k)]
                bad.append('%s differs%s' % (name, ' in ' + ', '.join(map(str, keys[:8])) if keys else ''))
        elif name.endswith('.npz'):
            why = same_npz(new, old)
            if why:
                bad.append('%s: %s' % (name, why))
    return bad


def main():
    """make_golden.py [what ...]          regenerate those fixtures in place (default: all but `big` and `stream`)
    make_golden.py --check [what ...]  regenerate into a temporary directory and compare with the committed fixtures by content
                                       (default: all but `big`; exit code 1 and a list when anything differs)"""
    global OUT
    argv = [a for a in sys.argv[1:] if a != '--check']
    checking = len(argv) != len(sys.argv) - 1
    sys.argv[1:] = argv  # (capture_dp reads its dp=<name> filters from there)
    if checking:
        import tempfile
        OUT = tempfile.mkdtemp(prefix='gu_golden_check_')
    os.makedirs(OUT, exist_ok=True)
    everything = {'kat', 'err', 'render', 'maze', 'dp', 'traj', 'mc', 'mcnp', 'driver', 'bfs', 'arrows', 'trail'}  # plus 'big' (slow) and 'stream' on request
    what = {a.split('=')[0] for a in argv} or (everything | {'stream'} if checking else everything)
    if 'kat' in what:
        reseed()
        json.dump(capture_kats(), open(os.path.join(OUT, 'kat.json'), 'w'), indent=1)
    if 'err' in what:
        reseed()
        json.dump(capture_errors(), open(os.path.join(OUT, 'errors.json'), 'w'), indent=1)
    if 'render' in what:
        reseed()
        json.dump(capture_render_and_quirks(), open(os.path.join(OUT, 'render_quirks.json'), 'w'), indent=1)
    if 'maze' in what:
        reseed()
        mazes, levels = capture_mazes_and_levels()
        json.dump(mazes, open(os.path.join(OUT, 'mazes.json'), 'w'), indent=1)
        json.dump(levels, open(os.path.join(OUT, 'levels.json'), 'w'), indent=1)
    if 'dp' in what:
        reseed()
        capture_dp()
    if 'mc' in what:
        reseed()
        capture_mc()
    if 'mcnp' in what:
        reseed()
        capture_mc_numpy_rng()
    if 'driver' in what:
        reseed()
        capture_driver()
    if 'bfs' in what:
        reseed()
        json.dump(capture_bfs(), open(os.path.join(OUT, 'bfs.json'), 'w'), indent=1)
    if 'arrows' in what:
        reseed()
        json.dump(capture_arrows(), open(os.path.join(OUT, 'arrows.json'), 'w'))
    if 'trail' in what:
        reseed()
        json.dump(capture_trail(), open(os.path.join(OUT, 'trail.json'), 'w'))
    if 'big' in what:
        reseed()
        capture_big_digest()
    if 'stream' in what:
        reseed()
        capture_stream_digests()
    if 'traj' in what:
        reseed()
        path = os.path.join(OUT, 'digests.json')
        store = json.load(open(path)) if os.path.exists(path) else {}
        store.update(capture_trajectories())  # keeps the separately captured full-size digest ('big')
        json.dump(store, open(path, 'w'), indent=1)
    assert not any('__pycache__' in d for d, _, _ in os.walk(REF)), 'bytecode was written into the reference'
    if checking:
        import shutil
        bad = check(OUT, HERE)
        n = len(os.listdir(OUT))
        shutil.rmtree(OUT, ignore_errors=True)
        if bad:
            print('make_golden --check: %d of %d regenerated fixtures DIFFER from the committed ones:' % (len(bad), n))
            for b in bad:
                print('  ' + b)
            sys.exit(1)
        print('make_golden --check: %d fixtures regenerated from the reference, all identical to the committed ones' % n)


if __name__ == '__main__':
    main()
