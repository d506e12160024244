#!/usr/bin/env python3
"""Throughput of every BASELINE.json config and every call path on one MI355X (not the driver's bench;
numbers quoted in DESIGN.md come from here).  Usage: python tools/bench_configs.py [out.json]"""
import json
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import griduniverse_amd as gua  # noqa: E402


def maze(w, h, k):
    random.seed(k)
    np.random.seed(k)
    return gua.GridUniverseEnv(grid_shape=(w, h), random_maze=True)


def timed(eng, fn, reps):
    fn()
    eng.sync()
    t0 = time.perf_counter()
    eng.timer_begin()
    for _ in range(reps):
        fn()
    ms = eng.timer_end()
    eng.sync()
    return time.perf_counter() - t0, ms / 1e3


def paths(name, template, N, T=1000, reps=20):
    out = {'config': name, 'N': N}
    eng = gua.Engine(N, gua.GridSpec.from_env(template), seed=1)
    eng.reset()
    eng.reserve_trajectory(T)
    wall, dev = timed(eng, lambda: eng.rollout(T, 'uniform', True, True), reps)
    out['rollout_traj_steps_per_s'] = N * T * reps / wall
    out['rollout_traj_GBps_12B'] = 12 * N * T * reps / dev / 1e9
    wall, dev = timed(eng, lambda: eng.rollout(T, 'uniform', True, False, True), reps)
    out['rollout_stats_only_steps_per_s'] = N * T * reps / wall
    if template.world.size <= 65536:
        wall, dev = timed(eng, lambda: eng.rollout(T, 'uniform', True, 'packed'), reps)
        out['rollout_packed_traj_steps_per_s'] = N * T * reps / wall
        out['rollout_packed_traj_GBps_4B'] = 4 * N * T * reps / dev / 1e9
    wall, dev = timed(eng, lambda: eng.rollout(T, 'uniform', False, True), reps)
    out['rollout_traj_no_autoreset_steps_per_s'] = N * T * reps / wall
    # table-driven policies: first-argmax (greedy) and inverse-CDF sampling of a random stochastic policy
    S = template.world.size
    eng.vi_set(np.zeros(S), np.random.RandomState(1).dirichlet(np.ones(4), S))
    wall, dev = timed(eng, lambda: eng.rollout(T, 'sample', True, True), reps)
    out['rollout_sample_policy_traj_steps_per_s'] = N * T * reps / wall
    out['rollout_sample_policy_traj_GBps_12B'] = 12 * N * T * reps / dev / 1e9
    wall, dev = timed(eng, lambda: eng.rollout(T, 'greedy', True, True), reps)
    out['rollout_greedy_policy_traj_steps_per_s'] = N * T * reps / wall
    out['rollout_greedy_policy_traj_GBps_12B'] = 12 * N * T * reps / dev / 1e9
    # PCIe-inclusive: trajectory copied back to host numpy after every launch
    t0 = time.perf_counter()
    for _ in range(3):
        eng.rollout(T, 'uniform', True, True)
        eng.read_trajectory(0, T)
    out['rollout_traj_plus_D2H_steps_per_s'] = N * T * 3 / (time.perf_counter() - t0)
    eng.read_trajectory(0, T, pinned=True)
    t0 = time.perf_counter()
    for _ in range(3):
        eng.rollout(T, 'uniform', True, True)
        eng.read_trajectory(0, T, pinned=True)
    dt = time.perf_counter() - t0
    out['rollout_traj_plus_pinned_D2H_steps_per_s'] = N * T * 3 / dt
    out['pinned_D2H_GBps'] = 12 * N * T * 3 / dt / 1e9
    # device-resident action stream
    Ts = 256
    acts = np.random.RandomState(0).randint(0, 4, (Ts, N)).astype(np.int32)
    eng.upload_actions(acts)
    eng.reserve_trajectory(Ts)
    wall, dev = timed(eng, lambda: eng.rollout(Ts, 'stream', True, True), reps)
    out['rollout_stream_steps_per_s'] = N * Ts * reps / wall
    # (the stream kernel reads PACKED actions, two bits each: 12 B of rows + 0.25 B of actions per env-step)
    out['rollout_stream_GBps_12.25B'] = 12.25 * N * Ts * reps / dev / 1e9
    # one launch per env-step, actions already in HBM (eager launches vs one hipGraph replay)
    def eager():
        for t in range(Ts):
            eng.step_device(t, True)
    wall, dev = timed(eng, eager, 5)
    out['step_device_eager_steps_per_s'] = N * Ts * 5 / wall
    out['step_device_eager_us_per_launch'] = wall / (Ts * 5) * 1e6
    wall, dev = timed(eng, lambda: eng.step_graph(0, Ts, True), 5)
    out['step_graph_steps_per_s'] = N * Ts * 5 / wall
    out['step_graph_us_per_launch'] = wall / (Ts * 5) * 1e6
    out['step_graph_GBps_24B'] = 24 * N * Ts * 5 / dev / 1e9
    # host round trip per step (actions from numpy, results to numpy)
    n_host = 200
    t0 = time.perf_counter()
    for t in range(n_host):
        eng.step(acts[t % Ts], True)
    dt = time.perf_counter() - t0
    out['step_host_steps_per_s'] = N * n_host / dt
    out['step_host_us_per_call'] = dt / n_host * 1e6
    t0 = time.perf_counter()
    for t in range(n_host):
        eng.pinned_actions[:] = acts[t % Ts]
        eng.step_pinned(True)
    dt = time.perf_counter() - t0
    out['step_pinned_steps_per_s'] = N * n_host / dt
    out['step_pinned_us_per_call'] = dt / n_host * 1e6
    eng.close()
    return out


def config5(N=65536):
    env = maze(64, 64, 5)
    S = env.world.size
    eng = gua.Engine(N, gua.GridSpec.from_env(env), seed=5)
    eng.reset()
    eng.vi_set(np.zeros(S), np.ones((S, 4)) / 4)
    reps = 200
    wall, dev = timed(eng, lambda: eng.vi_sweep_step(1.0, True, want_delta=False), reps)
    out = {'config': 'c5 64x64 maze, fused V1+V2 sweep + greedy env step per launch', 'N': N,
           'launches_per_s': reps / wall, 'env_steps_per_s': N * reps / wall, 'us_per_launch_device': dev / reps * 1e6,
           'state_updates_per_s': S * reps / wall}
    # the same loop as ONE launch (workgroup cluster, grid barrier per round)
    eng.vi_set(np.zeros(S), np.ones((S, 4)) / 4)
    eng.vi_sweep_step_run(1.0, 50, True)
    rounds = 2000
    t0 = time.perf_counter()
    eng.vi_sweep_step_run(1.0, rounds, True)
    dt = time.perf_counter() - t0
    out['run_rounds_per_s (one launch for all rounds)'] = rounds / dt
    out['run_env_steps_per_s'] = N * rounds / dt
    out['run_us_per_round'] = dt / rounds * 1e6
    eng.vi_set(np.zeros(S), np.ones((S, 4)) / 4)
    t0 = time.perf_counter()
    eng.vi_sweep(1.0, 1000, True)
    eng.sync()
    dt = time.perf_counter() - t0
    out['vi_sweep_only_iters_per_s'] = 1000 / dt
    out['vi_sweep_only_state_updates_per_s'] = 1000 * S / dt
    eng.close()
    return out


def mc_bench(N=4096, T=1000):
    from griduniverse_amd.algorithms.monte_carlo import discount_table
    env = maze(8, 8, 1)
    S = env.world.size
    eng = gua.Engine(N, gua.GridSpec.from_env(env), seed=3)
    eng.vi_set(np.zeros(S), np.ones((S, 4)) / 4)
    first = eng.reset()
    eng.reserve_trajectory(T)
    t0 = time.perf_counter()
    eng.rollout(T, 'sample', False, True)
    eng.sync()
    t1 = time.perf_counter()
    pw, keep = discount_table(0.99, 1e-4, T)
    v, visits = eng.mc_evaluate(T, first, pw, keep, True, True, True, 0.001)
    t2 = time.perf_counter()
    traj = eng.read_trajectory(0, T)
    lengths = np.where(traj['done'].any(0), traj['done'].argmax(0) + 1, T)
    eng.close()
    return {'config': 'monte_carlo_evaluation, 8x8 maze, uniform policy, every-visit incremental mean', 'episodes': N,
            'max_steps': T, 'mean_episode_length': float(lengths.mean()), 'rollout_s': t1 - t0, 'evaluate_s': t2 - t1,
            'episodes_per_s': N / (t2 - t0)}


def maze_bench():
    out = {'config': 'on-device maze generation + rollout on per-group distinct 32x32 mazes'}
    N, T = 65536, 1000
    for n_grids in (65536, 1024, 64):
        eng = gua.Engine(N, gua.GridSpec(32, 32, [0], [1023], [], []), seed=1)
        t0 = time.perf_counter()
        eng.generate_mazes(n_grids, 32, 32, 7)
        dt = time.perf_counter() - t0
        eng.reset()
        eng.reserve_trajectory(T)
        wall, dev = timed(eng, lambda: eng.rollout(T, 'uniform', True, True), 5)
        out['G%d' % n_grids] = {'generate_s': dt, 'mazes_per_s': n_grids / dt, 'rollout_traj_steps_per_s': N * T * 5 / wall,
                                'rollout_traj_GBps_12B': 12 * N * T * 5 / dev / 1e9}
        eng.close()
    return out


def main():
    res = []
    res.append(paths('c2 open 8x8', gua.GridUniverseEnv(grid_shape=(8, 8)), 4096))
    res.append(paths('c3 maze 32x32 seed 123', maze(32, 32, 123), 65536))
    res.append(paths('c4 lava 32x32 (one of 8 shards)', gua.GridUniverseEnv(grid_shape=(32, 32), lava_states=[16 + 32 * r for r in range(24)]), 32768))
    res.append(paths('c3 grid at 1M envs', maze(32, 32, 123), 1 << 20, T=250, reps=10))
    res.append(config5())
    res.append(mc_bench())
    res.append(maze_bench())
    text = json.dumps(res, indent=1)
    print(text)
    if len(sys.argv) > 1:
        open(sys.argv[1], 'w').write(text)


if __name__ == '__main__':
    main()
