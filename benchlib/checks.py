"""Checks bench.py runs on what the GPU produced.  oracle/ is used here as the CHECKER, never as the thing measured."""
import hashlib
import json
import os

import numpy as np

from .workloads import REFERENCE_DIGEST

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sha256_triplet(traj):
    """sha256 over obs | reward | done, each int32 little-endian [T, N] -- tests/golden/make_golden.py: digest()."""
    h = hashlib.sha256()
    for k in ('obs', 'reward', 'done'):
        h.update(np.ascontiguousarray(traj[k], dtype='<i4').tobytes())
    return h.hexdigest()


def reference_digest(workload, template, seed, N, T, env_id0):
    """The sha256 the REFERENCE's own step() produced for this very run, if this run is the one that was captured
    (tests/golden/digests.json: same grid, seed, batch, length, env ids 0..N-1, from reset, auto-reset)."""
    try:
        entry = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'digests.json')))[REFERENCE_DIGEST[workload]]
    except (OSError, KeyError, ValueError):
        return None
    same = (env_id0 == 0 and entry['N'] == N and entry['T'] == T and entry['seed'] == seed and entry['auto_reset']
            and (entry['W'], entry['H']) == (template.x_max, template.y_max)
            and entry['starts'] == [int(s) for s in template.starting_states]
            and entry['goals'] == [int(s) for s in template.goal_states]
            and entry['lava'] == [int(s) for s in template.lava_states]
            and entry['walls'] == [int(s) for s in template.wall_indices])
    return entry['sha256'] if same else None


def cpu_baseline_check_prefix(template, seed, env_id0, traj, n_check=4096):
    """First `n_check` envs of a from-reset launch against the C oracle (used when no reference digest covers the run)."""
    from oracle import c_oracle as C
    T, N = traj['obs'].shape
    n = min(n_check, N)
    grid = C.Grid.from_env(template)
    st = C.State(n, env_id0)
    C.reset(grid, seed, st)
    want = C.rollout(grid, seed, st, T, True)
    return all(np.array_equal(traj[k][:, :n], want[k]) for k in ('obs', 'reward', 'done'))


def cpu_baseline_check_final_state(template, seed, env_id0, N, total_steps, state, budget_steps=4.0e8):
    """After ALL launches of the run (checked one, warm-up, probe, timed, instrumented): the final pos / done / episode /
    step count of a sample of envs -- the first and the last ones of the shard -- against the C oracle advanced by the same
    number of steps.  The whole batch would take the scalar oracle about an hour; the sample is sized to seconds."""
    from oracle import c_oracle as C
    per_block = int(max(1, min(N // 2, budget_steps // max(1, total_steps) // 2)))
    grid = C.Grid.from_env(template)
    ok, checked = True, 0
    for lo in sorted({0, N - per_block}):
        st = C.State(per_block, env_id0 + lo)
        C.reset(grid, seed, st)
        C.rollout(grid, seed, st, total_steps, True, trajectory=False)
        sl = slice(lo, lo + per_block)
        ok = ok and all(np.array_equal(state[k][sl], getattr(st, k)) for k in ('pos', 'done', 'episode', 'tcount'))
        checked += per_block
    return dict(equal=bool(ok), envs_checked=checked, env_steps_each=int(total_steps),
                fields='pos, done, episode, tcount', checker='oracle/gu_oracle.c')


def cpu_baseline_check_stats(template, seed, env_id0, T, ret, episodes, n_check=2048):
    """Per-env return and episode count of a from-reset, statistics-only launch against the C oracle (first `n_check` envs)."""
    from oracle import c_oracle as C
    n = min(n_check, ret.size)
    grid = C.Grid.from_env(template)
    st = C.State(n, env_id0)
    C.reset(grid, seed, st)
    want = C.rollout(grid, seed, st, T, True, trajectory=False, stats=True)
    return bool(np.array_equal(ret[:n], want['ret']) and np.array_equal(episodes[:n], want['episodes']))


def cpu_baseline_check_c5(template, seed, gamma, rounds, v, pi, state, rewards):
    """Config 5 against the C oracle: `rounds` x { V1 + V2 sweep (value_iteration_step, itself pinned to the reference's
    value-iteration trace by tests/test_oracle_c.py); every env steps greedily on the updated policy (np.argmax of its row,
    examples/griduniverse_alg_examples.py:76), lazy reset first } from reset with zero values and the uniform policy --
    tables as raw bytes, every env's position / done flag / episode count and last reward."""
    from oracle import c_oracle as C
    grid = C.Grid.from_env(template)
    S, N = template.world.size, state['pos'].size
    st = C.State(N)
    C.reset(grid, seed, st)
    v_o, pi_o = np.zeros(S), np.ones((S, 4)) / 4
    want = None
    for _ in range(rounds):
        v_o, pi_o, _ = C.value_iteration_step(grid, gamma, pi_o, v_o)
        acts = np.argmax(pi_o, axis=1).astype(np.int32)
        if st.done.any():
            C.reset(grid, seed, st, mask=st.done.astype(bool))
        want = C.rollout(grid, seed, st, 1, False, actions=acts[st.pos][None, :])
    return bool(v.tobytes() == v_o.tobytes() and pi.tobytes() == pi_o.tobytes() and np.array_equal(state['pos'], st.pos)
                and np.array_equal(state['done'], st.done) and np.array_equal(state['episode'], st.episode)
                and np.array_equal(rewards, want['reward'][0]))



def cpu_baseline_check_distinct_grids(got, first, N, G, T, seed, maze_seed):
    """Config 3 on G device-generated 32x32 mazes: the envs of six grids (up to eight each), reset state and the whole first launch,
    against the C oracle stepping on ITS restatement of the same mazes (oracle/gu_oracle.c: gu_oracle_generate_maze)."""
    from oracle import c_oracle as C
    group, ok = N // G, True
    for g in sorted({0, 1, G // 3, G // 2, G - 2, G - 1}):
        wall, start, goal = C.generate_maze(maze_seed, g, 32, 32)
        grid = C.Grid.from_lists(32, 32, walls=np.flatnonzero(wall).tolist(), goals=[int(goal)], starts=[int(start)])
        n = min(group, 8)
        st = C.State(n, g * group)
        ok = ok and bool(np.array_equal(C.reset(grid, seed, st), first[g * group:g * group + n]))
        want = C.rollout(grid, seed, st, T, True)
        ok = ok and all(np.array_equal(got[k][:, g * group:g * group + n], want[k]) for k in ('obs', 'reward', 'done'))
    return bool(ok)
