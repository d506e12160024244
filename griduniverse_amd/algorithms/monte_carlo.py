"""Monte-Carlo policy evaluation with the reference's names (core/algorithms/monte_carlo.py).

`run_episode` is the reference's scalar loop on the N = 1 facade.  `monte_carlo_evaluation` keeps
the reference's signature and arithmetic (float64, bit-exact given the same episodes) but generates
all `num_episodes` episodes in ONE fused rollout launch -- env e is episode e, actions sampled from
`policy[state]` by the per-env device RNG -- and reduces them on the device (csrc/gu_mc.hip).

`rng='numpy'` is the drop-in mode: the episodes are the ones the REFERENCE would draw -- start cells from the stdlib's
global `random`, actions from numpy's global stream, one `np.random.choice(4, p=policy[obs])` per step, episodes strictly one
after the other (monte_carlo.py:20, 46-52) -- so that `random.seed(k); np.random.seed(k); monte_carlo_evaluation(...)` returns
the reference's value function byte for byte and leaves both global streams where the reference leaves them.
"""
import bisect
import random  # noqa: F401  (the facade's reset() draws from it, like the reference's)

import numpy as np

from .utils import engine_of


def run_episode(policy, env, max_steps_per_episode=1000):
    """Reset, then sample-action / step until done or the step cap (monte_carlo.py:7-26).  Uses the
    numpy GLOBAL rng for np.random.choice, like the reference.  Returns (states, rewards, done)."""
    states_hist, rewards_hist = [], []
    observation = env.reset()
    states_hist.append(observation)
    done = False
    for _ in range(max_steps_per_episode):
        action = np.random.choice(policy[observation].size, p=policy[observation])
        observation, reward, done, _ = env.step(action)
        states_hist.append(observation)
        rewards_hist.append(reward)
        if done:
            break
    return states_hist, rewards_hist, done


def discount_table(discount_factor, threshold, length):
    """discount_factor ** i and the reference's truncation test (monte_carlo.py:69-70) for i < length,
    evaluated with Python's float pow exactly as the reference evaluates them."""
    pw = np.array([discount_factor ** i for i in range(length)], dtype=np.float64)
    return pw, pw > threshold


def _check_visited_policy_rows(policy, first, obs, done):
    """np.random.choice(4, p=policy[obs]) in the reference's run_episode raises ValueError for a row that is negative or does
    not sum to 1 -- but only when an episode actually draws an action in that state.  Called only when some row IS bad."""
    p = np.asarray(policy, dtype=np.float64)
    bad = (p < 0).any(axis=1) | (np.abs(p.sum(axis=1) - 1.0) > np.sqrt(np.finfo(np.float64).eps)) | ~np.isfinite(p).all(axis=1)
    T = obs.shape[0]
    ended = np.where(done.any(axis=0), done.argmax(axis=0) + 1, T)          # steps each episode took
    drew = np.arange(T)[:, None] < (ended[None, :] - 1)                       # obs[t] is a state an action was drawn in
    visited = np.union1d(np.asarray(first), obs[drew])
    hit = visited[bad[visited]]
    if hit.size:
        row = p[int(hit[0])]
        raise ValueError('probabilities are not non-negative' if (row < 0).any() else 'probabilities do not sum to 1')


def _choice_error(row):
    """The ValueError np.random.choice(4, p=row) raises for a row that is no distribution (numpy's legacy `choice`), or None."""
    row = np.asarray(row, dtype=np.float64)
    total = float(np.sum(row))
    if np.isnan(total):
        return 'probabilities contain NaN'
    if (row < 0).any():
        return 'probabilities are not non-negative'
    if abs(total - 1.0) > np.sqrt(np.finfo(np.float64).eps):
        return 'probabilities do not sum to 1'
    return None


def reference_rng_episodes(policy, env, num_episodes, max_steps_per_episode=1000):
    """The episodes the reference's `run_episode` would generate, drawn from the SAME global streams in the same order: per
    episode one `random.choice(starting_states)` (env.reset(), core/envs/griduniverse_env.py:189), then per step one uniform of
    numpy's global RandomState turned into an action the way `np.random.choice(4, p=policy[obs])` does it -- cdf = p.cumsum(),
    cdf /= cdf[-1], cdf.searchsorted(u, side='right') -- until done or the step cap (monte_carlo.py:15-25).  Transitions come
    from the (state, action) table the look-ahead kernel computed for the grid.  The uniforms of an episode are drawn in one
    block and the stream is then put back to exactly as many draws as the episode used.
    Returns (first_state int32[N], actions int32[T, N] zero-padded, lengths int32[N], the last episode's state list and
    whether it ended on a terminal step)."""
    nxt, _, don = env._transition_table(True)
    nxt_l, don_l = nxt.tolist(), don.tolist()
    p = np.asarray(policy, dtype=np.float64)
    if p.ndim != 2:
        raise IndexError('policy must be a table of one row of 4 action probabilities per state')
    wrong_width = p.shape[1] != 4  # np.random.choice(4, p=row): "'a' and 'p' must have same size", at the first draw
    with np.errstate(invalid='ignore', divide='ignore'):
        cdf = p.cumsum(axis=1)
        cdf /= cdf[:, -1:]
    cdf_l = cdf.tolist()
    bad = [_choice_error(row) for row in p] if not (np.isfinite(p).all() and (p >= 0).all()
                                                    and (np.abs(p.sum(axis=1) - 1.0) <= 1e-9).all()) else None
    cap = int(max_steps_per_episode)
    first, acts, lengths, states, done = [], [], [], [], False
    for _ in range(int(num_episodes)):
        s = env.reset()
        first.append(s)
        states = [s]
        before = np.random.get_state()
        u = np.random.random_sample(cap).tolist()
        row = []
        error = None
        done = False
        for t in range(cap):
            if s >= len(cdf_l):
                np.random.set_state(before)
                raise IndexError('index {} is out of bounds for axis 0 with size {}'.format(s, len(cdf_l)))  # policy[obs]
            if wrong_width:
                error = "'a' and 'p' must have same size"
                break
            if bad is not None and bad[s] is not None:
                error = bad[s]  # (np.random.choice validates p BEFORE it draws)
                break
            a = bisect.bisect_right(cdf_l[s], u[t])
            row.append(a)
            done = don_l[s][a]
            s = nxt_l[s][a]
            states.append(s)
            if done:
                break
        np.random.set_state(before)
        if row:
            np.random.random_sample(len(row))  # the stream has advanced by exactly the draws the episode made
        if error is not None:
            raise ValueError(error)
        acts.append(row)
        lengths.append(len(row))
    T = max(1, max(lengths) if lengths else 1)
    actions = np.zeros((T, len(acts)), np.int32)
    for e, row in enumerate(acts):
        actions[:len(row), e] = row
    return np.asarray(first, np.int32), actions, np.asarray(lengths, np.int32), states, bool(done)


def reference_rng_offsets(policy, env, eng, num_episodes, max_steps_per_episode=1000):
    """The reference's episodes WITHOUT walking them on the host (csrc/gu_mc_ref.hip): start cells from the stdlib's global stream
    (one env.reset() per episode -- a stream of its own, so all of them can be drawn first), uniforms of numpy's global stream drawn
    in bulk; the device walks the episode that would begin at EVERY offset of a block of uniforms (gu_mc_walk_lengths), the host
    follows the chain offset -> offset + length, one table look-up per episode, drawing further blocks while episodes are left; the
    global stream ends exactly where the reference leaves it.  Returns (first_state, offsets, lengths, uniforms, cdf), or None when
    the policy is not an (S, 4) table or a policy row is no distribution (np.random.choice raises for it the moment an episode draws from it: the host walk knows how)."""
    p = np.asarray(policy, dtype=np.float64)
    if p.shape != (env.world.size, 4):
        return None  # (the host walk indexes the rows it visits, and raises where the reference's policy[obs] would)
    if not (np.isfinite(p).all() and (p >= 0).all() and (np.abs(p.sum(axis=1) - 1.0) <= 1e-9).all()):
        return None
    cdf = p.cumsum(axis=1)
    cdf /= cdf[:, -1:]
    N, cap = int(num_episodes), int(max_steps_per_episode)
    first = np.array([env.reset() for _ in range(N)], np.int32)
    starts, sidx = np.unique(first, return_inverse=True)
    starts = starts.astype(np.int32)
    sidx = sidx.tolist()
    # the uniforms are drawn in pieces of at most 65 536, the stream's state remembered in front of each: putting the stream
    # back to "exactly `pos` draws made" then costs one set_state and at most 65 535 draws, whatever was drawn ahead
    pieces, marks, drawn = [], [], 0

    def draw(upto):
        nonlocal drawn
        while drawn < upto:
            n = min(65536, upto - drawn)
            marks.append((drawn, np.random.get_state()))
            pieces.append(np.random.random_sample(n))
            drawn += n

    U = np.empty(0, np.float64)
    offsets, lengths = np.zeros(N, np.int64), np.zeros(N, np.int32)
    pos, e = 0, 0
    while e < N:
        base = pos
        # offsets walked speculatively this time: a guess at first (64 uniforms per episode), then what the episodes so far say the
        # rest will need, and a quarter more (episodes that run into the step cap: 1000 uniforms each)
        guess = 64 * (N - e) if e == 0 else int((N - e) * (pos / e) * 1.25) + 4096
        block = int(min(max(16384, guess), (N - e) * cap, 1 << 20))
        if drawn < base + block + cap:
            draw(base + block + cap)
            U = np.concatenate(pieces) if len(pieces) > 1 else pieces[0]
        table = eng.mc_walk_lengths(U[base:base + block + cap], block, starts, cap, cdf)
        while e < N and pos - base < block:
            length = int(table[sidx[e], pos - base])
            offsets[e], lengths[e] = pos, length
            pos += length
            e += 1
    at, state = next((m for m in reversed(marks) if m[0] <= pos), (0, None)) if marks else (0, None)
    if state is not None:
        np.random.set_state(state)
        if pos > at:
            np.random.random_sample(pos - at)  # the stream has advanced by exactly the draws the episodes made
    return first, offsets, lengths, U[:pos + cap], cdf


def _states_of_episode(env, first_state, uniforms, cdf, cap):
    """The states one episode visits (the reset state first) and whether it ended on a terminal step: the host's copy of the
    device's walk, for the ONE episode whose end the caller's env instance is left at."""
    nxt, _, don = env._transition_table(True)
    s, states, done = int(first_state), [int(first_state)], False
    for t in range(min(cap, len(uniforms))):
        a = bisect.bisect_right(cdf[s].tolist(), float(uniforms[t]))
        done = bool(don[s][a])
        s = int(nxt[s][a])
        states.append(s)
        if done:
            break
    return states, done


def monte_carlo_evaluation(policy, env, every_visit=False, incremental_mean=True, stationary_env=True,
                           discount_factor=0.99, threshold=0.0001, alpha=0.001, num_episodes=100, *,
                           max_steps_per_episode=1000, seed=0, return_details=False, rng='device'):
    """Value function of `policy` from `num_episodes` sampled episodes (monte_carlo.py:29-99).

    Episodes are processed in index order, so the result equals the reference's sequential loop fed
    with the same episodes.  Extra keyword-only arguments: the step cap of each episode (run_episode's
    default 1000), the RNG seed of the batch, `return_details` to also get the visit counters and
    the raw trajectory, and `rng`: 'device' (default: every episode sampled by its own counter-RNG stream inside one fused
    launch) or 'numpy' (the reference's own draws from the global `random` / `np.random` streams: the returned array is then
    byte-identical to the reference's for the same seeds; `seed` is unused)."""
    from ..vec_env import VecGridUniverse
    if rng not in ('device', 'numpy'):
        raise ValueError("rng must be 'device' or 'numpy'")
    numpy_rng = rng == 'numpy'
    device = getattr(engine_of(env), 'device', 0)
    # the batch engine of the previous call on this env is kept (creating one costs a few ms of allocations, the
    # evaluation itself well under one): re-seeding restores exactly the state of a fresh engine
    key = (int(num_episodes), device)
    kept = getattr(env, '_episode_batch', None) if hasattr(env, '__dict__') else None
    if kept is not None and kept[0] == key:
        batch = kept[1]
        batch.engine.seed(seed)
    else:
        if kept is not None:
            kept[1].close()
            env._episode_batch = None
        batch = VecGridUniverse(num_episodes, template=env, seed=seed, auto_reset=False, device=device)
        if hasattr(env, '_drop_engine'):  # the facade closes it together with its own engine
            env._episode_batch = (key, batch)
    try:
        eng = batch.engine
        walked = reference_rng_offsets(policy, env, eng, num_episodes, max_steps_per_episode) if numpy_rng and hasattr(eng, 'mc_walk_lengths') else None
        if numpy_rng and walked is not None:
            # the reference's episodes, found and walked on the device (csrc/gu_mc_ref.hip), reduced by gu_mc_evaluate
            first, offsets, lengths_np, uniforms, cdf = walked
            T = max(1, int(lengths_np.max()) if lengths_np.size else 1)
            eng.reserve_trajectory(T)
            eng.mc_walk_episodes(uniforms, cdf, offsets, first, int(max_steps_per_episode), T)
            last = int(num_episodes) - 1
            last_states, last_done = _states_of_episode(env, first[last], uniforms[offsets[last]:offsets[last] + lengths_np[last]], cdf, int(max_steps_per_episode))
        elif numpy_rng:
            # (a policy row that is no distribution: the host walks, and raises where np.random.choice would)  The reference's
            # episodes replayed on the device: start cells installed, actions as a caller-supplied stream, no auto-reset (an env past
            # its terminal step is absorbing and its rows are not read), reduced by gu_mc_evaluate
            first_np, actions_np, lengths_np, last_states, last_done = reference_rng_episodes(policy, env, num_episodes, max_steps_per_episode)
            T = int(actions_np.shape[0])
            first = first_np
            eng.set_state(pos=first, done=np.zeros(first.size, np.int32))
            eng.upload_actions(actions_np)
            eng.reserve_trajectory(T)
            eng.rollout(T, 'stream', auto_reset=False, trajectory=True)
        else:
            eng.vi_set(np.zeros(env.world.size), policy)
            first = batch.reset()
            T = int(max_steps_per_episode)
            eng.reserve_trajectory(T)
            eng.rollout(T, 'sample', auto_reset=False, trajectory=True)
        pw, keep = discount_table(discount_factor, threshold, T)
        value, visits = eng.mc_evaluate(T, first, pw, keep, every_visit, incremental_mean, stationary_env, alpha)
        if numpy_rng and hasattr(env, '_drop_engine'):
            # the reference's env ends up where its last episode ended (run_episode steps the instance itself)
            env.reset_to_trail(last_states, last_done)
        rows = np.asarray(policy, dtype=np.float64)
        if not numpy_rng and not (np.isfinite(rows).all() and (rows >= 0).all()
                and (np.abs(rows.sum(axis=1) - 1.0) <= np.sqrt(np.finfo(np.float64).eps)).all()):
            traj = eng.read_trajectory(0, T)  # rare path: some row is no distribution -- did an episode draw from it?
            _check_visited_policy_rows(rows, first, traj['obs'], traj['done'])
        if return_details:
            return value, dict(total_visit_counter=visits, first_state=first, **eng.read_trajectory(0, T))
        return value
    finally:
        if getattr(env, '_episode_batch', None) is None or env._episode_batch[1] is not batch:
            batch.close()
