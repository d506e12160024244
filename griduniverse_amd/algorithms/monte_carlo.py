"""Monte-Carlo policy evaluation with the reference's names (core/algorithms/monte_carlo.py).

`run_episode` is the reference's scalar loop on the N = 1 facade.  `monte_carlo_evaluation` keeps
the reference's signature and arithmetic (float64, bit-exact given the same episodes) but generates
all `num_episodes` episodes in ONE fused rollout launch -- env e is episode e, actions sampled from
`policy[state]` by the per-env device RNG -- and reduces them on the device (csrc/gu_mc.hip).
"""
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


def monte_carlo_evaluation(policy, env, every_visit=False, incremental_mean=True, stationary_env=True,
                           discount_factor=0.99, threshold=0.0001, alpha=0.001, num_episodes=100, *,
                           max_steps_per_episode=1000, seed=0, return_details=False):
    """Value function of `policy` from `num_episodes` sampled episodes (monte_carlo.py:29-99).

    Episodes are processed in index order, so the result equals the reference's sequential loop fed
    with the same episodes.  Extra keyword-only arguments: the step cap of each episode (run_episode's
    default 1000), the RNG seed of the batch, and `return_details` to also get the visit counters and
    the raw trajectory."""
    from ..vec_env import VecGridUniverse
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
        eng.vi_set(np.zeros(env.world.size), policy)
        first = batch.reset()
        T = int(max_steps_per_episode)
        eng.reserve_trajectory(T)
        eng.rollout(T, 'sample', auto_reset=False, trajectory=True)
        pw, keep = discount_table(discount_factor, threshold, T)
        value, visits = eng.mc_evaluate(T, first, pw, keep, every_visit, incremental_mean, stationary_env, alpha)
        rows = np.asarray(policy, dtype=np.float64)
        if not (np.isfinite(rows).all() and (rows >= 0).all()
                and (np.abs(rows.sum(axis=1) - 1.0) <= np.sqrt(np.finfo(np.float64).eps)).all()):
            traj = eng.read_trajectory(0, T)  # rare path: some row is no distribution -- did an episode draw from it?
            _check_visited_policy_rows(rows, first, traj['obs'], traj['done'])
        if return_details:
            return value, dict(total_visit_counter=visits, first_state=first, **eng.read_trajectory(0, T))
        return value
    finally:
        if getattr(env, '_episode_batch', None) is None or env._episode_batch[1] is not batch:
            batch.close()
