"""CPU restatement of the reference Monte-Carlo evaluation (test infrastructure).

Follows TheMTank/GridUniverse core/algorithms/monte_carlo.py:29-99, with the episode source made
explicit: `episodes` yields (states_hist, rewards_hist, done) exactly as run_episode (:7-26) returns
them.  Everything else -- first/every-visit bookkeeping, the truncated discounted return
`sum([(g**i) * r ... if (g**i) > threshold])`, the three update rules incl. the quirk that the
incremental-mean rule also pulls never-revisited states toward 0 -- is restated operation for
operation.  Pinned by golden G9 (tests/golden/mc_*.npz), captured from the real function with its
run_episode patched to replay the same episodes.
"""
import numpy as np


def episodes_from_trajectory(first_state, obs, reward, done):
    """Split a [T, N] no-auto-reset trajectory into run_episode-shaped triples, env e = episode e."""
    T, N = obs.shape
    for e in range(N):
        hits = np.flatnonzero(done[:, e])
        L = int(hits[0]) + 1 if hits.size else T
        states = [int(first_state[e])] + [int(s) for s in obs[:L, e]]
        rewards = [np.int64(r) for r in reward[:L, e]]
        yield states, rewards, bool(done[L - 1, e])


def monte_carlo_evaluation(n_states, episodes, every_visit=False, incremental_mean=True, stationary_env=True,
                           discount_factor=0.99, threshold=0.0001, alpha=0.001):
    total_visit_counter = np.zeros(n_states)
    total_return = np.zeros(n_states)
    value_function = np.zeros(n_states)
    for states_hist, rewards_hist, _done in episodes:
        visits = np.zeros(n_states)
        returns = np.zeros(n_states)
        for idx, state in enumerate(states_hist):
            if visits[state] != 0 and not every_visit:
                continue  # :57-62
            visits[state] += 1
            g = 0
            for i, r in enumerate(rewards_hist[idx:]):  # :69-70, builtin sum = left-to-right adds from int 0
                if (discount_factor ** i) > threshold:
                    g = g + (discount_factor ** i) * r
            returns[state] += g
        for state in range(n_states):  # :73-91
            total_visit_counter[state] += visits[state]
            if not incremental_mean:
                total_return[state] += returns[state]
            elif stationary_env:
                if total_visit_counter[state] > 0.0:
                    value_function[state] += (1 / total_visit_counter[state]) * (returns[state] - value_function[state])
            else:
                value_function[state] += alpha * (returns[state] - value_function[state])
    if not incremental_mean:  # :93-97
        for state in range(n_states):
            if total_visit_counter[state] > 0.0:
                value_function[state] = total_return[state] / total_visit_counter[state]
    return value_function, total_visit_counter
