"""CPU restatement of the reference tabular DP sweeps (test infrastructure).

Follows TheMTank/GridUniverse:
  core/algorithms/utils.py:15-27                 single_step_policy_evaluation  (V1)
  core/algorithms/utils.py:55-72                 greedy_policy_from_value_function (V2)
  core/algorithms/dynamic_programming.py:8-28    value_iteration  (V3)
  core/algorithms/dynamic_programming.py:31-57   policy_iteration

All arithmetic is IEEE-754 float64 with one rounding per operation, in the
reference's left-to-right order; `env` is any object with the reference env's
surface (`world.size`, `reward_matrix`, `look_step_ahead`, `is_terminal`,
`action_space.n`).  Pinned by golden G4 (tests/golden/dp_*.npz).
"""
import warnings

import numpy as np


def single_step_policy_evaluation(policy, env, discount_factor=1.0, value_function=None):
    """One synchronous (Jacobi) Bellman expectation sweep.  utils.py:15-27.

    v'[s] = ((0.0 + R[s]) + p0*(g*v[n0])) + p1*(g*v[n1]) ... ; walls and terminal
    states are swept like any other state (quirk 12).
    """
    n_states = env.world.size
    v_old = np.zeros(n_states) if value_function is None else value_function
    v_new = np.zeros(n_states)
    for s in range(n_states):
        acc = np.float64(0.0) + env.reward_matrix[s]
        for a, p in enumerate(policy[s]):
            nxt = env.look_step_ahead(s, a)[0]
            acc = acc + p * (discount_factor * v_old[nxt])
        v_new[s] = acc
    return v_new


def greedy_policy_from_value_function(policy, env, value_function, discount_factor=1.0):
    """Tie-aware greedy improvement, mutating and returning `policy`.  utils.py:55-72.

    q[s,a] = 0.0 + (R[n] + g*v[n]); ties are detected after np.around(., 8); a
    terminal state's row becomes all zeros.  (The reference builds the row with
    the removed alias np.float == float64; utils.py:71.)
    """
    n_states, n_actions = env.world.size, env.action_space.n
    q = np.zeros((n_states, n_actions))
    for s in range(n_states):
        for a in range(n_actions):
            nxt, reward, _ = env.look_step_ahead(s, a)
            q[s][a] += reward + discount_factor * value_function[nxt]
        best = np.where(np.around(q[s], 8) == np.around(np.amax(q[s]), 8))[0]
        terminal = env.is_terminal(s)
        share = 1 / len(best)
        policy[s] = np.array([share if (a in best and not terminal) else 0 for a in range(n_actions)],
                             dtype=np.float64)
    return policy


def value_iteration(policy, env, value_function=None, threshold=0.00001, max_steps=1000, **kwargs):
    """dynamic_programming.py:8-28: {V1; signed delta; V2} until delta < threshold."""
    v = np.zeros(env.world.size) if value_function is None else value_function
    pi = policy
    for it in range(max_steps):
        v_next = single_step_policy_evaluation(pi, env, value_function=v, **kwargs)
        delta = np.max(v - v_next)  # signed, not absolute (:17)
        v = v_next
        pi = greedy_policy_from_value_function(pi, env, value_function=v, **kwargs)
        if delta < threshold:
            break
        if it == max_steps - 1:
            warnings.warn('Value iteration did not reach the selected threshold. Finished after reaching '
                          'the maximum {} steps'.format(it + 1), UserWarning)
    return v, pi


def policy_iteration(policy, env, value_function=None, threshold=0.00001, max_steps=1000, **kwargs):
    """dynamic_programming.py:31-57."""
    v = converged_v = np.zeros(env.world.size) if value_function is None else value_function
    pi = policy
    for it in range(max_steps):
        v_next = single_step_policy_evaluation(pi, env, value_function=v, **kwargs)
        delta_eval = np.max(v - v_next)
        v = v_next
        if delta_eval < threshold:
            candidate = greedy_policy_from_value_function(pi, env, value_function=v, **kwargs)
            delta = np.max(converged_v - v_next)
            converged_v = v_next
            if delta < threshold:
                break
            pi = candidate
        elif it == max_steps - 1:
            pi = greedy_policy_from_value_function(pi, env, value_function=converged_v, **kwargs)
            warnings.warn('Policy iteration did not reach the selected threshold. Finished after reaching '
                          'the maximum {} steps with delta_eval {}'.format(it + 1, delta_eval), UserWarning)
    return converged_v, pi
