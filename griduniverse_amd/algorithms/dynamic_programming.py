"""value_iteration / policy_iteration with the reference's signatures and stopping rules
(core/algorithms/dynamic_programming.py:8-57); the sweeps run on the device and only the
per-iteration delta (one float64) returns to the host."""
import warnings

import numpy as np

from .utils import engine_of


def value_iteration(policy, env, value_function=None, threshold=0.00001, max_steps=1000, discount_factor=1.0):
    """{evaluation sweep; delta = max(v - v') (signed); greedy update} until delta < threshold
    (dynamic_programming.py:8-28).  Returns (value_function, policy); `policy` is updated in place."""
    eng = engine_of(env)
    v0 = np.zeros(env.world.size) if value_function is None else value_function
    eng.vi_set(v0, policy)
    # the whole loop is queued at once; `delta < threshold` is evaluated on the device after every round
    steps, deltas = eng.vi_run(discount_factor, threshold, max_steps)
    if steps == max_steps and max_steps > 0 and not deltas[-1] < threshold:
        warnings.warn('Value iteration did not reach the selected threshold. Finished after reaching '
                      'the maximum {} steps'.format(max_steps), UserWarning)
    v, pi = eng.vi_get()
    if max_steps > 0:
        policy[...] = pi
        return v, policy
    return v0, policy


def policy_iteration(policy, env, value_function=None, threshold=0.00001, max_steps=1000, discount_factor=1.0):
    """Evaluate the current policy sweep by sweep until it converges, improve it greedily, repeat
    until two consecutive converged value functions agree (dynamic_programming.py:31-57)."""
    eng = engine_of(env)
    converged_v = np.zeros(env.world.size) if value_function is None else value_function
    eng.vi_set(converged_v, policy)
    steps_left = max_steps
    while steps_left > 0:
        # the evaluation sweeps up to the next converged one run on the device in one call (stopping rule included)
        swept, deltas_eval = eng.vi_eval_run(discount_factor, threshold, steps_left)
        steps_left -= swept
        delta_eval = deltas_eval[-1]
        if delta_eval < threshold:
            v_now, pi_now = eng.vi_get()
            eng.vi_greedy(discount_factor)
            delta = np.max(converged_v - v_now)
            converged_v = v_now
            # the reference computes the improved policy into the SAME array it returns
            policy[...] = eng.vi_get()[1]
            if delta < threshold:
                break
        else:  # the last allowed sweep did not converge
            eng.vi_set(converged_v, eng.vi_get()[1])
            eng.vi_greedy(discount_factor)
            policy[...] = eng.vi_get()[1]
            warnings.warn('Policy iteration did not reach the selected threshold. Finished after reaching '
                          'the maximum {} steps with delta_eval {}'.format(max_steps, delta_eval), UserWarning)
    return converged_v, policy
