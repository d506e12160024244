"""Import-path shim: `core.algorithms.monte_carlo` of the reference, on the engine.  Under THIS path monte_carlo_evaluation
defaults to the reference's own random draws (rng='numpy': start cells from the stdlib's global `random`, actions from numpy's
global stream, as core/algorithms/monte_carlo.py:20, 46-52 draws them), so an unmodified driver that seeds those streams gets the
reference's numbers byte for byte; `griduniverse_amd.algorithms.monte_carlo` itself defaults to the per-env device RNG."""
import functools

from griduniverse_amd.algorithms import monte_carlo as _mc
from griduniverse_amd.algorithms.monte_carlo import *  # noqa: F401,F403
from griduniverse_amd.algorithms.monte_carlo import run_episode  # noqa: F401


@functools.wraps(_mc.monte_carlo_evaluation)
def monte_carlo_evaluation(policy, env, *args, **kwargs):
    kwargs.setdefault('rng', 'numpy')
    return _mc.monte_carlo_evaluation(policy, env, *args, **kwargs)
