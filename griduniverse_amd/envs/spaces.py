"""The sliver of `gym.spaces` the reference's callers use:
`Discrete(n)` with `.n`, `.sample()` and `.contains()`.  gym itself is not a dependency."""
import numpy as np

_sampler = np.random.RandomState()


def seed(value=None):
    """Seed the module-level sampler behind Discrete.sample() (gym keeps one too)."""
    _sampler.seed(value)


class Discrete(object):
    def __init__(self, n):
        self.n = int(n)

    def sample(self):
        return int(_sampler.randint(self.n))

    def contains(self, x):
        try:
            return 0 <= int(x) < self.n and int(x) == x
        except (TypeError, ValueError):
            return False

    def __repr__(self):
        return 'Discrete({})'.format(self.n)

    def __eq__(self, other):
        return isinstance(other, Discrete) and other.n == self.n
