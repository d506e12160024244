import numpy as np

_rng = np.random.RandomState()


class Discrete(object):
    def __init__(self, n):
        self.n = n

    def sample(self):
        return int(_rng.randint(self.n))

    def contains(self, x):
        return 0 <= int(x) < self.n
