import numpy as np


def np_random(seed=None):
    rng = np.random.RandomState()
    rng.seed(seed if seed is None else int(seed) % (2 ** 32))
    return rng, seed
