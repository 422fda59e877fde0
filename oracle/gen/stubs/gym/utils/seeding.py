import numpy as np


def np_random(seed=None):
    # the reference never draws from this generator (MGR:132-134 only stores it)
    return np.random.RandomState(seed if seed is not None else 0), seed
