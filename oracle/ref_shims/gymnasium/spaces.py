"""Descriptive containers with the constructor surface the reference uses."""
import numpy as np


class Space:
    pass


class Box(Space):
    def __init__(self, low=None, high=None, shape=None, dtype=float):
        self.low = np.asarray(low, dtype=float)
        self.high = np.asarray(high, dtype=float)
        self.shape = tuple(shape) if shape is not None else np.shape(self.low)
        self.dtype = dtype


class Dict(Space):
    def __init__(self, spaces=None):
        self.spaces = dict(spaces or {})

    def __setitem__(self, key, value):
        self.spaces[key] = value

    def __getitem__(self, key):
        return self.spaces[key]
