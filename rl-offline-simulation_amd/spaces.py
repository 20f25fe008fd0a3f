"""Minimal stand-ins for gym.spaces.Discrete / Box.

The reference validates `isinstance(space, gym.spaces.Discrete)` (per_state_rejection.py:16-25).
gym is not part of this stack, so spaces are duck-typed: anything with an integer `.n` is discrete;
real gym spaces pass straight through.
"""
import numpy as np


class Discrete:
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
        self.dtype = np.int64

    def contains(self, x):
        return 0 <= int(x) < self.n

    def __repr__(self):
        return f"Discrete({self.n})"

    def __eq__(self, other):
        return is_discrete(other) and other.n == self.n


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.shape = tuple(shape) if shape is not None else np.shape(low)
        self.low = np.broadcast_to(np.asarray(low, dtype), self.shape)
        self.high = np.broadcast_to(np.asarray(high, dtype), self.shape)
        self.dtype = dtype

    def __repr__(self):
        return f"Box{self.shape}"


def is_discrete(space):
    return hasattr(space, "n") and not hasattr(space, "low")
