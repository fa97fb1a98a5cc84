import numpy as np


class Space:
    pass


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is None:
            shape = np.shape(low)
        self.shape = tuple(shape)
        self.dtype = dtype
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), self.shape).copy()

    def __repr__(self):
        return "Box%s" % (self.shape,)


class Discrete(Space):
    def __init__(self, n):
        self.n = int(n)
        self.shape = ()
