"""The `data` attribute of the analysis classes.

The reference's constructors start every object with an empty DataFrame holding the first column (`amof/rdf.py:33-35,144-146`,
`amof/msd.py:64-66,152-154`, `amof/bad.py:66-68`, `amof/cn.py:30-32`).  Building it costs ~0.1 ms of pandas per object --
4 % of one rank's 10 ms step in an 8-GPU run (profiles/r04/shards.txt) -- for a frame that `compute_*` replaces at once,
so it is built when somebody looks at it before a result has been stored: same object for every reader afterwards.
"""
import numpy as np
import pandas as pd


class EmptyUntilComputed:
    def __init__(self, column):
        self.column = column

    def __set_name__(self, owner, name):
        self.slot = "_" + name

    def __get__(self, obj, owner=None):
        if obj is None:
            return self
        val = obj.__dict__.get(self.slot)
        if val is None:
            val = pd.DataFrame({self.column: np.empty([0])})
            obj.__dict__[self.slot] = val
        return val

    def __set__(self, obj, value):
        obj.__dict__[self.slot] = value
