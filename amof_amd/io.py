"""Minimal extended-XYZ reader (fixture ingest only).

Reads the ``Lattice="..."`` comment-line convention used by the reference's
fixture ``examples/files/ZIF-4.xyz``.  The reference itself goes through
``ase.io.read`` (amof/trajectory.py:56); this reader only covers what the
parity tests need.  Full trajectory ingest is a later row.
"""

import re

import numpy as np

from . import data as _data
from .frames import Frame


def read_extxyz(path, index=None):
    """Return a list of :class:`Frame` (or one frame if ``index`` is an int)."""
    frames = []
    with open(path, "r") as fh:
        lines = fh.read().splitlines()
    i = 0
    while i < len(lines):
        if not lines[i].strip():
            i += 1
            continue
        n = int(lines[i].split()[0])
        comment = lines[i + 1]
        m = re.search(r'Lattice="([^"]*)"', comment)
        if m is None:
            raise ValueError("no Lattice= in comment line of %s" % path)
        cell = np.array([float(x) for x in m.group(1).split()]).reshape(3, 3)
        pbc = (True, True, True)
        mp = re.search(r'pbc="([^"]*)"', comment)
        if mp is not None:
            pbc = tuple(t.upper().startswith("T") for t in mp.group(1).split())
        symbols, pos = [], []
        for line in lines[i + 2:i + 2 + n]:
            w = line.split()
            symbols.append(w[0])
            pos.append([float(w[1]), float(w[2]), float(w[3])])
        numbers = [_data.atomic_numbers[s] for s in symbols]
        frames.append(Frame(numbers, np.array(pos), cell, pbc))
        i += 2 + n
    if isinstance(index, int):
        return frames[index]
    return frames
