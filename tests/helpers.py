"""Shared builders for the parity tests (seeded synthetic trajectories)."""

import os
import re

import numpy as np

from amof_amd import data as _data
from amof_amd.frames import Frame, PackedTrajectory

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# Independent, pure-Python extended-XYZ reader (float() per field) for the fixtures: the yardstick the native
# reader (amof_amd.trajectory.read_xyz) is compared with bit for bit in tests/test_ingest.py.
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


def species_of(numbers):
    kinds = sorted(set(int(z) for z in numbers))
    return kinds, np.array([kinds.index(int(z)) for z in numbers], dtype=np.int32)


def zif4_frame():
    return read_extxyz(os.path.join(GOLDEN, "ZIF-4.xyz"), 0)


def replicate(frame, reps):
    """replicate a frame reps = (nx, ny, nz) times along its cell vectors"""
    pos, num = [], []
    for a in range(reps[0]):
        for b in range(reps[1]):
            for c in range(reps[2]):
                pos.append(frame.positions + a * frame.cell[0] + b * frame.cell[1] + c * frame.cell[2])
                num.append(frame.numbers)
    cell = frame.cell * np.array(reps)[:, None]
    return Frame(np.concatenate(num), np.concatenate(pos), cell, frame.pbc)


def random_walk(base, F, sigma, seed, wrap=True, cell_jitter=0.0, ortho=False):
    """PackedTrajectory: base + cumulative Gaussian steps, wrapped into the cell"""
    rng = np.random.default_rng(seed)
    cell0 = np.diag(np.diag(base.cell)) if ortho else base.cell.copy()
    pos = base.positions.copy()
    P = np.empty((F,) + pos.shape)
    C = np.empty((F, 3, 3))
    for k in range(F):
        cell = cell0 * (1.0 + cell_jitter * rng.normal()) if cell_jitter else cell0
        p = pos
        if wrap:
            s = np.linalg.solve(cell.T, p.T).T
            s -= np.floor(s)
            p = s @ cell
        P[k] = p
        C[k] = cell
        pos = pos + rng.normal(scale=sigma, size=pos.shape)
    if not cell_jitter:
        C = C[:1]
    return PackedTrajectory(P, C, base.numbers, pbc=base.pbc)


def random_gas(N, cell, numbers, seed, F=1):
    rng = np.random.default_rng(seed)
    cell = np.asarray(cell, dtype=float)
    if cell.shape == (3,):
        cell = np.diag(cell)
    # uniform in the cell, then displaced by whole lattice vectors (still uniform modulo the cell)
    P = (rng.uniform(0, 1, size=(F, N, 3)) + rng.integers(-1, 2, size=(F, N, 3))) @ cell
    return PackedTrajectory(P, cell, numbers)


def device_walk(device, reps, n_frames, sigma, seed, base=None):
    """Synthetic ZIF-4 supercell trajectory generated directly in HBM (torch, float64): Gaussian random walk,
    sigma per frame and axis, wrapped each frame into the constant orthorhombic cell.  The headline workload of
    bench.py (BASELINE.json configs[2]/[3]) and of tests/test_gpu_headline.py.  Returns a PackedTrajectory whose
    ``pos`` is a CUDA tensor; the caller synchronises (or relies on the library's stream ordering)."""
    import torch
    base = zif4_frame() if base is None else base
    rep = replicate(base, reps)
    lengths = np.diag(base.cell) * np.array(reps)          # constant orthorhombic cell
    cell = np.diag(lengths)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    n = len(rep.numbers)
    L = torch.tensor(lengths, dtype=torch.float64, device=device)
    traj = torch.empty((n_frames, n, 3), dtype=torch.float64, device=device)
    cur = torch.tensor(rep.positions, dtype=torch.float64, device=device)   # unwrapped position of the last frame
    chunk = 250
    for f0 in range(0, n_frames, chunk):
        f1 = min(f0 + chunk, n_frames)
        steps = torch.randn((f1 - f0, n, 3), dtype=torch.float64, device=device, generator=g) * sigma
        if f0 == 0:
            steps[0] = 0.0                                  # frame 0 is the base structure
        walk = cur + torch.cumsum(steps, dim=0)
        cur = walk[-1].clone()
        traj[f0:f1] = walk - torch.floor(walk / L) * L     # wrapped into the cell
        del steps, walk
    return PackedTrajectory(traj, cell, rep.numbers)


def device_walk_cell(device, base, cells, n_frames, sigma, seed):
    """Like device_walk, for ANY cell: a Gaussian random walk in Cartesian space from the (replicated) Frame `base`,
    wrapped each frame into `cells` -- one [3][3] cell or a per-frame [F][3][3] series (NPT) -- in fractional
    coordinates.  Atoms keep their fractional coordinates when the cell breathes.  Generated in HBM (torch, float64)."""
    import torch
    cells = np.asarray(cells, dtype=np.float64)
    per_frame = cells.ndim == 3 and cells.shape[0] > 1
    if cells.ndim == 2:
        cells = cells.reshape(1, 3, 3)
    assert cells.shape[0] in (1, n_frames)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    n = len(base.numbers)
    C = torch.tensor(cells, dtype=torch.float64, device=device)
    Cinv = torch.linalg.inv(C)
    traj = torch.empty((n_frames, n, 3), dtype=torch.float64, device=device)
    # the walk lives in the fractional coordinates of the first cell (steps of sigma Angstrom there)
    cur = torch.tensor(np.linalg.solve(np.asarray(base.cell).T, base.positions.T).T, dtype=torch.float64, device=device)
    chunk = 250
    for f0 in range(0, n_frames, chunk):
        f1 = min(f0 + chunk, n_frames)
        steps = torch.randn((f1 - f0, n, 3), dtype=torch.float64, device=device, generator=g) * sigma
        if f0 == 0:
            steps[0] = 0.0
        walk = cur + torch.cumsum(steps @ Cinv[0], dim=0)
        cur = walk[-1].clone()
        s = walk - torch.floor(walk)
        traj[f0:f1] = torch.matmul(s, C[f0:f1] if per_frame else C[0])
        del steps, walk, s
    return PackedTrajectory(traj, cells if per_frame else cells[0], base.numbers)


# ---------------------------------------------------------------------------------------------------------------
# A second, stricter stand-in for ase.Atoms (ASE itself is not installed here).  Unlike amof_amd.frames.Frame --
# written by the same hand as the code that consumes it -- this one copies the SHAPES of ASE 3.20's objects: the
# cell is a Cell object (not an ndarray: `.array`, `__array__`, `.lengths()`, `.volume`), `get_cell()` returns
# such an object, `pbc` is a bool ndarray, `positions` / `numbers` are properties over a private arrays dict,
# `symbols.formula._count` is a plain dict in order of first appearance.  Nothing here inherits from the product.
class CellLike(object):
    def __init__(self, array):
        self.array = np.array(array, dtype=float).reshape(3, 3)

    def __array__(self, dtype=None, copy=None):
        return self.array if dtype is None else self.array.astype(dtype)

    def __getitem__(self, item):
        return self.array[item]

    def __len__(self):
        return 3

    def copy(self):
        return CellLike(self.array.copy())

    def lengths(self):
        return np.linalg.norm(self.array, axis=1)

    def cellpar(self):
        ang = []
        for i in range(3):
            j, k = (i + 1) % 3, (i + 2) % 3
            ang.append(np.degrees(np.arccos(np.dot(self.array[j], self.array[k]) /
                                            (np.linalg.norm(self.array[j]) * np.linalg.norm(self.array[k])))))
        return np.array(list(self.lengths()) + ang)

    @property
    def volume(self):
        return abs(np.linalg.det(self.array))


class _FormulaLike(object):
    def __init__(self, symbols):
        self._count = {}
        for s in symbols:
            self._count[s] = self._count.get(s, 0) + 1


class _SymbolsLike(object):
    def __init__(self, numbers):
        self.numbers = numbers

    def __iter__(self):
        return (_data.chemical_symbols[int(z)] for z in self.numbers)

    @property
    def formula(self):
        return _FormulaLike(list(self))


class AseLikeAtoms(object):
    def __init__(self, numbers, positions, cell, pbc=True):
        self.arrays = {"numbers": np.array(numbers, dtype=int), "positions": np.array(positions, dtype=float)}
        self._cellobj = CellLike(cell)
        self._pbc = np.zeros(3, bool)
        self._pbc[:] = pbc

    cell = property(lambda self: self._cellobj)
    pbc = property(lambda self: self._pbc)
    numbers = property(lambda self: self.arrays["numbers"])
    positions = property(lambda self: self.arrays["positions"],
                         lambda self, v: self.arrays["positions"].__setitem__(slice(None), v))
    symbols = property(lambda self: _SymbolsLike(self.arrays["numbers"]))

    def __len__(self):
        return len(self.arrays["positions"])

    def copy(self):
        return AseLikeAtoms(self.numbers.copy(), self.positions.copy(), self.cell.array.copy(), self.pbc.copy())

    def get_positions(self):
        return self.arrays["positions"].copy()

    def set_positions(self, newpositions):
        self.arrays["positions"][:] = newpositions

    def get_atomic_numbers(self):
        return self.arrays["numbers"].copy()

    def get_cell(self, complete=False):
        return self._cellobj.copy()

    def get_pbc(self):
        return self._pbc.copy()

    def get_cell_lengths_and_angles(self):
        return self._cellobj.cellpar()

    def get_volume(self):
        return self._cellobj.volume

    def get_masses(self):
        return np.array([_data.atomic_masses[int(z)] for z in self.numbers])

    def get_center_of_mass(self):
        m = self.get_masses()
        return np.dot(m, self.positions) / m.sum()

    def translate(self, displacement):
        self.arrays["positions"] += np.array(displacement)


def as_ase_like(packed):
    """the frames of a host PackedTrajectory as AseLikeAtoms objects"""
    return [AseLikeAtoms(packed.numbers, packed.pos[k], packed.cell_of(k), packed.pbc) for k in range(packed.n_frames)]
