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
