"""numpy restatements for the aMOF pair-distance hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Two groups:

1. Brute-force float64 pair / neighbour / angle references written from the
   specification in SURVEY 8a (explicit enumeration of lattice shifts; no cell
   list, no rounding tricks).  They are an independent second opinion on the C
   oracle -- "parity unpinned" like it, because the reference holds no vectors
   for RDF / CN / BAD.

2. The MSD path, restated line by line from the reference's own numpy code:
   ``get_delta_pos`` (amof/trajectory.py:285-303), ``compute_msd_of_m``
   (amof/msd.py:185-205) and ``compute_msd`` (amof/msd.py:207-268), plus
   ``construct_step`` (amof/trajectory.py:244-283).  ``compute_msd_of_m`` and
   ``construct_step`` are PINNED by golden vectors generated from the reference
   itself (tests/golden/make_reference_goldens.py).  ``wrap_positions`` is
   ase==3.20.1's published algorithm ([3P-memory]; call site
   amof/trajectory.py:302), unpinned.
"""

import itertools

import numpy as np


# --------------------------------------------------------------------------
# group 1: brute force
# --------------------------------------------------------------------------

def _shift_range(cell, pbc, R):
    inv = np.linalg.inv(cell)
    h = 1.0 / np.linalg.norm(inv, axis=0)
    M = [int(np.ceil(R / h[k])) + 1 if pbc[k] else 0 for k in range(3)]
    return [np.array(n) for n in itertools.product(*[range(-m, m + 1) for m in M])]


def _wrapped(pos, cell, pbc):
    """positions folded into the cell (periodic axes only)"""
    s = np.linalg.solve(cell.T, pos.T).T
    for k in range(3):
        if pbc[k]:
            s[:, k] -= np.floor(s[:, k])
    return s @ cell


def pair_distances(pos, cell, pbc, R):
    """yield (i_idx, j_idx, r) arrays for every ordered pair image with r < R
    (zero-shift self pairs excluded)."""
    pbc = np.broadcast_to(np.asarray(pbc, dtype=bool), (3,))
    p = _wrapped(np.asarray(pos, dtype=np.float64), cell, pbc)
    n = len(p)
    d0 = p[None, :, :] - p[:, None, :]  # d0[i, j] = r_j - r_i
    out_i, out_j, out_r, out_v = [], [], [], []
    for nvec in _shift_range(cell, pbc, R):
        T = nvec @ cell
        d = d0 + T
        r = np.sqrt((d ** 2).sum(axis=2))
        mask = r < R
        if not nvec.any():
            mask &= ~np.eye(n, dtype=bool)
        ii, jj = np.nonzero(mask)
        out_i.append(ii); out_j.append(jj); out_r.append(r[ii, jj]); out_v.append(d[ii, jj])
    return (np.concatenate(out_i), np.concatenate(out_j), np.concatenate(out_r), np.concatenate(out_v))


def rdf_hist(pos, cell, species, S, rmax, nbins, pbc=(True, True, True)):
    """ordered-pair histogram u64 [S][S][nbins] of one frame
    (bin = int(r / (rmax/nbins)), r < rmax; asap3 RawRDF semantics)."""
    species = np.asarray(species)
    ii, jj, r, _ = pair_distances(pos, cell, pbc, rmax)
    b = (r / (rmax / nbins)).astype(np.int64)
    ok = b < nbins
    hist = np.zeros((S, S, nbins), dtype=np.uint64)
    np.add.at(hist, (species[ii[ok]], species[jj[ok]], b[ok]), 1)
    return hist


def neighbour_lists(pos, cell, species, S, rcm, pbc=(True, True, True)):
    """list of (j, vector) per atom under per-pair cutoffs rcm[S][S]
    (ASE neighbor_list 'ij' semantics: strict <, all images, both directions)."""
    species = np.asarray(species)
    rcm = np.asarray(rcm, dtype=np.float64).reshape(S, S)
    R = rcm.max()
    nl = [[] for _ in range(len(pos))]
    if R <= 0:
        return nl
    ii, jj, r, v = pair_distances(pos, cell, pbc, R)
    keep = r < rcm[species[ii], species[jj]]
    for i, j, vec in zip(ii[keep], jj[keep], v[keep]):
        nl[i].append((int(j), vec))
    return nl


def cn_sums(pos, cell, species, S, rcm, sets, pbc=(True, True, True)):
    """per-set (sum of counts, per-atom counts dict) for one frame
    (reference amof/cn.py:67-73)."""
    species = np.asarray(species)
    nl = neighbour_lists(pos, cell, species, S, rcm, pbc)
    res = []
    for A, B in sets:
        counts = [sum(1 for (j, _) in nl[i] if species[j] == B) for i in range(len(pos)) if species[i] == A]
        res.append(np.array(counts, dtype=np.int64))
    return res


def angles(pos, cell, species, S, rcm, A, B, pbc=(True, True, True)):
    """all B-A-B angles in degrees for one frame (reference amof/bad.py:70-101;
    A or B = -1 stands for "X").  Valid for cells where every neighbour vector
    is the minimum-image one (cutoff below half the perpendicular heights)."""
    species = np.asarray(species)
    nl = neighbour_lists(pos, cell, species, S, rcm, pbc)
    out = []
    for a in range(len(pos)):
        if not (A < 0 or species[a] == A):
            continue
        vs = [v for (j, v) in nl[a] if (B < 0 or species[j] == B)]
        for v1, v2 in itertools.combinations(vs, 2):
            c = np.dot(v1 / np.linalg.norm(v1), v2 / np.linalg.norm(v2))
            out.append(180.0 / np.pi * np.arccos(np.clip(c, -1.0, 1.0)))
    return np.array(out)


def angles_of_triples(pos, cell, triples):
    """ase.Atoms.get_angles(triples, mic=True) ([3P-memory] of ase 3.20.1, assumption A6): the angle at triples[:, 1]
    between the minimum-image vectors to triples[:, 0] and triples[:, 2], degrees; the minimum image is searched over
    the 27 neighbouring lattice translations of the rounded fractional difference."""
    pos, cell, triples = np.asarray(pos, float), np.asarray(cell, float), np.asarray(triples, int)

    def mic(v):
        s = np.linalg.solve(cell.T, v.T).T
        s -= np.round(s)
        best = s @ cell
        bestn = (best ** 2).sum(axis=1)
        for n in np.ndindex(3, 3, 3):
            cand = s @ cell + (np.array(n) - 1) @ cell
            cn = (cand ** 2).sum(axis=1)
            better = cn < bestn - 1e-12
            best[better], bestn[better] = cand[better], cn[better]
        return best

    v1 = mic(pos[triples[:, 0]] - pos[triples[:, 1]])
    v2 = mic(pos[triples[:, 2]] - pos[triples[:, 1]])
    v1 /= np.linalg.norm(v1, axis=1)[:, None]
    v2 /= np.linalg.norm(v2, axis=1)[:, None]
    return 180.0 / np.pi * np.arccos(np.einsum('ij,ij->i', v1, v2).clip(-1.0, 1.0))


# --------------------------------------------------------------------------
# group 2: MSD path (reference-owned numpy, restated)
# --------------------------------------------------------------------------

def wrap_positions(positions, cell, pbc=True, center=(0.5, 0.5, 0.5), eps=1e-7):
    """ase.geometry.wrap_positions as published in ase 3.20.1 ([3P-memory]):
    fractional = solve(cell.T, pos.T).T - shift; periodic axes: %= 1, += shift."""
    if not hasattr(center, '__len__'):
        center = (center,) * 3
    pbc = np.broadcast_to(np.asarray(pbc, dtype=bool), (3,))
    shift = np.asarray(center, dtype=np.float64) - 0.5 - eps
    shift[np.logical_not(pbc)] = 0.0
    cell = np.asarray(cell, dtype=np.float64)
    fractional = np.linalg.solve(cell.T, np.asarray(positions).T).T - shift
    for i, periodic in enumerate(pbc):
        if periodic:
            fractional[:, i] %= 1.0
            fractional[:, i] += shift[i]
    return np.dot(fractional, cell)


def get_delta_pos(pos, cell, pbc=True):
    """reference amof/trajectory.py:285-303: delta[0] = pos[0] (aliased),
    delta[k+1] = wrap(pos[k+1] - pos[k]) in the EARLIER frame's cell."""
    delta_pos = [pos[0]]
    for k in range(len(pos) - 1):
        delta_pos.append(wrap_positions(pos[k + 1] - pos[k], cell[k], pbc=pbc, center=(0., 0., 0.)))
    return delta_pos


def compute_msd_of_m(delta_pos, m):
    """reference amof/msd.py:185-205, including its quirks: time origin 0 is
    never written (stays 0) while the mean still divides by F-m, and the
    running sum aliases (and mutates) delta_pos[0]."""
    msd_partial = np.zeros(len(delta_pos) - m)
    r_k_minus_m = delta_pos[0]
    r_k = r_k_minus_m * 0
    for k in range(0, m + 1):
        r_k += delta_pos[k]
    for k in range(m + 1, len(delta_pos)):
        r_k += delta_pos[k]
        r_k_minus_m += delta_pos[k - m]
        msd_partial[k - m] = np.linalg.norm(r_k - r_k_minus_m) ** 2 / len(r_k_minus_m)
    return np.mean(msd_partial)


def msd_window_setup(n_frames, delta_time=100, max_time="half", timestep=1):
    """reference amof/msd.py:173-181"""
    half_time = (n_frames // 2) * timestep
    if max_time == "half" or max_time > half_time:
        max_time = half_time
    delta_m = delta_time // timestep
    window = np.arange(0, max_time // timestep, delta_m)
    time = timestep * window
    return window, time


def window_msd(pos, cell, numbers, masses, window, pbc=True, unwrap=False, atom_subset=None):
    """reference amof/msd.py:207-261 on packed arrays (no mutation of inputs).

    Args:
        pos: [F][N][3]; cell: [F][3][3] or [3][3]; numbers, masses: [N]
        atom_subset: optional boolean mask of atoms entering the per-element
            loops (the centre of mass always uses every atom) -- used by the
            bounded CPU-baseline sample in bench.py.
    Returns:
        (elements, msd) with elements = list(set(numbers)) (reference order,
        amof/atom.py:44-46) and msd[e] = array over window.
    """
    pos = np.array(pos, dtype=np.float64)  # private copy: the reference mutates
    F = len(pos)
    cell = np.asarray(cell, dtype=np.float64).reshape(-1, 3, 3)
    cells = [cell[k if len(cell) > 1 else 0] for k in range(F)]
    numbers = np.asarray(numbers)
    masses = np.asarray(masses, dtype=np.float64)
    elements = list(set(numbers))
    if unwrap:
        positions = [pos[k].copy() for k in range(F)]
        delta_pos = get_delta_pos(positions, cells, pbc)
        new_pos = positions[0]
        for i in range(1, F):
            new_pos += delta_pos[i]
            pos[i] = new_pos
    for k in range(F):
        cg = np.dot(masses, pos[k]) / masses.sum()
        pos[k] -= cg
    out = []
    for x in elements:
        sel = numbers == x
        if atom_subset is not None:
            sel = sel & atom_subset
        positions = [pos[k][sel] for k in range(F)]
        delta_pos = get_delta_pos(positions, cells, pbc)
        out.append(np.array([compute_msd_of_m(delta_pos, int(m)) for m in window]))
    return elements, out


def window_msd_fast(pos, cell, numbers, masses, window, pbc=True, unwrap=False, atom_subset=None):
    """Vectorised equivalent of :func:`window_msd` (cumsum instead of the
    reference's running sums) for larger parity cases; validated against the
    loop version in tests/test_oracle_msd.py.

    ``atom_subset`` (boolean mask, not with ``unwrap``): the centre of mass
    uses every atom, the per-element sums only the selected ones -- headline-
    sized parity cases check a slice of the atoms."""
    if atom_subset is not None:
        if unwrap:
            raise ValueError("atom_subset is not supported together with unwrap")
        pos = np.asarray(pos, dtype=np.float64)
        masses = np.asarray(masses, dtype=np.float64)
        com = np.einsum('n,fnc->fc', masses, pos) / masses.sum()
        sel = np.asarray(atom_subset, dtype=bool)
        sub = pos[:, sel] - com[:, None, :]
        return _window_msd_centered(sub, cell, np.asarray(numbers)[sel], window, pbc)
    pos = np.array(pos, dtype=np.float64)
    F = len(pos)
    cell = np.asarray(cell, dtype=np.float64).reshape(-1, 3, 3)
    cells = [cell[k if len(cell) > 1 else 0] for k in range(F)]
    numbers = np.asarray(numbers)
    masses = np.asarray(masses, dtype=np.float64)
    elements = list(set(numbers))

    def deltas(p):
        d = np.zeros_like(p)
        for k in range(F - 1):
            d[k + 1] = wrap_positions(p[k + 1] - p[k], cells[k], pbc=pbc, center=(0., 0., 0.))
        return d

    if unwrap:
        pos = pos[0][None] + np.cumsum(deltas(pos), axis=0)
    pos = pos - (np.einsum('n,fnc->fc', masses, pos) / masses.sum())[:, None, :]
    u = np.cumsum(deltas(pos), axis=0)
    out = []
    for x in elements:
        ux = u[:, numbers == x]
        n = ux.shape[1]
        vals = []
        for m in window:
            m = int(m)
            if F - m - 1 >= 1:
                diff = ux[m + 1:] - ux[1:F - m]
                vals.append((diff ** 2).sum() / n / (F - m))
            else:
                vals.append(0.0)
        out.append(np.array(vals))
    return elements, out


def _window_msd_centered(pos, cell, numbers, window, pbc=True):
    """tail of :func:`window_msd_fast` for positions whose centre of mass is already removed"""
    F = len(pos)
    cell = np.asarray(cell, dtype=np.float64).reshape(-1, 3, 3)
    cells = [cell[k if len(cell) > 1 else 0] for k in range(F)]
    d = np.zeros_like(pos)
    for k in range(F - 1):
        d[k + 1] = wrap_positions(pos[k + 1] - pos[k], cells[k], pbc=pbc, center=(0., 0., 0.))
    u = np.cumsum(d, axis=0)
    elements = list(set(numbers))
    out = []
    for x in elements:
        ux = u[:, numbers == x]
        n = ux.shape[1]
        vals = []
        for m in window:
            m = int(m)
            if F - m - 1 >= 1:
                diff = ux[m + 1:] - ux[1:F - m]
                vals.append((diff ** 2).sum() / n / (F - m))
            else:
                vals.append(0.0)
        out.append(np.array(vals))
    return elements, out


def construct_step(**kwargs):
    """reference amof/trajectory.py:244-283"""
    delta_Step = kwargs.get('delta_Step', None)
    first_frame = kwargs.get('first_frame', None)
    last_frame = kwargs.get('last_frame', None)
    number_of_frames = kwargs.get('number_of_frames', None)
    step = kwargs.get('step', None)
    if step is not None:
        if isinstance(step, slice):
            return np.array(list(range(step.start or 0, step.stop, step.step or 1)))
        return np.array(step)
    if delta_Step is not None:
        if first_frame is not None and last_frame is not None:
            return np.arange(first_frame, last_frame, delta_Step)
        if number_of_frames is not None:
            if first_frame is None and last_frame is not None:
                first_frame = last_frame - number_of_frames * delta_Step
            if first_frame is not None:
                return np.arange(first_frame, first_frame + number_of_frames * delta_Step, delta_Step)
        return None
    if number_of_frames is not None:
        if first_frame is not None and last_frame is not None:
            return np.linspace(first_frame, last_frame, number_of_frames)
    return None


def direct_msd(pos, cell, numbers):
    """reference amof/msd.py:83-107 (DirectMsd.compute_species_msd) for 'X' and every element,
    vectorised over atoms (the reference loops over atoms in Python); orthogonal cells only.
    Returns (elements, {None or Z: msd[F]})."""
    pos = np.asarray(pos, dtype=np.float64)
    F = len(pos)
    cell = np.asarray(cell, dtype=np.float64).reshape(-1, 3, 3)
    numbers = np.asarray(numbers)
    elements = list(set(numbers))
    out = {}
    for x in [None] + elements:
        sel = slice(None) if x is None else (numbers == x)
        r_0 = pos[0][sel]
        r_t = r_0
        msd = np.zeros(F)
        for t in range(1, F):
            r_prev = r_t
            dr = np.zeros((len(r_0), 3))
            for j in range(3):
                a = cell[t if len(cell) > 1 else 0][j, j]
                dr[:, j] = (pos[t][sel] - r_prev % a)[:, j]
                dr[:, j] = np.where(dr[:, j] > a / 2, dr[:, j] - a, np.where(dr[:, j] < -a / 2, dr[:, j] + a, dr[:, j]))
            r_t = dr + r_prev
            msd[t] = np.linalg.norm(r_t - r_0) ** 2 / len(r_0)
        out[x] = msd
    return elements, out
