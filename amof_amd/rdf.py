"""Radial distribution functions on MI355X (mirror of reference amof/rdf.py).

``Rdf`` keeps the reference's signatures and ``.data`` schema
(amof/rdf.py:28-122).  The per-frame neighbour search + histogramming that the
reference delegates to ``asap3.analysis.rdf.RadialDistributionFunction``
(amof/rdf.py:88-93) runs in the HIP kernels behind ``amof_rdf_accumulate``;
this module keeps the O(bins) host logic: rmax clamp, bin-count arithmetic with
Python float semantics, asap3-style normalisation, column naming.
"""

import functools
import logging
import os

import numpy as np
import pandas as pd

from ._lazy import Deferred, EmptyUntilComputed

from . import _hip
from . import data as _data
from . import dist as _dist
from .files import path as _path
from .frames import pack_trajectory, resident_source

logger = logging.getLogger(__name__)


DEFAULT_SHELL = "exact"


def normalize_rdf(hist, ncount, natoms, mean_volume, rmax, nbins):
    """asap3 ``RadialDistributionFunction.get_rdf`` normalisation ([3P-memory],
    consumed at amof/rdf.py:96,109):

        g[b] = V * H[b] / (shell_b * N * ncount),
        shell_b = 4 pi delta (r_b^2 + delta^2 / 12), delta = rmax / nbins, r_b = (b + 1/2) delta

    ``ncount`` = frames * N for the total RDF, frames * N_a for the partial
    a -> b (partials are normalised with the TOTAL density, so that
    g_XX = sum_ab (N_a/N) g_ab; corroborated by amof/rdf.py:114,216-227).

    Shell volume (assumption A1, DESIGN 5.1 -- asap3 is not available here and the reference holds no test that pins
    it): the default is the EXACT volume of the spherical shell, ``(4 pi / 3)(r_hi^3 - r_lo^3) = 4 pi delta (r_b^2 +
    delta^2 / 12)`` -- what asap3's ``get_rdf`` and its port ``ase.geometry.rdf.get_rdf`` divide by according to two
    independent reviewer readings (rounds 2 and 3), and the only choice for which an ideal gas gives g = 1 in EVERY bin,
    the first ones included (``tests/test_oracle_kat.py::test_ideal_gas_discriminates_the_shell_volume``).
    ``AMOF_RDF_SHELL=midpoint`` (or ``shell='midpoint'``) switches to ``4 pi r_b^2 delta``; the two differ by
    ``delta^2 / (12 r_b^2)`` relative: 8e-6 at r = 1 A, dr = 0.01 (above the 1e-6 bar, shrinking as 1/r^2).  A
    maintainer with asap3 installed settles it for good with ``tests/golden/make_thirdparty_goldens.py`` +
    ``tests/test_thirdparty_goldens.py``."""
    return normalize_rdf_shell(hist, ncount, natoms, mean_volume, rmax, nbins, os.environ.get("AMOF_RDF_SHELL", DEFAULT_SHELL))


@functools.lru_cache(maxsize=8)
def _shell_volumes(rmax, nbins, shell):
    """the bins' shell volumes (the same array for every column of a result and for the next result of the same grid)"""
    delta = rmax / nbins
    r = (np.arange(nbins) + 0.5) * delta
    vol = 4 * np.pi * r * r * delta if shell == "midpoint" else 4 * np.pi * delta * (r * r + delta * delta / 12.0)
    vol.flags.writeable = False
    return vol


@functools.lru_cache(maxsize=32)
def _column_index(names):
    return pd.Index(names)


def normalize_rdf_shell(hist, ncount, natoms, mean_volume, rmax, nbins, shell):
    if shell not in ("midpoint", "exact"):
        raise ValueError("AMOF_RDF_SHELL must be 'midpoint' or 'exact', not %r" % (shell,))
    vol = _shell_volumes(float(rmax), int(nbins), shell)
    return np.asarray(hist, dtype=np.float64) * (mean_volume / (natoms * ncount)) / vol


class Rdf(Deferred):
    """
    Main class for rdf

    ``from_trajectory`` enqueues the analysis on its device's first lane and returns; ``.data`` (and ``.hist``,
    ``.rmax``, ...) wait for it (amof_amd/_lazy.py; ``AMOF_ASYNC=0``: synchronous).
    """

    data = EmptyUntilComputed("r")      # (the reference's empty first-column frame, built on first look)

    def __init__(self):
        """default constructor"""
        self.data = None

    @classmethod
    def from_trajectory(cls, trajectory, dr=0.01, rmax='half_cell', device=None, distributed=None):
        """
        Constructor of rdf class

        Args:
            trajectory: list of ase.Atoms-like frames, or a PackedTrajectory
            dr, rmax: floats in Angstrom
                If rmax is 'half_cell', half of the minimum cell length over
                the trajectory is used (reference amof/rdf.py:74-79).
            device: GPU index (default: LOCAL_RANK or 0)
            distributed: None -> shard frames over the ranks of an initialised
                torch.distributed group (every rank holds the whole
                trajectory); 'local' -> ``trajectory`` already is this rank's
                own block of frames; False -> single process.
        """
        rdf_class = cls()
        rdf_class.compute_rdf(trajectory, dr, rmax, device=device, distributed=distributed)
        return rdf_class

    @classmethod
    def from_rdf(cls, *args):
        logger.exception('from_rdf is deprecated, use from_file instead')

    @classmethod
    def from_file(cls, path_to_rdf):
        """constructor of rdf class from rdf file"""
        rdf_class = cls()
        rdf_class.read_rdf_file(path_to_rdf)
        return rdf_class

    def compute_rdf(self, trajectory, dr, rmax, device=None, distributed=None):
        """compute rdf from a trajectory (reference amof/rdf.py:67-114)"""
        packed = pack_trajectory(trajectory, device=device if device is not None else _hip.default_device())
        atomic_numbers_unique = packed.unique_numbers()
        N_species = len(atomic_numbers_unique)
        rank, world = (0, 1) if distributed is False else _dist.world()
        merge = distributed is not False and _dist.merging(world)
        dev = device if device is not None else getattr(packed, "device_index", None)
        ctx = _hip.lane_context(dev, 0)
        # a host trajectory gets ONE device copy, uploaded while its first analyses walk the part that has arrived
        source = resident_source(packed, ctx.device, allow=not merge and hasattr(ctx, "submit"))

        # min over ALL frames of the three cell lengths, halved (amof/rdf.py:74)
        rmax_half_cell = np.min(packed.cell_lengths()) / 2
        if distributed == 'local' and merge:
            rmax_half_cell = _dist.all_reduce_min(rmax_half_cell, device=ctx.device)
        if isinstance(rmax, str) and rmax == 'half_cell':
            rmax = rmax_half_cell
        elif rmax > rmax_half_cell:
            logger.info("Specified rmax %s is larger than half cell; will use half_cell rmax", rmax)
            rmax = rmax_half_cell
        rmax = float(rmax)

        logger.info("Start computing rdf for %s frames with dr = %s and rmax = %s", len(packed), dr, rmax)
        bins = int(rmax // dr)          # Python float floor-division, as the reference (amof/rdf.py:82)
        r = np.arange(bins) * dr
        if bins <= 0:
            self.data = pd.DataFrame({"r": r})
            raise ValueError("rmax // dr gives no bin")

        F_local = len(packed)
        if getattr(source, "is_stream", False):
            # frames are independent: the integer counts of the batches add up (a file stream parses its next batch in a
            # background thread, a host trajectory's next frames are on their way over PCIe, while this one is on the GPU)
            if merge:
                raise ValueError("a streamed trajectory is analysed by one process (distributed=False)")

            def walk():
                hist, vol_sum, kinds = None, 0.0, None
                for batch in source.batches():
                    h, v, kinds = ctx.rdf_accumulate(batch, rmax, bins)
                    hist = h if hist is None else hist + h
                    vol_sum += v
                if source.cell is not None:
                    vol_sum = source.volume_sum()      # (the library's own left-to-right sum: identical to the unstreamed result)
                return hist, vol_sum, kinds

            self._defer(ctx, walk, lambda raw: self._finish(packed, raw[0], raw[1], F_local, raw[2], atomic_numbers_unique,
                                                            rmax, bins, r))
            return
        if merge and distributed != 'local':
            frame_range = _dist.shard_range(F_local, rank, world)
        else:
            frame_range = (0, F_local)
        sharded = merge and distributed != 'local'
        on_device = merge and _dist.device_collectives()

        def local():
            # this rank's kernels (a lane job: amof_amd/_lazy.py)
            if on_device:
                # the histogram stays in HBM from the kernels through the RCCL all-reduce (amof_rdf_accumulate_dev)
                import torch
                S = len(_hip.packed_species(packed)[0])
                buf = torch.zeros(S * S * bins + 1, dtype=torch.int64, device=torch.device("cuda", ctx.device))
                out = buf[:S * S * bins].view(S, S, bins)
                _, vol_sum, kinds = ctx.rdf_accumulate(packed, rmax, bins, frame_range=frame_range, out=out)
                return buf, vol_sum, kinds
            return ctx.rdf_accumulate(packed, rmax, bins, frame_range=frame_range)

        def finish(raw):
            # the ranks' merge (the calling thread: collectives in program order) and the DataFrame
            hist, vol_sum, kinds = raw
            n_frames = frame_range[1] - frame_range[0]
            if on_device:
                # A frame-sharded run needs ONE collective: every rank holds every cell, so the volume sum of the whole
                # trajectory is computed locally (PackedTrajectory.volume_sum, the library's own operation order) and the
                # frame count is known.  Own-block ranks ('local') carry their frame count in a spare word of the same
                # tensor and all-reduce the float volume sums separately.
                buf = hist
                S = len(kinds)
                if sharded:
                    _dist.all_reduce_sum(buf[:S * S * bins])
                    vol_sum, n_frames = packed.volume_sum(), F_local
                else:
                    buf[-1] = n_frames
                    _dist.all_reduce_sum(buf)
                    vol_sum = float(_dist.all_reduce_sum(np.array([vol_sum]), device=ctx.device)[0])
                host = buf.cpu().numpy()
                hist = host[:S * S * bins].reshape(S, S, bins).view(np.uint64)
                if not sharded:
                    n_frames = int(host[-1])
            elif sharded:
                hist = _dist.all_reduce_sum(hist, device=ctx.device)
                vol_sum, n_frames = packed.volume_sum(), F_local
            elif merge:
                hist = _dist.all_reduce_sum(hist, device=ctx.device)
                tot = _dist.all_reduce_sum(np.array([vol_sum, float(n_frames)]), device=ctx.device)
                vol_sum, n_frames = float(tot[0]), int(round(tot[1]))
            self._finish(packed, hist, vol_sum, n_frames, kinds, atomic_numbers_unique, rmax, bins, r)

        self._defer(ctx, local, finish, collective=merge)

    def _finish(self, packed, hist, vol_sum, n_frames, kinds, atomic_numbers_unique, rmax, bins, r):
        """normalisation and column assembly (amof/rdf.py:96-114) from the integer counts"""
        N_species = len(atomic_numbers_unique)
        self.hist = hist                      # integer ordered-pair counts [S][S][bins]
        self.kinds = kinds
        self.n_frames = n_frames
        self.rmax = rmax

        natoms = packed.n_atoms
        mean_volume = vol_sum / n_frames
        counts = packed.species_counts()
        idx = {z: k for k, z in enumerate(kinds)}

        # every column from ONE broadcast normalisation, one DataFrame construction (column order as the reference:
        # r, X-X, the cartesian product of the species, then the A-X sums; amof/rdf.py:96-114)
        ncount = np.array([n_frames * counts[z] for z in kinds], dtype=np.float64)
        partial = normalize_rdf(hist, ncount[:, None, None], natoms, mean_volume, rmax, bins)      # [S][S][bins]
        names = ["r", "X-X"]
        table = np.empty((2 + N_species * N_species + N_species, bins), dtype=np.float64)   # one row per column
        table[0] = r
        table[1] = normalize_rdf(hist.sum(axis=(0, 1)), n_frames * natoms, natoms, mean_volume, rmax, bins)
        sidx = [idx[int(z)] for z in atomic_numbers_unique]
        syms = [_data.chemical_symbols[int(z)] for z in atomic_numbers_unique]
        # (block copies instead of a Python loop per column: the DataFrame of a rank's share is assembled behind a 9 ms
        #  launch in an 8-GPU run)
        sub = partial[sidx][:, sidx] if sidx != list(range(len(kinds))) else partial                 # [i][j][bins]
        names += [syms[i] + "-" + syms[j] for i in range(N_species) for j in range(N_species)]
        table[2:2 + N_species * N_species] = sub.reshape(N_species * N_species, bins)
        # sum([...]) of the reference: 0 + g_A0 + g_A1 + ... in species order (amof/rdf.py:114), for every A at once
        acc = 0
        for j in range(N_species):
            acc = acc + sub[:, j]
        names += [syms[i] + "-X" for i in range(N_species)]
        table[2 + N_species * N_species:] = acc
        self.data = pd.DataFrame(table.T, columns=_column_index(tuple(names)))     # (a single float64 block: no per-column handling)

    def write_to_file(self, filename):
        filename = _path.append_suffix(filename, 'rdf')
        self.data.to_feather(filename)

    def read_rdf_file(self, path_to_data):
        path_to_data = _path.append_suffix(path_to_data, 'rdf')
        self.data = pd.read_feather(path_to_data)

    def get_coordination_number(self, nn_set, cutoff, density):
        """
        return coordination number (reference amof/rdf.py:124-131)
        nn_set: str indicating pair of neighbours
        cutoff: float, in Angstrom
        density: float, no units
        """
        return get_coordination_number(self.data['r'], self.data[nn_set], cutoff, density)


def _basic_simps(y, start, stop, x):
    """scipy.integrate._quadrature._basic_simpson for sample points x (scipy 1.7.1, [3P-memory])"""
    step = 2
    h = np.diff(x)
    sl0, sl1, sl2 = slice(start, stop, step), slice(start + 1, stop + 1, step), slice(start + 2, stop + 2, step)
    h0 = h[sl0]
    h1 = h[sl1]
    hsum = h0 + h1
    hprod = h0 * h1
    h0divh1 = h0 / h1
    tmp = hsum / 6.0 * (y[sl0] * (2 - 1.0 / h0divh1) + y[sl1] * (hsum * hsum / hprod) + y[sl2] * (2 - h0divh1))
    return np.sum(tmp)


def simps(y, x):
    """``scipy.integrate.simps(y, x)`` as in scipy 1.7.1 (default ``even='avg'``), which the reference
    pins (requirements.txt:15) and calls at amof/rdf.py:226.  Modern ``scipy.integrate.simpson`` treats
    an even number of samples differently, so the algorithm is restated here ([3P-memory])."""
    y = np.asarray(y, dtype=np.float64)
    x = np.asarray(x, dtype=np.float64)
    N = len(y)
    if N < 2:
        return 0.0
    if N % 2 == 0:
        val = 0.5 * (x[-1] - x[-2]) * (y[-1] + y[-2]) + 0.5 * (x[1] - x[0]) * (y[1] + y[0])
        result = _basic_simps(y, 0, N - 3, x) + _basic_simps(y, 1, N - 2, x)
        return result / 2.0 + val / 2.0
    return _basic_simps(y, 0, N - 2, x)


def get_coordination_number(r, rdf, cutoff, density):
    """
    return coordination number (reference amof/rdf.py:216-227)
    r, rdf: arrays of same size
    cutoff: float, in Angstrom
    density: float, number density of the entire system (counting every species) in Angstrom^-3
    """
    r = np.asarray(r)
    rdf = np.asarray(rdf)
    mask = (r > 0) & (r < cutoff)
    r = r[mask]
    rdf = rdf[mask]
    integral = simps(rdf * (r ** 2), r)
    return 4 * np.pi * density * integral


class CoordinationNumber(object):
    """
    Class to compute CoordinationNumber from RDF (mirror of reference amof/rdf.py:135-214;
    deprecated there: "Subjected to numerical errors in the integration step.  Best to use
    amof.cn.CoordinationNumber").

    Per frame, the partial RDF of every neighbour set is histogrammed on the GPU with
    ``dr`` (default 1e-4 A) up to the largest cutoff and integrated with Simpson's rule.
    The reference asks asap3 for ``get_rdf`` on an object that never ran ``update()``
    (amof/rdf.py:181-185); what is computed here is the evident intent, the RDF of that frame.
    """

    data = EmptyUntilComputed("Step")      # (the reference's empty first-column frame, built on first look)

    def __init__(self):
        """default constructor"""
        logger.warning('Compute CoordinationNumber from RDF, best to use amof.cn.CoordinationNumber')
        self.data = None

    @classmethod
    def from_trajectory(cls, trajectory, nb_set_and_cutoff, delta_Step=1, first_frame=0, dr=0.0001, parallel=False,
                        device=None):
        from . import trajectory as _trajectory
        cn_class = cls()
        step = _trajectory.construct_step(delta_Step=delta_Step, first_frame=first_frame,
                                          number_of_frames=len(trajectory))
        cn_class.compute_cn(trajectory, nb_set_and_cutoff, step, dr, parallel, device=device)
        return cn_class

    def compute_cn(self, trajectory, nb_set_and_cutoff, step, dr, parallel=False, device=None):
        packed = pack_trajectory(trajectory)
        if getattr(packed, "is_stream", False):
            packed = packed.read_all()      # (this analysis does not add up batch by batch)
        rmax = float(np.max(list(nb_set_and_cutoff.values())))
        logger.info("Start computing coordination number for %s frames with dr = %s and rmax = %s", len(packed), dr, rmax)
        bins = int(rmax // dr)
        r = np.arange(bins) * dr
        dev = device if device is not None else getattr(packed, "device_index", None)
        ctx = _hip.get_context(dev)
        natoms = packed.n_atoms
        rows = []
        for k in range(len(packed)):
            hist, vol, kinds = ctx.rdf_accumulate(packed, rmax, bins, frame_range=(k, k + 1))
            idx = {z: i for i, z in enumerate(kinds)}
            dic = {'Step': step[k]}
            density = natoms / vol                                     # amof.atom.get_number_density
            for nn_set, cutoff in nb_set_and_cutoff.items():
                za, zb = tuple(_data.atomic_numbers[i] for i in nn_set.split('-'))
                n_a = int((packed.numbers == za).sum())
                g = normalize_rdf(hist[idx[za], idx[zb]], n_a, natoms, vol, rmax, bins)
                dic[nn_set] = get_coordination_number(r, g, cutoff, density)
            rows.append(dic)
        self.data = pd.DataFrame(rows)

    @classmethod
    def from_file(cls, filename):
        cn_class = cls()
        cn_class.read_cn_file(filename)
        return cn_class

    def read_cn_file(self, filename):
        filename = _path.append_suffix(filename, 'cn')
        self.data = pd.read_feather(filename)

    def write_to_file(self, filename):
        filename = _path.append_suffix(filename, 'cn')
        self.data.to_feather(filename)
