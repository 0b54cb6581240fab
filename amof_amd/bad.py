"""Bond-angle distributions on MI355X (mirror of reference amof/bad.py).

``Bad`` keeps the reference's signatures and ``.data`` schema
(amof/bad.py:33-169).  Neighbour search, minimum-image bond vectors, angles and
the histogram counts (amof/bad.py:70-114 + numpy.histogram at :160) run in the
HIP kernel behind ``amof_bad_hist``; the host keeps species / column logic and
the density normalisation.  Angles are never materialised as Python floats
(the reference keeps every angle of every frame in lists, amof/bad.py:156-160).

Deviation: when the cutoff keys cover every species the reference appends the
string "X" to its element list (amof/bad.py:127-128) and then indexes
``ase.data.chemical_symbols["X"]`` (amof/bad.py:111), which raises TypeError.
Here "X" (any species) works and is spelled 'X' in the column names.
"""

import logging

import numpy as np
import pandas as pd

from ._lazy import Deferred, EmptyUntilComputed

from . import _hip
from . import atom as amatom
from . import data as _data
from . import dist as _dist
from .files import path as _path
from .frames import pack_trajectory, resident_source

logger = logging.getLogger(__name__)


def _symbol(c):
    return "X" if isinstance(c, str) else _data.chemical_symbols[c]


class CoreBad(object):
    """
    Core material for every Bad function
    """

    @classmethod
    def from_trajectory(cls, trajectory, nb_set_and_cutoff, dtheta=0.05, normalization='total', parallel=False,
                        device=None, distributed=None):
        """
        constructor of bad class from a trajectory
        Args:
            nb_set_and_cutoff: dict, keys are str indicating pair of neighbours,
                values are cutoffs float, in Angstrom
            dtheta: float, in degrees
            normalization: str, unused (as in the reference, amof/bad.py:120)
            parallel: accepted for compatibility; frames always run in
                parallel on the GPU
        """
        bad_class = cls()
        bad_class.compute_bad(trajectory, nb_set_and_cutoff, dtheta, normalization, parallel,
                              device=device, distributed=distributed)
        return bad_class

    @classmethod
    def from_file(cls, filename):
        """constructor of bad class from bad file"""
        bad_class = cls()
        bad_class.read_bad_file(filename)
        return bad_class


class Bad(CoreBad, Deferred):
    """
    Main class for bad

    ``from_trajectory`` enqueues the analysis on its device's second lane and returns; ``.data``, ``.hist``,
    ``.n_angles`` wait for it (amof_amd/_lazy.py; ``AMOF_ASYNC=0``: synchronous).
    """

    data = EmptyUntilComputed("theta")      # (the reference's empty first-column frame, built on first look)

    def __init__(self):
        """default constructor"""
        self.data = None

    def compute_bad(self, trajectory, nb_set_and_cutoff, dtheta, normalization='total', parallel=False,
                    device=None, distributed=None):
        """compute bond-angle distributions (reference amof/bad.py:116-160)"""
        packed = pack_trajectory(trajectory, device=device if device is not None else _hip.default_device())
        atomic_numbers_unique = packed.unique_numbers()

        cutoff_dict = amatom.format_cutoff(nb_set_and_cutoff)
        elements_present_unique = list(set([_data.atomic_numbers[i] for nb_set in nb_set_and_cutoff.keys()
                                            for i in nb_set.split('-')]))
        if len(elements_present_unique) == len(atomic_numbers_unique):
            elements_present_unique.append("X")
        elements = [(a, b) for b in elements_present_unique for a in elements_present_unique
                    if (a not in [b, "X"] or ((a, b) == ("X", "X")))]

        logger.info("Start computing bad for %s frames with dtheta = %s", len(packed), dtheta)
        bins = int(180 // dtheta)       # Python float floor-division (amof/bad.py:142)
        theta_bins = np.arange(bins + 2) * dtheta
        theta = np.arange(bins + 1) * dtheta + dtheta / 2

        kinds, _ = _hip.packed_species(packed)
        lut = {z: k for k, z in enumerate(kinds)}
        rcm = amatom.cutoff_matrix(cutoff_dict, kinds)

        def sidx(c):
            return -1 if isinstance(c, str) else lut.get(c, None)

        names, triples = [], []
        for A, B in elements:
            aba_str = "-".join([_symbol(C) for C in [B, A, B]])
            ia, ib = sidx(A), sidx(B)
            if ia is None or ib is None:
                continue                # species absent from the trajectory: no angle, column omitted
            names.append(aba_str)
            triples.append((ia, ib))

        rank, world = (0, 1) if distributed is False else _dist.world()
        merge = distributed is not False and _dist.merging(world)
        F = len(packed)
        frame_range = _dist.shard_range(F, rank, world) if (merge and distributed != 'local') else (0, F)
        dev = device if device is not None else getattr(packed, "device_index", None)
        ctx = _hip.lane_context(dev, 1)
        db = np.array(np.diff(theta_bins), float)

        def assemble(hist, nang):
            self.hist = hist
            self.n_angles = nang
            self.columns = names
            cols = {"theta": theta}
            for k, aba_str in enumerate(names):
                if nang[k] != 0:            # columns without any angle are omitted (amof/bad.py:159)
                    n = hist[k].astype(np.int64)
                    cols[aba_str] = n / db / n.sum()      # numpy.histogram(density=True)
            self.data = pd.DataFrame(cols)

        source = resident_source(packed, ctx.device, allow=not merge and hasattr(ctx, "submit"))
        if getattr(source, "is_stream", False):
            if merge:
                raise ValueError("a streamed trajectory is analysed by one process (distributed=False)")

            def walk():
                hist, nang = np.zeros((len(triples), bins + 1), dtype=np.uint64), np.zeros(len(triples), dtype=np.uint64)
                for batch in source.batches():
                    if triples:
                        h, a = ctx.bad_hist(batch, rcm, triples, theta_bins)
                        hist, nang = hist + h, nang + a
                return hist, nang

            self._defer(ctx, walk, lambda raw: assemble(raw[0], raw[1]))
            return
        on_device = bool(triples) and merge and _dist.device_collectives()
        T, nb = len(triples), bins + 1

        def local():
            # this rank's kernels (a lane job: amof_amd/_lazy.py)
            if on_device:
                # counts stay in HBM from the kernels through ONE RCCL all-reduce (amof_bad_hist_dev)
                import torch
                both = torch.zeros(T * nb + T, dtype=torch.int64, device=torch.device("cuda", ctx.device))
                ctx.bad_hist(packed, rcm, triples, theta_bins, frame_range=frame_range, out=(both[:T * nb], both[T * nb:]))
                return both, None
            if triples:
                return ctx.bad_hist(packed, rcm, triples, theta_bins, frame_range=frame_range)
            return np.zeros((0, bins + 1), dtype=np.uint64), np.zeros(0, dtype=np.uint64)

        def finish(raw):
            hist, nang = raw
            if on_device:
                _dist.all_reduce_sum(hist)
                both = hist.cpu().numpy().view(np.uint64)
                hist, nang = both[:T * nb].reshape(T, nb), both[T * nb:]
            elif merge:
                hist = _dist.all_reduce_sum(hist, device=ctx.device)
                nang = _dist.all_reduce_sum(nang, device=ctx.device)
            assemble(hist, nang)

        self._defer(ctx, local, finish, collective=merge)

    def write_to_file(self, filename):
        filename = _path.append_suffix(filename, 'bad')
        self.data.to_feather(filename)

    def read_bad_file(self, path_to_data):
        path_to_data = _path.append_suffix(path_to_data, 'bad')
        self.data = pd.read_feather(path_to_data)


def _largest_neighbour_count(ctx, packed, rcm, frame_range, batch_bytes=256 << 20):
    """largest number of neighbours (over every species with a cutoff to the centre's) of any atom in any frame of the
    range: the per-atom output of the CN kernels, in frame batches of bounded size.  An upper bound of the coordination
    number any B-A-B / X-A-X triple can see, exact for the X triples."""
    S = rcm.shape[0]
    sets = [(a, b) for a in range(S) for b in range(S) if rcm[a, b] > 0]
    if not sets:
        return 0
    f0, f1 = frame_range
    step = max(1, int(batch_bytes // max(1, 4 * len(sets) * packed.n_atoms)))
    largest = 0
    for k in range(f0, f1, step):
        _, pa = ctx.cn_count(packed, rcm, sets, frame_range=(k, min(k + step, f1)), per_atom=True)
        # an atom's row is -1 in the sets whose centre species is not its own: count those as zero
        largest = max(largest, int(np.maximum(pa, 0).sum(axis=1).max()))
    return largest


class BadByCn(CoreBad):
    """
    Bond-angle distributions split by coordination number (mirror of reference
    amof/bad.py:172-309): BAD for A bonded to 2 B, 3 B, ...  With
    ``normalization='partial'`` each cn is weighted by its share of the angles, so
    that the partial BADs of a triple add up to the ``Bad`` result.

    The reference stores the result as an ``xarray.Dataset`` (variable 'bad',
    dims atom_triple x cn x theta).  ``.data`` is that Dataset when xarray is
    importable; the same numbers are always available as ``.bad``
    (``{triple: {cn: density[theta]}}``), ``.theta`` and the integer counts ``.hist``.
    """

    CN_MAX = 16      # first guess of the largest coordination number (slots of the device histogram); grows on demand

    def __init__(self):
        """default constructor"""
        self.data = None
        self.bad = {}
        self.theta = np.empty([0])

    def compute_bad(self, trajectory, nb_set_and_cutoff, dtheta, normalisation='total', parallel=False,
                    device=None, distributed=None):
        """compute bond-angle distributions by cn (reference amof/bad.py:240-301)"""
        packed = pack_trajectory(trajectory)
        if getattr(packed, "is_stream", False):
            packed = packed.read_all()      # (this analysis does not add up batch by batch)
        atomic_numbers_unique = packed.unique_numbers()
        cutoff_dict = amatom.format_cutoff(nb_set_and_cutoff)
        elements_present_unique = list(set([_data.atomic_numbers[i] for nb_set in nb_set_and_cutoff.keys()
                                            for i in nb_set.split('-')]))
        if len(elements_present_unique) == len(atomic_numbers_unique):
            elements_present_unique.append("X")
        elements = [(a, b) for b in elements_present_unique for a in elements_present_unique
                    if (a not in [b, "X"] or ((a, b) == ("X", "X")))]
        logger.info("Start computing bad for %s frames with dtheta = %s", len(packed), dtheta)
        bins = int(180 // dtheta)
        theta_bins = np.arange(bins + 2) * dtheta
        theta = np.arange(bins + 1) * dtheta + dtheta / 2
        kinds, _ = _hip.packed_species(packed)
        lut = {z: k for k, z in enumerate(kinds)}
        rcm = amatom.cutoff_matrix(cutoff_dict, kinds)
        names, triples = [], []
        for A, B in elements:
            ia = -1 if isinstance(A, str) else lut.get(A, None)
            ib = -1 if isinstance(B, str) else lut.get(B, None)
            if ia is None or ib is None:
                continue
            names.append("-".join([_symbol(C) for C in [B, A, B]]))
            triples.append((ia, ib))
        rank, world = (0, 1) if distributed is False else _dist.world()
        merge = distributed is not False and _dist.merging(world)
        F = len(packed)
        frame_range = _dist.shard_range(F, rank, world) if (merge and distributed != 'local') else (0, F)
        dev = device if device is not None else getattr(packed, "device_index", None)
        ctx = _hip.get_context(dev)
        # the last slot (cn_max) also collects every larger neighbour count -- the reference has no limit on the
        # coordination number (amof/bad.py:190-224).  When it is populated the slots are sized ONCE from a count pass (the
        # CN kernels' per-atom output: the largest number of neighbours any centre has in any frame) and the histogram is
        # taken a second and last time; rounds 2 - 3 grew the slots x4 per retry, each a whole pass and an all-reduce.
        cn_max = self.CN_MAX
        self.passes = 0
        while True:
            if triples:
                hist, nang = ctx.bad_hist_by_cn(packed, rcm, triples, theta_bins, cn_max=cn_max, frame_range=frame_range)
            else:
                hist = np.zeros((0, cn_max + 1, bins + 1), dtype=np.uint64)
                nang = np.zeros((0, cn_max + 1), dtype=np.uint64)
            self.passes += 1
            full = float(nang[:, cn_max].any()) if nang.size else 0.0
            if merge:
                full = _dist.all_reduce_sum(np.array([full]), device=ctx.device)[0]     # every rank must take the same decision
            if not full:
                break
            if self.passes > 3:
                raise RuntimeError("BadByCn: a centre has more neighbours than the count pass found (%d)" % (cn_max - 1))
            if self.passes > 1:
                # the count pass (the CN kernels' neighbour decision) and the BAD kernels disagree by a neighbour at the
                # cutoff: should not happen (one canonical arithmetic), but a doubling costs a pass, an exception the result
                cn_max = 2 * cn_max
                continue
            largest = _largest_neighbour_count(ctx, packed, rcm, frame_range)
            if merge:
                largest = -int(_dist.all_reduce_min(-float(largest), device=ctx.device))
            cn_max = max(cn_max + 1, int(largest) + 1)          # slot cn_max stays empty: it is the overflow slot
        if merge:
            hist = _dist.all_reduce_sum(hist, device=ctx.device)
            nang = _dist.all_reduce_sum(nang, device=ctx.device)
        self.hist, self.n_angles, self.columns, self.theta = hist, nang, names, theta
        db = np.array(np.diff(theta_bins), float)
        self.bad = {}
        for k, aba_str in enumerate(names):
            cns = [c for c in range(2, cn_max) if nang[k, c] != 0]
            if not cns:
                continue                          # no angle at all: triple omitted (amof/bad.py:282)
            num_angles_all = int(nang[k].sum())
            per_cn = {}
            for c in cns:
                n = hist[k, c].astype(np.int64)
                ratio = float(nang[k, c]) / num_angles_all if normalisation == 'partial' else 1
                per_cn[c] = ratio * (n / db / n.sum())
            self.bad[aba_str] = per_cn
        try:
            import xarray as xr
        except ImportError:
            self.data = None
            return
        dic_of_xarray = {aba: xr.DataArray([v[c] for c in v], coords={"cn": list(v), "theta": theta}, dims=("cn", "theta"))
                         for aba, v in self.bad.items()}
        xa = xr.Dataset(dic_of_xarray).to_array("atom_triple")
        self.data = xr.Dataset({'bad': xa})

    def write_to_file(self, filename):
        if self.data is None:
            raise ImportError("xarray (netCDF) is needed to write a BadByCn file, as in the reference")
        filename = _path.append_suffix(filename, 'bad')
        self.data.to_netcdf(filename)

    def read_bad_file(self, filename):
        import xarray as xr
        filename = _path.append_suffix(filename, 'bad')
        self.data = xr.open_dataset(filename)
