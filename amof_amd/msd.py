"""Mean-squared displacement on MI355X (mirror of reference amof/msd.py).

``WindowMsd`` keeps the reference's signatures and ``.data`` schema
(amof/msd.py:140-268).  Centre-of-mass removal, optional unwrap, wrapped
frame-to-frame displacements and the per-window sums
(amof/msd.py:185-205,222-248; amof/trajectory.py:285-303) run in the HIP
kernels behind ``amof_msd_window``; the host keeps the window arithmetic, the
reference's normalisation quirk and the formula-weighted total.

Unlike the reference (amof/msd.py:230,237) the caller's frames are NOT
mutated; ``.data`` is the same.
"""

import logging

import numpy as np
import pandas as pd

from ._lazy import Deferred, EmptyUntilComputed

from . import _hip
from . import data as _data
from . import dist as _dist
from .files import path as _path
from .frames import pack_trajectory, resident_source, PackedTrajectory

logger = logging.getLogger(__name__)


class Msd(object):
    """
    Main class for MSD
    """

    def write_to_file(self, path_to_output):
        """path_to_output: where the MSD object will be written"""
        path_to_output = _path.append_suffix(path_to_output, 'msd')
        self.data.to_feather(path_to_output)

    @classmethod
    def from_msd(cls, *args):
        logger.exception('from_msd is deprecated, use from_file instead')

    @classmethod
    def from_file(cls, path_to_msd):
        """constructor of msd class from msd file"""
        msd_class = cls()
        msd_class.read_msd_file(path_to_msd)
        return msd_class

    def read_msd_file(self, path_to_data):
        """path_to_data: where the MSD object is"""
        path_to_data = _path.append_suffix(path_to_data, 'msd')
        self.data = pd.read_feather(path_to_data)


class DirectMsd(Msd):
    """
    Direct MSD (mirror of reference amof/msd.py:54-137; deprecated there)

    MSD(t) = 1/N sum_i (r_i(t) - r_i(0))^2 with a running unwrap that only works
    for orthogonal cells.  Better to use WindowMsd.
    """

    data = EmptyUntilComputed("Step")      # (the reference's empty first-column frame, built on first look)

    def __init__(self):
        """default constructor"""
        self.data = None
        logger.warning('DirectMsd is deprecated and not suitable for non-orthogonal cells, use WindowMsd instead')

    @classmethod
    def from_trajectory(cls, trajectory, delta_Step=1, first_frame=0, parallel=False, device=None):
        """
        Args:
            trajectory: list of ase.Atoms-like frames, or a PackedTrajectory
            delta_Step: number of simulation steps between two frames
            parallel: accepted for compatibility
        """
        from . import trajectory as amtraj
        msd_class = cls()
        step = amtraj.construct_step(delta_Step=delta_Step, first_frame=first_frame, number_of_frames=len(trajectory))
        msd_class.compute_msd(trajectory, step, parallel, device=device)
        return msd_class

    @staticmethod
    def compute_species_msd(trajectory, atomic_number=None, device=None):
        """MSD(t) of one species -- all atoms if ``atomic_number`` is None -- as the reference's static helper returns it
        (amof/msd.py:84-107: one float per frame, 0 at t = 0), from the same kernel as the class (``amof_msd_direct``
        computes every column of a trajectory at once; a caller that wants all of them uses ``from_trajectory``)."""
        packed = pack_trajectory(trajectory)
        if getattr(packed, "is_stream", False):
            packed = packed.read_all()
        dev = device if device is not None else getattr(packed, "device_index", None)
        msd, kinds = _hip.get_context(dev).msd_direct(packed)
        if atomic_number is None:
            return np.array(msd[:, 0])
        if int(atomic_number) not in kinds:
            raise ValueError("no atom of atomic number %r in the trajectory" % (atomic_number,))
        return np.array(msd[:, 1 + kinds.index(int(atomic_number))])

    def compute_msd(self, trajectory, step, parallel=False, device=None):
        packed = pack_trajectory(trajectory)
        if getattr(packed, "is_stream", False):
            packed = packed.read_all()      # (this analysis does not add up batch by batch)
        logger.info("Start computing msd for %s frames", len(packed))
        elements = packed.unique_numbers()
        dev = device if device is not None else getattr(packed, "device_index", None)
        msd, kinds = _hip.get_context(dev).msd_direct(packed)
        self.data = pd.DataFrame({"Step": step})
        self.data["X"] = msd[:, 0]
        for x in elements:
            self.data[_data.chemical_symbols[int(x)]] = msd[:, 1 + kinds.index(int(x))]


class WindowMsd(Msd, Deferred):
    """
    Window MSD

    ``from_trajectory`` enqueues the analysis on its device's second lane (a high-priority stream beside an RDF
    launch) and returns; ``.data`` and ``.sumsq`` wait for it (amof_amd/_lazy.py; ``AMOF_ASYNC=0``: synchronous).

    MSD(m) = 1/N_particles sum_i 1/(N-m) sum_k (r_i(k+m) - r_i(k))^2, with the
    reference's actual summation range k = 1 .. N-m-1 (amof/msd.py:195-204:
    time origin 0 is never written but the mean still divides by N-m).
    Time is expressed in fs.
    """

    data = EmptyUntilComputed("Time")      # (the reference's empty first-column frame, built on first look)

    def __init__(self):
        """default constructor"""
        self.data = None

    @classmethod
    def from_trajectory(cls, trajectory, delta_time=100, max_time="half", timestep=1, parallel=False,
                        unwrap=False, device=None, distributed=None):
        """
        constructor of msd class

        Args:
            trajectory: list of ase.Atoms-like frames, or a PackedTrajectory
            delta_time: int, time between two computed values of the MSD, in fs
            max_time: int or "half"
            timestep: int, time between two frames of the trajectory
            parallel: accepted for compatibility (atoms always run in parallel on the GPU)
            unwrap: Boolean, unwrap the trajectory before computing the MSD
        """
        msd_class = cls()
        half_time = (len(trajectory) // 2) * timestep
        if (isinstance(max_time, str) and max_time == "half") or max_time > half_time:
            max_time = half_time
        if delta_time < timestep:
            logger.exception("Delta_time should be larger than timestep")
        delta_m = delta_time // timestep
        window = np.arange(0, max_time // timestep, delta_m)
        time = timestep * window
        msd_class.compute_msd(trajectory, window, time, parallel, unwrap, device=device, distributed=distributed)
        return msd_class

    @staticmethod
    def compute_msd_of_m(delta_pos, m, device=None):
        """MSD(m) of a list of successive displacements -- ``delta_pos[0]`` the initial positions, as
        ``amof.trajectory.get_delta_pos`` returns it -- with the reference's normalisation (amof/msd.py:185-205: the sum over
        the origins k = 1 .. F - m - 1 divided by F - m and by the number of atoms).  The reference's static helper, on the
        GPU: the displacements are summed into positions once (no cell, no wrapping: the caller's displacements already are
        the minimum images) and ONE window goes through the kernels of the class.  Deviation: ``delta_pos[0]`` is not
        modified (the reference accumulates into it; successive calls on one list give the same values either way)."""
        delta = np.asarray([np.asarray(d, dtype=np.float64) for d in delta_pos], dtype=np.float64)
        if delta.ndim != 3 or delta.shape[2] != 3:
            raise ValueError("delta_pos: a list of [n_atoms][3] arrays is expected")
        F, n = delta.shape[0], delta.shape[1]
        m = int(m)
        if not 0 <= m < F:
            raise ValueError("m = %d outside [0, %d)" % (m, F))
        if n == 0:
            return float("nan")                                     # (the reference divides by len(r) == 0)
        pos = np.cumsum(delta, axis=0)
        packed = PackedTrajectory(pos, np.eye(3), np.ones(n, dtype=np.int64), pbc=(False, False, False))
        ctx = _hip.get_context(device if device is not None else _hip.default_device())
        sumsq, _ = ctx.msd_window(packed, np.array([m], dtype=np.int32), unwrap=False, remove_com=False)
        return float(sumsq[0][0]) / n / (F - m)

    def compute_msd(self, trajectory, window, time, parallel=False, unwrap=False, device=None, distributed=None):
        """compute the window MSD (reference amof/msd.py:207-268)"""
        packed = pack_trajectory(trajectory, device=device if device is not None else _hip.default_device())
        if getattr(packed, "is_stream", False):
            packed = packed.read_all()      # a window couples frames half a trajectory apart: nothing to stream
        elements = packed.unique_numbers()
        if unwrap == True:  # noqa: E712  (the reference compares with ==)
            logger.info("Unwrap trajectory before computing msd")
        logger.info("Start computing msd at %s times on a trajectory of %s frames", len(window), len(packed))

        rank, world = (0, 1) if distributed is False else _dist.world()
        merge = distributed is not False and _dist.merging(world)
        N, F = packed.n_atoms, len(packed)
        atom_range = _dist.shard_range(N, rank, world) if merge and distributed != 'local' else (0, N)
        dev = device if device is not None else getattr(packed, "device_index", None)
        ctx = _hip.lane_context(dev, 1)
        sharded = merge and distributed != 'local'
        on_device = sharded and _dist.device_collectives()
        com = csum = None
        if on_device and unwrap != True and packed.on_device:  # noqa: E712  (the unwrapped centre of mass is another quantity)
            # atoms are sharded (reference: one joblib worker per element, amof/msd.py:252-256).  What couples the ranks is
            # the centre of mass of every frame, a sum over ALL atoms: each rank reads ITS atoms once and contributes their
            # mass-weighted coordinate sums per frame, ONE all-reduce of [F][3] (120 kB at 5000 frames) completes them, and
            # the rank's second pass finishes its atoms (amof_msd_shard_begin / _finish: 1 / world of the trajectory per rank
            # and pass).  The first half and the all-reduce run here, in the calling thread -- collectives are issued in
            # program order, never from a lane -- the second half on the lane.
            import torch
            csum = torch.empty((F, 3), dtype=torch.float64, device=torch.device("cuda", ctx.device))
            try:
                ctx.msd_shard_begin(packed, window, atom_range, csum)
                _dist.all_reduce_sum(csum)
            except _hip.Unsupported:
                # general cells, irregular windows: the centre of mass frame-sharded into a zeroed table (x + 0 = x, exact),
                # then the general kernels on the rank's atoms
                csum = None
                com = torch.zeros((F, 3), dtype=torch.float64, device=torch.device("cuda", ctx.device))
                ctx.msd_com(packed, _dist.shard_range(F, rank, world), com)
                _dist.all_reduce_sum(com)

        # (a host trajectory: its one device copy -- started by whichever analysis came first -- must be complete: the wait is
        #  part of the lane job)
        source = resident_source(packed, ctx.device, allow=not merge and hasattr(ctx, "submit"))

        def local():
            # this rank's kernels (a lane job: amof_amd/_lazy.py)
            if getattr(source, "is_stream", False):
                source.read_all()
            if on_device:
                # the S x W sums stay in HBM from the kernels through their all-reduce
                import torch
                out = torch.zeros((len(_hip.packed_species(packed)[0]), len(window)), dtype=torch.float64,
                                  device=torch.device("cuda", ctx.device))
                if csum is not None:
                    return ctx.msd_shard_finish(packed, window, atom_range, csum, out)
                return ctx.msd_window(packed, window, unwrap=(unwrap == True), remove_com=True,  # noqa: E712
                                      atom_range=atom_range, com=com, out=out)
            return ctx.msd_window(packed, window, unwrap=(unwrap == True), remove_com=True,  # noqa: E712
                                  atom_range=atom_range)

        def finish(raw):
            sumsq, kinds = raw
            if on_device:
                _dist.all_reduce_sum(sumsq)
                sumsq = sumsq.cpu().numpy()
            elif sharded:
                sumsq = _dist.all_reduce_sum(sumsq, device=ctx.device)
            self._assemble(formula_dict, packed, sumsq, kinds, window, time, elements)

        # formula of the first frame (amof/msd.py:263), read here: the caller's frames are not touched from a lane
        if isinstance(trajectory, PackedTrajectory) or getattr(trajectory, "is_stream", False):
            formula_dict = packed.formula_count()
        else:
            formula_dict = dict(trajectory[0].symbols.formula._count)
        self._defer(ctx, local, finish, collective=sharded)

    def _assemble(self, formula_dict, packed, sumsq, kinds, window, time, elements):
        """normalisation quirk and the formula-weighted total (amof/msd.py:195-204,259-268) from the raw sums"""
        F = len(packed)
        self.sumsq = sumsq
        idx = {z: k for k, z in enumerate(kinds)}

        denom = (F - np.asarray(window)).astype(np.float64)
        counts = packed.species_counts()
        cols = {"Time": time}
        for e in elements:
            cols[_data.chemical_symbols[int(e)]] = sumsq[idx[int(e)]] / counts[int(e)] / denom
        # formula-weighted total (amof/msd.py:263-268)
        acc = 0.0
        for k, v in formula_dict.items():
            acc = acc + cols[k] * v
        cols['X'] = acc / sum(formula_dict.values())
        self.data = pd.DataFrame(cols)
