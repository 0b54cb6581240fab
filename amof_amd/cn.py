"""Coordination numbers on MI355X (mirror of reference amof/cn.py).

``CoordinationNumber`` keeps the reference's signatures and ``.data`` schema
(amof/cn.py:25-100).  The ASE neighbour search that takes "92% of computation
time" in the reference (amof/cn.py:65) and the per-atom counting loop
(amof/cn.py:67-73) run in the HIP kernel behind ``amof_cn_count``.
"""

import logging

import numpy as np
import pandas as pd

from ._lazy import Deferred, EmptyUntilComputed

from . import _hip
from . import atom as amatom
from . import data as _data
from . import dist as _dist
from . import trajectory as _trajectory
from .files import path as _path
from .frames import pack_trajectory, resident_source

logger = logging.getLogger(__name__)


class CoordinationNumber(Deferred):
    """
    Main class to compute CoordinationNumber

    ``from_trajectory`` enqueues the analysis on its device's second lane and returns; ``.data`` waits for it
    (amof_amd/_lazy.py; ``AMOF_ASYNC=0``: synchronous).
    """

    data = EmptyUntilComputed("Step")      # (the reference's empty first-column frame, built on first look)

    def __init__(self):
        """default constructor"""
        self.data = None

    @classmethod
    def from_trajectory(cls, trajectory, nb_set_and_cutoff, delta_Step=1, first_frame=0, parallel=False,
                        device=None, distributed=None):
        """
        constructor of CoordinationNumber class from a trajectory
        Args:
            nb_set_and_cutoff: dict, keys are str indicating pair of neighbours,
                values are cutoffs float, in Angstrom
            parallel: accepted for compatibility; frames always run in
                parallel on the GPU
        """
        cn_class = cls()
        step = _trajectory.construct_step(delta_Step=delta_Step, first_frame=first_frame,
                                          number_of_frames=len(trajectory))
        cn_class.compute_cn(trajectory, nb_set_and_cutoff, step, parallel, device=device, distributed=distributed)
        return cn_class

    def compute_cn(self, trajectory, nb_set_and_cutoff, step, parallel=False, device=None, distributed=None):
        """compute coordination numbers (reference amof/cn.py:48-82)"""
        packed = pack_trajectory(trajectory, device=device if device is not None else _hip.default_device())
        logger.info("Start computing coordination number for %s frames", len(packed))
        cutoff_dict = amatom.format_cutoff(nb_set_and_cutoff)
        kinds, _ = _hip.packed_species(packed)
        lut = {z: k for k, z in enumerate(kinds)}
        rcm = amatom.cutoff_matrix(cutoff_dict, kinds)
        names, sets, present = [], [], []
        for nb_set in nb_set_and_cutoff.keys():
            a, b = tuple(_data.atomic_numbers[i] for i in nb_set.split('-'))
            names.append(nb_set)
            ok = a in lut and b in lut
            present.append(a in lut)
            sets.append((lut[a], lut[b]) if ok else None)
        live = [s for s in sets if s is not None]

        rank, world = (0, 1) if distributed is False else _dist.world()
        merge = distributed is not False and _dist.merging(world)
        F = len(packed)
        frame_range = _dist.shard_range(F, rank, world) if (merge and distributed != 'local') else (0, F)
        dev = device if device is not None else getattr(packed, "device_index", None)
        ctx = _hip.lane_context(dev, 1)
        counts = packed.species_counts()

        def assemble(sums):
            data = {'Step': np.asarray(step)[:len(sums)] if distributed == 'local' else step}
            k = 0
            for name, s, has_a in zip(names, sets, present):
                a = _data.atomic_numbers[name.split('-')[0]]
                n_a = counts.get(a, 0)
                if s is not None:
                    col = sums[:, k].astype(np.float64) / n_a   # np.mean of integer counts (amof/cn.py:73)
                    k += 1
                elif has_a:
                    col = np.zeros(len(sums))                   # centres exist, partner species absent
                else:
                    col = np.full(len(sums), np.nan)            # np.mean([]) in the reference
                data[name] = col
            self.data = pd.DataFrame(data)

        source = resident_source(packed, ctx.device, allow=not merge and hasattr(ctx, "submit"))
        if getattr(source, "is_stream", False):
            if merge:
                raise ValueError("a streamed trajectory is analysed by one process (distributed=False)")

            def walk():
                rows = [ctx.cn_count(batch, rcm, live) if live else np.zeros((len(batch), 0), dtype=np.int64)
                        for batch in source.batches()]
                return np.concatenate(rows, axis=0) if rows else np.zeros((0, len(live)), dtype=np.int64)

            self._defer(ctx, walk, assemble)
            return
        sharded = merge and distributed != 'local'

        def local():
            # this rank's kernels (a lane job: amof_amd/_lazy.py)
            if live:
                return ctx.cn_count(packed, rcm, live, frame_range=frame_range)
            return np.zeros((frame_range[1] - frame_range[0], 0), dtype=np.int64)

        def finish(sums):
            if sharded:
                sums = _dist.all_gather_rows(sums, device=ctx.device)
            assemble(sums)

        self._defer(ctx, local, finish, collective=sharded)

    @classmethod
    def from_file(cls, filename):
        """constructor of cn class from cn file"""
        cn_class = cls()
        cn_class.read_cn_file(filename)
        return cn_class

    def read_cn_file(self, filename):
        filename = _path.append_suffix(filename, 'cn')
        self.data = pd.read_feather(filename)

    def write_to_file(self, filename):
        filename = _path.append_suffix(filename, 'cn')
        self.data.to_feather(filename)
