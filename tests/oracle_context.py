"""A stand-in for `amof_amd._hip.Context` whose entry points are answered by the CPU oracle -- TEST INFRASTRUCTURE ONLY.

The `-m "not gpu"` suite uses it to drive the host side of the analysis classes (argument handling, frame / atom
sharding, the lanes of amof_amd/_lazy.py, the torch.distributed merge over gloo) on a machine without a GPU; the
product never imports it and has no CPU path (amof_amd/_hip.py raises without the library or a GPU).
"""
import numpy as np

from amof_amd import _hip
from oracle import clib, numpy_oracle as no


class OracleContext(_hip.Lane):
    """same method names and return values as `_hip.Context` for the calls the four classes make"""

    def __init__(self, name="oracle-lane"):
        self.device = None
        self._lane_name = name
        self.calls = []

    def job_stats(self):
        return {"kernel_s_all": 0.0, "kernel_s_dominant": 0.0, "kernel_launches": 0, "path": self.calls[-1] if self.calls else ""}

    @staticmethod
    def _frames(packed, frame_range):
        f0, f1 = (0, packed.n_frames) if frame_range is None else frame_range
        cell = packed.cell if packed.cell.shape[0] == 1 else packed.cell[f0:f1]
        return packed.pos_host()[f0:f1], cell

    def rdf_accumulate(self, packed, rmax, nbins, frame_range=None, out=None):
        assert out is None
        self.calls.append("rdf")
        kinds, sp = _hip.packed_species(packed)
        pos, cell = self._frames(packed, frame_range)
        hist, vol = clib.rdf_hist(pos, cell, sp, len(kinds), rmax, nbins)
        return hist, vol, kinds

    def cn_count(self, packed, cutoff, sets, frame_range=None, per_atom=False):
        self.calls.append("cn")
        kinds, sp = _hip.packed_species(packed)
        pos, cell = self._frames(packed, frame_range)
        return clib.cn_counts(pos, cell, sp, len(kinds), np.asarray(cutoff, dtype=np.float64), sets, per_atom=per_atom)

    def bad_hist(self, packed, cutoff, triples, edges, frame_range=None, out=None):
        assert out is None
        self.calls.append("bad")
        kinds, sp = _hip.packed_species(packed)
        pos, cell = self._frames(packed, frame_range)
        return clib.bad_hist(pos, cell, sp, len(kinds), np.asarray(cutoff, dtype=np.float64), triples, np.asarray(edges))

    def bad_hist_by_cn(self, packed, cutoff, triples, edges, cn_max=16, frame_range=None):
        self.calls.append("bad_by_cn")
        kinds, sp = _hip.packed_species(packed)
        pos, cell = self._frames(packed, frame_range)
        return clib.bad_hist_by_cn(pos, cell, sp, len(kinds), np.asarray(cutoff, dtype=np.float64), triples, np.asarray(edges), cn_max)

    def msd_window(self, packed, windows, unwrap=False, remove_com=True, atom_range=None, com=None, out=None):
        assert com is None and out is None and remove_com
        self.calls.append("msd")
        kinds, _ = _hip.packed_species(packed)
        a0, a1 = (0, packed.n_atoms) if atom_range is None else atom_range
        mask = np.zeros(packed.n_atoms, dtype=bool)
        mask[a0:a1] = True
        F = packed.n_frames
        windows = np.asarray(windows)
        sums = np.zeros((len(kinds), len(windows)))
        if a1 > a0:
            elements, ref = no.window_msd_fast(packed.pos_host(), packed.cell, packed.numbers, packed.masses, windows,
                                               unwrap=unwrap, atom_subset=mask)
            for e, r in zip(elements, ref):
                n_e = int((packed.numbers[a0:a1] == e).sum())
                sums[kinds.index(int(e))] = np.asarray(r) * n_e * (F - windows)
        return sums, kinds


def install(monkeypatch=None):
    """route `_hip.get_context` / `_hip.lane_context` to two oracle-backed lanes; returns them"""
    lanes = {0: OracleContext("oracle-lane-0"), 1: OracleContext("oracle-lane-1")}
    lanes[1]._follows = lanes[0]            # (as _hip.get_context(device, lane=1) sets it: lane 1 follows lane 0)

    def lane_context(device, lane):
        from amof_amd import _lazy
        return lanes[lane if _lazy.async_enabled() else 0]

    def get_context(device=None, lane=0):
        return lanes[lane]
    if monkeypatch is not None:
        monkeypatch.setattr(_hip, "lane_context", lane_context)
        monkeypatch.setattr(_hip, "get_context", get_context)
    else:
        _hip.lane_context, _hip.get_context = lane_context, get_context
    return lanes
