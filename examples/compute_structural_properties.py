#!/usr/bin/env python
"""The hot-path part of the reference's `examples/Compute structural properties.py`
(lines 58-118: RDF, BAD, coordination number, MSD) with the MI355X classes.

    python examples/compute_structural_properties.py [out_dir]

Only the imports differ from the reference script; ASE is replaced by the native extended-XYZ
reader because it is not installed here (a list of ase.Atoms works just as well).
"""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from amof_amd.trajectory import read_lammps_traj         # noqa: E402
from amof_amd.rdf import Rdf                             # noqa: E402
from amof_amd.bad import Bad                             # noqa: E402
from amof_amd.cn import CoordinationNumber               # noqa: E402
from amof_amd.msd import WindowMsd                       # noqa: E402


def main(out_dir):
    os.makedirs(out_dir, exist_ok=True)
    # one frame, like the reference example (native extended-XYZ reader; the Lattice= comment carries the cell)
    traj = read_lammps_traj(os.path.join(ROOT, "tests", "golden", "ZIF-4.xyz"), ":")

    rdf = Rdf.from_trajectory(traj)                                              # reference :58
    print(rdf.data[["r", "X-X", "Zn-N"]].iloc[195:206].to_string())
    rdf.write_to_file(os.path.join(out_dir, "zif4"))                             # reference :75-79
    assert np.allclose(Rdf.from_file(os.path.join(out_dir, "zif4")).data, rdf.data)

    bad = Bad.from_trajectory(traj, {'Zn-N': 2.5})                               # reference :89
    sel = bad.data[(bad.data.theta > 100) & (bad.data.theta < 120)]
    print("N-Zn-N angles: density peaks at %.2f degrees" % sel.theta.values[np.argmax(sel["N-Zn-N"].values)])

    cn = CoordinationNumber.from_trajectory(traj, {'Zn-N': 2.5})                 # reference :100
    print(cn.data.to_string())

    rng = np.random.default_rng(0)                                               # reference :110-118: 11 rattled frames
    base = traj.frame(0)                                                         # an ase.Atoms-like Frame
    mock = [base.copy() for _ in range(11)]
    for k in range(1, 11):
        mock[k].positions = mock[k - 1].positions + rng.normal(scale=0.5, size=mock[k].positions.shape)
    msd = WindowMsd.from_trajectory(mock, delta_time=1, timestep=1)
    print(msd.data.to_string())


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/tmp/amof_amd_example")
