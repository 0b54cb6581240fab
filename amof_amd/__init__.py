"""amof_amd -- MI355X-native kernels behind aMOF's pair-distance analysis API.

Drop-in mirrors of the reference's hot-path classes (coudertlab/amof v1.1.0):

    amof.rdf.Rdf                  -> amof_amd.rdf.Rdf
    amof.msd.WindowMsd            -> amof_amd.msd.WindowMsd
    amof.bad.Bad                  -> amof_amd.bad.Bad
    amof.cn.CoordinationNumber    -> amof_amd.cn.CoordinationNumber

All distance arithmetic runs in hand-written HIP kernels (gfx950) behind the C
ABI of ``include/amof_hip.h``; there is no CPU fallback.
"""

__version__ = "0.1.0"

from .frames import Frame, PackedTrajectory, pack_trajectory  # noqa: F401
