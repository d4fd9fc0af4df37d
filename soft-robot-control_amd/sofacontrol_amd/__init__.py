"""sofacontrol_amd -- MI355X-native drop-in for the sofacontrol ROM + optimal-control hot path.

Mirrors the module layout of the reference package `sofacontrol` (mor.pod, tpwl.tpwl, scp.gusto,
scp.locp, scp.standalone, scp.models.tpwl, lqr.ilqr, lqr.lqr, lqr.traj_tracking_lqr, utils) and keeps
its class / method surface; all arithmetic runs in libsofacontrol_hip.so (hand-written HIP for gfx950,
C ABI in include/sofacontrol_hip.h).  There is no CPU fallback: importing works anywhere (the library
cross-compiles without a GPU) but every compute call raises if the library or the GPU is missing.
"""
from . import _lib  # noqa: F401

__all__ = ['_lib']
