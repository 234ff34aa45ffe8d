"""geot_amd -- MI355X (gfx950) implementation of GeoT's point-cloud sampling / grouping /
interpolation hot path behind the reference's operator API.

    from geot_amd.pointnet2 import pointnet2_utils          # furthest_point_sample, ball_query, ...
    from geot_amd.pointops.functions import pointops        # fps, knn, ...
    from geot_amd.knn_cuda import KNN
    import geot_amd.aliases; geot_amd.aliases.install()     # reference import names -> this package

The compute path is libgeot_hip.so (C ABI in include/geot_hip.h); there is no CPU fallback.
"""
__version__ = "0.1.0"

import os as _os
import sys as _sys


def _hip_runtime_up():
    t = _sys.modules.get("torch")
    return bool(t is not None and hasattr(t, "cuda") and t.cuda.is_initialized())


# hipGraph replays (geot_amd/graph_step.py) need the runtime's "graph packet capture" OFF: with it on (the default of ROCm
# 7.0), the AQL packets of a graph's hipMemsetAsync nodes keep pointing at kernel-argument slots of the device's shared
# ring, and a few thousand eager launches between two replays recycle those slots -- the memsets then clear something
# else, and every torch reduction that zeroes its semaphores that way returns garbage, silently
# (profiles/r04_graph_capture_notes.txt has the reproducer).  The switch is read when the HIP runtime initialises, so it is
# set here, at import, unless the process has chosen a value itself; graph_step refuses to capture when it could not
# take effect (HIP already initialised) or was overridden to anything but 0.
GRAPH_PACKET_CAPTURE_ENV = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"
_GRAPH_ENV_SET_IN_TIME = not _hip_runtime_up()
if _GRAPH_ENV_SET_IN_TIME:
    _os.environ.setdefault(GRAPH_PACKET_CAPTURE_ENV, "0")


def graph_replay_is_safe():
    """True when hipGraph packet capture is off for this process (see above)."""
    return _os.environ.get(GRAPH_PACKET_CAPTURE_ENV) == "0" and _GRAPH_ENV_SET_IN_TIME
