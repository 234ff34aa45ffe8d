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


# hipGraph replays (geot_amd/graph_step.py) and the runtime's "graph packet capture" (on by default in ROCm 7.0): with it on,
# the AQL packets of a graph's hipMemsetAsync / hipMemcpyAsync NODES keep pointing at kernel-argument slots of the device's
# shared ring, and a few thousand eager launches between two replays recycle those slots -- the memsets then clear something
# else, and every torch reduction that zeroes its semaphores that way returns garbage, silently
# (profiles/r04_graph_capture_notes.txt has the reproducer).  Graphs of kernel nodes alone are not affected.  Two modes,
# chosen by GEOT_GRAPH_LAUNCH before the HIP runtime initialises (the switch is read then):
#   safe (default)  packet capture is turned OFF here, at import, unless the process has chosen a value itself; any graph
#                   replays correctly, a launch costs the host 7-20 ms (the runtime re-encodes the graph's packets);
#   fast            packet capture stays ON (0.5 ms per launch); graph_step then inspects every graph it captures
#                   (hipGraphGetNodes) and REFUSES one that holds anything but kernel nodes.  The training steps of this
#                   package capture kernel-only (tests/test_graph_step_gpu.py); code added around them may not.
# graph_step refuses to capture when neither holds (HIP already initialised at import, or the variable overridden).
GRAPH_PACKET_CAPTURE_ENV = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"
GRAPH_LAUNCH = _os.environ.get("GEOT_GRAPH_LAUNCH", "safe")
if GRAPH_LAUNCH not in ("safe", "fast"):
    raise ImportError("GEOT_GRAPH_LAUNCH must be 'safe' or 'fast', not %r" % GRAPH_LAUNCH)
_GRAPH_ENV_SET_IN_TIME = not _hip_runtime_up()
if _GRAPH_ENV_SET_IN_TIME and GRAPH_LAUNCH == "safe":
    _os.environ.setdefault(GRAPH_PACKET_CAPTURE_ENV, "0")


def graph_replay_is_safe():
    """True when hipGraph packet capture is off for this process (see above): any captured graph replays correctly."""
    return _os.environ.get(GRAPH_PACKET_CAPTURE_ENV) == "0" and _GRAPH_ENV_SET_IN_TIME
