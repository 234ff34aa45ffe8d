"""Stream plumbing for hipGraph capture (graph_step.py, bench.py --graph, the capture tests).

What ROCm 7.0's graphs can and cannot do shaped how this package uses them (profiles/r04_graph_capture_notes.txt):

* A captured graph with fork / join branches replays, but its launch costs the HOST 6-17 ms (the runtime walks the branches
  and synchronises them from the host), and two forks that wait on EACH OTHER -- a side stream started behind the
  teacher's stream and joined back into it -- put each stream into the other's "parallel capture streams" list, on which
  hipStreamEndCapture recurses without end: a stack overflow inside hip::Stream::EndCapture.  A SINGLE-STREAM graph of the
  same ~1000 kernels launches in 0.3 ms.
* So nothing forks while a capture is in progress (`may_fork` is false): the captured pieces are single-stream graphs, and
  the overlap the eager steps get from side streams comes from replaying two such graphs on two streams (graph_step.py).
* Events recorded during a capture are kept alive until it has ended (torch's Stream.wait_stream drops its temporary
  event at once; the runtime keeps the capture's events in a list it visits at the end).  Not observed to fault; it costs
  a list of handles.
"""
import contextlib

import torch

_KEPT = None        # events of the capture in progress (None: no capture)


def capturing():
    return _KEPT is not None


def event(**kw):
    """torch.cuda.Event(**kw), kept alive until the end of the capture in progress (if any)."""
    ev = torch.cuda.Event(**kw)
    if _KEPT is not None:
        _KEPT.append(ev)
    return ev


def may_fork(device):
    """May work be queued on a side stream behind the current stream of `device`?  Not while a capture is in progress."""
    return _KEPT is None


@contextlib.contextmanager
def capture(graph, device=None, pool=None):
    """`with torch.cuda.graph(graph, pool=pool)` during which may_fork() is false and every event stays alive."""
    global _KEPT
    if _KEPT is not None:
        raise RuntimeError("geot_amd.streams.capture: a capture is already in progress")
    orig = torch.cuda.Stream.record_event

    def record_event(self, event=None):
        ev = orig(self, event)
        if _KEPT is not None:
            _KEPT.append(ev)
        return ev
    _KEPT = []
    torch.cuda.Stream.record_event = record_event
    import gc
    gc.collect()     # dead cycles may hold an earlier backward's AccumulateGrad nodes (another stream): graph_step._run has the story
    try:
        with torch.cuda.graph(graph, pool=pool):
            yield
    finally:
        torch.cuda.Stream.record_event = orig
        kept, _KEPT = _KEPT, None
        del kept[:]


# ---- what a captured graph is made of --------------------------------------------------------------------------------------
NODE_TYPE_NAMES = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "waitEvent", 7: "eventRecord",
                   8: "extSemSignal", 9: "extSemWait", 10: "memAlloc", 11: "memFree", 12: "memcpyFromSymbol",
                   13: "memcpyToSymbol"}            # hipGraphNodeType
_hip = None


def node_types(graph):
    """{"kernel": n, "memset": n, ...} of a torch.cuda.CUDAGraph constructed with keep_graph=True (hipGraphGetNodes /
    hipGraphNodeGetType on its hipGraph_t)."""
    global _hip
    import collections
    import ctypes
    import os
    if _hip is None:
        _hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    raw = ctypes.c_void_p(graph.raw_cuda_graph())
    n = ctypes.c_size_t(0)
    if _hip.hipGraphGetNodes(raw, None, ctypes.byref(n)) != 0:
        raise RuntimeError("hipGraphGetNodes failed")
    nodes = (ctypes.c_void_p * max(n.value, 1))()
    if _hip.hipGraphGetNodes(raw, nodes, ctypes.byref(n)) != 0:
        raise RuntimeError("hipGraphGetNodes failed")
    kinds = collections.Counter()
    for nd in nodes[:n.value]:
        t = ctypes.c_int(-1)
        if _hip.hipGraphNodeGetType(ctypes.c_void_p(nd), ctypes.byref(t)) != 0:
            raise RuntimeError("hipGraphNodeGetType failed")
        kinds[NODE_TYPE_NAMES.get(t.value, str(t.value))] += 1
    return dict(kinds)
