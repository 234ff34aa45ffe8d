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
    try:
        with torch.cuda.graph(graph, pool=pool):
            yield
    finally:
        torch.cuda.Stream.record_event = orig
        kept, _KEPT = _KEPT, None
        del kept[:]
