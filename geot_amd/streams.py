"""Stream plumbing for hipGraph capture (graph_step.py, bench.py --graph, the capture tests).

Two properties of ROCm 7.0's stream capture shape how this package forks work onto side streams while capturing:

* **Forks only from the capture's origin stream.**  hipStreamWaitEvent re-registers every NON-origin stream that waits on a
  captured event as a "parallel capture stream" of the stream the event was recorded on.  Two forks that wait on each other
  (a side stream started behind the teacher's stream and joined back into it) are then in each other's lists, and
  hipStreamEndCapture -- which ends the capture on every stream of those lists, recursively -- never returns: the process
  dies of a stack overflow inside hip::Stream::EndCapture (native backtrace: profiles/r04_graph_capture_notes.txt).  Waits
  between the origin and a fork are fine in both directions.  `may_fork(dev)` is therefore false on any stream but the
  origin while a capture is in progress, and the segmentor keeps its sampling in line there.
* Events recorded during a capture are kept alive until it has ended (torch's Stream.wait_stream drops its temporary event
  at once; the runtime keeps a list of the capture's events and visits it at the end).  Not observed to fault; it costs a
  list of a few dozen handles.
"""
import contextlib

import torch

_KEPT = None        # events of the capture in progress (None: no capture)
_ORIGIN = None      # the stream the capture in progress was begun on


def event(**kw):
    """torch.cuda.Event(**kw), kept alive until the end of the capture in progress (if any)."""
    ev = torch.cuda.Event(**kw)
    if _KEPT is not None:
        _KEPT.append(ev)
    return ev


def may_fork(device):
    """May work be queued on a side stream behind the current stream of `device`?  Always outside a capture; inside one
    only from the origin stream (see the module docstring)."""
    return _ORIGIN is None or torch.cuda.current_stream(device) == _ORIGIN


@contextlib.contextmanager
def capture(graph, device, pool=None):
    """`with torch.cuda.graph(graph, pool=pool)` that records the origin stream for may_fork() and keeps the capture's
    events alive until hipStreamEndCapture has returned."""
    global _KEPT, _ORIGIN
    if _ORIGIN is not None:
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
            _ORIGIN = torch.cuda.current_stream(device)
            try:
                yield
            finally:
                _ORIGIN = None
    finally:
        torch.cuda.Stream.record_event = orig
        kept, _KEPT = _KEPT, None
        del kept[:]
