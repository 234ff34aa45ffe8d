"""Tensor validation + stream plumbing shared by the extension-shaped modules.

Error behaviour: every violated precondition raises RuntimeError (the reference
mixes AT_ASSERT -> RuntimeError, exit(-1) and no checks at all; we never exit and
never silently read a bad tensor).  CPU tensors are rejected like the
reference's AT_ASSERT(false, "CPU not supported") -- there is no CPU path.
"""
import os

import torch

from .. import _lib


def req(t, name, dtype, ndim=None):
    if not isinstance(t, torch.Tensor):
        raise RuntimeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s: CPU not supported (tensor must live on the GPU)" % name)
    if t.dtype != dtype:
        raise RuntimeError("%s must be a %s tensor, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise RuntimeError("%s must be a contiguous tensor" % name)
    if ndim is not None and t.dim() != ndim:
        raise RuntimeError("%s must have %d dimensions, got %d" % (name, ndim, t.dim()))
    return t


def f32(t, name, ndim=None):
    return req(t, name, torch.float32, ndim)


def i32(t, name, ndim=None):
    return req(t, name, torch.int32, ndim)


def same_device(*ts):
    dev = ts[0].device
    for t in ts[1:]:
        if t.device != dev:
            raise RuntimeError("all tensors must be on the same device (%s vs %s)" % (dev, t.device))
    return dev


def need(cond, msg):
    if not cond:
        raise RuntimeError(msg)


def check_index(idx, upper, name):
    """GEOT_DEBUG=1: verify 0 <= idx < upper before a gather-type launch (one host sync; off by default).
    The reference never checks, and an out-of-range neighbour index is an out-of-bounds read on the GPU --
    this turns it into an IndexError while a caller's index pipeline is being brought up."""
    if os.environ.get("GEOT_DEBUG", "0") != "1" or idx.numel() == 0:
        return
    lo, hi = int(idx.min().item()), int(idx.max().item())
    if lo < 0 or hi >= int(upper):
        raise IndexError("%s holds indices in [%d, %d], valid range is [0, %d)" % (name, lo, hi, int(upper)))


def stream_of(dev):
    return torch.cuda.current_stream(dev).cuda_stream


trace = None     # bench.py's attribution pass sets this to a callable(launch, name) that brackets the launch

# How a launch reaches the C ABI: through the compiled dispatcher (csrc_torch/gen_dispatch.py -> _geot_dispatch_cpp.so:
# one pybind11 forwarder per entry point; the current stream and the device guard are taken in C++) when it can be built
# and loaded, through ctypes otherwise (no compiler, another GEOT_DISTANCE than the one the module is linked against).
# GEOT_BINDING=ctypes forces the fallback, =cpp makes a missing dispatcher an error.  Same library, same arguments.
BINDING = os.environ.get("GEOT_BINDING", "auto")
_dispatch = None          # the module, False when unavailable, None before the first launch


def dispatcher():
    """The compiled dispatcher module, or None when launches go through ctypes."""
    global _dispatch
    if _dispatch is None:
        _dispatch = False
        if BINDING not in ("auto", "cpp", "ctypes"):
            raise RuntimeError("GEOT_BINDING must be auto, cpp or ctypes, got %r" % BINDING)
        if BINDING != "ctypes":
            try:
                from .. import build_torch_ext
                _dispatch = build_torch_ext.load("_geot_dispatch_cpp")
            except Exception as e:      # noqa: BLE001 -- no compiler / headers / matching library: ctypes serves
                if BINDING == "cpp":
                    raise RuntimeError("GEOT_BINDING=cpp: the compiled dispatcher is unavailable: %s" % e)
                import sys
                sys.stderr.write("geot_amd: the compiled dispatcher is unavailable (%s: %s); launches go through ctypes\n"
                                 % (type(e).__name__, str(e)[:300]))
                _dispatch = False
    return _dispatch or None


def call(name, dev, *args):
    """Launch C-ABI entry `name` on torch's current stream of `dev` (under a device guard when `dev` is not
    the current device; the composite steps are launch-bound at one cloud, so the common case stays lean)."""
    if trace is not None:
        return trace(lambda: _launch(name, dev, *args), name)
    return _launch(name, dev, *args)


def _launch(name, dev, *args):
    disp = _dispatch if _dispatch is not None else dispatcher()
    if disp:
        fn = getattr(disp, name, None)
        if fn is not None:
            err = fn(dev.index if dev.index is not None else torch.cuda.current_device(), *args)
            if err:
                _lib.check(err, name)
            return
    lib = _lib.load()
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx == torch.cuda.current_device():
        err = getattr(lib, name)(*args, torch.cuda.current_stream(idx).cuda_stream)
    else:
        with torch.cuda.device(idx):
            err = getattr(lib, name)(*args, torch.cuda.current_stream(idx).cuda_stream)
    if err:
        _lib.check(err, name)


def ptr(t):
    return t.data_ptr() if t is not None else None


def knn_workspace(dev, b, nq, nr, k):
    """(ptr, bytes) of scratch for the grid kNN path, or (None, 0) when the sizes do not qualify
    (the C entry point then runs the brute-force kernel).  The tensor is returned too so that it stays
    alive until the launch has been queued (torch's allocator is stream-ordered after that)."""
    lib = _lib.load()
    if not lib.geot_knn_grid_eligible(int(b), int(nq), int(nr), int(k)):
        return None, 0, None
    nbytes = int(lib.geot_knn_grid_ws_bytes(int(b), int(nr)))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    return ws.data_ptr(), nbytes, ws


def ball_workspace(dev, b, n, m, radius, nsample):
    """(ptr, bytes, keep-alive tensor) of scratch for the grid ball query, or (None, 0, None)."""
    lib = _lib.load()
    if not lib.geot_ball_grid_eligible(int(b), int(n), int(m), float(radius), int(nsample)):
        return None, 0, None
    nbytes = int(lib.geot_knn_grid_ws_bytes(int(b), int(n)))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    return ws.data_ptr(), nbytes, ws


def grad_needs_atomics(b, c, m, n_sources, slots):
    """True for the shapes the atomic-free gradient forms do not take (the *_grad_ws calls then accumulate with float
    atomics in a zero-filled channels-last workspace; with fewer than 16 channels the direct scatter is the better one)."""
    return bool(_lib.load().geot_grad_ws_needs_zero(int(b), int(c), int(m), int(n_sources), int(slots)))


def grad_workspace(dev, b, c, m, n_sources, slots, weighted=None):
    """Float scratch for the *_grad_ws entry points (geot_scatter_grad_ws_floats): the sorted pair stream of
    csrc/tile_scatter.hip, or -- for shapes it does not take -- the (B, m, C) channels-last accumulator, zero-filled
    only then."""
    lib = _lib.load()
    weighted = (slots == 3) if weighted is None else weighted
    floats = int(lib.geot_scatter_grad_ws_floats(int(b), int(c), int(m), int(n_sources), int(slots), int(bool(weighted))))
    need_zero = lib.geot_grad_ws_needs_zero(int(b), int(c), int(m), int(n_sources), int(slots))
    alloc = torch.zeros if need_zero else torch.empty
    return alloc(max(floats, 1), dtype=torch.float32, device=dev)
