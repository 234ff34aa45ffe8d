"""The training steps of train_step.py replayed from a hipGraph: the host queues ONE graph launch per iteration instead of
the ~1100 (supervised) / ~1500 (FixMatch+NTM) kernel launches, autograd bookkeeping included, that take 21-28 ms of host
time per step -- at the authors' own operating point (16 000 points, 2 + 2 clouds) the whole step.

    step    = SupervisedStep(model)                      # or build_fixmatch(...)
    graphed = GraphedSupervisedStep(step)                # or GraphedFixMatchStep(step)
    loss    = graphed(pos, cls, target, next_pos=...)    # same call, same results, bit for bit

What is captured is the step's own `iteration()` -- forward, loss, backward, AdamW, the side-stream index plan, the
teacher's stream and the look-ahead, as fork / joins of one graph -- over FIXED buffers:

* the batch is copied into static input tensors before every replay (a few small device-to-device copies);
* the look-ahead's product -- the geometry of the NEXT batch (Group, the 8192-sample FPS, the index plan) -- is written by
  the graph into buffers of its own and copied into a static geometry at the graph's tail, behind everything that read the
  current one; the next replay consumes it.  Whether the static geometry describes the batch a call passes is checked on
  the host (the tensor the previous call announced, unedited); otherwise it is recomputed before the replay;
* the optimisers run with capturable=True (the step counters live on the device either way under fused=True: the same
  kernel, the same arithmetic), the EMA transition matrix is updated in its buffer.

The first `warmup` calls run the same iteration eagerly over the same buffers (library handles, workspaces, lazily
initialised state), the next call captures and replays.  Shapes are fixed at the first call; a batch of another shape is an
error (build another wrapper).  DistributedDataParallel is not captured: its bucket hooks and RCCL's collectives are host
logic (train_step.ddp() wraps eagerly; graph_step refuses a DDP-wrapped model).

Mirrors the loop body of examples/segmentation/train.py:410-669; the reference has no counterpart (it never leaves eager
mode) -- this is the MI355X answer to a step that is host-bound at the reference's own batch sizes.
"""
import copy

import torch

from . import streams
from .fused_norm import ReverseIndex


# ---- structure helpers: geometries are dicts / tuples of tensors, ReverseIndex objects, events and python scalars ------
def tree_clone(obj):
    """Fresh buffers with the same contents (events dropped)."""
    if torch.is_tensor(obj):
        return obj.detach().clone()
    if isinstance(obj, dict):
        return {k: tree_clone(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(tree_clone(v) for v in obj)
    if isinstance(obj, ReverseIndex):
        new = copy.copy(obj)
        new.ws, new.order = tree_clone(obj.ws), tree_clone(obj.order)
        return new
    if isinstance(obj, torch.cuda.Event):
        return None
    return obj


def tree_copy_(dst, src, path="geometry"):
    """Copy every tensor of `src` into the matching buffer of `dst`; the structure (and every python scalar) must agree."""
    if torch.is_tensor(dst):
        if not torch.is_tensor(src) or dst.shape != src.shape or dst.dtype != src.dtype:
            raise RuntimeError("%s: the captured buffer does not fit (shape / dtype changed)" % path)
        if dst.data_ptr() != src.data_ptr():
            dst.copy_(src)
    elif isinstance(dst, dict):
        for k, v in dst.items():
            if k in ("pts", "version", "grouped", "src", "static"):
                continue
            tree_copy_(v, src[k], "%s[%r]" % (path, k))
    elif isinstance(dst, (list, tuple)):
        if len(dst) != len(src):
            raise RuntimeError("%s: structure changed" % path)
        for i, (d, s) in enumerate(zip(dst, src)):
            tree_copy_(d, s, "%s[%d]" % (path, i))
    elif isinstance(dst, ReverseIndex):
        if (dst.b, dst.n, dst.m, dst.nt, dst.ws_ints) != (src.b, src.n, src.m, src.nt, src.ws_ints):
            raise RuntimeError("%s: reverse index of another shape" % path)
        tree_copy_(dst.ws, src.ws, path + ".ws")
        if dst.order is not None:
            tree_copy_(dst.order, src.order, path + ".order")
    elif dst is None:
        if src is not None and not isinstance(src, torch.cuda.Event):
            raise RuntimeError("%s: structure changed" % path)
    elif dst != src:
        raise RuntimeError("%s: %r became %r" % (path, dst, src))


def _static_geometry(g, pts):
    """A geometry in buffers of its own that the model takes on the caller's word (transformer.py `static`)."""
    if g is None:
        return None
    out = tree_clone({k: v for k, v in g.items() if k not in ("pts", "version", "grouped", "src")})
    out.update(pts=pts, version=None, grouped=None, static=True)
    return out


class _Graphed:
    def __init__(self, step, warmup=3):
        for net in self._modules(step):
            if isinstance(net, torch.nn.parallel.DistributedDataParallel):
                raise RuntimeError("graph_step: a DistributedDataParallel model is not captured (its gradient buckets and "
                                   "collectives are host logic); run N > 1 eagerly")
        self.step = step
        self.warmup = int(warmup)
        self.calls = 0
        self.graphs = {}                     # variant (look-ahead or not) -> (CUDAGraph, static outputs)
        self._stream = None
        self._pool = None
        for opt in step.optimizers():
            for group in opt.param_groups:
                group["capturable"] = True   # fused AdamW: the step counter is a device tensor already; same kernel

    @staticmethod
    def _modules(step):
        return [m for m in (getattr(step, "model", None), getattr(step, "model_t", None), getattr(step, "T_predictor", None))
                if m is not None]

    def _run(self, variant, fn):
        """fn() = the iteration over the static buffers, returning its static outputs.  Eager for the first `warmup` calls
        (on a side stream, torch's capture recipe), then captured once per variant and replayed."""
        dev = self.device
        if variant not in self.graphs and self.calls < self.warmup:
            if self._stream is None:
                self._stream = torch.cuda.Stream(device=dev)
            main = torch.cuda.current_stream(dev)
            self._stream.wait_stream(main)
            with torch.cuda.stream(self._stream):
                out = fn()
            main.wait_stream(self._stream)
            self.calls += 1
            return out
        if variant not in self.graphs:
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()
            if self._pool is None:
                self._pool = torch.cuda.graph_pool_handle()
            with streams.capture(graph, dev, pool=self._pool):
                out = fn()
            self.graphs[variant] = (graph, out)
        graph, out = self.graphs[variant]
        graph.replay()
        self.calls += 1
        return out

    @property
    def captured(self):
        return bool(self.graphs)


class GraphedSupervisedStep(_Graphed):
    """SupervisedStep.__call__ from a hipGraph (see the module docstring).  The returned loss is a static tensor the next
    call overwrites: clone it to keep it."""

    def __init__(self, step, warmup=3):
        super().__init__(step, warmup)
        self.x = None            # static (pos, cls, target)
        self.next_pos = None     # static coordinates of the announced batch
        self.geometry = None     # static geometry of x's coordinates
        self._announced = None   # (tensor, version) the static geometry describes

    def _inner(self):
        m = self.step.model
        return m.module if hasattr(m, "module") else m

    def __call__(self, pos, cls, target, next_pos=None):
        inner = self._inner()
        look = next_pos is not None and hasattr(inner, "prefetch_geometry")
        if self.x is None:
            self.device = pos.device
            self.x = (pos.detach().clone().contiguous(), cls.detach().clone(), target.detach().clone())
            self.next_pos = torch.empty_like(self.x[0])
        for dst, src, name in zip(self.x, (pos, cls, target), ("pos", "cls", "target")):
            if dst.shape != src.shape or dst.dtype != src.dtype:
                raise RuntimeError("graphed step: %s is %s %s, captured for %s %s" % (name, tuple(src.shape), src.dtype,
                                                                                      tuple(dst.shape), dst.dtype))
        fresh = not (self._announced is not None and pos is self._announced[0] and pos._version == self._announced[1])
        self.x[0].copy_(pos)
        self.x[1].copy_(cls)
        self.x[2].copy_(target)
        geometry = None
        if look:
            if fresh or self.geometry is None:
                # not the batch the previous call announced (the first call, a reshuffle): its geometry now, in line
                from .train_step import _mode
                _mode(self.step.model, True)
                g = inner.prefetch_geometry(self.x[0])
                if g is None:
                    look = False
                else:
                    torch.cuda.current_stream(self.device).wait_stream(inner_side(inner, self.device))
                    if self.geometry is None:
                        self.geometry = _static_geometry(g, self.x[0])
                    else:
                        tree_copy_(self.geometry, g)
            geometry = self.geometry if look else None
        if look:
            self.next_pos.copy_(next_pos)
        self._announced = None

        def iteration():
            loss, g_next = self.step.iteration(self.x[0], self.x[1], self.x[2], geometry, self.next_pos if look else None,
                                               static=True)
            if look:
                with torch.no_grad():
                    tree_copy_(self.geometry, g_next)      # behind the joins: every reader of the current one is done
            return loss
        loss = self._run("lookahead" if look else "plain", iteration)
        if look:
            self._announced = (next_pos, next_pos._version)
        return loss


def inner_side(segmentor_or_wrapper, device):
    seg = getattr(segmentor_or_wrapper, "segmentor", segmentor_or_wrapper)
    return seg._side_stream(device)


_NEXT_KEYS = (("pos",), ("pos_s", "pos_w"))


class GraphedFixMatchStep(_Graphed):
    """FixMatchNTMStep.__call__ from a hipGraph.  The returned losses are static tensors the next call overwrites."""

    def __init__(self, step, warmup=3):
        super().__init__(step, warmup)
        self.data = self.data_u = None       # static batch dicts
        self.next = None                     # static coordinates of the announced batches
        self.geometry = None                 # [student, teacher] static geometries
        self._announced = None

    def _copy_in(self, dst, src, what):
        for k, v in dst.items():
            s = src[k]
            if v.shape != s.shape or v.dtype != s.dtype:
                raise RuntimeError("graphed FixMatch step: %s[%r] is %s, captured for %s" % (what, k, tuple(s.shape), tuple(v.shape)))
            v.copy_(s)

    def __call__(self, data, data_u, next_batches=None):
        step = self.step
        inner = step.model.module if hasattr(step.model, "module") else step.model
        look = next_batches is not None
        if self.data is None:
            self.device = data["pos"].device
            self.data = {k: v.detach().clone().contiguous() for k, v in data.items() if torch.is_tensor(v)}
            self.data_u = {k: v.detach().clone().contiguous() for k, v in data_u.items() if torch.is_tensor(v) and k != "T"}
            self.next = ({"pos": torch.empty_like(self.data["pos"])},
                         {"pos_s": torch.empty_like(self.data_u["pos_s"]), "pos_w": torch.empty_like(self.data_u["pos_w"])})
        src = (data["pos"], data_u["pos_s"], data_u["pos_w"])
        fresh = not (self._announced is not None and all(t is a for t, a in zip(src, self._announced[0]))
                     and all(t._version == v for t, v in zip(src, self._announced[1])))
        self._copy_in(self.data, data, "data")
        self._copy_in(self.data_u, data_u, "data_u")
        geoms = (None, None)
        if look:
            if fresh or self.geometry is None:
                from .train_step import _mode
                _mode(step.model, True)
                _mode(step.model_t, False)
                g_s = inner.prefetch_geometry(self.data, self.data_u, fixmatch=True)
                g_t = step.model_t.prefetch_geometry(self.data_u, if_teacher=True)
                if g_s is None or g_t is None:
                    look = False
                else:
                    main = torch.cuda.current_stream(self.device)
                    main.wait_stream(inner_side(inner, self.device))
                    main.wait_stream(inner_side(step.model_t, self.device))
                    if self.geometry is None:
                        self.geometry = [_static_geometry(g_s, None), _static_geometry(g_t, None)]
                    else:
                        tree_copy_(self.geometry[0], g_s)
                        tree_copy_(self.geometry[1], g_t)
            if look:
                geoms = tuple(self.geometry)
                for dst, batch, keys in zip(self.next, next_batches, _NEXT_KEYS):
                    for k in keys:
                        dst[k].copy_(batch[k])
        self._announced = None

        def iteration():
            losses, g_next = step.iteration(self.data, self.data_u, geoms, self.next if look else None, static=True)
            if look:
                with torch.no_grad():
                    tree_copy_(self.geometry[0], g_next[0])
                    tree_copy_(self.geometry[1], g_next[1])
            return losses
        losses = self._run("lookahead" if look else "plain", iteration)
        if look:
            nsrc = (next_batches[0]["pos"], next_batches[1]["pos_s"], next_batches[1]["pos_w"])
            self._announced = (nsrc, tuple(t._version for t in nsrc))
        return losses
