"""The training steps of train_step.py replayed from hipGraphs: per iteration the host issues a handful of small copies and
TWO graph launches instead of ~1100 (supervised) / ~1500 (FixMatch+NTM) kernel launches with their autograd bookkeeping
(18-24 ms of host time per iteration -- at one or two clouds per step, the whole step).

    step    = SupervisedStep(model)                      # or build_fixmatch(...)
    graphed = GraphedSupervisedStep(step)                # or GraphedFixMatchStep(step)
    loss    = graphed(pos, cls, target, next_pos=...)    # same call, same results, bit for bit

An iteration is two SINGLE-STREAM graphs (geot_amd/streams.py: on ROCm 7.0 a graph with fork / join branches costs the host
6-17 ms per launch, a single-stream one of the same kernels 0.3 ms):

* **P** -- the step's `lookahead_work()`: everything that depends on a batch and on frozen state alone.  Supervised step:
  the batch's geometry (Group, the 8192-sample FPS, the index plan).  FixMatch+NTM: the student's and the teacher's
  geometry, the frozen teacher's forward -> pseudo labels, the kNN graph and Morton order of the 3-D loss.
* **M** -- the training iteration proper over P's product: forward, losses, backward, AdamW (capturable=True; under
  fused=True the step counters live on the device either way: the same kernel), the EMA transition matrix updated in
  its buffer.

With `next_pos` / `next_batches` given, P of the NEXT batch replays on a side stream beside M of the current one -- the
overlap the eager steps build from side streams inside the iteration, here between two graphs; its product is copied into
M's input buffers at the head of the next call, behind M.  Whether that product describes the batch a call passes is
checked on the host (the very tensors the previous call announced, unedited); otherwise -- the first call, a reshuffled
loader, no look-ahead at all -- P runs for the current batch on the current stream before M.

A learning-rate schedule reaches a captured AdamW step through memory: make `lr` a float32 device tensor in the optimizer's
parameter groups and fill_ it between calls (tests/test_graph_step_gpu.py); a python float is baked into the graph.
Inputs are copied into static buffers before every replay; shapes are fixed by the first call (another shape is an error:
build another wrapper).  The first `warmup` executions of each graph's body run eagerly over the same buffers, the next
one captures.  N > 1: DistributedDataParallel itself is not captured (its reducer is host logic); a step built on the bare
SyncBatchNorm-converted modules with a gradient exchange of its own (train_step.sync_only / GradSync: one flat RCCL
all-reduce between backward and optimizer, a kernel node) replays like the single-GPU one -- SyncBatchNorm's statistic
all-reduces and the anchor all-gather of the class transition are captured where they stand.

Mirrors the loop body of examples/segmentation/train.py:410-669; the reference never leaves eager mode.
"""
import copy
import os

import torch

from . import streams
from .fused_norm import ReverseIndex

_IDENTITY_KEYS = ("pts", "version", "grouped", "src")     # of a geometry: tensor identities and events, never copied


# ---- structure helpers: P's product is a tree of dicts / tuples of tensors, ReverseIndex objects and python scalars ------
def tree_clone(obj):
    """Fresh buffers with the same contents."""
    if torch.is_tensor(obj):
        return obj.detach().clone()
    if isinstance(obj, dict):
        return {k: (None if k in _IDENTITY_KEYS else tree_clone(v)) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(tree_clone(v) for v in obj)
    if isinstance(obj, ReverseIndex):
        new = copy.copy(obj)
        new.ws, new.order = tree_clone(obj.ws), tree_clone(obj.order)
        return new
    return obj


def tree_detach(obj):
    """The same buffers without their autograd history (tensors only; containers rebuilt, everything else as it is)."""
    if torch.is_tensor(obj):
        return obj.detach()
    if isinstance(obj, dict):
        return {k: tree_detach(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(tree_detach(v) for v in obj)
    return obj


def tree_copy_(dst, src, path="pre"):
    """Copy every tensor of `src` into the matching buffer of `dst`; the structure (and every python scalar) must agree."""
    if torch.is_tensor(dst):
        if not torch.is_tensor(src) or dst.shape != src.shape or dst.dtype != src.dtype:
            raise RuntimeError("%s: the captured buffer does not fit (shape / dtype changed)" % path)
        dst.copy_(src)
    elif isinstance(dst, dict):
        for k, v in dst.items():
            if k not in _IDENTITY_KEYS:
                tree_copy_(v, src[k], "%s[%r]" % (path, k))
    elif isinstance(dst, (list, tuple)):
        if not isinstance(src, (list, tuple)) or len(dst) != len(src):
            raise RuntimeError("%s: structure changed" % path)
        for i, (d, s) in enumerate(zip(dst, src)):
            tree_copy_(d, s, "%s[%d]" % (path, i))
    elif isinstance(dst, ReverseIndex):
        if (dst.b, dst.n, dst.m, dst.nt, dst.ws_ints) != (src.b, src.n, src.m, src.nt, src.ws_ints):
            raise RuntimeError("%s: reverse index of another shape" % path)
        tree_copy_(dst.ws, src.ws, path + ".ws")
        if dst.order is not None:
            tree_copy_(dst.order, src.order, path + ".order")
    elif dst != src:
        raise RuntimeError("%s: %r became %r" % (path, dst, src))


def _fits(dst, src, what):
    if dst.shape != src.shape or dst.dtype != src.dtype:
        raise RuntimeError("graphed step: %s is %s %s, captured for %s %s" % (what, tuple(src.shape), src.dtype,
                                                                              tuple(dst.shape), dst.dtype))


class _Graphed:
    """P / M bookkeeping shared by the two steps; subclasses supply the static buffers and the two bodies."""

    def __init__(self, step, warmup=2, agree=None):
        """agree(ok: bool) -> bool, for N > 1: called after every capture, BEFORE its first replay, with this rank's verdict;
        returns the verdict of all ranks (e.g. an all-reduce MIN).  A capture one rank refuses is then refused by all of
        them at the same point of the collective sequence -- the ranks fall back to the eager step together instead of
        meeting each other with different collectives."""
        self.agree = agree
        for net in (getattr(step, n, None) for n in ("model", "model_t", "T_predictor")):
            if isinstance(net, torch.nn.parallel.DistributedDataParallel):
                raise RuntimeError(
                    "graph_step: a DistributedDataParallel model is not captured -- its reducer is host logic, and it keeps "
                    "every parameter's AccumulateGrad node on the stream of its construction, which forks the captured "
                    "backward (a crash in hipStreamEndCapture on this runtime).  For N > 1 hand the step the bare "
                    "SyncBatchNorm-converted modules and a gradient exchange of its own: train_step.sync_only + GradSync "
                    "(build_fixmatch(graph_sync=True)); over RCCL its one flat all-reduce is a kernel node of the graph")
        if getattr(step, "grad_sync", None) is not None:
            import torch.distributed as dist
            if dist.get_backend(step.grad_sync.group) != "nccl":
                raise RuntimeError("graph_step: the step's gradient exchange runs over the %r backend -- on the host, not "
                                   "capturable; run N > 1 eagerly or over nccl (RCCL)" % dist.get_backend(step.grad_sync.group))
        import geot_amd
        # (see geot_amd/__init__.py) packet capture exported off by the launcher: anything replays; otherwise (fast mode, or the
        # switch set by the package itself, which cannot be verified) only graphs of kernel nodes do -- checked per graph
        self.kernel_only = not geot_amd.graph_replay_is_safe()
        self.step = step
        self.warmup = int(warmup)
        self.calls = 0
        self.device = None
        self.graphs = {}          # "P" / "M" -> (CUDAGraph, static outputs)
        self.node_types = {}      # "P" / "M" -> {"kernel": n, ...} of the captured graph
        self._eager_runs = {"P": 0, "M": 0}
        self.side = None          # the stream P replays on beside M
        self.pre = None           # P's product in M's input buffers
        self._pre_next = None     # P's product as P left it (for the batch `_announced` describes)
        self._announced = None
        self._pending = False     # a P is in flight on the side stream
        for opt in step.optimizers():
            for group in opt.param_groups:
                group["capturable"] = True   # fused AdamW: the step counter is a device tensor already; same kernel
        # a python-float lr is baked into the captured AdamW step: remember it and refuse a call that finds another value
        # (a torch LR scheduler replaces group["lr"] every step -- the replay would silently train at the first one)
        self._float_lrs = [(group, group["lr"]) for opt in step.optimizers() for group in opt.param_groups
                           if not torch.is_tensor(group["lr"])]

    def _check_lr(self):
        for group, lr in self._float_lrs:
            if not torch.is_tensor(group["lr"]) and group["lr"] != lr:
                raise RuntimeError(
                    "graph_step: a parameter group's lr changed from %r to %r, but a python float is baked into the captured "
                    "AdamW step -- the replay would keep training at %r.  Make lr a float32 device tensor in the optimizer's "
                    "parameter groups before wrapping the step and fill_ it between calls (module docstring)" % (lr, group["lr"], lr))

    def _run(self, name, fn, pool_of=None):
        """fn() = a graph's body over the static buffers.  Eager for its first `warmup` executions, then captured once
        (into the memory pool of graph `pool_of`, if given) and replayed on the current stream."""
        if name not in self.graphs:
            if self._eager_runs.get(name, 0) < self.warmup:
                self._eager_runs[name] = self._eager_runs.get(name, 0) + 1
                return fn()
            torch.cuda.synchronize(self.device)
            # Dead python cycles can hold an earlier iteration's autograd graph -- eager steps taken before this wrapper, the
            # warm-up runs -- and with it every parameter's AccumulateGrad node, bound to the stream of THAT iteration; a
            # backward under capture would hop to that stream and back: a fork inside the capture, which this runtime answers
            # with a crash in hipStreamEndCapture (seen at 2 clouds after three eager steps; torch.cuda.graph collected
            # unconditionally until 2.x made it optional).  Collect before every capture: three times in a wrapper's life.
            import gc
            gc.collect()
            graph = torch.cuda.CUDAGraph(keep_graph=True)         # (the hipGraph_t stays: node_types below)
            pool = self.graphs[pool_of][0].pool() if pool_of is not None else None
            refusal = None
            try:
                with streams.capture(graph, self.device, pool=pool):
                    out = fn()
                self.node_types[name] = kinds = streams.node_types(graph)
                if self.kernel_only and set(kinds) - {"kernel"} and os.environ.get("GEOT_GRAPH_UNSAFE") != "1":
                    import geot_amd
                    refusal = RuntimeError(
                        "graph_step: graph %s holds %s; unless the launcher exported %s=0 (before the HIP runtime initialises) "
                        "only kernel nodes are known to replay correctly: with graph packet capture on, eager launches between "
                        "two replays corrupt a graph's memset / memcpy nodes -- wrong gradients, no error "
                        "(geot_amd/__init__.py).  tools/lab/find_nonkernel_ops.py names the operators that issue "
                        "hipMemsetAsync / hipMemcpyAsync" % (name, kinds, geot_amd.GRAPH_PACKET_CAPTURE_ENV))
            except RuntimeError as e:
                if self.agree is None:
                    raise
                refusal = e
            if self.agree is not None and not self.agree(refusal is None) and refusal is None:
                refusal = RuntimeError("graph_step: another rank refused its capture of graph %s; all ranks run eagerly" % name)
            if refusal is not None:
                del graph
                raise refusal
            self.graphs[name] = (graph, out)
        graph, out = self.graphs[name]
        graph.replay()
        return out

    @property
    def captured(self):
        return ("M" in self.graphs or "M2" in self.graphs) and "P" in self.graphs

    def _join_pending(self):
        """The current stream waits for the P of the previous call (its product, and its reads of P's static inputs)."""
        if self._pending:
            torch.cuda.current_stream(self.device).wait_stream(self.side)
            self._pending = False

    def _iterate(self, announced_now, next_src, load_next, lookahead, train):
        """announced_now: does the pending look-ahead describe the batch that was just copied in?  next_src: the tensors
        of the announced next batch (None: no look-ahead); load_next(current: bool): fill P's static inputs from the
        current batch / the announced one; lookahead() / train(): the bodies of P / M.  (_join_pending() has run.)"""
        dev = self.device
        main = torch.cuda.current_stream(dev)
        if self.side is None:
            self.side = torch.cuda.Stream(device=dev)
        if not (announced_now and self._pre_next is not None):
            # nobody looked ahead for this batch (first call, an unannounced batch, no look-ahead): P now, in line
            load_next(True)
            self._pre_next = self._run("P", lookahead)
        with torch.no_grad():
            if self.pre is None:
                self.pre = tree_clone(self._pre_next)
            else:
                tree_copy_(self.pre, self._pre_next)
        self._pre_next = None
        self._announced = None
        mid = None
        if isinstance(train, tuple):                 # (forward, backward): P starts between the two, beside the backward
            mid = self._run("M1", train[0])
        if next_src is not None:
            load_next(False)
            self.side.wait_stream(main)              # behind the copies above (and behind M's previous replay)
            with torch.cuda.stream(self.side):
                self._pre_next = self._run("P", lookahead)
            self._pending = True
            self._announced = tuple((t, t._version) for t in next_src)
        fresh = isinstance(train, tuple) and "M2" not in self.graphs
        out = self._run("M2", lambda: train[1](mid), pool_of="M1") if isinstance(train, tuple) else self._run("M", train)
        if fresh and "M2" in self.graphs:
            # M1's stored product (the cut: tensors with the capture-time iteration's autograd graph behind them) has done its
            # job -- M2 is captured and a replay never runs the python backward again.  Kept as it was, it would hold that
            # graph, every parameter's AccumulateGrad node with it (created on the capture stream), for the life of the
            # wrapper, and an eager step afterwards (bench.py's other leg) would find them on the wrong stream.
            g1, o1 = self.graphs["M1"]
            self.graphs["M1"] = (g1, tree_detach(o1))
        self.calls += 1
        return out

    def _is_announced(self, tensors):
        a = self._announced
        return a is not None and len(a) == len(tensors) and all(t is s and t._version == v for t, (s, v) in zip(tensors, a))


class GraphedSupervisedStep(_Graphed):
    """SupervisedStep.__call__ from hipGraphs (see the module docstring).  The returned loss is a static tensor the next
    call overwrites: clone it to keep it."""

    def __init__(self, step, warmup=2, split=None, agree=None):
        """split: capture the iteration as two graphs sharing a memory pool -- M1 = forward, loss and the backward of the
        head and the decoder, M2 = the backward of the transformer blocks and the patch encoder + AdamW (the model cuts its
        autograd graph between the two: SupervisedStep.forward_backward_head) -- and start P between them: beside the
        blocks' small GEMMs, where the eager step queues its look-ahead, instead of beside the forward.  Same bits.  It pays
        when what follows the cut outlasts P (~6.3 ms whatever the batch: its FPS runs one cloud per CU): at 8 clouds the
        replay goes from 0.45 ms behind the eager step to level, at 4 clouds 21.95 -> 21.08 ms, at 3 19.37 -> 18.59; at 1 / 2
        clouds M2 is shorter than P and the step waits for it (10.1 -> 12.4, 13.6 -> 15.2 ms).  Default (None):
        GEOT_GRAPH_SPLIT=1 / 0 if set, else split batches of at least 3 clouds.  agree: see _Graphed."""
        super().__init__(step, warmup, agree)
        env = os.environ.get("GEOT_GRAPH_SPLIT")
        self.split = bool(split) if split is not None else (None if env is None else env == "1")
        self.x = None            # static (pos, cls, target)
        self.next_pos = None     # static coordinates P works on

    def __call__(self, pos, cls, target, next_pos=None):
        if self.x is None:
            self.device = pos.device
            self.x = (pos.detach().clone().contiguous(), cls.detach().clone(), target.detach().clone())
            self.next_pos = torch.empty_like(self.x[0])
            if self.split is None:
                self.split = pos.shape[0] >= 3
        for dst, src, name in zip(self.x, (pos, cls, target), ("pos", "cls", "target")):
            _fits(dst, src, name)
        if next_pos is not None:
            _fits(self.next_pos, next_pos, "next_pos")
        self._check_lr()
        announced_now = self._is_announced((pos,))
        self._join_pending()
        for dst, src in zip(self.x, (pos, cls, target)):
            dst.copy_(src)

        def load_next(current):
            self.next_pos.copy_(self.x[0] if current else next_pos)

        def lookahead():
            return self.step.lookahead_work(self.next_pos)

        def train():
            return self.step.iteration(self.x[0], self.x[1], self.x[2], self.pre, None)[0]
        if self.split:
            def head():
                self._loss, rest = self.step.forward_backward_head(self.x[0], self.x[1], self.x[2], self.pre)
                return rest

            def rest_update(rest):
                self.step.backward_rest_update(rest)
                return self._loss
            train = (head, rest_update)
        return self._iterate(announced_now, None if next_pos is None else (next_pos,), load_next, lookahead, train)


_P_KEYS = (("pos",), ("pos_s", "pos_w", "x_w", "cls_w", "raw_pos"))      # what FixMatchNTMStep.lookahead_work reads


class GraphedFixMatchStep(_Graphed):
    """FixMatchNTMStep.__call__ from hipGraphs.  The returned losses are static tensors the next call overwrites."""

    def __init__(self, step, warmup=2, split=None, agree=None):
        """split: as GraphedSupervisedStep -- M1 ends where the backward reaches the student's transformer blocks, P starts
        there.  Same bits, but a loss here: P is 10 ms (the teacher's forward is in it) and does not fit beside M2 -- 29.0 ->
        30.4 ms per iteration (bench.py --workload fixmatch, alternating).  Off unless split=True / GEOT_GRAPH_SPLIT=1."""
        super().__init__(step, warmup, agree)
        self.split = bool(split) if split is not None else os.environ.get("GEOT_GRAPH_SPLIT") == "1"
        self.data = self.data_u = None       # static batch dicts
        self.next = None                     # static inputs of P

    def __call__(self, data, data_u, next_batches=None):
        step = self.step
        if self.data is None:
            self.device = data["pos"].device
            self.data = {k: v.detach().clone().contiguous() for k, v in data.items() if torch.is_tensor(v)}
            self.data_u = {k: v.detach().clone().contiguous() for k, v in data_u.items() if torch.is_tensor(v) and k != "T"}
            self.next = tuple({k: torch.empty_like(d[k]) for k in keys} for d, keys in zip((self.data, self.data_u), _P_KEYS))
        for dst, src, what in ((self.data, data, "data"), (self.data_u, data_u, "data_u")):
            for k, v in dst.items():
                _fits(v, src[k], "%s[%r]" % (what, k))
        self._check_lr()
        announced_now = self._is_announced([b[k] for b, keys in zip((data, data_u), _P_KEYS) for k in keys])
        self._join_pending()
        for dst, src in ((self.data, data), (self.data_u, data_u)):
            for k, v in dst.items():
                v.copy_(src[k])

        def load_next(current):
            srcs = (self.data, self.data_u) if current else next_batches
            for dst, src, keys in zip(self.next, srcs, _P_KEYS):
                for k in keys:
                    _fits(dst[k], src[k], "next[%r]" % k)
                    dst[k].copy_(src[k])

        def lookahead():
            return step.lookahead_work(self.next[0], self.next[1])

        def train():
            pre = self.pre
            return step.student_iteration(self.data, self.data_u, pre["geom_s"], pre["pseudo"], pre["knn"], ema_in_place=True)
        if self.split:
            def head():
                pre = self.pre
                self._losses, rest = step.student_iteration(self.data, self.data_u, pre["geom_s"], pre["pseudo"], pre["knn"],
                                                            ema_in_place=True, defer_rest=True)
                return rest

            def rest_update(rest):
                rest()
                return self._losses
            train = (head, rest_update)
        next_src = None if next_batches is None else [b[k] for b, keys in zip(next_batches, _P_KEYS) for k in keys]
        return self._iterate(announced_now, next_src, load_next, lookahead, train)
