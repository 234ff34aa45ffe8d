"""geot_amd/graph_step.py: the two training steps replayed from a hipGraph give the SAME bits as the eager steps --
losses of every iteration, every parameter, buffer and AdamW moment afterwards -- over alternating batches with the
look-ahead on (the geometry of batch k + 1 is produced by replay k and consumed by replay k + 1), with the random layers
(DropPath, the head's Dropout) left on: the philox offsets of a replay continue where eager execution would be."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")

SMALL = dict(trans_dim=384, depth=3, num_heads=4, group_size=32, num_group=128, encoder_dims=256, nclasses=17,
             drop_path_rate=0.1, downsample_targets=[2048, 1024, 512], extract_layers=[1, 2, 3])


def _sup_batches(b, n):
    from geot_amd.synth import make_batch, region_labels
    out = []
    for start in (0, 40, 90):
        xyz = make_batch(b, n, start_index=start)[0]
        out.append((torch.from_numpy(xyz).to(DEV), torch.randint(0, 2, (b, 1), device=DEV),
                    torch.from_numpy(region_labels(xyz)).to(DEV)))
    return out


def _state(step):
    out = {}
    for name, mod in (("model", step.model), ("T", getattr(step, "T_predictor", None))):
        if mod is not None:
            out.update({name + "." + k: v.detach().clone() for k, v in mod.state_dict().items()})
    for i, opt in enumerate(step.optimizers()):
        for j, p in enumerate(pp for g in opt.param_groups for pp in g["params"]):
            for k, v in opt.state.get(p, {}).items():
                if torch.is_tensor(v):
                    out["opt%d.%d.%s" % (i, j, k)] = v.detach().clone()
    if hasattr(step, "ema_t"):
        out["ema_t"] = step.ema_t.detach().clone()
    return out


def _same(a, b):
    assert set(a) == set(b)
    bad = [k for k in a if not torch.equal(a[k], b[k])]
    assert not bad, (len(bad), bad[:8])


@pytest.mark.parametrize("look,split", [(True, False), (False, False), (True, True)])
def test_supervised_step_from_a_graph_equals_eager(look, split):
    from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T
    from geot_amd import train_step as ts, graph_step as gs
    batches = _sup_batches(2, 6000)
    torch.manual_seed(0)
    init = PointTransformer_seg_T(**SMALL).state_dict()
    order = [0, 1, 0, 1, 2, 0, 1]          # batch 2 once, unannounced in the look-ahead run: its geometry is made in line
    runs = {}
    for mode in ("eager", "graph"):
        m = PointTransformer_seg_T(**SMALL).to(DEV)
        m.load_state_dict(init)
        step = ts.SupervisedStep(m)
        call = gs.GraphedSupervisedStep(step, warmup=2, split=split) if mode == "graph" else step
        torch.manual_seed(7)
        losses = []
        for i, k in enumerate(order):
            cur = batches[k]
            nxt = batches[order[i + 1]] if i + 1 < len(order) else batches[0]
            announce = look and not (i + 1 < len(order) and order[i + 1] == 2)   # batch 2 arrives unannounced
            wrong = batches[0][0].clone() if (look and not announce) else None   # ... a DIFFERENT tensor was announced
            losses.append(call(cur[0], cur[1], cur[2], next_pos=nxt[0] if announce else wrong).clone())
        torch.cuda.synchronize()
        if mode == "graph":
            assert call.captured and call.calls == len(order)
            assert sorted(call.node_types) == (["M1", "M2", "P"] if split else ["M", "P"])
            assert all(set(v) == {"kernel"} for v in call.node_types.values()), call.node_types     # (see the fast-mode test)
        runs[mode] = (losses, _state(step))
    for i, (a, b) in enumerate(zip(*[runs[m][0] for m in ("eager", "graph")])):
        assert torch.equal(a, b), (i, float(a), float(b))
    _same(runs["eager"][1], runs["graph"][1])


@pytest.mark.parametrize("split", [False, True])
def test_eager_steps_before_the_wrapper_do_not_reach_into_the_capture(split):
    """A step that has ALREADY run eagerly (look-ahead on) and is wrapped afterwards: dead python cycles still hold those
    iterations' autograd graphs -- every parameter's AccumulateGrad node with them, bound to the eager stream -- and a backward
    under capture that meets them forks the capture (this runtime: a crash in hipStreamEndCapture, seen at 2 clouds in the
    one-graph mode).  graph_step collects before every capture; the run continues with the bits of an all-eager run."""
    import gc
    from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T
    from geot_amd import train_step as ts, graph_step as gs
    batches = _sup_batches(2, 6000)
    torch.manual_seed(0)
    init = PointTransformer_seg_T(**SMALL).state_dict()
    order = [0, 1, 0, 1, 0, 1, 0, 1]
    runs = {}
    was = gc.isenabled()
    gc.disable()                   # (no collection by chance between the eager steps and the capture)
    try:
        for mode in ("eager", "mixed"):
            m = PointTransformer_seg_T(**SMALL).to(DEV)
            m.load_state_dict(init)
            step = ts.SupervisedStep(m)
            graphed = None
            torch.manual_seed(7)
            losses = []
            for i, k in enumerate(order):
                if mode == "mixed" and i == 3:
                    graphed = gs.GraphedSupervisedStep(step, warmup=2, split=split)
                cur, nxt = batches[k], batches[order[(i + 1) % len(order)]]
                call = graphed if graphed is not None else step
                # (the wrapper's first call finds the geometry the last eager step queued for this batch unused: it makes its own)
                losses.append(call(cur[0], cur[1], cur[2], next_pos=nxt[0]).clone())
            torch.cuda.synchronize()
            if graphed is not None:
                assert graphed.captured and all(set(v) == {"kernel"} for v in graphed.node_types.values())
            runs[mode] = (losses, _state(step))
    finally:
        if was:
            gc.enable()
    for i, (a, b) in enumerate(zip(runs["eager"][0], runs["mixed"][0])):
        assert torch.equal(a, b), (i, float(a), float(b))
    _same(runs["eager"][1], runs["mixed"][1])


def _fix_batch(seed, n=4096):
    from geot_amd.synth import make_batch, region_labels
    xl, xu = make_batch(2, n, start_index=seed)[0], make_batch(2, n, start_index=seed + 50)[0]
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)      # noqa: E731
    lab, unl, strong = T(xl), T(xu), T(xu * np.float32(1.04))
    z = torch.zeros(2, 1, dtype=torch.long, device=DEV)
    return ({"pos": lab, "x": lab.transpose(1, 2).contiguous(), "cls": z, "y": T(region_labels(xl))},
            {"pos_w": unl, "x_w": unl.transpose(1, 2).contiguous(), "cls_w": z, "pos_s": strong,
             "x_s": strong.transpose(1, 2).contiguous(), "cls_s": z, "raw_pos": unl})


@pytest.mark.parametrize("look,split", [(True, False), (False, False), (True, True)])
def test_fixmatch_iteration_from_a_graph_equals_eager(look, split):
    from geot_amd import train_step as ts, graph_step as gs
    cfg = dict(ts.NTM_CFG, threed_k=8)
    batches = [_fix_batch(3), _fix_batch(400)]
    runs = {}
    for mode in ("eager", "graph"):
        torch.manual_seed(5)
        step = ts.build_fixmatch(DEV, seg_cfg=SMALL, cfg=cfg, use_ddp=False)
        call = gs.GraphedFixMatchStep(step, warmup=2, split=split) if mode == "graph" else step
        torch.manual_seed(11)
        out = []
        for i in range(6):
            cur, nxt = batches[i % 2], batches[(i + 1) % 2]
            res = call(cur[0], cur[1], next_batches=nxt if look else None)
            out.append({k: v.clone() for k, v in res.items()})
        torch.cuda.synchronize()
        if mode == "graph":
            assert call.captured
            assert len(call.node_types) == (3 if split else 2) and all(set(v) == {"kernel"} for v in call.node_types.values()), call.node_types
        runs[mode] = (out, _state(step))
    for i, (a, b) in enumerate(zip(runs["eager"][0], runs["graph"][0])):
        for k in a:
            assert torch.equal(a[k], b[k]), (i, k, float(a[k]), float(b[k]))
    _same(runs["eager"][1], runs["graph"][1])


def test_fixmatch_eager_calls_between_replays_share_the_ema_buffer():
    """An eager iteration on a step that a GraphedFixMatchStep wraps updates ema_t IN its buffer (the captured graph holds that
    very tensor): replay, eager, replay equals three eager iterations -- bench.py runs exactly this sequence of legs."""
    from geot_amd import train_step as ts, graph_step as gs
    cfg = dict(ts.NTM_CFG, threed_k=8)
    batches = [_fix_batch(3), _fix_batch(400)]
    modes = ["g", "g", "g", "e", "g", "e", "g"]        # (the first two graphed calls are the eager warm-up over the static buffers)
    runs = {}
    for mode in ("eager", "mixed"):
        torch.manual_seed(5)
        step = ts.build_fixmatch(DEV, seg_cfg=SMALL, cfg=cfg, use_ddp=False)
        graphed = gs.GraphedFixMatchStep(step, warmup=2) if mode == "mixed" else None
        buf = step.ema_t
        torch.manual_seed(11)
        out = []
        for i, how in enumerate(modes):
            cur = batches[i % 2]
            call = graphed if (graphed is not None and how == "g") else step
            out.append({k: v.clone() for k, v in call(cur[0], cur[1]).items()})
            assert step.ema_t is buf                     # never rebound
        torch.cuda.synchronize()
        runs[mode] = (out, _state(step))
    for i, (a, b) in enumerate(zip(runs["eager"][0], runs["mixed"][0])):
        for k in a:
            assert torch.equal(a[k], b[k]), (i, k, float(a[k]), float(b[k]))
    _same(runs["eager"][1], runs["mixed"][1])


def test_a_changed_float_lr_is_refused():
    """A python-float lr is baked into the captured AdamW step; a scheduler that replaces it must not go unnoticed."""
    from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T
    from geot_amd import train_step as ts, graph_step as gs
    torch.manual_seed(0)
    m = PointTransformer_seg_T(**SMALL).to(DEV)
    step = ts.SupervisedStep(m)
    call = gs.GraphedSupervisedStep(step, warmup=1)
    a = _sup_batches(2, 6000)[0]
    call(*a)
    for opt in step.optimizers():
        for group in opt.param_groups:
            if not torch.is_tensor(group["lr"]):
                group["lr"] = group["lr"] * 0.5
    with pytest.raises(RuntimeError, match="lr changed"):
        call(*a)


def test_a_batch_of_another_shape_is_refused():
    from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T
    from geot_amd import train_step as ts, graph_step as gs
    torch.manual_seed(0)
    m = PointTransformer_seg_T(**SMALL).to(DEV)
    call = gs.GraphedSupervisedStep(ts.SupervisedStep(m), warmup=1)
    a = _sup_batches(2, 6000)[0]
    call(*a)
    b = _sup_batches(1, 6000)[0]
    with pytest.raises(RuntimeError, match="captured for"):
        call(*b)


def test_replays_survive_eager_launches_between_them():
    """The ROCm 7.0 hazard geot_amd/__init__.py switches off (graph packet capture: the captured packets of a graph's
    hipMemsetAsync nodes point into the device's shared kernel-argument ring): a captured torch reduction that zeroes its
    semaphores by memset returned garbage after ~20 000 eager launches between two replays (tools/lab/packet_capture_repro.py;
    in the training step: four bias / weight gradients at 1e36 after 13 eager iterations between two replays).  With the
    switch off -- what conftest.py and `import geot_amd` arrange -- the second replay equals the first."""
    import geot_amd
    assert geot_amd.graph_replay_is_safe()
    x = torch.randn(8, 5, 24000, device=DEV)
    out = torch.zeros(5, dtype=torch.float64, device=DEV)
    lin = torch.nn.Linear(128, 384).to(DEV)
    xin = torch.randn(4096, 128, device=DEV)
    gb = torch.zeros(384, device=DEV)

    def body():
        for _ in range(10):
            out.copy_(x.sum((0, 2), dtype=torch.float64))
            gb.copy_(torch.autograd.grad(lin(xin).square().sum(), lin.bias)[0])
    s = torch.cuda.Stream(device=DEV)
    s.wait_stream(torch.cuda.current_stream(DEV))
    with torch.cuda.stream(s):
        body()
    torch.cuda.current_stream(DEV).wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    g.replay()
    torch.cuda.synchronize()
    want, want_gb = out.clone(), gb.clone()
    a = torch.randn(1 << 16, device=DEV)
    for _ in range(60000):
        a.mul_(1.0)
    out.zero_()
    gb.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want) and torch.equal(gb, want_gb)


@pytest.mark.parametrize("which", ["supervised", "fixmatch"])
def test_fast_launch_mode_kernel_only_graphs(which):
    """GEOT_GRAPH_LAUNCH=fast: packet capture stays ON (a graph launch costs the host 0.5 ms instead of 7-20 ms); what makes
    that safe is that the captured steps hold kernel nodes only -- verified per graph through hipGraphGetNodes -- and they
    stay bit-equal to eager with 40 000 eager launches between replays (tests/_fast_graph_check.py, its own process: the
    switch is read when HIP initialises)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != "DEBUG_CLR_GRAPH_PACKET_CAPTURE"}
    env["GEOT_GRAPH_LAUNCH"] = "fast"
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "_fast_graph_check.py"), which], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "fast ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_a_learning_rate_schedule_reaches_the_replayed_step_through_a_tensor():
    """A captured AdamW step bakes a python-float lr into the graph.  The reference steps its scheduler every epoch
    (train.py:283 scheduler.step(epoch)): with `lr` a float32 DEVICE tensor in the parameter groups (fused AdamW reads it from
    memory) a schedule is a fill_ between two replays -- and the replayed run equals the eager run with the same tensor lr."""
    from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T
    from geot_amd import train_step as ts, graph_step as gs
    batches = _sup_batches(2, 6000)
    torch.manual_seed(0)
    init = PointTransformer_seg_T(**SMALL).state_dict()
    schedule = [1e-3, 1e-3, 1e-3, 5e-4, 5e-4, 2e-4]
    runs = {}
    for mode in ("eager", "graph"):
        m = PointTransformer_seg_T(**SMALL).to(DEV)
        m.load_state_dict(init)
        step = ts.SupervisedStep(m)
        lr = torch.tensor(schedule[0], dtype=torch.float32, device=DEV)
        for group in step.optimizer.param_groups:
            group["lr"] = lr
        call = gs.GraphedSupervisedStep(step, warmup=2) if mode == "graph" else step
        if mode == "eager":
            for group in step.optimizer.param_groups:
                group["capturable"] = True       # (a tensor lr needs it; the graphed wrapper sets it itself)
        torch.manual_seed(7)
        losses = []
        for i, value in enumerate(schedule):
            lr.fill_(value)
            cur, nxt = batches[i % 2], batches[(i + 1) % 2]
            losses.append(call(cur[0], cur[1], cur[2], next_pos=nxt[0]).clone())
        torch.cuda.synchronize()
        runs[mode] = (losses, _state(step))
    for a, b in zip(runs["eager"][0], runs["graph"][0]):
        assert torch.equal(a, b)
    _same(runs["eager"][1], runs["graph"][1])
    # and the schedule did something: the run with a constant lr ends elsewhere
    m = PointTransformer_seg_T(**SMALL).to(DEV)
    m.load_state_dict(init)
    step = ts.SupervisedStep(m)
    torch.manual_seed(7)
    for i in range(len(schedule)):
        cur, nxt = batches[i % 2], batches[(i + 1) % 2]
        step(cur[0], cur[1], cur[2], next_pos=nxt[0])
    assert not torch.equal(m.seg_head[3].weight, runs["graph"][1]["model.seg_head.3.weight"])
