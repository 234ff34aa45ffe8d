"""GPU suite, N-rank control flow on the one-GPU test box: two ranks share GPU 0 and talk over gloo (the RCCL path
itself needs >= 2 GPUs, which only the driver's 8-GPU node has).

* bench.py's own N > 1 tail, started the way the driver starts it (fresh child processes): rendezvous, SyncBatchNorm +
  DDP step, barriers, MAX over ranks, rank-0 JSON with the audit block (`comm`: ranks in the collective, exposed
  all-reduce time) -- for the configs[3] and the configs[4] workload.
* one full FixMatch+NTM iteration (train.py:455-602: frozen teacher, DDP student, DDP T_predictor, SyncBatchNorm through
  fused_norm.bn_act, anchor-row all-gather) on two ranks against the same iteration on the concatenated batch in one
  process: same losses, same EMA matrix, same averaged gradients."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = dict(trans_dim=384, depth=3, num_heads=4, group_size=16, num_group=64, encoder_dims=256, nclasses=17,
             drop_path_rate=0.0, downsample_targets=[1024, 512, 256], extract_layers=[1, 2, 3])
N = 2048


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("workload", ["model", "fixmatch"])
def test_bench_two_rank_rehearsal_emits_an_auditable_line(workload):
    env = dict(os.environ, GEOT_BENCH_REHEARSAL="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--clouds", "1", "--steps", "2",
                        "--warmup", "1", "--workload", workload, "--points", "8192"], env=env, capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]            # rank 0 alone prints
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    assert rec["value"] > 0 and np.isfinite(rec["ms_per_step"]) and "REHEARSAL" in rec["data"]
    assert rec["config"]["parallelism"].startswith("dp2") and rec["config"]["points"] == 8192
    clouds = {"model": 1, "fixmatch": 2}[workload]
    assert rec["config"]["clouds_per_gpu"] == clouds
    assert abs(rec["value"] - 2 * clouds * 2 / (rec["ms_per_step"] * 2 / 1e3)) < 1e-6 * rec["value"]   # whole-job clouds / max time
    comm = rec["comm"]
    assert comm["ranks_in_collective"] == 2 and comm["backend"] == "gloo"
    assert comm["gradient_allreduce_mb_per_step"] > 100            # 27 M fp32 parameters
    assert np.isfinite(comm["exposed_allreduce_ms"]) and comm["ms_per_step_without_gradient_allreduce"] > 0
    assert comm["ms_per_step_without_lookahead"] > 0
    assert "cpu_baseline" not in rec and rec["graph"]["replayed"] is False          # gloo: collectives on the host, not capturable -- eager DDP
    ranks = rec["ranks"]                                                            # what a scaling line is attributed with
    assert len(ranks["host_issue_ms_per_step"]["per_rank"]) == 2 and ranks["host_issue_ms_per_step"]["max"] >= ranks["host_issue_ms_per_step"]["mean"] > 0
    assert len(ranks["cores_visible_per_rank"]) == 2 and ranks["host_cpu_ms_per_step"]["max"] > 0
    coll = ranks["python_level_collectives_per_step"]
    # SyncBatchNorm through fused_norm.bn_act: one all-reduce of the statistics forward, one backward, per BatchNorm layer
    assert coll.get("all_reduce", 0) >= 2 * 9
    if workload == "fixmatch":
        assert coll.get("all_gather", 0) + coll.get("all_gather_into_tensor", 0) >= 1   # the class-anchor exchange
    assert np.isfinite(rec["final_loss"])


@pytest.mark.parametrize("workload", ["model", "fixmatch"])
def test_two_rank_rehearsal_gives_the_same_losses_wherever_the_lookahead_is_queued(workload):
    """GEOT_LOOKAHEAD_AT=blocks (inside the backward, beside DDP's bucketed all-reduces) and =forward (behind the forward):
    the look-ahead only moves kernels between streams -- under DDP + SyncBatchNorm on two ranks the last step's loss is the
    same to the last bit, and the same as with the look-ahead off."""
    losses = {}
    for at, extra in (("blocks", []), ("forward", []), ("off", ["--no-lookahead"])):
        env = dict(os.environ, GEOT_BENCH_REHEARSAL="1", GEOT_LOOKAHEAD_AT=at if at != "off" else "blocks")
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--clouds", "1", "--steps", "3",
                            "--warmup", "1", "--workload", workload, "--points", "8192"] + extra, env=env, capture_output=True,
                           text=True, timeout=900, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
        losses[at] = rec["final_loss"]
    assert losses["blocks"] == losses["forward"] == losses["off"], losses


# ---- FixMatch + NTM iteration: 2 ranks == 1 process -------------------------------------------------------------------
def _fixmatch_inputs(dev):
    from geot_amd.synth import make_batch, region_labels
    xyz_l = make_batch(2, N, start_index=11)[0]
    xyz_u = make_batch(2, N, start_index=31)[0]
    scale = np.array([[[0.9, 1.1, 1.05]], [[1.1, 0.95, 0.9]]], np.float32)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)      # noqa: E731
    lab, unl, strong = T(xyz_l), T(xyz_u), T(xyz_u * scale)
    data = {"pos": lab, "x": lab.transpose(1, 2).contiguous(), "cls": torch.tensor([[0], [1]], device=dev),
            "y": T(region_labels(xyz_l))}
    data_u = {"pos_w": unl, "x_w": unl.transpose(1, 2).contiguous(), "cls_w": torch.tensor([[1], [0]], device=dev),
              "pos_s": strong, "x_s": strong.transpose(1, 2).contiguous(), "cls_s": torch.tensor([[1], [0]], device=dev),
              "raw_pos": unl}
    return data, data_u


def _shard(d, r):
    return {k: v[r:r + 1].contiguous() for k, v in d.items()}


def _run_iteration(dev, rank=None):
    """One FixMatch+NTM iteration; returns (losses, ema_t, gradients of a few student / predictor parameters as they
    stand when optimizer.step() is called -- after DDP's averaging)."""
    import torch.distributed as dist
    from geot_amd import train_step as ts
    torch.manual_seed(77)
    cfg = dict(ts.NTM_CFG, threed_k=8)
    step = ts.build_fixmatch(dev, seg_cfg=SMALL, cfg=cfg, use_ddp=rank is not None,
                             group=dist.group.WORLD if rank is not None else None)
    inner = step.model.module if hasattr(step.model, "module") else step.model
    inner.segmentor.seg_head[2].p = 0.0                          # no dropout: the draws differ with the batch split
    data, data_u = _fixmatch_inputs(dev)
    if rank is not None:
        data, data_u = _shard(data, rank), _shard(data_u, rank)
    grads = {}

    def capture(opt, tag):
        orig = opt.step

        def stepper(*a, **k):
            mod = step.model if tag == "s" else step.T_predictor
            inner = mod.module if hasattr(mod, "module") else mod
            for n, p in inner.named_parameters():
                if p.grad is not None and (tag == "t" or n.endswith(("seg_head.3.weight", "reduce_dim.weight", "sigma",
                                                                     "propogation_0.mlp.layer0.conv.weight",
                                                                     "encoder.first_conv.1.weight", "blocks.blocks.2.attn.qkv.weight"))):
                    grads[tag + ":" + n] = p.grad.detach().cpu().numpy().copy()
            return orig(*a, **k)
        opt.step = stepper
    capture(step.optimizer, "s")
    capture(step.T_optimizer, "t")
    out = step(data, data_u)
    torch.cuda.synchronize()
    return {k: float(v) for k, v in out.items()}, step.ema_t.cpu().numpy(), grads


def _fixmatch_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank), LOCAL_RANK="0")
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    losses, ema, grads = _run_iteration(torch.device("cuda:0"), rank)
    q.put((rank, losses, ema, grads))
    dist.barrier()
    dist.destroy_process_group()


def test_fixmatch_iteration_on_two_ranks_equals_the_single_process_iteration():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fixmatch_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    import queue
    res = []
    while len(res) < len(procs):                     # a rank that dies must fail the test, not leave it waiting
        try:
            res.append(q.get(timeout=5))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead:
                for p in procs:
                    p.terminate()
                pytest.fail("a rank exited with %s" % dead)
    res.sort(key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    losses, ema, grads = _run_iteration(torch.device("cuda:0"))
    (_, l0, e0, g0), (_, l1, e1, g1) = res
    # every rank holds the whole batch's class anchors -> the same EMA matrix as the single process
    np.testing.assert_allclose(e0, ema, rtol=1e-5, atol=1e-7)
    np.testing.assert_array_equal(e0, e1)
    # per-rank losses are means over the rank's clouds: their average is the single-process loss
    for k in losses:
        avg = 0.5 * (l0[k] + l1[k])
        assert abs(avg - losses[k]) <= 2e-4 * max(abs(losses[k]), 1e-3), (k, l0[k], l1[k], losses[k])
    assert set(g0) == set(grads) and len(grads) >= 6
    for k, w in grads.items():
        np.testing.assert_array_equal(g0[k], g1[k])                 # DDP: both ranks step with the same averaged gradient
        err = np.linalg.norm(g0[k] - w) / (np.linalg.norm(w) + 1e-12)
        assert err <= 5e-3, (k, err)                                # fp32: batch-size dependent GEMM blocking / summation order


def _rccl_single_rank_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import geot_amd  # noqa: F401
    import torch.distributed as dist
    from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T
    from geot_amd import train_step as ts
    from geot_amd.synth import make_batch, region_labels
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    x = torch.arange(1 << 20, device=dev, dtype=torch.float32)
    dist.all_reduce(x)
    parts = [torch.empty(4, device=dev)]
    dist.all_gather(parts, torch.arange(4.0, device=dev))
    torch.manual_seed(3)
    init = PointTransformer_seg_T(**SMALL).state_dict()
    xyz = make_batch(2, N, start_index=5)[0]
    pos, tgt = torch.from_numpy(xyz).to(dev), torch.from_numpy(region_labels(xyz)).to(dev)
    cls = torch.zeros(2, 1, dtype=torch.long, device=dev)
    losses = {}
    for wrapped in (True, False):
        m = PointTransformer_seg_T(**SMALL).to(dev)
        m.load_state_dict(init)
        m.seg_head[2].p = 0.0
        net = ts.ddp(m, dev, unused=ts.UNUSED_SUPERVISED, min_world=1) if wrapped else m
        assert isinstance(net, torch.nn.parallel.DistributedDataParallel) == wrapped
        step = ts.SupervisedStep(net)
        losses[wrapped] = [float(step(pos, cls, tgt)) for _ in range(3)]
    torch.cuda.synchronize()
    q.put((dist.get_backend(), float(x[-1]), parts[0].tolist(), losses))
    dist.destroy_process_group()


def _rccl_graphed_ddp_worker(port, q, which):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.pop("DEBUG_CLR_GRAPH_PACKET_CAPTURE", None)
    os.environ["GEOT_GRAPH_LAUNCH"] = "fast"              # captures are inspected: kernel nodes only
    import geot_amd  # noqa: F401
    import torch.distributed as dist
    from geot_amd import train_step as ts, graph_step as gs
    from geot_amd.synth import make_batch, region_labels
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    out = {}
    try:
        if which == "supervised":
            from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T
            torch.manual_seed(3)
            init = PointTransformer_seg_T(**SMALL).state_dict()
            batches = []
            for start in (5, 60):
                xyz = make_batch(3, N, start_index=start)[0]
                batches.append((torch.from_numpy(xyz).to(dev), torch.zeros(3, 1, dtype=torch.long, device=dev),
                                torch.from_numpy(region_labels(xyz)).to(dev)))
            for mode in ("eager", "graph"):
                m = PointTransformer_seg_T(**SMALL).to(dev)
                m.load_state_dict(init)
                if mode == "eager":       # train.py:159-166: SyncBatchNorm + DistributedDataParallel, bucketed all-reduce in the backward
                    step = ts.SupervisedStep(ts.ddp(m, dev, unused=ts.UNUSED_SUPERVISED, min_world=1))
                else:                     # the bare converted module + one flat all-reduce: what a hipGraph can hold
                    net = ts.sync_only(m, min_world=1)
                    step = ts.SupervisedStep(net, grad_sync=ts.GradSync([net], dist.group.WORLD))
                call = gs.GraphedSupervisedStep(step, warmup=2) if mode == "graph" else step
                torch.manual_seed(7)
                losses = []
                for i in range(6):                                   # 2 eager warm-ups over the static buffers, capture, 3 replays
                    cur, nxt = batches[i % 2], batches[(i + 1) % 2]
                    losses.append(float(call(cur[0], cur[1], cur[2], next_pos=nxt[0])))
                torch.cuda.synchronize()
                state = {k: v.detach().float().cpu().numpy() for k, v in m.state_dict().items()}
                out[mode] = {"losses": losses, "state": state}
                if mode == "graph":
                    out["nodes"] = call.node_types
                    out["sync_calls"] = step.grad_sync.collectives
                    out["captured"] = call.captured
        else:
            cfg = dict(ts.NTM_CFG, threed_k=8)
            import importlib.util
            spec = importlib.util.spec_from_file_location("_gs_tests", os.path.join(ROOT, "tests", "test_graph_step_gpu.py"))
            gst = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(gst)
            _fix_batch, SMALL_FIX = gst._fix_batch, gst.SMALL
            batches = [_fix_batch(3), _fix_batch(400)]
            for mode in ("eager", "graph"):
                torch.manual_seed(5)
                step = ts.build_fixmatch(dev, seg_cfg=SMALL_FIX, cfg=cfg, use_ddp=(mode == "eager"), graph_sync=(mode == "graph"),
                                         group=dist.group.WORLD, min_world=1)
                assert isinstance(step.model, torch.nn.parallel.DistributedDataParallel) == (mode == "eager")
                call = gs.GraphedFixMatchStep(step, warmup=2) if mode == "graph" else step
                torch.manual_seed(11)
                losses = []
                for i in range(5):
                    cur, nxt = batches[i % 2], batches[(i + 1) % 2]
                    losses.append({k: float(v) for k, v in call(cur[0], cur[1], next_batches=nxt).items()})
                torch.cuda.synchronize()
                out[mode] = {"losses": losses, "ema_t": step.ema_t.cpu().numpy()}
                if mode == "graph":
                    out["nodes"] = call.node_types
                    out["sync_calls"] = step.grad_sync.collectives
                    out["captured"] = call.captured
        q.put(out)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("which", ["supervised", "fixmatch"])
def test_ddp_step_replayed_from_graphs_over_rccl_equals_eager(which):
    """N > 1 without the host: a DistributedDataParallel + SyncBatchNorm step over backend nccl (one rank: what this box
    allows) against the same step on the bare SyncBatchNorm-converted modules replayed from hipGraphs -- train_step.GradSync
    puts ONE flat gradient all-reduce between the backward and the optimizer, the captures hold kernel nodes only: the
    losses and the parameters of the eager DDP step bit for bit over alternating batches (the reducer divides by the world size and sums: at one rank both are the
    identity, so the two exchanges agree exactly).  examples/segmentation/train.py:159-166, 646-669."""
    import torch.multiprocessing as mp
    import queue
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_graphed_ddp_worker, args=(_free_port(), q, which))
    p.start()
    res = None
    for _ in range(160):
        try:
            res = q.get(timeout=5)
            break
        except queue.Empty:
            if p.exitcode not in (None, 0):
                pytest.fail("the RCCL rank exited with %s" % p.exitcode)
    p.join(timeout=120)
    assert res is not None and p.exitcode == 0
    assert res["captured"] and res["sync_calls"] >= 3
    assert all(set(v) == {"kernel"} for v in res["nodes"].values()), res["nodes"]
    assert res["eager"]["losses"] == res["graph"]["losses"], (res["eager"]["losses"], res["graph"]["losses"])
    if which == "supervised":
        for k, v in res["eager"]["state"].items():
            np.testing.assert_array_equal(v, res["graph"]["state"][k], err_msg=k)
    else:
        np.testing.assert_array_equal(res["eager"]["ema_t"], res["graph"]["ema_t"])


def test_rccl_backend_runs_a_ddp_step_on_one_rank():
    """The only RCCL execution a one-GPU box allows: a process group over backend "nccl" (= RCCL on ROCm) with ONE rank --
    all-reduce and all-gather go through the library, and DistributedDataParallel + SyncBatchNorm over it drive three
    supervised steps (bucketed gradient all-reduce in the backward, look-ahead off) to the same losses as the bare model.
    What it cannot show is a transfer between GPUs; that needs the driver's multi-GPU node."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_single_rank_worker, args=(_free_port(), q))
    p.start()
    import queue
    res = None
    for _ in range(120):
        try:
            res = q.get(timeout=5)
            break
        except queue.Empty:
            if p.exitcode not in (None, 0):
                pytest.fail("the RCCL rank exited with %s" % p.exitcode)
    p.join(timeout=120)
    assert res is not None and p.exitcode == 0
    backend, last, gathered, losses = res
    assert backend == "nccl" and last == float((1 << 20) - 1) and gathered == [0.0, 1.0, 2.0, 3.0]
    assert losses[True] == losses[False], losses


@pytest.mark.parametrize("packet_capture_env", [None, "1"])
def test_bench_one_gpu_line_has_both_modes_or_says_why_not(packet_capture_env):
    """bench.py on one GPU, as a fresh child: the replay leg runs in the runtime's fast graph mode (kernel-only graphs, their
    node counts in the JSON) next to the eager leg; with graph packet capture forced on by the environment and no
    GEOT_GRAPH_LAUNCH=fast the captures are inspected all the same (kernel nodes only: geot_amd/__init__.py) and replay."""
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "DEBUG_CLR_GRAPH_PACKET_CAPTURE",
                        "GEOT_GRAPH_LAUNCH")}
    if packet_capture_env is not None:
        env["DEBUG_CLR_GRAPH_PACKET_CAPTURE"] = packet_capture_env
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--clouds", "2", "--steps", "3", "--warmup", "1",
                        "--points", "8192", "--no-cpu-baseline", "--no-dense-reference", "--no-also"], env=env, capture_output=True, text=True,
                       timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 1 and rec["value"] > 0 and np.isfinite(rec["ms_per_step"])
    g = rec["graph"]
    if packet_capture_env is None:
        assert g["replayed"] is True and g["launch_mode"] == "fast" and "eager" in rec        # 2 clouds: the replay is primary
        assert all(set(v) == {"kernel"} for v in g["nodes"].values()) and set(g["nodes"]) >= {"P"}
        assert rec["host_issue_ms_per_step"] < rec["eager"]["host_issue_ms_per_step"]
    else:
        assert g["replayed"] is True and g["launch_mode"] == "inspected" and "replay_refused" not in g
        assert all(set(v) == {"kernel"} for v in g["nodes"].values())


@pytest.mark.parametrize("workload", ["model", "fixmatch"])
def test_bench_multi_gpu_code_paths_on_one_rank_over_rccl(workload):
    """GEOT_BENCH_SOLO_DP=1: every line the N > 1 bench will execute, on ONE rank over RCCL -- DDP + SyncBatchNorm wrapped, the
    short eager-DDP leg timed first with the replay guard armed, the step captured with its flat gradient all-reduce and
    replayed (kernel-only graphs), the eager leg after it, the comm audit (no_sync leg, per-rank host figures, collectives
    counted) -- so that the multi-GPU run meets no code for the first time.  (What one rank cannot show: a transfer.)"""
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "DEBUG_CLR_GRAPH_PACKET_CAPTURE",
                        "GEOT_GRAPH_LAUNCH")}
    env["GEOT_BENCH_SOLO_DP"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--clouds", "2" if workload == "model" else "1",
           "--steps", "3", "--warmup", "1", "--points", "8192", "--no-cpu-baseline", "--no-dense-reference", "--no-also"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines                                        # the guard stayed silent: ONE line
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["value"] > 0 and np.isfinite(rec["ms_per_step"])
    assert rec["config"]["parallelism"].startswith("dp1: D")
    g = rec["graph"]
    assert g["replayed"] is True and "replay_refused" not in g, g
    assert all(set(v) == {"kernel"} for v in g["nodes"].values())
    assert "eager" in rec and rec["eager"]["ms_per_step"] > 0          # the DDP step, timed after the replay
    assert rec["comm"]["backend"] == "nccl" and rec["comm"]["ranks_in_collective"] == 1
    assert rec["comm"]["gradient_allreduce_mb_per_step"] > 50
    assert "ranks" in rec and len(rec["ranks"]["host_issue_ms_per_step"]["per_rank"]) == 1
    assert "ReplayGuard" not in rec["config"]["workload"]
