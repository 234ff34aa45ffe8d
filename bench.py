#!/usr/bin/env python
"""bench.py -- throughput of the GeoT sampling/grouping hot path and of the model that calls it, on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--clouds B] [--workload model|sa|backbone_ops|ntm|fixmatch]

Default workload = BASELINE.json configs[2] (the config the metric "clouds/s ... fwd+bwd" is quoted on):
B = 8 synthetic 24 000-point tooth clouds per GPU through the configured backbone PointTransformer_seg_T
(cfgs/tooth_semi/transformer_finetune_fixmatch_ntm.yaml:6-15; random init) -- forward, Poly1FocalLoss, backward,
gradient all-reduce (N > 1), AdamW step -- fp32 throughout, inputs resident in HBM.  One "step" = one such
iteration over the rank's clouds.

--gpus N > 1 (configs[3]): this script starts N ranks itself (one process per GPU, before the parent touches a
GPU; under torchrun it uses the ranks it is given), binds rank -> GPU, converts BatchNorm to SyncBatchNorm and
wraps the model in DistributedDataParallel over backend "nccl" (= RCCL over xGMI) exactly as
examples/segmentation/train.py:159-166 does; the 108 MB gradient all-reduce is inside the timed step.  The timed
region is bracketed by barrier + synchronize and the MAX over ranks is used.  Weak scaling: B clouds per rank.

Other workloads: `sa` = configs[1] (one SetAbstraction forward, eval); `backbone_ops` = the hot-path ops of the
backbone alone (dense layers replaced by nothing); `ntm` = the unlabelled half of the NTM block alone;
`fixmatch` = configs[4], one full FixMatch+NTM iteration (teacher 2 clouds, student 6 clouds, NTM block, losses,
both optimisers) per rank.

Prints ONE JSON line on rank 0 (DESIGN.md section 7 explains every field).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# One GPU: the replay leg launches its graphs in the runtime's fast mode (graph packet capture left ON: 0.5 ms of host time per
# launch instead of 7-20 ms), which graph_step accepts only for graphs of kernel nodes alone -- it inspects what it captured
# and this file records the node counts (and falls back to the eager step if a graph is refused).  An exported
# GEOT_GRAPH_LAUNCH / DEBUG_CLR_GRAPH_PACKET_CAPTURE wins.  (The sa / ntm --graph capture below is inspected the same way.)
if "DEBUG_CLR_GRAPH_PACKET_CAPTURE" not in os.environ:      # (N > 1 too: the data-parallel step replays from graphs over RCCL)
    os.environ.setdefault("GEOT_GRAPH_LAUNCH", "fast")
import geot_amd  # noqa: E402,F401  (before torch touches the GPU: it pins the HIP runtime's graph switch, geot_amd/__init__.py)

N_POINTS = 24000
NPOINT, RADIUS, NSAMPLE, MLP = 6000, 0.1, 32, [3, 64, 64, 128]
FP32_VECTOR_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: peak FP32 vector (= fp32 MFMA) rate
FP32_MATRIX_PEAK_TFLOPS = 157.3   # dense fp32 MFMA peak (no xf32/TF32 on gfx950)
HBM_PEAK_GBS = 8000.0
FPS_FLOP_PER_UPDATE = 10          # 3 sub, 3 mul, 2 add, min, compare
# dense (GEMM) flops of one PointTransformer_seg_T forward per cloud as the REFERENCE executes it
DENSE_GFLOP_FWD_REFERENCE = 204.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--clouds", type=int, default=None,
                    help="clouds per GPU per step (model/backbone_ops/ntm: 8, sa: 1, fixmatch: 2 labelled + 2 unlabelled)")
    ap.add_argument("--points", type=int, default=N_POINTS,
                    help="points per cloud: 24000 = BASELINE.json's metric; 16000 = the authors' own operating point "
                         "(cfgs/tooth_semi/default.yaml:6 num_points)")
    ap.add_argument("--workload", choices=["model", "sa", "ops_only", "backbone_ops", "ntm", "fixmatch"], default="model",
                    help="ops_only (old name: backbone_ops): the hot-path ops of the backbone with the dense layers replaced "
                         "by NOTHING -- a kernel-timing stand-in, never the metric")
    ap.add_argument("--no-dense-reference", action="store_true",
                    help="model: skip the 3 extra steps in the reference's op order that fill the JSON's dense.reference_order")
    ap.add_argument("--no-saturated", action="store_true", help="sa: skip the extra run at 256 clouds per launch")
    ap.add_argument("--no-lookahead", action="store_true",
                    help="model: do not hand the step the next batch's coordinates (its sampling / grouping then runs at "
                         "the head of its own step instead of beside the previous backward)")
    ap.add_argument("--dense", choices=["factored", "reference"], default=None,
                    help="model: how the first 1x1 conv behind a gather is evaluated (see transformer.py)")
    ap.add_argument("--no-tuned-gemm", action="store_true",
                    help="model / fixmatch: leave the rocBLAS / hipBLASLt solution choice of the dense layers to the "
                         "library defaults instead of the recorded TunableOp selection (geot_amd/tuning)")
    ap.add_argument("--tune-gemm", default=None, metavar="CSV",
                    help="model / fixmatch: let TunableOp time every GEMM solution during warm-up and write CSV (minutes)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=24)
    ap.add_argument("--streams", type=int, default=1, help="sa only: HIP streams the steps are dealt to")
    ap.add_argument("--graph", action="store_true",
                    help="sa / ntm: capture one step into a hipGraph (torch.cuda.graph) and time replays.  model / fixmatch on one "
                         "GPU: make the hipGraph replay (geot_amd/graph_step.py) the PRIMARY mode whatever the size (default: the "
                         "replay at <= 2 clouds in the training forward, the eager step above; the other mode is timed in the "
                         "same run and reported beside it)")
    ap.add_argument("--no-also", action="store_true",
                    help="default run (one GPU, workload model, default sizes): do NOT append the SetAbstraction forward (configs[1]) "
                         "and the FixMatch+NTM iteration (configs[4]) as short legs under `also` in the one JSON line")
    ap.add_argument("--no-graph", action="store_true",
                    help="model / fixmatch: time the eager step only, no replay leg; N > 1 always runs eagerly "
                         "(DistributedDataParallel is host logic)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------
def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N ranks (children of a parent that has NOT touched the
    GPU), rank r -> GPU r, rendezvous on 127.0.0.1; rank 0 prints the JSON line; exit with the first failing rank's code."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), GEOT_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    failed = 0                                   # exit code of the FIRST rank that failed (not of the ranks we stop)
    try:
        live = list(procs)
        while live:                              # a rank that dies would leave the others waiting in a collective:
            for p in list(live):                 # poll, and take the rest down with it
                rc = p.poll()
                if rc is None:
                    continue
                live.remove(p)
                if rc != 0 and not failed:
                    failed = abs(rc)
                    for q in live:
                        q.terminate()
            time.sleep(0.2)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return failed


ALSO_KEYS = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "data", "config", "roofline", "roofline_hbm",
             "cpu_baseline", "saturated", "graph", "final_loss", "host_issue_ms_per_step")


def also_legs(args, run=subprocess.run):
    """The other single-GPU configurations of BASELINE.json as short legs of the default run, each a CHILD process of this
    file (this process has initialised the GPU: it must not exec; a child is the permitted way) started after the primary
    leg has finished its timing -- the parent only waits meanwhile.  Returns {"sa": {...}, "fixmatch": {...}}: the child's
    own JSON line cut down to ALSO_KEYS, or {"error": ...} (a failing leg never fails the primary line)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                            "GEOT_GRAPH_LAUNCH")}
    legs = {"sa": ["--workload", "sa", "--steps", "20", "--warmup", "3", "--cpu-steps", "12"],
            "fixmatch": ["--workload", "fixmatch", "--steps", str(max(4, min(args.steps, 10))), "--warmup", str(max(2, min(args.warmup, 3)))]}
    out = {}
    for name, extra in legs.items():
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--no-also"] + extra
        if args.no_cpu_baseline:
            cmd.append("--no-cpu-baseline")
        t0 = time.time()
        try:
            r = run(cmd, env=env, capture_output=True, text=True, timeout=900)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode != 0 or not lines:
                out[name] = {"error": "exit %d: %s" % (r.returncode, (r.stderr or r.stdout)[-400:])}
                continue
            rec = json.loads(lines[-1])
            leg = {k: rec[k] for k in ALSO_KEYS if k in rec}
            leg["wall_s"] = round(time.time() - t0, 1)
            leg["command"] = "python bench.py " + " ".join(cmd[2:])
            out[name] = leg
        except Exception as e:      # noqa: BLE001
            out[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[:400])}
    return out


def _finite(obj):
    """NaN / inf are not JSON: a figure that could not be measured is reported as null."""
    import math
    if isinstance(obj, dict):
        return {k: _finite(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_finite(v) for v in obj]
    if isinstance(obj, float) and not math.isfinite(obj):
        return None
    return obj


class EventTimer:
    """HIP events around one call, recorded on the stream that is current at the call (the C ABI and torch both
    launch there -- inside `with torch.cuda.stream(side)` that is the side stream)."""

    def __init__(self):
        self.pairs = []
        self.only = None        # wrap(): time a call only if only(*args) is true (e.g. the student's batch, not the teacher's)

    def wrap(self, fn):
        import torch

        def timed(*a, **k):
            if self.only is not None and not self.only(*a, **k):
                return fn(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **k)
            e1.record()
            self.pairs.append((e0, e1))
            return out
        return timed

    def hook(self, module):
        import torch

        def pre(mod, inp):
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
            mod._geot_e0 = e0

        def post(mod, inp, out):
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.pairs.append((mod._geot_e0, e1))
        return [module.register_forward_pre_hook(pre), module.register_forward_hook(post)]

    def mean_ms(self):
        import numpy as np
        return float(np.mean([a.elapsed_time(b) for a, b in self.pairs])) if self.pairs else float("nan")


def pmc_traffic(kernel, tag):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this command
    (profiles/*<tag>*_pmc_traffic.json: FETCH_SIZE and WRITE_SIZE in separate --pmc passes, FETCH doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  Counters cannot be read in-process: the figure is the profiled
    one; `source` names the file and the commit it was taken at so a stale figure is visible."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*%s*_pmc_traffic.json" % tag)),
                   key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))])
    if not files:
        return None, None
    with open(files[-1]) as f:
        prof = json.load(f)
    for name, rec in prof.get("kernels", {}).items():
        if kernel in name:
            source = {"file": os.path.relpath(files[-1], ROOT), "commit": prof.get("commit")}
            # the kernel's source file as profiled against the one this run was built from (None: the profile predates the field)
            import hashlib
            base = re.split(r"[<(]", kernel)[0].split("::")[-1].strip()
            unchanged = None
            for fn, sha in prof.get("csrc_sha16", {}).items():
                path = os.path.join(ROOT, "geot_amd", "csrc", fn)
                if fn.endswith(".hip") and os.path.exists(path) and base and base in open(path).read():
                    unchanged = hashlib.sha256(open(path, "rb").read()).hexdigest()[:16] == sha
            source["kernel_source_unchanged"] = unchanged
            return rec.get("traffic_bytes"), source
    return None, None


def pmc_launches(kernel, tag):
    """launches of `kernel` in the PMC pass pmc_traffic() quotes (None: the profile predates the field)"""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*%s*_pmc_traffic.json" % tag)),
                   key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))])
    if not files:
        return None
    with open(files[-1]) as f:
        for name, rec in json.load(f).get("kernels", {}).items():
            if kernel in name:
                return rec.get("launches")
    return None


def sq_share(kernel, tag):
    """Executed-work figure of `kernel` from the committed SQ counter passes of this command (profiles/*<tag>*_sq_counters.json,
    tools/profile_bench.sh): the share of the cycles its CUs were busy in which a vector instruction was issuing
    (SQ_ACTIVE_INST_VALU / SQ_BUSY_CU_CYCLES), with its source.  Pruned updates are not in it: only instructions that ran."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*%s*_sq_counters.json" % tag)),
                   key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))])
    for path in reversed(files):
        with open(path) as f:
            prof = json.load(f)
        for name, rec in prof.get("kernels", {}).items():
            if kernel in name and rec.get("valu_active_share_of_busy_cu_cycles") is not None:
                return {"value": rec["valu_active_share_of_busy_cu_cycles"], "file": os.path.relpath(path, ROOT), "kernel": name,
                        "valu_insts_per_wave_cycle": rec.get("valu_insts_per_wave_cycle"),
                        "what": "SQ_ACTIVE_INST_VALU / SQ_BUSY_CU_CYCLES of the profiled launches: vector-issue share of the cycles "
                                "the occupied CUs were busy (executed instructions only; pruned updates are not counted)"}
    return None


def affinity_cores(cap=16):
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return min(cap, n)


# ---------------------------------------------------------------------------------------------------------------
def cpu_baseline_sa(xyz_np, feats_np, sa_cpu, steps):
    """configs[1] on the host: oracle (C restatement, OpenMP) for FPS / ball query / grouping + the same SharedMLP
    on torch-CPU.  Checker code, used here only as the reported baseline."""
    import numpy as np
    import torch
    from oracle import capi
    cores = capi.set_threads(affinity_cores())
    torch.set_num_threads(cores)

    def one():
        inds = capi.fps_dense(xyz_np, NPOINT, 512, True)
        new_xyz = np.take_along_axis(xyz_np, inds[..., None].astype(np.int64).repeat(3, -1), 1)
        idx = capi.ball_query(new_xyz, xyz_np, RADIUS, NSAMPLE)
        gx = capi.group_points(np.ascontiguousarray(xyz_np.transpose(0, 2, 1)), idx)
        gx -= new_xyz.transpose(0, 2, 1)[..., None]
        gf = capi.group_points(feats_np, idx)
        with torch.no_grad():
            y = sa_cpu.mlp_module(torch.from_numpy(np.concatenate([gx, gf], 1)))
            return torch.nn.functional.max_pool2d(y, kernel_size=[1, y.size(3)]).squeeze(-1)

    one()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    dt = time.perf_counter() - t0
    return {"value": xyz_np.shape[0] * steps / dt, "unit": "clouds/s", "cores": cores, "kind": "port",
            "sample": "%d SetAbstraction forwards of %d cloud(s) x %d pts (oracle C/OpenMP FPS+ball_query+group, "
                      "torch-CPU SharedMLP+max), %.1f s" % (steps, xyz_np.shape[0], N_POINTS, dt)}


def cpu_baseline_model():
    """configs[2] on the host, bounded sample: ONE fwd+loss+bwd+AdamW step of the same PointTransformer_seg_T over
    2 clouds x 24 000 points on torch-CPU, the sampling / grouping ops (a) from the reference's own pure-torch
    fallbacks (pointmlp.py:45-143 + layers/knn.py:7-20, restated in oracle/torch_cpu_ref.py) = "reference-fallback",
    (b) from the C/OpenMP oracle = "port"."""
    import torch
    from oracle import torch_cpu_ref
    cores = affinity_cores()
    v_ref, t_ref = torch_cpu_ref.time_model_step("reference-fallback", 2, N_POINTS, cores)
    v_port, t_port = torch_cpu_ref.time_model_step("port", 2, N_POINTS, cores)
    return {"value": v_ref, "unit": "clouds/s", "cores": cores, "kind": "reference-fallback",
            "torch": torch.__version__,
            "sample": "1 step (fwd + Poly1Focal + bwd + AdamW) of PointTransformer_seg_T on 2 clouds x %d pts, torch-CPU "
                      "dense layers, hot-path ops = the reference's torch fallbacks (pointmlp.py:45-143, knn.py:7-20), "
                      "%.1f s" % (N_POINTS, t_ref),
            "port": {"value": v_port, "unit": "clouds/s", "cores": cores, "kind": "port",
                     "sample": "same step, index-producing ops from the C/OpenMP oracle (exact CUDA semantics), %.1f s" % t_port}}


def cpu_baseline_fixmatch(bl, bu):
    """configs[4] on the host, bounded sample, composed from two measured pieces: (a) the supervised step of the same
    model on 2 clouds (cpu_baseline_model: forward + loss + backward + AdamW) -- the student's 6-cloud forward + backward
    and the teacher's 2-cloud forward are priced from it by cloud count (a forward alone at 1/3 of a step); (b) the NTM block
    of ONE iteration (sig_t_mean, class transition, logit correction, threeD_space_loss with its k = 32 graph) on the bu
    unlabelled clouds, timed on oracle/np_ntm.py -- the fp64 numpy restatement of transformer.py:1099-1131,
    train.py:505-557 and utils/insT_loss.py:61-110 (the reference's own code for this block hard-codes .cuda()) -- with the
    neighbour search from the C/OpenMP oracle.  Checker code, used here only as the reported baseline."""
    import numpy as np
    from oracle import capi, np_ntm
    from geot_amd.synth import make_batch, make_logits
    base = cpu_baseline_model()
    cores = base["cores"]
    capi.set_threads(cores)
    xyz = make_batch(1, N_POINTS, start_index=10_000)[0]       # ONE unlabelled cloud (clouds are independent here): x bu below
    pw, ps = make_logits(xyz, index=0), make_logits(xyz, index=1, sharp=3.0)

    def softmax(z):
        e = np.exp(z - z.max(1, keepdims=True))
        return e / e.sum(1, keepdims=True)
    C = pw.shape[1]
    rng = np.random.default_rng(1609)
    W = rng.standard_normal((C, C, 2 * C)) * 0.1
    cm, ema, sigma = np.full((C, C), 1.0 / C), np.eye(C), np.full(C, 0.4)
    t0 = time.perf_counter()
    eta, p = softmax(pw), softmax(ps)
    tr = np_ntm.class_transition(eta, sigma, ema)
    ins = np_ntm.sig_t_mean(p, cm, W)
    np_ntm.correct_logits(ps, ins, tr["ema_t_corr"], 0.9)
    nbr = capi.knn_sorted(xyz, xyz, 33)[0][:, :, 1:]
    np_ntm.threed_space_loss(xyz, eta.argmax(1), ins, nbr, 1.0)
    t_ntm = (time.perf_counter() - t0) * bu
    t_step2 = 2.0 / base["value"]                        # seconds of one supervised step on 2 clouds
    t_iter = t_step2 * (bl + 2 * bu) / 2.0 + (t_step2 / 3.0) * bu / 2.0 + t_ntm
    return {"value": (bl + bu) / t_iter, "unit": "clouds/s", "cores": cores, "kind": "reference-fallback + port, composed",
            "torch": base.get("torch"),
            "sample": "one FixMatch+NTM iteration priced from two measured pieces: the supervised step on 2 clouds (%.1f s: %s) "
                      "scaled to the student's %d clouds fwd+bwd and the teacher's %d clouds fwd (1/3 of a step per cloud), "
                      "plus the NTM block (sig_t_mean + class transition + correction + threeD_space_loss k=32 with its "
                      "gradient) in fp64 numpy with the C/OpenMP kNN, timed on 1 cloud x %d points and scaled to the %d "
                      "unlabelled clouds (%.1f s)"
                      % (t_step2, base["kind"], bl + 2 * bu, bu, N_POINTS, bu, t_ntm),
            "ntm_block_s": t_ntm, "supervised_step_2_clouds_s": t_step2, "supervised": base}


# ---------------------------------------------------------------------------------------------------------------
def hot_path_attribution(step, steps=2):
    """Run `steps` extra (untimed) steps with HIP events around every C-ABI launch: per-entry-point GPU time.
    Returns {entry: ms per step}.  Overlapped launches (side streams) are counted at their own duration."""
    import torch
    from geot_amd.ext import _common
    rec = {}

    def bracket(launch, name):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        rec.setdefault(name, []).append((e0, e1))
    _common.trace = bracket
    try:
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
    finally:
        _common.trace = None
    return {k: sum(a.elapsed_time(b) for a, b in v) / steps for k, v in rec.items()}


HOST_ISSUE = {}     # seconds the host needed to queue the timed steps (the last call of timed_steps)
HOST_ISSUE_MAIN = {}


class ReplayGuard:
    """N > 1 only.  A hipGraph replay that holds an RCCL all-reduce has run here on ONE rank only (this pool has no multi-GPU
    box for the builder); if such a replay were to stall on a real node, every rank would sit in it until the driver's limit.
    The guard turns that into a finished run: the eager DistributedDataParallel step is timed FIRST (a short leg), and if the
    capture + replays that follow do not finish inside the deadline every rank leaves -- rank 0 after printing the bench line
    of that eager leg, marked as such.  Cancelled as soon as the timed replay region is over."""

    def __init__(self, seconds, line):
        import threading
        self.done = threading.Event()
        self.line = line
        self.thread = threading.Thread(target=self._watch, args=(seconds,), daemon=True)
        self.thread.start()

    def _watch(self, seconds):
        if self.done.wait(seconds):
            return
        sys.stderr.write("bench.py: the hipGraph replay did not finish in %.0f s -- reporting the eager leg measured before it\n" % seconds)
        sys.stderr.flush()
        if self.line is not None:
            print(self.line, flush=True)
        os._exit(0)

    def cancel(self):
        self.done.set()


def timed_steps(step, steps, dev, rehearsal):
    """barrier + synchronize, `steps` steps, synchronize + barrier; MAX over ranks of the elapsed seconds."""
    import torch
    from geot_amd import dist_utils
    torch.cuda.synchronize()
    dist_utils.barrier()
    torch.cuda.synchronize()
    t0, c0 = time.perf_counter(), time.process_time()
    for _ in range(steps):
        out = step()
    HOST_ISSUE["s"] = time.perf_counter() - t0          # the host has queued every step; the GPU may still be running
    HOST_ISSUE["cpu_s"] = time.process_time() - c0      # CPU time of all threads while queueing (blocked time excluded)
    torch.cuda.synchronize()
    dist_utils.barrier()
    torch.cuda.synchronize()
    return dist_utils.max_over_ranks(time.perf_counter() - t0, "cpu" if rehearsal else dev), out


def main():
    global N_POINTS
    args = parse()
    N_POINTS = args.points
    if args.workload == "backbone_ops":
        args.workload = "ops_only"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))

    import numpy as np
    import torch
    from geot_amd import dist_utils
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    world, rank, local = dist_utils.env_world()
    # GEOT_BENCH_REHEARSAL=1: N ranks share GPU 0 and talk over gloo -- rehearses the N > 1 control flow
    # (spawn, rendezvous, DDP all-reduce, barriers, MAX over ranks, rank-0 JSON) on a one-GPU box; never a measurement
    rehearsal = os.environ.get("GEOT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    elif world > 1 and torch.cuda.device_count() < world:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible (GEOT_BENCH_REHEARSAL=1 rehearses the control "
                         "flow on one GPU over gloo)" % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist_utils.init("gloo" if rehearsal else "nccl")
    # GEOT_BENCH_SOLO_DP=1 (one rank): the N > 1 code paths on ONE rank over RCCL -- DDP + SyncBatchNorm wrapped, the replay with
    # its captured all-reduce, the eager-first leg and its guard, the comm audit -- so that a one-GPU box executes every line
    # the multi-GPU run will (tests/test_dist_gpu.py); `multi` selects code paths, `world` stays the arithmetic
    solo = world == 1 and os.environ.get("GEOT_BENCH_SOLO_DP") == "1"
    if solo and not torch.distributed.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
        torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    multi = world > 1 or solo
    min_world = 1 if solo else 2

    from geot_amd import _lib, build as hip_build
    from geot_amd.synth import make_batch, region_labels
    if rank == 0:
        hip_build.build()          # mtime-incremental; a stale git-ignored .so must not be loaded as-is
        try:                       # the compiled torch bindings too, here and not lazily in every rank (same in-tree output files)
            from geot_amd import build_torch_ext
            build_torch_ext.build()
        except Exception as e:     # noqa: BLE001 -- no compiler: every rank binds through ctypes
            sys.stderr.write("bench.py: the compiled torch bindings did not build (%s); ctypes serves\n" % str(e)[:200])
    dist_utils.barrier()
    _lib.load()

    workload = args.workload
    gemm_file = None
    if workload in ("model", "fixmatch") and not args.no_tuned_gemm:
        from geot_amd import tuning
        gemm_file = tuning.enable(tune=args.tune_gemm is not None, path=args.tune_gemm)
    default_b = {"sa": 1, "fixmatch": 2}.get(workload, 8)
    B = args.clouds if args.clouds is not None else default_b
    clouds_per_step = B
    xyz_np, _ = make_batch(B, N_POINTS, start_index=dist_utils.cloud_range(rank, B)[0])
    xyz = torch.from_numpy(xyz_np).to(dev)
    fps_timer, gemm_timer, mlp_timer = EventTimer(), EventTimer(), EventTimer()
    feats_np = sa = model = None
    patch_owner = patch_name = None
    unpatch = []
    fps_rounds = 0
    fps_clouds = B
    ddp_modules = []
    parallelism = "independent clouds per rank, no collective"
    # N > 1: the step replayed from hipGraphs (no host on the critical path) needs its collectives captured -- RCCL only, and
    # never in the one-GPU rehearsal over gloo.  GEOT_BENCH_DP_GRAPH=0: the eager DistributedDataParallel step as primary.
    dp_graph = (multi and not rehearsal and workload in ("model", "fixmatch") and not args.no_graph
                and os.environ.get("GEOT_BENCH_DP_GRAPH", "1") != "0")
    trainer_g = None

    if workload == "model":
        from geot_amd.openpoints.models.backbone.transformer import PointTransformer_seg_T, TOOTH_SEG_CFG
        from geot_amd import train_step as ts
        from geot_amd.pointops.functions import pointops as pops
        torch.manual_seed(1609)
        model = PointTransformer_seg_T(**TOOTH_SEG_CFG, dense=args.dense).to(dev)
        dense_mode = model.dense
        net = ts.ddp(model, dev, unused=ts.UNUSED_SUPERVISED, min_world=min_world)
        ddp_modules = [net] if net is not model else []
        trainer = ts.SupervisedStep(net)
        target = torch.from_numpy(region_labels(xyz_np)).to(dev)
        cls = torch.from_numpy(np.random.default_rng(1609 + rank).integers(0, 2, size=(B, 1))).to(dev)
        patch_owner, patch_name = pops, "furthestsampling_uniform"
        fps_rounds = max(TOOTH_SEG_CFG["downsample_targets"]) - 1
        desc = ("configs[2]: B=%d x %dk-pt clouds, full transformer_finetune backbone PointTransformer_seg_T "
                "(trans_dim 384, depth 12, 512 groups x 32, targets 8192/4096/2048) fwd + Poly1FocalLoss + bwd + AdamW, "
                "train mode, random init" % (B, N_POINTS // 1000)) if not multi else \
               ("configs[3]: data-parallel %d x (B=%d x %dk-pt clouds), same model; SyncBatchNorm + DDP gradient "
                "all-reduce over %s" % (world, B, N_POINTS // 1000, "gloo (rehearsal)" if rehearsal else "RCCL"))
        if multi:
            parallelism = "dp%d: DistributedDataParallel (25 MB buckets, overlapped with backward) + SyncBatchNorm" % world

        # two batches, alternating, as a loader with one batch of look-ahead delivers them: the step is told the NEXT batch's
        # coordinates and queues their sampling / grouping / index work beside its own backward (SupervisedStep, same
        # results); --no-lookahead: every step does all of its own work between its own start and end
        xyz_b_np, _ = make_batch(B, N_POINTS, start_index=dist_utils.cloud_range(rank, B)[0] + 100003)
        batches = [(xyz, cls, target),
                   (torch.from_numpy(xyz_b_np).to(dev), cls.flip(0).contiguous(), torch.from_numpy(region_labels(xyz_b_np)).to(dev))]
        turn = [0]
        lookahead = not args.no_lookahead

        def step():
            cur, nxt = batches[turn[0] % 2], batches[(turn[0] + 1) % 2]
            turn[0] += 1
            return trainer(cur[0], cur[1], cur[2], next_pos=nxt[0] if lookahead else None)
        if dp_graph:
            # N > 1 without the host: the same model on the bare SyncBatchNorm-converted module with ONE flat gradient
            # all-reduce between backward and optimizer (train_step.sync_only / GradSync) -- the form a hipGraph can hold
            model_g = PointTransformer_seg_T(**TOOTH_SEG_CFG, dense=args.dense).to(dev)
            model_g.load_state_dict(model.state_dict())
            net_g = ts.sync_only(model_g, min_world=min_world)
            trainer_g = ts.SupervisedStep(net_g, grad_sync=ts.GradSync([net_g], torch.distributed.group.WORLD))
    elif workload == "fixmatch":
        from geot_amd import train_step as ts
        torch.manual_seed(1609)
        trainer = ts.build_fixmatch(dev, use_ddp=True, min_world=min_world,
                                    group=torch.distributed.group.WORLD if multi else None)
        if multi:
            ddp_modules = [trainer.model, trainer.T_predictor]
        if dp_graph:
            torch.manual_seed(1609)         # (the same initial weights as `trainer`)
            trainer_g = ts.build_fixmatch(dev, graph_sync=True, min_world=min_world, group=torch.distributed.group.WORLD)
        from geot_amd.pointops.functions import pointops as pops
        patch_owner, patch_name = pops, "furthestsampling_uniform"      # the student's 8192-sample FPS: its largest kernel
        from geot_amd.openpoints.models.backbone.transformer import TOOTH_SEG_CFG
        fps_rounds = max(TOOTH_SEG_CFG["downsample_targets"]) - 1
        bl = bu = B
        fps_clouds = bl + 2 * bu              # the student batch: labelled + strong + weak views (train.py:478-490)
        fps_timer.only = lambda flat, b, n, k: b == fps_clouds
        clouds_per_step = bl + bu
        xyz_u_np, _ = make_batch(bu, N_POINTS, start_index=10_000 + dist_utils.cloud_range(rank, bu)[0])
        xyz_u = torch.from_numpy(xyz_u_np).to(dev)
        rng = np.random.default_rng(1609 + rank)
        strong = (xyz_u * torch.from_numpy(rng.uniform(0.8, 1.2, size=(bu, 1, 3)).astype(np.float32)).to(dev)).contiguous()
        data = {"pos": xyz, "x": xyz.transpose(1, 2).contiguous(), "cls": torch.zeros(bl, 1, dtype=torch.long, device=dev),
                "y": torch.from_numpy(region_labels(xyz_np)).to(dev)}
        data_u = {"pos_w": xyz_u, "x_w": xyz_u.transpose(1, 2).contiguous(), "cls_w": torch.zeros(bu, 1, dtype=torch.long, device=dev),
                  "pos_s": strong, "x_s": strong.transpose(1, 2).contiguous(), "cls_s": torch.zeros(bu, 1, dtype=torch.long, device=dev),
                  "raw_pos": xyz_u}
        desc = ("configs[4]: FixMatch+NTM semi-supervised step per rank: teacher fwd on %d weak clouds, student fwd+bwd on "
                "%d labelled + %d strong + %d weak clouds x %dk pts, class transition + sig_t_mean + logit correction + "
                "threeD_space_loss(k=32) + Poly1Focal losses, AdamW x2" % (bu, bl, bu, bu, N_POINTS // 1000))
        if multi:
            parallelism = "dp%d: DDP(student) + DDP(T_predictor) + SyncBatchNorm + all-gather of the class anchors" % world

        # a second pair of batches: the iterations alternate, and each is told the next one's batches (FixMatchNTMStep look-ahead)
        xyz_l2_np, _ = make_batch(bl, N_POINTS, start_index=dist_utils.cloud_range(rank, bl)[0] + 100003)
        xyz_u2_np, _ = make_batch(bu, N_POINTS, start_index=110_003 + dist_utils.cloud_range(rank, bu)[0])
        xyz_l2, xyz_u2 = torch.from_numpy(xyz_l2_np).to(dev), torch.from_numpy(xyz_u2_np).to(dev)
        strong2 = (xyz_u2 * torch.from_numpy(rng.uniform(0.8, 1.2, size=(bu, 1, 3)).astype(np.float32)).to(dev)).contiguous()
        data2 = {"pos": xyz_l2, "x": xyz_l2.transpose(1, 2).contiguous(), "cls": data["cls"], "y": torch.from_numpy(region_labels(xyz_l2_np)).to(dev)}
        data_u2 = {"pos_w": xyz_u2, "x_w": xyz_u2.transpose(1, 2).contiguous(), "cls_w": data_u["cls_w"], "pos_s": strong2,
                   "x_s": strong2.transpose(1, 2).contiguous(), "cls_s": data_u["cls_s"], "raw_pos": xyz_u2}
        batches = [(data, data_u), (data2, data_u2)]
        turn = [0]
        lookahead = not args.no_lookahead

        def step():
            cur, nxt = batches[turn[0] % 2], batches[(turn[0] + 1) % 2]
            turn[0] += 1
            return trainer(cur[0], cur[1], next_batches=nxt if lookahead else None)["loss"]
    elif workload == "sa":
        from geot_amd.pointnet2.pointnet2_modules import PointnetSAModuleVotes
        import geot_amd.pointnet2.pointnet2_modules as mods
        feats_np = np.random.default_rng(1609 + rank).standard_normal((B, MLP[0], N_POINTS)).astype(np.float32)
        feats = torch.from_numpy(feats_np).to(dev)
        torch.manual_seed(1609)
        sa = PointnetSAModuleVotes(mlp=list(MLP), npoint=NPOINT, radius=RADIUS, nsample=NSAMPLE, use_xyz=True).to(dev).eval()
        patch_owner, patch_name = mods.pointnet2_utils, "furthest_point_sample"
        fps_rounds, desc = NPOINT - 1, ("configs[1]: PointNet++ SetAbstraction fwd (FPS 24000->6000, ball_query r=0.1 "
                                        "ns=32, group, SharedMLP [6,64,64,128], max), eval")

        def step():
            with torch.no_grad():
                return sa(xyz, feats)[1]
    elif workload == "ops_only":
        from geot_amd import workloads as wl
        from geot_amd.pointops.functions import pointops as pops
        hot = wl.BackboneHotPath().to(dev)
        tokens = torch.randn(B, wl.TRANS_DIM, wl.GROUPS, device=dev)
        patch_owner, patch_name = pops, "furthestsampling_uniform"
        fps_rounds, desc = 8191, ("STAND-IN, not a BASELINE config: the sampling / grouping / interpolation ops of "
                                  "PointTransformer_seg_T fwd+bwd (FPS 512+8192, kNN 32/4, three_nn+interpolate x3) with "
                                  "every dense layer replaced by nothing -- kernel timing only")

        def step():
            return wl.backbone_hotpath_step(hot, xyz, tokens)
    else:
        from geot_amd import workloads as wl
        from geot_amd.synth import make_logits
        nt = wl.NtmHotPath().to(dev)
        pw = torch.from_numpy(make_logits(xyz_np, index=2 * rank)).to(dev)
        ps = torch.from_numpy(make_logits(xyz_np, index=2 * rank + 1, sharp=3.0)).to(dev)
        desc = ("configs[4] NTM half-step: sig_t_mean + class transition + logit correction + "
                "threeD_space_loss(k=32) fwd+bwd on B_u clouds")

        def step():
            return wl.ntm_step(nt, xyz, pw, ps)

    graph_note = ""
    # model / fixmatch on one GPU: the iteration can be issued EAGERLY (one kernel launch at a time from the host) or REPLAYED
    # from two single-stream hipGraphs (geot_amd/graph_step.py: M = forward, losses, backward, AdamW over static buffers on
    # the current stream; P = the batch-only work of the NEXT batch on a side stream beside it; same kernels, same bits:
    # tests/test_graph_step_gpu.py).  Both are timed in every run and both are in the JSON; `value` is the PRIMARY mode, fixed
    # here and not picked after the fact: the replay where it was measured ahead -- the FixMatch+NTM iteration (30.4 against
    # 31.1 ms: its ~1500 launches keep the eager host on the critical path) and the supervised step at <= 3 clouds (the eager
    # step is host-bound at 17-19 ms; 10 / 13.5 / 18.6 ms replayed) -- the eager step otherwise (4-8 clouds: GPU-bound either
    # way; the replay 2 % behind at 4 clouds, level at 8 -- and the dominant kernel's HIP events, which the roofline block
    # needs from the timed region, can only be recorded on eager launches).
    # --graph / --no-graph force the primary mode (--no-graph also skips the replay leg).
    can_replay = workload in ("model", "fixmatch") and (not multi or dp_graph) and not args.no_graph
    # N > 1: the replay is the primary mode whenever it captures (on every rank: `agree`) -- an eager rank needs ~1.5 host cores
    # continuously, eight of them share one host
    use_graph = can_replay and (args.graph or workload == "fixmatch" or B <= 3 or multi)
    eager_step = step
    graphed = replay_step = None
    replay_refused = None

    def agree(ok):
        flag = torch.tensor([1.0 if ok else 0.0], device=dev)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
        return bool(flag.item() > 0.5)
    if can_replay:
        from geot_amd import graph_step as gs
        try:
            graphed = (gs.GraphedSupervisedStep if workload == "model" else gs.GraphedFixMatchStep)(
                trainer_g if multi else trainer, agree=agree if multi else None)
        except RuntimeError as e:                   # e.g. packet capture left on by the environment without GEOT_GRAPH_LAUNCH=fast
            replay_refused, use_graph, can_replay = "%s: %s" % (type(e).__name__, str(e)[:600]), False, False
        if multi and not agree(graphed is not None):          # one verdict for all ranks, before any of them replays
            replay_refused = replay_refused or "another rank could not construct its replay"
            graphed, use_graph, can_replay = None, False, False
    if can_replay:

        def replay_step():
            cur, nxt = batches[turn[0] % 2], batches[(turn[0] + 1) % 2]
            turn[0] += 1
            if workload == "model":
                return graphed(cur[0], cur[1], cur[2], next_pos=nxt[0] if lookahead else None)
            return graphed(cur[0], cur[1], next_batches=nxt if lookahead else None)["loss"]
    guard = None
    if use_graph and multi:
        k_g = max(2, min(args.steps, 5))
        for _ in range(2):
            eager_step()
        t_g, out_g = timed_steps(eager_step, k_g, dev, rehearsal)
        fallback = {"metric": "point-clouds/sec (24k pts, 17 classes) fwd+bwd", "value": world * clouds_per_step * k_g / t_g,
                    "unit": "clouds/s", "n_gpus": world, "steps": k_g, "warmup": 2, "ms_per_step": 1e3 * t_g / k_g,
                    "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                    "config": {"workload": desc + "; eager DistributedDataParallel step -- the hipGraph replay that is the "
                                           "primary mode at N > 1 did not finish inside its deadline on this node (ReplayGuard)",
                               "clouds_per_gpu": clouds_per_step, "points": N_POINTS, "parallelism": parallelism,
                               "stand_in": False},
                    "roofline": None, "cpu_baseline": None, "final_loss": float(out_g.float().mean())}
        guard = ReplayGuard(float(os.environ.get("GEOT_BENCH_REPLAY_DEADLINE", "300")) + 2.0 * args.steps,
                            json.dumps(_finite(fallback), allow_nan=False) if rank == 0 else None)
    if use_graph:
        try:
            for _ in range(graphed.warmup + 1):     # eager over the static buffers, then the capture + first replay
                replay_step()
            assert graphed.captured
            step = replay_step
            graph_note = "; the iteration replayed from single-stream hipGraphs (static buffers, batch copied in per step)"
        except RuntimeError as e:                   # e.g. fast launch mode and a graph that is not kernel-only: eager, and say so
            replay_refused, use_graph, can_replay = "%s: %s" % (type(e).__name__, str(e)[:600]), False, False
            step = eager_step
            torch.cuda.synchronize()
    if args.graph and not can_replay and replay_refused is None:
        assert workload in ("sa", "ntm"), "--graph: sa / ntm capture one step here; model / fixmatch have the replay of graph_step.py"
        from geot_amd import streams
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(3):
                eager_step()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize()
        hip_graph = torch.cuda.CUDAGraph(keep_graph=True)
        with streams.capture(hip_graph, dev):
            graph_out = eager_step()
        kinds = streams.node_types(hip_graph)
        if not geot_amd.graph_replay_is_safe() and set(kinds) - {"kernel"}:
            raise SystemExit("bench.py --graph: the captured step holds %s; only kernel nodes are known to replay correctly unless "
                             "the launcher exported DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 (geot_amd/__init__.py)" % kinds)

        def step():
            hip_graph.replay()
            return graph_out
        patch_owner = None          # per-kernel HIP events cannot be recorded inside a replay
        graph_note = "; one step captured into a hipGraph, replays timed"
    for _ in range(args.warmup):
        step()

    def install_timers():
        """HIP events around the dominant kernel's launch (and, model: the decoder's widest GEMM) -- on eager steps only:
        events cannot be recorded inside a graph replay (torch's `external` events are refused by this runtime), so under
        the default replay the kernel timings come from the eager leg that follows the timed region."""
        undo = []
        if patch_owner is not None:
            orig_fn = getattr(patch_owner, patch_name)
            setattr(patch_owner, patch_name, fps_timer.wrap(orig_fn))
            undo.append(lambda: setattr(patch_owner, patch_name, orig_fn))
        if workload == "model":
            # the decoder's widest GEMM (1536 -> 384 over B*N points): a module call in the channels-first layout, a
            # pointwise_from_cl() call (same rocBLAS GEMM, transposed operand) in the point-major one
            hooks = gemm_timer.hook(model.propogation_0.mlp.layer1.conv)
            import geot_amd.openpoints.models.backbone.transformer as tr_mod
            orig_from_cl = tr_mod.pointwise_from_cl
            gemm_timer.only = lambda w, z, **kw: tuple(w.shape) == (384, 1536) and z.shape[1] == N_POINTS
            tr_mod.pointwise_from_cl = gemm_timer.wrap(orig_from_cl)
            undo.append(lambda: [setattr(tr_mod, "pointwise_from_cl", orig_from_cl)] + [h.remove() for h in hooks])
        if workload == "sa" and not args.graph:
            import geot_amd.sa_fused as sa_fused_mod
            orig_mlp = sa_fused_mod.fused_group_mlp_max
            sa_fused_mod.fused_group_mlp_max = mlp_timer.wrap(orig_mlp)
            undo.append(lambda: setattr(sa_fused_mod, "fused_group_mlp_max", orig_mlp))
        return undo

    undo = [] if use_graph else install_timers()
    if args.streams > 1 and workload == "sa":
        pool = [torch.cuda.Stream(device=dev) for _ in range(args.streams)]
        dealt = [0]

        def dealt_step():
            with torch.cuda.stream(pool[dealt[0] % args.streams]):
                dealt[0] += 1
                return step()
        elapsed, out = timed_steps(dealt_step, args.steps, dev, rehearsal)
    else:
        elapsed, out = timed_steps(step, args.steps, dev, rehearsal)
    if guard is not None:
        guard.cancel()
    HOST_ISSUE_MAIN["ms"] = 1e3 * HOST_ISSUE["s"] / max(args.steps, 1)
    HOST_ISSUE_MAIN["cpu_ms"] = 1e3 * HOST_ISSUE["cpu_s"] / max(args.steps, 1)
    assert torch.isfinite(out).all()
    other_leg = None
    if can_replay:
        # the same iterations in the OTHER mode, in this run (same model state continuing): what the replay is worth at this
        # size.  The per-kernel HIP events are taken on eager steps -- no event can be recorded inside a replay.
        k_e = max(2, min(args.steps, 10))
        if use_graph:
            undo = install_timers()
            other = eager_step
            other()                             # untimed: the eager look-ahead restarts here
        else:
            for u in undo:
                u()
            undo = []
            other = replay_step
            try:
                for _ in range(graphed.warmup + 2):  # eager over the static buffers, capture, first replays
                    other()
                assert graphed.captured
            except RuntimeError as e:            # the timed (eager) region is done: a failed replay leg is reported, not fatal
                replay_refused, other = "%s: %s" % (type(e).__name__, str(e)[:600]), None
                torch.cuda.synchronize()
    if can_replay and other is not None:
        t_e, out_e = timed_steps(other, k_e, dev, rehearsal)
        assert torch.isfinite(out_e).all()
        other_leg = {"ms_per_step": 1e3 * t_e / k_e, "clouds_per_s": clouds_per_step * k_e / t_e,
                     "host_issue_ms_per_step": 1e3 * HOST_ISSUE["s"] / k_e,
                     "host_cpu_ms_per_step": 1e3 * HOST_ISSUE["cpu_s"] / k_e, "steps": k_e,
                     "note": ("the same iterations, same model state continuing, launched kernel by kernel from the host"
                              if use_graph and not multi else
                              "the eager DistributedDataParallel step (25-MB buckets overlapped with the backward) on the same "
                              "batches, its own copy of the model" if use_graph else
                              "the same iterations, same model state continuing, replayed from single-stream hipGraphs "
                              "(geot_amd/graph_step.py)")}
    for u in undo:
        u()

    ms_per_step = 1e3 * elapsed / args.steps
    fps_ms = fps_timer.mean_ms()
    fps_alone_ms = None
    if workload == "model" and fps_rounds:
        # the same launch with the chip to itself: inside the step the kernel shares its CUs with the GEMMs of the stream it
        # runs beside (the previous batch's backward under look-ahead, the encoder otherwise) and its loop waits for issue
        # slots -- the in-step duration above is what the roofline uses, this one says how much of it is contention
        alone = EventTimer()
        solo = alone.wrap(pops.furthestsampling_uniform)
        torch.cuda.synchronize()
        for _ in range(3):
            solo(xyz.reshape(-1, 3), B, N_POINTS, fps_rounds + 1)
            torch.cuda.synchronize()
        fps_alone_ms = alone.mean_ms()
    fps_flop = fps_clouds * N_POINTS * fps_rounds * FPS_FLOP_PER_UPDATE
    fps_tf = fps_flop / (fps_ms * 1e-3) / 1e12 if fps_rounds else float("nan")
    tag = {"model": "bench_model", "sa": "bench_sa"}.get(workload, workload)
    traffic, source = pmc_traffic("fps_pruned_kernel", tag) if (B == default_b and workload != "model") else (None, None)
    fps_roofline = {
        "kernel": "fps_pruned_kernel", "bound": "valu", "achieved": fps_tf, "peak": FP32_VECTOR_PEAK_TFLOPS,
        "unit": "TFLOP/s", "frac": fps_tf / FP32_VECTOR_PEAK_TFLOPS, "traffic": traffic, "traffic_source": source,
        "avg_launch_ms": fps_ms, "alone_launch_ms": fps_alone_ms, "cus_used": fps_clouds,
        # `frac` divides by all 256 CUs and is capped at clouds / 256 whatever the kernel does (one workgroup per cloud); what the
        # occupied CUs actually execute comes from the committed SQ counter pass: vector-issue share of their busy cycles
        "valu_issue_share_of_occupied_cus": sq_share("fps_pruned_kernel<768, 32, false", tag if workload != "model" else "bench_model"),
        "note": "FPS is fp32-VALU / round-latency bound, not HBM or MFMA bound; algorithmic flop = clouds*N*(m-1) updates "
                "* 10 (what the reference executes); the pruned kernel skips most of them exactly (so `frac` is not utilisation: "
                "see valu_issue_share_of_occupied_cus); one workgroup (one CU of 256) per cloud; avg_launch_ms is the launch as it ran inside the step (side stream, sharing its CUs "
                "with the main stream's GEMMs) -- under the default hipGraph replay taken on the eager leg of the same run, "
                "where HIP events can bracket one launch; alone_launch_ms the same launch with the chip to itself"}
    result = {
        "metric": "point-clouds/sec (24k pts, 17 classes) fwd+bwd" if workload in ("model", "fixmatch") else
                  "point-clouds/sec (24k pts, 17 classes)",
        "value": world * clouds_per_step * args.steps / elapsed,
        "unit": "clouds/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "host_issue_ms_per_step": HOST_ISSUE_MAIN.get("ms"),   # wall time until the host has queued a step (rank 0); it
        # includes time BLOCKED on a full hardware queue, so it approaches ms_per_step whenever the GPU is the bottleneck
        "host_cpu_ms_per_step": HOST_ISSUE_MAIN.get("cpu_ms"),  # CPU time (all threads) spent queueing a step: the host's cost
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic" if not rehearsal else "synthetic (REHEARSAL: all ranks on one GPU over gloo -- not a measurement)",
        "config": {"workload": desc + graph_note, "clouds_per_gpu": clouds_per_step, "points": N_POINTS,
                   "parallelism": parallelism, "stand_in": workload in ("ops_only", "ntm")},
        "roofline": fps_roofline if (fps_rounds and not args.graph) else None,
        "final_loss": float(out.float().mean()) if workload in ("model", "fixmatch") else None,   # of the last timed step (rank 0)
    }
    if workload == "model":
        # dominant kernel of the step by rocprofv3 time (profiles/r02_bench_model_*_window.csv): the 8192-sample FPS
        # (one workgroup per cloud, on the side stream beside the encoder).  Next to it the widest library GEMM of the
        # decoder (propogation_0.mlp.layer1: 1536 -> 384 over B*24000 points), timed from module hooks.
        t_fps, s_fps = pmc_traffic("fps_pruned_kernel<768, 32, false", tag) if B == default_b else (None, None)
        fps_roofline.update(kernel="fps_pruned_kernel<768,32,false,8> (pointops.fps 24000 -> 8192, K2 semantics)",
                            traffic=t_fps, traffic_source=s_fps)
        g_ms = gemm_timer.mean_ms()
        g_flop = 2.0 * 384 * 1536 * B * N_POINTS
        g_tf = g_flop / (g_ms * 1e-3) / 1e12
        result["roofline_secondary"] = {
            "kernel": "propogation_0.mlp.layer1.conv (1x1 conv 1536->384 over B*N points = one rocBLAS fp32 GEMM; its input "
                      "is point-major (B, N, 1536) under the default GEOT_FP_LAYOUT=cl, a transposed operand)",
            "bound": "mfma", "achieved": g_tf, "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": g_tf / FP32_MATRIX_PEAK_TFLOPS, "traffic": None, "avg_launch_ms": g_ms,
            "note": "algorithmic flop = 2*384*1536*B*N per forward launch; library kernel (dense layers are stock "
                    "PyTorch -> rocBLAS, SURVEY.md 2.1 row 12), timed with HIP events from module hooks"}
        att = hot_path_attribution(eager_step)
        hot_ms = sum(att.values())
        top = dict(sorted(att.items(), key=lambda kv: -kv[1])[:8])
        result["hot_path"] = {"c_abi_gpu_ms_per_step": hot_ms, "share_of_step": hot_ms / ms_per_step,
                              "dense_and_glue_share": 1.0 - hot_ms / ms_per_step, "top_entry_points_ms": top,
                              "note": "sum of HIP-event durations around every C-ABI launch in 2 extra untimed steps; the "
                                      "8192-point FPS runs on a side stream beside the encoder, so shares are of GPU work, "
                                      "not of wall time"}
        # the step's largest HBM-bound hand-written kernel: the point-major interpolation + BatchNorm gradient of the three FP
        # stages (gather_rows_csr_bn_cl: rows of 4 C bytes; reads y and dz of the n unknown points, writes the m known ones)
        gr_ms = att.get("geot_gather_rows_csr_bn_cl")
        if gr_ms:
            c_fp = 4 * TOOTH_SEG_CFG["trans_dim"]
            tg = TOOTH_SEG_CFG["downsample_targets"]
            stages = [(N_POINTS, tg[0]), (tg[0], TOOTH_SEG_CFG["num_group"]), (tg[1], TOOTH_SEG_CFG["num_group"])]   # (n, m) of prop0 / 1 / 2
            gr_bytes = sum(4.0 * B * c_fp * (2 * n_ + m_) for n_, m_ in stages)
            # (24000 <- 8192 is the list walk, the two 512-target stages the few-target form: per-launch PMC bytes of both kernels,
            # weighted by their launch counts in the profiled run)
            t3 = s3 = None
            if B == default_b:
                parts = [pmc_traffic(k, tag) + (pmc_launches(k, tag),) for k in ("gather_rows_csr_bn_cl_kernel", "gather_rows_chunks_bn_cl_kernel")]
                parts = [p for p in parts if p[0] is not None]
                if parts and all(p[2] for p in parts):
                    t3 = sum(p[0] * p[2] for p in parts) / sum(p[2] for p in parts)
                    s3 = parts[0][1]
                elif parts:
                    t3, s3 = parts[0][0], parts[0][1]
            result["roofline_hbm"] = {
                "kernel": "gather_rows_csr_bn_cl_kernel + gather_rows_chunks_bn_cl_kernel (geot_gather_rows_csr_bn_cl, 3 launches per step: "
                          "FP stages 24000<-8192, 8192<-512, 4096<-512 at C = %d)" % c_fp,
                "bound": "hbm", "achieved": gr_bytes / (gr_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": gr_bytes / (gr_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": t3, "traffic_source": s3,
                "algorithmic_bytes_per_launch": gr_bytes / 3, "avg_launch_ms": gr_ms / 3,
                "note": "algorithmic bytes = 4 C B (2 n + m) per stage (y and dz rows in, gradient rows out; the index is < 1 %); "
                        "HIP events around the three C-ABI launches of 2 extra eager steps (hot_path); traffic = PMC bytes per "
                        "launch averaged over the three stages"}
        f_ms = att.get("geot_furthestsampling_offset", float("nan"))
        k_ms = att.get("geot_knn_sorted_ws", float("nan"))
        result["micro"] = {"fps_24000_to_8192_Mpoints_per_s": B * N_POINTS / (fps_ms * 1e-3) / 1e6,
                           "fps_all_launches_ms_per_step": f_ms + att.get("geot_furthest_point_sampling", 0.0),
                           "knn_all_launches_ms_per_step": k_ms,
                           "knn_Mqueries_per_s": B * (512 + 4096 * 2 + 8192 * 2) / (k_ms * 1e-3) / 1e6,
                           "note": "kNN: k=32 512x24000 + k=4 {512->4096, 4096^2, 4096->8192, 8192^2} per cloud per step"}
        result["dense"] = {"mode": dense_mode,
                           "gemm_selection": ("PyTorch TunableOp, recorded: " + os.path.relpath(gemm_file, ROOT)) if gemm_file
                           else "library defaults",
                           "reference_gflop_fwd_per_cloud": DENSE_GFLOP_FWD_REFERENCE,
                           "note": "factored: first 1x1 conv behind a gather evaluated before the gather (same function, "
                                   "fewer flops); reference: the reference's op order"}
    if workload == "sa" and mlp_timer.pairs:
        mlp_ms = mlp_timer.mean_ms()
        widths = [MLP[0] + 3] + MLP[1:]   # use_xyz: 3 relative coordinates + 3 features
        mlp_flop = 2.0 * B * NPOINT * NSAMPLE * sum(a * b for a, b in zip(widths[:-1], widths[1:]))
        mlp_tf = mlp_flop / (mlp_ms * 1e-3) / 1e12
        t2, s2 = pmc_traffic("sa_group_mlp_max_kernel", tag) if B == 1 else (None, None)
        result["roofline_secondary"] = {
            "kernel": "sa_group_mlp_max_kernel", "bound": "mfma", "achieved": mlp_tf, "peak": FP32_MATRIX_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": mlp_tf / FP32_MATRIX_PEAK_TFLOPS, "traffic": t2, "traffic_source": s2,
            "avg_launch_ms": mlp_ms,
            "note": "fused group + [6,64,64,128] 1x1-conv stack on fp32 MFMA + max over nsample; algorithmic flop = "
                    "2*rows*sum(Cin*Cout) with the unpadded 3+3 input channels"}
    if workload == "fixmatch":
        fps_roofline.update(kernel="fps_pruned_kernel<768,32,false,8> (the student's pointops.fps 24000 -> 8192 over its "
                                   "%d clouds: the longest kernel of the iteration; the teacher runs the same on %d)" % (fps_clouds, bu))
        att = hot_path_attribution(eager_step)
        hot_ms = sum(att.values())
        result["hot_path"] = {"c_abi_gpu_ms_per_step": hot_ms, "share_of_step": hot_ms / ms_per_step,
                              "top_entry_points_ms": dict(sorted(att.items(), key=lambda kv: -kv[1])[:8]),
                              "note": "HIP-event durations around every C-ABI launch in 2 extra untimed iterations; the FPS "
                                      "launches and the teacher run on side streams, so this is GPU work, not wall time"}
    if multi:
        # Audit trail for a multi-GPU record: (a) an all-reduce of ones over the backend the step used -- the number of
        # ranks that really took part; (b) what the gradient all-reduce costs on the critical path: the same steps with
        # DDP's reduction switched off (no_sync: gradients stay local; SyncBatchNorm and the anchor exchange still talk)
        import contextlib
        import torch.distributed as dist
        ones = torch.ones(1, device="cpu" if rehearsal else dev)
        dist.all_reduce(ones)
        k2 = max(2, min(args.steps, 5))

        def local_step():          # (the eager DDP step: a replay holds its all-reduce inside the graph)
            with contextlib.ExitStack() as stack:
                for m in ddp_modules:
                    stack.enter_context(m.no_sync())
                return eager_step()
        local_step()
        t_local, _ = timed_steps(local_step, k2, dev, rehearsal)
        eager_ms = other_leg["ms_per_step"] if (use_graph and other_leg) else ms_per_step
        grad_bytes = sum(p.numel() * 4 for m in ddp_modules for p in m.parameters() if p.requires_grad)
        result["comm"] = {"backend": dist.get_backend(), "ranks_in_collective": int(ones.item()),
                          "gradient_allreduce_mb_per_step": grad_bytes / 1e6,
                          "ms_per_step_without_gradient_allreduce": 1e3 * t_local / k2,
                          "exposed_allreduce_ms": eager_ms - 1e3 * t_local / k2,
                          "note": "exposed = the eager DDP step minus the same step under DDP.no_sync() (%d steps); negative = noise" % k2}
        # What a scaling line needs to be attributed: the host side of every rank (N ranks share one host: issue time is
        # the first suspect when N x the one-GPU rate is not reached), the collectives a step really issues, and the step
        # with the look-ahead off (its FPS / index kernels run beside RCCL's all-reduce kernels when it is on)
        gdev = "cpu" if rehearsal else dev
        issue = dist_utils.gather_over_ranks(HOST_ISSUE_MAIN["ms"], gdev)
        cpu = dist_utils.gather_over_ranks(HOST_ISSUE_MAIN["cpu_ms"], gdev)
        cores = dist_utils.gather_over_ranks(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 0, gdev)
        counts = {}
        real = {n: getattr(dist, n) for n in ("all_reduce", "all_gather", "all_gather_into_tensor", "broadcast")}

        def counted(name):
            def fn(*a, **k):
                counts[name] = counts.get(name, 0) + 1
                return real[name](*a, **k)
            return fn
        for n_ in real:
            setattr(dist, n_, counted(n_))
        try:
            eager_step()           # one untimed EAGER step with the python-level collectives counted (SyncBatchNorm in bn_act,
            # anchors); a replayed step issues the same ones from inside its graphs, plus GradSync's one flat all-reduce
        finally:
            for n_, f_ in real.items():
                setattr(dist, n_, f_)
        torch.cuda.synchronize()
        result["ranks"] = {"host_issue_ms_per_step": {"max": max(issue), "mean": sum(issue) / len(issue), "per_rank": issue},
                           "host_cpu_ms_per_step": {"max": max(cpu), "mean": sum(cpu) / len(cpu)},
                           "cores_visible_per_rank": cores, "host_cpu_count": os.cpu_count(),
                           "python_level_collectives_per_step": counts,
                           "note": "python_level_collectives: the all-reduces / all-gathers this package issues itself in one step "
                                   "(SyncBatchNorm statistics each way in fused_norm.bn_act, the class-anchor exchange); DDP's "
                                   "bucketed gradient all-reduces are issued from C++ and are not in this count"}
        if workload in ("model", "fixmatch") and lookahead:
            def plain_ddp_step():
                cur = batches[turn[0] % 2]
                turn[0] += 1
                return trainer(cur[0], cur[1], cur[2]) if workload == "model" else trainer(cur[0], cur[1])["loss"]
            plain_ddp_step()
            t_pl, _ = timed_steps(plain_ddp_step, k2, dev, rehearsal)
            result["comm"]["ms_per_step_without_lookahead"] = 1e3 * t_pl / k2
            result["comm"]["lookahead_note"] = ("the same DDP step with next_pos / next_batches withheld (%d steps): lower than "
                                                "ms_per_step = the look-ahead's kernels hurt beside the collectives" % k2)
    if workload in ("model", "fixmatch"):
        result["graph"] = {"replayed": bool(use_graph),
                           "note": ("primary mode = hipGraph replay (geot_amd/graph_step.py: training graph + the next batch's "
                                    "look-ahead graph on a side stream, bit-identical to the eager step, "
                                    "tests/test_graph_step_gpu.py); the eager leg of the same run is `eager`") if use_graph else
                                   ("primary mode = eager: N > 1 runs DistributedDataParallel, whose buckets and collectives are "
                                    "host logic" if multi else
                                    "primary mode = eager" + ("; the hipGraph replay of the same iterations is `replay` (primary for "
                                                              "the FixMatch iteration and at <= 3 clouds, where the eager "
                                                              "step is host-bound)" if can_replay else
                                                             (" (the replay was refused: `replay_refused`)" if replay_refused else " (--no-graph)")))}
        if other_leg is not None:
            result["eager" if use_graph else "replay"] = other_leg
        if graphed is not None and graphed.node_types:
            result["graph"]["launch_mode"] = ("safe" if geot_amd.graph_replay_is_safe() else
                                              "fast" if geot_amd.GRAPH_LAUNCH == "fast" else "inspected")
            result["graph"]["nodes"] = graphed.node_types      # {"P": {"kernel": n}, "M": {...}}: hipGraphGetNodes on the captures
            result["graph"]["launch_note"] = (
                "fast = the runtime's graph packet capture left on (0.5 ms of host time per launch); correct only for graphs of "
                "kernel nodes alone, which graph_step verifies per capture (`nodes`); safe = packet capture exported off by the "
                "launcher (DEBUG_CLR_GRAPH_PACKET_CAPTURE=0), any graph, 7-20 ms per launch; inspected = the switch is whatever "
                "the environment left it and cannot be verified, so captures are held to kernel nodes as in fast mode "
                "(geot_amd/__init__.py)")
        if replay_refused:
            result["graph"]["replay_refused"] = replay_refused
        result["config"]["lookahead"] = ("the step is handed the next batch's coordinates (two batches alternate) and queues "
                                         "their sampling / grouping / index work beside its own backward"
                                         if lookahead else "off: all of a batch's work inside its own step")
    if workload in ("model", "fixmatch") and not multi and lookahead and not args.no_dense_reference:
        # the same steps without the look-ahead (every batch's sampling at the head of its own step), observed in this run
        runner = graphed if use_graph else trainer      # (the replay's "plain" variant is captured by the first call)

        def plain_step():
            cur = batches[turn[0] % 2]
            turn[0] += 1
            return runner(cur[0], cur[1], cur[2]) if workload == "model" else runner(cur[0], cur[1])["loss"]
        plain_step()
        k3 = min(args.steps, 10)
        t_plain, _ = timed_steps(plain_step, k3, dev, rehearsal)
        result["lookahead"] = {"ms_per_step_without": 1e3 * t_plain / k3, "clouds_per_s_without": clouds_per_step * k3 / t_plain,
                               "steps": k3, "note": "same model, same alternating batches, next_pos=None: every step samples "
                                                    "and groups its own batch before its encoder can start"}
    if workload == "model" and not multi and dense_mode != "reference" and not args.no_dense_reference:
        # the same step with every layer in the REFERENCE's op order (first 1x1 conv after the gather, op-by-op attention /
        # LayerNorm) on the same kernels: what the algebraic re-ordering is worth, observed in this run
        del trainer, net
        graphed = runner = eager_step = step = None      # (the replay holds the trainer and its graph's memory pool)
        torch.cuda.empty_cache()
        torch.manual_seed(1609)
        ref_model = PointTransformer_seg_T(**TOOTH_SEG_CFG, dense="reference").to(dev)
        ref_model.load_state_dict(model.state_dict())
        ref_trainer = ts.SupervisedStep(ref_model)

        def ref_step():
            return ref_trainer(xyz, cls, target)
        ref_step()
        t_ref, ref_out = timed_steps(ref_step, 3, dev, rehearsal)
        assert torch.isfinite(ref_out).all()
        result["dense"]["reference_order"] = {"clouds_per_s": clouds_per_step * 3 / t_ref, "ms_per_step": 1e3 * t_ref / 3,
                                              "steps": 3, "note": "--dense reference: the reference's op order, same HIP "
                                                                  "kernels and GEMM selection; 1 warm-up + 3 timed steps"}
        del ref_trainer, ref_model
    if workload == "sa" and not multi and not args.graph and not args.no_saturated and B < 64:
        # one cloud occupies ONE of the 256 CUs in FPS (96 % of this step): the same module at 256 clouds per launch
        # is what the chip does when every CU has a cloud (the reference's per-cloud semantics are unchanged)
        bs = 256
        xs = torch.from_numpy(make_batch(bs, N_POINTS, start_index=5000)[0]).to(dev)
        fs = torch.randn(bs, MLP[0], N_POINTS, device=dev)

        def sat_step():
            with torch.no_grad():
                return sa(xs, fs)[1]
        sat_step()
        t_sat, _ = timed_steps(sat_step, 3, dev, rehearsal)
        result["saturated"] = {"clouds_per_launch": bs, "clouds_per_s": bs * 3 / t_sat, "ms_per_step": 1e3 * t_sat / 3,
                               "steps": 3, "note": "same SetAbstraction forward with one cloud per CU (256 clouds in one "
                                                   "call): throughput when FPS fills the chip"}
        del xs, fs
    if rank == 0 and not multi and not args.no_cpu_baseline:
        if workload == "sa":
            from geot_amd.pointnet2.pointnet2_modules import PointnetSAModuleVotes
            torch.manual_seed(1609)
            sa_cpu = PointnetSAModuleVotes(mlp=list(MLP), npoint=NPOINT, radius=RADIUS, nsample=NSAMPLE, use_xyz=True).eval()
            sa_cpu.load_state_dict(sa.state_dict())
            result["cpu_baseline"] = cpu_baseline_sa(xyz_np, feats_np, sa_cpu, args.cpu_steps)
        elif workload == "model":
            result["cpu_baseline"] = cpu_baseline_model()
        elif workload == "fixmatch":
            result["cpu_baseline"] = cpu_baseline_fixmatch(bl, bu)
    if (rank == 0 and not multi and workload == "model" and not args.no_also and args.clouds is None
            and args.points == N_POINTS and not args.graph and not args.no_graph and args.dense is None):
        # the driver's default invocation: configs[1] and configs[4] into the same record, after the primary leg's timing
        torch.cuda.empty_cache()
        result["also"] = also_legs(args)
    if rank == 0:
        print(json.dumps(_finite(result), allow_nan=False), flush=True)
    if multi:
        import torch.distributed as dist
        dist.barrier()                      # rank 0 may still be printing: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
