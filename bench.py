#!/usr/bin/env python
"""bench.py -- throughput of the GeoT sampling/grouping hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--clouds B] [--workload sa|backbone_ops|ntm]

Workload (default, BASELINE.json configs[1]): one 24 000-point synthetic tooth cloud per
GPU through a PointNet++ SetAbstraction forward (PointnetSAModuleVotes npoint=6000,
radius=0.1, nsample=32, mlp=[3,64,64,128], use_xyz) = furthest_point_sample ->
gather_operation -> ball_query -> fused {group xyz, group features, centre subtraction,
SharedMLP on fp32 MFMA, max over nsample}, eval mode, inputs resident in HBM.
One "step" = one such forward over the rank's clouds.  N > 1: one process per GPU
(torchrun), every rank works on its own clouds (weak scaling, no data-path collective);
the timed region is bracketed by barrier + synchronize and the max over ranks is used.

Other workloads (not the judged default): `--workload backbone_ops` = every sampling / grouping /
interpolation op of one PointTransformer_seg_T forward+backward at the configs[2] shapes (B=8 clouds
per GPU unless --clouds; dense layers excluded, see geot_amd/workloads.py); `--workload ntm` = the
unlabelled half of a FixMatch+NTM step (sig_t_mean, class transition, logit correction, 3-D loss).

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

N_POINTS = 24000
NPOINT, RADIUS, NSAMPLE, MLP = 6000, 0.1, 32, [3, 64, 64, 128]
FP32_VECTOR_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: peak FP32 vector (= fp32 MFMA) rate
FP32_MATRIX_PEAK_TFLOPS = 157.3   # dense fp32 MFMA peak (same figure)
HBM_PEAK_GBS = 8000.0
FPS_FLOP_PER_UPDATE = 10          # 3 sub, 3 mul, 2 add, min, compare


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--clouds", type=int, default=None, help="clouds per GPU per step (sa: 1, backbone_ops/ntm: 8)")
    ap.add_argument("--workload", choices=["sa", "backbone_ops", "ntm"], default="sa")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=24)
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams the steps are dealt to round-robin (default 1 = strictly one after the other; "
                         "2 lets step i+1's FPS, which occupies one CU per cloud, run beside step i's ball query + MLP)")
    return ap.parse_args()


def build_module(device):
    from geot_amd.pointnet2.pointnet2_modules import PointnetSAModuleVotes
    torch.manual_seed(1609)
    sa = PointnetSAModuleVotes(mlp=list(MLP), npoint=NPOINT, radius=RADIUS, nsample=NSAMPLE, use_xyz=True)
    return sa.to(device).eval()


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of THIS command
    (profiles/*_pmc_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc passes, FETCH
    doubled as MI355X_MICROARCH.md prescribes for gfx950).  Counters cannot be read from inside the
    process, so the figure is the profiled one, not a live one; None if no profile is committed."""
    import glob
    here = os.path.dirname(os.path.abspath(__file__))
    import re
    files = sorted(glob.glob(os.path.join(here, "profiles", "*_pmc_traffic.json")),
                   key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))])   # r01 ... v11 after v7
    if not files:
        return None
    with open(files[-1]) as f:
        prof = json.load(f)
    for name, rec in prof.get("kernels", {}).items():
        if kernel in name:
            return rec.get("traffic_bytes")
    return None


class FpsTimer:
    """HIP events around the dominant kernel (FPS) on the stream it is launched on
    (torch's current stream, which is where the C ABI launcher puts it)."""

    def __init__(self):
        self.pairs = []

    def wrap(self, fn):
        def timed(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = fn(*a, **k)
            e1.record()
            self.pairs.append((e0, e1))
            return out
        return timed

    def mean_ms(self):
        return float(np.mean([a.elapsed_time(b) for a, b in self.pairs])) if self.pairs else float("nan")


def cpu_baseline(xyz_np, feats_np, sa_cpu, steps):
    """The CPU port of the same step: oracle (C restatement, OpenMP over all host cores) for
    FPS / ball query / grouping + the same SharedMLP on torch-CPU.  Checker code, used here
    only as the reported baseline, never as the thing shipped."""
    from oracle import capi
    # the GPU box gives a one-GPU job a 16-core share; use what the affinity mask allows up to that
    cores = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    cores = capi.set_threads(cores)
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
                      "torch-CPU SharedMLP+max; FPS rounds are sequential so its share runs on 1 thread per cloud), %.1f s" % (steps, xyz_np.shape[0], N_POINTS, dt)}


def main():
    args = parse()
    from geot_amd import dist_utils
    assert torch.cuda.is_available(), "bench.py needs a GPU (there is no CPU fallback)"
    world, rank, local = dist_utils.env_world()
    # GEOT_BENCH_REHEARSAL=1: N ranks share GPU 0 and talk over gloo -- rehearses the N > 1 control flow
    # (rendezvous, barriers, MAX over ranks, rank-0 JSON) on a one-GPU box; never a measurement
    rehearsal = os.environ.get("GEOT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist_utils.init("gloo" if rehearsal else "nccl")

    from geot_amd import _lib, build as hip_build
    from geot_amd.synth import make_batch
    from geot_amd.pointnet2 import pointnet2_utils
    if not os.path.exists(hip_build.LIB) and rank == 0:      # fresh checkout: built artefacts are git-ignored
        hip_build.build()
    dist_utils.barrier()
    _lib.load()

    workload = args.workload
    B = args.clouds if args.clouds is not None else (1 if workload == "sa" else 8)
    xyz_np, _ = make_batch(B, N_POINTS, start_index=dist_utils.cloud_range(rank, B)[0])
    xyz = torch.from_numpy(xyz_np).to(dev)
    timer = FpsTimer()
    feats_np = sa = None
    if workload == "sa":
        feats_np = np.random.default_rng(1609 + rank).standard_normal((B, MLP[0], N_POINTS)).astype(np.float32)
        feats = torch.from_numpy(feats_np).to(dev)
        sa = build_module(dev)
        import geot_amd.pointnet2.pointnet2_modules as mods
        patch_owner, patch_name = mods.pointnet2_utils, "furthest_point_sample"
        fps_rounds, desc = NPOINT - 1, ("configs[1]: PointNet++ SetAbstraction fwd (FPS 24000->6000, ball_query r=0.1 "
                                        "ns=32, group, SharedMLP [6,64,64,128], max), eval")

        def step():
            with torch.no_grad():
                return sa(xyz, feats)[1]
    elif workload == "backbone_ops":
        from geot_amd import workloads as wl
        from geot_amd.pointops.functions import pointops as pops
        hot = wl.BackboneHotPath().to(dev)
        tokens = torch.randn(B, wl.TRANS_DIM, wl.GROUPS, device=dev)
        patch_owner, patch_name = pops, "furthestsampling_uniform"
        fps_rounds, desc = 8191, ("configs[2] hot-path ops only: PointTransformer_seg_T sampling/grouping/interpolation "
                                  "fwd+bwd (FPS 512+8192, kNN 32/4, three_nn+interpolate x3), dense layers excluded")

        def step():
            return wl.backbone_hotpath_step(hot, xyz, tokens)
    else:
        from geot_amd import workloads as wl
        from geot_amd import knn_cuda as kmod
        nt = wl.NtmHotPath().to(dev)
        from geot_amd.synth import make_logits
        # spatially coherent predictions (weak and strong view of the same regions), see synth.region_labels
        pw = torch.from_numpy(make_logits(xyz_np, index=2 * rank)).to(dev)
        ps = torch.from_numpy(make_logits(xyz_np, index=2 * rank + 1, sharp=3.0)).to(dev)
        patch_owner, patch_name = None, None
        fps_rounds, desc = 0, ("configs[4] NTM half-step: sig_t_mean + class transition + logit correction + "
                               "threeD_space_loss(k=32) fwd+bwd on B_u clouds")

        def step():
            return wl.ntm_step(nt, xyz, pw, ps)

    for _ in range(args.warmup):
        step()
    if patch_owner is not None:
        orig_fn = getattr(patch_owner, patch_name)
        setattr(patch_owner, patch_name, timer.wrap(orig_fn))
    mlp_timer = FpsTimer()
    if workload == "sa":
        import geot_amd.sa_fused as sa_fused_mod
        orig_mlp = sa_fused_mod.fused_group_mlp_max
        sa_fused_mod.fused_group_mlp_max = mlp_timer.wrap(orig_mlp)
    torch.cuda.synchronize()
    dist_utils.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if args.streams > 1:
        pool = [torch.cuda.Stream(device=dev) for _ in range(args.streams)]
        for i in range(args.steps):
            with torch.cuda.stream(pool[i % args.streams]):
                out = step()
    else:
        for _ in range(args.steps):
            out = step()
    torch.cuda.synchronize()
    dist_utils.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if patch_owner is not None:
        setattr(patch_owner, patch_name, orig_fn)
    if workload == "sa":
        sa_fused_mod.fused_group_mlp_max = orig_mlp
    elapsed = dist_utils.max_over_ranks(elapsed, "cpu" if rehearsal else dev)
    assert torch.isfinite(out).all()

    fps_ms = timer.mean_ms()
    fps_flop = B * N_POINTS * fps_rounds * FPS_FLOP_PER_UPDATE
    achieved = fps_flop / (fps_ms * 1e-3) / 1e12 if fps_rounds else float("nan")
    result = {
        "metric": "point-clouds/sec (24k pts, 17 classes)",
        "value": world * B * args.steps / elapsed,
        "unit": "clouds/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic" if not rehearsal else "synthetic (REHEARSAL: all ranks on one GPU over gloo -- not a measurement)",
        "config": {"workload": desc, "clouds_per_gpu": B, "points": N_POINTS,
                   "parallelism": "independent clouds per rank, no collective" +
                                  ("" if args.streams == 1 else "; consecutive steps overlap on %d HIP streams" % args.streams)},
        "roofline": {"kernel": "fps_pruned_kernel", "bound": "valu",
                     "achieved": achieved, "peak": FP32_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / FP32_VECTOR_PEAK_TFLOPS,
                     "traffic": pmc_traffic("fps_pruned_kernel") if workload == "sa" and B == 1 else None,
                     "avg_launch_ms": fps_ms,
                     # one cloud = one workgroup = one CU: the share of the chip a launch can reach is B / 256
                     "cus_used": B, "frac_of_cus_used": achieved / (FP32_VECTOR_PEAK_TFLOPS * min(B, 256) / 256.0),
                     "note": "FPS is fp32-VALU / round-latency bound, not HBM or MFMA bound; algorithmic "
                             "flop = clouds*N*(m-1) updates * 10 (what the reference executes); the pruned kernel "
                             "skips most of them exactly; one workgroup (one CU of 256) per cloud"},
    }
    if workload == "ntm":
        result["roofline"] = None
    if workload == "sa" and mlp_timer.pairs:
        mlp_ms = mlp_timer.mean_ms()
        widths = [MLP[0] + 3] + MLP[1:]   # use_xyz: 3 relative coordinates + 3 features
        mlp_flop = 2.0 * B * NPOINT * NSAMPLE * sum(a * b for a, b in zip(widths[:-1], widths[1:]))
        mlp_tf = mlp_flop / (mlp_ms * 1e-3) / 1e12
        result["roofline_secondary"] = {
            "kernel": "sa_group_mlp_max_kernel", "bound": "mfma", "achieved": mlp_tf, "peak": FP32_MATRIX_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": mlp_tf / FP32_MATRIX_PEAK_TFLOPS,
            "traffic": pmc_traffic("sa_group_mlp_max_kernel") if B == 1 else None, "avg_launch_ms": mlp_ms,
            "note": "fused group + [6,64,64,128] 1x1-conv stack on fp32 MFMA + max over nsample; algorithmic flop = "
                    "2*rows*sum(Cin*Cout) with the unpadded 3+3 input channels"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline and workload == "sa":
        sa_cpu = build_module("cpu")
        sa_cpu.load_state_dict(sa.state_dict())
        result["cpu_baseline"] = cpu_baseline(xyz_np, feats_np, sa_cpu, args.cpu_steps)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()                      # rank 0 may still be printing: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
