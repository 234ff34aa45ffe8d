"""Per-op timings at the BASELINE shapes (developer tool; bench.py is the judged harness)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geot_amd.synth import make_batch, make_logits, region_labels  # noqa: E402
from geot_amd.ext import pointnet2_ext as p2, pointops_cuda as pops  # noqa: E402
from geot_amd.knn_cuda import knn_sorted  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    B = int(os.environ.get("B", "1"))
    N = 24000
    xyz_np, _ = make_batch(B, N)
    xyz = torch.from_numpy(xyz_np).to(DEV)
    rows = []
    for m in (512, 6000, 8192):
        t = timeit(lambda: p2.furthest_point_sampling(xyz, m), iters=3, warm=1)
        rows.append(("fps K1 N=%d m=%d B=%d" % (N, m, B), t, "%.3f us/round" % (1e3 * t / (m - 1))))
    flat = xyz.reshape(-1, 3)
    off = (torch.arange(1, B + 1, device=DEV, dtype=torch.int32) * N)
    noff = (torch.arange(1, B + 1, device=DEV, dtype=torch.int32) * 8192)
    idx = torch.zeros(B * 8192, dtype=torch.int32, device=DEV)

    def k2():
        tmp = torch.full((B * N,), 1e10, device=DEV)
        pops.furthestsampling_cuda(B, N, flat, off, noff, tmp, idx)
    t = timeit(k2, iters=3, warm=1)
    rows.append(("fps K2 N=%d m=8192 B=%d" % (N, B), t, "%.3f us/round" % (1e3 * t / 8191)))
    c6000 = p2.furthest_point_sampling(xyz, 6000)
    new_xyz = p2.gather_points(xyz.transpose(1, 2).contiguous(), c6000).transpose(1, 2).contiguous()
    t = timeit(lambda: p2.ball_query(new_xyz, xyz, 0.1, 32))
    rows.append(("ball_query 6000x24000 r=.1 ns=32", t, "%.1f Gpair/s" % (B * 6000 * 24000 / t / 1e6)))
    bq = p2.ball_query(new_xyz, xyz, 0.1, 32)
    feats = torch.randn(B, 64, N, device=DEV)
    t = timeit(lambda: p2.group_points(feats, bq))
    byts = 4 * (64 * 6000 * 32 + 6000 * 32 + 64 * N) * B
    rows.append(("group_points C=64 6000x32", t, "%.1f GB/s" % (byts / t / 1e6)))
    go = torch.randn(B, 64, 6000, 32, device=DEV)
    t = timeit(lambda: p2.group_points_grad(go, bq, N))
    rows.append(("group_points_grad C=64", t, "%.1f GB/s" % (byts / t / 1e6)))
    known_idx = idx[:8192].long()
    known = xyz[:, known_idx % N].contiguous()
    t = timeit(lambda: p2.three_nn(xyz, known))
    rows.append(("three_nn 24000x8192", t, "%.1f Gpair/s" % (B * N * 8192 / t / 1e6)))
    d2, i3 = p2.three_nn(xyz, known)
    w = torch.rand(B, N, 3, device=DEV)
    f384 = torch.randn(B, 384, 8192, device=DEV)
    t = timeit(lambda: p2.three_interpolate(f384, i3, w))
    byts = (4 * (384 * N + 384 * 8192) + 24 * N) * B
    rows.append(("three_interpolate C=384 8192->24000", t, "%.1f GB/s" % (byts / t / 1e6)))
    g384 = torch.randn(B, 384, N, device=DEV)
    t = timeit(lambda: p2.three_interpolate_grad(g384, i3, w, 8192))
    rows.append(("three_interpolate_grad C=384", t, "%.1f GB/s" % (byts / t / 1e6)))
    c512 = xyz[:, :512].contiguous()
    t = timeit(lambda: knn_sorted(c512, xyz, 32))
    rows.append(("knn_sorted 512x24000 k=32", t, "%.1f Gpair/s" % (B * 512 * N / t / 1e6)))
    k8 = xyz[:, :8192].contiguous()
    t = timeit(lambda: knn_sorted(k8, k8, 4))
    rows.append(("knn_sorted 8192x8192 k=4", t, "%.1f Gpair/s" % (B * 8192 * 8192 / t / 1e6)))
    t = timeit(lambda: knn_sorted(xyz, xyz, 33), iters=3, warm=1)
    rows.append(("knn_sorted 24000x24000 k=33", t, "%.1f Gpair/s" % (B * N * N / t / 1e6)))
    from geot_amd import workloads as wl
    from geot_amd import ntm as ntm_mod
    tokens = torch.randn(B, 384, 512, device=DEV)
    hot = wl.BackboneHotPath().to(DEV)
    t = timeit(lambda: wl.backbone_hotpath_step(hot, xyz, tokens), iters=5, warm=2)
    rows.append(("backbone hot-path ops fwd+bwd B=%d" % B, t, "%.1f clouds/s" % (B / t * 1e3)))
    C = 17
    pw = torch.from_numpy(make_logits(xyz_np, 0)).to(DEV)            # spatially coherent predictions
    ps = torch.from_numpy(make_logits(xyz_np, 1, sharp=3.0)).to(DEV)
    nt = wl.NtmHotPath().to(DEV)
    t = timeit(lambda: wl.ntm_step(nt, xyz, pw, ps), iters=5, warm=2)
    rows.append(("NTM step (sig_t_mean+correct+3D loss) fwd+bwd B=%d" % B, t, "%.1f clouds/s" % (B / t * 1e3)))
    cm = torch.softmax(torch.randn(C, C, device=DEV), 1)
    prob = torch.softmax(ps, 1)
    with torch.no_grad():
        t = timeit(lambda: nt.predictor(prob, cm))
        rows.append(("sig_t_mean fwd", t, "%.1f GB/s (write)" % (B * N * C * C * 4 / t / 1e6)))
        insT = nt.predictor(prob, cm)
        t = timeit(lambda: ntm_mod.correct_logits(ps, insT, cm, 0.9))
        rows.append(("correct_logits fwd", t, "%.1f GB/s (read)" % (B * N * C * C * 4 / t / 1e6)))
        crit = ntm_mod.threeD_space_loss(k=32)
        nbr = crit.neighbours(xyz)
        lab = torch.from_numpy(region_labels(xyz_np)).to(DEV)
        t = timeit(lambda: crit(xyz, lab, insT, nbr))
        rows.append(("threeD_space_loss fwd (region labels)", t, "%.1f GB/s (rows)" % (B * N * 33 * C * C * 4 / t / 1e6)))
    for name, ms, extra in rows:
        print("%-42s %10.3f ms   %s" % (name, ms, extra), flush=True)


if __name__ == "__main__":
    main()
