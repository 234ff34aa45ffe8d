"""Prototype: the channels-first group_points gradient (B, C, npoint, nsample) -> (B, C, N) routed through the point-major row
gather -- transpose, reverse index, gather_rows_csr_cl, transpose back -- against geot's own entry point, piece by piece."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import geot_amd
import torch
from geot_amd.synth import make_batch
from geot_amd.ext import pointnet2_ext as p2
from geot_amd import fused_norm as fnm
B, N, C = 8, 24000, int(os.environ.get("C", "64"))
dev = "cuda"
xyz = torch.from_numpy(make_batch(B, N)[0]).to(dev)
c6000 = p2.furthest_point_sampling(xyz, 6000)
new_xyz = p2.gather_points(xyz.transpose(1, 2).contiguous(), c6000).transpose(1, 2).contiguous()
bq = p2.ball_query(new_xyz, xyz, 0.1, 32)                      # (B, 6000, 32) int32
go = torch.randn(B, C, 6000, 32, device=dev)
M = 6000 * 32


def timed(fn, k=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k):
        out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3, out


alg = 4 * B * (C * M + M + C * N)
t_ref, ref = timed(lambda: p2.group_points_grad(go, bq, N))
print("geot group_points_grad (entry point)   %8.1f us   %.2f TB/s" % (t_ref, alg / t_ref / 1e6))
idx = bq.reshape(B, M, 1).contiguous()
t1, g_cl = timed(lambda: go.view(B, C, M).transpose(1, 2).contiguous())
t2, rix = timed(lambda: fnm.ReverseIndex(idx, None, N, None))
t3, out_cl = timed(lambda: rix.gather(g_cl))
t4, out = timed(lambda: out_cl.transpose(1, 2).contiguous())
print("transpose in %8.1f | reverse index %8.1f | row gather %8.1f | transpose out %8.1f | total %8.1f us  %.2f TB/s"
      % (t1, t2, t3, t4, t1 + t2 + t3 + t4, alg / (t1 + t2 + t3 + t4) / 1e6))
print("max |difference| vs the entry point: %.3e (scale %.3e)" % (float((out - ref).abs().max()), float(ref.abs().max())))
